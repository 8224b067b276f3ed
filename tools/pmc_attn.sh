# PMC passes over the attention micro-benchmark (separate passes; no tracing flags): where do the wave cycles of attn_kernel go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_attn/p$i -- python3 $R/tools/bench_ops.py attn > $R/gpurun_out/pmc_attn_p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_attn | grep -A1 -E "attn_kernel" | head -40
rm -rf gpurun_out/pmc_attn
