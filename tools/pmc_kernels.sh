# SQ counter passes (busy / wait / issue / LDS) over the conv, GEMM and attention micro-benchmarks -> gpurun_out/pmc_kernels.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sec in conv gemm attn; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" \
             "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_k/$sec/p$i -- python3 $R/tools/bench_ops.py $sec > $R/gpurun_out/pmc_k_${sec}_p$i.log 2>&1
  done
done
cd $R
for sec in conv gemm attn; do echo "== $sec"; python3 tools/pmc_summary.py gpurun_out/pmc_k/$sec | grep -A1 -E "conv3_halo|conv3_lw|gemm_big|gemm_w8|gemm_lw|attn_kernel|attn40_lw|igemm_kernel" | head -60; done > gpurun_out/pmc_kernels.txt
rm -rf gpurun_out/pmc_k gpurun_out/pmc_k_*.log
wc -l gpurun_out/pmc_kernels.txt
