# SQ counter passes (busy / wait / LDS) over the three upsamplers in both forms (tools/bench_ops.py upsub) -> gpurun_out/r06_pmc_up_sub.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_us/p$i -- python3 $R/tools/bench_ops.py upsub > $R/gpurun_out/pmc_us_p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_us | grep -A1 -E "conv3_lw" > gpurun_out/r06_pmc_up_sub.txt
tail -12 gpurun_out/pmc_us_p1.log >> gpurun_out/r06_pmc_up_sub.txt
rm -rf gpurun_out/pmc_us gpurun_out/pmc_us_p*.log
cat gpurun_out/r06_pmc_up_sub.txt | cut -c1-260
