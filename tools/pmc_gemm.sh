# PMC passes over the GEMM micro-benchmark (separate passes; no tracing flags)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
           "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU" \
           "SQ_WAVE_CYCLES TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  CS_TUNE="$1" rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_gemm/p$i -- python3 $R/tools/bench_ops.py gemm > $R/gpurun_out/pmc_gemm_p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_gemm | grep -A1 -E "gemm_big|gemm_med" 
