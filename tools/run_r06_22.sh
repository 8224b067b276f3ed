#!/bin/bash
# round 6, run 22: the UNet's output head on two planes: tests, the tightest gated numbers, forward time A/B
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_ops_gpu.py -q -m gpu -k "output_head or conv_out or group_norm" -s -x > gpurun_out/r06/head_22.log 2>&1; tail -8 gpurun_out/r06/head_22.log; grep "hi + lo head" gpurun_out/r06/head_22.log
CS_SCHED_NS=4,8,12 CS_SCHED_KS=auto timeout 1500 python tools/parity_schedule.py > gpurun_out/r06/sched_head2.txt 2>/dev/null; cat gpurun_out/r06/sched_head2.txt
for k in 0 1 0 1; do CS_TUNE=head_x2=$k CS_OUT_F32=1 timeout 300 python tools/bench_unet.py 10 2>/dev/null | head -1 | sed "s/^/head_x2=$k  /"; done
