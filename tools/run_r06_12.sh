#!/bin/bash
# round 6, run 12: the sub-pixel upsamplers: op test, per-launch and forward A/B
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "subpixel or upsample" -s -x > gpurun_out/r06/upsub_test.log 2>&1; tail -25 gpurun_out/r06/upsub_test.log
timeout 900 python tools/ab_up_sub.py > gpurun_out/r06_ab_up_sub.txt 2> gpurun_out/r06/ab_up_sub.err; cat gpurun_out/r06_ab_up_sub.txt; tail -5 gpurun_out/r06/ab_up_sub.err
