#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
timeout 2000 python -m pytest tests/test_parity_e2e_gpu.py -q -m gpu -k "rollout_function_and_the_pipeline_loop and (2 or 3)" -s > gpurun_out/r06/rollout_26.log 2>&1; tail -5 gpurun_out/r06/rollout_26.log; grep "step latents vs the fp32 oracle" gpurun_out/r06/rollout_26.log | cut -c1-330
