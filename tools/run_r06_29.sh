#!/bin/bash
# round 6, run 29: the two-plane head behind the model-dtype output too: UNet tests, the rollout / pipeline gate
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_residual_x2_gpu.py -q -m gpu -x -s > gpurun_out/r06/head_29.log 2>&1; tail -4 gpurun_out/r06/head_29.log; grep "eps vs the fp32 oracle" gpurun_out/r06/head_29.log
timeout 2000 python -m pytest tests/test_parity_e2e_gpu.py -q -m gpu -k "rollout_function_and_the_pipeline_loop or eight_step or budget" -s > gpurun_out/r06/rollout_29.log 2>&1; tail -3 gpurun_out/r06/rollout_29.log; grep "step latents vs the fp32 oracle" gpurun_out/r06/rollout_29.log | cut -c1-330
