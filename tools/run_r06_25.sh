#!/bin/bash
# round 6, run 25: the rollout function with the denoiser's fp32 output next to its fp32 state
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
timeout 2000 python -m pytest tests/test_parity_e2e_gpu.py tests/test_solver_gpu.py tests/test_ppo_gpu.py -q -m gpu -k "rollout or ppo or denoise" -s > gpurun_out/r06/rollout_25.log 2>&1; tail -5 gpurun_out/r06/rollout_25.log; grep "step latents vs the fp32 oracle" gpurun_out/r06/rollout_25.log | cut -c1-330
