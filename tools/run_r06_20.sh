#!/bin/bash
# round 6, run 20: FLUX with the output head on hi + lo planes: the FLUX tests (numbers), then the edit time
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
timeout 1700 python -m pytest tests/test_flux_gpu.py -q -m gpu -s > gpurun_out/r06/flux_20.log 2>&1; tail -6 gpurun_out/r06/flux_20.log
grep -i "rel l2\|fp32 output\|final latents\|rollout\|reduced flux\|full-width" gpurun_out/r06/flux_20.log | cut -c1-300
timeout 600 python tools/bench_flux.py 2>/dev/null | tail -4
