#!/bin/bash
# round 6, run 16: sub-pixel upsamplers in the VAE decoder: op cases at 128-column tiles, the VAE tests, decode time A/B
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_vae_gpu.py -q -m gpu -k "subpixel or vae or decoder or decode" -s -x > gpurun_out/r06/upsub_vae_test.log 2>&1; tail -12 gpurun_out/r06/upsub_vae_test.log; grep "fused-upsample\|full vae" gpurun_out/r06/upsub_vae_test.log
for k in 0 1 0 1; do CS_TUNE=up_fold=$k timeout 300 python tools/bench_vae.py 2>/dev/null | sed "s/^/up_fold=$k  /"; done | tee gpurun_out/r06_ab_vae_up_sub.txt
