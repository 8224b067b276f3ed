#!/bin/bash
# round 6, run 21: configs[3] at its real size against the streamed fp32 oracle, with the output head on hi + lo planes
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
timeout 2400 python tools/flux_full_size_parity.py > gpurun_out/r06_flux_full_size_parity_b.txt 2> gpurun_out/r06/flux_full_b.err; cat gpurun_out/r06_flux_full_size_parity_b.txt; tail -3 gpurun_out/r06/flux_full_b.err
