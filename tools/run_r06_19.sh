#!/bin/bash
# round 6, run 19: FLUX precision emulation with the softmax-probability, output-head and split-stream rounding points
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
timeout 1500 python tools/sim_precision_flux.py > gpurun_out/r06/sim_flux.txt 2> gpurun_out/r06/sim_flux.err; cat gpurun_out/r06/sim_flux.txt; tail -3 gpurun_out/r06/sim_flux.err
