#!/bin/bash
# round 6, run 14: sub-pixel upsamplers ALSO in the split-stream forwards (up_fold = 2): what it costs the gate
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
CS_TUNE=up_fold=2 CS_SCHED_NS=4,8,12 CS_SCHED_KS=auto timeout 1500 python tools/parity_schedule.py > gpurun_out/r06/sched_upfold2.txt 2>gpurun_out/r06/sched_upfold2.err; cat gpurun_out/r06/sched_upfold2.txt; tail -3 gpurun_out/r06/sched_upfold2.err
CS_TUNE=up_fold=2 CS_SCHED_SEED=8 CS_SCHED_NS=4,8 CS_SCHED_KS=auto timeout 1500 python tools/parity_schedule.py > gpurun_out/r06/sched_upfold2_s8.txt 2>gpurun_out/r06/sched_upfold2_s8.err; cat gpurun_out/r06/sched_upfold2_s8.txt
