#!/bin/bash
# round 6, run 13: the precision schedule with the sub-pixel upsamplers in the one-plane forwards (up_fold = 1, default) against up_fold = 0
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
CS_SCHED_NS=4,8,12,15 CS_SCHED_KS=auto timeout 1500 python tools/parity_schedule.py > gpurun_out/r06/sched_upfold1.txt 2>gpurun_out/r06/sched_upfold1.err; cat gpurun_out/r06/sched_upfold1.txt; tail -3 gpurun_out/r06/sched_upfold1.err
