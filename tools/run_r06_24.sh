#!/bin/bash
# round 6, run 24: the schedule sweep at the step counts the suite does not hold (2, 3, 5, 6, 15) with the round's final kernels
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
CS_SCHED_NS=2,3,5,6,15 CS_SCHED_KS=auto timeout 1700 python tools/parity_schedule.py > gpurun_out/r06/sched_small_n.txt 2>/dev/null; cat gpurun_out/r06/sched_small_n.txt
CS_SCHED_SEED=8 CS_SCHED_NS=4,8 CS_SCHED_KS=auto timeout 1700 python tools/parity_schedule.py > gpurun_out/r06/sched_final_s8.txt 2>/dev/null; cat gpurun_out/r06/sched_final_s8.txt
