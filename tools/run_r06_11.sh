#!/bin/bash
# round 6, run 11: the LayerNorm-fold detector at op level; the precision schedule on a second weight seed and at batch 2
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
python -m pytest tests/test_ln_fold_gpu.py -q -m gpu -s > gpurun_out/r06/lnfold_11.log 2>&1; tail -3 gpurun_out/r06/lnfold_11.log
grep -A1 "^dc\|^outl" gpurun_out/r06/lnfold_11.log | head -40
CS_SCHED_SEED=8 CS_SCHED_NS=4,8,12 CS_SCHED_KS=auto timeout 1500 python tools/parity_schedule.py > gpurun_out/r06/sched_seed8.txt 2>gpurun_out/r06/sched_seed8.err; cat gpurun_out/r06/sched_seed8.txt
CS_SCHED_B=2 CS_SCHED_NS=8 CS_SCHED_KS=auto timeout 1500 python tools/parity_schedule.py > gpurun_out/r06/sched_b2.txt 2>gpurun_out/r06/sched_b2.err; cat gpurun_out/r06/sched_b2.txt
