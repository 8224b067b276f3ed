#!/bin/bash
# round 6, run 18: the default schedule with the sub-pixel upsamplers on a second weight seed and at batch 2; the power sampler's card match
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
CS_SCHED_SEED=8 CS_SCHED_NS=4,8,12 CS_SCHED_KS=auto timeout 1500 python tools/parity_schedule.py > gpurun_out/r06/sched_upfold1_s8.txt 2>/dev/null; cat gpurun_out/r06/sched_upfold1_s8.txt
CS_SCHED_B=2 CS_SCHED_NS=8 CS_SCHED_KS=auto timeout 1500 python tools/parity_schedule.py > gpurun_out/r06/sched_upfold1_b2.txt 2>/dev/null; cat gpurun_out/r06/sched_upfold1_b2.txt
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --extras 0 --traffic none 2>gpurun_out/r06/bench_18.err | tail -1 > gpurun_out/r06/bench_18.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_18.json"))
print(d["value"], d["roofline"]["frac"], d["power_over_timed_region"])
PY
ls /sys/class/drm/ | head -30
