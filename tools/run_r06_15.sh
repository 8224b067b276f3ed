#!/bin/bash
# round 6, run 15: the whole GPU suite with the sub-pixel upsamplers, the A/B again (final kernel), bench
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
( time python -m pytest tests -q -m gpu --durations=10 ) > gpurun_out/r06/suite_15.log 2>&1
tail -22 gpurun_out/r06/suite_15.log
timeout 900 python tools/ab_up_sub.py > gpurun_out/r06_ab_up_sub.txt 2> gpurun_out/r06/ab_up_sub.err; cat gpurun_out/r06_ab_up_sub.txt
python bench.py --steps 3 --warmup 1 2> gpurun_out/r06/bench_15.err | tail -1 > gpurun_out/r06/bench_15.json; cut -c1-900 gpurun_out/r06/bench_15.json
