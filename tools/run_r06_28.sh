#!/bin/bash
# round 6, run 28: the bench's timed loop eager vs one hipGraph per generation, alternating on one box
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
for g in 0 1 0 1; do
  python bench.py --steps 5 --warmup 2 --graph $g --no-cpu-baseline --extras 0 --traffic none --ceilings 0 --profile-kernels 0 --decode 0 2>gpurun_out/r06/bench_28.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('graph=$g', round(d['value'],2), 'images/s  ms_per_step', round(d['ms_per_step'],2), ' forward', round(r['launch_ms'],3) if r.get('launch_ms') else None, ' frac', round(r['frac'],4))"
done
tail -3 gpurun_out/r06/bench_28.err
