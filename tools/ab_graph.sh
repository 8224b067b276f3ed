# hipGraph A/B of the headline bench on ONE box: eager, graph, eager, graph (steps 5 each, no extras / decode / cpu baseline)
R=$GRAFT_REPO_ROOT
for g in 0 1 0 1; do
  python3 $R/bench.py --steps 5 --warmup 2 --graph $g --decode 0 --extras 0 --no-cpu-baseline --profile-kernels 0 --ceilings 0 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('graph=$g', 'images/s', round(r['value'],2), 'ms_per_gen', round(r['ms_per_step'],2), 'fwd_ms', round(r['roofline']['launch_ms'],3))"
done
