# CFG shared-prefix A/B on ONE box: bench_unet (forward at effective batch 32) with the knob off / on, alternating
R=$GRAFT_REPO_ROOT
for v in 0 1 0 1; do echo "== cfg_share=$v"; CS_TUNE="cfg_share=$v" python3 $R/tools/bench_unet.py 5 2>&1 | grep -E "forward|ms" | head -3; done
