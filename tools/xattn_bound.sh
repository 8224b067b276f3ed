R=$GRAFT_REPO_ROOT
for d in 0 1 2 4 6 7 8 16 31; do echo "== debug=$d"; CS_TUNE="debug=$d" python3 $R/tools/bench_ops.py xattn 2>&1 | grep -E "xattn_block|4 kernels"; done
