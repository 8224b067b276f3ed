# what bounds the halo conv kernel (results wrong, timing only).  bits: 4 no weight staging, 8 no halo staging, 16 no MFMAs,
# 32 fragments read once, 64 no per-step barrier, 128 no stagger (read -> multiply in every wave), 256 multiply -> read in every wave
R=$GRAFT_REPO_ROOT
for d in ${1:-0 12 28 44 76 140 268 204}; do echo "== debug=$d"; CS_TUNE="debug=$d" python3 $R/tools/bench_ops.py conv 2>&1 | grep -E "conv3x3 s1 +L0 |conv3x3 s1 +L1 |conv3x3 s1 +L2 "; done
