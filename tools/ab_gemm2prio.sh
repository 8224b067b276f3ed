R=$GRAFT_REPO_ROOT
for v in 0 1 2 0 1 2; do echo "== gemm2_prio=$v"; CS_TUNE="gemm2_prio=$v" python3 $R/tools/bench_ops.py gemm2 2>&1 | grep -E "gemm2"; done
