# hand-placed fragment-read ring in gemm_big_kernel (gemm_ring=1) against the compiler's read order, ONE box
R=$GRAFT_REPO_ROOT
for v in 0 1 0 1; do echo "== gemm_ring=$v"; CS_TUNE="gemm_pers=0,gemm_ring=$v" python3 $R/tools/bench_ops.py gemm 2>&1 | grep -E "^linear"; done
for v in 0 1; do echo "== k loop only, gemm_ring=$v"; CS_TUNE="gemm_pers=0,gemm_ring=$v,debug=1" python3 $R/tools/bench_ops.py gemm 2>&1 | grep -E "^linear"; done
for v in 0 1; do echo "== k loop only without staging, gemm_ring=$v"; CS_TUNE="gemm_pers=0,gemm_ring=$v,debug=32769" python3 $R/tools/bench_ops.py gemm 2>&1 | grep -E "^linear"; done
