# staggered k32 GEMM (gemm_stag_kernel) vs gemm_big_kernel<.,320> on ONE box: per shape and whole forward, twice
R=$GRAFT_REPO_ROOT
for v in 0 1 0 1; do echo "== gemm_stag=$v"; CS_TUNE="gemm_stag=$v" python3 $R/tools/bench_ops.py gemm 2>&1 | grep -E "linear"; CS_TUNE="gemm_stag=$v" python3 $R/tools/bench_unet.py 5 2>&1 | grep -E "forward|gemm"; done
