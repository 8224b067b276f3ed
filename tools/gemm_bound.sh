# gemm_big: full / k loop only (debug 1: no epilogue) / epilogue only (debug 2: no k loop) on the UNet's linear shapes
R=$GRAFT_REPO_ROOT
for v in 0 1 2; do echo "== gemm debug=$v"; CS_TUNE="debug=$v" python3 $R/tools/bench_ops.py gemm 2>&1 | grep -E "linear"; done
