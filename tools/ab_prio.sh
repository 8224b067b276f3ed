# static-priority experiments on ONE box: attention (attn_prio 0/1) and gemm_big (debug bit 12), twice
R=$GRAFT_REPO_ROOT
for v in 0 1 0 1; do echo "== attn_prio=$v"; CS_TUNE="attn_prio=$v" python3 $R/tools/bench_ops.py attn 2>&1 | grep -E "attention"; done
for v in 0 4096 0 4096; do echo "== gemm debug=$v"; CS_TUNE="debug=$v" python3 $R/tools/bench_ops.py gemm 2>&1 | grep -E "linear"; done
