# experiment: half the CUs start their first GEMM tile late (gemm_big_kernel debug bit 8192, count of s_sleep(127) in bits 16..)
R=$GRAFT_REPO_ROOT
for n in 0 1 2 3 4 6; do v=$(( n == 0 ? 0 : 8192 + n * 65536 )); echo "== sleeps=$n debug=$v"; CS_TUNE="debug=$v" python3 $R/tools/bench_ops.py gemm 2>&1 | grep -E "^linear"; done
