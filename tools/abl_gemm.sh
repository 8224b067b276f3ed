for d in 0 2 3 4; do echo "== prefetch $d"; CS_TUNE="gemm_med=0,prefetch=$d" timeout 120 python tools/bench_ops.py gemm 2>&1 | grep -E "L0|L1|L2"; done
echo "== prefetch 3 kloop only";  CS_TUNE="gemm_med=0,prefetch=3,debug=1" timeout 120 python tools/bench_ops.py gemm 2>&1 | grep -E "L0|L1|L2"
