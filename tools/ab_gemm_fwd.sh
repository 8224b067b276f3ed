# forward-level A/B of the GEMM k-loop variants and tile order on ONE box: images/s, ms per UNet forward, GEMM class ms
R=$GRAFT_REPO_ROOT
for cfg in "gemm_ring=1,gemm_gm=1" "gemm_ring=2,gemm_gm=1" "gemm_ring=2" "gemm_ring=1,gemm_gm=1" "gemm_ring=2,gemm_gm=1" "gemm_ring=2"; do
  echo "== $cfg"; CS_TUNE="$cfg" python3 $R/bench.py --steps 3 --warmup 1 --extras 0 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['roofline']['launch_ms'],3), d['roofline_kernels']['gemm_1x1_linear'])"
done
