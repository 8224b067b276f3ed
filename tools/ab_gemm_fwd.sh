# forward-level A/B of the GEMM variants on ONE box: images/s, ms per UNet forward, GEMM class ms
R=$GRAFT_REPO_ROOT
for cfg in "gemm_pers=0,gemm_ring=0" "gemm_pers=0,gemm_ring=1" "gemm_pers=1" "gemm_pers=0,gemm_ring=0" "gemm_pers=0,gemm_ring=1" "gemm_pers=1"; do
  echo "== $cfg"; CS_TUNE="$cfg" python3 $R/bench.py --steps 3 --warmup 1 --extras 0 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['roofline']['launch_ms'],3), d['roofline_kernels']['gemm_1x1_linear'])"
done
