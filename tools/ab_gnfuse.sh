# GroupNorm statistics from the producers' epilogues (gn_fuse=1) against a statistics pass per GroupNorm, ONE box
R=$GRAFT_REPO_ROOT
for cfg in "gn_fuse=0" "gn_fuse=1" "gn_fuse=0" "gn_fuse=1"; do
  echo "== $cfg"; CS_TUNE="$cfg" python3 $R/bench.py --steps 3 --warmup 1 --extras 0 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline_kernels']; print(round(d['value'],2), round(d['roofline']['launch_ms'],3), 'gn', k['groupnorm_silu'], 'conv', k['conv3x3_igemm']['ms'], 'gemm', k['gemm_1x1_linear']['ms'])"
done
