# attn40_lw_kernel experiments on ONE box: static priority for the compute waves, conv3_lw_kernel: priority for its compute waves
for i in 1 2; do for t in "" "attn_prio=1" "attn_lw=0"; do echo "== CS_TUNE=$t"; CS_TUNE="$t" python tools/bench_ops.py attn 2>&1 | grep "dh=40 self"; done; done
for i in 1 2; do for t in "" "debug=4096"; do echo "== CS_TUNE=$t"; CS_TUNE="$t" python tools/bench_ops.py conv 2>&1 | grep -E "conv3x3" | head -8; done; done
