# attn40_lw_kernel on ONE box, alternating: default (16x16x16 second k step) | 16x16x32 for both k steps (attn_lw=2) | attn_kernel (attn_lw=0) | static priority (attn_prio=1)
for i in 1 2 3; do for t in "" "attn_lw=2" "attn_lw=0" "attn_prio=1"; do echo "== CS_TUNE=$t"; CS_TUNE="$t" python tools/bench_ops.py attn 2>&1 | grep "dh=40 self"; done; done
