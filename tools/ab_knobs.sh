# knob sweep of the whole UNet forward on ONE box (each setting twice, baseline in between)
R=$GRAFT_REPO_ROOT
for t in "" "attn_qt40=2" "" "conv_halo=4" "conv_halo=3" "" "gemm_big=3" "gemm_big=0" ""; do
  echo "== CS_TUNE='$t'"; CS_TUNE="$t" python3 $R/tools/bench_unet.py 5 2>&1 | grep -E "forward|conv3x3"; done
