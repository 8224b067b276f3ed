# fused cross-attention sub-block A/B on ONE box
R=$GRAFT_REPO_ROOT
for v in 0 1 0 1; do echo "== xattn_fused=$v"; CS_TUNE="xattn_fused=$v" python3 $R/tools/bench_unet.py 5 2>&1 | grep -E "forward|conv3x3"; done
