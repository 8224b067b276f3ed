# halo conv schedule A/B on ONE box: per-layer micro-benchmark and the whole UNet forward for conv_sched = 0 / 1 / 2, twice
R=$GRAFT_REPO_ROOT
for v in 0 1 2 0 1 2; do echo "== conv_sched=$v"; CS_TUNE="conv_sched=$v" python3 $R/tools/bench_ops.py conv 2>&1 | grep -E "conv3x3"; CS_TUNE="conv_sched=$v" python3 $R/tools/bench_unet.py 5 2>&1 | grep -E "forward|conv3x3"; done
