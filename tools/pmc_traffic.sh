#!/bin/bash
# HBM-side traffic per KERNEL against its algorithmic bytes.  On the GPU box (gpurun): bash tools/pmc_traffic.sh r04
#   (a) the UNet forward, summed per kernel name (both residual-stream modes), next to the executor's own algorithmic bytes per kernel class;
#   (b) every SD1.5 layer shape in isolation (tools/bench_ops.py gemm | conv | attn | norm): per shape the kernel that served it, FETCH / WRITE per launch,
#       the algorithmic bytes of the shape and the ratio.
# Counters in separate passes (FETCH_SIZE, WRITE_SIZE), no tracing in the same run; program directly after `--`.  Corrections as MI355X_MICROARCH.md
# (HBM section): FETCH_SIZE counts 128-B requests at 64 B -> x2; WRITE_SIZE exact for 16-B/lane stores; unit KB.
set -u
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in f16 f16x2; do
  export CS_RESIDUAL=$mode
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/fwd_${mode}_$c -- python3 $R/tools/bench_unet.py 2 > $OUT/fwd_${mode}_$c.log 2>&1
  done
  CS_PROFILE_JSON=$OUT/fwd_${mode}_classes.json python3 $R/tools/bench_unet.py 2 > $OUT/fwd_${mode}_plain.log 2>&1
done
unset CS_RESIDUAL
for sec in gemm conv attn norm; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/ops_${sec}_$c -- python3 $R/tools/bench_ops.py $sec > $OUT/ops_${sec}_$c.log 2>&1
  done
  CS_SHAPES_JSON=$OUT/ops_${sec}_shapes.json python3 $R/tools/bench_ops.py $sec > $OUT/ops_${sec}_plain.log 2>&1
done
cd $R
python3 tools/pmc_traffic_by_kernel.py $OUT > $R/gpurun_out/${TAG}_pmc_traffic_by_kernel.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +20M -delete
tail -5 $R/gpurun_out/${TAG}_pmc_traffic_by_kernel.txt
