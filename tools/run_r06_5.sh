set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_residual_x2_gpu.py tests/test_ln_fold_gpu.py tests/test_unet_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r06/t12.log
for i in 1 2 3; do
  echo "== r05 tree"; python tools/ab/r05/tools/bench_unet.py 10 2>&1 | tail -2
  echo "== r06 tree"; python tools/bench_unet.py 10 2>&1 | tail -2
done > gpurun_out/r06/ab_forward.log 2>&1
for cfg in "" "cfg_copy_async=0" "" "cfg_copy_async=0"; do echo "== CS_TUNE=$cfg"; CS_TUNE="$cfg" python tools/bench_unet.py 10 2>&1 | tail -2; done > gpurun_out/r06/ab_copy_async.log 2>&1
echo "== r05 f16"; CS_RESIDUAL=f16 python tools/ab/r05/tools/bench_unet.py 10 2>&1 | tail -2 >> gpurun_out/r06/ab_forward.log
echo "== r06 f16"; CS_RESIDUAL=f16 python tools/bench_unet.py 10 2>&1 | tail -2 >> gpurun_out/r06/ab_forward.log
bash tools/trace_timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline_unet.txt gpurun_out/r06/timeline_unet.txt
