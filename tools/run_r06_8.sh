set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time python -m pytest tests -q -m gpu --durations=15 -s 2>&1 | grep -v "^$" > gpurun_out/r06/suite_b_full.log ) 2> gpurun_out/r06/suite_b_time.log
grep -E "trajectory|latents vs the fp32|final latents|forward:|passed|failed|DC-heavy|smoke|reduced UNet|plain protocol|full-depth|flux 1 \+ 1|rel l2" gpurun_out/r06/suite_b_full.log | cut -c1-400 > gpurun_out/r06/suite_b.log
tail -25 gpurun_out/r06/suite_b_full.log >> gpurun_out/r06/suite_b.log
for i in 1 2 3; do
  echo "== r05 tree"; python tools/ab/r05/tools/bench_unet.py 10 2>&1 | tail -2
  echo "== r06 tree"; python tools/bench_unet.py 10 2>&1 | tail -2
done > gpurun_out/r06/ab_forward_final.log 2>&1
