set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_ln_fold_gpu.py -x -q -m gpu -s -k calibration 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06/t13.log
( time python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k bench_prints 2>&1 | tail -3 ) > gpurun_out/r06/t14.log 2>&1
