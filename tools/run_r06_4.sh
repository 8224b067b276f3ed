set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_residual_x2_gpu.py tests/test_ln_fold_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r06/t10.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke2.log 2>&1
python tools/parity_knobs_n4.py > gpurun_out/r06/parity_knobs_n4.log 2>&1
python -m pytest tests/test_flux_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06/t11.log
python bench.py --steps 3 --warmup 1 2> gpurun_out/r06/bench_a.err | tail -1 > gpurun_out/r06/bench_a.json
