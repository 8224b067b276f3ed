set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_parity_e2e_gpu.py tests/test_engine_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | grep -E "trajectory|latents vs|passed|failed|Error|^E " | cut -c1-360 > gpurun_out/r06/t15.log
python bench.py --steps 3 --warmup 1 2> gpurun_out/r06/bench_b.err | tail -1 > gpurun_out/r06/bench_b.json
