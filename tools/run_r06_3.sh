set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_residual_x2_gpu.py tests/test_unet_gpu.py tests/test_solver_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06/t7.log
python -m pytest tests/test_engine_gpu.py tests/test_diffusers_dropin.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06/t8.log
python -m pytest tests/test_parity_e2e_gpu.py -x -q -m gpu -s -k "eight_step or gate_holds or gate_on" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r06/t9.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke.log 2>&1
python tools/smoke_variants.py > gpurun_out/r06/smoke_variants.log 2>&1
