set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "ddim_baseline or reference_class or rccl or bench_prints or solver_state" 2>&1 | tail -15 > gpurun_out/r06/t1.log
python -m pytest tests/test_parity_e2e_gpu.py -x -q -m gpu -s -k "gate_on_the_rollout or gate_holds" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06/t2.log
python -m pytest tests/test_unet_gpu.py tests/test_solver_gpu.py tests/test_diffusers_dropin.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06/t3.log
bash tools/ab_unet_lib.sh r05 a27 > gpurun_out/r06/ab0.log 2>&1
python tools/flux_full_size_parity.py > gpurun_out/r06_flux_full_size_parity.txt 2> gpurun_out/r06/flux_full.err
bash tools/profile_flux.sh r06 > gpurun_out/r06/profile_flux.log 2>&1
