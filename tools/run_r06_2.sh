set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_ops_gpu.py tests/test_residual_x2_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r06/t4.log
python -m pytest tests/test_unet_gpu.py tests/test_ln_fold_gpu.py tests/test_vae_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06/t5.log
python -m pytest tests/test_parity_e2e_gpu.py -x -q -m gpu -s -k "eight_step or gate_holds" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06/t6.log
R05="lo8=0,epi_fast=1,x2_sc_skip=0,conv_out_mfma=0"
for i in 1 2; do
  for cfg in "$R05" "" "lo8=0" "lo8=0,epi_fast=1" "x2_sc_skip=0" "conv_out_mfma=0"; do
    echo "== CS_TUNE=$cfg"; CS_TUNE="$cfg" python tools/bench_unet.py 10 2>&1 | tail -2
  done
done > gpurun_out/r06/ab1.log 2>&1
