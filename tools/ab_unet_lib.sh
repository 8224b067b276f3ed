# usage: ab_unet_lib.sh libA libB   (alternating, same box; CS_RESIDUAL selects the stream mode)
for i in 1 2; do for v in $1 $2; do echo "== $v"; CONSOLVER_HIP_LIB=$PWD/tools/ab/lib_$v.so python tools/bench_unet.py 10 2>&1 | tail -2; done; done
