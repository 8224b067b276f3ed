# usage: ab_run.sh libA libB <bench_ops section> <grep pattern> [CS_TUNE]   (alternating, same box)
for v in $1 $2 $1 $2; do echo "== $v"; CS_TUNE="${5:-}" CONSOLVER_HIP_LIB=$PWD/tools/ab/lib_$v.so python tools/bench_ops.py $3 2>&1 | grep -E "$4"; done
