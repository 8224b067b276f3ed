# usage: ab_unet.sh "tuneA" "tuneB"   (alternating, same box)
for i in 1 2; do for t in "$1" "$2"; do echo "== CS_TUNE=$t"; CS_TUNE="$t" python tools/bench_unet.py 10 2>&1 | tail -2; done; done
