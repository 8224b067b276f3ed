#!/bin/bash
# Round profile: bench JSON, rocprofv3 kernel-trace stats of the same command, and two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) over tools/bench_unet.py for the HBM traffic per UNet forward.
# usage (on the GPU box, via gpurun): bash tools/profile_round.sh r01_d
set -u
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 3 --warmup 1 2>$OUT/bench.err | tail -1 > $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --profile-kernels 0 --extras 0 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/bench_unet.py 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/bench_unet.py 2 > $OUT/pmc_write.log 2>&1
cd $R
python3 - <<PY
import csv, glob, json
out = {}
for name in ("fetch", "write"):
    tot = 0.0; n = 0
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            tot += float(r["Counter_Value"]); n += 1
    out[name] = {"sum_counter": tot, "dispatches": n}
json.dump(out, open("$OUT/pmc_totals.json", "w"), indent=1)
print(out)
# per-forward HBM-side traffic with the gfx950 corrections of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-B requests at 64 B -> x2;
# WRITE_SIZE exact for 16-B/lane stores; both in KB.  tools/bench_unet.py 2 runs 6 forwards (3 warm-up + 2 timed + 1 profiled).
fw = 6
rec = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, no tracing) over 'python3 tools/bench_unet.py 2' = 6 UNet "
                 "forwards at effective batch 32 (3 warm-up + 2 timed + 1 profiled); tools/profile_round.sh $TAG",
       "forwards": fw, "FETCH_SIZE_KB_sum": out["fetch"]["sum_counter"], "WRITE_SIZE_KB_sum": out["write"]["sum_counter"],
       "dispatches": out["fetch"]["dispatches"],
       "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for 16-B/lane stores; units KB (x1024)",
       "fetch_bytes_per_forward": out["fetch"]["sum_counter"] * 1024 * 2 / fw, "write_bytes_per_forward": out["write"]["sum_counter"] * 1024 / fw}
rec["traffic_bytes_per_forward"] = rec["fetch_bytes_per_forward"] + rec["write_bytes_per_forward"]
json.dump(rec, open("$OUT/pmc_traffic.json", "w"), indent=1)
PY
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv | cut -c1-200
rm -rf $OUT/trace/*/*kernel_trace.csv $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/bench.json | cut -c1-600
