# per-kernel stats of the FLUX-Kontext edit (reduced depth by default: FLUX_LAYERS=4 FLUX_SINGLES=8) -> gpurun_out/flux_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export FLUX_LAYERS=${FLUX_LAYERS:-4} FLUX_SINGLES=${FLUX_SINGLES:-8}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_flux -- python3 $R/tools/bench_flux.py > $R/gpurun_out/trace_flux.log 2>&1
cd $R
tail -2 gpurun_out/trace_flux.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_flux/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
ours = sum(int(r["TotalDurationNs"]) for r in rows if "anonymous namespace" in r["Name"] or "_GLOBAL__N_" in r["Name"])
print(f"kernel time, all: {tot/1e6:.1f} ms; library kernels only: {ours/1e6:.1f} ms (2 edits x 8 forwards in the trace: warm-up + timed)")
for r in rows[:16]:
    print(f'{r["Name"][:90]:90s} n={int(r["Calls"]):5d} total={int(r["TotalDurationNs"])/1e6:9.2f} ms avg={float(r["AverageNs"])/1e3:9.1f} us {100*int(r["TotalDurationNs"])/tot:5.1f}%')
PY
rm -rf gpurun_out/trace_flux
