"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel name (mean per dispatch)."""
import csv, sys, collections, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"][:70]
        agg[(name, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, grid), cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = max(len(v) for v in cs.values())
    print(f"{name} grid={grid} n={n}")
    print("   " + "  ".join(f"{k}={sum(v)/len(v):.3g}" for k, v in sorted(cs.items())))
