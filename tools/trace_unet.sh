# per-dispatch kernel trace of a few UNet forwards -> table of the slowest GEMM-class dispatches of ONE forward
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_unet -- python3 $R/tools/bench_unet.py 1 > $R/gpurun_out/trace_unet.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/trace_unet/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward = the profiling pass; take the last N dispatches between two sinusoid kernels
idx = [i for i, r in enumerate(rows) if "sinusoid" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
fw = rows[a:b]
agg = collections.OrderedDict()
for r in fw:
    name = r["Kernel_Name"]
    import re
    mm = re.search(r"(conv3_lw_kernel<[^>]*>|conv3_halo_kernel<[^>]*>|gemm_w8_kernel<[^>]*>|gemm_big_kernel<[^>]*>|igemm_kernel<[^>]*>|splitk_reduce_kernel)", name)
    short = mm.group(1) if mm else re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)[:36]
    key = (short, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Workgroup_Size_X", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
tot = sum(sum(v) for v in agg.values())
print(f"one forward: {len(fw)} dispatches, {tot/1e3:.2f} ms of kernel time")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print(f"{k[0]:42s} grid={k[1]:>9s} n={len(v):3d} total={sum(v)/1e3:7.3f} ms  each={sum(v)/len(v):8.1f} us")
PY
rm -rf gpurun_out/trace_unet
