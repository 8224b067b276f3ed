# per-dispatch kernel trace of the VAE decoder (B = 1, 4, 16 passes of tools/bench_vae.py) -> slowest dispatch groups of the last B = 16 decode
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_vae -- python3 $R/tools/bench_vae.py > $R/gpurun_out/trace_vae.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/trace_vae/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pixel_linear" in r["Kernel_Name"]]
fw = rows[idx[-1]:]
agg = collections.OrderedDict()
for r in fw:
    name = r["Kernel_Name"]
    mm = re.search(r"(conv3_halo_kernel<[^>]*>|conv3_lw_kernel<[^>]*>|gemm_w8_kernel<[^>]*>|gemm_lw_kernel|gemm_big_kernel<[^>]*>|igemm_kernel<[^>]*>|splitk_reduce_kernel)", name)
    short = mm.group(1) if mm else re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)[:36]
    key = (short, r.get("Grid_Size_X", r.get("Grid_Size", "")))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"one decode (B=16): {len(fw)} dispatches, {tot/1e3:.2f} ms of kernel time")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:28]:
    print(f"{k[0]:42s} grid={k[1]:>10s} n={len(v):3d} total={sum(v)/1e3:7.3f} ms  each={sum(v)/len(v):8.1f} us")
PY
rm -rf gpurun_out/trace_vae
