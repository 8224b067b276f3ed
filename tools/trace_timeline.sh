# per-dispatch TIMELINE of one UNet forward in launch order: start offset, duration, gap to the previous dispatch, kernel, grid -> gpurun_out/timeline_unet.txt
# (which level / block the time goes to; tools/trace_unet.sh aggregates the same trace by kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_tl -- python3 $R/tools/bench_unet.py 1 > $R/gpurun_out/trace_tl.log 2>&1
cd $R
python3 - <<'PY' > gpurun_out/timeline_unet.txt
import csv, glob, re
f = glob.glob("gpurun_out/trace_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sinusoid" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
fw = rows[a:b]
t0 = int(fw[0]["Start_Timestamp"]); prev_end = t0
tot = 0.0; gaps = 0.0
for r in fw:
    name = r["Kernel_Name"]
    mm = re.search(r"(conv3_lw_kernel<[^>]*>|conv3_halo_kernel<[^>]*>|gemm_w8_kernel<[^>]*>|gemm_lw_kernel<[^>]*>|gemm_big_kernel<[^>]*>|igemm_kernel<[^>]*>|splitk_reduce_kernel|xattn64_kernel<[^>]*>|xattn_block_kernel<[^>]*>|attn40_lw_kernel|attn_kernel[A-Za-z0-9_]*|gn_[a-z_]*kernel|[a-z0-9_]*_kernel)", name)
    short = mm.group(1) if mm else name[:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3; g = (s - prev_end) / 1e3
    tot += d; gaps += max(g, 0.0)
    print(f"{(s - t0) / 1e3:9.1f} us  dur {d:8.1f}  gap {g:6.1f}  {short:44s} grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))}")
    prev_end = max(prev_end, e)
print(f"# {len(fw)} dispatches, kernel time {tot / 1e3:.3f} ms, gaps {gaps / 1e3:.3f} ms, span {(prev_end - t0) / 1e6:.3f} ms")
PY
rm -rf gpurun_out/trace_tl
tail -1 gpurun_out/timeline_unet.txt
