# idle time BETWEEN the kernels of one UNet forward (end of dispatch i -> start of dispatch i + 1 on the stream), from a rocprofv3 kernel trace:
# what launch latency / dependent-dispatch turnaround costs, and between which kernels.  On the GPU box: CS_RESIDUAL=f16x2 bash tools/trace_gaps.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_gaps -- python3 $R/tools/bench_unet.py 1 > $R/gpurun_out/trace_gaps.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/trace_gaps/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sinusoid" in r["Kernel_Name"]]
fw = rows[idx[-2]:idx[-1]]
def short(n):
    m = re.search(r"(conv3_lw_kernel|gemm_w8_kernel|gemm_lw_kernel|igemm_kernel|splitk_reduce_kernel|attn40_lw_kernel|attn_kernel|xattn_block_kernel|gn_apply_kernel|gn_finalize_kernel|gn_stats_kernel|rowvec_linear_kernel|row_stats_kernel|conv_out_patch_kernel|latent_to_nhwc64_kernel|copyBuffer|fillBuffer)", n)
    return m.group(1) if m else n[:30]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fw)
span = int(fw[-1]["End_Timestamp"]) - int(fw[0]["Start_Timestamp"])
gaps = collections.defaultdict(list)
for a, b in zip(fw, fw[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(g)
tot = sum(sum(v) for v in gaps.values())
print(f"one forward: {len(fw)} dispatches, span {span/1e6:.3f} ms, kernels busy {busy/1e6:.3f} ms, idle between kernels {tot/1e6:.3f} ms ({100*tot/span:.1f} %)")
neg = sum(min(0, g) for v in gaps.values() for g in v)
print(f"(overlap of consecutive dispatches, negative gaps: {neg/1e6:.3f} ms)")
print(f"{'after -> before':60s} {'n':>4s} {'total us':>9s} {'mean us':>8s}")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"{k[0] + ' -> ' + k[1]:60s} {len(v):4d} {sum(v)/1e3:9.1f} {sum(v)/len(v)/1e3:8.2f}")
PY
rm -rf gpurun_out/trace_gaps
