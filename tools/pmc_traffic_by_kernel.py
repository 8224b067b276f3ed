"""Joins rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE dispatch rows with algorithmic bytes (tools/pmc_traffic.sh writes the inputs).

Part (a): one UNet forward per residual mode, measured bytes per kernel name, classes next to the executor's own algorithmic bytes per class
          (cs_unet_profile_entry `bytes`: operands read once + outputs written once).
Part (b): isolated layer shapes (tools/bench_ops.py): dispatch rows are cut into shapes by their periodic (kernel, grid) pattern -- every shape is
          called 12 times back to back -- and set against the shape's algorithmic bytes.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import OrderedDict, defaultdict

out_dir = sys.argv[1]
FWD_PER_RUN = 7          # tools/bench_unet.py 2: 3 warm-up + 2 timed + 1 profiled + ... (counted from the dispatches below)


def rows_of(d):
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in fs:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")), float(r["Counter_Value"])))
    rows.sort()
    demangle([r[1] for r in rows])
    return rows


_DEMANGLED = {}


def demangle(names):
    """rocprofv3 leaves some kernel names mangled (_ZN12_GLOBAL__N_1<len><name>I<template args>E...; binutils' c++filt does not know the _Float16 code
    DF16_): pull the function name and the template argument codes out by hand"""
    for n in names:
        if n in _DEMANGLED or not n.startswith("_Z"):
            continue
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
        if not m:
            continue
        ln = int(m.group(1))
        st = m.end()
        ident = n[st:st + ln]
        rest = n[st + ln:]
        targs = ""
        if rest.startswith("I"):
            codes = re.findall(r"L([ib])(\d+)E|(DF16_)", rest[:rest.find("EEv") + 1] if "EEv" in rest else rest[:40])
            targs = "<" + ", ".join(("f16" if c[2] else (c[1] if c[0] == "i" else ("true" if c[1] == "1" else "false"))) for c in codes) + ">"
        _DEMANGLED[n] = ident + targs


def short(name):
    name = _DEMANGLED.get(name, name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:64]


def klass(name):
    n = short(name)
    if n.startswith(("conv3_lw", "conv3_halo")) or re.match(r"igemm_kernel<\d+, ?true", n):
        return "conv3x3_igemm"
    if n.startswith(("gemm_w8", "gemm_lw", "gemm_big", "igemm_kernel", "splitk_reduce")):
        return "gemm_1x1_linear"
    if n.startswith(("attn", "xattn")) or "attn" in n[:24]:
        return "attention (self + cross)"
    if n.startswith("gn_"):
        return "groupnorm_silu"
    if n.startswith("ln_"):
        return "layernorm"
    return "misc"


print("# HBM-side traffic per kernel vs algorithmic bytes (FETCH_SIZE x2 + WRITE_SIZE, KB -> bytes; separate PMC passes)\n")
# ---------------------------------------------------------------- (a) whole forward
for mode in ("f16", "f16x2"):
    f, w = rows_of(os.path.join(out_dir, f"fwd_{mode}_FETCH_SIZE")), rows_of(os.path.join(out_dir, f"fwd_{mode}_WRITE_SIZE"))
    if not f or not w:
        print(f"(no forward data for residual mode {mode})\n"); continue
    # the forwards of the run: conv_in runs exactly once per forward (conv_in_kernel, or latent_to_nhwc64_kernel when it is served by the MFMA conv)
    nfwd = sum(1 for r in f if short(r[1]).startswith(("conv_in_kernel", "latent_to_nhwc64_kernel")))
    if nfwd == 0:
        print(f"(no conv_in dispatch found for residual mode {mode}: cannot count forwards)\n"); continue
    agg = OrderedDict()
    for rows, col in ((f, 0), (w, 1)):
        for _, name, grid, v in rows:
            a = agg.setdefault(short(name), [0.0, 0.0, 0])
            a[col] += v * 1024 * (2 if col == 0 else 1)
            if col == 0:
                a[2] += 1
    cls_meas = defaultdict(float)
    tot = 0.0
    print(f"## (a) UNet forward, effective batch 32, residual stream {mode}: per kernel name, per forward ({nfwd} forwards in the run)")
    print(f"{'kernel':66s} {'launches':>8s} {'fetch GB':>9s} {'write GB':>9s} {'total GB':>9s}")
    for name, (fb, wb, n) in sorted(agg.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
        if (fb + wb) / nfwd < 5e6:
            continue
        print(f"{name:66s} {n / nfwd:8.1f} {fb / nfwd / 1e9:9.3f} {wb / nfwd / 1e9:9.3f} {(fb + wb) / nfwd / 1e9:9.3f}")
    for name, (fb, wb, n) in agg.items():
        cls_meas[klass(name)] += (fb + wb) / nfwd
        tot += (fb + wb) / nfwd
    alg = {}
    try:
        alg = json.load(open(os.path.join(out_dir, f"fwd_{mode}_classes.json")))
    except Exception:
        pass
    alg_m = defaultdict(float)
    for k, v in alg.items():
        alg_m["attention (self + cross)" if k.startswith("attention") else k] += v["bytes"]
    print(f"\n{'class':28s} {'measured GB':>12s} {'algorithmic GB':>15s} {'ratio':>7s}")
    for k, v in sorted(cls_meas.items(), key=lambda kv: -kv[1]):
        a = alg_m.get(k, 0.0)
        print(f"{k:28s} {v / 1e9:12.2f} {a / 1e9:15.2f} {(v / a if a else float('nan')):7.2f}")
    print(f"{'TOTAL':28s} {tot / 1e9:12.2f} {sum(alg_m.values()) / 1e9:15.2f} {tot / max(sum(alg_m.values()), 1):7.2f}\n")

# ---------------------------------------------------------------- (b) isolated shapes
CALLS = 12
for sec in ("gemm", "conv", "attn", "norm"):
    f, w = rows_of(os.path.join(out_dir, f"ops_{sec}_FETCH_SIZE")), rows_of(os.path.join(out_dir, f"ops_{sec}_WRITE_SIZE"))
    try:
        shapes = json.load(open(os.path.join(out_dir, f"ops_{sec}_shapes.json")))
    except Exception:
        shapes = []
    if not f or not w or not shapes:
        print(f"(no isolated-shape data for section {sec})\n"); continue
    # drop everything that is not one of the library's kernels (torch's randn / copies while the operands are built)
    ours = lambda r: not re.search(r"at::native|distribution|elementwise|vectorized|rocprim|fillFunctor|Cijk|memcpy|copy", r[1])
    f, w = [r for r in f if ours(r)], [r for r in w if ours(r)]
    print(f"## (b) isolated SD1.5 shapes, section {sec}: per launch (mean of {CALLS} back-to-back calls)")
    print(f"{'shape':44s} {'kernel(s)':50s} {'fetch MB':>9s} {'write MB':>9s} {'alg MB':>8s} {'ratio':>6s}")
    p = 0
    for sh in shapes:
        kpc = None
        for cand in (1, 2, 3, 4):
            seg = f[p:p + CALLS * cand]
            if len(seg) < CALLS * cand:
                break
            key = [(r[1], r[2]) for r in seg]
            if all(key[i] == key[i % cand] for i in range(len(key))):
                kpc = cand
                break
        if kpc is None:
            print(f"{sh['tag']:44s} (dispatch pattern not recognised at row {p}; stopping this section)")
            break
        segf, segw = f[p:p + CALLS * kpc], w[p:p + CALLS * kpc]
        p += CALLS * kpc
        fb = sum(r[3] for r in segf) * 1024 * 2 / CALLS
        wb = sum(r[3] for r in segw) * 1024 / CALLS
        names = " + ".join(short(r[1])[:46] for r in segf[:kpc])
        print(f"{sh['tag']:44s} {names:50s} {fb / 1e6:9.1f} {wb / 1e6:9.1f} {sh['alg_bytes'] / 1e6:8.1f} {(fb + wb) / sh['alg_bytes']:6.2f}")
    print()
