"""Where a step of conv3_lw_kernel goes (debug bit 16384): cycles per k step of compute wave 0, its share at the step barrier and in the
lgkmcnt(0) in front of it, the in-kernel shader clock, and what loader wave 4 spends waiting for its DMA / at the barriers.
Usage: python tools/conv_lw_trace.py  (GPU)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, ctypes as C
from consolver_amd import _lib as L, ops

lib = L.lib()
dev = torch.device("cuda:0")
B = 32
for tag, H, cin, cout, up in [("L0 320->320", 64, 320, 320, False), ("L0 960->320", 64, 960, 320, False), ("L1 640->640", 32, 640, 640, False),
                              ("L1 1920->640", 32, 1920, 640, False), ("L2 1280->1280", 16, 1280, 1280, False), ("L2 2560->1280", 16, 2560, 1280, False),
                              ("up 640->640 @64", 32, 640, 640, True)]:
    x = (torch.randn(B, H, H, cin, device=dev)).half()
    w = ops.pack_conv_weight((torch.randn(cout, cin, 3, 3, device=dev) * (9 * cin) ** -0.5).half())
    b = torch.randn(cout, device=dev).half()
    call = lambda: ops.conv2d(x, w, b, upsample=up, splitk=False)
    lib.cs_set_tuning(b"debug", 0)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    lib.cs_set_tuning(b"debug", 16384)
    call(); torch.cuda.synchronize()
    call(); torch.cuda.synchronize()
    lib.cs_set_tuning(b"debug", 0)
    buf = np.zeros((8192, 12), dtype=np.uint64)
    L.check(lib.cs_debug_trace_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes))
    Ho = 2 * H if up else H
    nwg = min(8192, (B * Ho * Ho // 256) * (cout // 160))
    t = buf[:nwg].astype(np.float64)
    steps = t[:, 5]
    ok = steps > 0
    t = t[ok]; steps = steps[ok]
    cyc, bar, ticks, lwait, lbar, lgkm = t[:, 6], t[:, 7], t[:, 8], t[:, 9], t[:, 10], t[:, 11]
    med = lambda v: float(np.median(v))
    print(f"== {tag}: {ms:.3f} ms ({2.0 * B * Ho * Ho * 9 * cin * cout / ms / 1e12:.0f} TFLOP/s), {len(t)} workgroups x {int(med(steps))} steps | per step, compute wave 0: "
          f"{med(cyc / steps):7.1f} cycles (MFMA floor 1280), barrier {med(bar / steps):6.1f}, lgkmcnt(0) {med(lgkm / steps):6.1f} | clock {med(cyc / ticks) * 0.1:.2f} GHz | "
          f"k loop {med(ticks) / 100:.1f} us per tile | loader wave 4: DMA wait {med(lwait / steps):6.1f}, barrier {med(lbar / steps):7.1f} cycles per step")
    # the workgroup's phases (100 MHz stamps) and the CU's turnaround to its next workgroup
    raw = buf[:nwg][ok]
    ent, ks, ke, ex = (raw[:, i].astype(np.float64) for i in range(4))
    print(f"      phases, us: entry -> k loop {med(ks - ent) / 100:.2f} | k loop {med(ke - ks) / 100:.2f} | epilogue until the stores have left {med(ex - ke) / 100:.2f} | workgroup {med(ex - ent) / 100:.2f} | "
          f"launch: first entry -> last exit {(ex.max() - ent.min()) / 100:.1f} us")
    cu = {}
    for r in raw:
        cu.setdefault((((int(r[4]) >> 32) & 0xF) << 8) | ((int(r[4]) >> 8) & 0xFF), []).append((int(r[0]), int(r[3])))
    gaps, per = [], []
    for lst in cu.values():
        lst.sort(); per.append(len(lst))
        gaps += [(lst[i + 1][0] - lst[i][1]) / 100 for i in range(len(lst) - 1)]
    xcc = (raw[:, 4].astype(np.uint64) >> np.uint64(32)).astype(np.int64) & 0xF
    dur = (ex - ent) / 100
    kl = (ke - ks) / 100
    print("      per XCD: mean workgroup us " + " ".join(f"{dur[xcc == x].mean():.1f}" for x in range(8)) + " | mean k loop us " + " ".join(f"{kl[xcc == x].mean():.1f}" for x in range(8))
          + " | last exit us " + " ".join(f"{(ex[xcc == x].max() - ent.min()) / 100:.0f}" for x in range(8)))
    if gaps:
        print(f"      {len(cu)} CUs, workgroups per CU min / mean / max {min(per)} / {np.mean(per):.1f} / {max(per)}; exit -> next entry on the same CU: median {np.median(gaps):.2f} us, mean {np.mean(gaps):.2f} us; "
              f"entry of a CU's first workgroup: spread {(max(l[0][0] for l in cu.values()) - ent.min()) / 100:.1f} us; exit of its last: spread {(ex.max() - min(l[-1][1] for l in cu.values())) / 100:.1f} us")
