"""Per-workgroup timeline of gemm_big_kernel (debug bit 16384): where does a CU's time go between k loops?
Usage: python tools/gemm_timeline.py  (GPU).  Prints, per shape, the median phase lengths per workgroup and the per-CU gaps between
one workgroup's last stamp and the next workgroup's entry."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, ctypes as C
from consolver_amd import _lib as L

lib = L.lib()
EXTRA = int(os.environ.get("CS_TIMELINE_DEBUG", "0"))      # extra debug bits (e.g. the touch distance << 20) for the stamped launches
dev = torch.device("cuda:0")
shapes = [("qkv L0", 131072, 320, 960, False, False), ("out+res L0", 131072, 320, 320, True, False), ("ff1 geglu L0", 131072, 320, 2560, False, True),
          ("ff2+res L0", 131072, 1280, 320, True, False), ("qkv L1", 32768, 640, 1920, False, False),
          ("out+res L1 (one round)", 32768, 640, 640, True, False), ("to_q L1 (one round)", 32768, 640, 640, False, False)]
if os.environ.get("CS_TIMELINE_ONLY"): shapes = [sh for sh in shapes if os.environ["CS_TIMELINE_ONLY"] in sh[0]]
for tag, M, K, N, res, geglu in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.05
    b = torch.randn(N, device=dev, dtype=torch.float16)
    No = N // 2 if geglu else N
    out = torch.empty(M, No, device=dev, dtype=torch.float16)
    r = torch.randn(M, No, device=dev, dtype=torch.float16) if res else None
    st = L.stream_ptr()
    call = lambda: L.check(lib.cs_op_linear(L.ptr(x), M, K, L.ptr(w), L.ptr(b), N, L.ptr(r) if res else None, L.ptr(out), int(geglu), st))
    lib.cs_set_tuning(b"debug", 0)
    for _ in range(3): call()
    lib.cs_set_tuning(b"debug", 16384 | EXTRA)
    call(); torch.cuda.synchronize()
    call(); torch.cuda.synchronize()
    lib.cs_set_tuning(b"debug", 0)
    nwg = min(8192, (M // 256) * (N // 320))
    buf = np.zeros((8192, 12), dtype=np.uint64)
    L.check(lib.cs_debug_trace_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes))
    t = buf[:nwg, :5].astype(np.int64)
    hw = buf[:nwg, 5]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0                       # 100 MHz wall clock
    cu = ((hw >> np.uint64(32)) << np.uint64(16)) | ((hw & np.uint64(0xffffffff)) >> np.uint64(8) & np.uint64(0xff)) | (((hw & np.uint64(0xffffffff)) >> np.uint64(13) & np.uint64(7)) << np.uint64(8))
    ph = np.diff(us, axis=1)
    print(f"== {tag}  M={M} K={K} N={N}: {nwg} workgroups, {len(np.unique(cu))} distinct CU keys, kernel span {us.max():.1f} us")
    print("   median per workgroup, us: prologue %.2f | k loop %.2f | epilogue issue %.2f | store drain %.2f | total %.2f" %
          (np.median(ph[:, 0]), np.median(ph[:, 1]), np.median(ph[:, 2]), np.median(ph[:, 3]), np.median(us[:, 4] - us[:, 0])))
    # k loop split (wave 0 of each workgroup, shader-clock cycles converted with the workgroup's own wall-clock span)
    cyc = buf[:nwg, 6:12].astype(np.float64)
    span_cyc = cyc[:, 5] - cyc[:, 4]; span_us = us[:, 2] - us[:, 1]
    mhz = np.median(span_cyc / np.maximum(span_us, 1e-3))
    if mhz > 0:
        parts = np.median(cyc[:, 0:4], axis=0) / mhz
        print("   k loop split, us per workgroup (counter %.0f MHz): DMA issue %.2f | MFMAs + fragment reads %.2f | wait for the DMA %.2f | barrier %.2f ; steps %d"
              % (mhz, parts[0], parts[1], parts[2], parts[3], K // 64))
    else:
        print("   k loop split: n/a (the in-loop cycle stamps belonged to the round-2 non-pipelined loops, removed in round 3)")
    gaps, firsts = [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        o = idx[np.argsort(us[idx, 0])]
        firsts.append(us[o[0], 0])
        for a, b2 in zip(o[:-1], o[1:]):
            gaps.append(us[b2, 0] - us[a, 4])
    gaps = np.array(gaps if gaps else [0.0])
    print("   per-CU turnaround (previous workgroup drained -> next workgroup's entry), us: median %.2f  p10 %.2f  p90 %.2f ; first entry median %.2f us"
          % (np.median(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90), np.median(firsts)))
    # spread of phases across the chip: how synchronised are the CUs?  (std of the epilogue start times modulo the period, first round)
    first_round = np.argsort(us[:, 0])[:256]
    print("   first round: k loop ends at %.2f +- %.2f us, drained at %.2f +- %.2f us" %
          (us[first_round, 2].mean(), us[first_round, 2].std(), us[first_round, 4].mean(), us[first_round, 4].std()))
