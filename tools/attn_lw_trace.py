"""Where a 64-key tile of attn40_lw_kernel goes (debug bit 16384): shader cycles per tile of compute wave 0 over the steady loop, the in-kernel
shader clock, and what loader wave 4 spends waiting for its DMA / at the tile barriers.  Usage: python tools/attn_lw_trace.py  (GPU)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, ctypes as C
from consolver_amd import _lib as L, ops

lib = L.lib()
dev = torch.device("cuda:0")
for B, N in [(32, 4096), (16, 4096), (32, 1024)]:
    H, dh = 8, 40
    qkv = torch.randn(B, N, 3 * H * dh, device=dev).half()
    q, k, v = qkv[..., :320], qkv[..., 320:640], qkv[..., 640:]
    out = torch.empty(B, N, 320, dtype=torch.float16, device=dev)
    call = lambda: L.check(lib.cs_op_attention(q.data_ptr(), 960, k.data_ptr(), 960, v.data_ptr(), 960, out.data_ptr(), 320, B, H, N, N, dh, dh ** -0.5, L.stream_ptr(dev)))
    for lw in (1, 0):
        lib.cs_set_tuning(b"attn_lw", lw)
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"B={B} N={N} attn_lw={lw}: {ms:.3f} ms ({4.0 * B * H * N * N * dh / ms / 1e9:.0f} TFLOP/s algorithmic)")
    lib.cs_set_tuning(b"attn_lw", 1)
    for dbg, tag in [(0, "full"), (1, "no exp / cvt"), (2, "no MFMA")]:
        lib.cs_set_tuning(b"debug", 16384 | dbg)
        call(); torch.cuda.synchronize()
        e0.record()
        for _ in range(5): call()
        e1.record(); torch.cuda.synchronize()
        print(f"   TRACE instantiation ({tag}): {e0.elapsed_time(e1) / 5:.3f} ms")
        lib.cs_set_tuning(b"debug", 0)
        buf = np.zeros((4096, 12), dtype=np.uint64)
        L.check(lib.cs_debug_attn_trace_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes))
        nwg = min(4096, B * H * N // 256)
        t = buf[:nwg].astype(np.float64)
        t = t[t[:, 2] > 0]
        med = lambda x: float(np.median(x))
        tiles = t[:, 2]
        print(f"   trace ({tag}): {len(t)} workgroups x {int(med(tiles))} tiles | compute wave 0: {med(t[:, 0] / tiles):7.1f} cycles per 64-key tile (56 MFMAs = 896) | clock {med(t[:, 0] / t[:, 1]) * 0.1:.2f} GHz | "
              f"{med(t[:, 1] / tiles) * 10:.0f} ns per tile | loader wave 4: DMA wait {med(t[:, 3] / tiles):6.1f}, barrier {med(t[:, 4] / tiles):7.1f} cycles per tile")
        if dbg == 0:
            # per-workgroup phases (us) and the gap between consecutive workgroups on one CU
            print(f"      phases, us: entry -> first tiles landed {med(t[:, 5]) / 100:.2f} | tile 0 {med(t[:, 6]) / 100:.2f} | steady loop {med(t[:, 1]) / 100:.2f} | drain + epilogue {med(t[:, 7]) / 100:.2f} | "
                  f"whole workgroup {med(t[:, 9] - t[:, 8]) / 100:.2f}")
            ids = buf[:nwg][buf[:nwg, 2] > 0]
            cu = {}
            for r in ids:
                cu.setdefault((((int(r[10]) >> 32) & 0xF) << 8) | ((int(r[10]) >> 8) & 0xFF), []).append((int(r[8]), int(r[9])))
            print(f"      first entry -> last exit: {(float(ids[:, 9].max()) - float(ids[:, 8].min())) / 100:.1f} us")
            gaps = []
            for k2, lst in cu.items():
                lst.sort()
                gaps += [(lst[i + 1][0] - lst[i][1]) / 100 for i in range(len(lst) - 1)]
            if gaps:
                print(f"      {len(cu)} hardware ids, {np.mean([len(v) for v in cu.values()]):.1f} workgroups each; exit -> next entry on the same id: median {np.median(gaps):.2f} us, mean {np.mean(gaps):.2f} us")
