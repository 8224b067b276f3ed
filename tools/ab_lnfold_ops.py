"""Where the folded LayerNorm costs GEMM time: per SD1.5 shape (batch 32), producer with / without row statistics and consumer linear vs linear_ln."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import ops
dev = "cuda:0"
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).half()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
print(f"{'shape':34s} {'plain us':>9s} {'folded us':>10s} {'delta':>7s}")
for tag, M, C in (("L0", 131072, 320), ("L1", 32768, 640), ("L2", 8192, 1280)):
    x, wo, bo = rnd(M, C), rnd(C, C, scale=C ** -0.5), rnd(C)
    r32 = torch.randn(M, C, device=dev) * 2
    rh, rl = ops.split_f16(r32)
    for mode, lo in (("f16", None), ("f16x2", rl)):
        oh, ol = torch.empty_like(rh), (torch.empty_like(rh) if lo is not None else None)
        t0 = timeit(lambda: ops.linear_x2(x, wo, bo, res=rh, res_lo=lo, want_lo=lo is not None, out=oh, out_lo=ol))
        t1 = timeit(lambda: ops.linear_x2(x, wo, bo, res=rh, res_lo=lo, want_lo=lo is not None, out=oh, out_lo=ol, row_stats=True))
        print(f"{'to_out+res ' + tag + ' ' + mode + ' (+row stats)':34s} {t0:9.1f} {t1:10.1f} {t1 - t0:+7.1f}")
        t0 = timeit(lambda: ops.linear_x2(x, wo, bo, want_lo=lo is not None, out=oh, out_lo=ol))
        t1 = timeit(lambda: ops.linear_x2(x, wo, bo, want_lo=lo is not None, out=oh, out_lo=ol, row_stats=True))
        print(f"{'proj_in ' + tag + ' ' + mode + ' (+row stats)':34s} {t0:9.1f} {t1:10.1f} {t1 - t0:+7.1f}")
    gam, bet = rnd(C) * 0.1 + 1, rnd(C, scale=0.1)
    st = ops.row_stats(rh, rl)
    for name, N, geglu in (("qkv", 3 * C, False), ("to_q", C, False), ("ff1 geglu", 8 * C, True)):
        w, b = rnd(N, C, scale=C ** -0.5), (rnd(N) if geglu else None)
        if geglu:
            wp, bp = ops.geglu_pack(w, b); w, b = wp.to(dev), bp.to(dev)
        wf, sf, bf = (t.to(dev) for t in ops.ln_fold_pack(w, b, gam, bet))
        t0 = timeit(lambda: ops.linear(rh, w, b, geglu=geglu))
        t1 = timeit(lambda: ops.linear_ln(rh, wf, sf, bf, st, 1, geglu=geglu))
        tl = timeit(lambda: ops.layer_norm(rh, gam, bet))
        print(f"{name + ' ' + tag + ' (LN kernel ' + format(tl, '.0f') + ' us)':34s} {t0:9.1f} {t1:10.1f} {t1 - t0:+7.1f}")
