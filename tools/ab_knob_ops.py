"""A/B of one kernel-selection knob (argv[1], default epi_fast) at the op level: bit comparison and time of value 0 against 1 per SD1.5
linear / conv shape (batch 32), every epilogue role."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import ops
dev = "cuda:0"
KNOB = sys.argv[1] if len(sys.argv) > 1 else "epi_fast"
torch.manual_seed(0)
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).half()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
def ab(name, fn):
    res, tm = [], []
    for v in (0, 1):
        ops.set_tuning(KNOB, v)
        flat = []
        def walk(o):
            if isinstance(o, (tuple, list)):
                for q in o: walk(q)
            elif torch.is_tensor(o): flat.append(o.clone())
        walk(fn())
        res.append(flat)
    for rep in range(2):
        for v in (0, 1):
            ops.set_tuning(KNOB, v)
            t = timeit(fn)
            if rep: tm.append(t)
    same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    print(f"{name:44s} {tm[0]:9.1f} {tm[1]:9.1f} {100 * (tm[1] / tm[0] - 1):+6.1f} %   {'bit-identical' if same else 'DIFFERENT'}", flush=True)
    return same
print(f"{'shape':44s} {KNOB + '=0 us':>9s} {'=1 us':>9s}")
ok = True
for tag, M, C in (("L0", 131072, 320), ("L1", 32768, 640), ("L2", 8192, 1280)):
    x, wo, bo = rnd(M, C), rnd(C, C, scale=C ** -0.5), rnd(C)
    r32 = torch.randn(M, C, device=dev) * 2
    rh, rl = ops.split_f16(r32)
    if C % 320 == 0:
        for mode, lo in (("f16", None), ("f16x2", rl)):
            ok &= ab(f"to_out+res {tag} {mode} (+row stats)", lambda: ops.linear_x2(x, wo, bo, res=rh, res_lo=lo, want_lo=lo is not None, row_stats=True))
            ok &= ab(f"proj_in {tag} {mode}", lambda: ops.linear_x2(x, wo, bo, want_lo=lo is not None))
    gam, bet = rnd(C) * 0.1 + 1, rnd(C, scale=0.1)
    st = ops.row_stats(rh, rl)
    for name, N, geglu in (("qkv", 3 * C, False), ("ff1 geglu", 8 * C, True)):
        w, b = rnd(N, C, scale=C ** -0.5), (rnd(N) if geglu else None)
        if geglu:
            wp, bp = ops.geglu_pack(w, b); w, b = wp.to(dev), bp.to(dev)
        wf, sf, bf = (t.to(dev) for t in ops.ln_fold_pack(w, b, gam, bet))
        ok &= ab(f"{name} {tag} plain", lambda: ops.linear(rh, w, b, geglu=geglu))
        ok &= ab(f"{name} {tag} folded LN", lambda: ops.linear_ln(rh, wf, sf, bf, st, 1, geglu=geglu))
    xf, w2, b2 = rnd(M, 4 * C), rnd(C, 4 * C, scale=(4 * C) ** -0.5), rnd(C)
    for mode, lo in (("f16", None), ("f16x2", rl)):
        ok &= ab(f"ff2+res {tag} {mode}", lambda: ops.linear_x2(xf, w2, b2, res=rh, res_lo=lo, want_lo=lo is not None))
# ragged M (rows past M clamped), a tile count that is no multiple of the grid
x, w, b = rnd(131072 - 77, 320), rnd(960, 320, scale=320 ** -0.5), rnd(960)
ok &= ab("qkv ragged M = 130995", lambda: ops.linear(x, w, b))
# 3x3 convs (conv3_lw_kernel): + time embedding (resnet conv1), + residual (conv2), both stream modes; the 1x1 shortcut on a concatenated input
for tag, B, H, C in (("64x64 320", 32, 64, 320), ("32x32 640", 32, 32, 640), ("16x16 1280", 32, 16, 1280), ("8x8 1280", 32, 8, 1280)):
    x = rnd(B, H, H, C)
    w = ops.pack_conv_weight(torch.randn(C, C, 3, 3) * (9 * C) ** -0.5).to(dev)
    b, temb = rnd(C), rnd(B, C)
    r32 = torch.randn(B, H, H, C, device=dev) * 2
    rh, rl = ops.split_f16(r32)
    ok &= ab(f"conv3x3 {tag} + temb", lambda: ops.conv2d(x, w, b, temb=temb))
    ok &= ab(f"conv3x3 {tag} + res f16", lambda: ops.conv2d(x, w, b, res=rh))
    ok &= ab(f"conv3x3 {tag} + res f16x2", lambda: ops.conv2d_x2(x, w, b, res=rh, res_lo=rl))
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
