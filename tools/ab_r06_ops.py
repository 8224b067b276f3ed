"""Per-launch A/B of the round-6 epilogue forms on the SD1.5 transformer shapes (batch 32), op level, one box:
  * the feed-forward's second linear in the split mode (residual + its lo plane, NO lo plane out): generic epilogue (epi_fast = 1: what round 5 ran) vs FAST 4, fp16 and byte lo planes;
  * to_out (residual + lo in, lo out, row statistics): fp16 vs byte lo planes;
  * proj_in (lo plane out, row statistics): fp16 vs byte;
  * the fused cross-attention block: fp16 vs byte lo planes;
  * conv_out (UNet eps head, VAE image head): v_dot2 patch kernel vs the MFMA kernel.
python tools/ab_r06_ops.py > gpurun_out/r06_ab_ops.txt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import ops
dev = "cuda:0"
def rnd(*s, scale=1.0, dt=torch.float16): return (torch.randn(*s, device=dev) * scale).to(dt)
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
print(f"{'shape':58s} {'us':>8s}")
for lvl, (M, C) in enumerate(((131072, 320), (32768, 640), (8192, 1280))):
    x4 = rnd(M, 4 * C); w2 = rnd(C, 4 * C, scale=(4 * C) ** -0.5); b = rnd(C, scale=0.1)
    r32 = rnd(M, C, scale=2.0, dt=torch.float32)
    rh, rl = ops.split_f16(r32); _, rl8 = ops.lo8_split(r32)
    ops.set_tuning("epi_fast", 1)
    t_gen = timeit(lambda: ops.linear_x2(x4, w2, b, res=rh, res_lo=rl, want_lo=False)); o_gen = ops.linear_x2(x4, w2, b, res=rh, res_lo=rl, want_lo=False)[0]
    ops.set_tuning("epi_fast", 3)
    t_f4 = timeit(lambda: ops.linear_x2(x4, w2, b, res=rh, res_lo=rl, want_lo=False)); o_f4 = ops.linear_x2(x4, w2, b, res=rh, res_lo=rl, want_lo=False)[0]
    t_f48 = timeit(lambda: ops.linear_lo8(x4, w2, b, res=rh, res_lo8=rl8, want_lo=False))
    t_pl = timeit(lambda: ops.linear(x4, w2, b, res=rh))
    print(f"ff2 + res L{lvl} [{M} x {4 * C} -> {C}]: generic (round 5) {t_gen:8.1f} | FAST 4 fp16 lo {t_f4:8.1f} ({'bit-identical' if torch.equal(o_gen, o_f4) else 'DIFFERENT'}) | FAST 4 byte lo {t_f48:8.1f} | one-plane stream {t_pl:8.1f}")
    a = rnd(M, C); wo = rnd(C, C, scale=C ** -0.5)
    t16 = timeit(lambda: ops.linear_x2(a, wo, b, res=rh, res_lo=rl, row_stats=True))
    t8 = timeit(lambda: ops.linear_lo8(a, wo, b, res=rh, res_lo8=rl8, row_stats=True))
    tp = timeit(lambda: ops.linear(a, wo, b, res=rh))
    print(f"to_out + res + row stats L{lvl} [{M} x {C} -> {C}]: fp16 lo planes {t16:8.1f} | byte lo planes {t8:8.1f} | one-plane stream (no stats) {tp:8.1f}")
    t16 = timeit(lambda: ops.linear_x2(a, wo, b, row_stats=True)); t8 = timeit(lambda: ops.linear_lo8(a, wo, b, row_stats=True))
    print(f"proj_in (lo out, row stats) L{lvl}: fp16 lo plane {t16:8.1f} | byte lo plane {t8:8.1f}")
    del x4, w2, r32, rh, rl, rl8, a, wo
B, HW, C, Nk = 32, 4096, 320, 77
M = B * HW
h32 = rnd(M, C, scale=4.0, dt=torch.float32)
hh, hl = ops.split_f16(h32); _, hl8 = ops.lo8_split(h32)
g, bb = (1.0 + 0.1 * rnd(C).float()).half(), rnd(C, scale=0.1)
wq, wo, bo = rnd(C, C, scale=C ** -0.5), rnd(C, C, scale=C ** -0.5), rnd(C, scale=0.1)
kv = rnd(B, Nk, 2 * C)
t16 = timeit(lambda: ops.xattn_block_x2(hh, hl, g, bb, wq, kv, wo, bo, hw=HW, row_stats=True)); t8 = timeit(lambda: ops.xattn_block_lo8(hh, hl8, g, bb, wq, kv, wo, bo, hw=HW, row_stats=True))
tp = timeit(lambda: ops.xattn_block(hh, g, bb, wq, kv, wo, bo, hw=HW))
print(f"fused cross-attention block L0 (batch 32): fp16 lo planes {t16:8.1f} | byte lo planes {t8:8.1f} | one-plane stream {tp:8.1f}")
del h32, hh, hl, hl8
for name, (Bc, H, cin, cout, post) in (("UNet eps head 320 -> 4 at 64 x 64, batch 32", (32, 64, 320, 4, False)), ("VAE image head 128 -> 3 at 512 x 512, batch 16", (16, 512, 128, 3, True))):
    x = rnd(Bc, H, H, cin); w = ops.pack_conv_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5).to(dev); bias = rnd(cout, scale=0.1)
    ops.set_tuning("conv_out_mfma", 0); t0 = timeit(lambda: ops.conv_out(x, w, bias, postprocess=post), 10); o0 = ops.conv_out(x, w, bias, postprocess=post)
    ops.set_tuning("conv_out_mfma", 1); t1 = timeit(lambda: ops.conv_out(x, w, bias, postprocess=post), 10); o1 = ops.conv_out(x, w, bias, postprocess=post)
    gb = x.numel() * 2 / 1e9
    print(f"conv_out, {name}: v_dot2 patch kernel {t0:8.1f} | MFMA kernel {t1:8.1f} ({gb / (t1 * 1e-6) / 1e3:.2f} TB/s of input; max |diff| {float((o0.float() - o1.float()).abs().max()):.1e})")
    del x
