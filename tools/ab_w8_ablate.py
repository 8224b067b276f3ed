"""Ablation timing of gemm_w8_kernel on the three L0 linear shapes (qkv LN-folded, FF1 GEGLU LN-folded, to_out + residual f16x2): run with CONSOLVER_HIP_LIB pointing at
a throw-away build whose kernel skips a part (results are wrong there; only the time matters)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).half()
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
M, C = 131072, 320
x = rnd(M, C); gam, bet = rnd(C) * 0.1 + 1, rnd(C, scale=0.1)
st = ops.row_stats(x)
out = []
for name, N, geglu in (("qkv", 3 * C, False), ("ff1", 8 * C, True)):
    w, b = rnd(N, C, scale=C ** -0.5), (rnd(N) if geglu else None)
    if geglu:
        wp, bp = ops.geglu_pack(w, b); w, b = wp.to(dev), bp.to(dev)
    wf, sf, bf = (t.to(dev) for t in ops.ln_fold_pack(w, b, gam, bet))
    out.append(f"{name} {timeit(lambda: ops.linear_ln(x, wf, sf, bf, st, 1, geglu=geglu)):7.1f}")
wo, bo = rnd(C, C, scale=C ** -0.5), rnd(C)
rh, rl = ops.split_f16(torch.randn(M, C, device=dev) * 2)
out.append(f"to_out+res_x2 {timeit(lambda: ops.linear_x2(x, wo, bo, res=rh, res_lo=rl, want_lo=True, row_stats=True)):7.1f}")
print("us:  " + "   ".join(out))
