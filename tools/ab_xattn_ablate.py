"""xattn_block_kernel (the fused cross-attention sub-block at C = 320) with parts switched off through its debug bits (results wrong there, only the time matters):
1 no attention, 2 no out GEMM, 4 no q GEMM, 8 no hidden-state loads, 16 no residual loads / stores."""
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
B, HW, C, Nk = 32, 4096, 320, 77
h32 = torch.randn(B * HW, C, device=dev) * 2
hh, hl = ops.split_f16(h32)
g, be = rnd(C) * 0.1 + 1, rnd(C, scale=0.1)
wq, wo, bo = rnd(C, C, scale=C ** -0.5), rnd(C, C, scale=C ** -0.5), rnd(C)
kv = rnd(B, Nk, 2 * C)
for dbg in (0, 1, 2, 4, 6, 7, 8, 16, 24, 31):
    ops.set_tuning("debug", dbg)
    t2 = timeit(lambda: ops.xattn_block_x2(hh, hl, g, be, wq, kv, wo, bo, hw=HW, row_stats=True))
    t1 = timeit(lambda: ops.xattn_block(hh, g, be, wq, kv, wo, bo, hw=HW))
    print(f"debug {dbg:2d}:  f16x2 {t2:7.1f} us   f16 {t1:7.1f} us")
ops.set_tuning("debug", 0)
