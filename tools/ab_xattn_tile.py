"""xattn64_kernel (64-row tiles, two workgroups per CU; knob xattn_tile = 64) against xattn_block_kernel (128-row tiles, one 160 KB workgroup per CU; 128) on the UNet's
shape (64 x 64 level, batch 32 and the CFG-shared half batch), alternating, both stream modes, with a bit comparison of every output plane."""
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
C, Nk = 320, 77
for B, HW in ((32, 4096), (16, 4096), (32, 1024)):
    h32 = torch.randn(B * HW, C, device=dev) * 2
    hh, hl = ops.split_f16(h32)
    g, be = rnd(C) * 0.1 + 1, rnd(C, scale=0.1)
    wq, wo, bo = rnd(C, C, scale=C ** -0.5), rnd(C, C, scale=C ** -0.5), rnd(C)
    kv = rnd(B, Nk, 2 * C)
    outs = {}
    for rnd_i in range(2):
        for tile in (128, 64):
            ops.set_tuning("xattn_tile", tile)
            t2 = timeit(lambda: ops.xattn_block_x2(hh, hl, g, be, wq, kv, wo, bo, hw=HW, row_stats=True))
            t1 = timeit(lambda: ops.xattn_block(hh, g, be, wq, kv, wo, bo, hw=HW))
            outs[tile] = ops.xattn_block_x2(hh, hl, g, be, wq, kv, wo, bo, hw=HW, row_stats=True)
            print(f"B={B} HW={HW} tile {tile:3d}:  f16x2 {t2:7.1f} us   f16 {t1:7.1f} us")
    print("   bit-identical:", all(torch.equal(a, b) for a, b in zip(outs[64], outs[128])))
ops.set_tuning("xattn_tile", 64)
