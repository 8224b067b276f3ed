"""Placement experiment behind profiles/r03_attn_pad_sweep.txt: needs a build whose launcher selects the PAD instantiations of attn40_lw_kernel through
cs_set_tuning("attn_pad", n) (removed from the tree after the sweep; the kernel keeps the PAD template parameter).  Without that knob this script times attn_lw 0 / 1 only."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch
from consolver_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0")
B, N, H, dh = 32, 4096, 8, 40
qkv = torch.randn(B, N, 3 * H * dh, device=dev).half()
q, k, v = qkv[..., :320], qkv[..., 320:640], qkv[..., 640:]
out = torch.empty(B, N, 320, dtype=torch.float16, device=dev)
call = lambda: L.check(lib.cs_op_attention(q.data_ptr(), 960, k.data_ptr(), 960, v.data_ptr(), 960, out.data_ptr(), 320, B, H, N, N, dh, dh ** -0.5, L.stream_ptr(dev)))
def t():
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
for rep in range(2):
    lib.cs_set_tuning(b"attn_lw", 0); print(f"attn_kernel: {t():.3f} ms")
    lib.cs_set_tuning(b"attn_lw", 1)
    for pad in range(16):
        if lib.cs_set_tuning(b"attn_pad", pad) != 0:
            print(f"attn40_lw_kernel: {t():.3f} ms"); break
        print(f"pad {pad:2d}: {t():.3f} ms")
