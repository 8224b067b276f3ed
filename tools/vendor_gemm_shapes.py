"""Reference point only (not used by the product): the vendor library's fp16 GEMM (torch.matmul -> hipBLASLt) on the UNet's linear shapes,
next to cs_op_linear on the same box.  No bias / residual / GEGLU on the vendor side: it is the k loop + plain store that is compared."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from consolver_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0")
shapes = [("square", 8192, 8192, 8192), ("qkv L0", 131072, 320, 960), ("out L0", 131072, 320, 320), ("ff1 L0 (no geglu)", 131072, 320, 2560),
          ("ff2 L0", 131072, 1280, 320), ("qkv L1", 32768, 640, 1920), ("ff1 L1 (no geglu)", 32768, 640, 5120), ("ff2 L1", 32768, 2560, 640),
          ("qkv L2", 8192, 1280, 3840), ("ff1 L2 (no geglu)", 8192, 1280, 10240), ("ff2 L2", 8192, 5120, 1280)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print(f"{'shape':22s} {'M':>7s} {'K':>6s} {'N':>6s} {'vendor ms':>10s} {'PF':>5s} {'ours ms':>9s} {'PF':>5s}")
for tag, M, K, N in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.float16); w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.05
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    wt = w.t()
    tv = timeit(lambda: torch.matmul(x, wt, out=out))
    st = L.stream_ptr()
    to = timeit(lambda: L.check(lib.cs_op_linear(L.ptr(x), M, K, L.ptr(w), None, N, None, L.ptr(out), 0, st)))
    fl = 2.0 * M * K * N
    print(f"{tag:22s} {M:7d} {K:6d} {N:6d} {tv:10.3f} {fl / tv / 1e12:5.2f} {to:9.3f} {fl / to / 1e12:5.2f}")
