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
import hashlib
for tag, M, C in (("L0", 131072, 320), ("L1", 32768, 640), ("L2", 8192, 1280)):
    x = rnd(M, C); gam, bet = rnd(C) * 0.1 + 1, rnd(C, scale=0.1)
    st = ops.row_stats(x)
    w, b = rnd(8 * C, C, scale=C ** -0.5), rnd(8 * C)
    wp, bp = ops.geglu_pack(w, b); w, b = wp.to(dev), bp.to(dev)
    wf, sf, bf = (t.to(dev) for t in ops.ln_fold_pack(w, b, gam, bet))
    o = ops.linear_ln(x, wf, sf, bf, st, 1, geglu=True)
    h = hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"ff1 geglu {tag} folded LN  {timeit(lambda: ops.linear_ln(x, wf, sf, bf, st, 1, geglu=True)):8.1f} us  sha {h}")
