import torch, sys, os
sys.path.insert(0, os.getcwd())
from consolver_amd import ops
DEV = "cuda:0"
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).half().to(DEV)
for C in (320, 640, 1280):
    gam, bet = (1 + 0.1 * rnd(C, seed=2).float()).half(), rnd(C, seed=3, scale=0.1)
    for name, N, geglu in (("qkv", 3 * C, False), ("to_q", C, False), ("ff1", 8 * C, True)):
        w, b = rnd(N, C, seed=4, scale=C ** -0.5), (rnd(N, seed=5) if geglu else None)
        if geglu:
            wp, bp = ops.geglu_pack(w, b); w, b = wp.to(DEV), bp.to(DEV)
        wf, sf, bf = (t.to(DEV) for t in ops.ln_fold_pack(w, b, gam, bet))
        for M in (16, 64, 256, 1024):
            x = rnd(M, C, seed=6, scale=2.0)
            st = ops.row_stats(x)
            full = ops.linear_ln(x, wf, sf, bf, st, 1, geglu=geglu)
            h = M // 2
            lo = ops.linear_ln(x[:h].contiguous(), wf, sf, bf, st[:h].contiguous(), 1, geglu=geglu)
            hi = ops.linear_ln(x[h:].contiguous(), wf, sf, bf, st[h:].contiguous(), 1, geglu=geglu)
            print(f"consumer {name} C={C} M={M}: first half equal {torch.equal(full[:h], lo)}, second half equal {torch.equal(full[h:], hi)}")
    # producer statistics: position independence
    wo, bo = rnd(C, C, seed=7, scale=C ** -0.5), rnd(C, seed=8, scale=0.1)
    for M in (16, 64, 256, 1024):
        a = rnd(M, C, seed=9); r = rnd(M, C, seed=10, scale=2.0)
        h = M // 2
        def stats(aa, rr):
            oh, ol, (st, G) = ops.linear_x2(aa, wo, bo, res=rr, want_lo=False, row_stats=True)
            v = st.reshape(-1)[: aa.shape[0] * G * 2].view(aa.shape[0], G, 2).sum(1)
            return oh, v, G
        of, sf_, Gf = stats(a, r)
        o2, s2, G2 = stats(a[h:].contiguous(), r[h:].contiguous())
        print(f"producer C={C} M={M}: G {Gf} vs {G2}; out second half equal {torch.equal(of[h:], o2)}; stats equal {torch.equal(sf_[h:], s2)} maxdiff {float((sf_[h:] - s2).abs().max()):.3e}")
