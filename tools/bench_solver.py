"""K1 (fused CFG + LMS combine + DDIM/Euler) bandwidth: algorithmic bytes / kernel time vs HBM peak.

Algorithmic bytes per step: passes = [x] + [eps_u, eps_c] + [hist m-1] + [x'] + [eps store]."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import _lib as L

dev = torch.device("cuda:0")
lib = L.lib()

def run(B, elems, dtype, m, cfg=True, euler=False, iters=50):
    t = lambda: torch.randn(B, elems, device=dev).to(dtype)
    x, eu, ec, out, eo = t(), t(), t(), t(), t()
    hist = [t() for _ in range(m - 1)]
    actions = torch.rand(B, 3, device=dev)
    a = L.CsStepArgs()
    a.x, a.eps_text, a.eps_uncond, a.guidance = x.data_ptr(), ec.data_ptr(), (eu.data_ptr() if cfg else None), 3.0
    for k, h in enumerate(hist): a.hist[k] = h.data_ptr()
    a.m, a.order_dim, a.scaler_dim = m, 4, 0
    a.actions, a.actions_stride, a.B, a.elems = actions.data_ptr(), 3, B, elems
    a.io_dtype = a.out_dtype = L.dtype_code(dtype)
    a.x_out, a.eps_out = out.data_ptr(), (eo.data_ptr() if cfg else None)
    a.sqrt_at, a.sqrt_1mat, a.sqrt_ap, a.sqrt_1map, a.dt = 0.3, 0.95, 0.5, 0.86, -0.1
    fn = lib.cs_lms_euler_step if euler else lib.cs_lms_ddim_step
    st = L.stream_ptr(dev)
    for _ in range(3): L.check(fn(C.byref(a), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.check(fn(C.byref(a), st))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    passes = 1 + (2 if cfg else 1) + (m - 1) + 1 + (1 if cfg else 0)
    nbytes = passes * B * elems * x.element_size()
    return dict(B=B, elems=elems, dtype=str(dtype).split(".")[-1], m=m, cfg=cfg, euler=euler, us=round(us, 2),
                MB=round(nbytes / 1e6, 2), GBps=round(nbytes / us / 1e3, 1), frac_of_8TBps=round(nbytes / us / 1e3 / 8000, 3))

rows = [run(16, 16384, torch.float16, 4),            # configs[1]: batch 16, steady state step
        run(80, 16384, torch.float16, 4),            # configs[4]: PPO rollout batch
        run(1, 262144, torch.bfloat16, 2, cfg=False, euler=True),   # configs[3]: FLUX packed latents
        run(4096, 16384, torch.float16, 4),          # 1 GB working set: what the kernel sustains when HBM-bound
        run(2048, 262144, torch.bfloat16, 2, cfg=False, euler=True)]
for r in rows: print(json.dumps(r))


def policy_us(B, hidden=256, K=11, order=4, iters=200):
    """policy MLP + softmax (cs_factor_probs) per sampling step: one conditioning row broadcast to B samples"""
    import consolver_amd
    net = consolver_amd.FactorNetPPO(hidden_dim=hidden, num_actions=K, order_dim=order, scaler_dim=0)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    net.to(dev)
    row = torch.tensor([[874.0, 749.0]], device=dev)
    rows = row.repeat(B, 1).contiguous()
    out = {}
    for name, fn in (("broadcast_row", lambda: net.probs_from(row, batch=B)), ("per_row", lambda: net.probs_from(rows))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        out[name + "_us"] = round(e0.elapsed_time(e1) / iters * 1e3, 2)     # includes the host-side launch path
    return dict(kernel="factor_probs", B=B, hidden=hidden, **out)

print(json.dumps(policy_us(16)))
print(json.dumps(policy_us(80)))
