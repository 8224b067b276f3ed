"""Precision SCHEDULE of the denoiser's residual stream along a trajectory: the first k forwards with the split (hi + lo) stream, the rest with one fp16 plane.
Per-step latent drift against the fp32 oracle on the full UNet for n = 4 / 8 / 12 / 15 and k = 0 .. n, one oracle trajectory per n:   python tools/parity_schedule.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_parity_e2e_gpu as T

g, B, wseed = 3.0, 1, 7
ux2, _ = T.build_full(seed=wseed, residual="f16x2")
for n in (4, 8, 12, 15):
    c = T._oracle_case(n, wseed, g, B)
    sch, idx, noise, ctx_d = c["sch"], c["idx"], c["noise"], c["ctx"].to(T.DEV)
    for k in [n, 0, 1, 2, 3, 4, 6][: (7 if n > 4 else 5)]:
        sch.factor_net.forced_action_idx = [torch.from_numpy(i).to(T.DEV) for i in idx]
        sch.set_timesteps(n, device=T.DEV)
        x = noise.to(T.DEV).float()
        d = []
        for i, t in enumerate(sch.timesteps):
            eps = ux2(x.half(), t, encoder_hidden_states=ctx_d, dup=2, reuse_kv=(i > 0), out_dtype=torch.float32, residual="f16x2" if i < k else "f16")[0]
            x = sch.step(eps[B:], t, x, return_dict=False, eps_uncond=eps[:B], guidance_scale=g)[0]
            d.append(T.rel_l2(x.float().cpu().numpy(), c["traj"][i]))
        ux2.set_residual_precision_keep("f16x2")
        print(f"n={n:2d} hi-precision steps k={k:2d}: max {max(d):.3e} final {d[-1]:.3e} | " + " ".join(f"{v * 1e3:.3f}" for v in d), flush=True)
