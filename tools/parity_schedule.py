"""Precision SCHEDULE of the denoiser's residual stream along a trajectory: the first k forwards with the split (hi + lo) stream, the rest with one fp16 plane.
Per-step latent drift against the fp32 oracle on the full UNet for n = 4 / 8 / 12 / 15 and k = 0 .. n, one oracle trajectory per n:   python tools/parity_schedule.py
Other weight seeds / batch sizes: CS_SCHED_SEED=8 CS_SCHED_B=2 CS_SCHED_NS=4,8 CS_SCHED_KS=auto python tools/parity_schedule.py ("auto" = all-split, all one-plane, and
the engine's default ceil(n / 4) with its two neighbours)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_parity_e2e_gpu as T
from consolver_amd import ops
for kv in os.environ.get("CS_TUNE", "").split(","):          # library knobs for the whole run: CS_TUNE="up_fold=2,..."
    if "=" in kv:
        ops.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))

g, B, wseed = 3.0, int(os.environ.get("CS_SCHED_B", "1")), int(os.environ.get("CS_SCHED_SEED", "7"))
NS = [int(v) for v in os.environ.get("CS_SCHED_NS", "4,8,12,15").split(",")]
AUTO = os.environ.get("CS_SCHED_KS", "") == "auto"
print(f"weight seed {wseed}, batch {B} (CFG {g}): per-step relative L2 of the latents vs the fp32 oracle, x 1e3", flush=True)
ux2, _ = T.build_full(seed=wseed, residual="f16x2")
for n in NS:
    c = T._oracle_case(n, wseed, g, B)
    sch, idx, noise, ctx_d = c["sch"], c["idx"], c["noise"], c["ctx"].to(T.DEV)
    ka = min(n, max(1, -(-n // 4)))
    for k in ([n, 0] + [v for v in (ka - 1, ka, ka + 1) if 0 < v < n] if AUTO else [n, 0, 1, 2, 3, 4, 6][: (7 if n > 4 else 5)]):
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
