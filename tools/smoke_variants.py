"""smoke()'s loop on other shapes (tools only): reduced-depth UNet at S x S latents, n steps, CFG 3, against the CPU oracle, to see where its 1.26e-3 at
16 x 16 / 2 steps comes from and which small configuration sits under north_star's 1e-3.   python tools/smoke_variants.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import consolver_amd
from consolver_amd.unet import HipUNet2DConditionModel
from consolver_amd.synth import synthetic_unet_state_dict, synthetic_prompt_embeds
from oracle.unet_oracle import UNetOracle
from oracle import solver_oracle as so

dev = torch.device("cuda:0")
for S, n, lpb in ((16, 2, 1), (16, 8, 1), (32, 2, 1), (32, 4, 1), (32, 8, 1), (64, 2, 1)):
    t0 = time.time()
    unet = HipUNet2DConditionModel(dict(layers_per_block=lpb, sample_size=S), device=dev)
    sd = synthetic_unet_state_dict(unet.manifest(), seed=1)
    unet.load_state_dict(sd)
    sch = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing", order_dim=4, scaler_dim=0,
                                     factor_net_kwargs=dict(hidden_dim=32, num_actions=11))
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for p in sch.factor_net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    w = {k: v.numpy().copy() for k, v in sch.factor_net.state_dict().items()}
    sch.factor_net.to(dev)
    B, cfg = 2, 3.0
    idx = np.random.default_rng(0).integers(0, 11, size=(n, B, 3))
    pe, ne = synthetic_prompt_embeds(B, seed=1001).half(), synthetic_prompt_embeds(B, seed=1002).half()
    noise = torch.randn(B, 4, S, S, generator=g).half()
    sch.set_timesteps(n, device=dev)
    x = noise.to(dev).float()
    ctx = torch.cat([ne, pe]).to(dev)
    orc_u = UNetOracle(sd, unet.config)
    orc_s = so.PPOSchedulerOracle(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing", order_dim=4, scaler_dim=0, num_actions=11, weights=w)
    orc_s.set_timesteps(n)
    xo = noise.float().numpy()
    ctx_f = torch.cat([ne, pe]).float()
    drift, fwd = [], []
    for i, t in enumerate(sch.timesteps):
        sch.factor_net.forced_action_idx = torch.from_numpy(idx[i]).to(dev)
        e = orc_u(torch.from_numpy(np.concatenate([xo, xo])), int(t), ctx_f).numpy()
        ef = unet(torch.from_numpy(xo).half().to(dev), t.float().reshape(1), encoder_hidden_states=ctx, dup=2, reuse_kv=False)[0].float().cpu().numpy()      # teacher-forced forward
        fwd.append(float(np.linalg.norm(ef - e) / np.linalg.norm(e)))
        eps = unet(x.half(), t.float().reshape(1), encoder_hidden_states=ctx, dup=2, reuse_kv=False)[0]
        x = sch.step(eps[B:], t, x, return_dict=False, eps_uncond=eps[:B], guidance_scale=cfg)[0]
        xo = orc_s.step(so.cfg_combine(e[:B], e[B:], cfg), int(t), xo, idx[i], cond_dtype="f16")["prev_sample"]
        drift.append(float(np.linalg.norm(x.float().cpu().numpy() - xo) / np.linalg.norm(xo)))
    print(f"S={S} n={n} layers_per_block={lpb}: per-forward eps error " + " ".join(f"{v:.2e}" for v in fwd) + "  | latents " + " ".join(f"{v:.2e}" for v in drift) + f"   ({time.time() - t0:.0f} s)", flush=True)
    del unet
