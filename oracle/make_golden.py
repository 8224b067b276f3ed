"""Generate tests/golden/*.npz from the IMPORTED reference (build container only).

TEST INFRASTRUCTURE ONLY.  Run as ``python oracle/make_golden.py`` in the
container that has /root/reference mounted.  It imports the reference solver
modules read-only through ``oracle/ref_stubs.py`` (one subprocess per flavour,
because both flavours ship a module called ``factor_net_ppo``), drives them on
seeded inputs and writes only *inputs and outputs* (data) to ``tests/golden``.
No reference source text is copied anywhere.

The stochastic call (``torch.multinomial``, factor_net_ppo.py:161) is replaced
during trajectory generation by a recorded index sequence so that the same
action indices can be replayed by the oracle and by the HIP path.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def eps_model_np(x, t, noise):
    """closed-form synthetic epsilon model used by every trajectory fixture."""
    return (0.9 * np.tanh(x) + 1e-5 * float(t) + 0.05 * noise).astype(np.float32)


# ----------------------------------------------------------------------------
def _seeded_weights(torch, net, seed, std=0.5):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * std / max(1.0, float(p.shape[-1]) ** 0.5) * 4.0)
    return {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}


class ForcedMultinomial:
    """context manager: torch.multinomial returns recorded indices."""
    def __init__(self, torch, rng):
        self.torch, self.rng, self.log = torch, rng, []

    def __enter__(self):
        self.orig = self.torch.multinomial
        def fake(probs, num_samples=1, **kw):
            K = probs.shape[-1]
            idx = self.rng.integers(0, K, size=(probs.shape[0], num_samples))
            self.log.append(idx.copy())
            return self.torch.from_numpy(idx).to(probs.device)
        self.torch.multinomial = fake
        return self

    def __exit__(self, *a):
        self.torch.multinomial = self.orig


def gen_sd():
    sys.path.insert(0, HERE)
    import ref_stubs
    ref_stubs.install("sd")
    import torch
    torch.set_num_threads(1)
    torch.set_grad_enabled(False)
    with ref_stubs.quiet():
        from scheduler_ppo import PPOScheduler
        from factor_net_ppo import FactorNetPPO
        import denoise_ppo

    # ---------------- tables -------------------------------------------------
    tab = {}
    with ref_stubs.quiet():
        s = PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                         factor_net_kwargs=dict(num_actions=3, hidden_dim=4))
        tab["ac_scaled_linear"] = s.alphas_cumprod.numpy()
        s = PPOScheduler(factor_net_kwargs=dict(num_actions=3, hidden_dim=4))
        tab["ac_linear"] = s.alphas_cumprod.numpy()
        s = PPOScheduler(beta_schedule="squaredcos_cap_v2", factor_net_kwargs=dict(num_actions=3, hidden_dim=4))
        tab["ac_cos"] = s.alphas_cumprod.numpy()
        for spacing in ("trailing", "leading", "linspace"):
            for off in ((0, 1) if spacing == "leading" else (0,)):
                s = PPOScheduler(timestep_spacing=spacing, steps_offset=off,
                                 factor_net_kwargs=dict(num_actions=3, hidden_dim=4))
                flat, offs = [], [0]
                for n in range(1, 51):
                    s.set_timesteps(n)
                    flat.append(s.timesteps.numpy().astype(np.int64))
                    offs.append(offs[-1] + len(flat[-1]))
                tab[f"ts_{spacing}_off{off}"] = np.concatenate(flat)
                tab[f"ts_{spacing}_off{off}_offsets"] = np.asarray(offs, np.int64)
        s = PPOScheduler(timestep_spacing="trailing", factor_net_kwargs=dict(num_actions=3, hidden_dim=4))
        s.set_timesteps(61)
        tab["ts_trailing_n61"] = s.timesteps.numpy().astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "sd_tables.npz"), **tab)

    # ---------------- factor net ---------------------------------------------
    fn = {}
    cases = [(4, 0, False, 11, 64), (4, 2, False, 11, 64), (4, 0, True, 11, 64),
             (2, 1, False, 161, 32), (3, 0, True, 11, 32), (2, 0, False, 11, 32)]
    fn["cases"] = np.asarray([[o, sc, int(uc), K, H] for o, sc, uc, K, H in cases], np.int64)
    for ci, (o, sc, uc, K, H) in enumerate(cases):
        with ref_stubs.quiet():
            net = FactorNetPPO(hidden_dim=H, num_actions=K, order_dim=o, scaler_dim=sc, use_conv=uc)
        w = _seeded_weights(torch, net, 1000 + ci)
        for k, v in w.items():
            fn[f"c{ci}_w_{k}"] = v
        B = 5
        g = torch.Generator().manual_seed(77 + ci)
        x = torch.tensor([[999., 874.], [124., -1.], [500., 375.], [0., -125.], [749., 499.]])
        eps = torch.randn(B, o, 4, 6, 6, generator=g)
        eps[1, o - 1] = 0  # a zero-padded slot
        eps[3, 1:] = 0
        xd = {"x": x, "epsilon": eps}
        with torch.no_grad():
            probs = net.forward_(xd)
            torch.manual_seed(4321 + ci)
            actions, aprobs = net.sample_action(xd)
            torch.manual_seed(4321 + ci)
            idx = torch.multinomial(probs.view(-1, K), 1).view(-1, net.action_dims)
            sel, ent = net.get_action_probs(xd, actions)
            # off-grid actions for the nearest-bin search
            pert = actions + 0.3 * (net.action_values[:, 1] - net.action_values[:, 0]).unsqueeze(0)
            sel2, _ = net.get_action_probs(xd, pert)
        fn[f"c{ci}_x"] = x.numpy()
        fn[f"c{ci}_eps"] = eps.numpy()
        fn[f"c{ci}_probs"] = probs.numpy()
        fn[f"c{ci}_idx"] = idx.numpy()
        fn[f"c{ci}_actions"] = actions.numpy()
        fn[f"c{ci}_aprobs"] = aprobs.numpy()
        fn[f"c{ci}_sel"] = sel.numpy()
        fn[f"c{ci}_entropy"] = ent.numpy()
        fn[f"c{ci}_pert"] = pert.numpy()
        fn[f"c{ci}_sel_pert"] = sel2.numpy()
    np.savez_compressed(os.path.join(OUT, "sd_factor_net.npz"), **fn)

    # ---------------- step trajectories --------------------------------------
    st = {}
    tcases = [  # order, scaler, use_conv, n_steps, spacing, pred_type
        (4, 0, False, 8, "trailing", "epsilon"),
        (4, 2, False, 6, "trailing", "epsilon"),
        (4, 1, True, 6, "trailing", "epsilon"),
        (2, 0, False, 4, "trailing", "epsilon"),
        (3, 2, True, 5, "leading", "epsilon"),
        (4, 0, False, 6, "linspace", "v_prediction"),
        (2, 1, False, 15, "trailing", "epsilon"),
    ]
    st["cases"] = np.asarray([[o, sc, int(uc), n, ["trailing", "leading", "linspace"].index(sp),
                               int(pt == "v_prediction")] for o, sc, uc, n, sp, pt in tcases], np.int64)
    for ci, (o, sc, uc, n, sp, pt) in enumerate(tcases):
        K, H, B = 11, 32, 3
        with ref_stubs.quiet():
            s = PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                             timestep_spacing=sp, prediction_type=pt, order_dim=o, scaler_dim=sc,
                             use_conv=uc, steps_offset=1 if sp == "leading" else 0,
                             factor_net_kwargs=dict(hidden_dim=H, num_actions=K))
        w = _seeded_weights(torch, s.factor_net, 2000 + ci)
        for k, v in w.items():
            st[f"t{ci}_w_{k}"] = v
        rng = np.random.default_rng(500 + ci)
        x = rng.standard_normal((B, 4, 8, 8)).astype(np.float32)
        st[f"t{ci}_x0"] = x.copy()
        s.set_timesteps(n)
        st[f"t{ci}_timesteps"] = s.timesteps.numpy().astype(np.int64)
        with ForcedMultinomial(torch, np.random.default_rng(900 + ci)) as fm, ref_stubs.quiet():
            for i, t in enumerate(s.timesteps):
                e = eps_model_np(x, int(t), rng.standard_normal(x.shape).astype(np.float32))
                out = s.step(torch.from_numpy(e), t, torch.from_numpy(x), return_dict=False)
                prev, actions, probs, conds, masks = out
                st[f"t{ci}_s{i}_eps"] = e
                st[f"t{ci}_s{i}_prev"] = prev.numpy()
                st[f"t{ci}_s{i}_actions"] = actions.numpy()
                st[f"t{ci}_s{i}_probs"] = probs.numpy()
                st[f"t{ci}_s{i}_condx"] = conds["x"].numpy()
                st[f"t{ci}_s{i}_masks"] = masks.numpy()
                st[f"t{ci}_s{i}_idx"] = fm.log[-1].reshape(B, -1)
                x = prev.numpy().copy()
        st[f"t{ci}_final"] = x
    # fp16-io case (documentation of the reference's dtype promotion)
    with ref_stubs.quiet():
        s = PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                         timestep_spacing="trailing", order_dim=4, scaler_dim=0,
                         factor_net_kwargs=dict(hidden_dim=32, num_actions=11))
    w = _seeded_weights(torch, s.factor_net, 2999)
    for k, v in w.items():
        st[f"h_w_{k}"] = v
    rng = np.random.default_rng(599)
    x = torch.from_numpy(rng.standard_normal((2, 4, 8, 8)).astype(np.float32)).half()
    st["h_x0"] = x.numpy()
    s.set_timesteps(4)
    with ForcedMultinomial(torch, np.random.default_rng(999)) as fm, ref_stubs.quiet():
        for i, t in enumerate(s.timesteps):
            e = torch.from_numpy(eps_model_np(x.float().numpy(), int(t),
                                              rng.standard_normal(x.shape).astype(np.float32))).half()
            prev = s.step(e, t, x, return_dict=False)[0]
            st[f"h_s{i}_eps"] = e.numpy()
            st[f"h_s{i}_prev"] = prev.float().numpy()
            st[f"h_s{i}_prev_dtype"] = np.asarray(str(prev.dtype))
            st[f"h_s{i}_idx"] = fm.log[-1].reshape(2, -1)
            x = prev.half()
    np.savez_compressed(os.path.join(OUT, "sd_steps.npz"), **st)

    # ---------------- denoise_diffusion rollout records ----------------------
    ro = {}

    class Tok:
        model_max_length = 7
        def __call__(self, text, **kw):
            class R: pass
            r = R()
            ids = torch.tensor([[(len(t) * 7 + j) % 13 for j in range(7)] for t in text])
            r.input_ids = ids
            return r

    def text_encoder(ids):
        return (torch.sin(ids.float()[..., None] * torch.arange(1, 9).float() * 0.37),)

    noise_bank = {}

    def unet(latent_in, t, encoder_hidden_states=None, return_dict=False):
        key = int(t)
        if key not in noise_bank:
            r = np.random.default_rng(7000 + key)
            noise_bank[key] = r.standard_normal(tuple(latent_in.shape)).astype(np.float32)
        ctx = encoder_hidden_states.mean(dim=(1, 2)).view(-1, 1, 1, 1).numpy()
        e = eps_model_np(latent_in.numpy(), key, noise_bank[key]) + 0.1 * ctx
        return (torch.from_numpy(e.astype(np.float32)),)

    # (the third case: use_conv under classifier-free guidance -- the cosine features see the COMBINED eps, scheduler_ppo.py:207-240 behind denoise_ppo.py:96-100)
    for ri, (o, sc, uc, n, cfg) in enumerate([(4, 0, False, 6, 3.0), (3, 1, True, 5, 1.0), (4, 0, True, 6, 3.0)]):
        noise_bank.clear()
        with ref_stubs.quiet():
            s = PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                             timestep_spacing="trailing", order_dim=o, scaler_dim=sc, use_conv=uc,
                             factor_net_kwargs=dict(hidden_dim=32, num_actions=11))
        w = _seeded_weights(torch, s.factor_net, 3000 + ri)
        for k, v in w.items():
            ro[f"r{ri}_w_{k}"] = v
        B = 2
        rng = np.random.default_rng(3100 + ri)
        noise = rng.standard_normal((B, 4, 8, 8)).astype(np.float32)
        text = ["a photo of a cat", "mi355x"]
        with ForcedMultinomial(torch, np.random.default_rng(3200 + ri)) as fm, ref_stubs.quiet():
            lat, conds, probs, actions, masks, pe = denoise_ppo.denoise_diffusion(
                text_encoder, s, unet, torch.from_numpy(noise), text, Tok(), cfg=cfg,
                num_inference_steps=n)
        ro[f"r{ri}_cfg"] = np.asarray([o, sc, int(uc), n], np.int64)
        ro[f"r{ri}_guidance"] = np.float32(cfg)
        ro[f"r{ri}_noise"] = noise
        ro[f"r{ri}_prompt_embeds"] = pe.numpy()
        ro[f"r{ri}_neg_embeds"] = text_encoder(Tok()([""] * B).input_ids)[0].numpy()
        ro[f"r{ri}_idx"] = np.stack([l.reshape(B, -1) for l in fm.log])
        for key in sorted(noise_bank):
            ro[f"r{ri}_unet_noise_{key}"] = noise_bank[key]
        ro[f"r{ri}_latents"] = lat.numpy()
        ro[f"r{ri}_conds_x"] = conds["x"].numpy()
        ro[f"r{ri}_conds_eps"] = conds["epsilon"].numpy()
        ro[f"r{ri}_probs"] = probs.numpy()
        ro[f"r{ri}_actions"] = actions.numpy()
        ro[f"r{ri}_masks"] = masks.numpy()
    np.savez_compressed(os.path.join(OUT, "sd_rollout.npz"), **ro)

    # ---------------- PPO policy update (train_ppo.py:404-437) -------------------
    # The reference's FactorNetPPO under torch autograd, the loss expression of the training loop, clip_grad_norm_ and
    # torch.optim.AdamW for two optimisation epochs on one collected batch.
    up = {}
    torch.set_grad_enabled(True)
    for ui, (o, sc, uc, K, H, R) in enumerate([(4, 0, False, 11, 32, 48), (2, 2, True, 5, 16, 21), (3, 1, False, 7, 24, 1)]):
        with ref_stubs.quiet():
            net = FactorNetPPO(hidden_dim=H, num_actions=K, order_dim=o, scaler_dim=sc, use_conv=uc)
        w0 = _seeded_weights(torch, net, 4000 + ui)
        A = o + sc - 1
        rng = np.random.default_rng(4100 + ui)
        ts = rng.integers(1, 999, size=(R, 1)).astype(np.float32)
        x = np.concatenate([ts, np.maximum(ts - 125, 0)], 1).astype(np.float32)
        conds = {"x": torch.from_numpy(x)}
        if uc:
            eps = rng.standard_normal((R, o, 4, 6, 6)).astype(np.float32)
            conds["epsilon"] = torch.from_numpy(eps)
            up[f"u{ui}_eps"] = eps
        av = net.action_values.numpy()
        idx = rng.integers(0, K, size=(R, A))
        actions = np.take_along_axis(np.broadcast_to(av, (R, A, K)), idx[..., None], 2)[..., 0].astype(np.float32)
        with torch.no_grad():
            cur, _ = net.get_action_probs(conds, torch.from_numpy(actions))
        old = np.clip(cur.numpy() * rng.uniform(0.55, 1.6, size=(R, A)), 1e-4, 1.0).astype(np.float32)   # ratios on both sides of the clip range
        adv = (rng.standard_normal((R, 1)) * 10).astype(np.float32) * (rng.random((R, A)) > 0.25).astype(np.float32)
        opt = torch.optim.AdamW(net.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
        clip_range, entropy_coef, max_norm = 0.2, 0.01, 1.0
        for ep in range(2):
            cur, ent = net(conds, torch.from_numpy(actions))
            lp = (cur + 1e-9).log().sum(dim=1).unsqueeze(1)
            olp = (torch.from_numpy(old) + 1e-9).log().sum(dim=1).unsqueeze(1)
            ratio = (lp - olp).exp()
            clipped = torch.clamp(ratio, 1 - clip_range, 1 + clip_range)
            a = torch.from_numpy(adv)
            loss = -torch.min(a * ratio, a * clipped).mean() - entropy_coef * ent.mean()
            loss.backward()
            up[f"u{ui}_e{ep}_loss"] = np.float32(loss.item())
            for k, prm in net.named_parameters():
                up[f"u{ui}_e{ep}_grad_{k}"] = prm.grad.detach().numpy().copy()
            up[f"u{ui}_e{ep}_norm"] = np.float32(torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm).item())
            opt.step()
            opt.zero_grad()
            for k, v in net.state_dict().items():
                up[f"u{ui}_e{ep}_after_{k}"] = v.detach().numpy().copy()
        up[f"u{ui}_cfg"] = np.asarray([o, sc, int(uc), K, H, R], np.int64)
        up[f"u{ui}_hyper"] = np.asarray([1e-3, 0.9, 0.999, 1e-2, 1e-8, clip_range, entropy_coef, max_norm], np.float64)
        up[f"u{ui}_x"] = x
        up[f"u{ui}_actions"] = actions
        up[f"u{ui}_old_probs"] = old
        up[f"u{ui}_adv"] = adv
        for k, v in w0.items():
            up[f"u{ui}_w_{k}"] = v
    torch.set_grad_enabled(False)
    np.savez_compressed(os.path.join(OUT, "sd_ppo_update.npz"), **up)
    print("sd fixtures written")


def gen_flux():
    sys.path.insert(0, HERE)
    import ref_stubs
    ref_stubs.install("flux")
    import torch
    torch.set_num_threads(1)
    torch.set_grad_enabled(False)
    with ref_stubs.quiet():
        from scheduler_fmppo import FMPPOScheduler
        from factor_net_ppo import FactorNetPPO

    fx = {}
    # sigma tables: pipeline passes sigmas=linspace(1, 1/n, n) and mu (pipeline.py:1009-1026)
    for n in range(2, 9):
        with ref_stubs.quiet():
            s = FMPPOScheduler(shift=3.0, use_dynamic_shifting=True, order_dim=2, scaler_dim=0, mu_dim=0,
                               factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
        s.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=1.15)
        fx[f"sig_dyn_n{n}"] = s.sigmas.numpy()
        fx[f"ts_dyn_n{n}"] = s.timesteps.numpy()
        with ref_stubs.quiet():
            s = FMPPOScheduler(shift=3.0, use_dynamic_shifting=False, order_dim=2, scaler_dim=0, mu_dim=0,
                               factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
        s.set_timesteps(n)
        fx[f"sig_static_n{n}"] = s.sigmas.numpy()
        fx[f"ts_static_n{n}"] = s.timesteps.numpy()
    for mu in (0.5, 0.8, 1.15):
        with ref_stubs.quiet():
            s = FMPPOScheduler(use_dynamic_shifting=True, order_dim=2, scaler_dim=0, mu_dim=0,
                               factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
        s.set_timesteps(sigmas=np.linspace(1.0, 1 / 5, 5), mu=mu)
        fx[f"sig_mu{mu}"] = s.sigmas.numpy()

    # factor net (flux variant)
    cases = [(2, 0, 0, False, 11, 32), (4, 2, 1, False, 11, 32), (3, 0, 0, True, 11, 32), (2, 1, 1, False, 21, 16)]
    fx["fn_cases"] = np.asarray([[o, sc, mu, int(uc), K, H] for o, sc, mu, uc, K, H in cases], np.int64)
    for ci, (o, sc, mu, uc, K, H) in enumerate(cases):
        with ref_stubs.quiet():
            net = FactorNetPPO(hidden_dim=H, num_actions=K, order_dim=o, scaler_dim=sc, mu_dim=mu, use_conv=uc)
        g = torch.Generator().manual_seed(1500 + ci)
        with torch.no_grad():
            for p in net.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        w = {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
        for k, v in w.items():
            fx[f"f{ci}_w_{k}"] = v
        x = torch.tensor([[1.0, 0.9567], [0.5128, 0.3109], [0.3109, 0.0]])
        eps = torch.randn(3, o, 5, 8, generator=g)
        eps[2, 1:] = 0
        with torch.no_grad():
            probs = net.forward_({"x": x, "epsilon": eps})
            torch.manual_seed(99 + ci)
            actions, aprobs = net.sample_action({"x": x, "epsilon": eps})
            sel, ent = net.get_action_probs({"x": x, "epsilon": eps}, actions)
        fx[f"f{ci}_x"] = x.numpy()
        fx[f"f{ci}_eps"] = eps.numpy()
        fx[f"f{ci}_probs"] = probs.numpy()
        fx[f"f{ci}_actions"] = actions.numpy()
        fx[f"f{ci}_aprobs"] = aprobs.numpy()
        fx[f"f{ci}_sel"] = sel.numpy()
        fx[f"f{ci}_entropy"] = ent.numpy()

    # step trajectories
    tcases = [(2, 0, 0, False, 8, "bf16"), (2, 0, 0, False, 5, "f32"), (4, 2, 1, False, 6, "bf16"),
              (3, 1, 0, True, 5, "f32")]
    fx["t_cases"] = np.asarray([[o, sc, mu, int(uc), n, int(dt == "bf16")] for o, sc, mu, uc, n, dt in tcases], np.int64)
    for ci, (o, sc, mu, uc, n, dt) in enumerate(tcases):
        K, H, B = 11, 32, 2
        with ref_stubs.quiet():
            s = FMPPOScheduler(shift=3.0, use_dynamic_shifting=True, order_dim=o, scaler_dim=sc, mu_dim=mu,
                               use_conv=uc, factor_net_kwargs=dict(hidden_dim=H, num_actions=K))
        g = torch.Generator().manual_seed(2500 + ci)
        with torch.no_grad():
            for p in s.factor_net.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        for k, v in s.factor_net.state_dict().items():
            fx[f"t{ci}_w_{k}"] = v.numpy().copy()
        tdt = torch.bfloat16 if dt == "bf16" else torch.float32
        rng = np.random.default_rng(2600 + ci)
        x = torch.from_numpy(rng.standard_normal((B, 24, 16)).astype(np.float32)).to(tdt)
        fx[f"t{ci}_x0"] = x.float().numpy()
        s.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=1.15)
        s.set_begin_index(0)
        fx[f"t{ci}_sigmas"] = s.sigmas.numpy()
        with ForcedMultinomial(torch, np.random.default_rng(2700 + ci)) as fm, ref_stubs.quiet():
            for i, t in enumerate(s.timesteps):
                v = torch.from_numpy(eps_model_np(x.float().numpy(), float(t) / 1000.0 * 1e3,
                                                  rng.standard_normal(x.shape).astype(np.float32))).to(tdt)
                prev, actions, probs, conds, masks = s.step(v, t, x, return_dict=False)
                fx[f"t{ci}_s{i}_v"] = v.float().numpy()
                fx[f"t{ci}_s{i}_prev"] = prev.float().numpy()
                fx[f"t{ci}_s{i}_actions"] = actions.numpy()
                fx[f"t{ci}_s{i}_probs"] = probs.numpy()
                fx[f"t{ci}_s{i}_condx"] = conds["x"].float().numpy()
                fx[f"t{ci}_s{i}_masks"] = masks.numpy()
                fx[f"t{ci}_s{i}_idx"] = fm.log[-1].reshape(B, -1)
                x = prev
        fx[f"t{ci}_final"] = x.float().numpy()
    np.savez_compressed(os.path.join(OUT, "flux.npz"), **fx)
    print("flux fixtures written")


def gen_forward_process():
    """the forward-process members of the scheduler protocol (row a6): PPOScheduler.add_noise (scheduler_ppo.py:336-358) and
    FMPPOScheduler.scale_noise (edit_ppo/scheduler_fmppo.py:457-484) of the imported reference, one subprocess-free pass each
    (the two flavours ship same-named modules, so FLUX runs in a child process)."""
    sys.path.insert(0, HERE)
    import ref_stubs
    flavour = os.environ.get("CS_GOLDEN_FLAVOUR", "sd")
    ref_stubs.install(flavour)
    import torch
    torch.set_num_threads(1)
    rng = np.random.default_rng(8100)
    x = rng.standard_normal((5, 4, 6, 6)).astype(np.float32)
    n = rng.standard_normal((5, 4, 6, 6)).astype(np.float32)
    if flavour == "sd":
        with ref_stubs.quiet():
            from scheduler_ppo import PPOScheduler
            s = PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing",
                             order_dim=4, scaler_dim=0, factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
        t = np.asarray([999, 874, 500, 1, 0], np.int64)
        out = s.add_noise(torch.from_numpy(x), torch.from_numpy(n), torch.from_numpy(t))
        np.savez_compressed(os.path.join(OUT, "forward_process_sd.npz"), x=x, noise=n, t=t, noisy=out.numpy())
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "forward_process"], env=dict(os.environ, CS_GOLDEN_FLAVOUR="flux"))
    else:
        with ref_stubs.quiet():
            from scheduler_fmppo import FMPPOScheduler
            s = FMPPOScheduler(shift=3.0, use_dynamic_shifting=True, order_dim=2, scaler_dim=0, mu_dim=0,
                               factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
        s.set_timesteps(sigmas=np.linspace(1.0, 1 / 6, 6), mu=1.15)
        ts = s.timesteps[[0, 2, 5, 3, 1]].clone()
        out = s.scale_noise(torch.from_numpy(x), ts, torch.from_numpy(n))
        np.savez_compressed(os.path.join(OUT, "forward_process_flux.npz"), x=x, noise=n, t=ts.numpy(), sigmas=s.sigmas.numpy(), noisy=out.numpy())
    print("forward-process fixtures written:", flavour)


def _ref_functions(path, names):
    """exec the named top-level function definitions of a reference file that cannot be imported as a whole (it pulls in
    the un-vendored diffusers package), from where the file lies -- nothing is copied into the repo."""
    import ast, inspect, typing
    import torch
    src = open(path).read()
    tree = ast.parse(src)
    ns = dict(inspect=inspect, torch=torch, Optional=typing.Optional, Union=typing.Union, List=typing.List)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def gen_flux_rollout():
    """edit_ppo/denoise_diffusion.denoise_diffusion (the FLUX PPO rollout, a18) driven with the closed-form stub pipe
    (oracle/flux_stub_pipe.py): inputs, the forced action indices and the returned 6-tuple."""
    sys.path.insert(0, HERE)
    import types
    import ref_stubs
    ref_stubs.install("flux")
    import torch
    torch.set_num_threads(1)
    # the module imports two helpers from diffusers' Kontext pipeline and the pipeline class (:6-7); the reference ships its
    # own copies of both helpers in edit_ppo/pipeline.py:119-183 -- use those, executed from where they lie
    calc, retr = _ref_functions(os.path.join(ref_stubs.REF_ROOT, "edit_ppo", "pipeline.py"), ["calculate_shift", "retrieve_timesteps"])
    pk = types.ModuleType("diffusers.pipelines.flux.pipeline_flux_kontext")
    pk.calculate_shift, pk.retrieve_timesteps = calc, retr
    for name in ("diffusers.pipelines", "diffusers.pipelines.flux"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["diffusers.pipelines.flux.pipeline_flux_kontext"] = pk
    sys.modules["diffusers"].FluxKontextPipeline = object
    with ref_stubs.quiet():
        from scheduler_fmppo import FMPPOScheduler
        import denoise_diffusion as ref_dd
    from flux_stub_pipe import StubKontextPipe

    fr = {}
    cases = [(2, 0, 0, 4, 1, 2.5), (3, 1, 0, 5, 2, 3.5)]          # order, scaler, mu_dim, n, B, guidance
    fr["cases"] = np.asarray([[o, sc, mu, n, B] for o, sc, mu, n, B, _ in cases], np.int64)
    for ci, (o, sc, mu, n, B, gs) in enumerate(cases):
        with ref_stubs.quiet():
            s = FMPPOScheduler(shift=3.0, use_dynamic_shifting=True, order_dim=o, scaler_dim=sc, mu_dim=mu,
                               factor_net_kwargs=dict(hidden_dim=32, num_actions=11))
        g = torch.Generator().manual_seed(4500 + ci)
        with torch.no_grad():
            for p in s.factor_net.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        for k, v in s.factor_net.state_dict().items():
            fr[f"c{ci}_w_{k}"] = v.numpy().copy()
        rng = np.random.default_rng(4600 + ci)
        noise = torch.from_numpy(rng.standard_normal((B, 16, 8, 8)).astype(np.float32)).to(torch.bfloat16)
        image = torch.from_numpy(np.tanh(rng.standard_normal((B, 3, 64, 64))).astype(np.float32))
        text = ["make it red", "mi355x"][:B]
        pipe = StubKontextPipe()
        with ForcedMultinomial(torch, np.random.default_rng(4700 + ci)) as fm, ref_stubs.quiet():
            lat, imgs, conds, probs, actions, masks = ref_dd.denoise_diffusion(
                s, pipe, noise, text, image, cfg=gs, num_inference_steps=n)
        fr[f"c{ci}_guidance"] = np.float32(gs)
        fr[f"c{ci}_noise"] = noise.float().numpy()
        fr[f"c{ci}_image"] = image.numpy()
        fr[f"c{ci}_idx"] = np.stack([l.reshape(B, -1) for l in fm.log])
        fr[f"c{ci}_sigmas"] = s.sigmas.numpy()
        fr[f"c{ci}_timestep_seen"] = np.stack([t for _, t in pipe.transformer.calls])       # what the DiT was called with (t / 1000 in bf16)
        fr[f"c{ci}_latents"] = lat.float().numpy()
        fr[f"c{ci}_pred_images"] = imgs.float().numpy()
        fr[f"c{ci}_conds_x"] = conds["x"].float().numpy()
        fr[f"c{ci}_conds_eps"] = conds["epsilon"].float().numpy()
        fr[f"c{ci}_probs"] = probs.float().numpy()
        fr[f"c{ci}_actions"] = actions.float().numpy()
        fr[f"c{ci}_masks"] = masks.float().numpy()
    np.savez_compressed(os.path.join(OUT, "flux_rollout.npz"), **fr)
    print("flux rollout fixtures written")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1:
        {"sd": gen_sd, "flux": gen_flux, "flux_rollout": gen_flux_rollout, "forward_process": gen_forward_process}[sys.argv[1]]()
    else:
        for fl in ("sd", "flux", "flux_rollout", "forward_process"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), fl])
        os.system(f"ls -la {OUT}")
