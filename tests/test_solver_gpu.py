"""GPU parity of the HIP solver path (through the C ABI) vs the golden vectors and the oracle."""
import numpy as np
import pytest
import torch

import consolver_amd
from oracle import solver_oracle as so

pytestmark = pytest.mark.gpu
SPACINGS = ["trailing", "leading", "linspace"]
DEV = "cuda:0"


def weights(npz, prefix):
    return {k[len(prefix):]: npz[k] for k in npz.files if k.startswith(prefix)}


def load_net(net, w):
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    net.to(DEV)


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def cu(x, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV).to(dtype)


def test_native_library_is_loaded():
    from consolver_amd import _lib
    _lib.lib()
    assert any("libconsolver_hip.so" in l for l in open("/proc/self/maps").read().splitlines())


def test_factor_net_probs_vs_golden(golden):
    g = golden["sd_factor_net"]
    for ci, (o, sc, uc, K, H) in enumerate(g["cases"]):
        net = consolver_amd.FactorNetPPO(hidden_dim=int(H), num_actions=int(K), order_dim=int(o), scaler_dim=int(sc),
                                         use_conv=bool(uc))
        load_net(net, weights(g, f"c{ci}_w_"))
        xd = {"x": cu(g[f"c{ci}_x"]), "epsilon": cu(g[f"c{ci}_eps"])}
        probs = net.forward_(xd)
        np.testing.assert_allclose(probs.cpu().numpy(), g[f"c{ci}_probs"], rtol=3e-4, atol=2e-6)
        net.forced_action_idx = cu(g[f"c{ci}_idx"], torch.int64)
        actions, aprobs = net.sample_action(xd)
        np.testing.assert_array_equal(actions.cpu().numpy(), g[f"c{ci}_actions"])
        np.testing.assert_allclose(aprobs.cpu().numpy(), g[f"c{ci}_aprobs"], rtol=3e-4, atol=2e-6)
        sel, ent = net(xd, cu(g[f"c{ci}_pert"]))
        np.testing.assert_allclose(sel.cpu().numpy(), g[f"c{ci}_sel_pert"], rtol=3e-4, atol=2e-6)
        np.testing.assert_allclose(ent.cpu().numpy(), g[f"c{ci}_entropy"], rtol=3e-4, atol=3e-6)
        if uc:
            cosf = net.cosine_features([xd["epsilon"][:, k].contiguous() for k in range(o)])
            np.testing.assert_allclose(cosf.cpu().numpy(), so.cosine_features(g[f"c{ci}_eps"]), atol=2e-6)
    g = golden["flux"]
    for ci, (o, sc, mu, uc, K, H) in enumerate(g["fn_cases"]):
        net = consolver_amd.FluxFactorNetPPO(hidden_dim=int(H), num_actions=int(K), order_dim=int(o),
                                             scaler_dim=int(sc), mu_dim=int(mu), use_conv=bool(uc))
        load_net(net, weights(g, f"f{ci}_w_"))
        probs = net.forward_({"x": cu(g[f"f{ci}_x"]), "epsilon": cu(g[f"f{ci}_eps"])})
        np.testing.assert_allclose(probs.cpu().numpy(), g[f"f{ci}_probs"], rtol=5e-3, atol=2e-5)


def test_samplers():
    net = consolver_amd.FactorNetPPO(hidden_dim=16, num_actions=11, order_dim=4, scaler_dim=0).to(DEV)
    p = torch.tensor([0.0, 0.1, 0.0, 0.5, 0.0, 0.0, 0.4, 0, 0, 0, 0], device=DEV).repeat(4000, 3, 1).contiguous()
    for mode in ("multinomial", "inverse_cdf"):
        net.sampler = mode
        torch.manual_seed(0)
        actions, aprobs, idx = net.draw(p)
        freq = torch.bincount(idx.flatten(), minlength=11).float() / idx.numel()
        assert torch.allclose(freq.cpu(), p[0, 0].cpu(), atol=0.02), (mode, freq)
        assert set(idx.unique().tolist()) <= {1, 3, 6}          # never a zero-probability bin
        av = net.action_values
        assert torch.equal(actions, av[torch.arange(3, device=DEV).expand_as(idx), idx])
        assert torch.equal(aprobs, p.gather(2, idx[..., None])[..., 0])
    # untrained net (zero last layer) is uniform (factor_net_ppo.py:82-83)
    probs = net.forward_({"x": torch.tensor([[999., 874.]], device=DEV)})
    assert torch.allclose(probs, torch.full_like(probs, 1 / 11), atol=1e-7)


@pytest.mark.parametrize("ci", range(7))
def test_sd_step_trajectories_vs_golden(golden, ci):
    g = golden["sd_steps"]
    o, sc, uc, n, sp, vp = [int(v) for v in g["cases"][ci]]
    s = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                   timestep_spacing=SPACINGS[sp], steps_offset=1 if sp == 1 else 0,
                                   prediction_type="v_prediction" if vp else "epsilon", order_dim=o, scaler_dim=sc,
                                   use_conv=bool(uc), factor_net_kwargs=dict(hidden_dim=32, num_actions=11))
    load_net(s.factor_net, weights(g, f"t{ci}_w_"))
    s.record_conds = True
    s.set_timesteps(n, device=DEV)
    assert np.array_equal(s.timesteps.cpu().numpy(), g[f"t{ci}_timesteps"])
    x = cu(g[f"t{ci}_x0"])
    for i, t in enumerate(s.timesteps):            # t is a 0-d CUDA tensor like in the pipeline loop
        s.factor_net.forced_action_idx = cu(g[f"t{ci}_s{i}_idx"], torch.int64)
        prev, actions, probs, conds, masks = s.step(cu(g[f"t{ci}_s{i}_eps"]), t, x, return_dict=False)
        np.testing.assert_array_equal(masks.cpu().numpy(), g[f"t{ci}_s{i}_masks"])
        np.testing.assert_array_equal(actions.cpu().numpy(), g[f"t{ci}_s{i}_actions"])
        np.testing.assert_array_equal(conds["x"].cpu().numpy(), g[f"t{ci}_s{i}_condx"])
        np.testing.assert_allclose(probs.cpu().numpy(), g[f"t{ci}_s{i}_probs"], rtol=5e-4, atol=2e-6)
        assert conds["epsilon"].shape == (3, o, 4, 8, 8)
        assert rel_l2(prev.cpu().numpy(), g[f"t{ci}_s{i}_prev"]) < 1e-6, (ci, i)
        x = prev
    assert rel_l2(x.cpu().numpy(), g[f"t{ci}_final"]) < 3e-6
    out = s.step(cu(g[f"t{ci}_s0_eps"]), int(s.timesteps[0]), x, return_dict=True)
    assert out.prev_sample.shape == x.shape and set(out) == {"prev_sample", "actions", "probs", "conds", "masks"}


@pytest.mark.parametrize("ci", range(4))
def test_flux_step_trajectories_vs_golden(golden, ci):
    g = golden["flux"]
    o, sc, mu, uc, n, bf = [int(v) for v in g["t_cases"][ci]]
    dt = torch.bfloat16 if bf else torch.float32
    s = consolver_amd.FMPPOScheduler(shift=3.0, use_dynamic_shifting=True, order_dim=o, scaler_dim=sc, mu_dim=mu,
                                     use_conv=bool(uc), factor_net_kwargs=dict(hidden_dim=32, num_actions=11))
    load_net(s.factor_net, weights(g, f"t{ci}_w_"))
    s.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=1.15, device=DEV)
    s.set_begin_index(0)
    x = cu(g[f"t{ci}_x0"], dt)
    for i, t in enumerate(s.timesteps):
        s.factor_net.forced_action_idx = cu(g[f"t{ci}_s{i}_idx"], torch.int64)
        prev, actions, probs, conds, masks = s.step(cu(g[f"t{ci}_s{i}_v"], dt), t, x, return_dict=False)
        assert prev.dtype == dt
        np.testing.assert_array_equal(masks.cpu().numpy(), g[f"t{ci}_s{i}_masks"])
        np.testing.assert_array_equal(actions.cpu().numpy(), g[f"t{ci}_s{i}_actions"])
        np.testing.assert_array_equal(conds["x"].float().cpu().numpy(), g[f"t{ci}_s{i}_condx"])
        got, want = prev.float().cpu().numpy(), g[f"t{ci}_s{i}_prev"]
        if bf:
            assert rel_l2(got, want) < 2e-3 and np.mean(got != want) < 0.02
        else:
            assert rel_l2(got, want) < 1e-6
        x = cu(want, dt)
    assert s.step_index == n


def _full_size_case(dtype, B=16, n=8, order=4, scaler=0, guidance=3.0, seed=0):
    """BASELINE config-2 shape: [B,4,64,64], 8 steps, CFG 3, fp16.  Oracle finishes in seconds."""
    rng = np.random.default_rng(seed)
    s = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                   timestep_spacing="trailing", order_dim=order, scaler_dim=scaler,
                                   factor_net_kwargs=dict(hidden_dim=256, num_actions=11))
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in s.factor_net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    w = {k: v.numpy().copy() for k, v in s.factor_net.state_dict().items()}
    s.factor_net.to(DEV)
    orc = so.PPOSchedulerOracle(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                timestep_spacing="trailing", order_dim=order, scaler_dim=scaler, num_actions=11, weights=w)
    s.set_timesteps(n, device=DEV)
    orc.set_timesteps(n)
    nm = "f16" if dtype == torch.float16 else "f32"
    x0 = so.round_like(rng.standard_normal((B, 4, 64, 64)).astype(np.float32), nm)
    x, xo = cu(x0, dtype), x0
    A = order + scaler - 1
    worst = 0.0
    for i, t in enumerate(s.timesteps):
        eu = so.round_like(rng.standard_normal(x0.shape).astype(np.float32), nm)
        ec = so.round_like((eu + 0.3 * rng.standard_normal(x0.shape)).astype(np.float32), nm)
        idx = rng.integers(0, 11, size=(B, A))
        s.factor_net.forced_action_idx = cu(idx, torch.int64)
        prev = s.step(cu(ec, dtype), t, x, return_dict=False, eps_uncond=cu(eu, dtype), guidance_scale=guidance)[0]
        e = so.round_like(so.cfg_combine(eu, ec, guidance), nm)
        xo = orc.step(e, int(t), xo, idx, cond_dtype=nm)["prev_sample"]
        worst = max(worst, rel_l2(prev.float().cpu().numpy(), xo))
        x = prev
        xo = so.round_like(xo, nm)   # the pipeline keeps latents in the model dtype
    return worst, rel_l2(x.float().cpu().numpy(), xo)


def test_full_size_fp16_cfg_trajectory_within_1e3():
    worst, final = _full_size_case(torch.float16)
    assert worst < 1e-3 and final < 1e-3, (worst, final)      # north_star gate: 1e-3 relative


def test_full_size_fp32_cfg_trajectory():
    worst, final = _full_size_case(torch.float32, B=4)
    assert worst < 1e-6 and final < 1e-6, (worst, final)


def test_fp32_solver_state_with_fp16_model_outputs():
    """round 5: the engine's solver state.  An fp32 `sample` next to fp16 eps tensors (CsStepArgs::x_is_f32) is read unrounded and stays fp32 -- torch's promotion, and what
    the reference's latents are from step 2 on with an fp32 policy net.  Against the fp32 oracle fed the same fp16-representable eps values the 8-step CFG trajectory
    is fp32-class (<= 2e-6: the CFG combine is rounded to fp16 as the history entry on both sides), where the fp16 state sits at ~3e-4 per step; a 16-bit sample keeps its
    dtype unless prev_sample_dtype asks for the promotion; out= of the wrong dtype is refused."""
    rng = np.random.default_rng(3)
    B, n, order, guidance = 4, 8, 4, 3.0
    s = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing", order_dim=order, scaler_dim=0,
                                   factor_net_kwargs=dict(hidden_dim=64, num_actions=11))
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in s.factor_net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    w = {k: v.numpy().copy() for k, v in s.factor_net.state_dict().items()}
    s.factor_net.to(DEV)
    orc = so.PPOSchedulerOracle(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing", order_dim=order, scaler_dim=0,
                                num_actions=11, weights=w)
    s.set_timesteps(n, device=DEV); orc.set_timesteps(n)
    x0 = rng.standard_normal((B, 4, 64, 64)).astype(np.float32)                 # NOT fp16-representable: an fp32 state
    x, xo = cu(x0, torch.float32), x0
    x16 = cu(so.round_like(x0, "f16"), torch.float16)
    for i, t in enumerate(s.timesteps):
        eu = so.round_like(rng.standard_normal(x0.shape).astype(np.float32), "f16")
        ec = so.round_like((eu + 0.3 * rng.standard_normal(x0.shape)).astype(np.float32), "f16")
        idx = rng.integers(0, 11, size=(B, order - 1))
        s.factor_net.forced_action_idx = cu(idx, torch.int64)
        prev = s.step(cu(ec, torch.float16), t, x, return_dict=False, eps_uncond=cu(eu, torch.float16), guidance_scale=guidance)[0]
        assert prev.dtype == torch.float32
        xo = orc.step(so.round_like(so.cfg_combine(eu, ec, guidance), "f16"), int(t), xo, idx, cond_dtype="f16")["prev_sample"]
        assert rel_l2(prev.cpu().numpy(), xo) < 2e-6, i
        x = prev
    # a 16-bit sample keeps its dtype ... unless the promotion is asked for
    s.set_timesteps(n, device=DEV)
    s.factor_net.forced_action_idx = cu(np.zeros((B, order - 1), np.int64), torch.int64)
    e16 = cu(so.round_like(rng.standard_normal(x0.shape).astype(np.float32), "f16"), torch.float16)
    assert s.step(e16, s.timesteps[0], x16, return_dict=False)[0].dtype == torch.float16
    s.set_timesteps(n, device=DEV)
    s.prev_sample_dtype = torch.float32
    try:
        p32 = s.step(e16, s.timesteps[0], x16, return_dict=False)[0]
        assert p32.dtype == torch.float32
        s.set_timesteps(n, device=DEV)
        with pytest.raises(ValueError):
            s.step(e16, s.timesteps[0], x16, return_dict=False, out=torch.empty_like(x16))
    finally:
        s.prev_sample_dtype = None
    s.set_timesteps(n, device=DEV)
    p16 = s.step(e16, s.timesteps[0], x16, return_dict=False)[0]
    assert torch.equal(p16, p32.to(torch.float16))                              # the same update, rounded once at the end


def test_properties_full_size():
    """size-independent properties at BASELINE shape: order-1 limit == DDIM, linearity in eps,
    coefficient sum == 1 (constant eps history is a fixed point of the combine)."""
    B, n = 16, 8
    s = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                   timestep_spacing="trailing", order_dim=4, scaler_dim=0,
                                   factor_net_kwargs=dict(hidden_dim=64, num_actions=11))
    s.factor_net.to(DEV)
    torch.manual_seed(1)
    x = torch.randn(B, 4, 64, 64, device=DEV)
    e = torch.randn(B, 4, 64, 64, device=DEV)
    # (1) constant history: any coefficients (they sum to 1) give the DDIM step with eps
    ac = s.alphas_cumprod
    outs = []
    for trial in range(2):
        s.set_timesteps(n, device=DEV)
        xs = x
        for i, t in enumerate(s.timesteps[:4]):
            s.factor_net.forced_action_idx = torch.randint(0, 11, (B, 3), device=DEV)
            xs = s.step(e, t, xs, return_dict=False)[0]
        outs.append(xs)
    assert rel_l2(outs[0].cpu().numpy(), outs[1].cpu().numpy()) < 2e-6
    # (2) first step == closed-form DDIM
    s.set_timesteps(n, device=DEV)
    s.factor_net.forced_action_idx = torch.randint(0, 11, (B, 3), device=DEV)
    p = s.step(e, s.timesteps[0], x, return_dict=False)[0]
    at, ap = ac[999].double(), ac[874].double()
    want = (ap / at).sqrt() * x.double().cpu() + ((1 - ap).sqrt() - ap.sqrt() * (1 - at).sqrt() / at.sqrt()) * e.double().cpu()
    assert rel_l2(p.cpu().numpy(), want.numpy()) < 1e-6
    # (3) the update is affine in (x, eps): step(a x1 + b x2, a e1 + b e2) = a step(x1,e1) + b step(x2,e2)
    x2, e2 = torch.randn_like(x), torch.randn_like(e)
    idx = torch.randint(0, 11, (B, 3), device=DEV)
    def two_steps(xa, ea):
        s.set_timesteps(n, device=DEV)
        s.factor_net.forced_action_idx = idx
        y = s.step(ea, s.timesteps[0], xa, return_dict=False)[0]
        return s.step(0.5 * ea, s.timesteps[1], y, return_dict=False)[0]
    lhs = two_steps(0.3 * x + 0.7 * x2, 0.3 * e + 0.7 * e2)
    rhs = 0.3 * two_steps(x, e) + 0.7 * two_steps(x2, e2)
    assert rel_l2(lhs.cpu().numpy(), rhs.cpu().numpy()) < 2e-6


def test_edge_cases():
    s = consolver_amd.PPOScheduler(timestep_spacing="trailing", order_dim=2, scaler_dim=2,
                                   factor_net_kwargs=dict(hidden_dim=8, num_actions=5))
    s.factor_net.to(DEV)
    s.set_timesteps(3, device=DEV)
    # empty batch
    z = torch.zeros(0, 4, 8, 8, device=DEV)
    out = s.step(z, s.timesteps[0], z, return_dict=False)
    assert out[0].shape == (0, 4, 8, 8) and out[1].shape == (0, 3)
    # ragged element count (not a multiple of the vector width) + history longer than n steps (stale history quirk)
    s.set_timesteps(3, device=DEV)
    w = {k: v.cpu().numpy() for k, v in s.factor_net.state_dict().items()}
    orc = so.PPOSchedulerOracle(timestep_spacing="trailing", order_dim=2, scaler_dim=2, num_actions=5, weights=w)
    orc.set_timesteps(3)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 3, 5, 7)).astype(np.float32)
    xo = x
    xg = cu(x)
    for i in range(5):                               # 2 steps past the schedule: history is not cleared
        t = int(s.timesteps[min(i, 2)])
        e = rng.standard_normal(x.shape).astype(np.float32)
        idx = rng.integers(0, 5, size=(2, 3))
        s.factor_net.forced_action_idx = cu(idx, torch.int64)
        xg = s.step(cu(e), t, xg, return_dict=False)[0]
        xo = orc.step(e, t, xo, idx)["prev_sample"]
        assert rel_l2(xg.cpu().numpy(), xo) < 1e-6
    # deterministic: same inputs twice -> bit-identical
    s.set_timesteps(3, device=DEV)
    a = s.step(cu(e), s.timesteps[0], cu(x), return_dict=False)[0]
    s.set_timesteps(3, device=DEV)
    b = s.step(cu(e), s.timesteps[0], cu(x), return_dict=False)[0]
    assert torch.equal(a, b)


def test_sd_rollout_records_vs_golden(golden):
    from consolver_amd.rollout import denoise_diffusion
    g = golden["sd_rollout"]
    from oracle.make_golden import eps_model_np
    for ri in range(3):
        o, sc, uc, n = [int(v) for v in g[f"r{ri}_cfg"]]
        cfg = float(g[f"r{ri}_guidance"])
        s = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                       timestep_spacing="trailing", order_dim=o, scaler_dim=sc, use_conv=bool(uc),
                                       factor_net_kwargs=dict(hidden_dim=32, num_actions=11))
        load_net(s.factor_net, weights(g, f"r{ri}_w_"))
        pe, ne = cu(g[f"r{ri}_prompt_embeds"]), cu(g[f"r{ri}_neg_embeds"])
        idx_seq = [cu(v, torch.int64) for v in g[f"r{ri}_idx"]]
        calls = {"i": 0}

        class Tok:
            model_max_length = 7
            def __call__(self, text, **kw):
                class R: pass
                r = R(); r.input_ids = torch.zeros(len(text), 7, dtype=torch.long, device=DEV); r.is_neg = text[0] == ""
                r.input_ids[:, 0] = int(r.is_neg)
                return r

        def text_encoder(ids):
            return (ne if int(ids[0, 0]) == 1 else pe,)

        def unet(lat, t, encoder_hidden_states=None, return_dict=False):
            s.factor_net.forced_action_idx = idx_seq[calls["i"]]
            calls["i"] += 1
            key = int(t)
            c = encoder_hidden_states.mean(dim=(1, 2)).view(-1, 1, 1, 1).cpu().numpy()
            e = eps_model_np(lat.cpu().numpy(), key, g[f"r{ri}_unet_noise_{key}"]) + (0.1 * c).astype(np.float32)
            return (cu(e),)

        lat, conds, probs, actions, masks, pe_out = denoise_diffusion(
            text_encoder, s, unet, cu(g[f"r{ri}_noise"]), ["a photo of a cat", "mi355x"], Tok(), cfg=cfg,
            num_inference_steps=n)
        assert conds["x"].shape == (2, n - 1, 2) and conds["epsilon"].shape == g[f"r{ri}_conds_eps"].shape
        np.testing.assert_array_equal(conds["x"].cpu().numpy(), g[f"r{ri}_conds_x"])
        np.testing.assert_array_equal(actions.cpu().numpy(), g[f"r{ri}_actions"])
        np.testing.assert_array_equal(masks.cpu().numpy(), g[f"r{ri}_masks"])
        np.testing.assert_allclose(probs.cpu().numpy(), g[f"r{ri}_probs"], rtol=5e-4, atol=2e-6)
        assert rel_l2(conds["epsilon"].cpu().numpy(), g[f"r{ri}_conds_eps"]) < 3e-6
        assert rel_l2(lat.cpu().numpy(), g[f"r{ri}_latents"]) < 3e-6
        assert torch.equal(pe_out, pe)


def test_ddim_baseline_scheduler_matches_oracle():
    """order-1 baseline (row f-4): eta = 0 DDIM on PPOScheduler's tables / grid / prev_t rule (scheduler_ppo.py:203,306-332)"""
    from consolver_amd.baselines import DDIMBaselineScheduler
    s = DDIMBaselineScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing")
    ac = so.alphas_cumprod(so.make_betas("scaled_linear", 0.00085, 0.012))
    rng = np.random.default_rng(0)
    for n in (1, 4, 8):
        s.set_timesteps(n, device=DEV)
        assert s.timesteps.cpu().tolist() == so.sd_timesteps(n, spacing="trailing").tolist()
        x = rng.standard_normal((3, 4, 8, 8)).astype(np.float32)
        xg, xo = cu(x), x
        for t in s.timesteps:
            e = (0.9 * np.tanh(xo) + 1e-5 * float(t)).astype(np.float32)
            xg = s.step(cu(e), t, xg, return_dict=False)[0]
            pt = so.sd_prev_timestep(int(t), n)
            xo = so.ddim_update(xo, e, ac[int(t)], ac[pt] if pt >= 0 else ac[0])
            assert rel_l2(xg.cpu().numpy(), xo) < 1e-6
    # fused CFG form and fp16 I/O
    s.set_timesteps(4, device=DEV)
    u, c = rng.standard_normal(x.shape).astype(np.float32), rng.standard_normal(x.shape).astype(np.float32)
    t = int(s.timesteps[0])
    got = s.step(cu(c), s.timesteps[0], cu(x), eps_uncond=cu(u), guidance_scale=3.0, return_dict=True).prev_sample
    want = so.ddim_update(x, so.cfg_combine(u, c, 3.0), ac[t], ac[so.sd_prev_timestep(t, 4)])
    assert rel_l2(got.cpu().numpy(), want) < 1e-6
    got16 = s.step(cu(c, torch.float16), s.timesteps[0], cu(x, torch.float16), return_dict=False)[0]
    assert got16.dtype == torch.float16
    want16 = so.ddim_update(so.round_f16(x), so.round_f16(c), ac[t], ac[so.sd_prev_timestep(t, 4)])
    assert rel_l2(got16.float().cpu().numpy(), want16) < 1e-3


def test_flow_match_euler_baseline_matches_oracle():
    """edit_ppo/scheduler_fm.py:405-410 (type == 'euler') on FMPPOScheduler's sigma schedule"""
    from consolver_amd.baselines import FlowMatchEulerBaselineScheduler
    s = FlowMatchEulerBaselineScheduler(shift=3.0, use_dynamic_shifting=True)
    n = 6
    sig = np.linspace(1.0, 1.0 / n, n)
    s.set_timesteps(sigmas=sig, mu=1.15, device=DEV)
    ref = consolver_amd.FMPPOScheduler(shift=3.0, use_dynamic_shifting=True, order_dim=2, scaler_dim=0, mu_dim=0,
                                       factor_net_kwargs=dict(hidden_dim=4, num_actions=3))
    ref.set_timesteps(sigmas=sig, mu=1.15, device=DEV)
    assert torch.equal(s.sigmas, ref.sigmas) and torch.equal(s.timesteps, ref.timesteps)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 16, 64)).astype(np.float32)
    xg, xo = cu(x), x
    sg = s.sigmas.cpu().numpy()
    for i, t in enumerate(s.timesteps):
        v = (0.7 * np.tanh(xo) - 0.1).astype(np.float32)
        xg = s.step(cu(v), t, xg, return_dict=False)[0]
        xo = (xo + np.float32(sg[i + 1] - sg[i]) * v).astype(np.float32)
        assert rel_l2(xg.cpu().numpy(), xo) < 1e-6
