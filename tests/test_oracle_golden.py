"""Pins oracle/solver_oracle.py (numpy restatement) against golden vectors that
oracle/make_golden.py produced by running the imported reference (CPU)."""
import numpy as np
import pytest

from oracle import solver_oracle as so

SPACINGS = ["trailing", "leading", "linspace"]


def weights(npz, prefix):
    return {k[len(prefix):]: npz[k] for k in npz.files if k.startswith(prefix)}


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# ---------------------------------------------------------------- tables
def test_alphas_cumprod_tables(golden):
    g = golden["sd_tables"]
    for key, kw in [("ac_scaled_linear", dict(beta_schedule="scaled_linear", beta_start=0.00085, beta_end=0.012)),
                    ("ac_linear", dict(beta_schedule="linear")),
                    ("ac_cos", dict(beta_schedule="squaredcos_cap_v2"))]:
        ac = so.alphas_cumprod(so.make_betas(**kw))
        # float table: ATen's vectorised linspace differs from the scalar formula by <= 1 ulp
        np.testing.assert_allclose(ac, g[key], rtol=2e-6, atol=0)
    ac = g["ac_scaled_linear"]
    assert abs(float(ac[999]) - 0.00466009508818388) < 1e-9
    assert abs(float(ac[0]) - 0.9991499781608582) < 1e-7


@pytest.mark.parametrize("spacing,off", [("trailing", 0), ("leading", 0), ("leading", 1), ("linspace", 0)])
def test_timestep_grids_bit_exact(golden, spacing, off):
    g = golden["sd_tables"]
    flat, offs = g[f"ts_{spacing}_off{off}"], g[f"ts_{spacing}_off{off}_offsets"]
    for n in range(1, 51):
        want = flat[offs[n - 1]:offs[n]]
        got = so.sd_timesteps(n, 1000, spacing, off)
        assert got.dtype == np.int64 and np.array_equal(got, want), (spacing, n)


def test_trailing_quirks(golden):
    g = golden["sd_tables"]
    assert np.array_equal(so.sd_timesteps(16, 1000, "trailing"),
                          [999, 937, 874, 811, 749, 687, 624, 561, 499, 437, 374, 311, 249, 187, 124, 61])
    got = so.sd_timesteps(61, 1000, "trailing")
    assert np.array_equal(got, g["ts_trailing_n61"]) and len(got) == 62 and got[-1] == -1
    assert so.sd_prev_timestep(999, 6) == 833  # floor division, not the next grid entry
    with pytest.raises(ValueError):
        so.sd_timesteps(1001)


# ---------------------------------------------------------------- factor net
def test_factor_net_sd(golden):
    g = golden["sd_factor_net"]
    for ci, (o, sc, uc, K, H) in enumerate(g["cases"]):
        w = weights(g, f"c{ci}_w_")
        np.testing.assert_allclose(so.action_values_sd(o, sc, K), w["action_values"], rtol=0, atol=2e-7)
        probs = so.factor_net_probs(w, g[f"c{ci}_x"], g[f"c{ci}_eps"], variant="sd", use_conv=bool(uc))
        np.testing.assert_allclose(probs, g[f"c{ci}_probs"], rtol=2e-4, atol=1e-6)
        actions, aprobs = so.gather_actions(probs, w["action_values"], g[f"c{ci}_idx"])
        np.testing.assert_array_equal(actions, g[f"c{ci}_actions"])
        np.testing.assert_allclose(aprobs, g[f"c{ci}_aprobs"], rtol=2e-4, atol=1e-6)
        idx = so.nearest_bins(g[f"c{ci}_actions"], w["action_values"])
        np.testing.assert_array_equal(idx, g[f"c{ci}_idx"])
        sel = np.take_along_axis(probs, so.nearest_bins(g[f"c{ci}_pert"], w["action_values"])[..., None], 2)[..., 0]
        np.testing.assert_allclose(sel, g[f"c{ci}_sel_pert"], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(so.normalized_entropy(probs), g[f"c{ci}_entropy"], rtol=2e-4, atol=2e-6)


def test_factor_net_flux(golden):
    g = golden["flux"]
    for ci, (o, sc, mu, uc, K, H) in enumerate(g["fn_cases"]):
        w = weights(g, f"f{ci}_w_")
        np.testing.assert_allclose(so.action_values_flux(o, sc, mu, K), w["action_values"], rtol=0, atol=2e-7)
        probs = so.factor_net_probs(w, g[f"f{ci}_x"], g[f"f{ci}_eps"], variant="flux", use_conv=bool(uc))
        # temperature 0.01 amplifies fp32 matmul-order noise 100x
        np.testing.assert_allclose(probs, g[f"f{ci}_probs"], rtol=5e-3, atol=1e-5)
        np.testing.assert_allclose(so.normalized_entropy(g[f"f{ci}_probs"]), g[f"f{ci}_entropy"], rtol=1e-3, atol=1e-5)


# ---------------------------------------------------------------- step trajectories
def test_sd_step_trajectories(golden):
    g = golden["sd_steps"]
    for ci, (o, sc, uc, n, sp, vp) in enumerate(g["cases"]):
        sch = so.PPOSchedulerOracle(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                    timestep_spacing=SPACINGS[sp], steps_offset=1 if sp == 1 else 0,
                                    prediction_type="v_prediction" if vp else "epsilon",
                                    order_dim=int(o), scaler_dim=int(sc), use_conv=bool(uc), num_actions=11,
                                    weights=weights(g, f"t{ci}_w_"))
        sch.set_timesteps(int(n))
        assert np.array_equal(sch.timesteps, g[f"t{ci}_timesteps"])
        x = g[f"t{ci}_x0"]
        for i, t in enumerate(sch.timesteps):
            out = sch.step(g[f"t{ci}_s{i}_eps"], t, x, g[f"t{ci}_s{i}_idx"])
            np.testing.assert_array_equal(out["masks"], g[f"t{ci}_s{i}_masks"])
            np.testing.assert_array_equal(out["actions"], g[f"t{ci}_s{i}_actions"])
            np.testing.assert_array_equal(out["conds_x"], g[f"t{ci}_s{i}_condx"])
            np.testing.assert_allclose(out["probs"], g[f"t{ci}_s{i}_probs"], rtol=5e-4, atol=1e-6)
            # chained on the reference's own state -> per-step error only
            assert rel_l2(out["prev_sample"], g[f"t{ci}_s{i}_prev"]) < 5e-7, (ci, i)
            x = g[f"t{ci}_s{i}_prev"]
        # free-running trajectory
        sch.set_timesteps(int(n))
        x = g[f"t{ci}_x0"]
        for i, t in enumerate(sch.timesteps):
            x = sch.step(g[f"t{ci}_s{i}_eps"], t, x, g[f"t{ci}_s{i}_idx"])["prev_sample"]
        assert rel_l2(x, g[f"t{ci}_final"]) < 2e-6, ci


def test_sd_fp16_io_vs_fp32_oracle(golden):
    """the reference's fp16 chain vs the fp32 oracle on the same fp16 inputs: the 1e-3 gate."""
    g = golden["sd_steps"]
    sch = so.PPOSchedulerOracle(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                timestep_spacing="trailing", order_dim=4, scaler_dim=0, num_actions=11,
                                weights=weights(g, "h_w_"))
    sch.set_timesteps(4)
    x = g["h_x0"].astype(np.float32)
    for i, t in enumerate(sch.timesteps):
        out = sch.step(g[f"h_s{i}_eps"].astype(np.float32), t, x, g[f"h_s{i}_idx"], cond_dtype="f16")
        assert rel_l2(out["prev_sample"], g[f"h_s{i}_prev"]) < 1e-3
        x = so.round_f16(g[f"h_s{i}_prev"])
    # reference promotes to fp32 from step 2 on (fp32 net, fp16 eps)
    assert str(g["h_s0_prev_dtype"]) == "torch.float16" and str(g["h_s1_prev_dtype"]) == "torch.float32"


def test_sd_rollout_records(golden):
    g = golden["sd_rollout"]
    for ri in range(3):
        o, sc, uc, n = [int(v) for v in g[f"r{ri}_cfg"]]
        cfg = float(g[f"r{ri}_guidance"])
        sch = so.PPOSchedulerOracle(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                    timestep_spacing="trailing", order_dim=o, scaler_dim=sc, use_conv=bool(uc),
                                    num_actions=11, weights=weights(g, f"r{ri}_w_"))
        pe, ne = g[f"r{ri}_prompt_embeds"], g[f"r{ri}_neg_embeds"]
        ctx = np.concatenate([ne, pe]) if cfg > 1 else pe

        def eps_model(lat_in, t):
            from oracle.make_golden import eps_model_np
            c = ctx.mean(axis=(1, 2)).reshape(-1, 1, 1, 1)
            return eps_model_np(lat_in, t, g[f"r{ri}_unet_noise_{t}"]) + (0.1 * c).astype(np.float32)

        lat, conds, probs, actions, masks = so.sd_rollout(sch, eps_model, g[f"r{ri}_noise"], n, cfg, g[f"r{ri}_idx"])
        assert conds["x"].shape == g[f"r{ri}_conds_x"].shape == (2, n - 1, 2)
        assert conds["epsilon"].shape == g[f"r{ri}_conds_eps"].shape
        np.testing.assert_array_equal(conds["x"], g[f"r{ri}_conds_x"])
        np.testing.assert_array_equal(actions, g[f"r{ri}_actions"])
        np.testing.assert_array_equal(masks, g[f"r{ri}_masks"])
        np.testing.assert_allclose(probs, g[f"r{ri}_probs"], rtol=5e-4, atol=1e-6)
        assert rel_l2(conds["epsilon"], g[f"r{ri}_conds_eps"]) < 2e-6
        assert rel_l2(lat, g[f"r{ri}_latents"]) < 2e-6


# ---------------------------------------------------------------- FLUX
def test_flux_sigma_tables(golden):
    g = golden["flux"]
    for n in range(2, 9):
        sig, ts = so.flux_sigmas(sigmas=np.linspace(1.0, 1 / n, n), mu=1.15, shift=3.0, use_dynamic_shifting=True)
        np.testing.assert_allclose(sig, g[f"sig_dyn_n{n}"], rtol=2e-7, atol=0)
        np.testing.assert_allclose(ts, g[f"ts_dyn_n{n}"], rtol=2e-7, atol=0)
        sig, ts = so.flux_sigmas(n, shift=3.0, use_dynamic_shifting=False)
        np.testing.assert_allclose(sig, g[f"sig_static_n{n}"], rtol=2e-7, atol=0)
        np.testing.assert_allclose(ts, g[f"ts_static_n{n}"], rtol=2e-7, atol=0)
    for mu in (0.5, 0.8, 1.15):
        sig, _ = so.flux_sigmas(sigmas=np.linspace(1.0, 1 / 5, 5), mu=mu, use_dynamic_shifting=True)
        np.testing.assert_allclose(sig, g[f"sig_mu{mu}"], rtol=2e-7, atol=0)
    assert abs(so.calculate_shift(4096) - 1.15) < 1e-12 and abs(so.calculate_shift(256) - 0.5) < 1e-12
    with pytest.raises(ValueError):
        so.flux_sigmas(sigmas=[1.0, 0.5], use_dynamic_shifting=True)


def test_flux_step_trajectories(golden):
    g = golden["flux"]
    for ci, (o, sc, mu, uc, n, bf) in enumerate(g["t_cases"]):
        io = "bf16" if bf else "f32"
        sch = so.FMPPOSchedulerOracle(shift=3.0, use_dynamic_shifting=True, order_dim=int(o), scaler_dim=int(sc),
                                      mu_dim=int(mu), use_conv=bool(uc), num_actions=11,
                                      weights=weights(g, f"t{ci}_w_"))
        sch.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=1.15)
        np.testing.assert_allclose(sch.sigmas, g[f"t{ci}_sigmas"], rtol=2e-7)
        x = g[f"t{ci}_x0"]
        for i in range(int(n)):
            out = sch.step(g[f"t{ci}_s{i}_v"], x, g[f"t{ci}_s{i}_idx"], io_dtype=io)
            np.testing.assert_array_equal(out["masks"], g[f"t{ci}_s{i}_masks"])
            np.testing.assert_array_equal(out["actions"], g[f"t{ci}_s{i}_actions"])
            np.testing.assert_array_equal(out["conds_x"], g[f"t{ci}_s{i}_condx"])
            want = g[f"t{ci}_s{i}_prev"]
            if bf:   # identical fp32 value rounded once to bf16: allow 1 bf16 ulp on a handful of ties
                assert rel_l2(out["prev_sample"], want) < 2e-3
                assert np.mean(out["prev_sample"] != want) < 0.02
            else:
                assert rel_l2(out["prev_sample"], want) < 5e-7
            x = want
        assert sch.step_index == n


# ---------------------------------------------------------------- reward / sharding
def test_psnr_known_answers():
    a = np.random.default_rng(0).random((3, 3, 16, 16)).astype(np.float32)
    r = so.image_psnr_reward(a, a)
    assert r.shape == (3, 1) and np.allclose(r, 80.0, atol=1e-3)       # 10*log10(1/1e-8)
    for delta in (0.1, 0.01):
        r = so.image_psnr_reward(np.full_like(a, 0.5), np.full_like(a, 0.5 + delta))
        assert np.allclose(r, -20 * np.log10(delta), atol=2e-3)
    assert np.all(so.image_psnr_reward(np.zeros_like(a), np.ones_like(a) * 2) == 0.0)   # clamp min 0
    assert np.allclose(so.depth_psnr_tail(a[:, 0], a[:, 0]), 80.0, atol=1e-3)


@pytest.mark.parametrize("n,world", [(5000, 8), (128, 8), (7, 8), (1001, 3), (16, 1)])
def test_shard_rules(n, world):
    spans = [so.shard_bounds(n, world, r) for r in range(world)]      # gen_ppo.py:349-357
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert all(e - s == n // world for s, e in spans[:-1])
    spans = [so.shard_bounds_ceil(n, world, r) for r in range(world)]  # generate_ours.py:176-177
    assert sum(e - s for s, e in spans) == n
    assert all(e - s <= -(-n // world) for s, e in spans)


def test_ppo_policy_update_vs_reference_autograd(golden):
    """train_ppo.py:404-437: the oracle's hand-derived gradients, clip_grad_norm_ and AdamW restatements against the
    reference's FactorNetPPO under torch autograd + torch.optim.AdamW (two optimisation epochs on one batch; plain,
    use_conv and single-row batches; ratios on both sides of the clip range)."""
    g = golden["sd_ppo_update"]
    for ui in range(3):
        o, sc, uc, K, H, R = [int(v) for v in g[f"u{ui}_cfg"]]
        lr, b1, b2, wd, eps, clip_range, entropy_coef, max_norm = [float(v) for v in g[f"u{ui}_hyper"]]
        w = weights(g, f"u{ui}_w_")
        eps_stack = g[f"u{ui}_eps"] if uc else None
        state = {}
        for ep in range(2):
            loss, grads = so.ppo_policy_grads(w, g[f"u{ui}_x"], g[f"u{ui}_actions"], g[f"u{ui}_old_probs"], g[f"u{ui}_adv"],
                                              eps_stack=eps_stack, use_conv=bool(uc), clip_range=clip_range, entropy_coef=entropy_coef)
            assert abs(loss - float(g[f"u{ui}_e{ep}_loss"])) < 2e-5 * max(1.0, abs(loss))
            for k, gr in grads.items():
                want = g[f"u{ui}_e{ep}_grad_{k}"]
                assert gr.shape == want.shape
                assert rel_l2(gr, want) < 2e-5, (ui, ep, k, rel_l2(gr, want))
            total, clipped = so.clip_grad_norm(grads, max_norm)
            assert abs(total - float(g[f"u{ui}_e{ep}_norm"])) < 2e-5 * max(1.0, total)
            w_new = so.adamw_step({k: v for k, v in w.items()}, clipped, state, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
            for k in grads:
                want = g[f"u{ui}_e{ep}_after_{k}"]
                assert np.abs(w_new[k] - want).max() < 2e-6 + 2e-5 * np.abs(want).max(), (ui, ep, k)
            np.testing.assert_array_equal(w_new["action_values"], g[f"u{ui}_e{ep}_after_action_values"])   # buffer, not a parameter
            w = w_new


def test_clip_text_oracle_vs_transformers(golden):
    """oracle/clip_oracle.py against the installed third-party transformers.CLIPTextModel (reduced config, seeded weights)"""
    import torch
    from oracle.clip_oracle import ClipTextOracle, clip_manifest
    g = golden["clip_text"]
    V, D, I, NL, H, P = [int(v) for v in g["cfg"]]
    cfg = dict(vocab_size=V, hidden_size=D, intermediate_size=I, num_hidden_layers=NL, num_attention_heads=H, max_position_embeddings=P)
    sd = {k[2:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("w_")}
    assert sorted(sd.keys()) == sorted(n for n, _ in clip_manifest(cfg))
    for n, shape in clip_manifest(cfg):
        assert tuple(sd[n].shape) == shape
    orc = ClipTextOracle(sd, cfg, round_weights_to_f16=False)
    for name in ("full", "short"):
        ids = torch.from_numpy(np.asarray(g[f"{name}_ids"]))
        out = orc(ids)[0].numpy()
        assert rel_l2(out, g[f"{name}_out"]) < 2e-6
        pooled = out[np.arange(ids.shape[0]), ids.numpy().argmax(-1)]            # end-of-text position
        assert rel_l2(pooled, g[f"{name}_pooled"]) < 2e-6


def test_t5_encoder_oracle_vs_transformers(golden):
    """oracle/t5_oracle.py against the installed third-party transformers.T5EncoderModel (reduced config, seeded weights,
    sequences inside and beyond the 128-position bucket range)"""
    import torch
    from oracle.t5_oracle import T5EncoderOracle, t5_manifest
    g = golden["t5_encoder"]
    V, D, dk, H, I, NL, NB, MD = [int(v) for v in g["cfg"]]
    cfg = dict(vocab_size=V, d_model=D, d_kv=dk, num_heads=H, d_ff=I, num_layers=NL, relative_attention_num_buckets=NB,
               relative_attention_max_distance=MD)
    sd = {k[2:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("w_")}
    assert sorted(sd.keys()) == sorted(n for n, _ in t5_manifest(cfg))
    orc = T5EncoderOracle(sd, cfg, round_weights_to_bf16=False)
    for name in ("short", "long"):
        out = orc(torch.from_numpy(np.asarray(g[f"{name}_ids"])))[0].numpy()
        assert rel_l2(out, g[f"{name}_out"]) < 3e-6


def test_flux_rollout_records(golden):
    """a18: the oracle's restatement of the FLUX rollout loop against the 6-tuple the IMPORTED reference function
    (edit_ppo/denoise_diffusion.py:11-176) returned for the closed-form stub pipe (oracle/flux_stub_pipe.py)."""
    import torch
    from oracle.flux_stub_pipe import StubKontextPipe, velocity_np
    g = golden["flux_rollout"]
    for ci, (o, sc, mu, n, B) in enumerate(g["cases"]):
        o, sc, mu, n, B = int(o), int(sc), int(mu), int(n), int(B)
        gs = float(g[f"c{ci}_guidance"])
        pipe = StubKontextPipe()
        text = ["make it red", "mi355x"][:B]
        pe, pooled, _ = pipe.encode_prompt(prompt=text, device="cpu")
        noise = torch.from_numpy(g[f"c{ci}_noise"]).to(torch.bfloat16)
        packed = pipe._pack_latents(noise, B, 16, 0, 0)
        _, il, lat_ids, img_ids = pipe.prepare_latents(image=torch.from_numpy(g[f"c{ci}_image"]), batch_size=B, dtype=torch.bfloat16,
                                                       device="cpu", latents=packed)
        ids = torch.cat([lat_ids, img_ids]).float().numpy()
        sch = so.FMPPOSchedulerOracle(shift=3.0, use_dynamic_shifting=True, order_dim=o, scaler_dim=sc, mu_dim=mu,
                                      num_actions=11, weights=weights(g, f"c{ci}_w_"))
        guidance = np.full((B,), gs, np.float32)

        def v_model(h, ts):
            return velocity_np(h, ts, guidance, pooled.float().numpy(), pe.float().numpy(), ids)

        lat, conds, probs, actions, masks, seen = so.flux_rollout(sch, v_model, packed.float().numpy(), il.float().numpy(), n,
                                                                  g[f"c{ci}_idx"], io_dtype="bf16")
        np.testing.assert_allclose(sch.sigmas, g[f"c{ci}_sigmas"], rtol=2e-7)
        np.testing.assert_array_equal(seen, g[f"c{ci}_timestep_seen"])            # t.to(bf16) / 1000 in bf16
        np.testing.assert_array_equal(conds["x"], g[f"c{ci}_conds_x"])
        np.testing.assert_array_equal(actions, g[f"c{ci}_actions"])
        np.testing.assert_array_equal(masks, g[f"c{ci}_masks"])
        np.testing.assert_allclose(probs, g[f"c{ci}_probs"], rtol=5e-3, atol=2e-5)
        assert conds["epsilon"].shape == g[f"c{ci}_conds_eps"].shape == (B, n - 1, o, 16, 64)
        # bf16 trajectories: equal up to 1 bf16 ulp on a small fraction of elements (tanh / reduction order of the stub DiT)
        for got, want in ((conds["epsilon"], g[f"c{ci}_conds_eps"]), (lat, g[f"c{ci}_latents"])):
            assert rel_l2(got, want) < 2e-3 and np.mean(got != want) < 0.02, (ci, rel_l2(got, want), np.mean(got != want))
