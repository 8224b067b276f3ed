"""PPO rollout consumer (config 5; train_ppo.py:352-427) through the C ABI vs the numpy oracle:
image-PSNR reward (edit_ppo/reward_model.py:484-509) incl. the known answers of SURVEY 8(c), the depth PSNR tail
(:404-422), advantage normalisation (train_ppo.py:376-390), the clipped-surrogate loss value (:408-421) and the
end-to-end collect_rollout on reduced SD1.5 / VAE networks.

Tolerances: fp32 reductions in a different summation order than numpy -> 1e-5 relative on the MSE = 5e-5 dB.
"""
import numpy as np
import pytest
import torch

import consolver_amd
from consolver_amd import ppo
from oracle import solver_oracle as so

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cu(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a)).to(DEV, dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(4, 3, 64, 64), (3, 3, 20, 24), (1, 3, 512, 512), (2, 3, 8, 12)])
def test_image_psnr_reward_matches_oracle(shape, dtype):
    rng = np.random.default_rng(7)
    pred = torch.from_numpy(rng.random(shape, dtype=np.float32)).to(dtype)
    tgt = (pred.float() + torch.from_numpy(rng.normal(0, 0.05, shape).astype(np.float32))).clamp(0, 1).to(dtype)
    got = ppo.calculate_reward("image_psnr", None, None, pred.to(DEV), tgt.to(DEV), DEV)
    want = so.image_psnr_reward(pred.float().numpy(), tgt.float().numpy())
    assert got.shape == (shape[0], 1) and got.dtype == torch.float32
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=2e-4)


def test_image_psnr_known_answers():
    x = torch.rand(3, 3, 32, 32)
    # pred == target -> 10 log10(1 / 1e-8) = 80
    got = ppo.calculate_image_psnr_reward(None, x.to(DEV), x.to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), np.full((3, 1), 80.0, np.float32), atol=1e-4)
    # constant offset d -> -20 log10 d
    for d in (0.5, 0.1, 0.01):
        a = torch.full((2, 3, 16, 16), 0.25)
        got = ppo.calculate_image_psnr_reward(None, (a + d).to(DEV), a.to(DEV))
        np.testing.assert_allclose(got.cpu().numpy(), np.full((2, 1), -20 * np.log10(d), np.float32), atol=2e-3)
    # images further apart than 1 -> negative PSNR clamps to 0; depth tail has no upper clamp
    got = ppo.calculate_image_psnr_reward(None, torch.full((1, 3, 8, 8), 3.0).to(DEV), torch.zeros(1, 3, 8, 8).to(DEV))
    assert float(got) == 0.0
    dm = torch.rand(2, 24, 40)
    np.testing.assert_allclose(ppo.depth_psnr_tail(dm.to(DEV), dm.to(DEV)).cpu().numpy(), so.depth_psnr_tail(dm.numpy(), dm.numpy()), atol=1e-4)
    dn = (dm + 0.1 * torch.rand_like(dm))
    np.testing.assert_allclose(ppo.depth_psnr_tail(dn.to(DEV), dm.to(DEV)).cpu().numpy(), so.depth_psnr_tail(dn.numpy(), dm.numpy()), atol=2e-4)
    # edge cases: empty batch, shape mismatch, backbone rewards are out of scope, unknown type
    assert ppo.calculate_image_psnr_reward(None, x[:0].to(DEV), x[:0].to(DEV)).shape == (0, 1)
    with pytest.raises(ValueError):
        ppo.calculate_image_psnr_reward(None, x.to(DEV), x[:, :, :16].to(DEV))
    with pytest.raises(NotImplementedError):
        ppo.calculate_reward("clip", None, None, x.to(DEV), x.to(DEV), DEV)
    with pytest.raises(ValueError):
        ppo.calculate_reward("nope", None, None, x.to(DEV), x.to(DEV), DEV)


@pytest.mark.parametrize("B,n,A", [(80, 8, 3), (5, 2, 5), (2, 15, 1), (300, 4, 3)])
def test_advantages_match_oracle(B, n, A):
    rng = np.random.default_rng(B)
    r = rng.normal(20, 4, (B, 1)).astype(np.float32)
    masks = (rng.random((B * (n - 1), A)) > 0.3).astype(np.float32)
    got = ppo.compute_advantages(cu(r), cu(masks).reshape(B, n - 1, A), n)
    want = so.ppo_advantages(r, masks, n)
    assert got.shape == (B * (n - 1), A)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=2e-5)


def test_advantages_single_trajectory_is_nan_like_torch_std():
    got = ppo.compute_advantages(cu([[3.0]]), cu(np.ones((3, 2))), 4)
    assert torch.isnan(got).all()


@pytest.mark.parametrize("R,A", [(560, 3), (7, 1), (1000, 5)])
def test_ppo_loss_matches_oracle(R, A):
    rng = np.random.default_rng(R)
    cur = rng.random((R, A)).astype(np.float32) * 0.9 + 0.05
    old = np.clip(cur * rng.uniform(0.6, 1.5, (R, A)), 1e-3, 1).astype(np.float32)
    ent = rng.random((R, A)).astype(np.float32)
    adv = (rng.normal(0, 10, (R, 1)) * (rng.random((R, A)) > 0.2)).astype(np.float32)
    got = ppo.ppo_loss(cu(cur), cu(old), cu(ent), cu(adv), clip_range=0.2, entropy_coef=0.01)
    want = so.ppo_loss(cur, old, ent, adv, 0.2, 0.01)
    assert abs(float(got) - float(want)) < 1e-4 * max(1.0, abs(float(want)))


def test_collect_rollout_reduced_networks():
    """train_ppo.py:352-403 on reduced SD1.5 / VAE networks: rollout records, decode of prediction and teacher latents,
    reward, advantages; every stage against the oracle applied to the product's own upstream tensors, plus the whole
    chain against the all-oracle chain within the fp16 pipeline's drift."""
    from consolver_amd.unet import HipUNet2DConditionModel
    from consolver_amd.vae import HipAutoencoderKL
    from consolver_amd.synth import synthetic_unet_state_dict, synthetic_vae_state_dict, synthetic_prompt_embeds
    from oracle.unet_oracle import UNetOracle
    from oracle import vae_oracle
    from tests._models import get_unet
    unet, usd = get_unet(dict(layers_per_block=1, sample_size=16), seed=5)
    vae = HipAutoencoderKL(dict(layers_per_block=1, sample_size=16), device=DEV)
    vsd = synthetic_vae_state_dict(vae.manifest(), seed=6)
    vae.load_state_dict(vsd)
    sch = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing",
                                     order_dim=4, scaler_dim=0, factor_net_kwargs=dict(hidden_dim=32, num_actions=11))
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in sch.factor_net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    w = {k: v.numpy().copy() for k, v in sch.factor_net.state_dict().items()}
    sch.factor_net.to(DEV)
    B, n, cfg = 3, 4, 3.0
    rng = np.random.default_rng(2)
    idx = rng.integers(0, 11, size=(n, B, 3))
    sch.factor_net.forced_action_idx = [cu(i, torch.int64) for i in idx]
    pe, ne = synthetic_prompt_embeds(B, seed=1001).half(), synthetic_prompt_embeds(B, seed=1002).half()
    noise = torch.randn(B, 4, 16, 16, generator=g).half()
    teacher = (torch.randn(B, 4, 16, 16, generator=g) * 0.18215).half()
    out = ppo.collect_rollout(None, sch, unet, vae, noise.to(DEV), ["a"] * B, None, teacher.to(DEV), cfg=cfg, num_inference_steps=n,
                              decode_batch_size=2, prompt_embeds=pe.to(DEV), negative_prompt_embeds=ne.to(DEV))
    R = B * (n - 1)
    assert out["conds"]["x"].shape == (R, 2) and out["conds"]["epsilon"].shape == (R, 4, 4, 16, 16)
    assert out["actions"].shape == (R, 3) and out["probs"].shape == (R, 3) and out["masks"].shape == (R, 3)
    assert out["advantages"].shape == (R, 3) and out["rewards"].shape == (B, 1)
    # stage checks on the product's own tensors
    vo = vae_oracle.VaeOracle(vsd, vars(vae.config))
    pred_img = vae_oracle.decode_latents(vo, out["model_pred"].float().cpu(), 2)
    tgt_img = vae_oracle.decode_latents(vo, teacher.float(), 2)
    np.testing.assert_allclose(out["rewards"].cpu().numpy(), so.image_psnr_reward(pred_img.numpy(), tgt_img.numpy()), atol=0.05)
    np.testing.assert_allclose(out["advantages"].cpu().numpy(),
                               so.ppo_advantages(out["rewards"].cpu().numpy(), out["masks"].cpu().numpy(), n), rtol=1e-4, atol=1e-4)
    # whole chain, all oracle
    uo = UNetOracle(usd, unet.config)
    so_s = so.PPOSchedulerOracle(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing",
                                 order_dim=4, scaler_dim=0, num_actions=11, weights=w)
    ctx = torch.cat([ne, pe]).float()
    eps_model = lambda lat, t: uo(torch.from_numpy(lat), int(t), ctx).numpy()
    lat_o, conds_o, probs_o, actions_o, masks_o = so.sd_rollout(so_s, eps_model, noise.float().numpy(), n, cfg, idx, cond_dtype="f16")
    np.testing.assert_array_equal(out["actions"].cpu().numpy(), actions_o.reshape(R, -1))
    np.testing.assert_array_equal(out["masks"].cpu().numpy(), masks_o.reshape(R, -1))
    np.testing.assert_array_equal(out["conds"]["x"].cpu().numpy(), conds_o["x"].reshape(R, 2))
    rel = np.linalg.norm(out["model_pred"].float().cpu().numpy() - lat_o) / np.linalg.norm(lat_o)
    assert rel < 1e-2, rel
    r_o = so.image_psnr_reward(vae_oracle.decode_latents(vo, torch.from_numpy(lat_o), 2).numpy(), tgt_img.numpy())
    np.testing.assert_allclose(out["rewards"].cpu().numpy(), r_o, atol=0.2)
    # the loss value of the collected batch under the same policy: ratio = 1 -> -mean(adv) - coef * mean(entropy)
    cur, ent = sch.factor_net(out["conds"], out["actions"])
    loss = ppo.ppo_loss(cur, out["probs"], ent, out["advantages"])
    want = so.ppo_loss(cur.cpu().numpy(), out["probs"].cpu().numpy(), ent.cpu().numpy(), out["advantages"].cpu().numpy())
    assert abs(float(loss) - float(want)) < 1e-4 * max(1.0, abs(float(want)))


def _trainer_from_golden(g, ui):
    o, sc, uc, K, H, R = [int(v) for v in g[f"u{ui}_cfg"]]
    net = consolver_amd.FactorNetPPO(hidden_dim=H, num_actions=K, order_dim=o, scaler_dim=sc, use_conv=bool(uc))
    sd = {k[len(f"u{ui}_w_"):]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith(f"u{ui}_w_")}
    net.load_state_dict(sd)
    net.to(DEV)
    lr, b1, b2, wd, eps, clip_range, entropy_coef, max_norm = [float(v) for v in g[f"u{ui}_hyper"]]
    tr = ppo.PolicyTrainer(net, lr=lr, betas=(b1, b2), weight_decay=wd, eps=eps, max_grad_norm=max_norm, clip_range=clip_range,
                           entropy_coef=entropy_coef)
    conds = {"x": cu(g[f"u{ui}_x"])}
    if uc:
        conds["epsilon"] = cu(g[f"u{ui}_eps"])
    return net, tr, conds, (o, sc, uc, K, H, R)


def test_policy_update_matches_reference_autograd_and_adamw(golden):
    """train_ppo.py:404-437 on the HIP library against the golden vectors produced by the reference's FactorNetPPO under
    torch autograd, clip_grad_norm_ and torch.optim.AdamW (two epochs on one batch; plain, use_conv, single-row)."""
    g = golden["sd_ppo_update"]
    for ui in range(3):
        net, tr, conds, _ = _trainer_from_golden(g, ui)
        actions, old, adv = cu(g[f"u{ui}_actions"]), cu(g[f"u{ui}_old_probs"]), cu(g[f"u{ui}_adv"])
        for ep in range(2):
            loss = float(tr.compute_grads(conds, actions, old, adv))
            want_loss = float(g[f"u{ui}_e{ep}_loss"])
            assert abs(loss - want_loss) < 5e-5 * max(1.0, abs(want_loss)), (ui, ep, loss, want_loss)
            for k, gv in tr.grad_views().items():
                want = g[f"u{ui}_e{ep}_grad_{k}"]
                err = np.linalg.norm(gv.cpu().numpy() - want) / max(np.linalg.norm(want), 1e-12)
                assert err < 1e-4, (ui, ep, k, err)
            loss2, norm = tr.step(conds, actions, old, adv)
            assert abs(float(loss2) - want_loss) < 5e-5 * max(1.0, abs(want_loss))
            assert abs(float(norm) - float(g[f"u{ui}_e{ep}_norm"])) < 1e-4 * max(1.0, float(norm))
            for k, v in net.state_dict().items():
                want = g[f"u{ui}_e{ep}_after_{k}"]
                assert np.abs(v.cpu().numpy() - want).max() < 3e-6 + 5e-5 * np.abs(want).max(), (ui, ep, k)


def test_policy_update_at_rollout_size_vs_oracle(tmp_path):
    """config-5 size: B = 80 trajectories x (n - 1) = 7 recorded steps, hidden 256, 11 bins; HIP vs the numpy oracle;
    deterministic; the probabilities the library itself evaluates after the step reflect the in-place update; checkpoint
    round trip in the reference's format."""
    torch.manual_seed(0)
    net = consolver_amd.FactorNetPPO(hidden_dim=256, num_actions=11, order_dim=4, scaler_dim=0)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape) * 0.3 / max(1.0, p.shape[-1] ** 0.5) * 4)
    net.to(DEV)
    w = {k: v.cpu().numpy().copy() for k, v in net.state_dict().items()}
    R, A = 560, 3
    rng = np.random.default_rng(1)
    t = rng.integers(1, 999, size=(R, 1)).astype(np.float32)
    x = np.concatenate([t, np.maximum(t - 125, 0)], 1)
    idx = rng.integers(0, 11, size=(R, A))
    actions = np.take_along_axis(np.broadcast_to(w["action_values"], (R, A, 11)), idx[..., None], 2)[..., 0]
    cur = so.gather_actions(so.factor_net_probs(w, x), w["action_values"], idx)[1]
    old = np.clip(cur * rng.uniform(0.6, 1.5, (R, A)), 1e-4, 1).astype(np.float32)
    adv = (rng.normal(0, 10, (R, 1)) * (rng.random((R, A)) > 0.2)).astype(np.float32)
    tr = ppo.PolicyTrainer(net, lr=1e-4, weight_decay=1e-3)
    conds = {"x": cu(x)}
    loss = float(tr.compute_grads(conds, cu(actions), cu(old), cu(adv)))
    g1 = tr.grads.clone()
    want_loss, want = so.ppo_policy_grads(w, x, actions, old, adv)
    assert abs(loss - want_loss) < 5e-5 * max(1.0, abs(want_loss))
    for k, gv in tr.grad_views().items():
        err = np.linalg.norm(gv.cpu().numpy() - want[k]) / np.linalg.norm(want[k])
        assert err < 1e-4, (k, err)
    tr.compute_grads(conds, cu(actions), cu(old), cu(adv))
    assert torch.equal(g1, tr.grads)                              # no atomics: bit-identical
    _, norm = tr.step(conds, cu(actions), cu(old), cu(adv))
    total, clipped = so.clip_grad_norm(want, 1.0)
    assert abs(float(norm) - total) < 1e-4 * total
    w_new = so.adamw_step(w, clipped, {}, lr=1e-4, weight_decay=1e-3)
    for k, v in net.state_dict().items():
        assert np.abs(v.cpu().numpy() - w_new[k]).max() < 1e-6 + 2e-5 * np.abs(w_new[k]).max(), k
    # the library's own forward sees the updated parameters
    probs = net.forward_(conds).cpu().numpy()
    np.testing.assert_allclose(probs, so.factor_net_probs(w_new, x), rtol=2e-4, atol=2e-6)
    d = tr.save_checkpoint(str(tmp_path), 7)
    assert d.endswith("checkpoint-7") and sorted(torch.load(d + "/model.ckpt").keys()) == sorted(net.state_dict().keys())
    net2 = consolver_amd.FactorNetPPO(hidden_dim=256, num_actions=11, order_dim=4, scaler_dim=0).to(DEV)
    ppo.PolicyTrainer(net2).load_checkpoint(d)
    for k, v in net2.state_dict().items():
        assert torch.equal(v, net.state_dict()[k])


def test_score_image_pairs_on_gpu(tmp_path):
    """compute_reward.py pair scoring reduced to the arithmetic-only reward: PNG files -> GPU PSNR -> statistics JSON"""
    from consolver_amd import evaluation as ev
    g = torch.Generator().manual_seed(3)
    want = []
    for i in range(5):
        a = torch.rand(3, 32, 32, generator=g)
        b = (a + 0.05 * torch.randn(3, 32, 32, generator=g)).clamp(0, 1)
        ev.save_generation(str(tmp_path / "a"), 0, i, a, "p")
        ev.save_generation(str(tmp_path / "b"), 0, i, b, "p")
        qa, qb = (a * 255).round() / 255, (b * 255).round() / 255
        want.append(float(so.image_psnr_reward(qa[None].numpy(), qb[None].numpy())[0, 0]))
    pairs = ev.find_image_pairs(str(tmp_path / "a"), str(tmp_path / "b"))
    res = ev.score_image_pairs(pairs, batch_size=2, device=DEV)
    np.testing.assert_allclose(res["image_psnr"], want, atol=2e-4)
    out = ev.write_results(str(tmp_path / "out.json"), res, {"num_pairs": len(pairs)})
    assert out["statistics"]["image_psnr"]["count"] == 5 and abs(out["statistics"]["image_psnr"]["mean"] - np.mean(want)) < 2e-4


def test_train_iteration_reduced_networks_improves_reward_direction():
    """train_ppo.py:322-437 in one call on reduced networks: random sample repeat, rollout, decode, reward, advantages, two PPO
    epochs; the policy parameters move, the loss is finite, and the same iteration with a 1-rank gloo-free ``dist=None`` path is
    deterministic given the same forced actions."""
    from consolver_amd.unet import HipUNet2DConditionModel
    from consolver_amd.vae import HipAutoencoderKL
    from consolver_amd.synth import synthetic_unet_state_dict, synthetic_vae_state_dict, synthetic_prompt_embeds
    import random
    from tests._models import get_unet
    unet, _ = get_unet(dict(layers_per_block=1, sample_size=16), seed=5)
    vae = HipAutoencoderKL(dict(layers_per_block=1, sample_size=16), device=DEV)
    vae.load_state_dict(synthetic_vae_state_dict(vae.manifest(), seed=6))
    sch = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing",
                                     order_dim=4, scaler_dim=0, factor_net_kwargs=dict(hidden_dim=32, num_actions=11))
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in sch.factor_net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    sch.factor_net.to(DEV)
    sch.factor_net.sampler = "inverse_cdf"
    B = 6
    batch = ([f"p{i}" for i in range(B)], torch.randn(B, 4, 16, 16, generator=g).half().to(DEV),
             (torch.randn(B, 4, 16, 16, generator=g) * 0.18215).half().to(DEV))
    pe, ne = synthetic_prompt_embeds(B, seed=1001).half().to(DEV), synthetic_prompt_embeds(B, seed=1002).half().to(DEV)
    before = [p.detach().clone() for p in sch.factor_net.parameters()]
    tr = ppo.PolicyTrainer(sch.factor_net, lr=1e-3)
    out = ppo.train_iteration(tr, None, sch, unet, vae, batch, None, cfg=3.0, ppo_epochs=2, prompt_embeds=pe, negative_prompt_embeds=ne,
                              rng=random.Random(0))
    assert 2 <= out["num_inference_steps"] <= 15 and torch.isfinite(out["loss"]) and torch.isfinite(out["norm"]) and torch.isfinite(out["reward"])
    assert tr.step_count == 2
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, sch.factor_net.parameters()))
