"""End-to-end latent parity of configs[1] (SD1.5 + PPOScheduler, 8 steps, CFG 3, fp16) and where its error comes from.

north_star gate: final latents within 1e-3 relative L2 of the reference on identical seeds / prompts (GATE below).  The
reference pipeline is itself fp16 (gen_ppo.py:193-195): fp16 weights, fp16 activation storage between torch ops,
fp32 accumulation inside the vendor kernels; its own distance from the fp32 oracle is 1.9e-3.  The HIP engine has two
residual-stream storage modes (include/consolver_hip.h):
  * residual="f16x2" (split-fp16 stream, hi + lo planes): asserted AGAINST THE GATE, final latents <= 1.0e-3;
  * residual="f16"   (one plane, the reference's own arithmetic class): 1.26e-3 measured; it cannot meet the gate (every add onto the
    stream rounds it: tools/sim_precision.py reproduces the executor's 1.56e-3 per forward from the rounding points alone), so its
    assertions are "measured + 10 %" regression bounds and the gate is printed next to them.
These tests measure

  (a) the HIP engine's 8-step trajectory on the FULL SD1.5 UNet against UNetOracle + PPOSchedulerOracle (fp32, CPU)
      with replayed action indices, with the per-step drift and the teacher-forced per-forward eps error printed;
  (b) the SAME restatement run as a plain torch-fp16 graph on the GPU (in this test only) against the same oracle,
      and assert  HIP-vs-oracle <= 1.25 x torch-fp16-vs-oracle  per forward;
  (c) the per-forward error attributed to kernel classes: the torch-fp16 graph with ONE op class replaced by the HIP
      kernel of that class (through the op-level C ABI) -- the table recorded in DESIGN.md section 3.

The torch-fp16 graph is a comparator, not an oracle: nothing is checked against it except the error budget.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import consolver_amd
from consolver_amd import ops
from consolver_amd.engine import SDSamplingEngine
from consolver_amd.synth import synthetic_prompt_embeds, synthetic_unet_state_dict
from consolver_amd.unet import HipUNet2DConditionModel
from oracle import solver_oracle as so
from oracle.unet_oracle import UNetOracle
from tests._models import get_unet, get_oracle, drop

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GATE = 1.0e-3          # north_star: latents within 1e-3 relative fp32 of the reference
FWD_X2_BOUND = 1.0e-3   # per-forward eps error of the f16x2 stream on random latents: measured 0.857e-3 / 0.910e-3 / 0.873e-3 at t = 999 / 499 / 124, + 10 % (the 8-step latents are what the gate is on)


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a)).double() if not isinstance(a, torch.Tensor) else a.double().cpu()
    b = torch.as_tensor(np.asarray(b)).double() if not isinstance(b, torch.Tensor) else b.double().cpu()
    return float((a - b).norm() / b.norm())


# ---------------------------------------------------------------- HIP op hooks for the hybrid graphs
def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


class HipHooks:
    """callables with the UNetOracle hook signatures, backed by the op-level C ABI (consolver_hip_ops.h)."""

    def __init__(self):
        self._packed = {}

    def _w(self, w, kind):
        # keyed on the tensor OBJECT, which the cache keeps alive: a (data_ptr) key would match another layer's weight that the
        # allocator later placed at a freed address
        key = (id(w), kind)
        if key not in self._packed:
            if kind == "conv":
                self._packed[key] = (w, ops.pack_conv_weight(w).to(DEV))
            else:
                self._packed[key] = (w, w.to(DEV, torch.float16).contiguous())
        return self._packed[key][1]

    def conv3(self, x, w, b, stride):
        return _nchw(ops.conv2d(_nhwc(x), self._w(w, "conv"), b, taps=9, stride=stride))

    def conv1(self, x, w, b, stride):
        assert stride == 1
        return _nchw(ops.conv2d(_nhwc(x), self._w(w, "conv"), b, taps=1))

    def linear(self, x, w, b):
        B, N, K = x.shape
        return ops.linear(x.reshape(B * N, K).contiguous(), self._w(w, "lin"), b).reshape(B, N, -1)

    def geglu(self, x, w, b):
        key = (id(w), "geglu")
        if key not in self._packed:
            wp, bp = ops.geglu_pack(w, b)
            self._packed[key] = (w, wp.to(DEV), bp.to(DEV))
        _, wp, bp = self._packed[key]
        B, N, K = x.shape
        return ops.linear(x.reshape(B * N, K).contiguous(), wp, bp, geglu=True).reshape(B, N, -1)

    def sdpa(self, q, k, v, H):
        return ops.attention(q.contiguous(), k.contiguous(), v.contiguous(), H)

    def group_norm(self, x, w, b, groups, eps, silu):
        return _nchw(ops.group_norm(_nhwc(x), w, b, groups, eps, silu))

    def layer_norm(self, x, w, b):
        B, N, C = x.shape
        return ops.layer_norm(x.reshape(B * N, C).contiguous(), w, b).reshape(B, N, C)


CLASSES = ["conv3", "conv1", "linear", "geglu", "sdpa", "group_norm", "layer_norm"]


def cast32(hook):
    """the HIP fp16 kernel as one op of an otherwise fp32 graph: fp16 in (inputs rounded), fp16 out, cast back up"""
    def f(*args):
        a = [x.half() if isinstance(x, torch.Tensor) and x.is_floating_point() else x for x in args]
        return hook(*a).float()
    return f


def build_full(seed=7, residual="f16"):
    """the full SD1.5 UNet on seeded synthetic weights: ONE handle per seed and pytest process (tests/_models.py), switched between the residual-stream modes"""
    return get_unet({}, seed=seed, residual=residual)


@pytest.mark.timeout(1800)
def test_forward_error_budget_and_class_attribution():
    """(b) + (c): one CFG dual-batch forward of the full SD1.5 UNet at three timesteps of the 8-step grid."""
    u, sd = build_full(residual="f16")
    orc = get_oracle({}, seed=7)                                              # the oracle: CPU fp32
    t16 = UNetOracle(sd, u.config, device=DEV, dtype=torch.float16)          # comparator: plain torch fp16 on the GPU
    hooks = HipHooks()
    g = torch.Generator().manual_seed(5)
    rows = []
    inputs = []
    worst_ratio = 0.0
    for t in (999, 499, 124):
        lat = torch.randn(1, 4, 64, 64, generator=g).half()
        ctx = synthetic_prompt_embeds(2, seed=13 + t).half()
        want = orc(torch.cat([lat.float()] * 2), t, ctx.float())
        inputs.append((lat, ctx, want))
        e_t16 = rel_l2(t16(torch.cat([lat] * 2), t, ctx).float(), want)
        got = u(lat.to(DEV), t, encoder_hidden_states=ctx.to(DEV), dup=2, reuse_kv=False)[0]
        e_hip = rel_l2(got.float(), want)
        row = dict(t=t, torch_fp16=e_t16, hip_executor=e_hip)
        if t == 499:
            for c in CLASSES + ["all"]:
                hyb = UNetOracle(sd, u.config, device=DEV, dtype=torch.float16)
                for k in (CLASSES if c == "all" else [c]):
                    hyb.ops[k] = getattr(hooks, k)
                row["hybrid_" + c] = rel_l2(hyb(torch.cat([lat] * 2), t, ctx).float(), want)
                del hyb
            # the reverse attribution: an fp32 torch graph on the GPU with ONE class computed by the fp16 HIP kernel
            g32 = UNetOracle(sd, u.config, device=DEV, dtype=torch.float32)
            row["gpu_fp32_graph"] = rel_l2(g32(torch.cat([lat.float()] * 2), t, ctx.float()), want)
            for c in CLASSES + ["all"]:
                for k in CLASSES:
                    g32.ops[k] = cast32(getattr(hooks, k)) if (c == "all" or k == c) else None
                row["f32+hip_" + c] = rel_l2(g32(torch.cat([lat.float()] * 2), t, ctx.float()), want)
            del g32
        rows.append(row)
        worst_ratio = max(worst_ratio, e_hip / e_t16)
    # the same three forwards on the split-fp16 residual stream (the same handle, switched)
    ux2, _ = build_full(residual="f16x2")
    for row, (lat, ctx, want) in zip(rows, inputs):
        row["hip_executor_f16x2"] = rel_l2(ux2(lat.to(DEV), row["t"], encoder_hidden_states=ctx.to(DEV), dup=2, reuse_kv=False)[0].float(), want)
    print("\nper-forward eps error vs the fp32 oracle (relative L2), full SD1.5 UNet, CFG dual batch:")
    for r in rows:
        print("  " + "\n    ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in r.items()))
    print(f"  worst HIP / torch-fp16 ratio = {worst_ratio:.3f}")
    print(f"  gate (north_star, on the 8-step latents): {GATE:.1e}; per-forward eps error is printed for attribution, the latents are what is gated")
    for r in rows:
        assert r["hip_executor"] <= 1.25 * r["torch_fp16"], r
        assert r["hip_executor"] < 1.75e-3, r                     # f16 stream: regression bound = measured 1.47e-3 .. 1.59e-3 + 10 % (gate 1.0e-3 not met in this mode)
        assert r["hip_executor_f16x2"] < FWD_X2_BOUND, r          # f16x2 stream: measured + 10 % (see FWD_X2_BOUND)
        assert r["hip_executor_f16x2"] < 0.72 * r["hip_executor"], r


def _scheduler(seed=11):
    sch = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                     timestep_spacing="trailing", order_dim=4, scaler_dim=0,
                                     factor_net_kwargs=dict(hidden_dim=64, num_actions=11))
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in sch.factor_net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    w = {k: v.numpy().copy() for k, v in sch.factor_net.state_dict().items()}
    sch.factor_net.to(DEV)
    return sch, w


def _oracle_sched(w):
    s = so.PPOSchedulerOracle(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                              timestep_spacing="trailing", order_dim=4, scaler_dim=0, num_actions=11, weights=w)
    return s


def _hip_trajectory(unet, sch, idx, noise, ctx_d, B, n, g, hi_steps=None):
    """the product's loop, step by step as SDSamplingEngine runs it (fp32 solver state and fp32 eps, the denoiser reads the state's fp16 copy; per-step latents kept for the drift table)"""
    sch.factor_net.forced_action_idx = [torch.from_numpy(i).to(DEV) for i in idx]
    sch.set_timesteps(n, device=DEV)
    x = noise.to(DEV).float()
    traj = []
    for i, t in enumerate(sch.timesteps):
        # (round 6: the engine takes the denoiser's output in fp32; hi_steps: its precision schedule -- the first hi_steps forwards on the split stream, the rest on one plane)
        kw = {} if hi_steps is None else {"residual": "f16x2" if i < hi_steps else "f16"}
        eps = unet(x.half(), t, encoder_hidden_states=ctx_d, dup=2, reuse_kv=(i > 0), out_dtype=torch.float32, **kw)[0]
        x = sch.step(eps[B:], t, x, return_dict=False, eps_uncond=eps[:B], guidance_scale=g)[0]
        assert x.dtype == torch.float32
        traj.append(x.float().cpu().numpy())
    if hi_steps is not None:
        unet.set_residual_precision_keep("f16x2")
    return traj


# B = 16 is configs[1]'s own batch: ~10 minutes of fp32 CPU oracle on the GPU box, so it runs on request (CS_PARITY_B16=1; its output is kept
# under profiles/) and the default suite keeps B = 2 -- the per-sample arithmetic does not depend on the batch.
@pytest.mark.timeout(6000)
@pytest.mark.parametrize("B", [2] + ([16] if os.environ.get("CS_PARITY_B16") == "1" else []))
def test_eight_step_trajectory_full_unet_vs_oracle(B):
    """(a): configs[1]'s 8 steps (trailing grid 999..124, CFG 3, order 4) on the full UNet."""
    sch, w = _scheduler()
    n, g = 8, 3.0
    idx = np.random.default_rng(6).integers(0, 11, size=(n, B, 3))
    pe, ne = synthetic_prompt_embeds(B, seed=1001).half(), synthetic_prompt_embeds(B, seed=1002).half()
    noise = torch.randn(B, 4, 64, 64, generator=torch.Generator().manual_seed(43)).half()
    ctx = torch.cat([ne, pe])
    ctx_d = ctx.to(DEV)

    # --- the oracle (fp32 CPU) and the torch-fp16 comparator pipeline (fp16 UNet graph + fp16-rounded solver arithmetic); the oracle's per-step latents are
    #     kept for the teacher-forced per-forward errors below
    u, sd = build_full(residual="f16")
    orc_u = get_oracle({}, seed=7)
    t16_u = UNetOracle(sd, u.config, device=DEV, dtype=torch.float16)
    s_or, s_16 = _oracle_sched(w), _oracle_sched(w)
    s_or.set_timesteps(n); s_16.set_timesteps(n)
    xo = noise.float().numpy()
    x16 = noise.float().numpy()
    orc_in, orc_eps, orc_traj, t16_traj, fwd_t16 = [], [], [], [], []
    for i, t in enumerate(s_or.timesteps):
        t = int(t)
        orc_in.append(xo)
        e_or = orc_u(torch.cat([torch.from_numpy(xo)] * 2), t, ctx.float()).numpy()     # the oracle itself stays fp32 end to end
        orc_eps.append(e_or)
        fwd_t16.append(rel_l2(t16_u(torch.cat([torch.from_numpy(xo).half()] * 2), t, ctx).float().cpu().numpy(), e_or))
        xo = s_or.step(so.cfg_combine(e_or[:B], e_or[B:], g), t, xo, idx[i], cond_dtype="f16")["prev_sample"]
        f16 = t16_u(torch.cat([torch.from_numpy(x16).half()] * 2), t, ctx).float().cpu().numpy()
        ec = so.round_f16(so.cfg_combine(so.round_f16(f16[:B]), so.round_f16(f16[B:]), g))
        x16 = so.round_f16(s_16.step(ec, t, x16, idx[i], cond_dtype="f16")["prev_sample"])
        orc_traj.append(xo); t16_traj.append(x16)
    del t16_u

    # --- the product in both residual-stream modes: free-running trajectory + teacher-forced forwards on the ORACLE's latents of each step
    res = {}
    for mode in ("f16", "f16x2"):
        un, _ = build_full(residual=mode)
        traj = _hip_trajectory(un, sch, idx, noise, ctx_d, B, n, g)
        if mode == "f16":      # the engine runs the same loop (ring buffers, no allocation): bit-identical final latents
            sch.factor_net.forced_action_idx = [torch.from_numpy(i).to(DEV) for i in idx]
            eng = SDSamplingEngine(un, sch, guidance_scale=g)
            got = eng.generate(pe.to(DEV), ne.to(DEV), latents=noise.to(DEV), num_inference_steps=n).float().cpu().numpy()
            assert np.array_equal(got, traj[-1])
        fwd = []
        for i, t in enumerate(s_or.timesteps):
            e = un(torch.from_numpy(orc_in[i]).half().to(DEV), int(t), encoder_hidden_states=ctx_d, dup=2, reuse_kv=(i > 0))[0].float().cpu().numpy()
            fwd.append(rel_l2(e, orc_eps[i]))
        res[mode] = (traj, fwd)
    rows = []
    for i, t in enumerate(s_or.timesteps):
        rows.append(dict(step=i, t=int(t), fwd_hip=res["f16"][1][i], fwd_x2=res["f16x2"][1][i], fwd_t16=fwd_t16[i],
                         drift_hip=rel_l2(res["f16"][0][i], orc_traj[i]), drift_x2=rel_l2(res["f16x2"][0][i], orc_traj[i]), drift_t16=rel_l2(t16_traj[i], orc_traj[i])))
    print(f"\n8-step trajectory, full SD1.5 UNet, B={B}, CFG 3 (relative L2 vs the fp32 oracle; gate on the final latents {GATE:.1e}):")
    for r in rows:
        print("  step {step} t={t:3d}  forward: f16x2 {fwd_x2:.3e} f16 {fwd_hip:.3e} torch-fp16 {fwd_t16:.3e}   "
              "latents: f16x2 {drift_x2:.3e} f16 {drift_hip:.3e} torch-fp16 {drift_t16:.3e}".format(**r))
    final_hip, final_x2, final_t16 = rows[-1]["drift_hip"], rows[-1]["drift_x2"], rows[-1]["drift_t16"]
    print(f"  final latents: f16x2 {final_x2:.3e} (gate {GATE:.1e}: {'MET' if final_x2 <= GATE else 'NOT MET'}, margin {100 * (1 - final_x2 / GATE):.1f} %), "
          f"f16 {final_hip:.3e} (gate not met in this mode; regression bound 1.39e-3), torch-fp16 class {final_t16:.3e}")
    assert np.isfinite(res["f16"][0][-1]).all() and np.isfinite(res["f16x2"][0][-1]).all()
    # ---- the gate: split-fp16 residual stream
    assert final_x2 <= GATE, final_x2
    for r in rows:
        assert r["drift_x2"] <= GATE, r                          # at every step of the trajectory, not only the last
    # ---- f16 stream: the reference's own arithmetic class; regression bounds = measured + 10 %
    for r in rows:
        assert r["fwd_hip"] <= 1.25 * r["fwd_t16"], r
    assert final_hip <= 1.25 * final_t16 + 2e-4, (final_hip, final_t16)
    assert final_hip < 1.39e-3, final_hip                         # measured 1.263e-3 at B = 2 with the fp32 solver state (fp16 state, round 4: 1.40e-3; torch-fp16 class: 1.97e-3), + 10 %


# the step counts the reference publishes besides 8 (readme.md:158-163: 5 / 8 / 10 / 12; train_ppo.py:345 draws 2..15) and a second weight seed: the gated mode only,
# B = 1 (the per-sample arithmetic does not depend on the batch), every step of every trajectory against the gate
_ORACLE_CASES = {}


def _oracle_case(n, wseed, g=3.0, B=1):
    """one seeded (prompts, noise, replayed action indices) case per (n, weight seed) and its fp32 CPU oracle trajectory (UNetOracle + PPOSchedulerOracle),
    computed once per pytest process: the engine loop, the rollout function and the diffusers pipeline loop are all held against the same numbers"""
    key = (n, wseed, g, B)
    if key not in _ORACLE_CASES:
        sch, w = _scheduler()
        idx = np.random.default_rng(60 + n).integers(0, 11, size=(n, B, 3))
        pe, ne = synthetic_prompt_embeds(B, seed=2001 + n).half(), synthetic_prompt_embeds(B, seed=2002 + n).half()
        noise = torch.randn(B, 4, 64, 64, generator=torch.Generator().manual_seed(143 + wseed)).half()
        ctx = torch.cat([ne, pe])
        build_full(seed=wseed, residual="f16x2")                  # (the oracle shares the handle's host weights: tests/_models.py)
        orc_u = get_oracle({}, seed=wseed)
        s_or = _oracle_sched(w)
        s_or.set_timesteps(n)
        xo = noise.float().numpy()
        traj = []
        for i, t in enumerate(s_or.timesteps):
            e_or = orc_u(torch.cat([torch.from_numpy(xo)] * 2), int(t), ctx.float()).numpy()
            xo = s_or.step(so.cfg_combine(e_or[:B], e_or[B:], g), int(t), xo, idx[i], cond_dtype="f16")["prev_sample"]
            traj.append(xo)
        _ORACLE_CASES[key] = dict(sch=sch, w=w, idx=idx, pe=pe, ne=ne, noise=noise, ctx=ctx, traj=traj)
    return _ORACLE_CASES[key]


@pytest.mark.timeout(3000)
@pytest.mark.parametrize("n,wseed", [(4, 7), (12, 7), (8, 8)])
def test_gate_holds_at_other_step_counts_and_weights(n, wseed):
    B, g = 1, 3.0
    c = _oracle_case(n, wseed, g, B)
    sch, idx, noise, ctx = c["sch"], c["idx"], c["noise"], c["ctx"]
    ux2, sd = build_full(seed=wseed, residual="f16x2")
    traj = _hip_trajectory(ux2, sch, idx, noise, ctx.to(DEV), B, n, g)
    drift = [rel_l2(traj[i], c["traj"][i]) for i in range(n)]
    print(f"\n{n}-step trajectory, weight seed {wseed}, f16x2: latents vs the fp32 oracle per step " + " ".join(f"{d:.3e}" for d in drift) +
          f"  -> final {drift[-1]:.3e}, gate {GATE:.1e}, margin {100 * (1 - max(drift) / GATE):.1f} %")
    if wseed != 7:
        drop({}, seed=wseed)                                      # (3.4 GB of host weights + the oracle's copy)
    assert np.isfinite(traj[-1]).all()
    assert max(drift) <= GATE, drift


# The ENGINE'S DEFAULT: a precision schedule of the residual stream (engine.py, `hi_precision_steps="auto"`): the first ceil(n / 4) forwards of a generation on the
# split (hi + lo) stream, the rest on one fp16 plane -- the gate's budget is spent in the first steps (profiles/r06_parity_schedule.txt).  Every step of every trajectory
# the defaults produce is held to the gate here, and the engine's own output is the loop's, bit for bit.
@pytest.mark.timeout(3000)
@pytest.mark.parametrize("n,wseed", [(4, 7), (8, 7), (12, 7), (15, 7), (8, 8)])
def test_gate_holds_under_the_engine_precision_schedule(n, wseed):
    B, g = 1, 3.0
    c = _oracle_case(n, wseed, g, B)
    sch, idx, noise, ctx = c["sch"], c["idx"], c["noise"], c["ctx"]
    ux2, _ = build_full(seed=wseed, residual="f16x2")
    eng = SDSamplingEngine(ux2, sch, guidance_scale=g)
    k = eng.hi_steps(n)
    assert k == {4: 1, 8: 2, 12: 3, 15: 4}[n] and eng.hi_precision_steps == "auto"
    traj = _hip_trajectory(ux2, sch, idx, noise, ctx.to(DEV), B, n, g, hi_steps=k)
    drift = [rel_l2(traj[i], c["traj"][i]) for i in range(n)]
    sch.factor_net.forced_action_idx = [torch.from_numpy(i).to(DEV) for i in idx]
    got = eng.generate(c["pe"].to(DEV), c["ne"].to(DEV), latents=noise.to(DEV), num_inference_steps=n).float().cpu().numpy()
    assert ux2.residual == "f16x2"                                # the engine hands the handle back in its own mode
    assert np.array_equal(got, traj[-1])
    print(f"\n{n}-step trajectory, weight seed {wseed}, under the engine's default schedule ({k} of {n} forwards on the split stream): latents vs the fp32 oracle per step "
          + " ".join(f"{d:.3e}" for d in drift) + f"  -> final {drift[-1]:.3e}, gate {GATE:.1e}, margin {100 * (1 - max(drift) / GATE):.1f} %")
    assert max(drift) <= GATE, drift


# north_star names two call paths as the drop-in (besides the native engine, which the tests above hold to the gate): the rollout function the trainer calls
# (denoise_ppo.py:52-113; train_ppo.py:345 draws n from 2..15) and the diffusers pipeline loop (gen_pretrain/pipeline.py:1045-1098: torch.cat dual batch, CFG
# combined by the pipeline, scheduler.step(noise_pred, t, latents)).  Both carry the latents between the steps: `rollout.denoise_diffusion` keeps them fp32
# internally (solver_state_dtype, default); a diffusers user sets `scheduler.prev_sample_dtype = torch.float32` -- what the reference's own
# latents are from step 2 on with an fp32 policy net.  The all-fp16 state is the explicitly non-default mode: printed, regression-bounded.
@pytest.mark.timeout(3000)
@pytest.mark.parametrize("n", [2, 3, 4, 8, 12, 15])      # train_ppo.py:345 draws the step count from 2 .. 15
def test_gate_on_the_rollout_function_and_the_pipeline_loop(n):
    from consolver_amd.rollout import denoise_diffusion
    from tests import fake_diffusers as fd
    B, g, wseed = 1, 3.0, 7
    c = _oracle_case(n, wseed, g, B)
    sch, idx, noise, pe, ne = c["sch"], c["idx"], c["noise"], c["pe"], c["ne"]
    want = c["traj"][-1]
    ux2, _ = build_full(seed=wseed, residual="f16x2")
    queue = lambda: [torch.from_numpy(i).to(DEV) for i in idx]

    def rollout(**kw):
        sch.factor_net.forced_action_idx = queue()
        out = denoise_diffusion(None, sch, ux2, noise.to(DEV), ["a"] * B, None, cfg=g, num_inference_steps=n, prompt_embeds=pe.to(DEV),
                                negative_prompt_embeds=ne.to(DEV), **kw)
        assert not sch.factor_net.forced_action_idx                  # every recorded index was consumed: n steps ran
        return out
    lat, conds, probs, actions, masks, _ = rollout()
    assert lat.dtype == noise.dtype and conds["epsilon"].dtype == torch.float16 and actions.shape == (B, n - 1, 3)      # the boundary keeps the caller's dtypes
    e_roll = rel_l2(lat.float().cpu().numpy(), want)
    e_roll16 = rel_l2(rollout(solver_state_dtype=None)[0].float().cpu().numpy(), want)

    # the diffusers loop (tests/fake_diffusers.py restates StableDiffusionPipeline.__call__'s denoising loop) through the plain scheduler protocol
    pipe = fd.StableDiffusionPipeline(vae=None, text_encoder=None, tokenizer=None, unet=ux2, scheduler=sch, safety_checker=None)

    def pipeline(prev_dtype):
        sch.factor_net.forced_action_idx = queue()
        sch.prev_sample_dtype = prev_dtype
        try:
            return pipe(prompt_embeds=pe.to(DEV), negative_prompt_embeds=ne.to(DEV), latents=noise.to(DEV).clone(), num_inference_steps=n, guidance_scale=g).images
        finally:
            sch.prev_sample_dtype = None
    lp = pipeline(torch.float32)
    assert lp.dtype == torch.float32
    e_pipe = rel_l2(lp.cpu().numpy(), want)
    e_pipe16 = rel_l2(pipeline(None).float().cpu().numpy(), want)
    print(f"\n{n}-step latents vs the fp32 oracle, full SD1.5 UNet, f16x2 stream: rollout.denoise_diffusion {e_roll:.3e} (fp16 state, non-default: {e_roll16:.3e}); "
          f"diffusers pipeline loop with prev_sample_dtype=float32 {e_pipe:.3e} (fp16 latents, the scheduler's default under a foreign pipeline: {e_pipe16:.3e}); gate {GATE:.1e}")
    assert e_roll <= GATE, e_roll
    assert e_pipe <= GATE, e_pipe
    # the all-fp16 state is the reference pipeline's own arithmetic class for the latents (one fp16 rounding per step): regression bound only, the gate is printed above
    assert e_roll16 < 1.35e-3 and e_pipe16 < 1.35e-3, (e_roll16, e_pipe16)


def test_two_rollouts_with_different_prompts_do_not_share_kv():
    """regression: the cross-attention K/V cache must not survive into a rollout with other prompts (a fresh torch.cat
    of the same size lands on the freed address of the previous one)."""
    from consolver_amd.rollout import denoise_diffusion
    u, sd = get_unet(dict(layers_per_block=1, sample_size=16), seed=3)
    sch, _ = _scheduler()
    B, n = 2, 3
    noise = torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(1)).half().to(DEV)
    idx = [torch.full((B, 3), 5, dtype=torch.long, device=DEV) for _ in range(n)]

    def run(seed, fresh):
        un = u
        if fresh:
            un = HipUNet2DConditionModel(dict(layers_per_block=1, sample_size=16), device=DEV)
            un.load_state_dict(sd)
        pe = synthetic_prompt_embeds(B, seed=seed).half().to(DEV)
        ne = synthetic_prompt_embeds(B, seed=seed + 1).half().to(DEV)
        sch.factor_net.forced_action_idx = list(idx)
        return denoise_diffusion(None, sch, un, noise, ["a"] * B, None, cfg=3.0, num_inference_steps=n, prompt_embeds=pe,
                                 negative_prompt_embeds=ne)[0].clone()
    a1 = run(100, False)
    b1 = run(200, False)          # second rollout on the SAME UNet object, other prompts
    b2 = run(200, True)           # the same rollout on a fresh UNet (no cache history)
    assert torch.equal(b1, b2)
    assert not torch.equal(a1, b1)
    # default reuse_kv=None: a new tensor (even at a recycled address) is never taken for the cached one
    ctx1 = torch.cat([synthetic_prompt_embeds(B, seed=1).half(), synthetic_prompt_embeds(B, seed=2).half()]).to(DEV)
    y1 = u(noise, 499, encoder_hidden_states=ctx1, dup=2)[0].clone()
    del ctx1
    ctx2 = torch.cat([synthetic_prompt_embeds(B, seed=3).half(), synthetic_prompt_embeds(B, seed=4).half()]).to(DEV)
    y2 = u(noise, 499, encoder_hidden_states=ctx2, dup=2)[0].clone()
    y2_ref = u(noise, 499, encoder_hidden_states=ctx2, dup=2, reuse_kv=False)[0]
    assert torch.equal(y2, y2_ref) and not torch.equal(y1, y2)
    # the same unmodified tensor object IS reused (and gives the same result)
    assert torch.equal(u(noise, 499, encoder_hidden_states=ctx2, dup=2)[0], y2_ref)


def test_step_accepts_cloned_timesteps_without_per_step_sync():
    """a caller that clones ``t`` (or builds its own CUDA scalars) gets the same trajectory; a wrong value is detected."""
    sch, _ = _scheduler()
    B, n = 2, 6
    g = torch.Generator().manual_seed(0)
    eps = [torch.randn(B, 4, 8, 8, generator=g).to(DEV) for _ in range(n)]
    x0 = torch.randn(B, 4, 8, 8, generator=g).to(DEV)
    idx = [torch.randint(0, 11, (B, 3), generator=g).to(DEV) for _ in range(n)]

    def run(mode):
        sch.set_timesteps(n, device=DEV)
        sch.factor_net.forced_action_idx = list(idx)
        x = x0
        for i, t in enumerate(sch.timesteps):
            tt = {"elem": t, "clone": t.clone(), "host": int(sch._timesteps[i]), "cpu": t.cpu()}[mode]
            x = sch.step(eps[i], tt, x, return_dict=False)[0]
        return x
    ref = run("elem")
    for mode in ("clone", "host", "cpu"):
        assert torch.equal(run(mode), ref), mode
    sch.verify_timesteps()
    # a foreign CUDA timestep that is NOT the next grid entry is reported
    sch.set_timesteps(n, device=DEV)
    sch.factor_net.forced_action_idx = list(idx)
    x = sch.step(eps[0], sch.timesteps[0].clone(), x0, return_dict=False)[0]
    sch.step(eps[1], sch.timesteps[3].clone(), x, return_dict=False)
    with pytest.raises(RuntimeError):
        sch.verify_timesteps()


def test_policy_broadcast_row_equals_per_row_evaluation():
    """cs_factor_probs with x_row_stride = 0 (one conditioning row for all B samples: every sampling step) evaluates the MLP
    once and writes B rows; it must equal the per-row launch bit for bit."""
    for H, K, A in ((256, 11, 3), (64, 11, 3), (30, 7, 2)):
        net = consolver_amd.FactorNetPPO(hidden_dim=H, num_actions=K, order_dim=A + 1, scaler_dim=0)
        g = torch.Generator().manual_seed(H)
        with torch.no_grad():
            for p in net.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
        net.to(DEV)
        row = torch.tensor([[874.0, 749.0]], device=DEV)
        B = 16
        bc = net.probs_from(row, batch=B)
        full = net.probs_from(row.repeat(B, 1))
        assert bc.shape == (B, A, K) and torch.equal(bc, full)
        want = so.factor_net_probs({k: v.cpu().numpy() for k, v in net.state_dict().items()}, row.cpu().numpy())
        np.testing.assert_allclose(bc[:1].cpu().numpy(), np.asarray(want).reshape(1, A, K), rtol=3e-4, atol=2e-6)


def test_fmppo_fp32_sample_is_consumed_as_fp32():
    """scheduler_fmppo.py:354,429-436: the sample is upcast to fp32, the update runs in fp32 and only the result is rounded to
    the model dtype -- an fp32 sample must not be rounded to bf16 on the way in."""
    s = consolver_amd.FMPPOScheduler(shift=3.0, use_dynamic_shifting=True, order_dim=2, scaler_dim=0, mu_dim=0,
                                     factor_net_kwargs=dict(hidden_dim=32, num_actions=11))
    s.factor_net.to(DEV)
    n, B = 4, 2
    s.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=1.15, device=DEV)
    s.set_begin_index(0)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 64, 64, generator=g)                         # fp32 sample with bits below bf16 precision
    v = torch.randn(B, 64, 64, generator=g).bfloat16()
    s.factor_net.forced_action_idx = torch.zeros(B, 1, dtype=torch.long, device=DEV)
    out = s.step(v.to(DEV), s.timesteps[0], x.to(DEV), return_dict=False)[0]
    assert out.dtype == torch.bfloat16
    dt = np.float32(s._sigmas[1] - s._sigmas[0])
    step = torch.tensor(dt) * v                                     # m = 1: v_eff = v (bf16) and 0-d fp32 dt -> a bf16 product (:429)
    assert step.dtype == torch.bfloat16
    want = (x + step).bfloat16()
    rounded_in = (x.bfloat16().float() + step.float()).bfloat16()
    assert torch.equal(out.cpu(), want)
    assert not torch.equal(want, rounded_in)                        # the case distinguishes the two behaviours


def test_rollout_shares_denoiser_calls_for_identical_inputs():
    """the trainer's batch is B copies of one sample (repeat_random_sample, data_processing.py:65-83): with identical_inputs=True the
    rollout evaluates steps 0 and 1 (scaler_dim = 0) for one row and broadcasts; records and latents equal the full-batch rollout to
    fp16 rounding (the one-row denoiser call may pick other tile / split-K shapes than the batch-B call)."""
    from consolver_amd.rollout import denoise_diffusion
    u, _ = get_unet(dict(layers_per_block=1, sample_size=16), seed=3)
    sch, _ = _scheduler()
    B, n = 4, 5
    noise = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(1)).half().to(DEV).repeat(B, 1, 1, 1)
    pe = synthetic_prompt_embeds(1, seed=100).half().to(DEV).repeat(B, 1, 1)
    ne = synthetic_prompt_embeds(1, seed=101).half().to(DEV).repeat(B, 1, 1)
    idx = [torch.randint(0, 11, (B, 3), generator=torch.Generator().manual_seed(10 + i)).to(DEV) for i in range(n)]

    def run(flag):
        sch.factor_net.forced_action_idx = list(idx)
        return denoise_diffusion(None, sch, u, noise, ["a"] * B, None, cfg=3.0, num_inference_steps=n, prompt_embeds=pe,
                                 negative_prompt_embeds=ne, identical_inputs=flag)
    full, shared = run(False), run(True)
    lat_f, conds_f, probs_f, act_f, masks_f, _ = full
    lat_s, conds_s, probs_s, act_s, masks_s, _ = shared
    assert torch.equal(act_f, act_s) and torch.equal(masks_f, masks_s) and torch.equal(conds_f["x"], conds_s["x"])
    assert torch.allclose(probs_f, probs_s, rtol=1e-5, atol=1e-7)
    assert conds_s["epsilon"].shape == conds_f["epsilon"].shape == (B, n - 1, 4, 4, 16, 16)
    assert rel_l2(conds_s["epsilon"].float(), conds_f["epsilon"].float()) < 2e-3
    assert rel_l2(lat_s.float(), lat_f.float()) < 2e-3
    assert not torch.equal(lat_s[0], lat_s[1])                   # the trajectories do diverge (different sampled coefficients)
