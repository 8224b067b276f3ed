"""The schedulers as diffusers components: the call sequence of gen_ppo.py:110-195 and generate_ours.py:127-134.

diffusers is not installed here, so ``tests/fake_diffusers.py`` (written from the public diffusers API, independent of the
product) is injected as ``diffusers`` BEFORE ``consolver_amd`` is imported -- class bases are chosen at import time, hence a
fresh interpreter per case.  The stand-alone mixins (no diffusers importable) are tested in-process.
"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = """
import os, sys, json, tempfile
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import fake_diffusers
fake_diffusers.install()
import torch, numpy as np
import diffusers
from diffusers import StableDiffusionPipeline
from diffusers.schedulers.scheduling_utils import SchedulerMixin
from diffusers.configuration_utils import ConfigMixin, FrozenDict
import consolver_amd
from consolver_amd import PPOScheduler, FMPPOScheduler
assert consolver_amd.HAVE_DIFFUSERS
"""


def run_script(body, timeout=600):
    code = PRELUDE.format(root=ROOT) + textwrap.dedent(body)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    return r.stdout


def test_gen_ppo_load_pipeline_sequence_under_diffusers():
    """gen_ppo.py:116-195: PPOScheduler(...) -> StableDiffusionPipeline.from_pretrained(path, scheduler=noise_scheduler,
    revision=, torch_dtype=, safety_checker=None) -> set_progress_bar_config -> scheduler.factor_net.load_state_dict ->
    pipeline.to(device, dtype=weight_dtype) -> scheduler.factor_net.to(device, dtype=weight_dtype)."""
    out = run_script("""
    assert issubclass(PPOScheduler, SchedulerMixin) and issubclass(PPOScheduler, ConfigMixin)
    assert issubclass(FMPPOScheduler, SchedulerMixin) and issubclass(FMPPOScheduler, ConfigMixin)
    factor_net_kwargs = dict(embedding_dim=32, hidden_dim=64, num_actions=21)
    noise_scheduler = PPOScheduler(beta_end=0.012, beta_schedule="scaled_linear", beta_start=0.00085, num_train_timesteps=1000,
                                   steps_offset=1, trained_betas=None, timestep_spacing="trailing", order_dim=4, scaler_dim=0,
                                   use_conv=False, factor_net_kwargs=factor_net_kwargs)
    assert isinstance(noise_scheduler.config, FrozenDict) and noise_scheduler.config.steps_offset == 1
    unet = torch.nn.Linear(2, 2)
    pipeline = StableDiffusionPipeline.from_pretrained("runwayml/stable-diffusion-v1-5", scheduler=noise_scheduler, revision=None,
                                                       torch_dtype=torch.float16, safety_checker=None,
                                                       components=dict(vae=None, text_encoder=None, tokenizer=None, unet=unet))
    assert pipeline.scheduler is noise_scheduler and pipeline.deprecations == []
    pipeline.set_progress_bar_config(disable=True)
    ref = consolver_amd.FactorNetPPO(order_dim=4, scaler_dim=0, use_conv=False, **factor_net_kwargs)
    with torch.no_grad():
        for p_ in ref.parameters():
            p_.copy_(torch.randn(p_.shape) * 0.1)
    weight = {k: v.clone() for k, v in ref.state_dict().items()}
    pipeline.scheduler.factor_net.load_state_dict(weight)
    pipeline = pipeline.to("cpu", dtype=torch.float16)
    assert unet.weight.dtype == torch.float16
    pipeline.scheduler.factor_net.to("cpu", dtype=torch.float16)
    assert all(v.dtype == torch.float16 for v in pipeline.scheduler.factor_net.state_dict().values())
    pipeline.enable_vae_slicing()
    # a component that is not a SchedulerMixin is refused by the same check
    class NotAScheduler:
        config = FrozenDict(steps_offset=1)
    try:
        StableDiffusionPipeline.from_pretrained("x", scheduler=NotAScheduler(), components=dict(vae=None, text_encoder=None, tokenizer=None, unet=unet))
        raise SystemExit("the fake's component check accepted a non-scheduler")
    except ValueError:
        pass
    # steps_offset != 1: the pipeline rewrites scheduler._internal_dict and the scheduler must follow (leading spacing uses it)
    s0 = PPOScheduler(beta_end=0.012, beta_schedule="scaled_linear", beta_start=0.00085, steps_offset=0, timestep_spacing="leading",
                      order_dim=2, scaler_dim=0, factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
    s0.set_timesteps(8)
    before = s0.timesteps.tolist()
    p0 = StableDiffusionPipeline.from_pretrained("x", scheduler=s0, components=dict(vae=None, text_encoder=None, tokenizer=None, unet=unet))
    assert p0.deprecations == ["steps_offset!=1"] and s0.config.steps_offset == 1
    s0.set_timesteps(8)
    assert s0.timesteps.tolist() == [t + 1 for t in before], (before, s0.timesteps.tolist())
    print("OK", json.dumps(sorted(k for k in noise_scheduler.config.keys() if not k.startswith("_"))))
    """)
    keys = json.loads(out.strip().splitlines()[-1][3:])
    assert keys == sorted(["num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "trained_betas", "prediction_type",
                           "timestep_spacing", "steps_offset", "order_dim", "scaler_dim", "use_conv", "ppo_type", "factor_net_kwargs"])


def test_config_round_trips_under_diffusers(tmp_path):
    """save_config / from_config / save_pretrained / from_pretrained / compatibles through diffusers' mixins; the FLUX scheduler's
    ``from_pretrained(repo, subfolder="scheduler", order_dim=..., ...)`` of edit_ppo/generate_ours.py:127-134."""
    run_script(f"""
    d = {str(tmp_path)!r}
    s = PPOScheduler(beta_end=0.012, beta_schedule="scaled_linear", beta_start=0.00085, steps_offset=1, timestep_spacing="trailing",
                     order_dim=3, scaler_dim=1, trained_betas=None, factor_net_kwargs=dict(embedding_dim=16, hidden_dim=32, num_actions=11))
    s.save_pretrained(os.path.join(d, "sd"))
    j = json.load(open(os.path.join(d, "sd", "scheduler_config.json")))
    assert j["_class_name"] == "PPOScheduler" and j["order_dim"] == 3 and j["factor_net_kwargs"]["num_actions"] == 11
    t = PPOScheduler.from_pretrained(os.path.join(d, "sd"))
    pub = lambda c: {{k: v for k, v in dict(c).items() if not k.startswith("_")}}
    assert pub(t.config) == pub(s.config)
    assert t.factor_net.action_dims == s.factor_net.action_dims
    u = PPOScheduler.from_config(s.config, timestep_spacing="leading")
    assert u.config.timestep_spacing == "leading" and u.config.order_dim == 3
    assert s.compatibles == [diffusers.PNDMScheduler]                    # (the one Karras-family class the fake library defines)
    assert "DDIMScheduler" in PPOScheduler._compatibles and "UniPCMultistepScheduler" in PPOScheduler._compatibles
    for n in (4, 8):
        s.set_timesteps(n); t.set_timesteps(n)
        assert s.timesteps.tolist() == t.timesteps.tolist()
    # trained_betas as an array survives the JSON round trip
    b = np.linspace(1e-4, 2e-2, 1000).astype(np.float32)
    sb = PPOScheduler(trained_betas=b, order_dim=2, scaler_dim=0, factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
    sb.save_config(os.path.join(d, "tb"))
    tb = PPOScheduler.from_pretrained(os.path.join(d, "tb"))
    assert np.allclose(tb.betas.numpy(), b, rtol=1e-6)
    # FLUX: hub id unreachable -> the published Kontext config + overrides; a local folder -> loaded through diffusers
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f = FMPPOScheduler.from_pretrained("black-forest-labs/FLUX.1-Kontext-dev", subfolder="scheduler", order_dim=2, scaler_dim=0, mu_dim=0,
                                           factor_net_kwargs=dict(embedding_dim=8, hidden_dim=16, num_actions=5))
    assert f.config.shift == 3.0 and f.config.use_dynamic_shifting and f.config.order_dim == 2 and f.config.get("max_shift") == 1.15
    f.save_pretrained(os.path.join(d, "flux", "scheduler"))
    g = FMPPOScheduler.from_pretrained(os.path.join(d, "flux"), subfolder="scheduler", order_dim=3)
    assert g.config.order_dim == 3 and g.config.shift == 3.0 and g.config.mu_dim == 0
    f.set_timesteps(8, mu=1.0); g.set_timesteps(8, mu=1.0)
    assert np.array_equal(f.sigmas.numpy(), g.sigmas.numpy())
    print("OK")
    """)


def test_standalone_mixins_round_trip_and_cross_load(tmp_path):
    """without diffusers (this process): the same surface, the same file, and a file written here loads under the fake diffusers."""
    import consolver_amd
    from consolver_amd import PPOScheduler, FMPPOScheduler
    from consolver_amd._scheduler_base import ConfigMixin, SchedulerMixin
    assert not consolver_amd.HAVE_DIFFUSERS
    assert issubclass(PPOScheduler, SchedulerMixin) and issubclass(FMPPOScheduler, ConfigMixin)
    s = PPOScheduler(beta_end=0.012, beta_schedule="scaled_linear", beta_start=0.00085, steps_offset=1, timestep_spacing="trailing",
                     order_dim=3, scaler_dim=1, factor_net_kwargs=dict(embedding_dim=16, hidden_dim=32, num_actions=11))
    assert s.config.order_dim == 3 and s.config.get("nope", 7) == 7 and s.config["scaler_dim"] == 1
    with pytest.raises(TypeError):
        s.config["order_dim"] = 2
    d = str(tmp_path / "sd")
    s.save_pretrained(d)
    t = PPOScheduler.from_pretrained(d)
    pub = lambda c: {k: v for k, v in dict(c).items() if not k.startswith("_")}
    assert pub(t.config) == pub(s.config)
    u = PPOScheduler.from_config(s.config, order_dim=2)
    assert u.config.order_dim == 2 and u.factor_net.order_dim == 2
    assert s.compatibles == [PPOScheduler]
    with pytest.raises(EnvironmentError):
        PPOScheduler.from_pretrained(str(tmp_path / "missing"))
    # what StableDiffusionPipeline.__init__ does when steps_offset != 1: replace _internal_dict; the scheduler follows
    s0 = PPOScheduler(steps_offset=0, timestep_spacing="leading", order_dim=2, scaler_dim=0, factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
    s0.set_timesteps(8)
    before = s0.timesteps.tolist()
    cfg = dict(s0.config); cfg["steps_offset"] = 1
    s0._internal_dict = type(s0.config)(cfg)
    s0.set_timesteps(8)
    assert s0.timesteps.tolist() == [v + 1 for v in before]
    # the stand-alone file loads under (fake) diffusers and the other way round
    out = run_script(f"""
    t = PPOScheduler.from_pretrained({d!r})
    assert t.config.order_dim == 3 and t.config.factor_net_kwargs["num_actions"] == 11 and t.config.timestep_spacing == "trailing"
    t.save_pretrained({str(tmp_path / "sd2")!r})
    print("OK")
    """)
    assert "OK" in out
    v = PPOScheduler.from_pretrained(str(tmp_path / "sd2"))
    assert pub(v.config) == pub(s.config)
    # frozen, but copyable and picklable like diffusers' FrozenDict: deepcopy / pickle of a config and of a whole scheduler
    import copy, pickle
    c2, c3, c4 = copy.copy(s.config), copy.deepcopy(s.config), pickle.loads(pickle.dumps(s.config))
    assert dict(c2) == dict(s.config) == dict(c3) == dict(c4) and type(c3) is type(s.config)
    with pytest.raises(TypeError):
        c3["order_dim"] = 9
    s2 = copy.deepcopy(s)
    assert pub(s2.config) == pub(s.config) and s2.factor_net is not s.factor_net
    # FMPPOScheduler.from_pretrained without diffusers: a hub id falls back to the published Kontext config WITH a warning, a path to a JSON file is read,
    # a filesystem-looking path that does not exist raises instead of silently yielding plausible-but-unintended shift parameters
    with pytest.warns(UserWarning):
        f = FMPPOScheduler.from_pretrained("black-forest-labs/FLUX.1-Kontext-dev", subfolder="scheduler", order_dim=2, scaler_dim=0, mu_dim=0,
                                           factor_net_kwargs=dict(embedding_dim=8, hidden_dim=16, num_actions=5))
    assert f.config.shift == 3.0 and f.config.order_dim == 2
    f.save_pretrained(str(tmp_path / "flux" / "scheduler"))
    g = FMPPOScheduler.from_pretrained(str(tmp_path / "flux" / "scheduler" / "scheduler_config.json"), order_dim=2, scaler_dim=0, mu_dim=0,
                                       factor_net_kwargs=dict(embedding_dim=8, hidden_dim=16, num_actions=5))
    assert g.config.shift == f.config.shift and g.config.get("max_shift") == f.config.get("max_shift")
    for bad in (str(tmp_path / "no_such_dir"), "./typo/scheduler", "a/b/c"):
        with pytest.raises(EnvironmentError):
            FMPPOScheduler.from_pretrained(bad, subfolder="scheduler")


@pytest.mark.gpu
def test_pipeline_call_loop_under_diffusers_gpu():
    """the denoising loop of StableDiffusionPipeline.__call__ (fake_diffusers restates it) driving the HIP UNet and the HIP scheduler
    through the plain protocol (CFG combined by the pipeline, ``scheduler.step(noise_pred, t, latents, return_dict=False)[0]``),
    against the native engine's fused path with the same replayed action indices."""
    out = run_script("""
    from consolver_amd.unet import HipUNet2DConditionModel
    from consolver_amd.engine import SDSamplingEngine
    from consolver_amd.synth import synthetic_unet_state_dict, synthetic_prompt_embeds
    dev = torch.device("cuda:0")
    unet = HipUNet2DConditionModel(dict(layers_per_block=1, sample_size=16), device=dev)
    unet.load_state_dict(synthetic_unet_state_dict(unet.manifest(), seed=1))
    sch = PPOScheduler(beta_end=0.012, beta_schedule="scaled_linear", beta_start=0.00085, steps_offset=1, timestep_spacing="trailing",
                       order_dim=4, scaler_dim=0, factor_net_kwargs=dict(hidden_dim=32, num_actions=11))
    pipe = StableDiffusionPipeline.from_pretrained("x", scheduler=sch, torch_dtype=torch.float16, safety_checker=None,
                                                   components=dict(vae=None, text_encoder=None, tokenizer=None, unet=unet))
    pipe.scheduler.factor_net.to(dev)
    B, n, cfg = 2, 4, 3.0
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, 11, (n, B, 3), generator=g)
    pe, ne = synthetic_prompt_embeds(B, seed=11).half().to(dev), synthetic_prompt_embeds(B, seed=12).half().to(dev)
    noise = torch.randn(B, 4, 16, 16, generator=g).half().to(dev)

    class Replay:                      # per-step forced indices, advanced by the scheduler's own step counter
        pass
    net = pipe.scheduler.factor_net
    calls = {"i": 0}
    orig_step = sch.step
    def step(*a, **k):
        net.forced_action_idx = idx[calls["i"]].to(dev); calls["i"] += 1
        return orig_step(*a, **k)
    sch.step = step
    lat = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=noise.clone(), num_inference_steps=n, guidance_scale=cfg).images
    assert calls["i"] == n
    sch.step = orig_step
    # native fused path
    sch.set_timesteps(n, device=dev)
    x = noise.clone(); ctx = torch.cat([ne, pe])
    for i, t in enumerate(sch.timesteps):
        net.forced_action_idx = idx[i].to(dev)
        eps = unet(x, t.float().reshape(1), encoder_hidden_states=ctx, dup=2, reuse_kv=(i > 0))[0]
        x = sch.step(eps[B:], t, x, return_dict=False, eps_uncond=eps[:B], guidance_scale=cfg)[0]
    err = float((lat.float() - x.float()).norm() / x.float().norm())
    print("ERR", err)
    assert torch.isfinite(lat.float()).all() and err < 3e-3, err
    """)
    assert "ERR" in out
