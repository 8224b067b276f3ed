"""CPU-side checks of the product's host logic and of the C-ABI surface (no GPU compute)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import consolver_amd
from consolver_amd import _lib, tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    """the ABI checks need the in-tree library; hipcc cross-compiles it without a GPU (a fresh checkout has no .so: build artefacts are git-ignored)"""
    if not os.path.exists(_lib.LIB_PATH):
        from consolver_amd.build import build
        build()


def test_library_exports_every_declared_symbol():
    _ensure_built()
    hdr = "".join(open(os.path.join(ROOT, "include", f)).read() for f in sorted(os.listdir(os.path.join(ROOT, "include"))))
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    l = _lib.lib()
    assert l.cs_abi_version() == 2 and l.cs_target_arch() == b"gfx950"
    assert l.cs_error_string(-6) == b"not implemented"


def test_upsampler_subpixel_filter_pack_is_the_upsampled_conv():
    """cs_op_conv_up_fold_pack (host code behind the C ABI): the four 2 x 2-tap phase filters it writes are the per-neighbour SUMS of the 3 x 3 taps, rounded to fp16
    once, and a conv of the INPUT with them, scattered to the phases, is F.interpolate(nearest, x2) + conv2d(pad 1) of the reference graph (diffusers Upsample2D)."""
    import torch
    import torch.nn.functional as F
    from consolver_amd import ops
    g = torch.Generator().manual_seed(0)
    N, C, H = 16, 8, 6
    w = torch.randn(N, C, 3, 3, generator=g).half()
    ws = ops.conv_up_fold_pack(ops.pack_conv_weight(w))
    assert ws.shape == (4, N, 4 * C) and ws.dtype == torch.float16
    ws = ws.float().reshape(4, N, 4, C)
    taps = {0: [[0], [1, 2]], 1: [[0, 1], [2]]}                 # phase -> neighbour -> filter taps that land on it
    x = torch.randn(2, C, H, H, generator=g, dtype=torch.float64)
    xp = F.pad(x, (1, 1, 1, 1))
    exact = torch.zeros(2, N, 2 * H, 2 * H, dtype=torch.float64)
    packed = torch.zeros_like(exact)
    for py in (0, 1):
        for px in (0, 1):
            for a in (0, 1):
                for b in (0, 1):
                    wk = sum(w.double()[:, :, dy, dx] for dy in taps[py][a] for dx in taps[px][b])
                    assert torch.equal(ws[2 * py + px, :, 2 * a + b, :], wk.float().half().float())
                    sl = xp[:, :, py + a:py + a + H, px + b:px + b + H]       # input rows y - 1 + py + a
                    exact[:, :, py::2, px::2] += torch.einsum("nc,bchw->bnhw", wk, sl)
                    packed[:, :, py::2, px::2] += torch.einsum("nc,bchw->bnhw", ws[2 * py + px, :, 2 * a + b, :].double(), sl)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w.double(), padding=1)
    assert float((exact - ref).abs().max()) < 1e-12
    assert float((packed - ref).norm() / ref.norm()) < 4e-4        # the one fp16 rounding of the summed taps


def test_struct_layout_matches_header():
    # field order / sizes mirrored by hand in _lib.py; a C-side sizeof check guards it
    assert ctypes.sizeof(_lib.CsFactorNet) == 6 * 8 + 4 * 4 + 2 * 4
    assert ctypes.sizeof(_lib.CsStepArgs) % 8 == 0
    l = _lib.lib()
    a = _lib.CsStepArgs()
    a.B, a.elems = 1, 8
    # argument validation happens before any launch -> usable without a GPU
    assert l.cs_lms_ddim_step(ctypes.byref(a), None) == -1
    assert b"required" in l.cs_last_error()
    a.x = a.eps_text = a.x_out = 8
    a.order_dim, a.m, a.scaler_dim = 4, 5, 0
    assert l.cs_lms_ddim_step(ctypes.byref(a), None) == -1 and b"history length" in l.cs_last_error()
    a.m, a.scaler_dim = 1, 3
    assert l.cs_lms_ddim_step(ctypes.byref(a), None) == -6
    a.scaler_dim, a.B = 0, 0
    assert l.cs_lms_ddim_step(ctypes.byref(a), None) == 0     # empty batch is a no-op


@pytest.mark.parametrize("spacing,off", [("trailing", 0), ("leading", 0), ("leading", 1), ("linspace", 0)])
def test_scheduler_timestep_grids_bit_exact(golden, spacing, off):
    g = golden["sd_tables"]
    flat, offs = g[f"ts_{spacing}_off{off}"], g[f"ts_{spacing}_off{off}_offsets"]
    s = consolver_amd.PPOScheduler(timestep_spacing=spacing, steps_offset=off,
                                   factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
    for n in range(1, 51):
        want = flat[offs[n - 1]:offs[n]]
        if want.min() < 0 or want.max() > 999:
            with pytest.raises(ValueError):
                s.set_timesteps(n)
            continue
        s.set_timesteps(n)
        assert s.timesteps.dtype == torch.int64 and np.array_equal(s.timesteps.numpy(), want)
        assert s.ets == [] and s.num_inference_steps == n


def test_scheduler_tables_and_protocol(golden):
    g = golden["sd_tables"]
    s = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                   timestep_spacing="trailing", order_dim=4, scaler_dim=0,
                                   factor_net_kwargs=dict(embedding_dim=32, hidden_dim=256, num_actions=11))
    np.testing.assert_allclose(s.alphas_cumprod.numpy(), g["ac_scaled_linear"], rtol=2e-6)
    assert s.init_noise_sigma == 1.0 and s.order == 1 and "DDIMScheduler" in s._compatibles and len(s) == 1000
    assert s.config.order_dim == 4 and s.config.get("scaler_dim") == 0 and s.config["beta_schedule"] == "scaled_linear"
    x = torch.zeros(1, 4, 8, 8)
    assert s.scale_model_input(x, 5) is x
    with pytest.raises(ValueError):
        s.step(x, 999, x)                      # before set_timesteps (scheduler_ppo.py:199-200)
    with pytest.raises(ValueError):
        s.set_timesteps(1001)
    with pytest.raises(ValueError):
        s.set_timesteps(61)                    # reference grid would end in -1
    s.set_timesteps(8)
    with pytest.raises(RuntimeError, match="no|CUDA|HIP"):
        s.step(x, 999, x)                      # CPU tensors: fail loudly, never fall back
    # state dict layout (train_ppo.py:174-190 checkpoints)
    sd = s.factor_net.state_dict()
    assert list(sd) == ["action_values", "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias",
                        "mlp.4.weight", "mlp.4.bias"]
    assert sd["action_values"].shape == (3, 11) and sd["mlp.4.weight"].shape == (33, 256)
    assert sum(v.numel() for k, v in sd.items() if k != "action_values") == 75041
    assert float(sd["mlp.4.weight"].abs().max()) == 0.0        # zero-init last layer
    with pytest.raises(NotImplementedError):
        consolver_amd.PPOScheduler(beta_schedule="nope")
    with pytest.raises(ValueError):
        consolver_amd.PPOScheduler(timestep_spacing="nope", factor_net_kwargs=dict(hidden_dim=8, num_actions=3)).set_timesteps(4)


def test_action_value_grids(golden):
    g = golden["sd_factor_net"]
    for ci, (o, sc, uc, K, H) in enumerate(g["cases"]):
        net = consolver_amd.FactorNetPPO(hidden_dim=int(H), num_actions=int(K), order_dim=int(o), scaler_dim=int(sc),
                                         use_conv=bool(uc))
        np.testing.assert_allclose(net.action_values.numpy(), g[f"c{ci}_w_action_values"], atol=2e-7, rtol=0)
        assert net.mlp[0].in_features == 2 + (o - 1 if uc else 0)
    g = golden["flux"]
    for ci, (o, sc, mu, uc, K, H) in enumerate(g["fn_cases"]):
        net = consolver_amd.FluxFactorNetPPO(hidden_dim=int(H), num_actions=int(K), order_dim=int(o),
                                             scaler_dim=int(sc), mu_dim=int(mu), use_conv=bool(uc))
        np.testing.assert_allclose(net.action_values.numpy(), g[f"f{ci}_w_action_values"], atol=2e-7, rtol=0)
        assert net.action_dims == o + sc + mu - 1


def test_flux_scheduler_tables(golden):
    g = golden["flux"]
    for n in range(2, 9):
        s = consolver_amd.FMPPOScheduler.from_pretrained("black-forest-labs/FLUX.1-Kontext-dev", subfolder="scheduler",
                                                         order_dim=2, scaler_dim=0, mu_dim=0,
                                                         factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
        assert s.config.get("base_image_seq_len", 256) == 256 and s.config.use_dynamic_shifting
        s.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=1.15)
        np.testing.assert_allclose(s.sigmas.numpy(), g[f"sig_dyn_n{n}"], rtol=2e-7)
        np.testing.assert_allclose(s.timesteps.numpy(), g[f"ts_dyn_n{n}"], rtol=2e-7)
        assert s.step_index is None and s.begin_index is None
        s2 = consolver_amd.FMPPOScheduler(shift=3.0, order_dim=2, scaler_dim=0, mu_dim=0,
                                          factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
        s2.set_timesteps(n)
        np.testing.assert_allclose(s2.sigmas.numpy(), g[f"sig_static_n{n}"], rtol=2e-7)
    s = consolver_amd.FMPPOScheduler(use_dynamic_shifting=True, order_dim=2, scaler_dim=0, mu_dim=0,
                                     factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
    with pytest.raises(ValueError):
        s.set_timesteps(sigmas=[1.0, 0.5])                      # mu missing
    with pytest.raises(ValueError):
        s.set_timesteps(3, sigmas=[1.0, 0.5], mu=1.0)           # length mismatch
    with pytest.raises(ValueError):
        s.step(torch.zeros(1, 4, 4), 1.0, torch.zeros(1, 4, 4))  # before set_timesteps
    s.set_timesteps(sigmas=[1.0, 0.5], mu=1.0)
    with pytest.raises(ValueError):
        s.step(torch.zeros(1, 4, 4), 3, torch.zeros(1, 4, 4))    # integer timestep
    assert abs(tables.calculate_shift(4096) - 1.15) < 1e-12


def test_teacher_pair_dataset_roundtrip(tmp_path):
    """on-disk teacher-pair format (generate_data.py:180-213 / data_processing.py:10-83)"""
    import torch
    from consolver_amd import ppo_data as pd
    g = torch.Generator().manual_seed(0)
    want = {}
    for i in range(5):
        sid = pd.teacher_pair_id(3, i)
        assert sid == f"3_{i:08d}"
        noise, lat = torch.randn(4, 8, 8, generator=g), torch.randn(4, 8, 8, generator=g)
        pd.save_teacher_pair(str(tmp_path), sid, f"prompt {i}\n", noise, lat)
        want[sid] = (f"prompt {i}", noise, lat)
    assert sorted(os.listdir(tmp_path))[:2] == ["3_00000000.txt", "3_00000001.txt"]
    ds = pd.TeacherPairDataset(str(tmp_path), strict=True)
    assert len(ds) == 5
    for i in range(5):
        text, noise, lat = ds[i]
        t, n, l = want[ds.ids[i]]
        assert text == t and torch.equal(noise, n) and torch.equal(lat, l)
    text, noise, lat = pd.collate_teacher_pairs([ds[i] for i in range(4)])
    assert noise.shape == (4, 4, 8, 8) and len(text) == 4
    t2, n2, l2 = pd.repeat_random_sample((text, noise, lat))
    assert len(set(t2)) == 1 and all(torch.equal(n2[0], n2[k]) for k in range(4)) and l2.shape == lat.shape
    j = text.index(t2[0])
    assert torch.equal(n2[0], noise[j]) and torch.equal(l2[0], lat[j])
    # NaN latents are rejected at write time and skipped (replaced by another sample) at read time unless strict
    with pytest.raises(ValueError):
        pd.save_teacher_pair(str(tmp_path), "9_00000000", "x", noise[0], torch.full((4, 8, 8), float("nan")))
    torch.save(torch.full((4, 8, 8), float("nan")), os.path.join(tmp_path, "latent_3_00000002.pth"))
    with pytest.raises(FileNotFoundError):
        ds[2]
    assert pd.TeacherPairDataset(str(tmp_path))[2][0].startswith("prompt")


def test_evaluation_harness_files_pairs_statistics(tmp_path):
    """gen_ppo.py:318-325 naming, compute_reward.py:52-95 pair finder / loader, :332-365 statistics, :447-462 JSON schema"""
    import json
    import torch
    from consolver_amd import evaluation as ev
    g = torch.Generator().manual_seed(0)
    d1, d2 = tmp_path / "ours", tmp_path / "teacher"
    imgs = {}
    for i in range(3):
        img = torch.rand(3, 16, 24, generator=g)
        png, txt = ev.save_generation(str(d1 / "sub"), 2, i, img, f"prompt {i}")
        assert os.path.basename(png) == f"2_{i:08d}.png" and open(txt).read() == f"prompt {i}"
        imgs[i] = img
        if i < 2:
            ev.save_generation(str(d2 / "sub"), 2, i, img.flip(-1), "x")
    back = ev.load_image_tensor(str(d1 / "sub" / "2_00000001.png"), "cpu")
    assert back.shape == (3, 16, 24) and back.dtype == torch.float32
    assert torch.equal(back, (imgs[1] * 255).round() / 255)                 # 8-bit quantisation, (x * 255).round()
    pairs = ev.find_image_pairs(str(d1), str(d2))
    assert [os.path.basename(a) for a, _ in pairs] == ["2_00000000.png", "2_00000001.png"]      # the unpaired third file is skipped
    assert all(os.path.relpath(a, d1) == os.path.relpath(b, d2) for a, b in pairs)
    st = ev.calculate_statistics({"image_psnr": [10.0, 20.0, 40.0], "clip": []})
    assert st["image_psnr"] == {"mean": 70.0 / 3, "std": float(np.std([10.0, 20.0, 40.0])), "min": 10.0, "max": 40.0, "median": 20.0, "count": 3}
    assert st["clip"] == {"mean": 0.0, "std": 0.0, "min": 0.0, "max": 0.0, "median": 0.0, "count": 0}
    out = ev.write_results(str(tmp_path / "r.json"), {"image_psnr": [1.0, 2.0]}, {"dir1": "a", "dir2": "b", "num_pairs": 2})
    assert sorted(json.load(open(tmp_path / "r.json")).keys()) == ["config", "raw_scores", "statistics"] and out["statistics"]["image_psnr"]["count"] == 2


def test_flux_driver_host_logic(tmp_path):
    """edit_ppo/generate_ours.py: JSONL loading (invalid lines skipped), ceil chunking, unique paths, folder names"""
    from consolver_amd import generate_flux as gf
    from oracle import solver_oracle as so
    p = tmp_path / "d.jsonl"
    p.write_text('{"key": "a", "category": "c/1", "file_name": "x.jpg", "instruction": "i"}\nnot json\n{"key": "b", "category": "c", "file_name": "y.jpg", "instruction": "j"}\n')
    data = gf.load_jsonl(str(p))
    assert [e["key"] for e in data] == ["a", "b"]
    for n, world in [(10, 4), (7, 8), (16, 8), (0, 4), (5, 1)]:
        chunks = gf.chunk_entries(list(range(n)), world)
        assert sum(chunks, []) == list(range(n)) and len(chunks) <= world
        for r in range(world):
            lo, hi = so.shard_bounds_ceil(n, world, r)
            assert gf.shard_for_rank(list(range(n)), world, r) == list(range(lo, hi))
    assert gf.sanitize_folder_name(" style/change: x-1 ") == "style_change__x_1" and gf.sanitize_folder_name("") == "Unknown"
    f = tmp_path / "edited_image.jpg"
    assert gf.ensure_unique_path(str(f)) == str(f)
    f.write_text("x")
    assert gf.ensure_unique_path(str(f)).endswith("edited_image_1.jpg")
    (tmp_path / "edited_image_1.jpg").write_text("x")
    assert gf.ensure_unique_path(str(f)).endswith("edited_image_2.jpg")


def test_forward_process_members_vs_reference(golden):
    """row a6: PPOScheduler.add_noise (scheduler_ppo.py:336-358) and FMPPOScheduler.scale_noise (edit_ppo/scheduler_fmppo.py:457-484)
    against the imported reference (oracle/make_golden.py forward_process).  Plain table look-ups + torch elementwise on the caller's
    tensors (not on the sampling path), so they run on CPU tensors too."""
    import consolver_amd
    g = golden["forward_process_sd"]
    s = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing",
                                   order_dim=4, scaler_dim=0, factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
    out = s.add_noise(torch.from_numpy(g["x"]), torch.from_numpy(g["noise"]), torch.from_numpy(g["t"]))
    np.testing.assert_allclose(out.numpy(), g["noisy"], rtol=1e-6, atol=1e-7)
    g = golden["forward_process_flux"]
    f = consolver_amd.FMPPOScheduler(shift=3.0, use_dynamic_shifting=True, order_dim=2, scaler_dim=0, mu_dim=0,
                                     factor_net_kwargs=dict(hidden_dim=8, num_actions=3))
    f.set_timesteps(sigmas=np.linspace(1.0, 1 / 6, 6), mu=1.15)
    np.testing.assert_allclose(f.sigmas.numpy(), g["sigmas"], rtol=2e-7)
    out = f.scale_noise(torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["noise"]))
    np.testing.assert_allclose(out.numpy(), g["noisy"], rtol=1e-6, atol=1e-7)


PUBLISHED_PARAMS = {   # parameter counts of the REAL checkpoints (literals, not derived from this repo): runwayml/stable-diffusion-v1-5 unet 859,520,964;
    # its AutoencoderKL decoder 49,490,179 + post_quant_conv 20 (whole VAE 83,653,863 = encoder 34,163,592 + quant_conv 72 + these); openai/clip-vit-large-patch14
    # text model 123,060,480; black-forest-labs/FLUX.1-Kontext-dev transformer 11,901,408,320
    "sd15_unet": 859_520_964, "sd15_vae_decoder": 49_490_179 + 20, "clip_l_text": 123_060_480, "flux_kontext_dit": 11_901_408_320}


@pytest.mark.parametrize("name", sorted(PUBLISHED_PARAMS))
def test_weight_manifests_match_published_counts_and_committed_key_lists(name):
    """structural pin of the third-party denoisers (their packages cannot be installed here): the executor's weight manifest
    -- the diffusers / transformers state-dict names and shapes a real checkpoint is loaded by -- has the published parameter count and equals
    the committed list (tools/make_manifests.py) tensor for tensor.  Host only: cs_*_create touches no GPU."""
    import json
    import math
    _ensure_built()
    from consolver_amd.unet import HipUNet2DConditionModel
    from consolver_amd.vae import HipAutoencoderKL
    from consolver_amd.text_encoder import HipCLIPTextModel
    from consolver_amd.flux import HipFluxTransformer2DModel
    ctor = {"sd15_unet": HipUNet2DConditionModel, "sd15_vae_decoder": HipAutoencoderKL, "clip_l_text": HipCLIPTextModel,
            "flux_kontext_dit": HipFluxTransformer2DModel}[name]
    m = ctor(device="cpu").manifest()
    assert sum(math.prod(s) for _, s in m) == PUBLISHED_PARAMS[name]
    want = json.load(open(os.path.join(ROOT, "tests", "golden", name + "_manifest.json")))
    assert want["params"] == PUBLISHED_PARAMS[name]
    assert [[k, list(s)] for k, s in m] == want["tensors"]
    assert len({k for k, _ in m}) == len(m)                    # no duplicate names
    if name == "sd15_unet":
        names = {k for k, _ in m}
        # spot checks of names every SD1.5 checkpoint carries (diffusers UNet2DConditionModel state dict)
        for k in ("conv_in.weight", "time_embedding.linear_1.weight", "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight",
                  "down_blocks.2.resnets.1.time_emb_proj.bias", "mid_block.attentions.0.proj_out.weight", "up_blocks.1.upsamplers.0.conv.weight",
                  "up_blocks.3.resnets.2.conv_shortcut.weight", "up_blocks.3.attentions.2.transformer_blocks.0.ff.net.0.proj.bias", "conv_norm_out.bias"):
            assert k in names, k
        assert dict(m)["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight"] == (320, 768)
        assert dict(m)["up_blocks.0.resnets.0.conv1.weight"] == (1280, 2560, 3, 3)
