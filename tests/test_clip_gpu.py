"""HIP CLIP text encoder (cs_clip_encode through the C ABI) vs the golden vectors of the installed third-party
transformers.CLIPTextModel (reduced config) and vs the fp32 oracle at the full CLIP-L size; the causal head-64
attention op on its own; the prompt-embedding cache format.

Tolerance: 12 layers of fp16 activations -> relative L2 <= 5e-3 (measured ~1e-3)."""
import numpy as np
import pytest
import torch

from consolver_amd import _lib as L
from consolver_amd.text_encoder import HipCLIPTextModel, save_prompt_cache, load_prompt_cache
from consolver_amd.synth import synthetic_clip_state_dict
from oracle.clip_oracle import ClipTextOracle, clip_manifest

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_reduced_clip_matches_transformers_golden(golden):
    g = golden["clip_text"]
    V, D, I, NL, H, P = [int(v) for v in g["cfg"]]
    cfg = dict(vocab_size=V, hidden_size=D, intermediate_size=I, num_hidden_layers=NL, num_attention_heads=H, max_position_embeddings=P)
    m = HipCLIPTextModel(cfg, device=DEV)
    assert m.manifest() == clip_manifest(cfg)
    sd = {"text_model." + k[2:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("w_")}   # prefixed names load too
    m.load_state_dict(sd)
    for name in ("full", "short"):
        ids = torch.from_numpy(np.asarray(g[f"{name}_ids"])).to(DEV)
        res = m(ids)
        out = res[0]
        assert out.dtype == torch.float16 and out.shape == g[f"{name}_out"].shape
        assert res.last_hidden_state is out and rel_l2(res.pooler_output.float().cpu().numpy(), g[f"{name}_pooled"]) < 5e-3
        err = rel_l2(out.float().cpu().numpy(), g[f"{name}_out"])
        print("clip reduced", name, err)
        assert err < 5e-3, err
    assert m(ids[:0])[0].shape == (0, ids.shape[1], D)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 78, dtype=torch.long, device=DEV))        # longer than max_position_embeddings


def test_full_clip_l_matches_oracle():
    m = HipCLIPTextModel(device=DEV)
    sd = synthetic_clip_state_dict(m.manifest(), seed=3)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 49407, (3, 77), generator=g)
    ids[:, 0] = 49406
    ids[0, 10:] = 49407
    out = m(ids.to(DEV))[0]
    torch.set_num_threads(16)
    want = ClipTextOracle(sd)(ids)[0]
    assert out.shape == (3, 77, 768)
    err = rel_l2(out.float().cpu().numpy(), want.numpy())
    print("clip-L rel l2", err)
    assert err < 5e-3, err
    assert abs(m.flops(1) / 2e9 - 6.6) < 0.4            # ~6.6 GMAC per prompt (SURVEY f-2)


@pytest.mark.parametrize("B,H,N", [(2, 12, 77), (1, 2, 20), (3, 1, 130), (1, 4, 64)])
def test_causal_attention_head64(B, H, N):
    import ctypes as C
    g = torch.Generator().manual_seed(N)
    q, k, v = (torch.randn(B, N, H * 64, generator=g).half().to(DEV) for _ in range(3))
    out = torch.empty_like(q)
    L.check(L.lib().cs_op_attention_causal(L.ptr(q), H * 64, L.ptr(k), H * 64, L.ptr(v), H * 64, L.ptr(out), H * 64, B, H, N, 64,
                                           0.125, L.stream_ptr(q.device)))
    qf, kf, vf = (t.float().cpu().view(B, N, H, 64).transpose(1, 2) for t in (q, k, v))
    mask = torch.full((N, N), float("-inf")).triu(1)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * 0.125 + mask, -1) @ vf).transpose(1, 2).reshape(B, N, H * 64)
    assert rel_l2(out.float().cpu().numpy(), ref.numpy()) < 2e-3
    assert float((out.float().cpu() - ref).abs().max()) < 1e-2


def test_prompt_cache_roundtrip(tmp_path):
    prompts = ["a photo of a cat", "mi355x été", "x"]
    pe, ne = torch.randn(3, 77, 768).half(), torch.randn(3, 77, 768).half()
    path = str(tmp_path / "prompts.safetensors")
    save_prompt_cache(path, prompts, pe.to(DEV), ne.to(DEV))
    p2, pe2, ne2 = load_prompt_cache(path, device=DEV)
    assert p2 == prompts and torch.equal(pe2.cpu(), pe) and torch.equal(ne2.cpu(), ne)
    p3, pe3, ne3 = load_prompt_cache(path, start=1, end=3)             # a rank's contiguous shard
    assert p3 == prompts[1:] and torch.equal(pe3, pe[1:]) and torch.equal(ne3, ne[1:])
    save_prompt_cache(path, prompts, pe)
    assert load_prompt_cache(path)[2] is None
    with pytest.raises(ValueError):
        save_prompt_cache(path, prompts[:2], pe)


def test_t5_encoder_matches_transformers_golden(golden):
    """FLUX text_encoder_2 (edit_ppo/pipeline.py:279-330): HIP T5 encoder vs the golden vectors of the installed third-party
    transformers.T5EncoderModel (reduced config, 64- and 200-token sequences).  bf16 storage: rel L2 <= 2e-2 (f16: 3e-3)."""
    from consolver_amd.text_encoder import HipT5EncoderModel
    from oracle.t5_oracle import t5_manifest
    g = golden["t5_encoder"]
    V, D, dk, H, I, NL, NB, MD = [int(v) for v in g["cfg"]]
    cfg = dict(vocab_size=V, d_model=D, d_kv=dk, num_heads=H, d_ff=I, num_layers=NL, relative_attention_num_buckets=NB,
               relative_attention_max_distance=MD)
    sd = {k[2:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("w_")}
    for dt, tol in ((torch.float16, 3e-3), (torch.bfloat16, 2e-2)):
        m = HipT5EncoderModel(cfg, device=DEV, dtype=dt)
        assert m.manifest() == t5_manifest(cfg)
        m.load_state_dict(sd)
        for name in ("short", "long"):
            ids = torch.from_numpy(np.asarray(g[f"{name}_ids"])).to(DEV)
            out = m(ids)[0]
            assert out.dtype == dt and out.shape == g[f"{name}_out"].shape
            err = rel_l2(out.float().cpu().numpy(), g[f"{name}_out"])
            print("t5", dt, name, err)
            assert err < tol, (dt, name, err)


def test_t5_xxl_width_two_blocks_matches_oracle():
    """full T5-XXL widths (d_model 4096, 64 heads, d_ff 10240), two blocks, 512 tokens, synthetic weights, bf16"""
    from consolver_amd.text_encoder import HipT5EncoderModel
    from oracle.t5_oracle import T5EncoderOracle
    cfg = dict(num_layers=2, vocab_size=1000)
    m = HipT5EncoderModel(cfg, device=DEV)
    g = torch.Generator().manual_seed(5)
    sd = {}
    for name, shape in m.manifest():
        if name.endswith("layer_norm.weight"):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif "relative_attention_bias" in name or name == "shared.weight":
            sd[name] = torch.randn(shape, generator=g)
        else:
            sd[name] = torch.randn(shape, generator=g) * ((0.3 if name.endswith("SelfAttention.q.weight") else 1.0) / shape[1] ** 0.5)
    m.load_state_dict(sd)
    ids = torch.randint(0, 1000, (1, 512), generator=g)
    out = m(ids.to(DEV))[0]
    torch.set_num_threads(16)
    want = T5EncoderOracle(sd, m.config)(ids)[0]
    err = rel_l2(out.float().cpu().numpy(), want.numpy())
    print("t5-xxl width rel l2", err)
    assert out.shape == (1, 512, 4096) and err < 2e-2, err
