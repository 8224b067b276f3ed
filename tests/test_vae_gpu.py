"""HIP AutoencoderKL decoder (cs_vae_decode through the C ABI) and decode_latents (utils.py:6-34)
vs the torch-fp32 oracle restatement on identical seeded weights.

The decoder network's oracle is "parity unpinned" w.r.t. diffusers (oracle/vae_oracle.py); decode_latents
itself (scaling, chunking, [0, 1] map) is the reference's code.  Tolerance: ~60 fp16-stored kernels in a
row -> relative L2 of the image <= 5e-3 and every pixel within 2e-2 of the fp32 image (range [0, 1]).
"""
import pytest
import torch

from consolver_amd.vae import HipAutoencoderKL, decode_latents
from consolver_amd.synth import synthetic_vae_state_dict
from oracle import vae_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def build(cfg_over, seed=5):
    v = HipAutoencoderKL(cfg_over, device=DEV)
    man = v.manifest()
    assert man == vae_oracle.vae_manifest(vars(v.config))       # the library and the oracle agree on names and shapes
    sd = synthetic_vae_state_dict(man, seed=seed)
    v.load_state_dict(sd)
    return v, vae_oracle.VaeOracle(sd, vars(v.config))


def test_reduced_decoder_matches_oracle():
    v, orc = build(dict(block_out_channels=(128, 256, 512, 512), layers_per_block=1, sample_size=16))
    g = torch.Generator().manual_seed(1)
    z = torch.randn(3, 4, 16, 16, generator=g)
    got = v.decode(z.half().to(DEV), return_dict=False)[0]
    want = orc.decode(z.half().float())[0]
    assert got.shape == (3, 3, 128, 128) and got.dtype == torch.float16
    err = rel_l2(got, want)
    print("reduced vae rel l2", err)
    assert err < 5e-3, err
    assert v.decode(z.half().to(DEV), return_dict=True).sample.shape == got.shape


def test_decode_latents_chunks_and_postprocess():
    v, orc = build(dict(layers_per_block=1, sample_size=16), seed=9)
    g = torch.Generator().manual_seed(2)
    lat = (torch.randn(5, 4, 16, 16, generator=g) * 0.18215 * 2).half()
    want = vae_oracle.decode_latents(orc, lat.float(), batch_size=2)
    for bs in (1, 2, 5, 8):                                       # ragged last chunk, one chunk, chunk > N
        got = decode_latents(v, lat.to(DEV), batch_size=bs)
        assert got.shape == (5, 3, 128, 128)
        assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0
        assert (got.float().cpu() - want).abs().max() < 2e-2
        assert rel_l2(got, want) < 5e-3
    frac_clamped = float(((want == 0) | (want == 1)).float().mean())
    print("clamped fraction", frac_clamped)
    # empty input
    assert decode_latents(v, lat[:0].to(DEV), batch_size=2).shape == (0, 3, 128, 128)
    with pytest.raises(ValueError):
        decode_latents(v, lat.to(DEV), batch_size=0)
    with pytest.raises(ValueError):
        v.decode(torch.zeros(1, 4, 8, 8, device=DEV, dtype=torch.float16))


def test_full_sd15_decoder_matches_oracle():
    v, orc = build({}, seed=11)
    g = torch.Generator().manual_seed(4)
    lat = (torch.randn(1, 4, 64, 64, generator=g) * 0.18215).half()
    got = decode_latents(v, lat.to(DEV), batch_size=1)
    want = vae_oracle.decode_latents(orc, lat.float(), batch_size=1)
    assert got.shape == (1, 3, 512, 512)
    err = rel_l2(got, want)
    print("full vae rel l2", err, "max abs", float((got.float().cpu() - want).abs().max()))
    assert err < 5e-3, err
    assert (got.float().cpu() - want).abs().max() < 2e-2
    # FLOP count: ~1.26 TMAC = 2.5 TFLOP per 512x512 image for the SD1.5 decoder
    assert 2.3e12 < v.flops(1) < 2.7e12
    # round 6: the three upsamplers ran the sub-pixel form (knob up_fold, default on: the decoder's stream is one fp16 plane); the fused-upsample kernels give
    # the same image within the one extra fp16 rounding of the summed filter taps
    from consolver_amd import ops
    ops.set_tuning("up_fold", 0)
    try:
        plain = decode_latents(v, lat.to(DEV), batch_size=1)
    finally:
        ops.set_tuning("up_fold", 1)
    e_plain = rel_l2(plain, want)
    print("  fused-upsample kernels:", e_plain, "between the two forms:", rel_l2(got, plain.float().cpu()))
    assert not torch.equal(plain, got)
    assert err < 1.1 * e_plain + 1e-4 and rel_l2(got, plain.float().cpu()) < 2e-3


def test_product_path_has_no_cpu_fallback():
    class NotHip:
        config = vae_oracle._Cfg(dict(scaling_factor=0.18215, out_channels=3))
    with pytest.raises(RuntimeError):
        decode_latents(NotHip(), torch.zeros(1, 4, 16, 16), batch_size=1)


def test_flux_vae_16_channel_latents_and_long_softmax_rows():
    """FLUX-side decode (edit_ppo/utils.py:11-28): 16 latent channels, no post_quant_conv, scaling + shift, packed token latents.
    Reduced depth; sample 32 -> 1024 tokens in the mid attention; a second config with 96 x 96 latents (9216 tokens) exercises the
    long-row softmax kernel."""
    from consolver_amd.vae import flux_decode_latents, FLUX_VAE_CONFIG
    from consolver_amd.flux import pack_latents
    cfg = dict(FLUX_VAE_CONFIG); cfg.update(layers_per_block=1, sample_size=32)
    v, orc = build(cfg, seed=21)
    g = torch.Generator().manual_seed(6)
    lat = (torch.randn(2, 16, 32, 32, generator=g) * 0.3611).half()
    packed = pack_latents(lat)                                       # [2, 256, 64] tokens like the sampling loop holds them
    got = flux_decode_latents(v, packed.to(DEV), height=256, width=256)
    want = vae_oracle.flux_decode_latents(orc, lat.float())
    assert got.shape == (2, 3, 256, 256)
    err = rel_l2(got, want)
    print("flux vae rel l2", err)
    assert err < 5e-3 and (got.float().cpu() - want).abs().max() < 2e-2
    cfg.update(sample_size=96, block_out_channels=(128, 128, 128, 128))
    v2, orc2 = build(cfg, seed=22)
    lat2 = (torch.randn(1, 16, 96, 96, generator=g) * 0.3611).half()
    got2 = v2.decode(lat2.to(DEV))[0]
    want2 = orc2.decode(lat2.float())[0]
    assert rel_l2(got2, want2) < 5e-3


@pytest.mark.parametrize("flavour", ["sd", "flux"])
def test_vae_encoder_mode_matches_oracle(flavour):
    """edit_ppo/pipeline.py:613-623 ``_encode_vae_image`` (argmax of the posterior, shift, scale): reduced depth, 128 x 128 images;
    SD flavour (4 latent channels, quant_conv mixes the 8 moments) and FLUX flavour (16 channels, no quant convs);
    then encode -> decode through the same handle."""
    from consolver_amd.vae import encode_image_latents, FLUX_VAE_CONFIG, SD15_VAE_CONFIG
    base = dict(SD15_VAE_CONFIG if flavour == "sd" else FLUX_VAE_CONFIG)
    base.update(layers_per_block=1, sample_size=16, with_encoder=True)
    v, orc = build(base, seed=31)
    g = torch.Generator().manual_seed(8)
    img = (torch.rand(3, 3, 128, 128, generator=g) * 2 - 1).half()
    got = encode_image_latents(v, img.to(DEV))
    want = vae_oracle.encode_image_latents(orc, img.float())
    assert got.shape == (3, base["latent_channels"], 16, 16) and got.dtype == torch.float16
    err = rel_l2(got, want)
    print(flavour, "vae encoder rel l2", err)
    assert err < 5e-3, err
    mode = v.encode(img.to(DEV)).latent_dist.mode()
    assert rel_l2(mode, orc.encode_mode(img.float())) < 5e-3
    rec = v.decode(mode)[0]
    assert rec.shape == (3, 3, 128, 128) and rel_l2(rec, orc.decode(mode.float().cpu())[0]) < 5e-3
    with pytest.raises(NotImplementedError):
        v.encode(img.to(DEV)).latent_dist.sample()
    plain = HipAutoencoderKL(dict(layers_per_block=1, sample_size=16), device=DEV)
    plain.load_state_dict(synthetic_vae_state_dict(plain.manifest(), seed=1))
    with pytest.raises(RuntimeError):
        plain.encode(img.to(DEV))
