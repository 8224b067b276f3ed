"""Split-fp16 residual stream (CS_RESIDUAL_F16X2, include/consolver_hip.h): every kernel that touches a stream tensor, against torch fp32.

A stream tensor is two fp16 planes, value = hi + lo.  The adds onto the stream (conv / GEMM epilogues, the split-K reduce, the fused
cross-attention block) must take hi + lo in fp32 and leave hi + lo equal to the fp32 sum to ~22 bits; the norms must normalise hi + lo.
References are plain torch fp32 ops on the same fp16-rounded operands; the fp16 GEMM operands themselves are exact in both, so what the
tolerances see is fp32 accumulation order (1e-6 class), NOT fp16 storage (5e-4 class): a dropped lo plane fails these by two orders.
The end-to-end effect (the 1e-3 latent gate of north_star) is tests/test_parity_e2e_gpu.py.
"""
import pytest
import torch
import torch.nn.functional as F

from consolver_amd import ops
from consolver_amd.synth import synthetic_prompt_embeds, synthetic_unet_state_dict
from consolver_amd.unet import HipUNet2DConditionModel
from tests._models import get_unet, get_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 3e-6          # fp32 accumulation-order class; fp16 storage of a unit-scale tensor is 2.4e-4 rms


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def nchw(x):
    return x.permute(0, 3, 1, 2).float()


def hi_is_rounding(hi, lo):
    """hi == f16(hi + lo) except where lo (itself rounded to fp16) sits within one of its own ulps of a rounding tie"""
    return float((hi != (hi.float() + lo.float()).half()).float().mean()) < 2e-3


def test_split_representation_is_fp32_class():
    x = rnd(1 << 16, seed=1, scale=3.0, dtype=torch.float32)
    hi, lo = ops.split_f16(x)
    assert rel_l2(hi.float() + lo.float(), x) < 2e-7
    assert rel_l2(hi.float(), x) > 1e-4                      # the hi plane alone is an ordinary fp16 rounding


LINEAR_CASES = [   # M, K, N, bias, res, in_place : every linear / 1x1 kernel family of igemm.hip
    (8192, 320, 320, True, True, True),        # gemm_w8_kernel (256 x 320 tiles), residual in place (to_out)
    (8192, 1280, 320, True, True, False),      # FF2 shape
    (2048, 640, 640, True, True, True),        # 256 x 320, two column tiles
    (8192, 1280, 1280, True, True, True),      # gemm_lw_kernel (256 x 160 tiles, one round)
    (128, 1280, 1280, True, True, True),       # 8 x 8 level: generic tile + split-K reduce
    (320, 192, 128, False, False, False),      # generic 128-wide tile, ragged M, no residual: lo of a plain product (proj_in / conv outputs)
    (8192, 320, 320, True, False, False),      # proj_in: no residual, lo plane requested
]


@pytest.mark.parametrize("case", LINEAR_CASES)
def test_linear_x2(case):
    M, K, N, use_b, use_r, in_place = case
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    b = rnd(N, seed=3, scale=0.1) if use_b else None
    r32 = rnd(M, N, seed=4, scale=2.0, dtype=torch.float32) if use_r else None
    ref = x.float() @ w.float().t()
    if use_b:
        ref = ref + b.float()
    if use_r:
        ref = ref + r32
    rh, rl = ops.split_f16(r32) if use_r else (None, None)
    if in_place:
        oh, ol = ops.linear_x2(x, w, b, res=rh, res_lo=rl, out=rh, out_lo=rl)
    else:
        oh, ol = ops.linear_x2(x, w, b, res=rh, res_lo=rl)
    assert rel_l2(oh.float() + ol.float(), ref) < TOL, case
    assert hi_is_rounding(oh, ol)                                       # hi is the fp16 rounding of the value: the plane a GEMM may read alone
    if use_r:
        # dropping the residual's lo plane is visible at this tolerance (the test would not notice a no-op otherwise)
        oh2, ol2 = ops.linear_x2(x, w, b, res=ops.split_f16(r32)[0], res_lo=None)
        assert rel_l2(oh2.float() + ol2.float(), ref) > 20 * TOL
    # want_lo=False (FF2 in front of proj_out): the fp16 rounding of the same fp32 sum
    if use_r and not in_place:
        o1, none = ops.linear_x2(x, w, b, res=rh, res_lo=rl, want_lo=False)
        assert none is None and torch.equal(o1, oh)


SPLIT_A_CASES = [   # M, c0, c1, N : the resnet shortcut 1x1 over [x | skip] with BOTH operands split-fp16 (every kernel family that serves it)
    (8192, 640, 320, 320),      # gemm_w8_kernel, two sources (up block, 64 x 64 level shape)
    (4096, 320, 0, 640),        # gemm_w8_kernel, one source (down block shortcut)
    (8192, 1280, 1280, 1280),   # gemm_lw_kernel (16 x 16 level)
    (128, 1280, 1280, 1280),    # generic tile + split-K (8 x 8 level)
    (320, 128, 64, 128),        # generic 128-wide tile, ragged M
]


@pytest.mark.parametrize("case", SPLIT_A_CASES)
def test_split_a_operand_1x1(case):
    M, c0, c1, N = case
    x32 = rnd(M, c0, seed=1, scale=2.0, dtype=torch.float32)
    y32 = rnd(M, c1, seed=2, scale=2.0, dtype=torch.float32) if c1 else None
    w = rnd(N, c0 + c1, seed=3, scale=(c0 + c1) ** -0.5)
    b = rnd(N, seed=4, scale=0.1)
    xh, xl = ops.split_f16(x32)
    yh, yl = ops.split_f16(y32) if c1 else (None, None)
    full = torch.cat([x32, y32], -1) if c1 else x32
    ref = full @ w.float().t() + b.float()
    B = 1
    v = lambda t: t.view(B, M, 1, t.shape[-1]) if t is not None else None      # [B, H, W, C] view of the token rows
    oh, ol = ops.conv2d_x2(v(xh), w.view(N, 1, -1).contiguous(), b, x1=v(yh), taps=1, x0_lo=v(xl), x1_lo=v(yl))
    got = (oh.float() + ol.float()).view(M, N)
    assert rel_l2(got, ref) < TOL, case
    # the hi planes alone (what the layer computes without the lo k steps) are two orders away
    ph, pl = ops.conv2d_x2(v(xh), w.view(N, 1, -1).contiguous(), b, x1=v(yh), taps=1)
    assert rel_l2((ph.float() + pl.float()).view(M, N), ref) > 30 * TOL
    if not c1:
        lh, ll = ops.linear_x2(xh, w, b, x_lo=xl)
        assert torch.equal(lh, oh.view(M, N)) and torch.equal(ll, ol.view(M, N))


CONV_CASES = [   # B, H, W, c0, c1, N, taps, stride, up, temb, res
    (2, 64, 64, 64, 0, 320, 9, 1, False, False, True),     # conv3_lw_kernel, conv2 + residual
    (2, 32, 32, 128, 0, 640, 9, 1, False, True, True),     # temb + residual
    (4, 8, 8, 128, 0, 320, 9, 1, False, False, True),      # 8 x 8 level (four images per tile)
    (2, 16, 16, 128, 0, 128, 9, 2, False, False, False),   # downsample conv: generic tile (+ split-K), lo of a plain conv output
    (2, 16, 16, 64, 0, 320, 9, 1, True, False, False),     # upsample conv
    (2, 32, 32, 128, 64, 320, 1, 1, False, False, False),  # shortcut 1x1 over a skip concat
    (1, 8, 8, 640, 0, 1280, 9, 1, False, True, True),      # split-K halo conv at 8 x 8
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_x2(case):
    B, H, W, c0, c1, N, taps, stride, up, use_t, use_r = case
    k = 3 if taps == 9 else 1
    x0 = rnd(B, H, W, c0, seed=1)
    x1 = rnd(B, H, W, c1, seed=2) if c1 else None
    w = rnd(N, c0 + c1, k, k, seed=3, scale=((c0 + c1) * taps) ** -0.5)
    bias = rnd(N, seed=4, scale=0.1)
    temb = rnd(B, N, seed=5, scale=0.5) if use_t else None
    Ho = 2 * H if up else (H // 2 if stride == 2 else H)
    Wo = 2 * W if up else (W // 2 if stride == 2 else W)
    r32 = rnd(B, Ho, Wo, N, seed=6, scale=2.0, dtype=torch.float32) if use_r else None
    rh, rl = ops.split_f16(r32) if use_r else (None, None)
    oh, ol = ops.conv2d_x2(x0, ops.pack_conv_weight(w), bias, x1=x1, taps=taps, stride=stride, upsample=up, temb=temb, res=rh, res_lo=rl)
    xin = nchw(torch.cat([x0, x1], -1) if c1 else x0)
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.float(), bias.float(), stride=stride, padding=k // 2)
    if use_t:
        ref = ref + temb.float()[:, :, None, None]
    if use_r:
        ref = ref + nchw(r32)
    got = nchw(oh) + nchw(ol)
    assert rel_l2(got, ref) < TOL, case
    assert hi_is_rounding(oh, ol)
    # the hi plane equals the plain op's output where that path also rounds once (residual / temb layers): same kernels, same fp32 sum
    if use_r and not use_t:
        plain = ops.conv2d(x0, ops.pack_conv_weight(w), bias, x1=x1, taps=taps, stride=stride, upsample=up, res=rh)
        assert rel_l2(plain.float(), oh.float()) < 6e-4      # differs only by the residual's lo plane


@pytest.mark.parametrize("C,HW,B,c1", [(320, 4096, 2, 0), (640, 1024, 2, 320), (1280, 64, 3, 0), (320, 256, 1, 640)])
def test_group_norm_x2(C, HW, B, c1):
    x32 = rnd(B, HW, C, seed=1, scale=2.0, dtype=torch.float32) + 0.5
    y32 = rnd(B, HW, c1, seed=2, scale=1.5, dtype=torch.float32) if c1 else None
    g, b = (1.0 + 0.1 * rnd(C + c1, seed=3).float()).half(), rnd(C + c1, seed=4, scale=0.1)
    xh, xl = ops.split_f16(x32)
    yh, yl = ops.split_f16(y32) if c1 else (None, None)
    out = ops.group_norm_x2(xh, xl, g, b, 32, 1e-5, True, x1=yh, x1_lo=yl)
    full = torch.cat([x32, y32], -1) if c1 else x32
    ref = F.silu(F.group_norm(full.permute(0, 2, 1), 32, g.float(), b.float(), 1e-5)).permute(0, 2, 1)
    # output is an fp16 tensor (the next conv's operand): compare with the fp16 rounding of the fp32 result
    assert rel_l2(out.float(), ref.half().float()) < 1.5e-4
    # and it must be closer to the fp32 normalisation of hi + lo than a normalisation of the hi planes alone is
    hi_only = ops.group_norm(xh, g, b, 32, 1e-5, True, x1=yh)
    assert rel_l2(out.float(), ref) < rel_l2(hi_only.float(), ref)


@pytest.mark.parametrize("M,C", [(4096, 320), (1024, 640), (300, 1280)])
def test_layer_norm_x2(M, C):
    x32 = rnd(M, C, seed=1, scale=3.0, dtype=torch.float32) + 1.0
    g, b = (1.0 + 0.1 * rnd(C, seed=2).float()).half(), rnd(C, seed=3, scale=0.1)
    xh, xl = ops.split_f16(x32)
    out = ops.layer_norm_x2(xh, xl, g, b)
    ref = F.layer_norm(x32, (C,), g.float(), b.float(), 1e-5)
    assert rel_l2(out.float(), ref.half().float()) < 1.5e-4
    assert rel_l2(out.float(), ref) < rel_l2(ops.layer_norm(xh, g, b).float(), ref)


def test_xattn_block_x2_residual_is_fp32_class():
    """the fused cross-attention block adds onto hi + lo: out_hi + out_lo - (h_hi + h_lo) must equal the plain kernel's delta to fp16-of-the-delta
    precision, i.e. the stream itself is not re-rounded."""
    B, HW, C, Nk = 2, 1024, 320, 77
    M = B * HW
    h32 = rnd(M, C, seed=1, scale=4.0, dtype=torch.float32)
    hh, hl = ops.split_f16(h32)
    g, b = (1.0 + 0.1 * rnd(C, seed=2).float()).half(), rnd(C, seed=3, scale=0.1)
    wq, wo, bo = rnd(C, C, seed=4, scale=C ** -0.5), rnd(C, C, seed=5, scale=C ** -0.5), rnd(C, seed=6, scale=0.1)
    kv = rnd(B, Nk, 2 * C, seed=7)
    oh, ol = ops.xattn_block_x2(hh, hl, g, b, wq, kv, wo, bo, hw=HW)
    plain = ops.xattn_block(hh, g, b, wq, kv, wo, bo, hw=HW)
    # torch fp32 reference of the block on the hi plane's LayerNorm (the kernel normalises the hi plane in both forms)
    n = F.layer_norm(hh.float(), (C,), g.float(), b.float(), 1e-5).half().float()
    q = (n @ wq.float().t()).half().float().view(B, HW, 8, 40).transpose(1, 2)
    k = kv[..., :C].float().view(B, Nk, 8, 40).transpose(1, 2)
    v = kv[..., C:].float().view(B, Nk, 8, 40).transpose(1, 2)
    a = (torch.softmax(q @ k.transpose(-1, -2) * 40 ** -0.5, -1) @ v).transpose(1, 2).reshape(M, C).half().float()
    delta = a @ wo.float().t() + bo.float()
    ref = h32 + delta
    got = oh.float() + ol.float()
    assert rel_l2(got, ref) < 1.2e-4                      # bounded by the fp16 roundings INSIDE the branch (q, P, O, patch), relative to a stream of scale 4
    assert rel_l2(got, ref) < 0.5 * rel_l2(plain.float(), ref)
    assert rel_l2(got - h32, delta) < 2e-3                # the delta itself to fp16 class
    assert hi_is_rounding(oh, ol)


# ---- 8-bit lo planes (round 6): value = hi (fp16) + lo8 (e5m2), 14 significant bits -----------------------------------------------------------------
LO8_TOL = 3e-5      # hi + lo8 of a unit-scale tensor: ~1.5e-5 rms (fp16 alone: 2.4e-4; hi + fp16 lo: 1e-7)


def test_lo8_representation_and_hardware_format():
    """(a) hi + lo8 is a 14-bit representation; (b) the hardware's e5m2 conversion (v_cvt_pk_bf8_f32, round to nearest even) writes the bytes torch.float8_e5m2
    holds: a linear layer whose fp32 result is EXACT (operands on a 2^-8 / 2^-6 grid) must leave exactly torch's (fp16, e5m2) split of that result."""
    x = rnd(1 << 16, seed=1, scale=3.0, dtype=torch.float32)
    hi, lo8 = ops.lo8_split(x)
    e = rel_l2(ops.lo8_value(hi, lo8), x)
    assert 3e-6 < e < LO8_TOL, e
    g = torch.Generator().manual_seed(2)
    M, K, N = 512, 64, 320
    xi = (torch.randint(-1024, 1025, (M, K), generator=g).float() / 256).half().to(DEV)
    wi = (torch.randint(-128, 129, (N, K), generator=g).float() / 64).half().to(DEV)
    ref = xi.float() @ wi.float().t()                                  # multiples of 2^-14 below 2^9: exact in fp32 in any summation order
    assert torch.equal(ref, (xi.double() @ wi.double().t()).float())
    oh, ol8 = ops.linear_lo8(xi, wi)
    wh, wl8 = ops.lo8_split(ref)
    assert torch.equal(oh, wh)
    assert torch.equal(ol8, wl8), float((ol8 != wl8).float().mean())
    assert float((wl8 != 0).float().mean()) > 0.5                      # (the lo plane is exercised: most results do not fit fp16)


@pytest.mark.parametrize("case", [c for c in LINEAR_CASES if c[4]] + [(8192, 320, 320, True, False, False)])
def test_linear_lo8(case):
    """every linear kernel family with 8-bit lo planes: residual hi + lo8 in, hi + lo8 out (to_out), hi only out (the feed-forward's second linear: FAST 4), no
    residual (proj_in).  The reference adds the residual's REPRESENTED value, so the tolerance sees the output's 14-bit storage only."""
    M, K, N, use_b, use_r, _ = case
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    b = rnd(N, seed=3, scale=0.1) if use_b else None
    ref = x.float() @ w.float().t()
    if use_b:
        ref = ref + b.float()
    rh = rl8 = None
    if use_r:
        rh, rl8 = ops.lo8_split(rnd(M, N, seed=4, scale=2.0, dtype=torch.float32))
        ref = ref + ops.lo8_value(rh, rl8)
    oh, ol8 = ops.linear_lo8(x, w, b, res=rh, res_lo8=rl8)
    e = rel_l2(ops.lo8_value(oh, ol8), ref)
    assert e < LO8_TOL, (case, e)
    # hi is A nearest fp16 of the represented value (the plane a GEMM may read alone): |lo8| <= ulp(hi) / 2 -- with equality where the 3-bit lo rounds up to a tie
    ulp = oh.float().abs().clamp_min(2.0 ** -14).log2().floor().exp2() * 2.0 ** -10
    assert bool((ol8.view(torch.float8_e5m2).float().abs() <= 0.5 * ulp).all())
    if use_r:
        o1, none = ops.linear_lo8(x, w, b, res=rh, res_lo8=rl8, want_lo=False)
        assert none is None and torch.equal(o1, oh)                     # the hi-only form stores the same fp16 rounding of the same fp32 sum
        # the 8-bit residual plane is really read: dropping it is visible
        o2, l2 = ops.linear_lo8(x, w, b, res=rh, res_lo8=None)
        assert rel_l2(ops.lo8_value(o2, l2), ref) > 3 * LO8_TOL
        # the generic (load-where-added) epilogue agrees bit for bit with the FAST forms
        ops.set_tuning("epi_fast", 0)
        try:
            g1, g2 = ops.linear_lo8(x, w, b, res=rh, res_lo8=rl8)
            g3, _ = ops.linear_lo8(x, w, b, res=rh, res_lo8=rl8, want_lo=False)
        finally:
            ops.set_tuning("epi_fast", 3)
        assert torch.equal(g1, oh) and torch.equal(g2, ol8) and torch.equal(g3, oh)


def test_linear_x2_hi_only_fast_form_is_bit_identical():
    """FAST 4 (round 6): residual + its fp16 lo plane, no lo plane out -- what the feed-forward's second linear runs in the split mode -- against the generic epilogue"""
    for (M, K, N) in ((8192, 1280, 320), (4096, 2560, 640), (2048, 5120, 1280)):
        x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
        rh, rl = ops.split_f16(rnd(M, N, seed=4, scale=2.0, dtype=torch.float32))
        a, _ = ops.linear_x2(x, w, b, res=rh, res_lo=rl, want_lo=False)
        ops.set_tuning("epi_fast", 1)
        try:
            c, _ = ops.linear_x2(x, w, b, res=rh, res_lo=rl, want_lo=False)
        finally:
            ops.set_tuning("epi_fast", 3)
        assert torch.equal(a, c), (M, K, N)


@pytest.mark.parametrize("tile", [64, 128])
def test_xattn_block_lo8(tile):
    B, HW, C, Nk = 2, 1024, 320, 77
    M = B * HW
    hh, hl8 = ops.lo8_split(rnd(M, C, seed=1, scale=4.0, dtype=torch.float32))
    h = ops.lo8_value(hh, hl8)
    g, b = (1.0 + 0.1 * rnd(C, seed=2).float()).half(), rnd(C, seed=3, scale=0.1)
    wq, wo, bo = rnd(C, C, seed=4, scale=C ** -0.5), rnd(C, C, seed=5, scale=C ** -0.5), rnd(C, seed=6, scale=0.1)
    kv = rnd(B, Nk, 2 * C, seed=7)
    ops.set_tuning("xattn_tile", tile)
    try:
        oh, ol8, rs = ops.xattn_block_lo8(hh, hl8, g, b, wq, kv, wo, bo, hw=HW, row_stats=True)
        # the fp16-lo kernel on the SAME represented stream (every e5m2 value is an fp16 value): same LayerNorm, same branch, same fp32 sum -- only the output's lo plane differs
        xh, xl = ops.xattn_block_x2(hh, hl8.view(torch.float8_e5m2).to(torch.float16), g, b, wq, kv, wo, bo, hw=HW)
    finally:
        ops.set_tuning("xattn_tile", 64)
    got = ops.lo8_value(oh, ol8)
    assert torch.equal(oh, xh)                                # the hi plane is the fp16 rounding of the same fp32 sum
    assert rel_l2(got, xh.float() + xl.float()) < LO8_TOL     # hi + lo8 against hi + lo: the 14-bit storage of the output
    assert rel_l2(got - h, (xh.float() + xl.float()) - h) < 2e-3
    ulp = oh.float().abs().clamp_min(2.0 ** -14).log2().floor().exp2() * 2.0 ** -10
    assert bool((ol8.view(torch.float8_e5m2).float().abs() <= 0.5 * ulp).all())
    # the row statistics (norm3 folded into the GEGLU GEMM) are those of the fp32 values the planes were rounded from
    want = torch.stack([got.sum(1), (got * got).sum(1)], 1)
    assert rel_l2(rs.reshape(M, 2), want) < 1e-4
    assert rel_l2(ops.row_stats_lo8(oh, ol8).reshape(M, 2), want) < 1e-5


def _small_unet(residual):
    cfg = dict(layers_per_block=1, sample_size=16)
    return get_unet(cfg, seed=3, residual=residual)


def test_unet_x2_mode_is_closer_to_the_fp32_oracle_and_api_round_trips():
    from oracle.unet_oracle import UNetOracle
    u16, sd = _small_unet("f16")
    ux2 = HipUNet2DConditionModel(dict(layers_per_block=1, sample_size=16), device=DEV, residual="residual_fp32")   # a second handle on the same weights, created through the alias
    ux2.load_state_dict(sd)
    assert ux2.residual == "f16x2" and u16.residual == "f16"
    with pytest.raises(ValueError):
        u16.set_residual_precision("fp64")
    lat = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(1)).half()
    ctx = torch.cat([synthetic_prompt_embeds(2, seed=5), synthetic_prompt_embeds(2, seed=6)]).half()
    want = get_oracle(dict(layers_per_block=1, sample_size=16), seed=3)(torch.cat([lat.float()] * 2), 499, ctx.float())
    e16 = rel_l2(u16(lat.to(DEV), 499, encoder_hidden_states=ctx.to(DEV), dup=2, reuse_kv=False)[0].float().cpu(), want)
    ex2 = rel_l2(ux2(lat.to(DEV), 499, encoder_hidden_states=ctx.to(DEV), dup=2, reuse_kv=False)[0].float().cpu(), want)
    print(f"\nsmall UNet eps error vs fp32 oracle: f16 stream {e16:.3e}, f16x2 stream {ex2:.3e}")
    assert ex2 < 0.8 * e16
    # the same object switched back and forth reproduces both results bit for bit (workspace re-sized, K/V cache dropped)
    a = u16(lat.to(DEV), 499, encoder_hidden_states=ctx.to(DEV), dup=2, reuse_kv=False)[0].clone()
    u16.set_residual_precision("f16x2")
    b = u16(lat.to(DEV), 499, encoder_hidden_states=ctx.to(DEV), dup=2, reuse_kv=False)[0].clone()
    u16.set_residual_precision("f16")
    c = u16(lat.to(DEV), 499, encoder_hidden_states=ctx.to(DEV), dup=2, reuse_kv=False)[0].clone()
    assert torch.equal(a, c) and not torch.equal(a, b)
    assert torch.equal(b, ux2(lat.to(DEV), 499, encoder_hidden_states=ctx.to(DEV), dup=2, reuse_kv=False)[0])


def test_unet_lo8_hidden_state_matches_the_fp16_lo_plane():
    """knob lo8: the transformer blocks' hidden state with an 8-bit lo plane (default) against an fp16 one -- 14 vs 22 bits on tensors whose storage error is two
    orders below the branch tensors' fp16 roundings: both forwards sit at the same distance from the oracle"""
    cfg = dict(layers_per_block=1, sample_size=32)
    u, _ = get_unet(cfg, seed=3, residual="f16x2")
    lat = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(1)).half()
    ctx = torch.cat([synthetic_prompt_embeds(2, seed=5), synthetic_prompt_embeds(2, seed=6)]).half()
    want = get_oracle(cfg, seed=3)(torch.cat([lat.float()] * 2), 499, ctx.float())
    run = lambda: u(lat.to(DEV), 499, encoder_hidden_states=ctx.to(DEV), dup=2, reuse_kv=False)[0].float().cpu()
    a = run()
    outs = {}
    for knobs in (dict(lo8=0), dict(lo8=1, cfg_share=0), dict(lo8=1, xattn_fused=0), dict(lo8=1, ln_fold=0), dict(lo8=1, x2_split_a=3)):
        for k, v in knobs.items():
            ops.set_tuning(k, v)
        try:
            outs[tuple(knobs.items())] = run()
        finally:
            ops.reset_tuning()
    b = outs[(("lo8", 0),)]
    ea, eb = rel_l2(a, want), rel_l2(b, want)
    print(f"\nreduced UNet (32 x 32) eps error vs the fp32 oracle: lo8 hidden state {ea:.4e}, fp16 lo plane {eb:.4e}; the two outputs differ by {rel_l2(a, b):.2e}")
    # (the two outputs themselves differ by about one per-forward error: a 1e-5 perturbation of the stream re-draws the fp16 roundings of every branch tensor behind it;
    #  what the byte planes must not do is move the DISTANCE from the oracle)
    assert rel_l2(a, b) < 2e-3 and abs(ea - eb) < 0.05 * eb, (ea, eb)
    assert torch.equal(outs[(("lo8", 1), ("cfg_share", 0))], a)                       # the shared CFG prefix stays bit-identical with byte planes
    assert rel_l2(outs[(("lo8", 1), ("xattn_fused", 0))], a) < 1.5e-3                 # unfused cross-attention: igemm epilogues carry the byte planes
    for k in ((("lo8", 1), ("ln_fold", 0)), (("lo8", 1), ("x2_split_a", 3))):       # consumers that need an fp16 lo plane switch the byte planes off by themselves
        assert rel_l2(outs[k], a) < 1.5e-3 and torch.isfinite(outs[k]).all()


@pytest.mark.parametrize("knobs", [dict(cfg_share=0), dict(xattn_fused=0), dict(gn_fuse=0), dict(cfg_share=0, xattn_fused=0)])
def test_unet_x2_execution_variants_agree(knobs):
    """CFG shared prefix on / off is bit-identical in the split mode too; the unfused cross-attention block and the statistics-pass GroupNorm
    differ from the default by fp16 roundings inside a branch only."""
    u, _ = _small_unet("f16x2")
    u64, _ = get_unet(dict(layers_per_block=1, sample_size=32), seed=3, residual="f16x2")   # 32 x 32: the C = 320 level is fusable (HW % 128 == 0)
    for net, S in ((u, 16), (u64, 32)):
        lat = torch.randn(2, 4, S, S, generator=torch.Generator().manual_seed(1)).half().to(DEV)
        ctx = torch.cat([synthetic_prompt_embeds(2, seed=5), synthetic_prompt_embeds(2, seed=6)]).half().to(DEV)
        base = net(lat, 499, encoder_hidden_states=ctx, dup=2, reuse_kv=False)[0].clone()
        for k, v in knobs.items():
            ops.set_tuning(k, v)
        try:
            alt = net(lat, 499, encoder_hidden_states=ctx, dup=2, reuse_kv=False)[0].clone()
        finally:
            for k in knobs:
                ops.set_tuning(k, 1)
        if set(knobs) == {"cfg_share"}:
            assert torch.equal(base, alt)
        else:
            assert rel_l2(alt.float(), base.float()) < 1.5e-3
