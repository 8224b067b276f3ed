"""Per-kernel parity of the HIP ops (through the C ABI) against plain torch fp32 references of
the same op evaluated on the same fp16-rounded inputs.  Tolerances: outputs are stored in fp16
(eps 9.8e-4) with fp32 accumulation -> relative L2 <= 1e-3, max abs error a few fp16 ulps of the
output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from consolver_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().to(DEV)


def nchw(x):
    return x.permute(0, 3, 1, 2).float()


def nhwc(x):
    return x.permute(0, 2, 3, 1)


CONV_CASES = [
    # B, H, W, c0, c1, N, taps, stride, up, bias, temb, res
    (2, 16, 16, 64, 0, 128, 9, 1, False, True, False, False),
    (1, 8, 8, 320, 0, 320, 9, 1, False, True, True, True),      # BN=160 path, temb per sample
    (3, 8, 8, 128, 0, 160, 9, 1, False, False, False, False),
    (2, 16, 16, 128, 0, 128, 9, 2, False, True, False, False),  # downsample
    (2, 8, 8, 64, 0, 256, 9, 1, True, True, False, False),      # fused nearest x2 upsample
    (2, 8, 8, 128, 64, 320, 1, 1, False, True, False, False),   # 1x1 over a skip concat
    (2, 8, 8, 64, 64, 128, 9, 1, False, True, False, True),     # 3x3 over a concat + residual
    (1, 4, 4, 640, 0, 640, 9, 1, False, True, True, True),      # M = 16 << tile (tail rows)
    (5, 8, 8, 192, 0, 128, 1, 1, False, True, False, True),     # M = 320, ragged m tiles
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case):
    B, H, W, c0, c1, N, taps, stride, up, use_b, use_t, use_r = case
    k = 3 if taps == 9 else 1
    x0 = rnd(B, H, W, c0, seed=1)
    x1 = rnd(B, H, W, c1, seed=2) if c1 else None
    w = rnd(N, c0 + c1, k, k, seed=3, scale=(1.0 / ((c0 + c1) * taps)) ** 0.5)
    bias = rnd(N, seed=4, scale=0.1) if use_b else None
    temb = rnd(B, N + 64, seed=5, scale=0.5) if use_t else None
    Ho = 2 * H if up else (H // 2 if stride == 2 else H)
    res = rnd(B, Ho, Ho * W // H, N, seed=6) if use_r else None
    out = ops.conv2d(x0, ops.pack_conv_weight(w), bias, x1=x1, taps=taps, stride=stride, upsample=up,
                     temb=temb[:, 32:32 + N].contiguous() if use_t else None, res=res)
    xin = nchw(torch.cat([x0, x1], -1) if c1 else x0)
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.float(), bias.float() if use_b else None, stride=stride, padding=k // 2)
    if use_t:
        ref = ref + temb[:, 32:32 + N].float()[:, :, None, None]
    if use_r:
        ref = ref + nchw(res)
    assert out.shape == (B, Ho, ref.shape[3], N)
    assert rel_l2(nchw(out), ref) < 1e-3
    assert float((nchw(out) - ref).abs().max()) < 4e-3 * float(ref.abs().max())


HALO_CASES = [  # B, H, W, cin, N : halo-resident conv3x3 kernel forced on small shapes (covers every tile geometry:
    # 16x16 patches of 16..256-wide images, whole 8-wide images 4 per tile with a ragged last tile, non-square images,
    # 160- and 128-channel tiles)
    (2, 64, 64, 64, 320), (3, 32, 32, 128, 160), (2, 16, 16, 64, 320), (5, 8, 8, 64, 160), (1, 8, 8, 128, 320), (1, 64, 64, 320, 640),
    (1, 128, 128, 64, 128), (1, 256, 256, 64, 256), (2, 64, 64, 128, 512), (3, 8, 8, 64, 128), (1, 16, 16, 64, 384),
    (2, 16, 32, 64, 160), (1, 32, 16, 64, 128), (3, 16, 8, 64, 160), (1, 48, 80, 64, 128)]


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv3x3_halo_kernel(case):
    B, H, W, cin, N = case
    x, w, b = rnd(B, H, W, cin, seed=1), rnd(N, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(N, seed=3, scale=0.1)
    temb, res = rnd(B, N, seed=4, scale=0.5), rnd(B, H, W, N, seed=5)
    ref = F.conv2d(nchw(x), w.float(), b.float(), padding=1) + temb.float()[:, :, None, None] + nchw(res)
    ops.set_tuning("conv_halo", 2); ops.set_tuning("conv_lw", 0)
    try:
        out = ops.conv2d(x, ops.pack_conv_weight(w), b, temb=temb, res=res)
    finally:
        ops.set_tuning("conv_halo", 1); ops.set_tuning("conv_lw", 1)
    generic = None
    ops.set_tuning("conv_halo", 0)
    try:
        generic = ops.conv2d(x, ops.pack_conv_weight(w), b, temb=temb, res=res)
    finally:
        ops.set_tuning("conv_halo", 1)
    assert rel_l2(nchw(out), ref) < 1e-3
    assert float((nchw(out) - ref).abs().max()) < 4e-3 * float(ref.abs().max())
    assert rel_l2(out.float(), generic.float()) < 5e-4      # same math, different k order


@pytest.mark.parametrize("B,H,cin,N", [(2, 32, 64, 320), (1, 16, 128, 160), (3, 8, 64, 320), (1, 64, 64, 256), (2, 128, 64, 128)])
def test_conv3x3_halo_kernel_fused_upsample(B, H, cin, N):
    x, w, b = rnd(B, H, H, cin, seed=1), rnd(N, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(N, seed=3, scale=0.1)
    ref = F.conv2d(F.interpolate(nchw(x), scale_factor=2.0, mode="nearest"), w.float(), b.float(), padding=1)
    ops.set_tuning("conv_halo", 2); ops.set_tuning("conv_lw", 0)
    try:
        out = ops.conv2d(x, ops.pack_conv_weight(w), b, upsample=True)
    finally:
        ops.set_tuning("conv_halo", 1); ops.set_tuning("conv_lw", 1)
    assert out.shape == (B, 2 * H, 2 * H, N)
    assert rel_l2(nchw(out), ref) < 1e-3


WIDE_CASES = [  # B, H, W, cin, N (multiples of 320 / 256): the 256x320 and 256x256 tiles with the k32 inner step, forced (conv_halo = 3) on every
    # tile geometry: 16x16 patches, whole 8-wide images four per tile with a ragged last tile, non-square images, several column tiles
    (2, 64, 64, 64, 320), (1, 32, 32, 128, 640), (2, 16, 16, 64, 1280), (5, 8, 8, 64, 320), (1, 64, 64, 64, 256), (1, 32, 32, 128, 512),
    (3, 8, 8, 128, 256), (2, 16, 32, 64, 320), (1, 48, 80, 64, 256),
    (8, 8, 8, 1280, 320), (2, 16, 16, 512, 256)]        # + split over the channel chunks (fp32 partials)


@pytest.mark.parametrize("case", WIDE_CASES)
def test_conv3x3_halo_kernel_wide_tiles(case):
    B, H, W, cin, N = case
    x, w, b = rnd(B, H, W, cin, seed=1), rnd(N, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(N, seed=3, scale=0.1)
    temb, res = rnd(B, N, seed=4, scale=0.5), rnd(B, H, W, N, seed=5)
    ref = F.conv2d(nchw(x), w.float(), b.float(), padding=1) + temb.float()[:, :, None, None] + nchw(res)
    outs = {}
    for mode in (3, 4):                     # 3: force the wide tiles, 4: forbid them (k64 tiles of 160 / 128 columns)
        ops.set_tuning("conv_halo", mode); ops.set_tuning("conv_lw", 0)
        try:
            outs[mode] = ops.conv2d(x, ops.pack_conv_weight(w), b, temb=temb, res=res)
        finally:
            ops.set_tuning("conv_halo", 1); ops.set_tuning("conv_lw", 1)
    assert rel_l2(nchw(outs[3]), ref) < 1e-3
    assert float((nchw(outs[3]) - ref).abs().max()) < 4e-3 * float(ref.abs().max())
    assert rel_l2(outs[3].float(), outs[4].float()) < 5e-4      # same math, different k order


@pytest.mark.parametrize("B,H,cin,N", [(2, 32, 64, 320), (1, 16, 128, 640), (1, 64, 64, 256)])
def test_conv3x3_halo_kernel_wide_tiles_fused_upsample(B, H, cin, N):
    x, w, b = rnd(B, H, H, cin, seed=1), rnd(N, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(N, seed=3, scale=0.1)
    ref = F.conv2d(F.interpolate(nchw(x), scale_factor=2.0, mode="nearest"), w.float(), b.float(), padding=1)
    ops.set_tuning("conv_halo", 3); ops.set_tuning("conv_lw", 0)
    try:
        out = ops.conv2d(x, ops.pack_conv_weight(w), b, upsample=True)
    finally:
        ops.set_tuning("conv_halo", 1); ops.set_tuning("conv_lw", 1)
    assert rel_l2(nchw(out), ref) < 1e-3


LW_CASES = [  # B, H, W, cin, N (N % 160 == 0, or N % 128 == 0 below): the loader-wave kernel (conv3_lw_kernel) forced on every tile geometry -- 16 x 16 patches, whole
    # 8-wide images four per tile with a ragged last tile, non-square images, one / several column tiles, one / many channel chunks
    (2, 64, 64, 64, 320), (3, 32, 32, 128, 160), (2, 16, 16, 64, 320), (5, 8, 8, 64, 160), (1, 8, 8, 128, 320), (1, 64, 64, 320, 640),
    (2, 16, 32, 64, 160), (3, 16, 8, 64, 160), (1, 48, 80, 64, 480), (1, 32, 32, 640, 640), (2, 16, 16, 1280, 1280), (7, 8, 8, 192, 960),
    # N % 128 == 0 (BN 128: the VAE's widths)
    (2, 64, 64, 64, 128), (1, 32, 32, 256, 256), (1, 64, 64, 128, 512), (5, 8, 8, 64, 256), (1, 48, 80, 64, 384), (2, 16, 16, 512, 512)]


@pytest.mark.parametrize("case", LW_CASES)
def test_conv3x3_loader_wave_kernel(case):
    B, H, W, cin, N = case
    x, w, b = rnd(B, H, W, cin, seed=1), rnd(N, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(N, seed=3, scale=0.1)
    temb, res = rnd(B, N, seed=4, scale=0.5), rnd(B, H, W, N, seed=5)
    ref = F.conv2d(nchw(x), w.float(), b.float(), padding=1) + temb.float()[:, :, None, None] + nchw(res)
    outs = {}
    for lw in (1, 0):
        ops.set_tuning("conv_halo", 2); ops.set_tuning("conv_lw", lw)
        try:
            outs[lw] = ops.conv2d(x, ops.pack_conv_weight(w), b, temb=temb, res=res, splitk=False)     # (no split over the chunks: bit-comparable)
            if lw:
                again = ops.conv2d(x, ops.pack_conv_weight(w), b, temb=temb, res=res, splitk=False)
        finally:
            ops.set_tuning("conv_halo", 1); ops.set_tuning("conv_lw", 1)
    assert rel_l2(nchw(outs[1]), ref) < 1e-3
    assert float((nchw(outs[1]) - ref).abs().max()) < 4e-3 * float(ref.abs().max())
    assert torch.equal(outs[1], outs[0])        # same k order (chunk-major, tap-minor, two k halves) and same epilogue as the 8-wave halo kernel
    assert torch.equal(outs[1], again)


@pytest.mark.parametrize("B,H,cin,N", [(2, 32, 64, 320), (1, 16, 128, 160), (3, 8, 64, 320), (1, 32, 320, 640), (1, 32, 128, 256), (2, 16, 64, 128), (1, 64, 256, 512)])
def test_conv3x3_loader_wave_kernel_fused_upsample(B, H, cin, N):
    x, w, b = rnd(B, H, H, cin, seed=1), rnd(N, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(N, seed=3, scale=0.1)
    ref = F.conv2d(F.interpolate(nchw(x), scale_factor=2.0, mode="nearest"), w.float(), b.float(), padding=1)
    ops.set_tuning("conv_halo", 2)
    try:
        out = ops.conv2d(x, ops.pack_conv_weight(w), b, upsample=True)
    finally:
        ops.set_tuning("conv_halo", 1)
    assert out.shape == (B, 2 * H, 2 * H, N)
    assert rel_l2(nchw(out), ref) < 1e-3
    assert float((nchw(out) - ref).abs().max()) < 4e-3 * float(ref.abs().max())


SUBPIXEL_CASES = [   # B, H (input), cin, N
    (5, 8, 128, 160),      # four whole 8 x 8 input images per tile (the ragged last tile holds one)
    (2, 8, 64, 320),
    (1, 16, 64, 320),      # one 16 x 16 input patch per image
    (3, 16, 192, 160),
    (2, 32, 128, 160),     # four patches per image
    (1, 48, 64, 160),      # nine patches: not a power of two per row
    (1, 32, 128, 256),     # 128-column tiles (the VAE decoder's widths)
    (2, 16, 64, 128),
]


@pytest.mark.parametrize("B,H,cin,N", SUBPIXEL_CASES)
def test_upsample_conv_subpixel_form(B, H, cin, N):
    """cs_op_conv_up_sub (round 6): nearest x2 + 3x3 conv as four 2 x 2-tap phases over the INPUT pixels on pre-summed taps.  (1) small-integer data, where every
    sum is exact in fp16 / fp32: bit-identical to the fused-upsample kernel -- the indexing, the borders, the phase scatter; (2) random data: within the fp16
    rounding of the summed weights of the fp64 reference and of the plain kernel; (3) the GroupNorm statistics of the epilogue equal the sums of what it stored."""
    gi = torch.Generator().manual_seed(11)
    xi = torch.randint(-2, 3, (B, H, H, cin), generator=gi).to(torch.float16).to(DEV)
    wi = torch.randint(-1, 2, (N, cin, 3, 3), generator=gi).to(torch.float16).to(DEV)
    bi = torch.randint(-3, 4, (N,), generator=gi).to(torch.float16).to(DEV)
    wp = ops.pack_conv_weight(wi)
    ws = ops.conv_up_fold_pack(wp).to(DEV)
    assert ws.shape == (4, N, 4 * cin)
    assert float(ws.float().abs().max()) <= 4.0                     # (sums of 1, 2 or 4 taps)
    plain = ops.conv2d(xi, wp, bi, upsample=True)
    sub = ops.conv_up_sub(xi, wp, ws, bi)
    ref = F.conv2d(F.interpolate(nchw(xi), scale_factor=2.0, mode="nearest"), wi.float(), bi.float(), padding=1)
    assert float(ref.abs().max()) < 2048                            # integers below 2^11: exact in fp16
    assert torch.equal(nchw(plain), ref)
    assert torch.equal(sub, plain)
    # random data
    x, w, b = rnd(B, H, H, cin, seed=1), rnd(N, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(N, seed=3, scale=0.1)
    wp = ops.pack_conv_weight(w)
    ws = ops.conv_up_fold_pack(wp).to(DEV)
    ref = F.conv2d(F.interpolate(nchw(x).double(), scale_factor=2.0, mode="nearest"), w.double(), b.double(), padding=1).float()
    plain = ops.conv2d(x, wp, b, upsample=True)
    sub, st = ops.conv_up_sub(x, wp, ws, b, gn_stats=True)
    again = ops.conv_up_sub(x, wp, ws, b)
    assert torch.equal(sub, again)
    e_plain, e_sub = rel_l2(nchw(plain), ref), rel_l2(nchw(sub), ref)
    print(f"\nB={B} {H}x{H} {cin}->{N}: plain {e_plain:.3e}, sub-pixel {e_sub:.3e} vs fp64")
    assert e_plain < 4e-4 and e_sub < 6e-4, (e_plain, e_sub)        # one fp16 rounding of the output; + one of the summed weights
    assert float((nchw(sub) - ref).abs().max()) < 4e-3 * float(ref.abs().max())
    Ho = 2 * H
    o = sub.float().reshape(B, Ho * Ho, N // 2, 2)
    want_sum, want_sq = o.sum(dim=(1, 3)).double(), (o * o).sum(dim=(1, 3)).double()
    got = st.double().sum(dim=1)
    scale = want_sq.sqrt().clamp_min(1.0)
    assert float(((got[..., 0] - want_sum).abs() / (scale * (Ho * Ho) ** 0.5)).max()) < 2e-5
    assert float(((got[..., 1] - want_sq).abs() / want_sq.clamp_min(1e-6)).max()) < 2e-5
    assert float(st.abs().reshape(B, -1, N).sum(-1).min()) > 0       # every 64-pixel block of every sample was written by exactly one wave
    g, bt = (1 + 0.2 * rnd(N, seed=7)), 0.1 * rnd(N, seed=8)
    flat = sub.reshape(B, Ho * Ho, N)
    a = ops.group_norm(flat, g, bt, 32 if (N // 32) % 2 == 0 else 16, 1e-5, True)
    c = ops.group_norm(flat, g, bt, 32 if (N // 32) % 2 == 0 else 16, 1e-5, True, stats0=st)
    assert rel_l2(c.float(), a.float()) < 2e-4


@pytest.mark.parametrize("B,H,cin,N", [(8, 8, 1280, 320), (2, 16, 512, 320), (3, 8, 256, 160), (2, 16, 512, 256), (4, 8, 256, 128)])
def test_conv3x3_loader_wave_kernel_split_k(B, H, cin, N):
    x, w, b = rnd(B, H, H, cin, seed=1), rnd(N, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(N, seed=3, scale=0.1)
    temb, res = rnd(B, N, seed=4, scale=0.5), rnd(B, H, H, N, seed=5)
    ref = F.conv2d(nchw(x), w.float(), b.float(), padding=1) + temb.float()[:, :, None, None] + nchw(res)
    ops.set_tuning("conv_halo", 2)
    try:
        out = ops.conv2d(x, ops.pack_conv_weight(w), b, temb=temb, res=res)
        out2 = ops.conv2d(x, ops.pack_conv_weight(w), b, temb=temb, res=res.clone())
    finally:
        ops.set_tuning("conv_halo", 1)
    assert rel_l2(nchw(out), ref) < 1e-3
    assert torch.equal(out, out2)


@pytest.mark.parametrize("B,H,cin,N", [(8, 8, 1280, 320), (2, 16, 512, 320), (3, 8, 256, 160)])
def test_conv3x3_halo_kernel_split_k(B, H, cin, N):
    """small images: channel chunks split over workgroups, fp32 partials + fused reduce epilogue"""
    x, w, b = rnd(B, H, H, cin, seed=1), rnd(N, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(N, seed=3, scale=0.1)
    temb, res = rnd(B, N, seed=4, scale=0.5), rnd(B, H, H, N, seed=5)
    ref = F.conv2d(nchw(x), w.float(), b.float(), padding=1) + temb.float()[:, :, None, None] + nchw(res)
    ops.set_tuning("conv_halo", 2); ops.set_tuning("conv_lw", 0)
    try:
        out = ops.conv2d(x, ops.pack_conv_weight(w), b, temb=temb, res=res)
        res2 = res.clone()
        out2 = ops.conv2d(x, ops.pack_conv_weight(w), b, temb=temb, res=res2)   # deterministic (no atomics)
    finally:
        ops.set_tuning("conv_halo", 1); ops.set_tuning("conv_lw", 1)
    assert rel_l2(nchw(out), ref) < 1e-3
    assert torch.equal(out, out2)


@pytest.mark.parametrize("B,H,W,cin,cout,post", [(2, 64, 64, 320, 4, False),      # the UNet's eps head (16 patches per sample, image borders on every patch side)
                                                  (3, 16, 16, 64, 4, False),       # one patch = the whole image: every halo border is padding
                                                  (1, 32, 48, 128, 3, True),       # the VAE's image head (non-square, postprocess)
                                                  (2, 16, 32, 128, 8, False), (1, 32, 16, 64, 16, False),       # the encoder's moments head widths
                                                  (2, 8, 8, 320, 4, False), (1, 24, 24, 40, 3, False)])          # shapes off the patch kernels (one wave per pixel)
def test_conv_out_small_cout_kernels(B, H, W, cin, cout, post):
    """cs_op_conv_out: NHWC -> NCHW 3x3 conv with 3 / 4 / 8 / 16 output channels against torch conv2d on the same fp16-rounded inputs; on the 16 x 16-patch
    shapes the matrix-core kernel (conv_out_mfma_kernel, default) and the v_dot2 patch kernel must agree to accumulation-order precision."""
    x = rnd(B, H, W, cin, seed=3)
    w = rnd(cout, cin, 3, 3, seed=4, scale=(1.0 / (9 * cin)) ** 0.5)
    b = rnd(cout, seed=5, scale=0.1)
    ref = F.conv2d(nchw(x), w.float(), b.float(), padding=1)
    if post:
        ref = (ref / 2 + 0.5).clamp(0, 1)
    wp = ops.pack_conv_weight(w)
    outs = {}
    try:
        for mode in (1, 0):
            ops.set_tuning("conv_out_mfma", mode)
            outs[mode] = ops.conv_out(x, wp, b, postprocess=post)
            assert outs[mode].shape == (B, cout, H, W)
            assert rel_l2(outs[mode].float(), ref) < 1e-3, (mode, rel_l2(outs[mode].float(), ref))
            assert float((outs[mode].float() - ref).abs().max()) < 4e-3
            assert torch.equal(outs[mode], ops.conv_out(x, wp, b, postprocess=post))          # deterministic
    finally:
        ops.set_tuning("conv_out_mfma", 1)
    assert float((outs[1].float() - outs[0].float()).abs().max()) < 2e-3


def test_conv2d_temb_broadcast_and_inplace_residual():
    x = rnd(2, 8, 8, 64, seed=1)
    w = rnd(128, 64, 3, 3, seed=2, scale=0.05)
    temb = rnd(1, 128, seed=3)
    res = rnd(2, 8, 8, 128, seed=4)
    want = F.conv2d(nchw(x), w.float(), padding=1) + temb.float()[0][None, :, None, None] + nchw(res)
    out = ops.conv2d(x, ops.pack_conv_weight(w), None, temb=temb, res=res)
    assert rel_l2(nchw(out), want) < 1e-3


@pytest.mark.parametrize("M,K,N", [(256, 64, 128), (100, 320, 960), (4096, 1280, 320), (77 * 3, 768, 640), (1, 64, 160)])
def test_linear(M, K, N):
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    out = ops.linear(x, w, b, res=r)
    ref = F.linear(x.float(), w.float(), b.float()) + r.float()
    assert rel_l2(out.float(), ref) < 1e-3
    # in-place residual (out aliases res), as the UNet executor uses it
    r2 = r.clone()
    ops.linear(x, w, b, res=r2, out=r2)
    assert torch.equal(r2, out)


@pytest.mark.parametrize("M,K,N,geglu", [(256, 64, 320, False), (1000, 320, 960, False), (300, 128, 640, True), (513, 64, 2560, True)])
def test_linear_big_tile_kernel(M, K, N, geglu):
    """256x320 tile kernel forced on small shapes (ragged M tiles, GEGLU pairs, residual)"""
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.2)
    ops.set_tuning("gemm_big", 2)
    try:
        if geglu:
            wp, bp = ops.geglu_pack(w, b)
            out = ops.linear(x, wp.to(DEV), bp.to(DEV), geglu=True)
            val, gate = F.linear(x.float(), w.float(), b.float()).chunk(2, dim=-1)
            ref = val * F.gelu(gate)
        else:
            r = rnd(M, N, seed=4)
            out = ops.linear(x, w, b, res=r)
            ref = F.linear(x.float(), w.float(), b.float()) + r.float()
    finally:
        ops.set_tuning("gemm_big", 1)
    assert rel_l2(out.float(), ref) < 1.5e-3


@pytest.mark.parametrize("M,K,N,geglu", [(256, 64, 320, False), (1000, 320, 960, False), (300, 128, 640, True), (513, 64, 2560, True), (8192, 1280, 1280, False),
                                         (4097, 640, 1920, False), (2048, 2560, 640, False), (1, 192, 320, False), (8192, 1280, 2560, True)])
def test_linear_hand_scheduled_256x320_kernel(M, K, N, geglu):
    """gemm_w8_kernel (the 8-wave 256 x 320 tile with buffer-load LDS-DMA, hand-counted LDS reads, in-place MFMAs) against the reference and, bit for bit,
    against gemm_big_kernel (same k order, same epilogue): one / odd / even k-step counts, ragged M (clamped rows), GEGLU, residual."""
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.2)
    outs = {}
    for w8 in (1, 0):
        ops.set_tuning("gemm_w8", w8); ops.set_tuning("gemm_big", 2)
        try:
            if geglu:
                wp, bp = ops.geglu_pack(w, b)
                outs[w8] = ops.linear(x, wp.to(DEV), bp.to(DEV), geglu=True)
                val, gate = F.linear(x.float(), w.float(), b.float()).chunk(2, dim=-1)
                ref = val * F.gelu(gate)
            else:
                r = rnd(M, N, seed=4)
                outs[w8] = ops.linear(x, w, b, res=r)
                ref = F.linear(x.float(), w.float(), b.float()) + r.float()
        finally:
            ops.set_tuning("gemm_w8", 1); ops.set_tuning("gemm_big", 1)
    assert rel_l2(outs[1].float(), ref) < 1.5e-3
    assert torch.equal(outs[1], outs[0])


def test_conv1x1_over_concat_hand_scheduled_kernel():
    B, H, c0, c1, N = 2, 16, 640, 320, 640
    x0, x1 = rnd(B, H, H, c0, seed=1), rnd(B, H, H, c1, seed=2)
    w, b = rnd(N, c0 + c1, 1, 1, seed=3, scale=(c0 + c1) ** -0.5), rnd(N, seed=4, scale=0.1)
    ref = F.conv2d(nchw(torch.cat([x0, x1], -1)), w.float(), b.float())
    for w8 in (1, 0):
        ops.set_tuning("gemm_big", 2); ops.set_tuning("gemm_w8", w8)
        try:
            out = ops.conv2d(x0, ops.pack_conv_weight(w), b, x1=x1, taps=1)
        finally:
            ops.set_tuning("gemm_big", 1); ops.set_tuning("gemm_w8", 1)
        assert rel_l2(nchw(out), ref) < 1e-3


@pytest.mark.parametrize("M,K,N", [(256, 64, 160), (1000, 320, 480), (8192, 1280, 1280), (513, 128, 1280), (1, 192, 160), (4097, 640, 640), (700, 1024, 480)])
def test_linear_big_tile_kernel_160(M, K, N):
    """256x160 variant of the 8-wave kernel (the 1280-wide layers at 16x16), forced on small and ragged shapes"""
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.2), rnd(M, N, seed=4)
    ref = F.linear(x.float(), w.float(), b.float()) + r.float()
    outs = {}
    for lw in (1, 0):                          # 1: gemm_lw_kernel (waves 0-3 multiply, 4-7 stage, three stage buffers), 0: the 8-wave 256 x 160 kernel
        ops.set_tuning("gemm_big", 3); ops.set_tuning("gemm_lw", lw)
        try:
            outs[lw] = ops.linear(x, w, b, res=r)
        finally:
            ops.set_tuning("gemm_big", 1); ops.set_tuning("gemm_lw", 1)
        assert rel_l2(outs[lw].float(), ref) < 1.5e-3
    assert torch.equal(outs[1], outs[0])        # same k order, same epilogue


@pytest.mark.parametrize("M,C", [(128, 64), (300, 320), (64, 1280)])
def test_linear_geglu(M, C):
    x = rnd(M, C, seed=1)
    w, b = rnd(8 * C, C, seed=2, scale=C ** -0.5), rnd(8 * C, seed=3, scale=0.2)
    wp, bp = ops.geglu_pack(w, b)
    out = ops.linear(x, wp.to(DEV), bp.to(DEV), geglu=True)
    pr = F.linear(x.float(), w.float(), b.float())
    val, gate = pr.chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    assert out.shape == (M, 4 * C)
    assert rel_l2(out.float(), ref) < 1.5e-3


@pytest.mark.parametrize("B,H,Nq,Nk,dh", [(2, 8, 256, 256, 40), (1, 8, 1024, 1024, 40), (2, 8, 128, 77, 40), (2, 8, 256, 256, 80),
                                           (1, 8, 64, 77, 80), (2, 8, 64, 64, 160), (1, 8, 256, 77, 160), (1, 2, 100, 130, 40),
                                           (1, 3, 16, 5, 80)])
def test_attention(B, H, Nq, Nk, dh):
    C = H * dh
    q, k, v = rnd(B, Nq, C, seed=1), rnd(B, Nk, C, seed=2), rnd(B, Nk, C, seed=3)
    out = ops.attention(q, k, v, H)
    qf, kf, vf = (t.float().view(B, -1, H, dh).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * dh ** -0.5, -1) @ vf).transpose(1, 2).reshape(B, Nq, C)
    assert rel_l2(out.float(), ref) < 2e-3
    assert float((out.float() - ref).abs().max()) < 1e-2


def test_attention_fused_qkv_strides_and_peaky_softmax():
    B, H, N, dh = 2, 8, 192, 40
    C = H * dh
    qkv = rnd(B, N, 3 * C, seed=7, scale=3.0)      # large logits: exercises the running-max rescale
    out = ops.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], H, q_stride=None) if False else None
    from consolver_amd import _lib as L
    out = torch.empty(B, N, C, dtype=torch.float16, device=DEV)
    L.check(L.lib().cs_op_attention(qkv.data_ptr(), 3 * C, qkv.data_ptr() + 2 * C, 3 * C, qkv.data_ptr() + 4 * C, 3 * C,
                                    out.data_ptr(), C, B, H, N, N, dh, dh ** -0.5, L.stream_ptr(qkv.device)))
    q, k, v = (t.float().reshape(B, N, H, dh).transpose(1, 2) for t in qkv.split(C, dim=-1))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, -1) @ v).transpose(1, 2).reshape(B, N, C)
    assert rel_l2(out.float(), ref) < 2e-3


@pytest.mark.parametrize("B,HW,c0,c1,silu,eps", [(2, 256, 320, 0, True, 1e-5), (3, 64, 640, 320, True, 1e-5), (1, 1024, 320, 0, False, 1e-6),
                                                  (2, 64, 1280, 1280, True, 1e-5), (2, 16, 1280, 640, True, 1e-5), (1, 4, 640, 0, True, 1e-5)])
def test_group_norm(B, HW, c0, c1, silu, eps):
    x0 = rnd(B, HW, c0, seed=1) * 2 + 0.5
    x1 = (rnd(B, HW, c1, seed=2) - 0.3) if c1 else None
    C = c0 + c1
    g, b = (1 + 0.2 * rnd(C, seed=3)), 0.1 * rnd(C, seed=4)
    out = ops.group_norm(x0, g, b, 32, eps, silu, x1=x1)
    xin = torch.cat([x0, x1], -1) if c1 else x0
    ref = F.group_norm(xin.float().transpose(1, 2), 32, g.float(), b.float(), eps)
    ref = (F.silu(ref) if silu else ref).transpose(1, 2)
    assert rel_l2(out.float(), ref) < 1e-3
    assert float((out.float() - ref).abs().max()) < 6e-3


GN_FUSE_CASES = [
    # B, H, cin, c1 (second input source), N, taps, stride, up, temb, res   -- which kernel the shape selects
    (2, 64, 64, 0, 320, 9, 1, False, True, False),     # halo conv, 256 x 320 k32 tiles (resnet conv1: + temb)
    (2, 32, 128, 0, 640, 9, 1, False, False, True),    # halo conv wide tiles, + residual (resnet conv2)
    (3, 16, 128, 0, 160, 9, 1, False, True, True),     # halo conv 256 x 160
    (4, 8, 320, 0, 320, 9, 1, False, False, True),     # 8 x 8 images: four per tile, one per wave
    (2, 16, 64, 0, 320, 9, 1, True, False, False),     # fused x2 upsample (32 x 32 out)
    (2, 32, 128, 0, 128, 9, 2, False, False, False),   # stride-2 downsample: generic kernel
    (8, 8, 1280, 0, 320, 9, 1, False, False, True),    # split-K form at 8 x 8: the statistics come from splitk_reduce_stats_kernel (round 5)
    (32, 8, 1280, 0, 1280, 9, 1, False, True, False),  # ... resnet conv1 at the UNet's 8 x 8 level (+ temb)
    (8, 16, 640, 0, 1280, 9, 2, False, False, False),  # ... stride-2 downsample into 8 x 8 (generic kernel + split-K)
    (6, 8, 1280, 0, 1280, 1, 1, False, False, True),   # ... 1x1 + residual at 8 x 8
    (32, 64, 320, 0, 320, 1, 1, False, False, True),   # 1x1 (proj_out + residual): 256 x 320 GEMM
    (2, 16, 640, 0, 1280, 1, 1, False, False, True),   # 1x1, few tiles: generic / 256 x 160 GEMM
    (2, 8, 64, 64, 128, 9, 1, False, True, True),      # two-source input (generic kernel)
]


@pytest.mark.parametrize("case", GN_FUSE_CASES)
def test_conv_output_group_norm_statistics(case):
    """cs_op_conv2d_gn: the partial sums the epilogue (or the fallback pass) leaves equal the sums of the fp16 output it stored, and
    cs_op_group_norm_pre on them equals cs_op_group_norm on the tensor."""
    B, H, cin, c1, N, taps, stride, up, use_t, use_r = case
    k = 3 if taps == 9 else 1
    x0 = rnd(B, H, H, cin, seed=1)
    x1 = rnd(B, H, H, c1, seed=2) if c1 else None
    w = rnd(N, cin + c1, k, k, seed=3, scale=(1.0 / ((cin + c1) * taps)) ** 0.5)
    bias = rnd(N, seed=4, scale=0.1)
    temb = rnd(B, N, seed=5, scale=0.5) if use_t else None
    Ho = 2 * H if up else (H // 2 if stride == 2 else H)
    res = rnd(B, Ho, Ho, N, seed=6) if use_r else None
    wp = ops.pack_conv_weight(w)
    plain = ops.conv2d(x0, wp, bias, x1=x1, taps=taps, stride=stride, upsample=up, temb=temb, res=res)
    out, st = ops.conv2d(x0, wp, bias, x1=x1, taps=taps, stride=stride, upsample=up, temb=temb, res=res, gn_stats=True)
    assert torch.equal(out, plain)                                   # the statistics do not change what is stored
    assert st.shape == (B, Ho * Ho // 64, N // 2, 2)
    o = out.float().reshape(B, Ho * Ho, N // 2, 2)
    want_sum, want_sq = o.sum(dim=(1, 3)).double(), (o * o).sum(dim=(1, 3)).double()
    got = st.double().sum(dim=1)                                     # over the 64-pixel blocks of a sample
    scale = (o * o).sum(dim=(1, 3)).double().sqrt().clamp_min(1.0)
    assert float(((got[..., 0] - want_sum).abs() / (scale * (Ho * Ho) ** 0.5)).max()) < 2e-5
    assert float(((got[..., 1] - want_sq).abs() / want_sq.clamp_min(1e-6)).max()) < 2e-5
    g, b = (1 + 0.2 * rnd(N, seed=7)), 0.1 * rnd(N, seed=8)
    flat = out.reshape(B, Ho * Ho, N)
    G = 32 if (N // 32) % 2 == 0 else 16                             # the statistics are kept per channel PAIR: even channels per group
    a = ops.group_norm(flat, g, b, G, 1e-5, True)
    c = ops.group_norm(flat, g, b, G, 1e-5, True, stats0=st)
    assert rel_l2(c.float(), a.float()) < 2e-4
    ref = F.silu(F.group_norm(flat.float().transpose(1, 2), G, g.float(), b.float(), 1e-5)).transpose(1, 2)
    assert rel_l2(c.float(), ref) < 1e-3


def test_group_norm_two_sources_with_one_precomputed():
    """skip-concat GroupNorm (960 = 640 + 320 channels: groups straddle the sources) with the statistics of one / both sources supplied."""
    B, H = 2, 16
    a_in, b_in = rnd(B, H, H, 128, seed=1), rnd(B, H, H, 64, seed=2)
    wa = rnd(640, 128, 3, 3, seed=3, scale=(1.0 / (128 * 9)) ** 0.5); wb = rnd(320, 64, 3, 3, seed=4, scale=(1.0 / (64 * 9)) ** 0.5)
    xa, sa = ops.conv2d(a_in, ops.pack_conv_weight(wa), None, gn_stats=True)
    xb, sb = ops.conv2d(b_in, ops.pack_conv_weight(wb), None, gn_stats=True)
    xa, xb = xa.reshape(B, H * H, 640), xb.reshape(B, H * H, 320)
    g, b = (1 + 0.2 * rnd(960, seed=7)), 0.1 * rnd(960, seed=8)
    ref = F.silu(F.group_norm(torch.cat([xa, xb], -1).float().transpose(1, 2), 32, g.float(), b.float(), 1e-5)).transpose(1, 2)
    for s0, s1 in ((sa, sb), (sa, None), (None, sb)):
        out = ops.group_norm(xa, g, b, 32, 1e-5, True, x1=xb, stats0=s0, stats1=s1)
        assert rel_l2(out.float(), ref) < 1e-3


@pytest.mark.parametrize("M,C", [(7, 320), (256, 640), (1000, 1280)])
def test_layer_norm(M, C):
    x = rnd(M, C, seed=1) * 3 + 1
    g, b = 1 + 0.2 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    out = ops.layer_norm(x, g, b)
    ref = F.layer_norm(x.float(), (C,), g.float(), b.float())
    assert rel_l2(out.float(), ref) < 1e-3


def test_ops_are_deterministic():
    x, w = rnd(512, 320, seed=1), rnd(640, 320, seed=2, scale=0.05)
    assert torch.equal(ops.linear(x, w), ops.linear(x, w))
    q = rnd(1, 512, 320, seed=3)
    assert torch.equal(ops.attention(q, q, q, 8), ops.attention(q, q, q, 8))


@pytest.mark.parametrize("B,Nq,Nk", [(2, 256, 64), (2, 256, 128), (5, 512, 576), (1, 512, 1024), (1, 4096, 4096), (3, 256, 768), (40, 1024, 1024)])
def test_attention_dh40_loader_wave_kernel_matches_attn_kernel(B, Nq, Nk):
    """attn40_lw_kernel (loader waves, software-pipelined hand-placed stream), through fused-QKV strides like the UNet's.  With 16x16x32 MFMAs for both k steps
    (attn_lw = 2) it runs the same MFMA chains, exponent arguments, roundings and accumulation order as attn_kernel<f16, 40, 4>: bit-identical.  The default
    (attn_lw = 1) multiplies head dims 32..39 with a 16x16x16 MFMA: the same products summed in another association -- deterministic, and as close to the fp32
    reference as attn_kernel is"""
    from consolver_amd import _lib as L
    H, dh = 8, 40
    C = H * dh
    qa, kva = rnd(B, Nq, 3 * C, seed=11, scale=1.5), rnd(B, Nk, 3 * C, seed=12, scale=1.5)
    outs = {}
    for lw in (1, 2, 0, 1):
        ops.set_tuning("attn_lw", lw)
        try:
            out = torch.empty(B, Nq, C, dtype=torch.float16, device=DEV)
            L.check(L.lib().cs_op_attention(qa.data_ptr(), 3 * C, kva.data_ptr() + 2 * C, 3 * C, kva.data_ptr() + 4 * C, 3 * C,
                                            out.data_ptr(), C, B, H, Nq, Nk, dh, dh ** -0.5, L.stream_ptr(qa.device)))
            if lw == 1 and 1 in outs: again = out
            else: outs[lw] = out
        finally:
            ops.set_tuning("attn_lw", 1)
    q = qa[:, :, :C].float().reshape(B, Nq, H, dh).transpose(1, 2)
    k = kva[:, :, C:2 * C].float().reshape(B, Nk, H, dh).transpose(1, 2)
    v = kva[:, :, 2 * C:].float().reshape(B, Nk, H, dh).transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, -1) @ v).transpose(1, 2).reshape(B, Nq, C)
    assert torch.equal(outs[2], outs[0])
    assert torch.equal(outs[1], again)
    e1, e0 = rel_l2(outs[1].float(), ref), rel_l2(outs[0].float(), ref)
    assert e1 < 2e-3 and e1 < 1.1 * e0 + 1e-5, (e1, e0)
    assert float((outs[1].float() - outs[0].float()).abs().max()) < 2e-3 * float(ref.abs().max()) + 1e-3


@pytest.mark.parametrize("Nq,Nk,pos", [(256, 256, 200), (128, 77, 70), (64, 4096, 3000), (256, 4096, 3000), (512, 512, 300)])
def test_attention_dh40_outlier_key_takes_the_safe_path(Nq, Nk, pos):
    """head dim 40 exponentiates later tiles against the first tile's maximum; a key whose score exceeds it by more than the fp16
    range of P (2^16) must trigger the maxima-tracking redo, not produce inf/NaN"""
    B, H, dh = 2, 8, 40
    q, k, v = rnd(B, Nq, H * dh, seed=1), rnd(B, Nk, H * dh, seed=2), rnd(B, Nk, H * dh, seed=3)
    k = k.clone()
    k[:, pos] = 6.0 * q[:, 5]                                  # score ~ 36 |q|^2 dh^-0.5 log2e >> 16 above the first tile for query 5
    out = ops.attention(q, k, v, H)
    qf, kf, vf = (t.float().view(B, -1, H, dh).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * dh ** -0.5, -1) @ vf).transpose(1, 2).reshape(B, Nq, H * dh)
    assert torch.isfinite(out).all()
    assert rel_l2(out.float(), ref) < 2e-3
    assert float((out.float() - ref).abs().max()) < 1e-2


@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("B,HW,Nk", [(3, 256, 77), (1, 128, 80), (2, 384, 13), (2, 4096, 77), (5, 64, 1), (1, 192, 77)])
def test_xattn_block_fused_kernel(B, HW, Nk, tile):
    """the fused cross-attention sub-block (LayerNorm2 -> to_q -> <= 80-key attention -> to_out + residual in one kernel, C = 320,
    8 heads) against plain torch fp32 of the same fp16 inputs, and against the four unfused HIP ops it replaces.  tile: 64 = xattn64_kernel (round 5
    default: 64-row tiles, two workgroups per CU), 128 = xattn_block_kernel; the two must agree bit for bit (same per-wave arithmetic)."""
    if tile == 128 and HW % 128:
        pytest.skip("the 128-row kernel needs HW % 128 == 0")
    ops.set_tuning("xattn_tile", tile)
    g = torch.Generator().manual_seed(B * 1000 + HW + Nk)
    C, H = 320, 8
    M = B * HW
    h = torch.randn(M, C, generator=g).half().to(DEV)
    gam = (1 + 0.1 * torch.randn(C, generator=g)).half().to(DEV); bet = (0.05 * torch.randn(C, generator=g)).half().to(DEV)
    wq = (torch.randn(C, C, generator=g) * C ** -0.5).half().to(DEV)
    wo = (torch.randn(C, C, generator=g) * C ** -0.5).half().to(DEV); bo = (0.05 * torch.randn(C, generator=g)).half().to(DEV)
    kv = torch.randn(B, Nk, 2 * C, generator=g).half().to(DEV)
    got = ops.xattn_block(h, gam, bet, wq, kv, wo, bo, heads=H, hw=HW)
    # fp32 reference
    hf = h.float()
    ln = torch.nn.functional.layer_norm(hf, (C,), gam.float(), bet.float(), 1e-5)
    q = (ln @ wq.float().T).view(B, HW, H, C // H).transpose(1, 2)
    k = kv[..., :C].float().view(B, Nk, H, C // H).transpose(1, 2)
    v = kv[..., C:].float().view(B, Nk, H, C // H).transpose(1, 2)
    a = (torch.softmax(q @ k.transpose(-1, -2) * (C // H) ** -0.5, -1) @ v).transpose(1, 2).reshape(M, C)
    want = hf + a @ wo.float().T + bo.float()
    err = float((got.float() - want).norm() / want.norm())
    assert torch.isfinite(got).all() and err < 1.5e-3, err
    # the unfused HIP ops (same rounding points up to the softmax formulation)
    lnh = ops.layer_norm(h, gam, bet)
    qh = ops.linear(lnh, wq)
    ah = ops.attention(qh.view(B, HW, C), kv[..., :C], kv[..., C:], H)
    unf = ops.linear(ah.view(M, C), wo, bo, res=h)
    err2 = float((got.float() - unf.float()).norm() / unf.float().norm())
    assert err2 < 1.5e-3, err2
    # in place on the residual stream
    h2 = h.clone()
    ops.xattn_block(h2, gam, bet, wq, kv, wo, bo, heads=H, hw=HW, out=h2)
    assert torch.equal(h2, got)
    if tile == 64 and HW % 128 == 0:
        ops.set_tuning("xattn_tile", 128)
        assert torch.equal(ops.xattn_block(h, gam, bet, wq, kv, wo, bo, heads=H, hw=HW), got)
        # ... and on the split-fp16 stream, with the row statistics
        hl = (0.001 * torch.randn(M, C, generator=g)).half().to(DEV)
        a128 = ops.xattn_block_x2(h, hl, gam, bet, wq, kv, wo, bo, hw=HW, row_stats=True)
        ops.set_tuning("xattn_tile", 64)
        a64 = ops.xattn_block_x2(h, hl, gam, bet, wq, kv, wo, bo, hw=HW, row_stats=True)
        assert all(torch.equal(x, y) for x, y in zip(a128, a64))


XCD_GRID_CASES = [   # weight-heavy SD1.5 layers (16 x 16 / 8 x 8 levels, batch 32) where launch_igemm_impl maps the XCDs as a 2-D grid (tile_of, pn > 0)
    ("conv", 32, 8, 1280, 1280), ("conv", 32, 8, 2560, 1280), ("conv", 32, 16, 1280, 1280), ("conv", 32, 16, 2560, 1280),
    ("linear", 8192, 1280, 3840, False), ("linear", 8192, 1280, 10240, True), ("linear", 8192, 5120, 1280, False), ("linear", 2048, 1280, 10240, True),
    ("linear", 2048, 1280, 1280, False),
]


@pytest.mark.parametrize("case", XCD_GRID_CASES)
def test_xcd_grid_tile_order_is_a_permutation(case):
    """the 2-D XCD grid only changes WHICH workgroup computes a tile: results are bit-identical to the contiguous order, and every tile is computed
    (a wrong bijection leaves tiles of the poisoned output untouched)"""
    if case[0] == "conv":
        _, B, H, cin, N = case
        x, w, b = rnd(B, H, H, cin, seed=1), ops.pack_conv_weight(rnd(N, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)), rnd(N, seed=3, scale=0.1)
        run = lambda: ops.conv2d(x, w, b)
    else:
        _, M, K, N, geglu = case
        x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
        if geglu:
            wp, bp = ops.geglu_pack(w, b)
            w, b = wp.to(DEV), bp.to(DEV)
        run = lambda: ops.linear(x, w, b, geglu=geglu)
    grid = run()
    ops.set_tuning("xcd_grid", 0)
    try:
        contiguous = run()
    finally:
        ops.set_tuning("xcd_grid", 1)
    assert torch.isfinite(grid.float()).all() and torch.equal(grid, contiguous)


EPI_FAST_CASES = [   # (kind, rows / image side, K / Cin, N, what the epilogue adds)
    ("linear", 131072, 320, 320, "res"), ("linear", 131072, 320, 320, "res_x2"), ("linear", 32768, 2560, 640, "res_x2"), ("linear", 8192, 1280, 1280, "res"),
    ("linear", 8192 - 40, 1280, 1280, "res_x2"),                       # ragged M: the last row tile takes the generic code
    ("conv", 64, 320, 320, "temb"), ("conv", 64, 320, 320, "res_x2"), ("conv", 16, 1280, 1280, "res"), ("conv", 8, 1280, 1280, "temb"),
]


@pytest.mark.parametrize("case", EPI_FAST_CASES)
def test_fp32_patch_epilogue_fast_forms_are_bit_identical(case):
    """knob epi_fast: the FAST forms of the fp32-patch epilogue (residual / residual + lo plane / time-embedding loads issued ahead of their use) perform the
    same additions in the same order as the generic code: every output plane and the row statistics are bit-identical"""
    kind, M, K, N, adds = case
    if kind == "linear":
        x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
        rh, rl = ops.split_f16(torch.randn(M, N, device=DEV, generator=torch.Generator(DEV).manual_seed(4)) * 2)
        if adds == "res":
            run = lambda: ops.linear_x2(x, w, b, res=rh, want_lo=False, row_stats=True)
        else:
            run = lambda: ops.linear_x2(x, w, b, res=rh, res_lo=rl, want_lo=True, row_stats=True)
    else:
        B, H = 8, M
        x, w, b = rnd(B, H, H, K, seed=1), ops.pack_conv_weight(rnd(N, K, 3, 3, seed=2, scale=(9 * K) ** -0.5)), rnd(N, seed=3, scale=0.1)
        rh, rl = ops.split_f16(torch.randn(B, H, H, N, device=DEV, generator=torch.Generator(DEV).manual_seed(4)) * 2)
        temb = rnd(B, N, seed=5)
        if adds == "temb":
            run = lambda: ops.conv2d(x, w, b, temb=temb, gn_stats=True)
        elif adds == "res":
            run = lambda: ops.conv2d(x, w, b, res=rh)
        else:
            run = lambda: ops.conv2d_x2(x, w, b, res=rh, res_lo=rl)

    def flat(o, acc):
        if isinstance(o, (tuple, list)):
            for q in o:
                flat(q, acc)
        elif torch.is_tensor(o):
            acc.append(o.clone())
        return acc
    fast = flat(run(), [])
    ops.set_tuning("epi_fast", 0)
    try:
        generic = flat(run(), [])
    finally:
        ops.set_tuning("epi_fast", 3)
    assert len(fast) == len(generic) and len(fast) >= 1
    for a, g in zip(fast, generic):
        assert torch.isfinite(a.float()).all() and torch.equal(a, g)


def test_conv3x3_wider_than_the_loader_wave_zero_region_falls_back():
    """conv3_lw_kernel's padded halo rows read g_zero_region + chunk offset, which covers 64 chunks; Cin = 4160 (65 chunks) must take the halo
    kernels and still pad with zeros"""
    B, H, cin, N = 1, 16, 4160, 160
    x, w, b = rnd(B, H, H, cin, seed=1), rnd(N, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5), rnd(N, seed=3, scale=0.1)
    out = ops.conv2d(x, ops.pack_conv_weight(w), b)
    ref = F.conv2d(nchw(x), w.float(), b.float(), padding=1)
    assert rel_l2(nchw(out), ref) < 1e-3
    border = torch.ones(H, H, dtype=torch.bool); border[1:-1, 1:-1] = False
    assert rel_l2(nchw(out)[..., border], ref[..., border]) < 1e-3
