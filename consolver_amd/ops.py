"""Thin torch-tensor wrappers over the op-level C ABI (include/consolver_hip_ops.h).

Used by the kernel parity tests and available to callers that want the individual HIP ops.
Activations are NHWC fp16 CUDA tensors.
"""
import ctypes as C

import torch

from . import _lib as L


def _f16(t, name):
    L.require_cuda(t, name)
    if t.dtype != torch.float16 or not t.is_contiguous():
        raise TypeError(f"{name} must be a contiguous float16 tensor")
    return t


def pack_conv_weight(w):
    """[Cout, Cin, kh, kw] -> [Cout, kh*kw, Cin] fp16 (tap-major, channel-minor)."""
    co, ci = w.shape[0], w.shape[1]
    return w.reshape(co, ci, -1).permute(0, 2, 1).contiguous().to(torch.float16)


_SPLITK_WS = {}


def conv2d(x0, w_packed, bias=None, x1=None, taps=9, stride=1, upsample=False, temb=None, res=None, splitk=True, gn_stats=False):
    """gn_stats=True: also return the GroupNorm partial sums of the output, fp32 [B, Ho*Wo/64, N/2, 2] (cs_op_conv2d_gn)."""
    _f16(x0, "x0")
    B, Hi, Wi, c0 = x0.shape
    c1 = x1.shape[-1] if x1 is not None else 0
    N = w_packed.shape[0]
    Ho = 2 * Hi if upsample else (Hi // 2 if stride == 2 else Hi)
    Wo = 2 * Wi if upsample else (Wi // 2 if stride == 2 else Wi)
    out = torch.empty(B, Ho, Wo, N, dtype=torch.float16, device=x0.device)
    tstride = 0 if (temb is None or temb.shape[0] == 1) else temb.shape[1]
    ws = None
    if splitk and taps == 9:
        ws = _SPLITK_WS.get(x0.device)
        if ws is None:
            ws = _SPLITK_WS[x0.device] = torch.empty(64 << 20, dtype=torch.uint8, device=x0.device)
    if gn_stats:
        st = torch.zeros(B, Ho * Wo // 64, N // 2, 2, dtype=torch.float32, device=x0.device)
        L.check(L.lib().cs_op_conv2d_gn(L.ptr(x0), c0, L.ptr(x1), c1, B, Hi, Wi, taps, stride, int(upsample), L.ptr(w_packed),
                                        L.ptr(bias), N, L.ptr(temb), tstride, L.ptr(res), L.ptr(out),
                                        L.ptr(ws), ws.numel() if ws is not None else 0, L.ptr(st), L.stream_ptr(x0.device)))
        return out, st
    L.check(L.lib().cs_op_conv2d(L.ptr(x0), c0, L.ptr(x1), c1, B, Hi, Wi, taps, stride, int(upsample), L.ptr(w_packed),
                                 L.ptr(bias), N, L.ptr(temb), tstride, L.ptr(res), L.ptr(out),
                                 L.ptr(ws), ws.numel() if ws is not None else 0, L.stream_ptr(x0.device)))
    return out


def conv_up_fold_pack(w_packed):
    """the sub-pixel filters of a nearest-x2 upsample + 3x3 conv: w_packed [N, 9 Cin] (tap-major) -> [4, N, 4 Cin] on the CPU (cs_op_conv_up_fold_pack;
    the executor packs its three upsamplers with the same function)"""
    w = w_packed.detach().to("cpu", torch.float16).reshape(w_packed.shape[0], -1).contiguous()
    N, K = w.shape
    if K % 9:
        raise ValueError("conv_up_fold_pack: a 3x3 filter packed [N, 9 Cin]")
    out = torch.empty(4, N, 4 * (K // 9), dtype=torch.float16)
    L.check(L.lib().cs_op_conv_up_fold_pack(C.c_void_p(w.data_ptr()), N, K // 9, C.c_void_p(out.data_ptr())))
    return out


def conv_up_sub(x, w_packed, w_sub, bias=None, gn_stats=False):
    """nearest-x2 upsample + 3x3 conv through the sub-pixel kernel (cs_op_conv_up_sub): x [B, Hi, Wi, Cin] -> [B, 2 Hi, 2 Wi, N]"""
    _f16(x, "x")
    B, Hi, Wi, Cin = x.shape
    N = w_packed.shape[0]
    out = torch.empty(B, 2 * Hi, 2 * Wi, N, dtype=torch.float16, device=x.device)
    st = torch.zeros(B, 4 * Hi * Wi // 64, N // 2, 2, dtype=torch.float32, device=x.device) if gn_stats else None
    L.check(L.lib().cs_op_conv_up_sub(L.ptr(x), Cin, B, Hi, Wi, L.ptr(w_packed), L.ptr(w_sub), L.ptr(bias), N, L.ptr(out), L.ptr(st), L.stream_ptr(x.device)))
    return (out, st) if gn_stats else out


def linear(x, w, bias=None, res=None, geglu=False, out=None):
    _f16(x, "x")
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N // 2 if geglu else N, dtype=torch.float16, device=x.device)
    L.check(L.lib().cs_op_linear(L.ptr(x), M, K, L.ptr(w), L.ptr(bias), N, L.ptr(res), L.ptr(out), int(geglu),
                                 L.stream_ptr(x.device)))
    return out


def split_f16(x):
    """fp32 tensor -> (hi, lo) fp16 planes with hi + lo == x to 22 bits (the split-fp16 residual-stream representation, CS_RESIDUAL_F16X2)."""
    hi = x.to(torch.float16)
    lo = (x.float() - hi.float()).to(torch.float16)
    return hi.contiguous(), lo.contiguous()


def conv2d_x2(x0, w_packed, bias=None, x1=None, taps=9, stride=1, upsample=False, temb=None, res=None, res_lo=None, splitk=True, x0_lo=None, x1_lo=None,
              row_stats=False, want_lo=True):
    """cs_op_conv2d_x2: returns (out_hi, out_lo); res / res_lo are the planes of a split-fp16 residual (res_lo may be None); x0_lo / x1_lo (1x1 only)
    the lo planes of a split-fp16 A operand."""
    _f16(x0, "x0")
    B, Hi, Wi, c0 = x0.shape
    c1 = x1.shape[-1] if x1 is not None else 0
    N = w_packed.shape[0]
    Ho = 2 * Hi if upsample else (Hi // 2 if stride == 2 else Hi)
    Wo = 2 * Wi if upsample else (Wi // 2 if stride == 2 else Wi)
    out = torch.empty(B, Ho, Wo, N, dtype=torch.float16, device=x0.device)
    out_lo = torch.empty_like(out) if want_lo else None
    tstride = 0 if (temb is None or temb.shape[0] == 1) else temb.shape[1]
    ws = None
    if splitk:
        ws = _SPLITK_WS.get(x0.device)
        if ws is None:
            ws = _SPLITK_WS[x0.device] = torch.empty(64 << 20, dtype=torch.uint8, device=x0.device)
    rs, G = (torch.zeros(B * Ho * Wo, N // 64, 2, dtype=torch.float32, device=x0.device), C.c_int(0)) if row_stats else (None, None)
    L.check(L.lib().cs_op_conv2d_x2(L.ptr(x0), L.ptr(x0_lo), c0, L.ptr(x1), L.ptr(x1_lo), c1, B, Hi, Wi, taps, stride, int(upsample), L.ptr(w_packed),
                                    L.ptr(bias), N, L.ptr(temb), tstride, L.ptr(res), L.ptr(res_lo), L.ptr(out), L.ptr(out_lo),
                                    L.ptr(rs), C.byref(G) if row_stats else None,
                                    L.ptr(ws), ws.numel() if ws is not None else 0, L.stream_ptr(x0.device)))
    if row_stats:
        return out, out_lo, (rs, G.value)
    return out, out_lo


def linear_x2(x, w, bias=None, res=None, res_lo=None, splitk=True, want_lo=True, x_lo=None, row_stats=False, out=None, out_lo=None):
    """cs_op_linear_x2: returns (out_hi, out_lo) (out_lo None with want_lo=False: the fp16 rounding of the fp32 sum only); row_stats=True appends
    (stats [M, N/64, 2] fp32 of which the first G groups are valid, G): the row statistics a folded LayerNorm's consumer (linear_ln) reads."""
    _f16(x, "x")
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float16, device=x.device)
        out_lo = torch.empty_like(out) if want_lo else None
    ws = None
    if splitk:
        ws = _SPLITK_WS.get(x.device)
        if ws is None:
            ws = _SPLITK_WS[x.device] = torch.empty(64 << 20, dtype=torch.uint8, device=x.device)
    rs, G = (torch.zeros(M, N // 64, 2, dtype=torch.float32, device=x.device), C.c_int(0)) if row_stats else (None, None)
    L.check(L.lib().cs_op_linear_x2(L.ptr(x), L.ptr(x_lo), M, K, L.ptr(w), L.ptr(bias), N, L.ptr(res), L.ptr(res_lo), L.ptr(out), L.ptr(out_lo),
                                    L.ptr(rs), C.byref(G) if row_stats else None,
                                    L.ptr(ws), ws.numel() if ws is not None else 0, L.stream_ptr(x.device)))
    if row_stats:
        return out, out_lo, (rs, G.value)
    return out, out_lo


def lo8_split(v):
    """fp32 value -> (hi fp16, lo8 uint8): the representation of a stream tensor with an 8-bit lo plane (cs_op_linear_lo8): lo8 = e5m2(v - float(hi))"""
    hi = v.to(torch.float16)
    return hi, (v.float() - hi.float()).to(torch.float8_e5m2).view(torch.uint8)


def lo8_value(hi, lo8):
    """hi fp16 + lo8 (uint8 holding e5m2) -> fp32 value"""
    return hi.float() + lo8.view(torch.float8_e5m2).float()


def linear_lo8(x, w, bias=None, res=None, res_lo8=None, splitk=True, want_lo=True, row_stats=False):
    """cs_op_linear_lo8: linear_x2 with 8-bit (e5m2) lo planes: returns (out_hi fp16, out_lo8 uint8 or None)"""
    _f16(x, "x")
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=torch.float16, device=x.device)
    out_lo = torch.empty(M, N, dtype=torch.uint8, device=x.device) if want_lo else None
    ws = None
    if splitk:
        ws = _SPLITK_WS.get(x.device)
        if ws is None:
            ws = _SPLITK_WS[x.device] = torch.empty(64 << 20, dtype=torch.uint8, device=x.device)
    rs, G = (torch.zeros(M, N // 64, 2, dtype=torch.float32, device=x.device), C.c_int(0)) if row_stats else (None, None)
    L.check(L.lib().cs_op_linear_lo8(L.ptr(x), M, K, L.ptr(w), L.ptr(bias), N, L.ptr(res), L.ptr(res_lo8), L.ptr(out), L.ptr(out_lo),
                                     L.ptr(rs), C.byref(G) if row_stats else None,
                                     L.ptr(ws), ws.numel() if ws is not None else 0, L.stream_ptr(x.device)))
    if row_stats:
        return out, out_lo, (rs, G.value)
    return out, out_lo


def xattn_block_lo8(h, h_lo8, ln_gamma, ln_beta, wq, kv, wo, bo, heads=8, eps=1e-5, hw=None, row_stats=False):
    """cs_op_xattn_block_lo8: xattn_block_x2 on a stream with 8-bit lo planes"""
    _f16(h, "h")
    M, Cc = h.shape
    hw = hw or M // kv.shape[0]
    out = torch.empty_like(h)
    out_lo = torch.empty(M, Cc, dtype=torch.uint8, device=h.device)
    rs = torch.zeros(M, 1, 2, dtype=torch.float32, device=h.device) if row_stats else None
    L.check(L.lib().cs_op_xattn_block_lo8(L.ptr(h), L.ptr(h_lo8), L.ptr(ln_gamma), L.ptr(ln_beta), float(eps), L.ptr(wq), L.ptr(kv), kv.shape[1],
                                          L.ptr(wo), L.ptr(bo), M, hw, Cc, heads, float((Cc // heads) ** -0.5), L.ptr(out), L.ptr(out_lo), L.ptr(rs),
                                          L.stream_ptr(h.device)))
    if row_stats:
        return out, out_lo, rs
    return out, out_lo


def row_stats_lo8(x, x_lo8):
    _f16(x, "x")
    M, Cc = x.shape
    st = torch.empty(M, 1, 2, dtype=torch.float32, device=x.device)
    L.check(L.lib().cs_op_row_stats_lo8(L.ptr(x), L.ptr(x_lo8), M, Cc, L.ptr(st), L.stream_ptr(x.device)))
    return st


def ln_dc_ratio(stats, C_in, eps=1e-5):
    """RMS over the rows of |mean| / sigma, from row statistics [M, groups, 2] (cs_op_ln_dc_ratio): the figure cs_unet_calibrate_ln_fold compares with its
    bound to decide whether a transformer block keeps its LayerNorms folded into the consuming GEMMs"""
    M, G, _ = stats.shape
    acc = torch.zeros(1, dtype=torch.float32, device=stats.device)
    L.check(L.lib().cs_op_ln_dc_ratio(L.ptr(stats), M, G, C_in, eps, L.ptr(acc), L.stream_ptr(stats.device)))
    return float((acc / M).sqrt())


def ln_fold_pack(w, bias, gamma, beta):
    """host-side folding of a LayerNorm (gamma, beta) into the linear layer w [N, K] (+ bias) that consumes it: returns (W' fp16 [N, K], s fp32 [N], b' fp32 [N])
    on the CPU (cs_op_ln_fold_pack; the executor packs its weights with the same function)."""
    w = w.detach().to("cpu", torch.float16).contiguous()
    g = gamma.detach().to("cpu", torch.float16).contiguous()
    be = beta.detach().to("cpu", torch.float16).contiguous()
    b = bias.detach().to("cpu", torch.float16).contiguous() if bias is not None else None
    N, K = w.shape
    wo, so, bo = torch.empty_like(w), torch.empty(N, dtype=torch.float32), torch.empty(N, dtype=torch.float32)
    L.check(L.lib().cs_op_ln_fold_pack(C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()) if b is not None else None, C.c_void_p(g.data_ptr()),
                                       C.c_void_p(be.data_ptr()), N, K, C.c_void_p(wo.data_ptr()), C.c_void_p(so.data_ptr()), C.c_void_p(bo.data_ptr())))
    return wo, so, bo


def linear_ln(x, w_folded, ln_s, ln_b, stats, groups, eps=1e-5, geglu=False, splitk=True):
    """out = LayerNorm(h) W^T + b with the LayerNorm folded in (cs_op_linear_ln): x is the RAW hidden state [M, K] fp16, (stats, groups) the row statistics
    its producer left (linear_x2 / conv2d_x2 / xattn_block_x2 with row_stats=True, or row_stats())."""
    _f16(x, "x")
    M, K = x.shape
    N = w_folded.shape[0]
    out = torch.empty(M, N // 2 if geglu else N, dtype=torch.float16, device=x.device)
    ws = None
    if splitk:
        ws = _SPLITK_WS.get(x.device)
        if ws is None:
            ws = _SPLITK_WS[x.device] = torch.empty(64 << 20, dtype=torch.uint8, device=x.device)
    L.check(L.lib().cs_op_linear_ln(L.ptr(x), M, K, L.ptr(w_folded), L.ptr(ln_s), L.ptr(ln_b), N, L.ptr(stats), int(groups), float(eps), L.ptr(out), int(geglu),
                                    L.ptr(ws), ws.numel() if ws is not None else 0, L.stream_ptr(x.device)))
    return out


def row_stats(x, x_lo=None):
    """(sum, sum of squares) of every row of x [M, C] (+ x_lo): fp32 [M, 1, 2] (cs_op_row_stats)"""
    _f16(x, "x")
    M, Cc = x.shape
    st = torch.empty(M, 1, 2, dtype=torch.float32, device=x.device)
    L.check(L.lib().cs_op_row_stats(L.ptr(x), L.ptr(x_lo), M, Cc, L.ptr(st), L.stream_ptr(x.device)))
    return st


def group_norm_x2(x0, x0_lo, gamma, beta, groups=32, eps=1e-5, silu=False, x1=None, x1_lo=None):
    _f16(x0, "x0")
    B = x0.shape[0]
    c0 = x0.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    HW = x0.numel() // (B * c0)
    ws = torch.empty(int(L.lib().cs_op_group_norm_workspace(B, c0 + c1)), dtype=torch.uint8, device=x0.device)
    out = torch.empty(x0.shape[:-1] + (c0 + c1,), dtype=torch.float16, device=x0.device)
    L.check(L.lib().cs_op_group_norm_x2(L.ptr(x0), L.ptr(x0_lo), c0, L.ptr(x1), L.ptr(x1_lo), c1, B, HW, groups, float(eps), int(silu),
                                        L.ptr(gamma), L.ptr(beta), L.ptr(ws), L.ptr(out), L.stream_ptr(x0.device)))
    return out


def layer_norm_x2(x, x_lo, gamma, beta, eps=1e-5):
    _f16(x, "x")
    M, Cc = x.shape
    out = torch.empty_like(x)
    L.check(L.lib().cs_op_layer_norm_x2(L.ptr(x), L.ptr(x_lo), L.ptr(gamma), L.ptr(beta), L.ptr(out), M, Cc, float(eps), L.stream_ptr(x.device)))
    return out


def xattn_block_x2(h, h_lo, ln_gamma, ln_beta, wq, kv, wo, bo, heads=8, eps=1e-5, hw=None, row_stats=False):
    """h_lo=None: plain fp16 stream (out_lo is None then); row_stats=True appends the [M, 1, 2] row statistics of the output"""
    _f16(h, "h")
    M, Cc = h.shape
    hw = hw or M // kv.shape[0]
    out = torch.empty_like(h)
    out_lo = torch.empty_like(h) if h_lo is not None else None
    rs = torch.zeros(M, 1, 2, dtype=torch.float32, device=h.device) if row_stats else None
    L.check(L.lib().cs_op_xattn_block_x2(L.ptr(h), L.ptr(h_lo), L.ptr(ln_gamma), L.ptr(ln_beta), float(eps), L.ptr(wq), L.ptr(kv), kv.shape[1],
                                         L.ptr(wo), L.ptr(bo), M, hw, Cc, heads, float((Cc // heads) ** -0.5), L.ptr(out), L.ptr(out_lo), L.ptr(rs),
                                         L.stream_ptr(h.device)))
    if row_stats:
        return out, out_lo, rs
    return out, out_lo


def geglu_pack(w, b):
    """host-side permutation of a [2*Hd, K] GEGLU projection into (16 value | 16 gate) row blocks."""
    w = w.detach().to("cpu", torch.float16).contiguous()
    b = b.detach().to("cpu", torch.float16).contiguous()
    wo, bo = torch.empty_like(w), torch.empty_like(b)
    L.check(L.lib().cs_op_geglu_pack(C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), w.shape[0] // 2, w.shape[1],
                                     C.c_void_p(wo.data_ptr()), C.c_void_p(bo.data_ptr())))
    return wo, bo


def attention(q, k, v, heads, scale=None, q_stride=None, k_stride=None, v_stride=None):
    """q: [B, Nq, H*dh], k/v: [B, Nk, H*dh] (or strided views described by *_stride in halfs)."""
    B, Nq = q.shape[0], q.shape[1]
    Nk = k.shape[1]
    C_ = q.shape[2] if q_stride is None else None
    dh = (C_ or 0) // heads if C_ else None
    if dh is None:
        raise ValueError("pass dense q")
    out = torch.empty(B, Nq, heads * dh, dtype=torch.float16, device=q.device)
    L.check(L.lib().cs_op_attention(L.ptr(q), q_stride or q.stride(1), L.ptr(k), k_stride or k.stride(1), L.ptr(v),
                                    v_stride or v.stride(1), L.ptr(out), heads * dh, B, heads, Nq, Nk, dh,
                                    float(scale if scale is not None else dh ** -0.5), L.stream_ptr(q.device)))
    return out


def conv_out(x, w_packed, bias, postprocess=False):
    """3x3 conv (pad 1) NHWC x [B, H, W, Cin] -> NCHW [B, Cout, H, W] for a small Cout (cs_op_conv_out: the UNet's eps head, the VAE's image / moments heads)."""
    _f16(x, "x")
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    out = torch.empty(B, Cout, H, W, dtype=torch.float16, device=x.device)
    L.check(L.lib().cs_op_conv_out(L.ptr(x.contiguous()), B, Cin, H, W, L.ptr(w_packed.contiguous()), L.ptr(bias.to(x.device, torch.float16).contiguous()), Cout,
                                   L.ptr(out), int(postprocess), L.stream_ptr(x.device)))
    return out


def group_norm(x0, gamma, beta, groups=32, eps=1e-5, silu=False, x1=None, stats0=None, stats1=None):
    """stats0 / stats1: partial sums of x0 / x1 written by their producer (conv2d(gn_stats=True)); that source's statistics pass is skipped."""
    _f16(x0, "x0")
    B = x0.shape[0]
    c0 = x0.shape[-1]
    c1 = x1.shape[-1] if x1 is not None else 0
    HW = x0.numel() // (B * c0)
    ws = torch.empty(int(L.lib().cs_op_group_norm_workspace(B, c0 + c1)), dtype=torch.uint8, device=x0.device)
    out = torch.empty(x0.shape[:-1] + (c0 + c1,), dtype=torch.float16, device=x0.device)
    if stats0 is not None or stats1 is not None:
        L.check(L.lib().cs_op_group_norm_pre(L.ptr(x0), c0, L.ptr(stats0), L.ptr(x1), c1, L.ptr(stats1), B, HW, groups, float(eps), int(silu),
                                             L.ptr(gamma), L.ptr(beta), L.ptr(ws), L.ptr(out), L.stream_ptr(x0.device)))
        return out
    L.check(L.lib().cs_op_group_norm(L.ptr(x0), c0, L.ptr(x1), c1, B, HW, groups, float(eps), int(silu), L.ptr(gamma),
                                     L.ptr(beta), L.ptr(ws), L.ptr(out), L.stream_ptr(x0.device)))
    return out


def layer_norm(x, gamma, beta, eps=1e-5):
    _f16(x, "x")
    M, Cc = x.shape
    out = torch.empty_like(x)
    L.check(L.lib().cs_op_layer_norm(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(out), M, Cc, float(eps), L.stream_ptr(x.device)))
    return out


def xattn_block(h, ln_gamma, ln_beta, wq, kv, wo, bo, heads=8, eps=1e-5, hw=None, out=None):
    """fused cross-attention sub-block: out = h + to_out(softmax(scale q k^T) v) with q = LayerNorm(h) wq^T; h [M, C] (M = samples * hw rows),
    kv [samples, Nk, 2C] = (K | V) projections of the context.  C = 320, 8 heads, Nk <= 80."""
    _f16(h, "h")
    M, Cc = h.shape
    hw = hw or M // kv.shape[0]
    if out is None:
        out = torch.empty_like(h)
    L.check(L.lib().cs_op_xattn_block(L.ptr(h), L.ptr(ln_gamma), L.ptr(ln_beta), float(eps), L.ptr(wq), L.ptr(kv), kv.shape[1], L.ptr(wo),
                                      L.ptr(bo), M, hw, Cc, heads, float((Cc // heads) ** -0.5), L.ptr(out), L.stream_ptr(h.device)))
    return out


def set_tuning(key, value):
    """kernel-selection knob, e.g. set_tuning("conv_halo", 2) forces the halo-resident conv3x3 kernel."""
    L.check(L.lib().cs_set_tuning(key.encode(), int(value)))


def get_tuning(key):
    v = C.c_int(0)
    L.check(L.lib().cs_get_tuning(key.encode(), C.byref(v)))
    return v.value


def reset_tuning():
    """every kernel-selection knob back to its default"""
    L.check(L.lib().cs_reset_tuning())
