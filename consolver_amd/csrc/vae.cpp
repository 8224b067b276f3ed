// AutoencoderKL decoder executor (SD1.5 VAE): latents [B,4,h,w] -> images [B,3,8h,8w].
//
// Replaces `vae.decode(latents / vae.config.scaling_factor).sample` followed by
// `(image / 2 + 0.5).clamp(0, 1)` in decode_latents (utils.py:6-34; same two lines in
// gen_pretrain/pipeline.py:589-593).  diffusers==0.26.3 AutoencoderKL: post_quant_conv (1x1), Decoder =
// conv_in -> UNetMidBlock2D (resnet, 1-head attention, resnet) -> 4 UpDecoderBlock2D (3 resnets each,
// nearest-x2 upsample + conv on the first three) -> GroupNorm(32, eps 1e-6) -> SiLU -> conv_out.
//
// Activations are NHWC fp16 like the UNet executor; every 3x3 conv runs through the same implicit-GEMM
// MFMA kernels (igemm.hip).  The mid-block attention has ONE head of dim 512 over 4096 tokens, which is
// GEMM-shaped rather than flash-shaped: scores = Q K^T and O = P V are plain GEMMs through the same
// kernel, with a row-softmax kernel in between; V is produced directly transposed (V^T = Wv X^T) by
// swapping the GEMM operand roles, and its bias is applied after P V (rows of P sum to 1).
#include "ops.h"
#include "consolver_hip.h"

#include <map>
#include <string>
#include <vector>
#include <algorithm>
#include <cmath>


namespace {

struct HostT { std::vector<int64_t> shape; std::vector<f16> data; };
struct VConv { f16* w = nullptr; f16* b = nullptr; int cin = 0, cout = 0, taps = 1; f16* w_sub = nullptr; };     // w_sub: an upsampler's sub-pixel filters (IgemmArgs::w_up_sub)
struct VNorm { f16* g = nullptr; f16* b = nullptr; int c = 0; };
struct VResnet { VNorm n1, n2; VConv c1, c2, sc; bool has_sc = false; int cin = 0, cout = 0; };
struct VAttn { VNorm gn; f16 *wq = nullptr, *bq = nullptr, *wk = nullptr, *bk = nullptr, *wv = nullptr, *bv = nullptr, *wo = nullptr, *bo = nullptr; };

struct VArena {   // bump allocator with explicit stack discipline (mark / rewind)
    char* base = nullptr; size_t cap = 0, top = 0, peak = 0; bool dry = false;
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (!dry && top + bytes > cap) return nullptr;
        void* p = base + top; top += bytes; peak = std::max(peak, top);
        return p;
    }
};

constexpr int VAE_GN_SPLITS = 256;

}  // namespace

struct CsVae {
    CsVaeConfig cfg;
    std::vector<std::string> names;
    std::map<std::string, std::vector<int64_t>> expect;
    std::map<std::string, HostT> host;
    std::vector<void*> dev_allocs;
    bool finalized = false;
    f16 *pq_w = nullptr, *pq_b = nullptr;
    VConv conv_in, conv_out; VNorm norm_out;
    VResnet mid_res[2];
    VAttn att;
    std::vector<VResnet> up_res[4]; VConv up_samp[4];
    // encoder (optional, cfg.with_encoder): images -> mode of the latent distribution
    VConv e_conv_in, e_conv_out; VNorm e_norm_out; VResnet e_mid_res[2]; VAttn e_att;
    std::vector<VResnet> e_down_res[4]; VConv e_down_samp[4];
    f16 *q_w = nullptr, *q_b = nullptr;            // quant_conv rows of the mean half [L][2L], [L]
    VArena arena; double dry_flops = 0;
};

namespace {

void expect_tensor(CsVae* v, const std::string& n, std::vector<int64_t> shape) { v->names.push_back(n); v->expect[n] = std::move(shape); }
void expect_resnet(CsVae* v, const std::string& p, int cin, int cout) {
    expect_tensor(v, p + ".norm1.weight", {cin}); expect_tensor(v, p + ".norm1.bias", {cin});
    expect_tensor(v, p + ".conv1.weight", {cout, cin, 3, 3}); expect_tensor(v, p + ".conv1.bias", {cout});
    expect_tensor(v, p + ".norm2.weight", {cout}); expect_tensor(v, p + ".norm2.bias", {cout});
    expect_tensor(v, p + ".conv2.weight", {cout, cout, 3, 3}); expect_tensor(v, p + ".conv2.bias", {cout});
    if (cin != cout) { expect_tensor(v, p + ".conv_shortcut.weight", {cout, cin, 1, 1}); expect_tensor(v, p + ".conv_shortcut.bias", {cout}); }
}

void expect_attn(CsVae* v, const std::string& a, int top) {
    expect_tensor(v, a + ".group_norm.weight", {top}); expect_tensor(v, a + ".group_norm.bias", {top});
    for (const char* q : {".to_q", ".to_k", ".to_v", ".to_out.0"}) { expect_tensor(v, a + q + ".weight", {top, top}); expect_tensor(v, a + q + ".bias", {top}); }
}

void build_manifest(CsVae* v) {
    const CsVaeConfig& c = v->cfg;
    const int L = c.latent_channels, top = c.block_out_channels[3];
    if (c.with_encoder) {
        expect_tensor(v, "encoder.conv_in.weight", {c.block_out_channels[0], c.out_channels, 3, 3}); expect_tensor(v, "encoder.conv_in.bias", {c.block_out_channels[0]});
        int prev = c.block_out_channels[0];
        for (int i = 0; i < 4; ++i) {
            const int out = c.block_out_channels[i];
            const std::string b = "encoder.down_blocks." + std::to_string(i);
            for (int j = 0; j < c.layers_per_block; ++j) expect_resnet(v, b + ".resnets." + std::to_string(j), j == 0 ? prev : out, out);
            if (i < 3) { expect_tensor(v, b + ".downsamplers.0.conv.weight", {out, out, 3, 3}); expect_tensor(v, b + ".downsamplers.0.conv.bias", {out}); }
            prev = out;
        }
        expect_resnet(v, "encoder.mid_block.resnets.0", top, top);
        expect_attn(v, "encoder.mid_block.attentions.0", top);
        expect_resnet(v, "encoder.mid_block.resnets.1", top, top);
        expect_tensor(v, "encoder.conv_norm_out.weight", {top}); expect_tensor(v, "encoder.conv_norm_out.bias", {top});
        expect_tensor(v, "encoder.conv_out.weight", {2 * L, top, 3, 3}); expect_tensor(v, "encoder.conv_out.bias", {2 * L});
        if (c.use_quant_conv) { expect_tensor(v, "quant_conv.weight", {2 * L, 2 * L, 1, 1}); expect_tensor(v, "quant_conv.bias", {2 * L}); }
    }
    if (c.use_post_quant_conv) { expect_tensor(v, "post_quant_conv.weight", {L, L, 1, 1}); expect_tensor(v, "post_quant_conv.bias", {L}); }
    expect_tensor(v, "decoder.conv_in.weight", {top, L, 3, 3}); expect_tensor(v, "decoder.conv_in.bias", {top});
    expect_resnet(v, "decoder.mid_block.resnets.0", top, top);
    expect_attn(v, "decoder.mid_block.attentions.0", top);
    expect_resnet(v, "decoder.mid_block.resnets.1", top, top);
    int prev = top;
    for (int i = 0; i < 4; ++i) {
        const int out = c.block_out_channels[3 - i];
        const std::string b = "decoder.up_blocks." + std::to_string(i);
        for (int j = 0; j < c.layers_per_block + 1; ++j) expect_resnet(v, b + ".resnets." + std::to_string(j), j == 0 ? prev : out, out);
        if (i < 3) { expect_tensor(v, b + ".upsamplers.0.conv.weight", {out, out, 3, 3}); expect_tensor(v, b + ".upsamplers.0.conv.bias", {out}); }
        prev = out;
    }
    const int c0 = c.block_out_channels[0];
    expect_tensor(v, "decoder.conv_norm_out.weight", {c0}); expect_tensor(v, "decoder.conv_norm_out.bias", {c0});
    expect_tensor(v, "decoder.conv_out.weight", {c.out_channels, c0, 3, 3}); expect_tensor(v, "decoder.conv_out.bias", {c.out_channels});
}

f16* upload(CsVae* v, const std::vector<f16>& h) {
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(h.size() * sizeof(f16), 256)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(f16), hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return nullptr; }
    v->dev_allocs.push_back(d);
    return (f16*)d;
}
const HostT& T(CsVae* v, const std::string& n) { return v->host.at(n); }

std::vector<f16> pack_conv(const HostT& t) {   // [Cout][Cin][kh][kw] -> [Cout][kh*kw][Cin]
    const int64_t co = t.shape[0], ci = t.shape[1], kk = t.shape.size() == 4 ? t.shape[2] * t.shape[3] : 1;
    std::vector<f16> o((size_t)co * ci * kk);
    for (int64_t n = 0; n < co; ++n)
        for (int64_t c = 0; c < ci; ++c)
            for (int64_t k = 0; k < kk; ++k) o[(n * kk + k) * ci + c] = t.data[(n * ci + c) * kk + k];
    return o;
}
bool make_conv(CsVae* v, const std::string& p, VConv& c) {
    const HostT& w = T(v, p + ".weight");
    c.cout = (int)w.shape[0]; c.cin = (int)w.shape[1]; c.taps = w.shape.size() == 4 ? (int)(w.shape[2] * w.shape[3]) : 1;
    c.w = upload(v, pack_conv(w)); c.b = upload(v, T(v, p + ".bias").data);
    return c.w && c.b;
}
bool make_norm(CsVae* v, const std::string& p, VNorm& n) {
    n.c = (int)T(v, p + ".weight").shape[0];
    n.g = upload(v, T(v, p + ".weight").data); n.b = upload(v, T(v, p + ".bias").data);
    return n.g && n.b;
}
bool make_resnet(CsVae* v, const std::string& p, VResnet& r) {
    bool ok = make_norm(v, p + ".norm1", r.n1) && make_conv(v, p + ".conv1", r.c1) && make_norm(v, p + ".norm2", r.n2) && make_conv(v, p + ".conv2", r.c2);
    r.cin = r.c1.cin; r.cout = r.c1.cout;
    r.has_sc = v->host.count(p + ".conv_shortcut.weight") > 0;
    if (r.has_sc) ok = ok && make_conv(v, p + ".conv_shortcut", r.sc);
    return ok;
}

bool make_attn(CsVae* v, const std::string& a, VAttn& t) {
    bool ok = make_norm(v, a + ".group_norm", t.gn);
    t.wq = upload(v, T(v, a + ".to_q.weight").data); t.bq = upload(v, T(v, a + ".to_q.bias").data);
    t.wk = upload(v, T(v, a + ".to_k.weight").data); t.bk = upload(v, T(v, a + ".to_k.bias").data);
    t.wv = upload(v, T(v, a + ".to_v.weight").data); t.bv = upload(v, T(v, a + ".to_v.bias").data);
    t.wo = upload(v, T(v, a + ".to_out.0.weight").data); t.bo = upload(v, T(v, a + ".to_out.0.bias").data);
    return ok && t.wq && t.bq && t.wk && t.bk && t.wv && t.bv && t.wo && t.bo;
}
// 3x3 conv whose few input channels are zero-padded to 64 (the operand then arrives as NHWC-64): [co][ci][3][3] -> [co][9][64]
bool make_conv_padded64(CsVae* v, const std::string& p, VConv& c) {
    const HostT& w = T(v, p + ".weight");
    const int64_t co = w.shape[0], ci = w.shape[1];
    std::vector<f16> o((size_t)co * 9 * 64, (f16)0.f);
    for (int64_t n = 0; n < co; ++n) for (int64_t ch = 0; ch < ci; ++ch) for (int64_t k = 0; k < 9; ++k) o[(n * 9 + k) * 64 + ch] = w.data[(n * ci + ch) * 9 + k];
    c.cout = (int)co; c.cin = 64; c.taps = 9;
    c.w = upload(v, o); c.b = upload(v, T(v, p + ".bias").data);
    return c.w && c.b;
}

struct Run {
    CsVae* v; hipStream_t s; bool dry; int B; int rc = CS_OK;
    float* gn_ws = nullptr;
    // GroupNorm statistics a conv left for its output (IgemmArgs::gn_stats, [B][HW/64][C/2][2] partial sums): three rotating buffers are enough, a tensor is
    // normalised at most two convolutions after it was written (x -> conv1 -> conv2 (+ shortcut) -> next resnet)
    float* st_buf[3] = {nullptr, nullptr, nullptr}; const f16* st_of[3] = {nullptr, nullptr, nullptr}; int st_S[3] = {0, 0, 0}; int st_next = 0;
    size_t st_floats = 0;
    // gn_fuse is snapshotted once per decode / encode (the process-wide knob cannot change a run in flight); the buffers are ALWAYS reserved, so the
    // workspace size does not depend on the knob's value at query time
    bool v_gn_fuse = true;
    void init_stats(size_t floats) {
        v_gn_fuse = tune().gn_fuse != 0;
        st_floats = floats;
        for (auto& b : st_buf) b = (float*)alloc(floats * 2);           // (alloc counts halfs)
    }
    float* stats_for_output(const f16* out, int HW, int C) {
        if (!v_gn_fuse || !st_buf[0] || HW % 64 || C % 2 || (size_t)B * (HW / 64) * C > st_floats) return nullptr;
        const int k = st_next; st_next = (st_next + 1) % 3;
        st_of[k] = out; st_S[k] = HW / 64;
        return st_buf[k];
    }
    const float* stats_of(const f16* x, int* S) const {
        for (int k = 0; k < 3; ++k) if (st_of[k] == x && x) { *S = st_S[k]; return st_buf[k]; }
        return nullptr;
    }
    void forget_stats(const f16* t) { for (auto& o : st_of) if (o == t) o = nullptr; }      // the tensor is about to be overwritten by something that leaves no statistics

    f16* alloc(size_t halfs) {
        void* p = v->arena.alloc(halfs * sizeof(f16));
        if (!p && rc == CS_OK) { cs_set_error("vae: workspace too small"); rc = CS_E_ARG; }
        return p ? (f16*)p : reinterpret_cast<f16*>((uintptr_t)1 << 41);      // poison base, never dereferenced (nothing is launched once rc is set)
    }
    template <typename F> void launch(double flops, F&& f) {
        if (dry) { v->dry_flops += flops; return; }
        if (rc == CS_OK) rc = f();
    }
    void conv(const VConv& c, const f16* x, int H, int W, int up, const f16* res, f16* out) {
        IgemmArgs a{};
        a.a0 = x; a.c0 = c.cin; a.B = B; a.Hi = H; a.Wi = W; a.Ho = up ? 2 * H : H; a.Wo = up ? 2 * W : W; a.taps = c.taps; a.stride = 1;
        a.upsample = up; a.N = c.cout; a.w = c.w; a.bias = c.b; a.res = res; a.out = out;
        // the decoder's stream is one fp16 plane: its upsamplers run the sub-pixel form (16 instead of 36 multiplies per input pixel) unless up_fold is 0
        if (up && c.w_sub && tune().up_fold != 0 && !res && H % 16 == 0 && W % 16 == 0) a.w_up_sub = c.w_sub;
        forget_stats(out);
        a.gn_stats = c.taps == 9 ? stats_for_output(out, a.Ho * a.Wo, c.cout) : nullptr;
        launch(igemm_flops(a), [&] { return launch_igemm(a, s); });             // (a kernel without a statistics epilogue is followed by a statistics pass in the same layout)
    }
    void conv_down(const VConv& c, const f16* x, int H, int W, f16* out) {       // pad (0,1,0,1) + 3x3 stride 2 (encoder downsample)
        IgemmArgs a{};
        a.a0 = x; a.c0 = c.cin; a.B = B; a.Hi = H; a.Wi = W; a.Ho = H / 2; a.Wo = W / 2; a.taps = 9; a.stride = 2; a.pad_after_only = 1;
        a.N = c.cout; a.w = c.w; a.bias = c.b; a.out = out;
        forget_stats(out);
        launch(igemm_flops(a), [&] { return launch_igemm(a, s); });
    }
    void gemm(const f16* x, int M, int K, const f16* w, const f16* b, int N, const f16* res, f16* out) {
        IgemmArgs a{};
        a.a0 = x; a.c0 = K; a.B = 1; a.Hi = M; a.Wi = 1; a.Ho = M; a.Wo = 1; a.taps = 1; a.stride = 1; a.N = N; a.w = w; a.bias = b; a.res = res; a.out = out;
        forget_stats(out);
        launch(igemm_flops(a), [&] { return launch_igemm(a, s); });
    }
    void group_norm(const VNorm& n, const f16* x, int HW, bool silu, f16* out) {
        GroupNormArgs a{};
        a.x0 = x; a.c0 = n.c; a.B = B; a.HW = HW; a.groups = v->cfg.norm_num_groups; a.eps = 1e-6f; a.silu = silu;
        a.gamma = n.g; a.beta = n.b; a.partial = gn_ws; a.out = out; a.splits = VAE_GN_SPLITS;
        int S = 0;
        if (const float* st = stats_of(x, &S)) { a.stats0 = st; a.S0 = S; }
        forget_stats(out);
        launch(0, [&] { return launch_group_norm(a, s); });
    }
    // x [B,HW,cin] -> out [B,HW,cout]; n/h are scratch of the larger channel count
    void resnet(const VResnet& r, const f16* x, int H, int W, f16* n, f16* h, f16* out) {
        const int HW = H * W;
        group_norm(r.n1, x, HW, true, n);
        conv(r.c1, n, H, W, 0, nullptr, h);
        group_norm(r.n2, h, HW, true, n);
        const f16* res = x;
        if (r.has_sc) { conv(r.sc, x, H, W, 0, nullptr, out); res = out; }
        conv(r.c2, n, H, W, 0, res, out);
    }
    // single-head attention of the mid block: x [B,HW,C] -> y (= x + to_out(softmax(q k^T / sqrt C) v)); n, h scratch
    void mid_attention(const VAttn& t, const f16* x, f16* y, f16* n, f16* h, int HW, int C) {
        const size_t M = (size_t)B * HW;
        group_norm(t.gn, x, HW, false, n);
        f16* q = h;                       // [M][C]
        f16* k = y;                       // [M][C]
        gemm(n, (int)M, C, t.wq, t.bq, C, nullptr, q);
        gemm(n, (int)M, C, t.wk, t.bk, C, nullptr, k);
        const size_t mark = v->arena.top;
        f16* scores = alloc((size_t)HW * HW);
        f16* vt = alloc((size_t)C * HW);
        f16* att = alloc(M * C);
        if (rc != CS_OK) return;
        for (int b = 0; b < B; ++b) {
            const f16* qb = q + (size_t)b * HW * C; const f16* kb = k + (size_t)b * HW * C; const f16* nb = n + (size_t)b * HW * C;
            gemm(qb, HW, C, kb, nullptr, HW, nullptr, scores);                       // S = Q K^T          [HW][HW]
            launch(0, [&] { return launch_row_softmax(scores, HW, HW, 1.0f / sqrtf((float)C), s); });
            gemm(t.wv, C, C, nb, nullptr, HW, nullptr, vt);                          // V^T = Wv X^T       [C][HW]
            gemm(scores, HW, HW, vt, t.bv, C, nullptr, att + (size_t)b * HW * C);     // O = P V + bv       [HW][C]
        }
        gemm(att, (int)M, C, t.wo, t.bo, C, x, y);                                   // to_out + residual
        v->arena.top = mark;
    }
};

size_t gn_ws_bytes(const CsVae* v, int B) {
    return (((size_t)B * (VAE_GN_SPLITS + 1) * v->cfg.block_out_channels[3] * 2 * sizeof(float)) + 255) & ~(size_t)255;
}

int run_decode(CsVae* v, bool dry, const f16* latents, int B, float in_scale, float in_shift, f16* out, int postprocess, char* ws, size_t ws_bytes,
               hipStream_t s) {
    const CsVaeConfig& c = v->cfg;
    Run R{v, s, dry, B};
    v->dry_flops = 0;
    size_t off = 0;
    if (!dry) {
        if (ws_bytes < gn_ws_bytes(v, B)) CS_FAIL(CS_E_ARG, "vae: workspace too small");
        R.gn_ws = (float*)ws; off = gn_ws_bytes(v, B);
    }
    v->arena.base = dry ? (char*)256 : ws + off; v->arena.cap = dry ? 0 : ws_bytes - off; v->arena.top = 0; v->arena.peak = 0; v->arena.dry = dry;

    int H = c.sample_size, W = c.sample_size;
    const int top = c.block_out_channels[3], L = c.latent_channels;
    // ping-pong buffers sized for the largest activation of the whole decoder
    size_t maxact = 0;
    {
        int h = H, ch = top; maxact = (size_t)h * h * ch;
        for (int i = 0; i < 4; ++i) { ch = c.block_out_channels[3 - i]; const int cin = i ? c.block_out_channels[4 - i] : top;
            maxact = std::max(maxact, (size_t)h * h * std::max(ch, cin)); if (i < 3) { h *= 2; maxact = std::max(maxact, (size_t)h * h * ch); } }
    }
    const bool wide = L != 4;                    // FLUX: 16 latent channels -> NHWC-64 + implicit-GEMM conv_in
    f16* z = R.alloc((size_t)B * (wide ? 64 : L) * H * W);
    f16* bufs[4];
    for (auto& b : bufs) b = R.alloc((size_t)B * maxact);
    R.init_stats((size_t)B * (maxact / 64));
    if (R.rc != CS_OK) return R.rc;
    f16 *x = bufs[0], *y = bufs[1], *n = bufs[2], *h = bufs[3];

    if (!wide) {
        R.launch(0, [&] { return launch_pixel_linear_nchw(latents, v->pq_w, v->pq_b, z, B, L, H * W, in_scale, in_shift, s); });
        R.launch(2.0 * B * H * W * 9.0 * L * top, [&] { return launch_conv_in(z, B, B, L, H, W, v->conv_in.w, v->conv_in.b, top, x, s); });
    } else {
        R.launch(0, [&] { return launch_latent_to_nhwc64(latents, v->pq_w, v->pq_b, z, B, L, H * W, in_scale, in_shift, s); });
        R.conv(v->conv_in, z, H, W, 0, nullptr, x);          // weights zero-padded to 64 input channels at load
    }

    // ---- mid block
    R.resnet(v->mid_res[0], x, H, W, n, h, y); std::swap(x, y);
    R.mid_attention(v->att, x, y, n, h, H * W, top); std::swap(x, y);
    if (R.rc != CS_OK) return R.rc;
    R.resnet(v->mid_res[1], x, H, W, n, h, y); std::swap(x, y);

    // ---- up blocks
    for (int i = 0; i < 4; ++i) {
        for (auto& r : v->up_res[i]) { R.resnet(r, x, H, W, n, h, y); std::swap(x, y); }
        if (i < 3) { R.conv(v->up_samp[i], x, H, W, 1, nullptr, y); std::swap(x, y); H *= 2; W *= 2; }
    }
    R.group_norm(v->norm_out, x, H * W, true, n);
    const int c0 = c.block_out_channels[0];
    R.launch(2.0 * B * H * W * 9.0 * c0 * c.out_channels,
             [&] { return launch_conv_out3(n, B, c0, H, W, v->conv_out.w, v->conv_out.b, out, postprocess, s); });
    return R.rc;
}


int run_encode(CsVae* v, bool dry, const f16* images, int B, float out_scale, float out_shift, f16* latents, char* ws, size_t ws_bytes, hipStream_t s) {
    const CsVaeConfig& c = v->cfg;
    Run R{v, s, dry, B};
    v->dry_flops = 0;
    size_t off = 0;
    if (!dry) {
        if (ws_bytes < gn_ws_bytes(v, B)) CS_FAIL(CS_E_ARG, "vae: workspace too small");
        R.gn_ws = (float*)ws; off = gn_ws_bytes(v, B);
    }
    v->arena.base = dry ? (char*)256 : ws + off; v->arena.cap = dry ? 0 : ws_bytes - off; v->arena.top = 0; v->arena.peak = 0; v->arena.dry = dry;
    int H = 8 * c.sample_size, W = 8 * c.sample_size;
    const int top = c.block_out_channels[3], L = c.latent_channels;
    size_t maxact = 0;
    { int h = H; for (int i = 0; i < 4; ++i) { const int ch = c.block_out_channels[i], cin = i ? c.block_out_channels[i - 1] : c.block_out_channels[0];
          maxact = std::max(maxact, (size_t)h * h * std::max(std::max(ch, cin), 64)); if (i < 3) h /= 2; } }
    f16* bufs[4];
    for (auto& b : bufs) b = R.alloc((size_t)B * maxact);
    const int mo = c.use_quant_conv ? 2 * L : L;
    f16* mom = R.alloc((size_t)B * mo * c.sample_size * c.sample_size);
    R.init_stats((size_t)B * (maxact / 64));
    if (R.rc != CS_OK) return R.rc;
    f16 *x = bufs[0], *y = bufs[1], *n = bufs[2], *h = bufs[3];
    R.launch(0, [&] { return launch_latent_to_nhwc64(images, nullptr, nullptr, n, B, c.out_channels, H * W, 1.0f, 0.0f, s); });
    R.conv(v->e_conv_in, n, H, W, 0, nullptr, x);
    for (int i = 0; i < 4; ++i) {
        for (auto& r : v->e_down_res[i]) { R.resnet(r, x, H, W, n, h, y); std::swap(x, y); }
        if (i < 3) { R.conv_down(v->e_down_samp[i], x, H, W, y); std::swap(x, y); H /= 2; W /= 2; }
    }
    R.resnet(v->e_mid_res[0], x, H, W, n, h, y); std::swap(x, y);
    R.mid_attention(v->e_att, x, y, n, h, H * W, top); std::swap(x, y);
    if (R.rc != CS_OK) return R.rc;
    R.resnet(v->e_mid_res[1], x, H, W, n, h, y); std::swap(x, y);
    R.group_norm(v->e_norm_out, x, H * W, true, n);
    R.launch(2.0 * B * H * W * 9.0 * top * mo, [&] { return launch_conv_out_small(n, B, top, H, W, v->e_conv_out.w, v->e_conv_out.b, mo, mom, s); });
    // mode of the distribution = mean [= quant_conv rows of the mean half]; (mean - shift) * scale folded into the same pass
    R.launch(0, [&] { return launch_pixel_affine_nchw(mom, mo, v->q_w, v->q_b, L, latents, B, H * W, out_scale, out_shift, s); });
    return R.rc;
}

}  // namespace

extern "C" {

int cs_vae_create(const CsVaeConfig* cfg, CsVae** out) {
    if (!cfg || !out) CS_FAIL(CS_E_ARG, "cfg/out is NULL");
    if ((cfg->latent_channels != 4 && cfg->latent_channels != 16) || cfg->out_channels != 3) CS_FAIL(CS_E_UNSUPPORTED, "vae: built for 4 or 16 latent / 3 image channels");
    if (cfg->latent_channels == 4 && !cfg->use_post_quant_conv) CS_FAIL(CS_E_UNSUPPORTED, "vae: the 4-channel path expects post_quant_conv");
    for (int i = 0; i < 4; ++i)
        if (cfg->block_out_channels[i] % 128) CS_FAIL(CS_E_SHAPE, "vae: block_out_channels[%d]=%d must be a multiple of 128", i, cfg->block_out_channels[i]);
    if (cfg->block_out_channels[3] != cfg->block_out_channels[2]) CS_FAIL(CS_E_UNSUPPORTED, "vae: the two deepest blocks must have equal width");
    if (cfg->sample_size <= 0 || (cfg->sample_size * cfg->sample_size) % 128 || cfg->sample_size * cfg->sample_size > 65536)
        CS_FAIL(CS_E_SHAPE, "vae: latent sample_size %d unsupported (tokens must be a multiple of 128 and <= 65536)", cfg->sample_size);
    if (cfg->layers_per_block < 1 || cfg->norm_num_groups < 1) CS_FAIL(CS_E_ARG, "vae: bad layers_per_block / norm_num_groups");
    if (cfg->with_encoder && cfg->sample_size % 2) CS_FAIL(CS_E_SHAPE, "vae: the encoder needs an even latent size");
    CsVae* v = new CsVae();
    v->cfg = *cfg;
    build_manifest(v);
    *out = v;
    return CS_OK;
}

void cs_vae_destroy(CsVae* v) {
    if (!v) return;
    for (void* p : v->dev_allocs) hipFree(p);
    delete v;
}

int cs_vae_num_weights(const CsVae* v) { return v ? (int)v->names.size() : 0; }

const char* cs_vae_weight_name(const CsVae* v, int i, int64_t* shape4, int* ndim) {
    if (!v || i < 0 || i >= (int)v->names.size()) return nullptr;
    const auto& sh = v->expect.at(v->names[i]);
    if (ndim) *ndim = (int)sh.size();
    if (shape4) for (size_t k = 0; k < 4; ++k) shape4[k] = k < sh.size() ? sh[k] : 1;
    return v->names[i].c_str();
}

int cs_vae_set_weight(CsVae* v, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!v || !name || !data || !shape) CS_FAIL(CS_E_ARG, "null argument");
    if (v->finalized) CS_FAIL(CS_E_STATE, "weights are already packed");
    auto it = v->expect.find(name);
    if (it == v->expect.end()) CS_FAIL(CS_E_ARG, "unexpected tensor name '%s'", name);
    if ((int)it->second.size() != ndim) CS_FAIL(CS_E_SHAPE, "%s: rank %d, expected %zu", name, ndim, it->second.size());
    int64_t n = 1;
    for (int k = 0; k < ndim; ++k) {
        if (shape[k] != it->second[k]) CS_FAIL(CS_E_SHAPE, "%s: dim %d is %lld, expected %lld", name, k, (long long)shape[k], (long long)it->second[k]);
        n *= shape[k];
    }
    HostT t; t.shape.assign(shape, shape + ndim); t.data.resize(n);
    for (int64_t i = 0; i < n; ++i) t.data[i] = (f16)data[i];
    v->host[name] = std::move(t);
    return CS_OK;
}

int cs_vae_finalize(CsVae* v) {
    if (!v) CS_FAIL(CS_E_ARG, "null");
    if (v->finalized) return CS_OK;
    for (auto& n : v->names) if (!v->host.count(n)) CS_FAIL(CS_E_STATE, "missing weight '%s'", n.c_str());
    bool ok = true;
    if (v->cfg.use_post_quant_conv) { v->pq_w = upload(v, T(v, "post_quant_conv.weight").data); v->pq_b = upload(v, T(v, "post_quant_conv.bias").data); ok = v->pq_w && v->pq_b; }
    if (v->cfg.latent_channels != 4) ok = ok && make_conv_padded64(v, "decoder.conv_in", v->conv_in);
    else ok = ok && make_conv(v, "decoder.conv_in", v->conv_in);
    ok = ok && make_conv(v, "decoder.conv_out", v->conv_out) && make_norm(v, "decoder.conv_norm_out", v->norm_out);
    ok = ok && make_resnet(v, "decoder.mid_block.resnets.0", v->mid_res[0]) && make_resnet(v, "decoder.mid_block.resnets.1", v->mid_res[1]);
    ok = ok && make_attn(v, "decoder.mid_block.attentions.0", v->att);
    for (int i = 0; i < 4 && ok; ++i) {
        const std::string b = "decoder.up_blocks." + std::to_string(i);
        v->up_res[i].resize(v->cfg.layers_per_block + 1);
        for (size_t j = 0; j < v->up_res[i].size() && ok; ++j) ok = ok && make_resnet(v, b + ".resnets." + std::to_string(j), v->up_res[i][j]);
        if (i < 3) {
            ok = ok && make_conv(v, b + ".upsamplers.0.conv", v->up_samp[i]);
            VConv& uc = v->up_samp[i];
            if (ok && uc.taps == 9 && (uc.cout % 160 == 0 || uc.cout % 128 == 0) && uc.cin % 64 == 0) {
                std::vector<f16> sub((size_t)4 * uc.cout * 4 * uc.cin);
                conv_up_fold_pack_host(pack_conv(T(v, b + ".upsamplers.0.conv.weight")).data(), uc.cout, uc.cin, sub.data());
                uc.w_sub = upload(v, sub);
                ok = ok && uc.w_sub;
            }
        }
    }
    if (v->cfg.with_encoder && ok) {
        const int L = v->cfg.latent_channels;
        ok = make_conv_padded64(v, "encoder.conv_in", v->e_conv_in) && make_norm(v, "encoder.conv_norm_out", v->e_norm_out) &&
             make_resnet(v, "encoder.mid_block.resnets.0", v->e_mid_res[0]) && make_resnet(v, "encoder.mid_block.resnets.1", v->e_mid_res[1]) &&
             make_attn(v, "encoder.mid_block.attentions.0", v->e_att);
        for (int i = 0; i < 4 && ok; ++i) {
            const std::string b = "encoder.down_blocks." + std::to_string(i);
            v->e_down_res[i].resize(v->cfg.layers_per_block);
            for (size_t j = 0; j < v->e_down_res[i].size() && ok; ++j) ok = ok && make_resnet(v, b + ".resnets." + std::to_string(j), v->e_down_res[i][j]);
            if (i < 3) ok = ok && make_conv(v, b + ".downsamplers.0.conv", v->e_down_samp[i]);
        }
        // conv_out: with quant_conv every one of the 2L moments feeds the mean; without it only the mean half is needed
        const HostT& w = T(v, "encoder.conv_out.weight"); const HostT& bb = T(v, "encoder.conv_out.bias");
        const int rows = v->cfg.use_quant_conv ? 2 * L : L;
        HostT wt; wt.shape = {rows, w.shape[1], 3, 3}; wt.data.assign(w.data.begin(), w.data.begin() + (size_t)rows * w.shape[1] * 9);
        v->e_conv_out.cout = rows; v->e_conv_out.cin = (int)w.shape[1]; v->e_conv_out.taps = 9;
        v->e_conv_out.w = upload(v, pack_conv(wt)); v->e_conv_out.b = upload(v, std::vector<f16>(bb.data.begin(), bb.data.begin() + rows));
        ok = ok && v->e_conv_out.w && v->e_conv_out.b;
        if (v->cfg.use_quant_conv) {
            const HostT& qw = T(v, "quant_conv.weight"); const HostT& qb = T(v, "quant_conv.bias");
            v->q_w = upload(v, std::vector<f16>(qw.data.begin(), qw.data.begin() + (size_t)L * 2 * L));
            v->q_b = upload(v, std::vector<f16>(qb.data.begin(), qb.data.begin() + L));
            ok = ok && v->q_w && v->q_b;
        }
    }
    if (!ok) CS_FAIL(CS_E_HIP, "vae: weight upload failed (hipMalloc/hipMemcpy)");
    v->host.clear();
    v->finalized = true;
    return CS_OK;
}

size_t cs_vae_workspace_bytes(const CsVae* cv, int batch) {
    CsVae* v = const_cast<CsVae*>(cv);
    if (!v || !v->finalized || batch <= 0) return 0;
    run_decode(v, true, nullptr, batch, 1.f, 0.f, nullptr, 0, nullptr, 0, nullptr);
    return gn_ws_bytes(v, batch) + v->arena.peak + 4096;
}

double cs_vae_flops(const CsVae* cv, int batch) {
    CsVae* v = const_cast<CsVae*>(cv);
    if (!v || !v->finalized || batch <= 0) return 0;
    run_decode(v, true, nullptr, batch, 1.f, 0.f, nullptr, 0, nullptr, 0, nullptr);
    return v->dry_flops;
}

int cs_vae_decode(CsVae* v, const void* latents, int batch, float in_scale, float in_shift, void* images, int postprocess, void* workspace,
                  size_t workspace_bytes, void* stream) {
    if (!v) CS_FAIL(CS_E_ARG, "vae is NULL");
    if (!v->finalized) CS_FAIL(CS_E_STATE, "cs_vae_finalize has not been called");
    if (batch < 0) CS_FAIL(CS_E_ARG, "negative batch");
    if (batch == 0) return CS_OK;
    if (!latents || !images || !workspace) CS_FAIL(CS_E_ARG, "null pointer");
    return run_decode(v, false, (const f16*)latents, batch, in_scale, in_shift, (f16*)images, postprocess, (char*)workspace, workspace_bytes,
                      (hipStream_t)stream);
}

size_t cs_vae_encode_workspace_bytes(const CsVae* cv, int batch) {
    CsVae* v = const_cast<CsVae*>(cv);
    if (!v || !v->finalized || !v->cfg.with_encoder || batch <= 0) return 0;
    run_encode(v, true, nullptr, batch, 1.f, 0.f, nullptr, nullptr, 0, nullptr);
    return gn_ws_bytes(v, batch) + v->arena.peak + 4096;
}

int cs_vae_encode(CsVae* v, const void* images, int batch, float out_scale, float out_shift, void* latents, void* workspace, size_t workspace_bytes,
                  void* stream) {
    if (!v) CS_FAIL(CS_E_ARG, "vae is NULL");
    if (!v->finalized) CS_FAIL(CS_E_STATE, "cs_vae_finalize has not been called");
    if (!v->cfg.with_encoder) CS_FAIL(CS_E_STATE, "vae: created without the encoder (cfg.with_encoder = 0)");
    if (batch < 0) CS_FAIL(CS_E_ARG, "negative batch");
    if (batch == 0) return CS_OK;
    if (!images || !latents || !workspace) CS_FAIL(CS_E_ARG, "null pointer");
    return run_encode(v, false, (const f16*)images, batch, out_scale, out_shift, (f16*)latents, (char*)workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
