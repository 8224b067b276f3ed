// FLUX.1-Kontext DiT (FluxTransformer2DModel) forward executor on the HIP ops.
//
// Replaces the third-party call `transformer(hidden_states, timestep/1000, guidance, pooled_projections,
// encoder_hidden_states, txt_ids, img_ids)[0]` (edit_ppo/pipeline.py:1087-1097,
// edit_ppo/denoise_diffusion.py:135-144; diffusers git-main, un-vendored -> parity unpinned, see
// oracle/flux_oracle.py).  Architecture constants: SURVEY Appendix D (19 double-stream + 38 single-stream
// blocks, 24 heads x 128, adaLN-Zero modulation, RMSNorm on q/k, 3-axis RoPE, GELU-tanh MLP).
//
// Layout: token-major [B*tokens, C] 16-bit (bf16 by default).  The joint [context | image] sequence lives in
// ONE qkv buffer: the two streams' projection GEMMs write their row ranges through segment maps, so the
// torch.cat / split around attention never materialises; in the single-stream blocks attention output and
// the MLP activation are written side by side (ldc = 5 * hidden) so the cat before proj_out is free as well.
// All adaLN modulation linears of the 57 blocks are evaluated up front by one weight-streaming tiny-M kernel.
#include "ops.h"

#include <map>
#include <string>
#include <vector>
#include <algorithm>
#include <cstring>
#include <cmath>

namespace {

struct DevT { void* p = nullptr; std::vector<int64_t> shape; size_t elems = 0; };

struct Lin { u16* w = nullptr; u16* b = nullptr; int n = 0, k = 0; };

struct DoubleBlk {
    long mod_img = 0, mod_ctx = 0;       // offsets into the modulation vector (6 * D each)
    Lin qkv, add_qkv, out, add_out, ff1, ff2, cff1, cff2;
    u16 *nq = nullptr, *nk = nullptr, *naq = nullptr, *nak = nullptr;
};
struct SingleBlk { long mod = 0; Lin qkv_mlp, out; u16 *nq = nullptr, *nk = nullptr; };

struct Arena2 {
    char* base = nullptr; size_t cap = 0, top = 0, peak = 0; bool dry = false;
    void reset(char* b, size_t c, bool d) { base = b; cap = c; top = 0; peak = 0; dry = d; }
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (!dry && top + bytes > cap) return nullptr;
        void* p = base + top; top += bytes; peak = std::max(peak, top); return p;
    }
};

}  // namespace

struct CsFlux {
    CsFluxConfig cfg;
    int D = 0;   // hidden = heads * head_dim
    std::vector<std::string> names;
    std::map<std::string, std::vector<int64_t>> expect;
    std::map<std::string, DevT> raw;
    std::vector<void*> owned;
    bool finalized = false;
    Lin x_emb, c_emb, proj_out;
    Lin t1, t2, g1, g2, p1, p2;
    u16* mod_w = nullptr; u16* mod_b = nullptr; long mod_total = 0; long mod_final = 0;
    std::vector<DoubleBlk> dbl; std::vector<SingleBlk> sgl;
    Arena2 arena; double dry_flops = 0;
    int out_f32 = 0;                    // cs_flux_set_output_dtype: 1 = the forward's `out` is fp32 (split stream only: the head's two planes summed)
    int residual = CS_RESIDUAL_F16X2;   // cs_flux_set_residual_precision: the hidden-state stream as hi + lo planes of the model dtype (default) or one plane
};

namespace {

void expect(CsFlux* f, const std::string& n, std::vector<int64_t> s) { f->names.push_back(n); f->expect[n] = std::move(s); }
void expect_lin(CsFlux* f, const std::string& p, int n, int k, bool bias = true) { expect(f, p + ".weight", {n, k}); if (bias) expect(f, p + ".bias", {n}); }

void build_manifest(CsFlux* f) {
    const CsFluxConfig& c = f->cfg; const int D = f->D;
    expect_lin(f, "x_embedder", D, c.in_channels);
    expect_lin(f, "context_embedder", D, c.joint_attention_dim);
    expect_lin(f, "time_text_embed.timestep_embedder.linear_1", D, 256); expect_lin(f, "time_text_embed.timestep_embedder.linear_2", D, D);
    if (c.guidance_embeds) { expect_lin(f, "time_text_embed.guidance_embedder.linear_1", D, 256); expect_lin(f, "time_text_embed.guidance_embedder.linear_2", D, D); }
    expect_lin(f, "time_text_embed.text_embedder.linear_1", D, c.pooled_projection_dim); expect_lin(f, "time_text_embed.text_embedder.linear_2", D, D);
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string b = "transformer_blocks." + std::to_string(i);
        expect_lin(f, b + ".norm1.linear", 6 * D, D); expect_lin(f, b + ".norm1_context.linear", 6 * D, D);
        for (const char* q : {".attn.to_q", ".attn.to_k", ".attn.to_v", ".attn.add_q_proj", ".attn.add_k_proj", ".attn.add_v_proj", ".attn.to_out.0", ".attn.to_add_out"})
            expect_lin(f, b + q, D, D);
        for (const char* q : {".attn.norm_q.weight", ".attn.norm_k.weight", ".attn.norm_added_q.weight", ".attn.norm_added_k.weight"}) expect(f, b + q, {c.head_dim});
        expect_lin(f, b + ".ff.net.0.proj", 4 * D, D); expect_lin(f, b + ".ff.net.2", D, 4 * D);
        expect_lin(f, b + ".ff_context.net.0.proj", 4 * D, D); expect_lin(f, b + ".ff_context.net.2", D, 4 * D);
    }
    for (int i = 0; i < c.num_single_layers; ++i) {
        const std::string b = "single_transformer_blocks." + std::to_string(i);
        expect_lin(f, b + ".norm.linear", 3 * D, D);
        expect_lin(f, b + ".proj_mlp", 4 * D, D); expect_lin(f, b + ".proj_out", D, 5 * D);
        for (const char* q : {".attn.to_q", ".attn.to_k", ".attn.to_v"}) expect_lin(f, b + q, D, D);
        expect(f, b + ".attn.norm_q.weight", {c.head_dim}); expect(f, b + ".attn.norm_k.weight", {c.head_dim});
    }
    expect_lin(f, "norm_out.linear", 2 * D, D);
    expect_lin(f, "proj_out", c.in_channels, D);
}

u16* dev_alloc(CsFlux* f, size_t elems) {
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(elems * 2, 256)) != hipSuccess) return nullptr;
    f->owned.push_back(d);
    return (u16*)d;
}
const DevT& R(CsFlux* f, const std::string& n) { return f->raw.at(n); }

// concat rows of several [n_i, k] tensors (and their biases) into one packed Lin, zero padding rows to a multiple of 256
bool make_lin(CsFlux* f, std::initializer_list<std::string> prefixes, Lin& L) {
    int n = 0, k = 0;
    for (auto& p : prefixes) { const DevT& w = R(f, p + ".weight"); n += (int)w.shape[0]; k = (int)w.shape[1]; }
    const int npad = (n + 255) / 256 * 256;
    L.n = n; L.k = k;
    L.w = dev_alloc(f, (size_t)npad * k); L.b = dev_alloc(f, (size_t)npad);
    if (!L.w || !L.b) return false;
    if (hipMemset(L.w, 0, (size_t)npad * k * 2) != hipSuccess || hipMemset(L.b, 0, (size_t)npad * 2) != hipSuccess) return false;
    size_t ro = 0;
    for (auto& p : prefixes) {
        const DevT& w = R(f, p + ".weight"); const DevT& b = R(f, p + ".bias");
        if (hipMemcpy(L.w + ro * k, w.p, w.elems * 2, hipMemcpyDeviceToDevice) != hipSuccess) return false;
        if (hipMemcpy(L.b + ro, b.p, b.elems * 2, hipMemcpyDeviceToDevice) != hipSuccess) return false;
        ro += (size_t)w.shape[0];
    }
    return true;
}
u16* copy_vec(CsFlux* f, const std::string& n) {
    const DevT& t = R(f, n);
    u16* d = dev_alloc(f, t.elems);
    if (d && hipMemcpy(d, t.p, t.elems * 2, hipMemcpyDeviceToDevice) != hipSuccess) return nullptr;
    return d;
}

struct FRun {
    CsFlux* f; hipStream_t s; bool dry; int rc = CS_OK; int dt;
    void* attn_ws = nullptr; size_t attn_ws_bytes = 0;      // scratch of the attention's split-KV tail
    void* tail_ws = nullptr; size_t tail_ws_bytes = 0;      // scratch of the GEMMs' split-K tail
    void* alloc(size_t bytes) {
        void* p = f->arena.alloc(bytes);
        if (!p && rc == CS_OK) { cs_set_error("flux: workspace too small"); rc = CS_E_ARG; }
        return p ? p : reinterpret_cast<void*>((uintptr_t)1 << 41);      // poison base, never dereferenced (nothing is launched once rc is set)
    }
    static Gemm2Args gargs(const Lin& L, const void* a, long lda, int M, void* out, long ldc, int col_off, int act, const void* res,
                           const float* gate, long gate_stride, int rows_per_sample, int a_seg, int a_stride, long a_off, int c_seg, int c_stride, long c_off) {
        Gemm2Args g{};
        g.a = a; g.lda = lda; g.a_seg_rows = a_seg; g.a_seg_stride = a_stride; g.a_row_off = a_off;
        g.w = L.w; g.bias = L.b; g.M = M; g.N = L.n; g.K = L.k;
        g.out = out; g.res = res; g.ldc = ldc; g.c_col_off = col_off; g.c_seg_rows = c_seg; g.c_seg_stride = c_stride; g.c_row_off = c_off;
        g.gate = gate; g.gate_stride = gate_stride; g.rows_per_sample = rows_per_sample; g.act = act;
        return g;
    }
    void gemm(const Lin& L, const void* a, long lda, int M, void* out, long ldc, int col_off = 0, int act = 0, const void* res = nullptr,
              const float* gate = nullptr, long gate_stride = 0, int rows_per_sample = 0,
              int a_seg = 0, int a_stride = 0, long a_off = 0, int c_seg = 0, int c_stride = 0, long c_off = 0, void* lo = nullptr) {
        if (dry) { f->dry_flops += 2.0 * M * (double)L.n * L.k; return; }
        if (rc != CS_OK) return;
        Gemm2Args g = gargs(L, a, lda, M, out, ldc, col_off, act, res, gate, gate_stride, rows_per_sample, a_seg, a_stride, a_off, c_seg, c_stride, c_off);
        g.res_lo = lo; g.out_lo = lo;          // split residual stream: the lo plane is updated in place like the hi plane (out == res)
        g.dtype = dt; g.tail_ws = tail_ws; g.tail_ws_bytes = tail_ws_bytes;
        rc = launch_gemm2(g, s);
    }
    // the image-stream and text-stream linears of one stage in a single grouped launch
    void gemm_pair(Gemm2Args x, Gemm2Args y) {
        if (dry) { f->dry_flops += 2.0 * x.M * (double)x.N * x.K + 2.0 * y.M * (double)y.N * y.K; return; }
        if (rc != CS_OK) return;
        x.dtype = y.dtype = dt; x.tail_ws = tail_ws; x.tail_ws_bytes = tail_ws_bytes;
        rc = launch_gemm2_pair(x, y, s);
    }
    void small(const float* x, int Rr, int K, const u16* w, const u16* b, long N, float* out, int silu_in, int silu_out) {
        if (dry) { f->dry_flops += 2.0 * Rr * (double)N * K; return; }
        if (rc == CS_OK) rc = launch_small_linear(x, Rr, K, w, b, (int)N, out, silu_in, silu_out, dt, s);
    }
    void lnmod(const void* x, void* y, int M, int C, int rps, const float* shift, const float* scale, long stride, const void* x_lo = nullptr, void* y_lo = nullptr) {
        if (dry || rc != CS_OK) return;
        rc = launch_ln_modulate(x, y, M, C, rps, shift, scale, stride, 1e-6f, dt, s, x_lo, y_lo);
    }
    static Gemm2Args with_lo(Gemm2Args g, void* lo) { g.res_lo = lo; g.out_lo = lo; return g; }
    void attn(const u16* qkv, int B, int S, u16* out, long out_stride) {
        const CsFluxConfig& c = f->cfg; const int D = f->D;
        if (dry) { f->dry_flops += 4.0 * B * (double)S * S * D; return; }
        if (rc != CS_OK) return;
        AttnArgs a{};
        a.q = (const f16*)qkv; a.q_stride = 3 * D; a.k = (const f16*)(qkv + D); a.k_stride = 3 * D; a.v = (const f16*)(qkv + 2 * D); a.v_stride = 3 * D;
        a.out = (f16*)out; a.out_stride = (int)out_stride; a.B = B; a.H = c.num_heads; a.Nq = S; a.Nk = S; a.dh = c.head_dim;
        a.scale = 1.0f / sqrtf((float)c.head_dim); a.dtype = dt;
        a.split_ws = attn_ws; a.split_ws_bytes = attn_ws_bytes;
        rc = launch_attention(a, s);
    }
};

// hidden2 / I2: optional second block of image tokens per sample (the Kontext reference-image latents,
// edit_ppo/pipeline.py:1080 `cat([latents, image_latents], dim=1)`): the joint image sequence [hidden | hidden2] is never
// materialised -- the embedder reads both sources -- and only the I1 = I - I2 rows of `hidden` get an output row
// (`noise_pred[:, :latents.size(1)]`, edit_ppo/denoise_diffusion.py:145).
int flux_forward(CsFlux* f, bool dry, const void* hidden, int B, int I, const void* enc, int T, const float* pooled, const float* timestep,
                 const float* guidance, const float* rcos, const float* rsin, void* out, char* ws, size_t ws_bytes, hipStream_t s,
                 const void* hidden2 = nullptr, int I2 = 0) {
    const CsFluxConfig& c = f->cfg; const int D = f->D, S = T + I, H = c.num_heads, dh = c.head_dim;
    const int I1 = I - I2;                    // rows of `hidden` per sample = output rows per sample
    f->arena.reset(dry ? (char*)256 : ws, ws_bytes, dry); f->dry_flops = 0;
    FRun Rn{f, s, dry}; Rn.dt = c.dtype;
    const size_t e = 2;
    float* sin_t = (float*)Rn.alloc((size_t)B * 256 * 4); float* h1 = (float*)Rn.alloc((size_t)B * D * 4);
    float* emb_t = (float*)Rn.alloc((size_t)B * D * 4); float* emb_g = (float*)Rn.alloc((size_t)B * D * 4); float* emb_p = (float*)Rn.alloc((size_t)B * D * 4);
    float* temb = (float*)Rn.alloc((size_t)B * D * 4);
    float* mod = (float*)Rn.alloc((size_t)B * f->mod_total * 4);
    u16* img = (u16*)Rn.alloc((size_t)B * I * D * e); u16* ctx = (u16*)Rn.alloc((size_t)B * T * D * e);
    // Round 5, split residual stream: the hidden states of both streams (img / ctx, then the joint hs) carry a lo plane; every gated-residual epilogue adds hi + lo
    // in fp32 and writes both, the adaLN LayerNorms read hi + lo.  tools/sim_precision_flux.py: at full depth the bf16 STREAM is 1.16e-2 of the forward's 1.18e-2
    // relative error, everything else 2.5e-3.  The embedders' outputs start with lo = 0 (two of ~120 stream stores).
    const bool split = f->residual == CS_RESIDUAL_F16X2;
    u16* img_lo = split ? (u16*)Rn.alloc((size_t)B * I * D * e) : nullptr; u16* ctx_lo = split ? (u16*)Rn.alloc((size_t)B * T * D * e) : nullptr;
    u16* nimg = (u16*)Rn.alloc((size_t)B * I * D * e); u16* nctx = (u16*)Rn.alloc((size_t)B * T * D * e);
    u16* qkv = (u16*)Rn.alloc((size_t)B * S * 3 * D * e); u16* att = (u16*)Rn.alloc((size_t)B * S * D * e);
    u16* mlp = (u16*)Rn.alloc((size_t)B * I * 4 * D * e);            // double blocks: FF hidden of the image stream
    u16* cmlp = (u16*)Rn.alloc((size_t)B * T * 4 * D * e);           // ... and of the text stream (both FFs run in one grouped launch)
    u16* hs = (u16*)Rn.alloc((size_t)B * S * D * e); u16* nhs = (u16*)Rn.alloc((size_t)B * S * D * e);
    u16* hs_lo = split ? (u16*)Rn.alloc((size_t)B * S * D * e) : nullptr;
    u16* cat = (u16*)Rn.alloc((size_t)B * S * 5 * D * e);
    {   // long-K launches whose last round is partly empty: image + text FF2 (K = 4 D), single-stream proj_out (K = 5 D)
        const int tD = (D + 255) / 256;
        const size_t t1 = gemm2_tail_workspace_bytes(((B * I + 255) / 256 + (B * T + 255) / 256) * tD, 4 * D);
        const size_t t2 = gemm2_tail_workspace_bytes(((B * S + 255) / 256) * tD, 5 * D);
        const size_t t3 = gemm2_tail_workspace_bytes(((B * I + 255) / 256 + (B * T + 255) / 256) * 4 * tD, D);      // FF1 pair (K = D)
        const size_t t4 = gemm2_tail_workspace_bytes(((B * S + 255) / 256) * 4 * tD, D);                              // single-stream MLP
        const size_t t5 = gemm2_tail_workspace_bytes(((B * S + 255) / 256) * 3 * tD, D);                              // single-stream QKV
        const size_t t6 = gemm2_tail_workspace_bytes(((B * I + 255) / 256 + (B * T + 255) / 256) * 3 * tD, D);      // QKV pair
        const size_t t7 = gemm2_tail_workspace_bytes(((B * I + 255) / 256 + (B * T + 255) / 256) * tD, D);          // attention out pair
        Rn.tail_ws_bytes = std::max({t1, t2, t3, t4, t5, t6, t7});
        if (Rn.tail_ws_bytes) Rn.tail_ws = Rn.alloc(Rn.tail_ws_bytes);
    }
    Rn.attn_ws_bytes = attention_split_workspace_bytes(B, H, S, S, dh);
    if (Rn.attn_ws_bytes) Rn.attn_ws = Rn.alloc(Rn.attn_ws_bytes);
    if (Rn.rc != CS_OK) return Rn.rc;

    // device copies / memsets of the stream planes go through the run's error state: a failed one must not leave uninitialised workspace in the epilogues' adds
    auto hipok = [&](hipError_t e_, const char* what) {
        if (e_ != hipSuccess && Rn.rc == CS_OK) { cs_set_error("flux: %s failed: %s", what, hipGetErrorString(e_)); Rn.rc = CS_E_HIP; }
    };
    // ---- conditioning vector and all adaLN modulations -----------------------------------------------------
    if (!dry) {
        Rn.rc = launch_sinusoid_f32(timestep, 1000.0f, B, 256, sin_t, s);
    }
    Rn.small(sin_t, B, 256, f->t1.w, f->t1.b, D, h1, 0, 1); Rn.small(h1, B, D, f->t2.w, f->t2.b, D, emb_t, 0, 0);
    if (c.guidance_embeds) {
        if (!dry && Rn.rc == CS_OK) Rn.rc = launch_sinusoid_f32(guidance, 1000.0f, B, 256, sin_t, s);
        Rn.small(sin_t, B, 256, f->g1.w, f->g1.b, D, h1, 0, 1); Rn.small(h1, B, D, f->g2.w, f->g2.b, D, emb_g, 0, 0);
    }
    Rn.small(pooled, B, c.pooled_projection_dim, f->p1.w, f->p1.b, D, h1, 0, 1); Rn.small(h1, B, D, f->p2.w, f->p2.b, D, emb_p, 0, 0);
    if (!dry && Rn.rc == CS_OK) Rn.rc = launch_add3_f32(emb_t, c.guidance_embeds ? emb_g : nullptr, emb_p, temb, (long)B * D, s);
    Rn.small(temb, B, D, f->mod_w, f->mod_b, f->mod_total, mod, 1, 0);          // Linear(SiLU(temb)) for every block at once
    const long MS = f->mod_total;

    // ---- embedders ---------------------------------------------------------------------------------------------
    // split stream (round 6): the embedders' outputs ARE the initial hidden states -- rounding them to one plane of the model dtype put 2^-9 of relative error on the
    // whole stream before the first block.  They run the gated-residual epilogue's split form onto zeroed planes (no gate: hi + lo = the fp32 product + bias).
    const void* ires = nullptr; const void* cres = nullptr;
    if (split && !dry && Rn.rc == CS_OK) {
        hipok(hipMemsetAsync(img, 0, (size_t)B * I * D * e, s), "memset img"); hipok(hipMemsetAsync(ctx, 0, (size_t)B * T * D * e, s), "memset ctx");
        hipok(hipMemsetAsync(img_lo, 0, (size_t)B * I * D * e, s), "memset img_lo"); hipok(hipMemsetAsync(ctx_lo, 0, (size_t)B * T * D * e, s), "memset ctx_lo");
    }
    if (split) { ires = img; cres = ctx; }
    void* const ilo = split ? (void*)img_lo : nullptr; void* const clo = split ? (void*)ctx_lo : nullptr;
    if (I2 > 0)
        Rn.gemm_pair(FRun::with_lo(FRun::gargs(f->x_emb, hidden, c.in_channels, B * I1, img, D, 0, 0, ires, nullptr, 0, 0, 0, 0, 0, I1, I, 0), ilo),
                     FRun::with_lo(FRun::gargs(f->x_emb, hidden2, c.in_channels, B * I2, img, D, 0, 0, ires, nullptr, 0, 0, 0, 0, 0, I2, I, I1), ilo));
    else
        Rn.gemm(f->x_emb, hidden, c.in_channels, B * I, img, D, 0, 0, ires, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, ilo);
    Rn.gemm(f->c_emb, enc, c.joint_attention_dim, B * T, ctx, D, 0, 0, cres, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, clo);

    // ---- double-stream blocks -----------------------------------------------------------------------------------
    for (auto& k : f->dbl) {
        const float* mi = mod + k.mod_img; const float* mc = mod + k.mod_ctx;      // [shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp]
        Rn.lnmod(img, nimg, B * I, D, I, mi, mi + D, MS, img_lo); Rn.lnmod(ctx, nctx, B * T, D, T, mc, mc + D, MS, ctx_lo);
        Rn.gemm_pair(FRun::gargs(k.qkv, nimg, D, B * I, qkv, 3 * D, 0, 0, nullptr, nullptr, 0, 0, 0, 0, 0, I, S, T),
                     FRun::gargs(k.add_qkv, nctx, D, B * T, qkv, 3 * D, 0, 0, nullptr, nullptr, 0, 0, 0, 0, 0, T, S, 0));
        if (!dry && Rn.rc == CS_OK)
            Rn.rc = launch_qk_norm_rope(qkv, 3 * D, B * S, S, H, dh, 0, D, k.nq, k.nk, k.naq, k.nak, T, rcos, rsin, 1e-6f, c.dtype, s);
        Rn.attn(qkv, B, S, att, D);
        Rn.gemm_pair(FRun::with_lo(FRun::gargs(k.out, att, D, B * I, img, D, 0, 0, img, mi + 2 * D, MS, I, I, S, T, 0, 0, 0), img_lo),
                     FRun::with_lo(FRun::gargs(k.add_out, att, D, B * T, ctx, D, 0, 0, ctx, mc + 2 * D, MS, T, T, S, 0, 0, 0, 0), ctx_lo));
        Rn.lnmod(img, nimg, B * I, D, I, mi + 3 * D, mi + 4 * D, MS, img_lo); Rn.lnmod(ctx, nctx, B * T, D, T, mc + 3 * D, mc + 4 * D, MS, ctx_lo);
        Rn.gemm_pair(FRun::gargs(k.ff1, nimg, D, B * I, mlp, 4 * D, 0, 1, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0),
                     FRun::gargs(k.cff1, nctx, D, B * T, cmlp, 4 * D, 0, 1, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0));
        Rn.gemm_pair(FRun::with_lo(FRun::gargs(k.ff2, mlp, 4 * D, B * I, img, D, 0, 0, img, mi + 5 * D, MS, I, 0, 0, 0, 0, 0, 0), img_lo),
                     FRun::with_lo(FRun::gargs(k.cff2, cmlp, 4 * D, B * T, ctx, D, 0, 0, ctx, mc + 5 * D, MS, T, 0, 0, 0, 0, 0, 0), ctx_lo));
    }
    // ---- joint sequence [context | image] ---------------------------------------------------------------------------
    if (!dry && Rn.rc == CS_OK) {
        for (int b = 0; b < B; ++b) {
            hipok(hipMemcpyAsync(hs + ((size_t)b * S) * D, ctx + (size_t)b * T * D, (size_t)T * D * e, hipMemcpyDeviceToDevice, s), "copy ctx -> joint");
            hipok(hipMemcpyAsync(hs + ((size_t)b * S + T) * D, img + (size_t)b * I * D, (size_t)I * D * e, hipMemcpyDeviceToDevice, s), "copy img -> joint");
            if (split) {
                hipok(hipMemcpyAsync(hs_lo + ((size_t)b * S) * D, ctx_lo + (size_t)b * T * D, (size_t)T * D * e, hipMemcpyDeviceToDevice, s), "copy ctx_lo -> joint");
                hipok(hipMemcpyAsync(hs_lo + ((size_t)b * S + T) * D, img_lo + (size_t)b * I * D, (size_t)I * D * e, hipMemcpyDeviceToDevice, s), "copy img_lo -> joint");
            }
        }
    }
    // ---- single-stream blocks ---------------------------------------------------------------------------------------------
    for (auto& k : f->sgl) {
        const float* m = mod + k.mod;                                             // [shift, scale, gate]
        Rn.lnmod(hs, nhs, B * S, D, S, m, m + D, MS, hs_lo);
        // fused [q | k | v | mlp] projection: qkv part -> qkv buffer, GELU(mlp) part -> columns D.. of `cat`
        Lin lq = k.qkv_mlp; lq.n = 3 * D;
        Rn.gemm(lq, nhs, D, B * S, qkv, 3 * D);
        Lin lm; lm.w = k.qkv_mlp.w + (size_t)3 * D * D; lm.b = k.qkv_mlp.b + 3 * D; lm.n = 4 * D; lm.k = D;
        Rn.gemm(lm, nhs, D, B * S, cat, 5 * D, D, 1);
        if (!dry && Rn.rc == CS_OK)
            Rn.rc = launch_qk_norm_rope(qkv, 3 * D, B * S, S, H, dh, 0, D, k.nq, k.nk, nullptr, nullptr, 0, rcos, rsin, 1e-6f, c.dtype, s);
        Rn.attn(qkv, B, S, cat, 5 * D);
        Rn.gemm(k.out, cat, 5 * D, B * S, hs, D, 0, 0, hs, m + 2 * D, MS, S, 0, 0, 0, 0, 0, 0, hs_lo);
    }
    // ---- output head on the image tokens ----------------------------------------------------------------------------------
    if (!dry && Rn.rc == CS_OK)
        for (int b = 0; b < B; ++b) {
            hipok(hipMemcpyAsync(img + (size_t)b * I1 * D, hs + ((size_t)b * S + T) * D, (size_t)I1 * D * e, hipMemcpyDeviceToDevice, s), "copy joint -> img");
            if (split) hipok(hipMemcpyAsync(img_lo + (size_t)b * I1 * D, hs_lo + ((size_t)b * S + T) * D, (size_t)I1 * D * e, hipMemcpyDeviceToDevice, s), "copy joint_lo -> img_lo");
        }
    const float* mf = mod + f->mod_final;                                         // AdaLayerNormContinuous: [scale, shift]
    if (!split) {
        Rn.lnmod(img, nimg, B * I1, D, I1, mf + D, mf, MS, img_lo);
        Rn.gemm(f->proj_out, nimg, D, B * I1, out, c.in_channels);
        return Rn.rc;
    }
    // Split stream (round 6): the head's two tensors were the last places where the stream's value passed through ONE plane of the model dtype -- the modulated
    // LayerNorm output (proj_out's operand) and proj_out's result: 3.42e-3 per forward at full depth with them, 2.53e-3 without (tools/sim_precision_flux.py: the
    // emulation with exactly these two rounding points gives 3.41e-3).  The LayerNorm writes hi + lo; proj_out runs twice into one pair of planes -- W n_hi + b onto
    // zeroed planes, then W n_lo onto those (the gated-residual epilogue's split form, no gate) -- and `out` is the hi plane (= the rounding of the fp32-class value)
    // or, cs_flux_set_output_dtype(CS_F32), the sum of the two planes in fp32.  64 output columns: 0.1 % of a block's GEMM work.
    const size_t on = (size_t)B * I1 * c.in_channels;
    u16* nlo = (u16*)Rn.alloc((size_t)B * I1 * D * e);
    u16* oh = f->out_f32 ? (u16*)Rn.alloc(on * e) : (u16*)out;
    u16* ol = (u16*)Rn.alloc(on * e);
    Rn.lnmod(img, nimg, B * I1, D, I1, mf + D, mf, MS, img_lo, nlo);
    if (!dry && Rn.rc == CS_OK) { hipok(hipMemsetAsync(oh, 0, on * e, s), "memset head hi"); hipok(hipMemsetAsync(ol, 0, on * e, s), "memset head lo"); }
    Rn.gemm(f->proj_out, nimg, D, B * I1, oh, c.in_channels, 0, 0, oh, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, ol);
    Lin nb = f->proj_out; nb.b = nullptr;                                          // (the bias went in with the first product)
    Rn.gemm(nb, nlo, D, B * I1, oh, c.in_channels, 0, 0, oh, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, ol);
    if (f->out_f32 && !dry && Rn.rc == CS_OK) Rn.rc = launch_planes_to_f32(oh, ol, (float*)out, (long)on, c.dtype, s);
    return Rn.rc;
}

}  // namespace

extern "C" {

int cs_flux_create(const CsFluxConfig* cfg, CsFlux** out) {
    if (!cfg || !out) CS_FAIL(CS_E_ARG, "cfg/out is NULL");
    if (cfg->head_dim != 128) CS_FAIL(CS_E_UNSUPPORTED, "flux: head_dim must be 128");
    const int D = cfg->num_heads * cfg->head_dim;
    if (D % 256) CS_FAIL(CS_E_SHAPE, "flux: hidden size %d must be a multiple of 256", D);
    if (cfg->in_channels % 64 || cfg->joint_attention_dim % 64 || cfg->pooled_projection_dim % 8) CS_FAIL(CS_E_SHAPE, "flux: channel dims must be multiples of 64");
    if (cfg->axes_dims_rope[0] + cfg->axes_dims_rope[1] + cfg->axes_dims_rope[2] != cfg->head_dim) CS_FAIL(CS_E_SHAPE, "flux: rope axes must sum to head_dim");
    if (cfg->dtype != CS_BF16 && cfg->dtype != CS_F16) CS_FAIL(CS_E_DTYPE, "flux: dtype must be bf16 or f16");
    CsFlux* f = new CsFlux();
    f->cfg = *cfg; f->D = D;
    build_manifest(f);
    *out = f;
    return CS_OK;
}

void cs_flux_destroy(CsFlux* f) {
    if (!f) return;
    for (auto& kv : f->raw) if (kv.second.p) hipFree(kv.second.p);
    for (void* p : f->owned) hipFree(p);
    delete f;
}

int cs_flux_num_weights(const CsFlux* f) { return f ? (int)f->names.size() : 0; }
const char* cs_flux_weight_name(const CsFlux* f, int i, int64_t* shape2, int* ndim) {
    if (!f || i < 0 || i >= (int)f->names.size()) return nullptr;
    const auto& sh = f->expect.at(f->names[i]);
    if (ndim) *ndim = (int)sh.size();
    if (shape2) for (size_t k = 0; k < 2; ++k) shape2[k] = k < sh.size() ? sh[k] : 1;
    return f->names[i].c_str();
}

int cs_flux_set_weight(CsFlux* f, const char* name, const void* data, int on_device, const int64_t* shape, int ndim) {
    if (!f || !name || !data || !shape) CS_FAIL(CS_E_ARG, "null argument");
    if (f->finalized) CS_FAIL(CS_E_STATE, "weights are already packed");
    auto it = f->expect.find(name);
    if (it == f->expect.end()) CS_FAIL(CS_E_ARG, "unexpected tensor name '%s'", name);
    if ((int)it->second.size() != ndim) CS_FAIL(CS_E_SHAPE, "%s: rank %d, expected %zu", name, ndim, it->second.size());
    size_t n = 1;
    for (int k = 0; k < ndim; ++k) { if (shape[k] != it->second[k]) CS_FAIL(CS_E_SHAPE, "%s: dim %d is %lld, expected %lld", name, k, (long long)shape[k], (long long)it->second[k]); n *= (size_t)shape[k]; }
    DevT t; t.shape.assign(shape, shape + ndim); t.elems = n;
    CS_CHECK_HIP(hipMalloc(&t.p, std::max<size_t>(n * 2, 256)));
    CS_CHECK_HIP(hipMemcpy(t.p, data, n * 2, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    auto old = f->raw.find(name);
    if (old != f->raw.end() && old->second.p) hipFree(old->second.p);
    f->raw[name] = t;
    return CS_OK;
}

int cs_flux_finalize(CsFlux* f) {
    if (!f) CS_FAIL(CS_E_ARG, "null");
    if (f->finalized) return CS_OK;
    for (auto& n : f->names) if (!f->raw.count(n)) CS_FAIL(CS_E_STATE, "missing weight '%s'", n.c_str());
    const CsFluxConfig& c = f->cfg; const int D = f->D;
    bool ok = make_lin(f, {"x_embedder"}, f->x_emb) && make_lin(f, {"context_embedder"}, f->c_emb) && make_lin(f, {"proj_out"}, f->proj_out) &&
              make_lin(f, {"time_text_embed.timestep_embedder.linear_1"}, f->t1) && make_lin(f, {"time_text_embed.timestep_embedder.linear_2"}, f->t2) &&
              make_lin(f, {"time_text_embed.text_embedder.linear_1"}, f->p1) && make_lin(f, {"time_text_embed.text_embedder.linear_2"}, f->p2);
    if (c.guidance_embeds) ok = ok && make_lin(f, {"time_text_embed.guidance_embedder.linear_1"}, f->g1) && make_lin(f, {"time_text_embed.guidance_embedder.linear_2"}, f->g2);
    // modulation linears: one concatenated [mod_total, D] matrix
    std::vector<std::string> modp;
    f->dbl.resize(c.num_layers); f->sgl.resize(c.num_single_layers);
    long off = 0;
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string b = "transformer_blocks." + std::to_string(i);
        f->dbl[i].mod_img = off; modp.push_back(b + ".norm1.linear"); off += 6L * D;
        f->dbl[i].mod_ctx = off; modp.push_back(b + ".norm1_context.linear"); off += 6L * D;
    }
    for (int i = 0; i < c.num_single_layers; ++i) { f->sgl[i].mod = off; modp.push_back("single_transformer_blocks." + std::to_string(i) + ".norm.linear"); off += 3L * D; }
    f->mod_final = off; modp.push_back("norm_out.linear"); off += 2L * D;
    f->mod_total = off;
    f->mod_w = dev_alloc(f, (size_t)off * D); f->mod_b = dev_alloc(f, (size_t)off);
    ok = ok && f->mod_w && f->mod_b;
    long ro = 0;
    for (auto& p : modp) {
        if (!ok) break;
        const DevT& w = R(f, p + ".weight"); const DevT& b = R(f, p + ".bias");
        ok = hipMemcpy(f->mod_w + (size_t)ro * D, w.p, w.elems * 2, hipMemcpyDeviceToDevice) == hipSuccess &&
             hipMemcpy(f->mod_b + ro, b.p, b.elems * 2, hipMemcpyDeviceToDevice) == hipSuccess;
        ro += w.shape[0];
    }
    for (int i = 0; i < c.num_layers && ok; ++i) {
        const std::string b = "transformer_blocks." + std::to_string(i);
        DoubleBlk& k = f->dbl[i];
        ok = make_lin(f, {b + ".attn.to_q", b + ".attn.to_k", b + ".attn.to_v"}, k.qkv) &&
             make_lin(f, {b + ".attn.add_q_proj", b + ".attn.add_k_proj", b + ".attn.add_v_proj"}, k.add_qkv) &&
             make_lin(f, {b + ".attn.to_out.0"}, k.out) && make_lin(f, {b + ".attn.to_add_out"}, k.add_out) &&
             make_lin(f, {b + ".ff.net.0.proj"}, k.ff1) && make_lin(f, {b + ".ff.net.2"}, k.ff2) &&
             make_lin(f, {b + ".ff_context.net.0.proj"}, k.cff1) && make_lin(f, {b + ".ff_context.net.2"}, k.cff2);
        k.nq = copy_vec(f, b + ".attn.norm_q.weight"); k.nk = copy_vec(f, b + ".attn.norm_k.weight");
        k.naq = copy_vec(f, b + ".attn.norm_added_q.weight"); k.nak = copy_vec(f, b + ".attn.norm_added_k.weight");
        ok = ok && k.nq && k.nk && k.naq && k.nak;
    }
    for (int i = 0; i < c.num_single_layers && ok; ++i) {
        const std::string b = "single_transformer_blocks." + std::to_string(i);
        SingleBlk& k = f->sgl[i];
        ok = make_lin(f, {b + ".attn.to_q", b + ".attn.to_k", b + ".attn.to_v", b + ".proj_mlp"}, k.qkv_mlp) && make_lin(f, {b + ".proj_out"}, k.out);
        k.nq = copy_vec(f, b + ".attn.norm_q.weight"); k.nk = copy_vec(f, b + ".attn.norm_k.weight");
        ok = ok && k.nq && k.nk;
    }
    if (!ok) CS_FAIL(CS_E_HIP, "flux: weight packing failed (hipMalloc/hipMemcpy)");
    for (auto& kv : f->raw) if (kv.second.p) { hipFree(kv.second.p); kv.second.p = nullptr; }
    f->raw.clear();
    f->finalized = true;
    return CS_OK;
}

size_t cs_flux_workspace_bytes(const CsFlux* cf, int batch, int txt_len, int img_len) {
    CsFlux* f = const_cast<CsFlux*>(cf);
    if (!f || !f->finalized || batch <= 0) return 0;
    flux_forward(f, true, nullptr, batch, img_len, nullptr, txt_len, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
    return f->arena.peak + 4096;
}

double cs_flux_flops(const CsFlux* cf, int batch, int txt_len, int img_len) {
    CsFlux* f = const_cast<CsFlux*>(cf);
    if (!f || !f->finalized || batch <= 0) return 0;
    flux_forward(f, true, nullptr, batch, img_len, nullptr, txt_len, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
    return f->dry_flops;
}

int cs_flux_forward(CsFlux* f, const void* hidden_states, int batch, int img_len, const void* encoder_hidden_states, int txt_len,
                    const float* pooled_f32, const float* timestep, const float* guidance, const float* rope_cos, const float* rope_sin,
                    void* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!f) CS_FAIL(CS_E_ARG, "flux is NULL");
    if (!f->finalized) CS_FAIL(CS_E_STATE, "cs_flux_finalize has not been called");
    if (batch <= 0) return batch < 0 ? CS_E_SHAPE : CS_OK;
    if (!hidden_states || !encoder_hidden_states || !pooled_f32 || !timestep || !rope_cos || !rope_sin || !out || !workspace) CS_FAIL(CS_E_ARG, "null pointer");
    if (f->cfg.guidance_embeds && !guidance) CS_FAIL(CS_E_ARG, "guidance is required (guidance_embeds)");
    if (img_len <= 0 || txt_len <= 0) CS_FAIL(CS_E_SHAPE, "sequence lengths must be positive");
    if (f->out_f32 && f->residual != CS_RESIDUAL_F16X2) CS_FAIL(CS_E_STATE, "flux: an fp32 output (cs_flux_set_output_dtype) is the split stream's (CS_RESIDUAL_F16X2)");
    return flux_forward(f, false, hidden_states, batch, img_len, encoder_hidden_states, txt_len, pooled_f32, timestep, guidance, rope_cos, rope_sin, out,
                        (char*)workspace, workspace_bytes, (hipStream_t)stream);
}

int cs_flux_set_residual_precision(CsFlux* f, int mode) {
    if (!f) CS_FAIL(CS_E_ARG, "flux is NULL");
    if (mode != CS_RESIDUAL_F16 && mode != CS_RESIDUAL_F16X2) CS_FAIL(CS_E_ARG, "residual precision %d: CS_RESIDUAL_F16 (0, one plane) or CS_RESIDUAL_F16X2 (1, hi + lo planes)", mode);
    f->residual = mode;
    return CS_OK;
}
int cs_flux_get_residual_precision(const CsFlux* f) { return f ? f->residual : -1; }

int cs_flux_set_output_dtype(CsFlux* f, int dtype) {
    if (!f) CS_FAIL(CS_E_ARG, "flux is NULL");
    if (dtype != CS_F32 && dtype != f->cfg.dtype) CS_FAIL(CS_E_DTYPE, "flux output dtype %d: the model dtype (%d) or CS_F32", dtype, f->cfg.dtype);
    f->out_f32 = dtype == CS_F32;
    return CS_OK;
}
int cs_flux_get_output_dtype(const CsFlux* f) { return f ? (f->out_f32 ? CS_F32 : f->cfg.dtype) : -1; }

int cs_flux_forward_joint(CsFlux* f, const void* latents, int lat_len, const void* image_latents, int image_len, int batch,
                          const void* encoder_hidden_states, int txt_len, const float* pooled_f32, const float* timestep, const float* guidance,
                          const float* rope_cos, const float* rope_sin, void* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!f) CS_FAIL(CS_E_ARG, "flux is NULL");
    if (!f->finalized) CS_FAIL(CS_E_STATE, "cs_flux_finalize has not been called");
    if (batch <= 0) return batch < 0 ? CS_E_SHAPE : CS_OK;
    if (!latents || !encoder_hidden_states || !pooled_f32 || !timestep || !rope_cos || !rope_sin || !out || !workspace) CS_FAIL(CS_E_ARG, "null pointer");
    if (f->out_f32 && f->residual != CS_RESIDUAL_F16X2) CS_FAIL(CS_E_STATE, "flux: an fp32 output (cs_flux_set_output_dtype) is the split stream's (CS_RESIDUAL_F16X2)");
    if (f->cfg.guidance_embeds && !guidance) CS_FAIL(CS_E_ARG, "guidance is required (guidance_embeds)");
    if (lat_len <= 0 || txt_len <= 0 || image_len < 0) CS_FAIL(CS_E_SHAPE, "sequence lengths must be positive");
    if (image_len > 0 && !image_latents) CS_FAIL(CS_E_ARG, "image_latents is NULL but image_len > 0");
    return flux_forward(f, false, latents, batch, lat_len + image_len, encoder_hidden_states, txt_len, pooled_f32, timestep, guidance, rope_cos,
                        rope_sin, out, (char*)workspace, workspace_bytes, (hipStream_t)stream, image_len > 0 ? image_latents : nullptr, image_len);
}

}  // extern "C"
