// C ABI wrappers of the op-level launchers (include/consolver_hip_ops.h)
#include "ops.h"
#include "../../include/consolver_hip_ops.h"
#include <cstring>

int debug_trace_read(void* dst, size_t bytes);
int debug_attn_trace_read(void* dst, size_t bytes);

#include <mutex>
TuneSet g_tune;
thread_local const TuneSet* t_tune = nullptr;
static std::mutex g_tune_write_mutex;      // writers of the process-wide set (cs_set_tuning / cs_reset_tuning); per-handle overrides never write it (ops.h, TuneSet)

// every kernel-selection knob: name, field, accepted range (or the two-value set {lo, hi} when `pair`); the defaults are TuneSet's member initialisers
struct TuneKnob { const char* key; int TuneSet::* var; int lo, hi; bool pair; };
static const TuneKnob* tune_knobs(int* n) {
    static const TuneKnob k[] = {
        {"conv_halo", &TuneSet::halo, 0, 4, false},     {"gemm_lw", &TuneSet::gemm_lw, 0, 1, false},      {"gemm2_w8", &TuneSet::gemm2_w8, 0, 1, false},
        {"attn_lw", &TuneSet::attn_lw, 0, 2, false},    {"gemm_w8", &TuneSet::gemm_w8, 0, 1, false},      {"conv_lw", &TuneSet::conv_lw, 0, 3, false},
        {"gemm_big", &TuneSet::biggemm, 0, 3, false},   {"debug", &TuneSet::debug, 0, 0x7fffffff, false}, {"gemm_gm", &TuneSet::gemm_gm, -1, 64, false},
        {"gn_fuse", &TuneSet::gn_fuse, 0, 1, false},    {"xattn_fused", &TuneSet::xattn_fused, 0, 1, false}, {"cfg_share", &TuneSet::cfg_share, 0, 1, false},
        {"gemm2_prio", &TuneSet::gemm2_prio, -1, 1, false}, {"attn_prio", &TuneSet::attn_prio, -1, 1, false}, {"attn_qt40", &TuneSet::attn_qt40, 2, 4, true},
        {"x2_split_a", &TuneSet::x2_split_a, 0, 3, false}, {"x2_sc_skip", &TuneSet::x2_sc_skip, 0, 0xffff, false}, {"ln_fold", &TuneSet::ln_fold, 0, 1, false}, {"xcd_grid", &TuneSet::xcd_grid, 0, 1, false}, {"epi_fast", &TuneSet::epi_fast, 0, 3, false}, {"lo8", &TuneSet::lo8, 0, 1, false},
        {"conv_in_mfma", &TuneSet::conv_in_mfma, 0, 1, false}, {"xattn_tile", &TuneSet::xattn_tile, 64, 128, true},
        {"conv_out_mfma", &TuneSet::conv_out_mfma, 0, 1, false},
        {"up_fold", &TuneSet::up_fold, 0, 2, false},
        {"head_x2", &TuneSet::head_x2, 0, 1, false},
    };
    *n = (int)(sizeof(k) / sizeof(k[0]));
    return k;
}

int tune_apply(TuneSet& set, const char* key, int value) {
    if (!key) CS_FAIL(CS_E_ARG, "key is NULL");
    int n; const TuneKnob* k = tune_knobs(&n);
    for (int i = 0; i < n; ++i)
        if (!strcmp(key, k[i].key)) {
            const bool ok = k[i].pair ? (value == k[i].lo || value == k[i].hi) : (value >= k[i].lo && value <= k[i].hi);
            if (!ok) CS_FAIL(CS_E_ARG, "tuning key '%s': value %d outside %s%d%s%d%s", key, value, k[i].pair ? "{" : "[", k[i].lo, k[i].pair ? ", " : " .. ", k[i].hi, k[i].pair ? "}" : "]");
            set.*(k[i].var) = value;
            return CS_OK;
        }
    CS_FAIL(CS_E_ARG, "unknown tuning key '%s'", key);
}

extern "C" {

int cs_set_tuning(const char* key, int value) {
    std::lock_guard<std::mutex> lock(g_tune_write_mutex);
    return tune_apply(g_tune, key, value);
}

// the value the CALLING thread's launches would see (the process-wide one outside a per-handle forward)
int cs_get_tuning(const char* key, int* value) {
    if (!key || !value) CS_FAIL(CS_E_ARG, "key / value is NULL");
    int n; const TuneKnob* k = tune_knobs(&n);
    for (int i = 0; i < n; ++i)
        if (!strcmp(key, k[i].key)) { *value = tune().*(k[i].var); return CS_OK; }
    CS_FAIL(CS_E_ARG, "unknown tuning key '%s'", key);
}

int cs_reset_tuning(void) {
    std::lock_guard<std::mutex> lock(g_tune_write_mutex);
    g_tune = TuneSet();
    return CS_OK;
}

int cs_op_conv2d(const void* x0, int c0, const void* x1, int c1, int B, int Hi, int Wi, int taps, int stride, int upsample,
                 const void* w, const void* bias, int N, const void* temb, int temb_stride, const void* res, void* out,
                 void* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    IgemmArgs a{};
    a.a0 = (const f16*)x0; a.a1 = (const f16*)x1; a.c0 = c0; a.c1 = c1; a.B = B; a.Hi = Hi; a.Wi = Wi;
    if (stride != 1 && stride != 2) CS_FAIL(CS_E_ARG, "conv2d: stride must be 1 or 2");
    a.Ho = upsample ? 2 * Hi : (stride == 2 ? Hi / 2 : Hi);
    a.Wo = upsample ? 2 * Wi : (stride == 2 ? Wi / 2 : Wi);
    a.taps = taps; a.stride = stride; a.upsample = upsample; a.N = N; a.w = (const f16*)w; a.bias = (const f16*)bias;
    a.temb = (const f16*)temb; a.temb_stride = temb_stride; a.res = (const f16*)res; a.out = (f16*)out;
    a.splitk_ws = (float*)splitk_ws; a.splitk_ws_bytes = splitk_ws_bytes;
    return launch_igemm(a, (hipStream_t)stream);
}

int cs_op_conv2d_gn(const void* x0, int c0, const void* x1, int c1, int B, int Hi, int Wi, int taps, int stride, int upsample,
                    const void* w, const void* bias, int N, const void* temb, int temb_stride, const void* res, void* out,
                    void* splitk_ws, size_t splitk_ws_bytes, float* gn_stats, void* stream) {
    IgemmArgs a{};
    a.a0 = (const f16*)x0; a.a1 = (const f16*)x1; a.c0 = c0; a.c1 = c1; a.B = B; a.Hi = Hi; a.Wi = Wi;
    if (stride != 1 && stride != 2) CS_FAIL(CS_E_ARG, "conv2d: stride must be 1 or 2");
    a.Ho = upsample ? 2 * Hi : (stride == 2 ? Hi / 2 : Hi);
    a.Wo = upsample ? 2 * Wi : (stride == 2 ? Wi / 2 : Wi);
    a.taps = taps; a.stride = stride; a.upsample = upsample; a.N = N; a.w = (const f16*)w; a.bias = (const f16*)bias;
    a.temb = (const f16*)temb; a.temb_stride = temb_stride; a.res = (const f16*)res; a.out = (f16*)out;
    a.splitk_ws = (float*)splitk_ws; a.splitk_ws_bytes = splitk_ws_bytes; a.gn_stats = gn_stats;
    if (gn_stats && (a.Ho * a.Wo) % 64) CS_FAIL(CS_E_SHAPE, "conv2d_gn: Ho * Wo = %d must be a multiple of 64", a.Ho * a.Wo);
    return launch_igemm(a, (hipStream_t)stream);
}

int cs_op_conv_up_fold_pack(const void* w, int N, int Cin, void* out) {
    if (!w || !out || N <= 0 || Cin <= 0) CS_FAIL(CS_E_ARG, "conv_up_fold_pack: w [N][9 Cin], out [4][N][4 Cin] (host pointers)");
    conv_up_fold_pack_host((const f16*)w, N, Cin, (f16*)out);
    return CS_OK;
}

int cs_op_conv_up_sub(const void* x, int Cin, int B, int Hi, int Wi, const void* w, const void* w_sub, const void* bias, int N, void* out, float* gn_stats,
                      void* stream) {
    IgemmArgs a{};
    a.a0 = (const f16*)x; a.c0 = Cin; a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ho = 2 * Hi; a.Wo = 2 * Wi; a.taps = 9; a.stride = 1; a.upsample = 1; a.N = N;
    a.w = (const f16*)w; a.w_up_sub = (const f16*)w_sub; a.bias = (const f16*)bias; a.out = (f16*)out; a.gn_stats = gn_stats;
    if (!w_sub) CS_FAIL(CS_E_ARG, "conv_up_sub: w_sub (cs_op_conv_up_fold_pack) required");
    const bool f1 = Hi % 16 == 0 && Wi % 16 == 0 && (N % 160 == 0 || N % 128 == 0), f2 = Hi == 8 && Wi == 8 && N % 160 == 0;
    if (!(f1 || f2) || Cin % 64)
        CS_FAIL(CS_E_SHAPE, "conv_up_sub: input a multiple of 16 x 16 with N %% 160 == 0 or N %% 128 == 0, or 8 x 8 with N %% 160 == 0; Cin %% 64 == 0 (got %d x %d, N %d, Cin %d)", Hi, Wi, N, Cin);
    return launch_igemm(a, (hipStream_t)stream);
}

int cs_op_linear(const void* x, int M, int K, const void* w, const void* bias, int N, const void* res, void* out, int geglu, void* stream) {
    IgemmArgs a{};
    a.a0 = (const f16*)x; a.c0 = K; a.B = 1; a.Hi = M; a.Wi = 1; a.Ho = M; a.Wo = 1; a.taps = 1; a.stride = 1; a.N = N;
    a.w = (const f16*)w; a.bias = (const f16*)bias; a.res = (const f16*)res; a.out = (f16*)out; a.geglu = geglu;
    if (M <= 0) return M < 0 ? CS_E_SHAPE : CS_OK;
    return launch_igemm(a, (hipStream_t)stream);
}

int cs_op_xattn_block(const void* h, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* wq, const void* kv, int Nk,
                      const void* wo, const void* bo, int M, int HW, int C, int heads, float scale, void* out, void* stream) {
    XattnArgs a{};
    a.h = (const f16*)h; a.out = (f16*)out; a.ln_g = (const f16*)ln_gamma; a.ln_b = (const f16*)ln_beta; a.ln_eps = ln_eps;
    a.wq = (const f16*)wq; a.wo = (const f16*)wo; a.bo = (const f16*)bo; a.kv = (const f16*)kv;
    a.M = M; a.HW = HW; a.Nk = Nk; a.C = C; a.heads = heads; a.scale = scale;
    return launch_xattn_block(a, (hipStream_t)stream);
}

int cs_op_conv2d_x2(const void* x0, const void* x0_lo, int c0, const void* x1, const void* x1_lo, int c1, int B, int Hi, int Wi, int taps, int stride,
                    int upsample, const void* w, const void* bias, int N, const void* temb, int temb_stride, const void* res, const void* res_lo,
                    void* out, void* out_lo, float* row_stats, int* row_groups, void* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    IgemmArgs a{};
    a.a0 = (const f16*)x0; a.a1 = (const f16*)x1; a.c0 = c0; a.c1 = c1; a.B = B; a.Hi = Hi; a.Wi = Wi;
    a.a0_lo = (const f16*)x0_lo; a.a1_lo = (const f16*)x1_lo; a.row_stats = row_stats; a.row_stats_groups = row_groups;
    if (stride != 1 && stride != 2) CS_FAIL(CS_E_ARG, "conv2d: stride must be 1 or 2");
    if (res_lo && !res) CS_FAIL(CS_E_ARG, "conv2d_x2: res_lo without res");
    a.Ho = upsample ? 2 * Hi : (stride == 2 ? Hi / 2 : Hi);
    a.Wo = upsample ? 2 * Wi : (stride == 2 ? Wi / 2 : Wi);
    a.taps = taps; a.stride = stride; a.upsample = upsample; a.N = N; a.w = (const f16*)w; a.bias = (const f16*)bias;
    a.temb = (const f16*)temb; a.temb_stride = temb_stride; a.res = (const f16*)res; a.out = (f16*)out;
    a.res_lo = (const f16*)res_lo; a.out_lo = (f16*)out_lo;
    a.splitk_ws = (float*)splitk_ws; a.splitk_ws_bytes = splitk_ws_bytes;
    return launch_igemm(a, (hipStream_t)stream);
}

int cs_op_linear_x2(const void* x, const void* x_lo, int M, int K, const void* w, const void* bias, int N, const void* res, const void* res_lo,
                    void* out, void* out_lo, float* row_stats, int* row_groups, void* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    IgemmArgs a{};
    a.a0_lo = (const f16*)x_lo; a.row_stats = row_stats; a.row_stats_groups = row_groups;
    if (res_lo && !res) CS_FAIL(CS_E_ARG, "linear_x2: res_lo without res");
    a.a0 = (const f16*)x; a.c0 = K; a.B = 1; a.Hi = M; a.Wi = 1; a.Ho = M; a.Wo = 1; a.taps = 1; a.stride = 1; a.N = N;
    a.w = (const f16*)w; a.bias = (const f16*)bias; a.res = (const f16*)res; a.out = (f16*)out;
    a.res_lo = (const f16*)res_lo; a.out_lo = (f16*)out_lo;
    a.splitk_ws = (float*)splitk_ws; a.splitk_ws_bytes = splitk_ws_bytes;
    if (M <= 0) return M < 0 ? CS_E_SHAPE : CS_OK;
    return launch_igemm(a, (hipStream_t)stream);
}

int cs_op_linear_lo8(const void* x, int M, int K, const void* w, const void* bias, int N, const void* res, const void* res_lo8, void* out, void* out_lo8,
                     float* row_stats, int* row_groups, void* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    IgemmArgs a{};
    a.row_stats = row_stats; a.row_stats_groups = row_groups;
    if (res_lo8 && !res) CS_FAIL(CS_E_ARG, "linear_lo8: res_lo8 without res");
    a.a0 = (const f16*)x; a.c0 = K; a.B = 1; a.Hi = M; a.Wi = 1; a.Ho = M; a.Wo = 1; a.taps = 1; a.stride = 1; a.N = N;
    a.w = (const f16*)w; a.bias = (const f16*)bias; a.res = (const f16*)res; a.out = (f16*)out;
    a.res_lo = (const f16*)res_lo8; a.out_lo = (f16*)out_lo8; a.lo8 = 1;
    a.splitk_ws = (float*)splitk_ws; a.splitk_ws_bytes = splitk_ws_bytes;
    if (M <= 0) return M < 0 ? CS_E_SHAPE : CS_OK;
    return launch_igemm(a, (hipStream_t)stream);
}

int cs_op_xattn_block_lo8(const void* h, const void* h_lo8, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* wq, const void* kv,
                          int Nk, const void* wo, const void* bo, int M, int HW, int C, int heads, float scale, void* out, void* out_lo8, float* row_stats,
                          void* stream) {
    XattnArgs a{};
    a.row_stats = row_stats;
    a.h = (const f16*)h; a.out = (f16*)out; a.h_lo = (const f16*)h_lo8; a.out_lo = (f16*)out_lo8; a.lo8 = 1;
    a.ln_g = (const f16*)ln_gamma; a.ln_b = (const f16*)ln_beta; a.ln_eps = ln_eps;
    a.wq = (const f16*)wq; a.wo = (const f16*)wo; a.bo = (const f16*)bo; a.kv = (const f16*)kv;
    a.M = M; a.HW = HW; a.Nk = Nk; a.C = C; a.heads = heads; a.scale = scale;
    return launch_xattn_block(a, (hipStream_t)stream);
}

int cs_op_row_stats_lo8(const void* x, const void* x_lo8, int M, int C, float* stats, void* stream) {
    return launch_row_stats((const f16*)x, (const f16*)x_lo8, M, C, stats, (hipStream_t)stream, 1);
}

int cs_op_ln_dc_ratio(const float* row_stats, int M, int groups, int C, float eps, float* sum, void* stream) {
    return launch_ln_dc_ratio(row_stats, M, groups, C, eps, sum, (hipStream_t)stream);
}

int cs_op_conv_out(const void* x, int B, int Cin, int H, int W, const void* w, const void* bias, int Cout, void* out, int postprocess, void* stream) {
    if (postprocess && Cout != 3) CS_FAIL(CS_E_ARG, "conv_out: postprocess is the image head's (Cout 3)");
    if (Cout == 3) return launch_conv_out3((const f16*)x, B, Cin, H, W, (const f16*)w, (const f16*)bias, (f16*)out, postprocess, (hipStream_t)stream);
    if (Cout == 4) return launch_conv_out((const f16*)x, B, Cin, H, W, (const f16*)w, (const f16*)bias, Cout, (f16*)out, (hipStream_t)stream);
    return launch_conv_out_small((const f16*)x, B, Cin, H, W, (const f16*)w, (const f16*)bias, Cout, (f16*)out, (hipStream_t)stream);
}

int cs_op_group_norm_x2(const void* x0, const void* x0_lo, int c0, const void* x1, const void* x1_lo, int c1, int B, int HW, int groups,
                        float eps, int silu, const void* gamma, const void* beta, void* workspace, void* out, void* stream) {
    GroupNormArgs a{};
    a.x0 = (const f16*)x0; a.x1 = (const f16*)x1; a.c0 = c0; a.c1 = c1; a.B = B; a.HW = HW; a.groups = groups; a.eps = eps; a.silu = silu;
    a.gamma = (const f16*)gamma; a.beta = (const f16*)beta; a.partial = (float*)workspace; a.out = (f16*)out;
    a.x0_lo = (const f16*)x0_lo; a.x1_lo = (const f16*)x1_lo;
    return launch_group_norm(a, (hipStream_t)stream);
}

int cs_op_layer_norm_x2(const void* x, const void* x_lo, const void* gamma, const void* beta, void* out, int M, int C, float eps, void* stream) {
    return launch_layer_norm((const f16*)x, (const f16*)gamma, (const f16*)beta, (f16*)out, M, C, eps, (hipStream_t)stream, (const f16*)x_lo);
}

int cs_op_xattn_block_x2(const void* h, const void* h_lo, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* wq, const void* kv,
                         int Nk, const void* wo, const void* bo, int M, int HW, int C, int heads, float scale, void* out, void* out_lo, float* row_stats,
                         void* stream) {
    XattnArgs a{};
    a.row_stats = row_stats;
    a.h = (const f16*)h; a.out = (f16*)out; a.h_lo = (const f16*)h_lo; a.out_lo = (f16*)out_lo;
    a.ln_g = (const f16*)ln_gamma; a.ln_b = (const f16*)ln_beta; a.ln_eps = ln_eps;
    a.wq = (const f16*)wq; a.wo = (const f16*)wo; a.bo = (const f16*)bo; a.kv = (const f16*)kv;
    a.M = M; a.HW = HW; a.Nk = Nk; a.C = C; a.heads = heads; a.scale = scale;
    return launch_xattn_block(a, (hipStream_t)stream);
}

int cs_op_ln_fold_pack(const void* w_host, const void* bias_host, const void* gamma_host, const void* beta_host, int N, int K, void* w_out_host,
                       float* s_out_host, float* b_out_host) {
    if (!w_host || !gamma_host || !beta_host || !w_out_host || !s_out_host || !b_out_host || N <= 0 || K <= 0) CS_FAIL(CS_E_ARG, "ln_fold_pack: bad arguments");
    ln_fold_pack_host((const f16*)w_host, (const f16*)bias_host, (const f16*)gamma_host, (const f16*)beta_host, N, K, (f16*)w_out_host, s_out_host, b_out_host);
    return CS_OK;
}

int cs_op_linear_ln(const void* x, int M, int K, const void* w_folded, const float* ln_s, const float* ln_b, int N, const float* row_stats, int groups,
                    float eps, void* out, int geglu, void* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    IgemmArgs a{};
    a.a0 = (const f16*)x; a.c0 = K; a.B = 1; a.Hi = M; a.Wi = 1; a.Ho = M; a.Wo = 1; a.taps = 1; a.stride = 1; a.N = N;
    a.w = (const f16*)w_folded; a.out = (f16*)out; a.geglu = geglu;
    a.ln_stats = row_stats; a.ln_groups = groups; a.ln_eps = eps; a.ln_s = ln_s; a.ln_b = ln_b;
    a.splitk_ws = (float*)splitk_ws; a.splitk_ws_bytes = splitk_ws_bytes;
    if (!row_stats) CS_FAIL(CS_E_ARG, "linear_ln: row_stats required");
    if (M <= 0) return M < 0 ? CS_E_SHAPE : CS_OK;
    return launch_igemm(a, (hipStream_t)stream);
}

int cs_op_row_stats(const void* x, const void* x_lo, int M, int C, float* stats, void* stream) {
    return launch_row_stats((const f16*)x, (const f16*)x_lo, M, C, stats, (hipStream_t)stream);
}

int cs_op_geglu_pack(const void* w_host, const void* b_host, int Hd, int K, void* w_out_host, void* b_out_host) {
    if (!w_host || !w_out_host || Hd % 16) CS_FAIL(CS_E_ARG, "geglu_pack: bad arguments");
    const f16* w = (const f16*)w_host; const f16* b = (const f16*)b_host; f16* wo = (f16*)w_out_host; f16* bo = (f16*)b_out_host;
    for (int P = 0; P < Hd / 16; ++P)
        for (int i = 0; i < 16; ++i) {
            memcpy(wo + (size_t)(32 * P + i) * K, w + (size_t)(16 * P + i) * K, (size_t)K * sizeof(f16));
            memcpy(wo + (size_t)(32 * P + 16 + i) * K, w + (size_t)(Hd + 16 * P + i) * K, (size_t)K * sizeof(f16));
            if (b && bo) { bo[32 * P + i] = b[16 * P + i]; bo[32 * P + 16 + i] = b[Hd + 16 * P + i]; }
        }
    return CS_OK;
}

int cs_op_attention(const void* q, int q_stride, const void* k, int k_stride, const void* v, int v_stride, void* out, int out_stride,
                    int B, int H, int Nq, int Nk, int dh, float scale, void* stream) {
    AttnArgs a{};
    a.q = (const f16*)q; a.q_stride = q_stride; a.k = (const f16*)k; a.k_stride = k_stride; a.v = (const f16*)v; a.v_stride = v_stride;
    a.out = (f16*)out; a.out_stride = out_stride; a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.dh = dh; a.scale = scale;
    return launch_attention(a, (hipStream_t)stream);
}

int cs_op_gemm2(const void* x, int M, int K, const void* w, const void* bias, int N, const void* res, const float* gate, long gate_stride,
                int rows_per_sample, int act, void* out, long ldc, int col_off, int dtype, void* stream) {
    Gemm2Args g{};
    g.a = x; g.lda = K; g.w = w; g.bias = bias; g.M = M; g.N = N; g.K = K; g.out = out; g.res = res; g.ldc = ldc ? ldc : N; g.c_col_off = col_off;
    g.gate = gate; g.gate_stride = gate_stride; g.rows_per_sample = rows_per_sample; g.act = act; g.dtype = dtype;
    return launch_gemm2(g, (hipStream_t)stream);
}

int cs_op_gemm2_x2(const void* x, int M, int K, const void* w, const void* bias, int N, const void* res, const void* res_lo, const float* gate, long gate_stride,
                   int rows_per_sample, void* out, void* out_lo, int dtype, void* tail_ws, size_t tail_ws_bytes, void* stream) {
    Gemm2Args g{};
    g.a = x; g.lda = K; g.w = w; g.bias = bias; g.M = M; g.N = N; g.K = K; g.out = out; g.res = res; g.ldc = N; g.res_lo = res_lo; g.out_lo = out_lo;
    g.gate = gate; g.gate_stride = gate_stride; g.rows_per_sample = rows_per_sample; g.act = 0; g.dtype = dtype; g.tail_ws = tail_ws; g.tail_ws_bytes = tail_ws_bytes;
    return launch_gemm2(g, (hipStream_t)stream);
}
int cs_op_ln_modulate_x2(const void* x, const void* x_lo, void* y, int M, int C, int rows_per_sample, const float* shift, const float* scale, long mod_stride, float eps,
                         int dtype, void* stream) {
    return launch_ln_modulate(x, y, M, C, rows_per_sample, shift, scale, mod_stride, eps, dtype, (hipStream_t)stream, x_lo);
}

size_t cs_op_attention_workspace(int B, int H, int Nq, int Nk, int dh) { return attention_split_workspace_bytes(B, H, Nq, Nk, dh); }
int cs_op_attention_ws(const void* q, int q_stride, const void* k, int k_stride, const void* v, int v_stride, void* out, int out_stride,
                       int B, int H, int Nq, int Nk, int dh, float scale, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    AttnArgs a{};
    a.q = (const f16*)q; a.q_stride = q_stride; a.k = (const f16*)k; a.k_stride = k_stride; a.v = (const f16*)v; a.v_stride = v_stride;
    a.out = (f16*)out; a.out_stride = out_stride; a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.dh = dh; a.scale = scale; a.dtype = dtype;
    a.split_ws = workspace; a.split_ws_bytes = workspace_bytes;
    return launch_attention(a, (hipStream_t)stream);
}

static Gemm2Args g2_from(const CsGemm2Problem& q, int dtype) {
    Gemm2Args g{};
    g.a = q.x; g.lda = q.K; g.w = q.w; g.bias = q.bias; g.M = q.M; g.N = q.N; g.K = q.K; g.out = q.out; g.res = q.res; g.ldc = q.ldc ? q.ldc : q.N;
    g.c_col_off = q.col_off; g.gate = q.gate; g.gate_stride = q.gate_stride; g.rows_per_sample = q.rows_per_sample; g.act = q.act; g.dtype = dtype;
    return g;
}
int cs_op_gemm2_pair(const CsGemm2Problem* a, const CsGemm2Problem* b, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a) CS_FAIL(CS_E_ARG, "gemm2 pair: null problem");
    Gemm2Args ga = g2_from(*a, dtype);
    ga.tail_ws = workspace; ga.tail_ws_bytes = workspace_bytes;
    if (!b) return launch_gemm2(ga, (hipStream_t)stream);
    return launch_gemm2_pair(ga, g2_from(*b, dtype), (hipStream_t)stream);
}
size_t cs_op_gemm2_workspace(int tiles, int K) { return gemm2_tail_workspace_bytes(tiles, K); }

int cs_op_attention_ex(const void* q, int q_stride, const void* k, int k_stride, const void* v, int v_stride, void* out, int out_stride,
                       int B, int H, int Nq, int Nk, int dh, float scale, int dtype, void* stream) {
    AttnArgs a{};
    a.q = (const f16*)q; a.q_stride = q_stride; a.k = (const f16*)k; a.k_stride = k_stride; a.v = (const f16*)v; a.v_stride = v_stride;
    a.out = (f16*)out; a.out_stride = out_stride; a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.dh = dh; a.scale = scale; a.dtype = dtype;
    return launch_attention(a, (hipStream_t)stream);
}

int cs_op_attention_causal(const void* q, int q_stride, const void* k, int k_stride, const void* v, int v_stride, void* out, int out_stride,
                           int B, int H, int N, int dh, float scale, void* stream) {
    AttnArgs a{};
    a.q = (const f16*)q; a.q_stride = q_stride; a.k = (const f16*)k; a.k_stride = k_stride; a.v = (const f16*)v; a.v_stride = v_stride;
    a.out = (f16*)out; a.out_stride = out_stride; a.B = B; a.H = H; a.Nq = N; a.Nk = N; a.dh = dh; a.scale = scale; a.causal = 1;
    return launch_attention(a, (hipStream_t)stream);
}

int cs_op_rms_norm(const void* x, const void* weight, void* out, int M, int C, float eps, int dtype, void* stream) {
    return launch_rms_norm(x, weight, out, M, C, eps, dtype, (hipStream_t)stream);
}
int cs_op_gated_mul(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
    return launch_gated_mul(a, b, out, (long)n, dtype, (hipStream_t)stream);
}
int cs_op_embed_rows(const int64_t* ids, const void* table, void* out, int64_t rows, int C, int vocab, void* stream) {
    return launch_embed_rows(ids, table, out, (long)rows, C, vocab, (hipStream_t)stream);
}
int cs_op_attention_bias(const void* q, int q_stride, const void* k, int k_stride, const void* v, int v_stride, void* out, int out_stride,
                         int B, int H, int N, int dh, float scale, const float* bias_log2e, int dtype, void* stream) {
    if (!bias_log2e) CS_FAIL(CS_E_ARG, "attention_bias: bias is NULL");
    AttnArgs a{};
    a.q = (const f16*)q; a.q_stride = q_stride; a.k = (const f16*)k; a.k_stride = k_stride; a.v = (const f16*)v; a.v_stride = v_stride;
    a.out = (f16*)out; a.out_stride = out_stride; a.B = B; a.H = H; a.Nq = N; a.Nk = N; a.dh = dh; a.scale = scale; a.dtype = dtype; a.bias = bias_log2e;
    return launch_attention(a, (hipStream_t)stream);
}

size_t cs_op_group_norm_workspace(int B, int C) { return (size_t)B * (GN_SPLITS + 1) * C * 2 * sizeof(float); }

int cs_op_group_norm(const void* x0, int c0, const void* x1, int c1, int B, int HW, int groups, float eps, int silu,
                     const void* gamma, const void* beta, void* workspace, void* out, void* stream) {
    GroupNormArgs a{};
    a.x0 = (const f16*)x0; a.x1 = (const f16*)x1; a.c0 = c0; a.c1 = c1; a.B = B; a.HW = HW; a.groups = groups; a.eps = eps; a.silu = silu;
    a.gamma = (const f16*)gamma; a.beta = (const f16*)beta; a.partial = (float*)workspace; a.out = (f16*)out;
    return launch_group_norm(a, (hipStream_t)stream);
}

int cs_op_group_norm_pre(const void* x0, int c0, const float* stats0, const void* x1, int c1, const float* stats1, int B, int HW, int groups,
                         float eps, int silu, const void* gamma, const void* beta, void* workspace, void* out, void* stream) {
    if (HW % 64) CS_FAIL(CS_E_SHAPE, "group_norm_pre: HW = %d must be a multiple of 64", HW);
    GroupNormArgs a{};
    a.x0 = (const f16*)x0; a.x1 = (const f16*)x1; a.c0 = c0; a.c1 = c1; a.B = B; a.HW = HW; a.groups = groups; a.eps = eps; a.silu = silu;
    a.gamma = (const f16*)gamma; a.beta = (const f16*)beta; a.partial = (float*)workspace; a.out = (f16*)out;
    a.stats0 = stats0; a.S0 = HW / 64; a.stats1 = stats1; a.S1 = HW / 64;
    return launch_group_norm(a, (hipStream_t)stream);
}

int cs_op_layer_norm(const void* x, const void* gamma, const void* beta, void* out, int M, int C, float eps, void* stream) {
    return launch_layer_norm((const f16*)x, (const f16*)gamma, (const f16*)beta, (f16*)out, M, C, eps, (hipStream_t)stream);
}

int cs_debug_trace_read(void* dst_host, size_t bytes) { return debug_trace_read(dst_host, bytes); }
int cs_debug_attn_trace_read(void* dst_host, size_t bytes) { return debug_attn_trace_read(dst_host, bytes); }

}  // extern "C"
