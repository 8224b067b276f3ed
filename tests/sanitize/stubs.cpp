// Host-only stand-ins for the sanitizer harness (tests/sanitize/harness.cpp; TEST INFRASTRUCTURE, never linked into libconsolver_hip.so).
//
// The executors (unet.cpp / vae.cpp / flux.cpp / clip.cpp / ops_api.cpp / api.cpp) are host code: weight repacking, a first-fit arena over the caller's
// workspace, the dry-run workspace sizing and the launch sequence.  They are compiled here with -fsanitize=address,undefined (hipcc --cuda-host-only) and
// linked against THIS file instead of the HIP kernels and the HIP runtime:
//   * "device" memory is host malloc, so AddressSanitizer sees every buffer the executors carve out of the workspace;
//   * every launch_* stub TOUCHES the first and last byte of each tensor the real kernel would read or write (sizes from the launch arguments), so an
//     arena block that is too small, freed too early or placed past the end of the workspace is an ASan error, exactly as it would be a silent
//     corruption on the GPU.
#include "../../consolver_amd/csrc/ops.h"
#include <cstdlib>
#include <cstring>

// ---- HIP runtime -------------------------------------------------------------------------------------------------------
extern "C" {
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free((void*)s); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free((void*)e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
}

// ---- touch helpers -----------------------------------------------------------------------------------------------------
static volatile unsigned char g_sink;
static void rd(const void* p, size_t bytes) { if (p && bytes) { g_sink = ((const volatile unsigned char*)p)[0]; g_sink = ((const volatile unsigned char*)p)[bytes - 1]; } }
static void wr(void* p, size_t bytes) { if (p && bytes) { ((volatile unsigned char*)p)[0] = 1; ((volatile unsigned char*)p)[bytes - 1] = 1; } }
static size_t esz(int dtype) { return dtype == CS_F32 ? 4 : 2; }

double igemm_flops(const IgemmArgs& a) { return 2.0 * a.B * a.Ho * a.Wo * (double)a.N * a.taps * (a.c0 + a.c1); }
int launch_igemm(const IgemmArgs& a, hipStream_t) {
    const size_t Min = (size_t)a.B * a.Hi * a.Wi, M = (size_t)a.B * a.Ho * a.Wo, Nout = a.geglu ? a.N / 2 : a.N;
    rd(a.a0, Min * a.c0 * 2); rd(a.a1, Min * a.c1 * 2); rd(a.a0_lo, Min * a.c0 * 2); rd(a.a1_lo, Min * a.c1 * 2);
    rd(a.w, (size_t)a.N * a.taps * (a.c0 + a.c1) * 2); rd(a.bias, (size_t)a.N * 2);
    if (a.w_up_sub) rd(a.w_up_sub, (size_t)4 * a.N * 4 * a.c0 * 2);            // (an upsampler's sub-pixel filters: [4][N][4 Cin])
    if (a.temb) rd(a.temb, ((size_t)(a.temb_stride ? a.B - 1 : 0) * a.temb_stride + a.N) * 2);
    const size_t lob = a.lo8 ? 1 : 2;                  // (an 8-bit lo plane holds one byte per element: the executor allocates exactly that)
    rd(a.res, M * Nout * 2); rd(a.res_lo, M * Nout * lob);
    wr(a.out, M * Nout * 2); wr(a.out_lo, M * Nout * lob);
    if (a.gn_stats) wr(a.gn_stats, (size_t)a.B * (a.Ho * a.Wo / 64) * a.N * sizeof(float));
    if (a.row_stats) { const int G = a.N % 160 == 0 ? a.N / 160 : a.N / 64; wr(a.row_stats, M * G * 2 * sizeof(float)); *a.row_stats_groups = G; }   // (the widest layout a kernel may pick is N / 64 groups)
    if (a.ln_stats) { rd(a.ln_stats, M * a.ln_groups * 2 * sizeof(float)); rd(a.ln_s, (size_t)a.N * 4); rd(a.ln_b, (size_t)a.N * 4); }
    if (a.splitk_ws) wr(a.splitk_ws, a.splitk_ws_bytes);
    return CS_OK;
}
int launch_row_stats(const f16* x, const f16* x_lo, int M, int C, float* stats, hipStream_t, int lo8) { rd(x, (size_t)M * C * 2); rd(x_lo, (size_t)M * C * (lo8 ? 1 : 2)); wr(stats, (size_t)M * 8); return CS_OK; }
int launch_ln_dc_ratio(const float* rs, int M, int G, int, float, float* dst, hipStream_t) { rd(rs, (size_t)M * G * 8); wr(dst, 4); return CS_OK; }
int launch_attention(const AttnArgs& a, hipStream_t) {
    const size_t C = (size_t)a.H * a.dh;
    rd(a.q, (((size_t)a.B * a.Nq - 1) * a.q_stride + C) * 2); rd(a.k, (((size_t)a.B * a.Nk - 1) * a.k_stride + C) * 2);
    rd(a.v, (((size_t)a.B * a.Nk - 1) * a.v_stride + C) * 2); wr(a.out, (((size_t)a.B * a.Nq - 1) * a.out_stride + C) * 2);
    if (a.split_ws) wr(a.split_ws, a.split_ws_bytes);
    if (a.bias) rd(a.bias, (size_t)a.H * a.Nq * a.Nk * 4);
    return CS_OK;
}
size_t attention_split_workspace_bytes(int B, int H, int Nq, int, int dh) { return (size_t)B * H * Nq * (dh + 2) * 4 * 5; }
int launch_gn_stats64(const f16* x, int B, int HW, int C, float* partial, hipStream_t) { rd(x, (size_t)B * HW * C * 2); wr(partial, (size_t)B * (HW / 64) * C * 4); return CS_OK; }
int launch_group_norm(const GroupNormArgs& a, hipStream_t) {
    const size_t M = (size_t)a.B * a.HW; const int C = a.c0 + a.c1, smax = a.splits > 0 ? a.splits : GN_SPLITS;
    rd(a.x0, M * a.c0 * 2); rd(a.x1, M * a.c1 * 2); rd(a.x0_lo, M * a.c0 * 2); rd(a.x1_lo, M * a.c1 * 2); rd(a.gamma, C * 2); rd(a.beta, C * 2);
    wr(a.partial, (size_t)a.B * (smax + 1) * C * 2 * sizeof(float)); wr(a.out, M * C * 2); wr(a.out_lo, M * C * 2);
    if (a.stats0) rd(a.stats0, (size_t)a.B * a.S0 * a.c0 * 4);
    if (a.stats1) rd(a.stats1, (size_t)a.B * a.S1 * a.c1 * 4);
    return CS_OK;
}
int launch_layer_norm(const f16* x, const f16* g, const f16* b, f16* out, int M, int C, float, hipStream_t, const f16* x_lo) {
    rd(x, (size_t)M * C * 2); rd(x_lo, (size_t)M * C * 2); rd(g, C * 2); rd(b, C * 2); wr(out, (size_t)M * C * 2); return CS_OK;
}
int launch_xattn_block(const XattnArgs& a, hipStream_t) {
    const size_t n = (size_t)a.M * a.C * 2;
    const size_t nl = a.lo8 ? n / 2 : n;
    rd(a.h, n); rd(a.h_lo, nl); wr(a.out, n); wr(a.out_lo, nl); wr(a.row_stats, (size_t)a.M * 8); rd(a.kv, (size_t)(a.M / a.HW) * a.Nk * 2 * a.C * 2); rd(a.wq, (size_t)a.C * a.C * 2); rd(a.wo, (size_t)a.C * a.C * 2);
    return CS_OK;
}
int launch_time_embedding(const float* t, int Bt, int C0, int D, const f16* w1, const f16*, const f16* w2, const f16*, f16* scratch, f16* out, hipStream_t) {
    rd(t, Bt * 4); rd(w1, (size_t)D * C0 * 2); rd(w2, (size_t)D * D * 2); wr(scratch, (size_t)Bt * (C0 + D) * 2); wr(out, (size_t)Bt * D * 2); return CS_OK;
}
int launch_rowvec_linear(const f16* x, int R, int K, const f16* w, const f16*, int N, f16* out, int, hipStream_t) { rd(x, (size_t)R * K * 2); rd(w, (size_t)N * K * 2); wr(out, (size_t)R * N * 2); return CS_OK; }
int launch_conv_in(const f16* lat, int n_lat, int B, int Cin, int H, int W, const f16*, const f16*, int Cout, f16* out, hipStream_t, f16* out_lo) {
    rd(lat, (size_t)n_lat * Cin * H * W * 2); wr(out, (size_t)B * H * W * Cout * 2); wr(out_lo, (size_t)B * H * W * Cout * 2); return CS_OK;
}
int launch_conv_out(const f16* x, int B, int Cin, int H, int W, const f16*, const f16*, int Cout, f16* out, hipStream_t, int out_f32, const f16* x_lo, float* scratch32) { rd(x, (size_t)B * H * W * Cin * 2); rd(x_lo, (size_t)B * H * W * Cin * 2); wr(scratch32, (size_t)B * Cout * H * W * 4); wr(out, (size_t)B * Cout * H * W * (out_f32 ? 4 : 2)); return CS_OK; }
int launch_conv_out3(const f16* x, int B, int Cin, int H, int W, const f16*, const f16*, f16* out, int, hipStream_t) { rd(x, (size_t)B * H * W * Cin * 2); wr(out, (size_t)B * 3 * H * W * 2); return CS_OK; }
int launch_conv_out_small(const f16* x, int B, int Cin, int H, int W, const f16*, const f16*, int Cout, f16* out, hipStream_t) { rd(x, (size_t)B * H * W * Cin * 2); wr(out, (size_t)B * Cout * H * W * 2); return CS_OK; }
int launch_pixel_linear_nchw(const f16* x, const f16*, const f16*, f16* out, int B, int C, int HW, float, float, hipStream_t) { rd(x, (size_t)B * C * HW * 2); wr(out, (size_t)B * C * HW * 2); return CS_OK; }
int launch_pixel_affine_nchw(const f16* x, int Cin, const f16*, const f16*, int Cout, f16* out, int B, int HW, float, float, hipStream_t) { rd(x, (size_t)B * Cin * HW * 2); wr(out, (size_t)B * Cout * HW * 2); return CS_OK; }
int launch_latent_to_nhwc64(const f16* x, const f16*, const f16*, f16* out, int B, int C, int HW, float, float, hipStream_t) { rd(x, (size_t)B * C * HW * 2); wr(out, (size_t)B * HW * 64 * 2); return CS_OK; }
int launch_row_softmax(f16* x, long rows, int cols, float, hipStream_t) { wr(x, (size_t)rows * cols * 2); return CS_OK; }
int launch_embed_tokens(const int64_t* ids, const f16*, const f16*, f16* out, long rows, int, int C, int, hipStream_t) { rd(ids, rows * 8); wr(out, (size_t)rows * C * 2); return CS_OK; }
int launch_quick_gelu(f16* x, long n, hipStream_t) { wr(x, (size_t)n * 2); return CS_OK; }
int launch_gemm2(const Gemm2Args& a, hipStream_t) {
    rd(a.a, 2); rd(a.w, (size_t)((a.N + 255) / 256 * 256) * a.K * 2); wr(a.out, 2);
    if (!a.c_seg_rows) wr(a.out, (((size_t)a.M - 1 + a.c_row_off) * a.ldc + a.c_col_off + a.N) * 2);
    if (a.out_lo) {          // split residual stream: the lo planes are addressed like res / out
        rd(a.res_lo, 2); wr(a.out_lo, 2); rd(a.res, 2);
        if (!a.c_seg_rows) { const size_t n = (((size_t)a.M - 1 + a.c_row_off) * a.ldc + a.c_col_off + a.N) * 2; rd(a.res_lo, n); wr(a.out_lo, n); rd(a.res, n); }
    }
    if (!a.a_seg_rows) rd(a.a, (((size_t)a.M - 1 + a.a_row_off) * a.lda + a.K) * 2);
    if (a.tail_ws) wr(a.tail_ws, a.tail_ws_bytes);
    return CS_OK;
}
int launch_gemm2_pair(const Gemm2Args& a, const Gemm2Args& b, hipStream_t s) { launch_gemm2(a, s); return launch_gemm2(b, s); }
size_t gemm2_tail_workspace_bytes(int tiles, int K) { return K >= 6144 && tiles % 256 ? (size_t)3 * 256 * 256 * 256 * 4 : 0; }
int launch_small_linear(const float* x, int R, int K, const void* w, const void*, int N, float* out, int, int, int, hipStream_t) { rd(x, (size_t)R * K * 4); rd(w, (size_t)N * K * 2); wr(out, (size_t)R * N * 4); return CS_OK; }
int launch_planes_to_f32(const void* hi, const void* lo, float* out, long n, int, hipStream_t) { rd(hi, (size_t)n * 2); rd(lo, (size_t)n * 2); wr(out, (size_t)n * 4); return CS_OK; }
int launch_ln_modulate(const void* x, void* y, int M, int C, int, const float* sh, const float* sc, long, float, int, hipStream_t, const void* x_lo, void* y_lo) { rd(x, (size_t)M * C * 2); rd(x_lo, (size_t)M * C * 2); wr(y, (size_t)M * C * 2); wr(y_lo, (size_t)M * C * 2); rd(sh, C * 4); rd(sc, C * 4); return CS_OK; }
int launch_qk_norm_rope(void* qkv, long ld, int rows, int, int heads, int dh, int, int k_col, const void*, const void*, const void*, const void*, int, const float*, const float*, float, int, hipStream_t) {
    wr(qkv, (((size_t)rows - 1) * ld + k_col + (size_t)heads * dh) * 2); return CS_OK;
}
int launch_sinusoid_f32(const float* t, float, int R, int C, float* out, hipStream_t) { rd(t, R * 4); wr(out, (size_t)R * C * 4); return CS_OK; }
int launch_add3_f32(const float* a, const float* b, const float* c, float* out, long n, hipStream_t) { rd(a, n * 4); rd(b, n * 4); rd(c, n * 4); wr(out, n * 4); return CS_OK; }
int launch_cast_f32(const float* x, void* out, long n, int, hipStream_t) { rd(x, n * 4); wr(out, n * 2); return CS_OK; }
int launch_rms_norm(const void* x, const void*, void* y, int M, int C, float, int, hipStream_t) { rd(x, (size_t)M * C * 2); wr(y, (size_t)M * C * 2); return CS_OK; }
int launch_gated_mul(const void* a, const void* b, void* out, long n, int, hipStream_t) { rd(a, n * 2); rd(b, n * 2); wr(out, n * 2); return CS_OK; }
int launch_embed_rows(const int64_t* ids, const void*, void* out, long rows, int C, int, hipStream_t) { rd(ids, rows * 8); wr(out, (size_t)rows * C * 2); return CS_OK; }
int debug_trace_read(void*, size_t) { return CS_OK; }
int debug_attn_trace_read(void*, size_t) { return CS_OK; }
