// ConsistencySolver update kernels (HBM-bound streaming + latency-bound policy MLP).
//
//   cs_lms_ddim_step  : CFG combine + coefficient fix-up + linear-multistep combine + DDIM
//                       (scheduler_ppo.py:165-175,253-283,306-332; denoise_ppo.py:96-100)
//   cs_lms_euler_step : same with the flow-matching Euler epilogue
//                       (edit_ppo/scheduler_fmppo.py:400-436)
//   cs_factor_probs   : FactorNetPPO.forward_ (factor_net_ppo.py:137-157)
//   cs_cosine_features: compute_cosine_similarity (factor_net_ppo.py:108-130)
//   cs_sample_actions / cs_gather_actions / cs_action_probs / cs_step_masks / cs_stack_history
//
// Arithmetic is fp32 in the reference's operation order (file is built with
// -ffp-contract=off so that a*b+c is two roundings like torch's separate ops);
// half/bfloat I/O is converted on load and rounded once on store.
#include "common.h"

namespace {

template <typename T> struct Io;
template <> struct Io<float> {
    static constexpr int VEC = 4;
    typedef f32x4 vec_t;
    __device__ static void load(const void* p, int64_t i, float (&o)[4]) {
        f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + i);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
    __device__ static void store(void* p, int64_t i, const float (&o)[4]) {
        f32x4 v = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p) + i) = v;
    }
    __device__ static float load1(const void* p, int64_t i) { return reinterpret_cast<const float*>(p)[i]; }
    __device__ static void store1(void* p, int64_t i, float v) { reinterpret_cast<float*>(p)[i] = v; }
    __device__ static float round(float v) { return v; }
};
template <> struct Io<f16> {
    static constexpr int VEC = 8;
    __device__ static void load(const void* p, int64_t i, float (&o)[8]) {
        f16x8 v = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(p) + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
    }
    __device__ static void store(void* p, int64_t i, const float (&o)[8]) {
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (f16)o[j];
        *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p) + i) = v;
    }
    __device__ static float load1(const void* p, int64_t i) { return (float)reinterpret_cast<const f16*>(p)[i]; }
    __device__ static void store1(void* p, int64_t i, float v) { reinterpret_cast<f16*>(p)[i] = (f16)v; }
    __device__ static float round(float v) { return (float)(f16)v; }
};
struct bf16_tag {};
template <> struct Io<bf16_tag> {
    static constexpr int VEC = 8;
    __device__ static void load(const void* p, int64_t i, float (&o)[8]) {
        u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const u16*>(p) + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(v[j] << 16); o[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u); }
    }
    __device__ static void store(void* p, int64_t i, const float (&o)[8]) {
        u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (unsigned)f32_to_bf16(o[2 * j]) | ((unsigned)f32_to_bf16(o[2 * j + 1]) << 16);
        *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p) + i) = v;
    }
    __device__ static float load1(const void* p, int64_t i) { return bf16_to_f32(reinterpret_cast<const u16*>(p)[i]); }
    __device__ static void store1(void* p, int64_t i, float v) { reinterpret_cast<u16*>(p)[i] = f32_to_bf16(v); }
    __device__ static float round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

// the low-precision copy of a result vector (CsStepArgs::x_out_lp): V = 4 or 8 values as fp16 (lp_bf16 == 0) or bf16, one 8- / 16-byte store
template <int V>
__device__ __forceinline__ void store_lp(void* p, int64_t i, const float (&o)[V], int lp_bf16) {
    unsigned w[V / 2];
#pragma unroll
    for (int j = 0; j < V / 2; ++j) {
        if (lp_bf16) w[j] = (unsigned)f32_to_bf16(o[2 * j]) | ((unsigned)f32_to_bf16(o[2 * j + 1]) << 16);
        else { union { f16 h[2]; unsigned u; } c; c.h[0] = (f16)o[2 * j]; c.h[1] = (f16)o[2 * j + 1]; w[j] = c.u; }
    }
    if constexpr (V == 8) *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p) + i) = u32x4{w[0], w[1], w[2], w[3]};
    else { typedef unsigned u32x2v __attribute__((ext_vector_type(2))); *reinterpret_cast<u32x2v*>(reinterpret_cast<u16*>(p) + i) = u32x2v{w[0], w[1]}; }
}
__device__ __forceinline__ void store_lp1(void* p, int64_t i, float v, int lp_bf16) {
    if (lp_bf16) reinterpret_cast<u16*>(p)[i] = f32_to_bf16(v); else reinterpret_cast<f16*>(p)[i] = (f16)v;
}

struct StepParams {
    const void* x; const void* ec; const void* eu; float g;
    const void* hist[CS_MAX_ORDER];
    int m, order, scaler;
    const float* actions; int astride;
    int64_t elems;
    void* x_out; void* eps_out;
    float sat, s1mat, sap, s1map; int vpred; float dt;
    int x_f32;   // x is fp32 although eps / history are TI (scheduler_fmppo.py:354 upcasts the sample, not the model output)
    void* x_lp;  // optional 16-bit copy of an fp32 result (CsStepArgs::x_out_lp)
    int lp_bf16; // ... as bf16 (else fp16)
};

// coefficient fix-up, identical for every thread of a sample (b is block-uniform)
__device__ __forceinline__ void make_coeffs(const StepParams& p, int b, float (&c)[CS_MAX_ORDER], float& sc0, float& sc1) {
    const float* a = p.actions + (int64_t)b * p.astride;
#pragma unroll
    for (int k = 0; k < CS_MAX_ORDER; ++k) c[k] = 0.f;
    if (p.m > 1) {
        // list [a0+1, a1, ..., a_{order-2}, placeholder]; entry m-1 := 1 - sum(first m-1)
#pragma unroll
        for (int k = 0; k < CS_MAX_ORDER - 1; ++k)
            if (k < p.m - 1) c[k] = (k == 0) ? (a[0] + 1.0f) : a[k];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CS_MAX_ORDER - 1; ++k)
            if (k < p.m - 1) s = s + c[k];
#pragma unroll
        for (int k = 1; k < CS_MAX_ORDER; ++k)
            if (k == p.m - 1) c[k] = 1.0f - s;
    }
    sc0 = (p.scaler >= 1) ? a[p.order - 1] + 1.0f : 1.0f;
    sc1 = (p.scaler >= 2) ? a[p.order] + 1.0f : 1.0f;
}

template <typename TI, typename TO, bool EULER, bool OUT_X_TI>
__device__ __forceinline__ float solve_one(const StepParams& p, float x, float e0, const float* h, const float (&c)[CS_MAX_ORDER], float sc0, float sc1) {
    float eff;
    if (p.m == 1) {
        eff = e0;
    } else {
        eff = c[0] * e0;
#pragma unroll
        for (int k = 1; k < CS_MAX_ORDER; ++k)
            if (k < p.m) eff = eff + c[k] * h[k - 1];
    }
    if (p.scaler >= 1) eff = eff * sc0;
    if (p.scaler >= 2) x = x * sc1;
    if (EULER) {
        // scheduler_fmppo.py:429 `sample + dt * effective_model_output`: with ONE history entry and no scale the effective
        // output is the bf16 / f16 model output itself and dt is a 0-d fp32 tensor, so torch runs the product as a 16-bit op
        // (dt cast to the model dtype, product rounded to it) before the fp32 add; otherwise the [B,1,1] fp32 coefficients
        // have promoted everything to fp32.
        if (p.m == 1 && p.scaler == 0) return x + Io<TI>::round(Io<TI>::round(p.dt) * eff);
        return x + p.dt * eff;
    }
    if (p.vpred) eff = p.sat * eff + p.s1mat * x;
    float x0 = (x - p.s1mat * eff) / p.sat;
    return p.sap * x0 + p.s1map * eff;
}

template <typename TI, typename TO, bool EULER>
__global__ __launch_bounds__(256) void lms_step_kernel(StepParams p) {
    constexpr int V = Io<TI>::VEC;
    const int b = blockIdx.y;
    float c[CS_MAX_ORDER], sc0, sc1;
    make_coeffs(p, b, c, sc0, sc1);
    const int64_t base = (int64_t)b * p.elems;
    const int64_t nvec = p.elems / V;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + v * V;
        float x[V], e[V], u[V], h[CS_MAX_ORDER - 1][V], o[V];
        if (p.x_f32) {   // block-uniform
#pragma unroll
            for (int j = 0; j < V; ++j) x[j] = reinterpret_cast<const float*>(p.x)[i + j];
        } else {
            Io<TI>::load(p.x, i, x);
        }
        Io<TI>::load(p.ec, i, e);
        if (p.eu) {
            Io<TI>::load(p.eu, i, u);
#pragma unroll
            for (int j = 0; j < V; ++j) e[j] = Io<TI>::round(u[j] + p.g * (e[j] - u[j]));
        }
#pragma unroll
        for (int k = 0; k < CS_MAX_ORDER - 1; ++k)
            if (k < p.m - 1) Io<TI>::load(p.hist[k], i, h[k]);
        if (p.eps_out) Io<TI>::store(p.eps_out, i, e);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float hh[CS_MAX_ORDER - 1];
#pragma unroll
            for (int k = 0; k < CS_MAX_ORDER - 1; ++k) hh[k] = (k < p.m - 1) ? h[k][j] : 0.f;
            o[j] = solve_one<TI, TO, EULER, false>(p, x[j], e[j], hh, c, sc0, sc1);
        }
        if constexpr (Io<TO>::VEC == V) {
            Io<TO>::store(p.x_out, i, o);
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) Io<TO>::store1(p.x_out, i + j, o[j]);
        }
        if (p.x_lp) store_lp<V>(p.x_lp, i, o, p.lp_bf16);        // (block-uniform) the 16-bit view of an fp32 state: what `x.to(model dtype)` would round the stored result to
    }
    // scalar tail (elems % V != 0): handled by the last block of each sample
    if (blockIdx.x == 0) {
        for (int64_t t = nvec * V + threadIdx.x; t < p.elems; t += blockDim.x) {
            const int64_t i = base + t;
            float x = p.x_f32 ? reinterpret_cast<const float*>(p.x)[i] : Io<TI>::load1(p.x, i), e = Io<TI>::load1(p.ec, i);
            if (p.eu) { float u = Io<TI>::load1(p.eu, i); e = Io<TI>::round(u + p.g * (e - u)); }
            float hh[CS_MAX_ORDER - 1];
#pragma unroll
            for (int k = 0; k < CS_MAX_ORDER - 1; ++k) hh[k] = (k < p.m - 1) ? Io<TI>::load1(p.hist[k], i) : 0.f;
            if (p.eps_out) Io<TI>::store1(p.eps_out, i, e);
            const float r = solve_one<TI, TO, EULER, false>(p, x, e, hh, c, sc0, sc1);
            Io<TO>::store1(p.x_out, i, r);
            if (p.x_lp) store_lp1(p.x_lp, i, r, p.lp_bf16);
        }
    }
}

template <bool EULER>
int launch_step(const CsStepArgs* a, void* stream) {
    if (!a) CS_FAIL(CS_E_ARG, "args is NULL");
    if (a->B < 0 || a->elems < 0) CS_FAIL(CS_E_SHAPE, "negative shape");
    const bool empty = (a->B == 0 || a->elems == 0);
    if (!empty && (!a->x || !a->eps_text || !a->x_out)) CS_FAIL(CS_E_ARG, "x, eps_text and x_out are required");
    if (a->order_dim < 1 || a->order_dim > CS_MAX_ORDER) CS_FAIL(CS_E_ARG, "order_dim %d out of range [1,%d]", a->order_dim, CS_MAX_ORDER);
    if (a->scaler_dim < 0 || a->scaler_dim > 2) CS_FAIL(CS_E_UNSUPPORTED, "scaler_dim %d > 2 is not implemented (scheduler_ppo.py:280)", a->scaler_dim);
    if (a->m < 1 || a->m > a->order_dim) CS_FAIL(CS_E_ARG, "history length m=%d not in [1, order_dim=%d]", a->m, a->order_dim);
    if (empty) return CS_OK;
    if ((a->m > 1 || a->scaler_dim > 0) && !a->actions) CS_FAIL(CS_E_ARG, "actions required when m > 1 or scaler_dim > 0");
    if (a->actions && a->actions_stride < a->order_dim + a->scaler_dim - 1) CS_FAIL(CS_E_SHAPE, "actions_stride %d < order+scaler-1", a->actions_stride);
    if (a->eps_uncond && !a->eps_out) CS_FAIL(CS_E_ARG, "eps_out is required with CFG (the combined eps is the history entry)");
    for (int k = 0; k < a->m - 1; ++k)
        if (!a->hist[k]) CS_FAIL(CS_E_ARG, "hist[%d] is NULL but m=%d", k, a->m);
    StepParams p;
    p.x = a->x; p.ec = a->eps_text; p.eu = a->eps_uncond; p.g = a->guidance;
    for (int k = 0; k < CS_MAX_ORDER; ++k) p.hist[k] = a->hist[k];
    p.m = a->m; p.order = a->order_dim; p.scaler = a->scaler_dim;
    p.actions = a->actions; p.astride = a->actions_stride; p.elems = a->elems;
    p.x_out = a->x_out; p.eps_out = a->eps_out;
    p.sat = a->sqrt_at; p.s1mat = a->sqrt_1mat; p.sap = a->sqrt_ap; p.s1map = a->sqrt_1map;
    p.vpred = a->v_prediction; p.dt = a->dt;
    p.x_f32 = (a->x_is_f32 != 0 && a->io_dtype != CS_F32);
    p.x_lp = a->x_out_lp; p.lp_bf16 = a->lp_dtype == CS_BF16;
    if (a->x_out_lp && (a->out_dtype != CS_F32 || (a->lp_dtype != CS_F16 && a->lp_dtype != CS_BF16)))
        CS_FAIL(CS_E_ARG, "x_out_lp is the 16-bit copy (lp_dtype CS_F16 or CS_BF16) of an fp32 result (out_dtype CS_F32)");
    const int vec = (a->io_dtype == CS_F32) ? 4 : 8;
    int64_t nvec = a->elems / vec;
    int gx = (int)((nvec + 255) / 256);
    // memory-bound: cap the grid at ~8 blocks per CU overall and grid-stride the rest
    int cap = (2048 + a->B - 1) / a->B; if (cap < 1) cap = 1;
    if (gx > cap) gx = cap; if (gx < 1) gx = 1;
    dim3 grid(gx, a->B), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(TI, TO) hipLaunchKernelGGL((lms_step_kernel<TI, TO, EULER>), grid, block, 0, s, p)
    const int io = a->io_dtype, od = a->out_dtype;
    if (io == CS_F32 && od == CS_F32) LAUNCH(float, float);
    else if (io == CS_F16 && od == CS_F16) LAUNCH(f16, f16);
    else if (io == CS_F16 && od == CS_F32) LAUNCH(f16, float);
    else if (io == CS_BF16 && od == CS_BF16) LAUNCH(bf16_tag, bf16_tag);
    else if (io == CS_BF16 && od == CS_F32) LAUNCH(bf16_tag, float);
    else CS_FAIL(CS_E_DTYPE, "unsupported io/out dtype pair (%d,%d)", io, od);
#undef LAUNCH
    CS_CHECK_LAUNCH();
    return CS_OK;
}

// ---------------------------------------------------------------- policy MLP
// one workgroup per DISTINCT conditioning row; weights (<= ~1 MB) stay in L2.  When one row is
// broadcast to all B samples (x_row_stride == 0, no cosine features: every sampling step), a single
// workgroup evaluates the MLP once and writes the B identical probability rows.  Each wave keeps
// RW weight rows in flight (one 16-byte load per lane and row when hidden % 4 == 0) so the layer is
// a few L2 round trips instead of hidden / waves dependent ones.
template <int RW>
__device__ __forceinline__ void dense_rows(const float* __restrict__ w, const float* __restrict__ bias, const float* in_lds,
                                           float* out_lds, int rows, int K, int wave, int nw, int lane, bool relu, float post) {
    const bool vec = (K & 3) == 0 && ((uintptr_t)w & 15) == 0;
    for (int j0 = wave * RW; j0 < rows; j0 += nw * RW) {
        float acc[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r] = 0.f;
        if (vec) {
            for (int k = lane * 4; k < K; k += 256) {
                f32x4 wv[RW];
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const int j = j0 + r < rows ? j0 + r : rows - 1;
                    wv[r] = *reinterpret_cast<const f32x4*>(w + (int64_t)j * K + k);
                }
                const f32x4 xv = *reinterpret_cast<const f32x4*>(in_lds + k);
#pragma unroll
                for (int r = 0; r < RW; ++r) acc[r] += xv[0] * wv[r][0] + xv[1] * wv[r][1] + xv[2] * wv[r][2] + xv[3] * wv[r][3];
            }
        } else {
            for (int k = lane; k < K; k += 64) {
                const float xv = in_lds[k];
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const int j = j0 + r < rows ? j0 + r : rows - 1;
                    acc[r] += xv * w[(int64_t)j * K + k];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const float v = wave_sum(acc[r]);
            if (lane == 0 && j0 + r < rows) {
                const float o = (v + bias[j0 + r]) * post;
                out_lds[j0 + r] = relu ? fmaxf(o, 0.f) : o;
            }
        }
    }
}

__global__ __launch_bounds__(1024) void factor_probs_kernel(CsFactorNet n, const float* x, int xstride, const float* cosf, float* probs,
                                                            int rows_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* in = sm;                               // [in_dim] (<= 16)
    float* h0 = sm + 16;                          // [H] (padded to a multiple of 4)
    const int Hp = (n.hidden + 3) & ~3;
    float* h1 = h0 + Hp;                          // [H]
    float* lg = h1 + Hp;                          // [A*K] logits, then probabilities
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    if (tid < n.in_dim) in[tid] = (tid < 2) ? x[(int64_t)b * xstride + tid] * n.input_scale
                                            : cosf[(int64_t)b * (n.in_dim - 2) + tid - 2];
    __syncthreads();
    for (int j = tid; j < n.hidden; j += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < n.in_dim; ++k) acc += in[k] * n.w0[j * n.in_dim + k];
        h0[j] = fmaxf(acc + n.b0[j], 0.f);
    }
    __syncthreads();
    dense_rows<8>(n.w1, n.b1, h0, h1, n.hidden, n.hidden, wave, nw, lane, true, 1.f);
    __syncthreads();
    const int AK = n.action_dims * n.num_actions;
    dense_rows<4>(n.w2, n.b2, h1, lg, AK, n.hidden, wave, nw, lane, false, n.inv_temperature);
    __syncthreads();
    for (int a = wave; a < n.action_dims; a += nw) {
        float* row = lg + a * n.num_actions;
        float mx = -INFINITY;
        for (int k = lane; k < n.num_actions; k += 64) mx = fmaxf(mx, row[k]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int k = lane; k < n.num_actions; k += 64) s += expf(row[k] - mx);
        s = wave_sum(s);
        for (int k = lane; k < n.num_actions; k += 64) row[k] = expf(row[k] - mx) / s;
    }
    __syncthreads();
    // rows_out == 1: this workgroup's own row; > 1: the broadcast row written to every sample
    float* out = probs + (int64_t)b * AK;
    for (int r = wave; r < rows_out; r += nw)
        for (int k = lane; k < AK; k += 64) out[(int64_t)r * AK + k] = lg[k];
}

struct HistPtrs { const void* p[CS_MAX_ORDER]; };

// CFG form (eu != null): hist[0] is the TEXT branch and the newest history entry is the combined eps u + g (c - u), rounded to
// the model dtype exactly as lms_step_kernel rounds it; it is formed on the fly and the blocks of feature 1 also write it to
// eps_out (every launch has them, whatever m is), so the update kernel that follows takes it as an already combined eps.
template <typename T>
__global__ __launch_bounds__(256) void cosine_kernel(HistPtrs h, int m, int order, int64_t elems, float* out, const void* eu, float g, void* eps_out) {
    // grid (order-1, B): cos(hist[i], hist[0]), i = blockIdx.x + 1
    const int i = blockIdx.x + 1, b = blockIdx.y;
    __shared__ float red[3][4];
    float dot = 0.f, na = 0.f, nb = 0.f;
    const bool writer = eps_out != nullptr && i == 1;
    if (i < m || writer) {
        constexpr int V = Io<T>::VEC;
        const int64_t base = (int64_t)b * elems, nvec = elems / V;
        for (int64_t v = threadIdx.x; v < nvec; v += blockDim.x) {
            float a[V], c[V];
            Io<T>::load(h.p[0], base + v * V, c);
            if (eu) {
                float u[V];
                Io<T>::load(eu, base + v * V, u);
#pragma unroll
                for (int j = 0; j < V; ++j) c[j] = Io<T>::round(u[j] + g * (c[j] - u[j]));
                if (writer) Io<T>::store(eps_out, base + v * V, c);
            }
            if (i < m) {
                Io<T>::load(h.p[i], base + v * V, a);
#pragma unroll
                for (int j = 0; j < V; ++j) { dot += a[j] * c[j]; na += a[j] * a[j]; nb += c[j] * c[j]; }
            }
        }
        for (int64_t t = nvec * V + threadIdx.x; t < elems; t += blockDim.x) {
            float c = Io<T>::load1(h.p[0], base + t);
            if (eu) {
                const float u = Io<T>::load1(eu, base + t);
                c = Io<T>::round(u + g * (c - u));
                if (writer) Io<T>::store1(eps_out, base + t, c);
            }
            if (i < m) { const float a = Io<T>::load1(h.p[i], base + t); dot += a * c; na += a * a; nb += c * c; }
        }
    }
    dot = wave_sum(dot); na = wave_sum(na); nb = wave_sum(nb);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = dot; red[1][wave] = na; red[2][wave] = nb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float d = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        float a = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        float c = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        float r = 0.f;
        if (i < m) r = d / (fmaxf(sqrtf(a), 1e-8f) * fmaxf(sqrtf(c), 1e-8f));
        out[(int64_t)b * (order - 1) + (i - 1)] = r;
    }
}

__global__ void sample_kernel(const float* probs, const float* uni, const int64_t* idx_in, const float* av,
                              int B, int A, int K, int64_t* idx, float* actions, float* aprobs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * A) return;
    const int a = t % A;
    const float* p = probs + (int64_t)t * K;
    int k;
    if (idx_in) {
        k = (int)idx_in[t];
        k = k < 0 ? 0 : (k >= K ? K - 1 : k);
    } else {
        const float u = uni[t];
        float c = 0.f; k = K - 1;
        for (int j = 0; j < K; ++j) { c += p[j]; if (u < c) { k = j; break; } }
        // never return a zero-probability bin because of rounding at the tail
        while (k > 0 && p[k] <= 0.f) --k;
    }
    if (idx) idx[t] = k;
    if (actions) actions[t] = av[a * K + k];
    if (aprobs) aprobs[t] = p[k];
}

__global__ void action_probs_kernel(const float* probs, const float* actions, const float* av, int B, int A, int K,
                                    float* sel, float* ent) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * A) return;
    const int a = t % A;
    const float* p = probs + (int64_t)t * K;
    const float act = actions[t];
    int best = 0; float bd = fabsf(act - av[a * K]);
    float s = 0.f;
    for (int j = 0; j < K; ++j) {
        float d = fabsf(act - av[a * K + j]);
        if (d < bd) { bd = d; best = j; }
        s += p[j];
    }
    float h = 0.f;
    // torch.distributions.Categorical(probs=p).entropy(): -sum q log(clamp(q, eps, 1 - eps)) with q = p / sum p
    // (the same clamp the PPO gradient kernel differentiates, ppo.hip policy_row_kernel)
    for (int j = 0; j < K; ++j) { float q = p[j] / s; h -= q * logf(fminf(fmaxf(q, 1.1920929e-07f), 1.f - 1.1920929e-07f)); }
    if (sel) sel[t] = p[best];
    if (ent) ent[t] = h / logf((float)K);
}

__global__ void masks_kernel(int B, int A, int m, int order, float* masks) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * A) return;
    const int a = t % A;
    masks[t] = (a >= m - 1 && a < order - 1) ? 0.f : 1.f;
}

template <typename T>
__global__ __launch_bounds__(256) void stack_kernel(HistPtrs h, int m, int order, int64_t elems, void* out) {
    // grid (chunks, order, B)
    const int k = blockIdx.y, b = blockIdx.z;
    constexpr int V = Io<T>::VEC;
    const int64_t nvec = elems / V;
    const int64_t src = (int64_t)b * elems, dst = ((int64_t)b * order + k) * elems;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        float a[V];
        if (k < m) Io<T>::load(h.p[k], src + v * V, a);
        else {
#pragma unroll
            for (int j = 0; j < V; ++j) a[j] = 0.f;
        }
        Io<T>::store(out, dst + v * V, a);
    }
    if (blockIdx.x == 0)
        for (int64_t t = nvec * V + threadIdx.x; t < elems; t += blockDim.x)
            Io<T>::store1(out, dst + t, k < m ? Io<T>::load1(h.p[k], src + t) : 0.f);
}

}  // namespace

extern "C" {

int cs_lms_ddim_step(const CsStepArgs* args, void* stream) { return launch_step<false>(args, stream); }
int cs_lms_euler_step(const CsStepArgs* args, void* stream) { return launch_step<true>(args, stream); }

int cs_factor_probs(const CsFactorNet* n, const float* x, int x_row_stride, const float* cos_feat, int B, float* probs, void* stream) {
    if (!n) CS_FAIL(CS_E_ARG, "net is NULL");
    if (B < 0) CS_FAIL(CS_E_SHAPE, "negative batch");
    if (B == 0) return CS_OK;
    if (!x || !probs) CS_FAIL(CS_E_ARG, "x and probs are required");
    if (!n->w0 || !n->b0 || !n->w1 || !n->b1 || !n->w2 || !n->b2) CS_FAIL(CS_E_ARG, "missing weight pointer");
    if (n->in_dim < 2 || n->in_dim > 16) CS_FAIL(CS_E_SHAPE, "in_dim %d not in [2,16]", n->in_dim);
    if (n->hidden < 1 || n->hidden > 1024) CS_FAIL(CS_E_SHAPE, "hidden %d not in [1,1024]", n->hidden);
    if (n->action_dims < 1 || n->action_dims > CS_MAX_ACTION_DIMS) CS_FAIL(CS_E_SHAPE, "action_dims %d out of range", n->action_dims);
    if (n->num_actions < 1 || n->num_actions > 1024) CS_FAIL(CS_E_SHAPE, "num_actions %d out of range", n->num_actions);
    if (n->in_dim > 2 && !cos_feat) CS_FAIL(CS_E_ARG, "use_conv=True requires epsilon (factor_net_ppo.py:109-110)");
    const size_t Hp = ((size_t)n->hidden + 3) & ~(size_t)3;
    size_t lds = (16 + 2 * Hp + (size_t)n->action_dims * n->num_actions) * sizeof(float);
    if (lds > 64 * 1024) CS_FAIL(CS_E_SHAPE, "policy net too large for one workgroup (%zu B LDS)", lds);
    // one broadcast conditioning row and no per-sample features: evaluate once, write B rows
    const bool bcast = (x_row_stride == 0 && n->in_dim == 2 && B > 1);
    const int threads = n->hidden >= 128 ? 1024 : 256;
    hipLaunchKernelGGL(factor_probs_kernel, dim3(bcast ? 1 : B), dim3(threads), lds, (hipStream_t)stream, *n, x,
                       x_row_stride, cos_feat, probs, bcast ? B : 1);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_cosine_features_cfg(const void* const* hist, int m, int order, int B, int64_t elems, int dtype, const void* eps_uncond, float guidance,
                           void* eps_out, float* out, void* stream) {
    if (B <= 0 || elems <= 0) return B < 0 || elems < 0 ? CS_E_SHAPE : CS_OK;
    if (!hist || !out) CS_FAIL(CS_E_ARG, "use_conv=True requires epsilon (factor_net_ppo.py:109-110)");
    if (order < 2 || order > CS_MAX_ORDER || m < 1 || m > order) CS_FAIL(CS_E_ARG, "bad order/m (%d,%d)", order, m);
    if (eps_uncond && !eps_out) CS_FAIL(CS_E_ARG, "eps_out is required with eps_uncond (the combined eps is the history entry)");
    if (!eps_uncond && eps_out) CS_FAIL(CS_E_ARG, "eps_out without eps_uncond");
    HistPtrs h;
    for (int k = 0; k < CS_MAX_ORDER; ++k) h.p[k] = (k < m) ? hist[k] : nullptr;
    for (int k = 0; k < m; ++k) if (!h.p[k]) CS_FAIL(CS_E_ARG, "hist[%d] is NULL", k);
    dim3 grid(order - 1, B), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CS_F32) hipLaunchKernelGGL(cosine_kernel<float>, grid, block, 0, s, h, m, order, elems, out, eps_uncond, guidance, eps_out);
    else if (dtype == CS_F16) hipLaunchKernelGGL(cosine_kernel<f16>, grid, block, 0, s, h, m, order, elems, out, eps_uncond, guidance, eps_out);
    else if (dtype == CS_BF16) hipLaunchKernelGGL(cosine_kernel<bf16_tag>, grid, block, 0, s, h, m, order, elems, out, eps_uncond, guidance, eps_out);
    else CS_FAIL(CS_E_DTYPE, "bad dtype %d", dtype);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_cosine_features(const void* const* hist, int m, int order, int B, int64_t elems, int dtype, float* out, void* stream) {
    return cs_cosine_features_cfg(hist, m, order, B, elems, dtype, nullptr, 0.f, nullptr, out, stream);
}

static int launch_sample(const float* probs, const float* uni, const int64_t* idx_in, const float* av, int B, int A, int K,
                         int64_t* idx, float* actions, float* aprobs, void* stream) {
    if (B < 0 || A < 1 || K < 1) CS_FAIL(CS_E_SHAPE, "bad shape B=%d A=%d K=%d", B, A, K);
    if (B == 0) return CS_OK;
    if (!probs || !av) CS_FAIL(CS_E_ARG, "probs and action_values are required");
    int n = B * A;
    hipLaunchKernelGGL(sample_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, probs, uni, idx_in, av, B, A, K, idx, actions, aprobs);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_sample_actions(const float* probs, const float* uniforms, const float* action_values, int B, int A, int K,
                      int64_t* idx, float* actions, float* action_probs, void* stream) {
    if (B > 0 && !uniforms) CS_FAIL(CS_E_ARG, "uniforms is NULL");
    return launch_sample(probs, uniforms, nullptr, action_values, B, A, K, idx, actions, action_probs, stream);
}

int cs_gather_actions(const float* probs, const int64_t* idx, const float* action_values, int B, int A, int K,
                      float* actions, float* action_probs, void* stream) {
    if (B > 0 && !idx) CS_FAIL(CS_E_ARG, "idx is NULL");
    return launch_sample(probs, nullptr, idx, action_values, B, A, K, nullptr, actions, action_probs, stream);
}

int cs_action_probs(const float* probs, const float* actions, const float* action_values, int B, int A, int K,
                    float* selected, float* entropy, void* stream) {
    if (B < 0 || A < 1 || K < 1) CS_FAIL(CS_E_SHAPE, "bad shape");
    if (B == 0) return CS_OK;
    if (!probs || !actions || !action_values) CS_FAIL(CS_E_ARG, "probs, actions and action_values are required");
    int n = B * A;
    hipLaunchKernelGGL(action_probs_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, probs, actions, action_values, B, A, K, selected, entropy);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_step_masks(int B, int A, int m, int order, float* masks, void* stream) {
    if (B < 0 || A < 1) CS_FAIL(CS_E_SHAPE, "bad shape");
    if (B == 0) return CS_OK;
    if (!masks) CS_FAIL(CS_E_ARG, "masks is NULL");
    int n = B * A;
    hipLaunchKernelGGL(masks_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, B, A, m, order, masks);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_stack_history(const void* const* hist, int m, int order, int B, int64_t elems, int dtype, void* out, void* stream) {
    if (B <= 0 || elems <= 0) return B < 0 || elems < 0 ? CS_E_SHAPE : CS_OK;
    if (!hist || !out) CS_FAIL(CS_E_ARG, "hist and out are required");
    if (order < 1 || order > CS_MAX_ORDER || m < 0 || m > order) CS_FAIL(CS_E_ARG, "bad order/m");
    HistPtrs h;
    for (int k = 0; k < CS_MAX_ORDER; ++k) h.p[k] = (k < m) ? hist[k] : nullptr;
    const int vec = dtype == CS_F32 ? 4 : 8;
    int gx = (int)((elems / vec + 255) / 256); if (gx < 1) gx = 1; if (gx > 64) gx = 64;
    dim3 grid(gx, order, B), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CS_F32) hipLaunchKernelGGL(stack_kernel<float>, grid, block, 0, s, h, m, order, elems, out);
    else if (dtype == CS_F16) hipLaunchKernelGGL(stack_kernel<f16>, grid, block, 0, s, h, m, order, elems, out);
    else if (dtype == CS_BF16) hipLaunchKernelGGL(stack_kernel<bf16_tag>, grid, block, 0, s, h, m, order, elems, out);
    else CS_FAIL(CS_E_DTYPE, "bad dtype %d", dtype);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

}  // extern "C"
