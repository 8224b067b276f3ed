// PPO rollout consumer arithmetic (train_ppo.py:352-427; edit_ppo/reward_model.py:404-422,484-509):
// per-image PSNR rewards of decoded images, reward -> advantage normalisation tiled over the recorded
// steps, and the clipped-surrogate loss value.  The PSNR is the only one with real data behind it
// (2 x B x 3 x 512 x 512 halfs): an HBM-bound two-stage reduction, 16 bytes per lane per load, fp32
// accumulation.  The other two are launch-latency sized ([B] and [B (n-1), A]) and run as one workgroup.
#include "ops.h"
#include <cmath>
#include "consolver_hip.h"

namespace {

constexpr int PSNR_SPLITS = 64;

// partial[b][s] = sum over the s-th slice of (pred - target)^2
template <typename T>
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const T* __restrict__ pred, const T* __restrict__ target, long n,
                                                             float* __restrict__ partial) {
    const int b = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    const long nv = n >> 3;                                   // 8-element vectors
    const long per = (nv + S - 1) / S, v0 = s * per, v1 = v0 + per < nv ? v0 + per : nv;
    const T* p = pred + (size_t)b * n;
    const T* t = target + (size_t)b * n;
    float acc = 0.f;
    for (long v = v0 + threadIdx.x; v < v1; v += 256) {
        float a[8], c[8];
        if constexpr (sizeof(T) == 2) {
            const f16x8 pa = *reinterpret_cast<const f16x8*>(p + v * 8), ta = *reinterpret_cast<const f16x8*>(t + v * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) { a[k] = (float)pa[k]; c[k] = (float)ta[k]; }
        } else {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(p + v * 8), p1 = *reinterpret_cast<const f32x4*>(p + v * 8 + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(t + v * 8), t1 = *reinterpret_cast<const f32x4*>(t + v * 8 + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] = p0[k]; a[4 + k] = p1[k]; c[k] = t0[k]; c[4 + k] = t1[k]; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = a[k] - c[k]; acc += d * d; }
    }
    if (s == S - 1)                                            // ragged tail (n not a multiple of 8)
        for (long i = (nv << 3) + threadIdx.x; i < n; i += 256) { const float d = (float)p[i] - (float)t[i]; acc += d * d; }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * S + s] = red[0] + red[1] + red[2] + red[3];
}

// psnr[b] = clamp(10 log10(1 / (mse + 1e-8)), 0, hi)
__global__ __launch_bounds__(64) void psnr_finalize_kernel(const float* __restrict__ partial, int S, long n, float hi, float* __restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float a = 0.f;
    for (int s = lane; s < S; s += 64) a += partial[(size_t)b * S + s];
    a = wave_sum(a);
    if (lane == 0) {
        const float mse = a / (float)n;
        float psnr = 10.0f * log10f(1.0f / (mse + 1e-8f));
        psnr = fmaxf(psnr, 0.f);
        if (hi > 0.f) psnr = fminf(psnr, hi);
        out[b] = psnr;
    }
}

// adv[b*(n-1) + j][a] = (r[b] - mean) / (std_unbiased + 1e-8) * 10 * masks[b*(n-1) + j][a]      (train_ppo.py:376-390)
__global__ __launch_bounds__(256) void advantages_kernel(const float* __restrict__ r, int B, int steps, int A, const float* __restrict__ masks,
                                                         float* __restrict__ out) {
    __shared__ float red[4];
    __shared__ float stat[2];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) s += r[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)B;
    __syncthreads();
    float q = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) { const float d = r[i] - mean; q += d * d; }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        stat[0] = mean;
        stat[1] = sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)(B - 1));      // B == 1 -> 0/0 = NaN, like torch.std
    }
    __syncthreads();
    const float inv = 1.0f / (stat[1] + 1e-8f);
    const long total = (long)B * steps * A;
    for (long i = threadIdx.x; i < total; i += 256) {
        const int b = (int)(i / ((long)steps * A));
        out[i] = (r[b] - stat[0]) * inv * 10.0f * masks[i];
    }
}

// loss = -mean_{r,a} min(adv ratio, adv clip(ratio)) - coef * mean(entropy);  ratio[r] = exp(sum_a log(p+1e-9) - sum_a log(q+1e-9))
__global__ __launch_bounds__(256) void ppo_loss_kernel(const float* __restrict__ cur, const float* __restrict__ old, const float* __restrict__ ent,
                                                       const float* __restrict__ adv, int R, int A, float clip, float coef, float* __restrict__ out) {
    __shared__ float red[2][4];
    float pol = 0.f, en = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) {
        float lp = 0.f, lq = 0.f;
        for (int a = 0; a < A; ++a) { lp += logf(cur[(size_t)r * A + a] + 1e-9f); lq += logf(old[(size_t)r * A + a] + 1e-9f); en += ent[(size_t)r * A + a]; }
        const float ratio = expf(lp - lq), cl = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
        for (int a = 0; a < A; ++a) { const float ad = adv[(size_t)r * A + a]; pol += fminf(ad * ratio, ad * cl); }
    }
    pol = wave_sum(pol); en = wave_sum(en);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = pol; red[1][threadIdx.x >> 6] = en; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float n = (float)R * (float)A;
        out[0] = -(red[0][0] + red[0][1] + red[0][2] + red[0][3]) / n - coef * (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / n;
    }
}


// ------------------------------------------------------------------------------------------------
// PPO policy update (train_ppo.py:404-437): gradient of the clipped-surrogate + entropy loss with
// respect to the factor net's 75 k parameters, global-norm clip and AdamW.  Hand-derived backward
// (formulas: oracle/solver_oracle.py::ppo_policy_grads, pinned against torch autograd on the
// reference's FactorNetPPO).  R = B (n-1) <= a few thousand rows, hidden 256: microseconds of work,
// written for determinism (no atomics: per-row pass, then one thread per weight sums over rows).
// ------------------------------------------------------------------------------------------------
constexpr int POL_MAX_IN = 16, POL_MAX_H = 1024, POL_MAX_AK = 4096, POL_MAX_A = 8;

struct PolicyRowArgs {
    CsFactorNet net;
    const float* x; const float* cosf; const float* actions; const float* action_values; const float* old_probs; const float* adv;
    int R; float clip, coef;
    float *h0, *h1, *h2, *dlog, *d2, *d1, *part;     // workspace rows
};

// one workgroup (256 threads) per row: forward, loss terms, dlogits, d(hidden) ; everything stays in LDS
__global__ __launch_bounds__(256) void policy_row_kernel(PolicyRowArgs a) {
    extern __shared__ float sm[];
    const CsFactorNet& n = a.net;
    const int IN = n.in_dim, H = n.hidden, A = n.action_dims, K = n.num_actions, AK = A * K;
    float* h0 = sm; float* h1 = h0 + POL_MAX_IN; float* h2 = h1 + H; float* lg = h2 + H; float* d2 = lg + AK; float* d1 = d2 + H;
    float* sc = d1 + H;                                // [A]: s, H, (scratch)
    __shared__ float rowstat[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < IN) h0[tid] = tid < 2 ? a.x[(size_t)r * 2 + tid] * n.input_scale : a.cosf[(size_t)r * (IN - 2) + tid - 2];
    __syncthreads();
    for (int j = tid; j < H; j += 256) {
        float v = n.b0[j];
        for (int i = 0; i < IN; ++i) v += n.w0[(size_t)j * IN + i] * h0[i];
        h1[j] = fmaxf(v, 0.f);
    }
    __syncthreads();
    for (int j = w; j < H; j += 4) {                   // one wave per output, lanes over the reduction
        float v = 0.f;
        for (int k = lane; k < H; k += 64) v += n.w1[(size_t)j * H + k] * h1[k];
        v = wave_sum(v);
        if (lane == 0) h2[j] = fmaxf(v + n.b1[j], 0.f);
    }
    __syncthreads();
    for (int j = w; j < AK; j += 4) {
        float v = 0.f;
        for (int k = lane; k < H; k += 64) v += n.w2[(size_t)j * H + k] * h2[k];
        v = wave_sum(v);
        if (lane == 0) lg[j] = (v + n.b2[j]) * n.inv_temperature;
    }
    __syncthreads();
    // softmax per action dim (one wave per dim), selected prob, entropy
    for (int ad = w; ad < A; ad += 4) {
        float mx = -INFINITY;
        for (int k = lane; k < K; k += 64) mx = fmaxf(mx, lg[ad * K + k]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int k = lane; k < K; k += 64) { const float e = expf(lg[ad * K + k] - mx); lg[ad * K + k] = e; se += e; }
        se = wave_sum(se);
        const float inv = 1.0f / se;
        float ent = 0.f;
        for (int k = lane; k < K; k += 64) { const float p = lg[ad * K + k] * inv; lg[ad * K + k] = p; ent -= p * logf(fminf(fmaxf(p, 1.1920929e-07f), 1.f - 1.1920929e-07f)); }
        ent = wave_sum(ent);
        // nearest grid bin (first minimum), factor_net_ppo.py:174-178
        const float act = a.actions[(size_t)r * A + ad];
        float bd = INFINITY; int bi = 0;
        for (int k = lane; k < K; k += 64) { const float d = fabsf(act - a.action_values[ad * K + k]); if (d < bd) { bd = d; bi = k; } }
        for (int off = 32; off; off >>= 1) {
            const float od = __shfl_xor(bd, off, 64); const int oi = __shfl_xor(bi, off, 64);
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        if (lane == 0) { sc[ad] = lg[ad * K + bi]; sc[POL_MAX_A + ad] = ent; sc[2 * POL_MAX_A + ad] = (float)bi; }
    }
    __syncthreads();
    if (tid == 0) {
        float lp = 0.f, lq = 0.f, ent = 0.f;
        for (int ad = 0; ad < A; ++ad) { lp += logf(sc[ad] + 1e-9f); lq += logf(a.old_probs[(size_t)r * A + ad] + 1e-9f); ent += sc[POL_MAX_A + ad]; }
        const float rho = expf(lp - lq), cl = fminf(fmaxf(rho, 1.f - a.clip), 1.f + a.clip);
        float pol = 0.f, dsum = 0.f;
        for (int ad = 0; ad < A; ++ad) {
            const float ad_v = a.adv[(size_t)r * A + ad];
            pol += fminf(ad_v * rho, ad_v * cl);
            const bool clipped_is_min = (rho > 1.f + a.clip && ad_v > 0.f) || (rho < 1.f - a.clip && ad_v < 0.f);
            dsum += clipped_is_min ? 0.f : -ad_v;
        }
        const float ra = (float)a.R * (float)A;
        rowstat[0] = dsum / ra * rho;                  // d loss / d (sum_a log(s_a + 1e-9))
        rowstat[1] = -a.coef / (ra * logf((float)K));  // entropy weight
        a.part[(size_t)r * 2] = pol; a.part[(size_t)r * 2 + 1] = ent / logf((float)K);
    }
    __syncthreads();
    for (int j = tid; j < AK; j += 256) {
        const int ad = j / K, k = j - ad * K;
        const float p = lg[j], s = sc[ad], Hh = sc[POL_MAX_A + ad];
        const float onehot = (k == (int)sc[2 * POL_MAX_A + ad]) ? 1.f : 0.f;
        float dz = rowstat[0] * (s / (s + 1e-9f)) * (onehot - p);
        dz += rowstat[1] * (-p * (logf(fmaxf(p, 1.1920929e-07f)) + Hh));
        lg[j] = dz * n.inv_temperature;
    }
    __syncthreads();
    for (int k = tid; k < H; k += 256) {
        float v = 0.f;
        for (int j = 0; j < AK; ++j) v += lg[j] * n.w2[(size_t)j * H + k];
        d2[k] = h2[k] > 0.f ? v : 0.f;
    }
    __syncthreads();
    for (int k = tid; k < H; k += 256) {
        float v = 0.f;
        for (int j = 0; j < H; ++j) v += d2[j] * n.w1[(size_t)j * H + k];
        d1[k] = h1[k] > 0.f ? v : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < IN; i += 256) a.h0[(size_t)r * IN + i] = h0[i];
    for (int k = tid; k < H; k += 256) {
        a.h1[(size_t)r * H + k] = h1[k]; a.h2[(size_t)r * H + k] = h2[k];
        a.d2[(size_t)r * H + k] = d2[k]; a.d1[(size_t)r * H + k] = d1[k];
    }
    for (int j = tid; j < AK; j += 256) a.dlog[(size_t)r * AK + j] = lg[j];
}

// gw[n][k] = sum_r G[r][n] X[r][k] ; gb[n] = sum_r G[r][n]   (one thread per weight, rows in order: deterministic)
__global__ __launch_bounds__(256) void outer_sum_kernel(const float* __restrict__ G, const float* __restrict__ X, int R, int N, int Kd,
                                                        float* __restrict__ gw, float* __restrict__ gb) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * Kd) return;
    const int nn = (int)(i / Kd), k = (int)(i - (long)nn * Kd);
    float v = 0.f, b = 0.f;
    for (int r = 0; r < R; ++r) { const float g = G[(size_t)r * N + nn]; v += g * X[(size_t)r * Kd + k]; b += g; }
    gw[i] = v;
    if (k == 0) gb[nn] = b;
}

__global__ __launch_bounds__(256) void policy_loss_kernel(const float* __restrict__ part, int R, int A, float coef, float* __restrict__ loss) {
    __shared__ float red[2][4];
    float p = 0.f, e = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) { p += part[(size_t)r * 2]; e += part[(size_t)r * 2 + 1]; }
    p = wave_sum(p); e = wave_sum(e);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = p; red[1][threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ra = (float)R * (float)A;
        loss[0] = -(red[0][0] + red[0][1] + red[0][2] + red[0][3]) / ra - coef * (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / ra;
    }
}

// total = ||g||_2 ; g *= min(max_norm / (total + 1e-6), 1)      (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(1024) void clip_grad_norm_kernel(float* __restrict__ g, long n, float max_norm, float* __restrict__ total_out) {
    __shared__ float red[16];
    __shared__ float coef;
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) s += g[i] * g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        t = sqrtf(t);
        if (total_out) total_out[0] = t;
        coef = fminf(max_norm / (t + 1e-6f), 1.0f);
    }
    __syncthreads();
    if (coef < 1.0f) for (long i = threadIdx.x; i < n; i += 1024) g[i] *= coef;
}

// torch.optim.AdamW: p *= 1 - lr wd ; m, v moments ; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

}  // namespace

extern "C" {

size_t cs_psnr_workspace_bytes(int batch) { return (size_t)(batch > 0 ? batch : 0) * PSNR_SPLITS * sizeof(float) + 256; }

int cs_image_psnr(const void* pred, const void* target, int B, int64_t elems, int dtype, float clamp_hi, float* out, void* workspace,
                  size_t workspace_bytes, void* stream) {
    if (B < 0 || elems < 0) CS_FAIL(CS_E_ARG, "negative size");
    if (B == 0) return CS_OK;
    if (elems == 0) CS_FAIL(CS_E_SHAPE, "psnr of empty images");
    if (!pred || !target || !out || !workspace) CS_FAIL(CS_E_ARG, "null pointer");
    if (workspace_bytes < cs_psnr_workspace_bytes(B)) CS_FAIL(CS_E_ARG, "psnr: workspace too small");
    if (dtype != CS_F16 && dtype != CS_F32) CS_FAIL(CS_E_UNSUPPORTED, "psnr: dtype must be f16 or f32");
    if (((uintptr_t)pred | (uintptr_t)target) & 15 || (elems * (dtype == CS_F16 ? 2 : 4)) % 16)
        CS_FAIL(CS_E_ARG, "psnr: images must be 16-byte aligned with a 16-byte multiple of bytes per image");
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    if (dtype == CS_F16) hipLaunchKernelGGL(sqdiff_partial_kernel<f16>, dim3(PSNR_SPLITS, B), dim3(256), 0, s, (const f16*)pred, (const f16*)target, (long)elems, partial);
    else hipLaunchKernelGGL(sqdiff_partial_kernel<float>, dim3(PSNR_SPLITS, B), dim3(256), 0, s, (const float*)pred, (const float*)target, (long)elems, partial);
    hipLaunchKernelGGL(psnr_finalize_kernel, dim3(B), dim3(64), 0, s, partial, PSNR_SPLITS, (long)elems, clamp_hi, out);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_ppo_advantages(const float* rewards, int B, int recorded_steps, int A, const float* masks, float* out, void* stream) {
    if (B < 0 || recorded_steps < 0 || A < 0) CS_FAIL(CS_E_ARG, "negative size");
    if (B == 0 || recorded_steps == 0 || A == 0) return CS_OK;
    if (!rewards || !masks || !out) CS_FAIL(CS_E_ARG, "null pointer");
    hipLaunchKernelGGL(advantages_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rewards, B, recorded_steps, A, masks, out);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_ppo_loss(const float* curr_probs, const float* old_probs, const float* entropy, const float* advantages, int R, int A, float clip_range,
                float entropy_coef, float* loss, void* stream) {
    if (R <= 0 || A <= 0) CS_FAIL(CS_E_SHAPE, "ppo_loss: empty batch");
    if (!curr_probs || !old_probs || !entropy || !advantages || !loss) CS_FAIL(CS_E_ARG, "null pointer");
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, curr_probs, old_probs, entropy, advantages, R, A, clip_range,
                       entropy_coef, loss);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

static size_t policy_param_count(const CsFactorNet* n) {
    const size_t H = n->hidden, IN = n->in_dim, AK = (size_t)n->action_dims * n->num_actions;
    return H * IN + H + H * H + H + AK * H + AK;
}
size_t cs_policy_param_count(const CsFactorNet* net) { return net ? policy_param_count(net) : 0; }

size_t cs_policy_workspace_bytes(const CsFactorNet* n, int R) {
    if (!n || R <= 0) return 0;
    const size_t H = n->hidden, AK = (size_t)n->action_dims * n->num_actions;
    return ((size_t)R * (n->in_dim + 4 * H + AK + 2)) * sizeof(float) + 256;
}

int cs_ppo_policy_grads(const CsFactorNet* net, const float* x, const float* cos_feat, const float* actions, const float* action_values,
                        const float* old_probs, const float* advantages, int R, float clip_range, float entropy_coef, float* grads,
                        float* loss, void* workspace, size_t workspace_bytes, void* stream) {
    if (!net || !x || !actions || !action_values || !old_probs || !advantages || !grads || !workspace) CS_FAIL(CS_E_ARG, "null pointer");
    if (R <= 0) CS_FAIL(CS_E_SHAPE, "policy_grads: empty batch");
    const int IN = net->in_dim, H = net->hidden, A = net->action_dims, K = net->num_actions, AK = A * K;
    if (IN < 2 || IN > POL_MAX_IN || H < 1 || H > POL_MAX_H || A < 1 || A > POL_MAX_A || AK > POL_MAX_AK) CS_FAIL(CS_E_SHAPE, "policy_grads: net dims out of range");
    if (IN > 2 && !cos_feat) CS_FAIL(CS_E_ARG, "policy_grads: cos_feat required when in_dim > 2");
    if (workspace_bytes < cs_policy_workspace_bytes(net, R)) CS_FAIL(CS_E_ARG, "policy_grads: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    PolicyRowArgs a;
    a.net = *net; a.x = x; a.cosf = cos_feat; a.actions = actions; a.action_values = action_values; a.old_probs = old_probs; a.adv = advantages;
    a.R = R; a.clip = clip_range; a.coef = entropy_coef;
    float* ws = (float*)workspace;
    a.h0 = ws; ws += (size_t)R * IN; a.h1 = ws; ws += (size_t)R * H; a.h2 = ws; ws += (size_t)R * H; a.dlog = ws; ws += (size_t)R * AK;
    a.d2 = ws; ws += (size_t)R * H; a.d1 = ws; ws += (size_t)R * H; a.part = ws;
    const size_t lds = (size_t)(POL_MAX_IN + 4 * H + AK + 3 * POL_MAX_A) * sizeof(float);
    hipLaunchKernelGGL(policy_row_kernel, dim3(R), dim3(256), lds, s, a);
    // grads packed in state-dict order: w0 [H, IN], b0 [H], w1 [H, H], b1 [H], w2 [AK, H], b2 [AK]
    float* g_w0 = grads; float* g_b0 = g_w0 + (size_t)H * IN; float* g_w1 = g_b0 + H; float* g_b1 = g_w1 + (size_t)H * H;
    float* g_w2 = g_b1 + H; float* g_b2 = g_w2 + (size_t)AK * H;
    hipLaunchKernelGGL(outer_sum_kernel, dim3((H * IN + 255) / 256), dim3(256), 0, s, a.d1, a.h0, R, H, IN, g_w0, g_b0);
    hipLaunchKernelGGL(outer_sum_kernel, dim3((H * H + 255) / 256), dim3(256), 0, s, a.d2, a.h1, R, H, H, g_w1, g_b1);
    hipLaunchKernelGGL(outer_sum_kernel, dim3((AK * H + 255) / 256), dim3(256), 0, s, a.dlog, a.h2, R, AK, H, g_w2, g_b2);
    if (loss) hipLaunchKernelGGL(policy_loss_kernel, dim3(1), dim3(256), 0, s, a.part, R, A, entropy_coef, loss);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_clip_grad_norm(float* grads, int64_t n, float max_norm, float* total_norm, void* stream) {
    if (n < 0) CS_FAIL(CS_E_ARG, "negative size");
    if (n == 0) return CS_OK;
    if (!grads) CS_FAIL(CS_E_ARG, "null pointer");
    hipLaunchKernelGGL(clip_grad_norm_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, grads, (long)n, max_norm, total_norm);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr, float beta1, float beta2,
                  float eps, float weight_decay, void* stream) {
    if (n < 0 || step < 1) CS_FAIL(CS_E_ARG, "bad size / step");
    if (n == 0) return CS_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) CS_FAIL(CS_E_ARG, "null pointer");
    const float bc1 = (float)(1.0 - pow((double)beta1, step)), bc2s = (float)sqrt(1.0 - pow((double)beta2, step));
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (long)n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2s);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

}  // extern "C"
