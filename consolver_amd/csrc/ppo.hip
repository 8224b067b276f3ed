// PPO rollout consumer arithmetic (train_ppo.py:352-427; edit_ppo/reward_model.py:404-422,484-509):
// per-image PSNR rewards of decoded images, reward -> advantage normalisation tiled over the recorded
// steps, and the clipped-surrogate loss value.  The PSNR is the only one with real data behind it
// (2 x B x 3 x 512 x 512 halfs): an HBM-bound two-stage reduction, 16 bytes per lane per load, fp32
// accumulation.  The other two are launch-latency sized ([B] and [B (n-1), A]) and run as one workgroup.
#include "ops.h"
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

}  // extern "C"
