// 16-bit element traits shared by the dtype-generic kernels (f16 / bf16): MFMA fragment type,
// the matching v_mfma_f32_16x16x32 builtin and scalar conversions on the raw 16-bit pattern.
#pragma once
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
struct bf16_el {};   // tag type: storage is u16

template <typename T> struct El;
template <> struct El<f16> {
    typedef f16x8 frag;
    static constexpr int dtype = CS_F16;
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float tof(u16 v) { union { u16 u; f16 h; } x; x.u = v; return (float)x.h; }
    static __device__ __forceinline__ u16 fromf(float f) { union { u16 u; f16 h; } x; x.h = (f16)f; return x.u; }
};
template <> struct El<bf16_el> {
    typedef bf16x8_t frag;
    static constexpr int dtype = CS_BF16;
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float tof(u16 v) { return bf16_to_f32(v); }
    // plain cast: hipcc -O3 emits v_cvt_pk_bf16_f32 (RNE, NaN preserving) on gfx950
    static __device__ __forceinline__ u16 fromf(float f) { union { u16 u; __bf16 h; } x; x.h = (__bf16)f; return x.u; }
};

template <typename T> __device__ __forceinline__ typename El<T>::frag as_frag(u32x4 v) {
    union { u32x4 u; typename El<T>::frag f; } x; x.u = v; return x.f;
}
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b);
template <> __device__ __forceinline__ unsigned pack2<f16>(float a, float b) {
    union { f16x2 v; unsigned u; } x; x.v = f16x2{(f16)a, (f16)b}; return x.u;
}
template <> __device__ __forceinline__ unsigned pack2<bf16_el>(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    union { bf16x2_t v; unsigned u; } x; x.v = bf16x2_t{(__bf16)a, (__bf16)b}; return x.u;
}
