// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the CLIP+FDT hot path.
// wave = 64 lanes everywhere; no multi-backend macros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

#include "../../include/ilvlm_hip.h"

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

// ---- error plumbing (host) -----------------------------------------------------------
void ilvlm_set_error(const char* fmt, ...);
#define ILVLM_FAIL(code, ...)          \
    do {                               \
        ilvlm_set_error(__VA_ARGS__);  \
        return (code);                 \
    } while (0)
#define ILVLM_REQUIRE(cond, ...) \
    do {                         \
        if (!(cond)) ILVLM_FAIL(ILVLM_ERR_ARG, __VA_ARGS__); \
    } while (0)
#define ILVLM_LAUNCH_CHECK(name)                                                      \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) ILVLM_FAIL((int)e__, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// ---- scalar conversions --------------------------------------------------------------
template <class T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return (float)v; }
template <class T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }

// 4-wide vector access; requires 4-element alignment of the address
template <class T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
    bf16x4 v = *(const bf16x4*)p;
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
template <class T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, f32x4 v) {
    bf16x4 r = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *(bf16x4*)p = r;
}

// ---- wave / block reductions ---------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// reduce over the 16 lanes that share (lane >> 4)
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for blockDim.x == 256 (4 waves); scratch must hold >= 4 floats. All threads get the result.
__device__ __forceinline__ float block_sum_256(float v, float* scratch) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}
__device__ __forceinline__ float block_max_256(float v, float* scratch) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
}

// ---- activations (fp32 math) ---------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// QuickGELU x*sigmoid(1.702x) (reference image_encoder/base_transformer.py:24-26) and its derivative
__device__ __forceinline__ float quick_gelu(float x) { return x * sigmoidf_(1.702f * x); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
    float s = sigmoidf_(1.702f * x);
    return s * (1.0f + 1.702f * x * (1.0f - s));
}
// exact-erf GELU (nn.GELU() default, reference clip_fdt.py:89) and its derivative
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// row remap used for "token stream" layouts: compact row r of a [B, group, C] tensor lives at
// stream row (r / group) * (group + skip) + skip + r % group of a [B, group + skip, C] tensor.
__host__ __device__ __forceinline__ long map_row(long r, int group, int skip) {
    return group > 0 ? (r / group) * (long)(group + skip) + skip + (r % group) : r;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---- fp8 (OCP) helpers shared by fp8.hip and the kernels that emit an fp8 copy of their output ------------------
// FMT 0 = e4m3 (max 448), 1 = e5m2 (max 57344); saturating
template <int FMT>
__device__ __forceinline__ unsigned fp8_pack4(float a, float b, float c, float d) {
    const float mx = FMT == 0 ? 448.f : 57344.f;
    a = fminf(fmaxf(a, -mx), mx); b = fminf(fmaxf(b, -mx), mx);
    c = fminf(fmaxf(c, -mx), mx); d = fminf(fmaxf(d, -mx), mx);
    int w = 0;
    if (FMT == 0) {
        w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    } else {
        w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
    }
    return (unsigned)w;
}
__device__ __forceinline__ unsigned fp8_pack4_fmt(int fmt, float a, float b, float c, float d) {
    return fmt == 0 ? fp8_pack4<0>(a, b, c, d) : fp8_pack4<1>(a, b, c, d);
}
// raise *amax to m (m >= 0).  Thousands of waves raising ONE word serialise at ~12 ns per atomic, so only a wave that would
// actually raise the maximum issues one (the plain read may be stale; the atomic max keeps the result exact).
__device__ __forceinline__ void fp8_amax_raise(float* amax, float m) {
    if (m > 0.f && m > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax((unsigned*)amax, __float_as_uint(m));
}
