// fp8 (OCP e4m3 / e5m2) quantisation for the fp8 MFMA GEMM path (BASELINE.json configs[4]; the reference has no
// low-precision path at all: prototype/model/image_encoder/base_transformer.py:35-48 runs fp32 nn.Linear).
// Per-tensor scaling with an amax history (delayed scaling): q = sat(x * scale), x ~ q * inv_scale.  Every quantising
// kernel also records max|x| of what it saw (atomic max on the non-negative float's bit pattern), from which the next
// step's scale is derived by ilvlm_fp8_scale_update.
#include "common.h"

namespace {

template <int FMT>
__device__ __forceinline__ unsigned pack4(float a, float b, float c, float d) { return fp8_pack4<FMT>(a, b, c, d); }

__device__ __forceinline__ void amax_commit(float m, float* amax, float* scratch) {
    m = block_max_256(m, scratch);
    if (threadIdx.x == 0) fp8_amax_raise(amax, m);
}

// dst may be null: observe only (amax of a tensor that is not quantised this step)
template <class T, int FMT>
__global__ __launch_bounds__(256) void fp8_quant_kernel(const T* __restrict__ src, unsigned char* __restrict__ dst, long n,
                                                        const float* __restrict__ scale, float* __restrict__ amax) {
    __shared__ float red[4];
    const float s = scale ? scale[0] : 1.f;
    float m = 0.f;
    const long stride = (long)gridDim.x * 256 * 8;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i + 8 <= n; i += stride) {
        f32x4 a = load4<T>(src + i), b = load4<T>(src + i + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, fmaxf(fabsf(a[j]), fabsf(b[j])));
        if (dst) {
            uint2 w;
            w.x = pack4<FMT>(a[0] * s, a[1] * s, a[2] * s, a[3] * s);
            w.y = pack4<FMT>(b[0] * s, b[1] * s, b[2] * s, b[3] * s);
            *(uint2*)(dst + i) = w;
        }
    }
    if (amax) amax_commit(m, amax, red);
}

// weights: 64 x 64 tiles of the fp32 master arena -> e4m3 in the same [rows, cols] layout (forward operand) and
// transposed [cols, rows] (input-gradient operand: dX = dY W needs W with the reduction index contiguous)
// W8P / W8TP (optional): the same bytes in the fragment order of the streaming kernel's fp8 form (gemm.hip, pack_b8_kernel):
// block (row / 16, k / 128) of 2 KiB, 16-byte chunk number 64 * ((k / 64) & 1) + (row & 15) + 16 * ((k / 16) & 3) -- a thread's 16
// bytes are one chunk, so the packed copies cost one more store each.  The packed image of a matrix starts at its arena offset.
__global__ __launch_bounds__(256) void fp8_weight_kernel(const float* __restrict__ P, unsigned char* __restrict__ W8,
                                                         unsigned char* __restrict__ W8T, unsigned char* __restrict__ W8P,
                                                         unsigned char* __restrict__ W8TP, const int* __restrict__ table,
                                                         const float* __restrict__ scale, float* __restrict__ amax) {
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) unsigned char tile[64][64 + 16];
    const int* e = table + (long)blockIdx.x * 6;
    const long off = (long)e[0] * 64;              // arena offsets are multiples of 64 elements
    const int rows = e[1], cols = e[2], slot = e[3], r0 = e[4], c0 = e[5];
    const float s = scale[slot];
    const int t = threadIdx.x, r = t >> 2, cq = (t & 3) * 16;
    const float* src = P + off + (long)(r0 + r) * cols + c0 + cq;
    float m = 0.f;
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 v = *(const f32x4*)(src + 4 * k);
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(v[j]));
        w[k] = pack4<0>(v[0] * s, v[1] * s, v[2] * s, v[3] * s);
    }
    *(uint4*)(W8 + off + (long)(r0 + r) * cols + c0 + cq) = make_uint4(w[0], w[1], w[2], w[3]);
    if (W8P) {
        const int n = r0 + r, k = c0 + cq;
        const long chunk = ((long)(n >> 4) * (cols >> 7) + (k >> 7)) * 128 + ((k >> 6) & 1) * 64 + (n & 15) + ((k >> 4) & 3) * 16;
        *(uint4*)(W8P + off + chunk * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[cq + 4 * k + j][r] = (unsigned char)(w[k] >> (8 * j));
    __syncthreads();
    // transposed tile: row c (a column of the weight), 64 consecutive original rows
    const uint4 tv = *(const uint4*)&tile[r][cq];
    *(uint4*)(W8T + off + (long)(c0 + r) * rows + r0 + cq) = tv;
    if (W8TP) {
        const int n = c0 + r, k = r0 + cq;
        const long chunk = ((long)(n >> 4) * (rows >> 7) + (k >> 7)) * 128 + ((k >> 6) & 1) * 64 + (n & 15) + ((k >> 4) & 3) * 16;
        *(uint4*)(W8TP + off + chunk * 16) = tv;
    }
    amax_commit(m, amax + slot, red);
}

// delayed scaling: hist[slot][pos] = amax_cur[slot]; scale = fmt_max / max(hist) (1 while nothing was seen).  amax_cur
// restarts at 0.9 x the value just recorded instead of 0: every wave of every quantising kernel compares its local maximum
// with amax_cur and raises it atomically when larger -- from 0 the whole first generation of waves (thousands) would hit the
// one word (12800 atomics at ~12 ns each made the fused LayerNorm 7x slower); from 0.9 x the last step's amax only genuine
// new maxima do.  A recorded amax can therefore fall by at most 10 % per step: after an outlier (one step at 100 x) the
// entries decay geometrically and the scale is back within a few per cent in HIST + ~45 steps.  (Round 2 restarted at 0.9 x
// the WINDOW maximum: the outlier then stayed the window maximum for HIST steps at a time and the decay was 10 % per HIST
// steps, ~700 steps at a 100 x too conservative scale -- found by tests/test_fp8_gpu.py's amax-jump test.)
__global__ __launch_bounds__(256) void fp8_scale_kernel(float* __restrict__ amax_cur, float* __restrict__ hist,
                                                        float* __restrict__ scale, float* __restrict__ inv_scale,
                                                        const float* __restrict__ fmt_max, int n, int hist_len, int pos) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float last = amax_cur[i];
    hist[(long)i * hist_len + pos] = last;
    float m = 0.f;
    for (int k = 0; k < hist_len; ++k) m = fmaxf(m, hist[(long)i * hist_len + k]);
    amax_cur[i] = (last > 0.f && isfinite(last)) ? 0.9f * last : 0.f;
    const float s = (m > 0.f && isfinite(m)) ? fmt_max[i] / m : 1.f;
    scale[i] = s;
    inv_scale[i] = 1.f / s;
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int ilvlm_fp8_quantize(const void* src, int src_dtype, void* dst, long n, const float* scale, float* amax, int fmt,
                                  void* stream) {
    ILVLM_REQUIRE(src && n > 0 && n % 8 == 0, "fp8_quantize: n=%ld must be a positive multiple of 8", n);
    ILVLM_REQUIRE(dst || amax, "fp8_quantize: nothing to do (no destination and no amax)");
    ILVLM_REQUIRE(fmt == 0 || fmt == 1, "fp8_quantize: fmt must be 0 (e4m3) or 1 (e5m2)");
    ILVLM_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, "fp8_quantize: alignment");
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
#define Q(T, F) hipLaunchKernelGGL((fp8_quant_kernel<T, F>), dim3((int)blocks), dim3(256), 0, S_, (const T*)src, (unsigned char*)dst, n, scale, amax)
    if (src_dtype == ILVLM_BF16) { if (fmt == 0) Q(bf16, 0); else Q(bf16, 1); }
    else if (src_dtype == ILVLM_F32) { if (fmt == 0) Q(float, 0); else Q(float, 1); }
    else ILVLM_FAIL(ILVLM_ERR_ARG, "fp8_quantize: bad source dtype %d", src_dtype);
#undef Q
    ILVLM_LAUNCH_CHECK("fp8_quantize");
    return ILVLM_OK;
}

extern "C" int ilvlm_fp8_quantize_weights(const float* params, void* w8, void* w8t, const int32_t* tile_table, int n_tiles,
                                          const float* scale, float* amax, void* stream) {
    ILVLM_REQUIRE(params && w8 && w8t && tile_table && scale && amax && n_tiles > 0, "fp8_quantize_weights: bad args");
    hipLaunchKernelGGL(fp8_weight_kernel, dim3(n_tiles), dim3(256), 0, S_, params, (unsigned char*)w8, (unsigned char*)w8t,
                       (unsigned char*)nullptr, (unsigned char*)nullptr, tile_table, scale, amax);
    ILVLM_LAUNCH_CHECK("fp8_quantize_weights");
    return ILVLM_OK;
}

// as above, plus the fragment-order copies the streaming fp8 GEMM reads (ilvlm_gemm's b_packed with fp8 compute).  Every matrix
// of the table needs rows %% 128 == 0 and cols %% 128 == 0 (the caller's check: the table holds 64 x 64 tiles only).
extern "C" int ilvlm_fp8_quantize_weights_packed(const float* params, void* w8, void* w8t, void* w8p, void* w8tp,
                                                 const int32_t* tile_table, int n_tiles, const float* scale, float* amax, void* stream) {
    ILVLM_REQUIRE(params && w8 && w8t && w8p && w8tp && tile_table && scale && amax && n_tiles > 0, "fp8_quantize_weights_packed: bad args");
    ILVLM_REQUIRE(((uintptr_t)w8p % 16) == 0 && ((uintptr_t)w8tp % 16) == 0, "fp8_quantize_weights_packed: alignment");
    hipLaunchKernelGGL(fp8_weight_kernel, dim3(n_tiles), dim3(256), 0, S_, params, (unsigned char*)w8, (unsigned char*)w8t,
                       (unsigned char*)w8p, (unsigned char*)w8tp, tile_table, scale, amax);
    ILVLM_LAUNCH_CHECK("fp8_quantize_weights_packed");
    return ILVLM_OK;
}

extern "C" int ilvlm_fp8_scale_update(float* amax_cur, float* hist, float* scale, float* inv_scale, const float* fmt_max,
                                      int n_slots, int hist_len, int pos, void* stream) {
    ILVLM_REQUIRE(amax_cur && hist && scale && inv_scale && fmt_max && n_slots > 0 && hist_len > 0 && pos >= 0 && pos < hist_len,
                  "fp8_scale_update: bad args");
    hipLaunchKernelGGL(fp8_scale_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, S_, amax_cur, hist, scale, inv_scale, fmt_max,
                       n_slots, hist_len, pos);
    ILVLM_LAUNCH_CHECK("fp8_scale_update");
    return ILVLM_OK;
}
