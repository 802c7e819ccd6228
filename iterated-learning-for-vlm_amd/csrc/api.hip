// Library-level entry points: version, thread-local error text, and a device self test of the MFMA / LDS
// transpose-read fragment maps every MFMA kernel here relies on.
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void ilvlm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ilvlm_version(void) { return ILVLM_VERSION; }
extern "C" const char* ilvlm_last_error(void) { return g_err; }

namespace {
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// out[0][lane][0..3] : K-coverage probe of v_mfma_f32_16x16x32_bf16 (A = 1, B[k][n] = k): 496 everywhere
// out[1][lane][0..7] : ds_read_b64_tr_b16 x2 of a [32 k][16 col] image holding k      -> expect 8*(lane>>4) + j
// out[4][lane][0..7] : same reads of an image holding col                              -> expect lane & 15
// out[2][lane][0..3] : D of the bf16 MFMA with A[m][0] = m, B[0][n] = n+1               -> D[m][n] = m*(n+1)
// out[3][lane][0..3] : same for v_mfma_f32_16x16x4_f32
__global__ void selftest_kernel(float* out) {
    __shared__ __attribute__((aligned(16))) bf16 imgk[32 * 16];
    __shared__ __attribute__((aligned(16))) bf16 imgc[32 * 16];
    const int lane = threadIdx.x;
    for (int i = lane; i < 32 * 16; i += 64) { imgk[i] = (bf16)(float)(i / 16); imgc[i] = (bf16)(float)(i % 16); }
    __syncthreads();
    int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g = lane >> 4;
    for (int which = 0; which < 2; ++which) {
        const bf16* a = (which ? imgc : imgk) + (8 * g + q) * 16 + 4 * p;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * 16));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
        for (int j = 0; j < 8; ++j) out[((which ? 4 : 1) * 64 + lane) * 8 + j] = (float)u.v[j];
    }
    bf16x8 fa = {0, 0, 0, 0, 0, 0, 0, 0}, fb = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g == 0) { fa[0] = (bf16)(float)(lane & 15); fb[0] = (bf16)(float)((lane & 15) + 1); }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) out[(2 * 64 + lane) * 8 + j] = acc[j];
    for (int j = 0; j < 8; ++j) { fa[j] = (bf16)1.0f; fb[j] = (bf16)(float)(8 * g + j); }
    acc = (f32x4){0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) out[(0 * 64 + lane) * 8 + j] = acc[j];
    float a32 = g == 0 ? (float)(lane & 15) : 0.f, b32 = g == 0 ? (float)((lane & 15) + 1) : 0.f;
    acc = (f32x4){0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, b32, acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) out[(3 * 64 + lane) * 8 + j] = acc[j];
}
}  // namespace

extern "C" int ilvlm_selftest_fragments(float* out, void* stream) {
    ILVLM_REQUIRE(out, "selftest: null pointer");
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    ILVLM_LAUNCH_CHECK("selftest");
    return ILVLM_OK;
}
