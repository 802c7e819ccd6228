// LayerNorm forward / backward: one 64-lane wave per row, float4 (or bf16x4) vector access, the whole row
// cached in registers (cols <= 1024), wavefront-shuffle reductions; dgamma/dbeta are reduced per workgroup in
// registers + LDS and flushed with one fp32 atomic per column per workgroup.  HBM-bound.
// Replaces nn.LayerNorm call sites (reference base_transformer.py:10-18, visual_transformer.py:66,73,
// text_transformer.py:238, clip_fdt.py:86-92).
#include "common.h"

namespace {

constexpr int MAXV = 4;   // 4 x (64 lanes x 4) = 1024 columns max (ViT-L width)

template <class TX, class TY>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TY* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long rows, int cols,
                                                     float eps, int group, int skip, unsigned char* __restrict__ y8,
                                                     const float* __restrict__ q_scale, float* __restrict__ q_amax) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float qs = (y8 && q_scale) ? q_scale[0] : 1.f;       // fp8 (e4m3) copy of the output for the fp8 GEMMs
    float qm = 0.f;
    const TX* xr = x + map_row(row, group, skip) * (long)cols;
    // Every load of the row is issued before the first is used, without a branch around it: behind `if (c < cols)` the
    // compiler sinks the consumer into the branch and waits for each load in turn (3-4 memory round trips per row instead
    // of one).  Chunks past the row are loaded from the row's last chunk and zeroed by a select.
    f32x4 v[MAXV], gm[MAXV], bt[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 256 + lane * 4, cc = c < cols ? c : cols - 4;
        v[i] = load4<TX>(xr + cc);
        gm[i] = *(const f32x4*)(gamma + cc);
        bt[i] = *(const f32x4*)(beta + cc);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c >= cols) v[i] = (f32x4){0, 0, 0, 0};
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mu = wave_sum(s) / cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = i * 256 + lane * 4;
        if (c < cols) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = v[i][j] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / cols + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    TY* yr = y + row * (long)cols;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = i * 256 + lane * 4;
        if (c < cols) {
            const f32x4 g = gm[i], b = bt[i];
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
            if (y) store4<TY>(yr + c, o);
            if (q_amax) qm = fmaxf(qm, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
            if (y8) *(unsigned*)(y8 + row * (long)cols + c) = fp8_pack4<0>(o[0] * qs, o[1] * qs, o[2] * qs, o[3] * qs);
        }
    }
    if (q_amax) {
        qm = wave_max(qm);
        if (lane == 0) fp8_amax_raise(q_amax, qm);
    }
}

// ACT: 0 none, 3 quickgelu', 4 gelu_erf' applied to the low-precision copy only
template <class TDY, class TX, class TLP, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ dx_f32, TLP* __restrict__ dx_lp, int act,
                                                     const TLP* __restrict__ act_aux, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, long rows, int cols, int group, int skip,
                                                     float* __restrict__ ws, unsigned char* __restrict__ dx8,
                                                     const float* __restrict__ q_scale, float* __restrict__ q_amax) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float qs = (dx8 && q_scale) ? q_scale[0] : 1.f;      // fp8 (e5m2) copy of the low-precision gradient
    float qm = 0.f;
    f32x4 ag[NV], ab[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { ag[i] = (f32x4){0, 0, 0, 0}; ab[i] = (f32x4){0, 0, 0, 0}; }
    f32x4 gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = i * 256 + lane * 4;
        if (c < cols) gm[i] = *(const f32x4*)(gamma + c);
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const long srow = map_row(row, group, skip);
        const TX* xr = x + srow * (long)cols;
        const TDY* dyr = dy + row * (long)cols;
        // all loads of the row first, branch-free (see ln_fwd_kernel): x, dy and the residual gradient; chunks past the row
        // read the row's last chunk and are zeroed by a select
        f32x4 xh[NV], gy[NV], dr[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 256 + lane * 4, cc = c < cols ? c : cols - 4;
            xh[i] = load4<TX>(xr + cc);
            gy[i] = load4<TDY>(dyr + cc);
        }
        if (dres) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 256 + lane * 4, cc = c < cols ? c : cols - 4;
                dr[i] = *(const f32x4*)(dres + srow * (long)cols + cc);
            }
        }
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 256 + lane * 4;
            const f32x4 xv = xh[i];
            f32x4 d = gy[i];
            if (c >= cols) d = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float h = (xv[j] - mu) * rs;
                xh[i][j] = h;
                ag[i][j] += d[j] * h;
                ab[i][j] += d[j];
                float t = d[j] * gm[i][j];
                gy[i][j] = t;
                s1 += t;
                s2 += t * h;
            }
        }
        s1 = wave_sum(s1) / cols;
        s2 = wave_sum(s2) / cols;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = i * 256 + lane * 4;
            if (c < cols) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rs * (gy[i][j] - s1 - xh[i][j] * s2);
                if (dres) o += dr[i];
                if (dx_f32) *(f32x4*)(dx_f32 + srow * (long)cols + c) = o;
                if (dx_lp || dx8 || q_amax) {      // the low-precision gradient; with dx8 alone only its fp8 copy is kept
                    if (act) {
                        const f32x4 u = load4<TLP>(act_aux + row * (long)cols + c);     // FDT head only: not prefetched
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            o[j] *= act == ILVLM_ACT_QUICKGELU_BWD ? quick_gelu_grad(u[j]) : gelu_erf_grad(u[j]);
                    }
                    if (dx_lp) store4<TLP>(dx_lp + srow * (long)cols + c, o);
                    if (q_amax) qm = fmaxf(qm, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                    if (dx8) *(unsigned*)(dx8 + srow * (long)cols + c) = fp8_pack4<1>(o[0] * qs, o[1] * qs, o[2] * qs, o[3] * qs);
                }
            }
        }
    }
    if (q_amax) {
        qm = wave_max(qm);
        if (lane == 0) fp8_amax_raise(q_amax, qm);
    }
    // workgroup reduction of the per-wave column partials, then one atomic per column
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = i * 256 + lane * 4;
            if (i * 256 >= cols) break;   // uniform
            __syncthreads();
            red[wave][lane] = pass == 0 ? ag[i] : ab[i];
            __syncthreads();
            if (wave == 0 && c < cols) {
                f32x4 t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
                if (ws) {   // stage 1 of the two-stage reduction: this workgroup's partial row
                    *(f32x4*)(ws + ((long)blockIdx.x * 2 + pass) * cols + c) = t;
                } else {
                    float* dst = (pass == 0 ? dgamma : dbeta) + c;
#pragma unroll
                    for (int j = 0; j < 4; ++j) atomicAdd(dst + j, t[j]);
                }
            }
        }
    }
}

// stage 2: dgamma[c] += sum_b ws[b][0][c], dbeta[c] += sum_b ws[b][1][c].  16 columns x 16 partial-row lanes per
// workgroup so the loads of one column are spread over 16 threads and stay independent (latency-bound otherwise).
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ ws, int nblocks, int cols,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[16][17];
    const int cx = threadIdx.x & 15, by = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    const int pass = blockIdx.y;
    float s = 0.f;
    if (c < cols) {
#pragma unroll 4
        for (int b = by; b < nblocks; b += 16) s += ws[((long)b * 2 + pass) * cols + c];
    }
    red[by][cx] = s;
    __syncthreads();
    if (by == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cx];
        float* dst = pass == 0 ? dgamma : dbeta;
        dst[c] += t;
    }
}

// stage 2 for many LayerNorms at once: slot z holds the partial rows of one backward launch (ws + z * stride); its sums
// go to the pointers dst[2z] (dgamma) and dst[2z+1] (dbeta).  One launch instead of one per LayerNorm.
__global__ __launch_bounds__(256) void ln_bwd_reduce_batched_kernel(const float* __restrict__ ws, long stride, int nblocks,
                                                                    int cols, float* const* __restrict__ dst) {
    __shared__ float red[16][17];
    const int cx = threadIdx.x & 15, by = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    const int pass = blockIdx.y;
    const float* w = ws + (long)blockIdx.z * stride;
    float s = 0.f;
    if (c < cols) {
#pragma unroll 4
        for (int b = by; b < nblocks; b += 16) s += w[((long)b * 2 + pass) * cols + c];
    }
    red[by][cx] = s;
    __syncthreads();
    if (by == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cx];
        dst[2 * blockIdx.z + pass][c] += t;
    }
}

}  // namespace

extern "C" int ilvlm_layernorm_bwd_reduce_batched(const float* ws, long slot_stride, int n_slots, long rows, int ws_blocks,
                                                  int cols, float* const* grad_ptrs, void* stream) {
    ILVLM_REQUIRE(ws && grad_ptrs && n_slots > 0 && rows > 0 && ws_blocks > 0 && cols > 0, "layernorm_bwd_reduce_batched: bad args");
    ILVLM_REQUIRE(slot_stride >= 2L * ws_blocks * cols, "layernorm_bwd_reduce_batched: slot stride %ld < 2 * %d * %d", slot_stride,
                  ws_blocks, cols);
    ILVLM_REQUIRE(n_slots <= 65535, "layernorm_bwd_reduce_batched: too many slots");
    int blocks = ceil_div(rows, 4);              // what the deferred backward launches used (ilvlm_layernorm_bwd)
    if (blocks > ws_blocks) blocks = ws_blocks;
    hipLaunchKernelGGL(ln_bwd_reduce_batched_kernel, dim3(ceil_div(cols, 16), 2, n_slots), dim3(256), 0, (hipStream_t)stream, ws,
                       slot_stride, blocks, cols, grad_ptrs);
    ILVLM_LAUNCH_CHECK("layernorm_bwd_reduce_batched");
    return ILVLM_OK;
}

extern "C" int ilvlm_layernorm_fwd_q8(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                                      float* mean, float* rstd, long rows, int cols, float eps, int in_group, int in_skip,
                                      void* y8, const float* q_scale, float* q_amax, void* stream);
extern "C" int ilvlm_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                                   float* mean, float* rstd, long rows, int cols, float eps, int in_group, int in_skip,
                                   void* stream) {
    return ilvlm_layernorm_fwd_q8(x, x_dtype, gamma, beta, y, y_dtype, mean, rstd, rows, cols, eps, in_group, in_skip, nullptr,
                                  nullptr, nullptr, stream);
}
extern "C" int ilvlm_layernorm_fwd_q8(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                                      float* mean, float* rstd, long rows, int cols, float eps, int in_group, int in_skip,
                                      void* y8, const float* q_scale, float* q_amax, void* stream) {
    ILVLM_REQUIRE(!y8 || (q_scale && in_group == 0), "layernorm_fwd_q8: the fp8 copy needs a scale and compact rows");
    ILVLM_REQUIRE(x && gamma && beta && (y || y8) && mean && rstd, "layernorm_fwd: null pointer");
    ILVLM_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= MAXV * 256, "layernorm_fwd: cols=%d must be a multiple of 4 and <= %d",
                  cols, MAXV * 256);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ceil_div(rows, 4)), block(256);
#define LN_FWD(TX, TY)                                                                                               \
    hipLaunchKernelGGL((ln_fwd_kernel<TX, TY>), grid, block, 0, s, (const TX*)x, gamma, beta, (TY*)y, mean, rstd, rows, \
                       cols, eps, in_group, in_skip, (unsigned char*)y8, q_scale, q_amax)
    if (x_dtype == ILVLM_F32 && y_dtype == ILVLM_F32) LN_FWD(float, float);
    else if (x_dtype == ILVLM_F32 && y_dtype == ILVLM_BF16) LN_FWD(float, bf16);
    else if (x_dtype == ILVLM_BF16 && y_dtype == ILVLM_BF16) LN_FWD(bf16, bf16);
    else if (x_dtype == ILVLM_BF16 && y_dtype == ILVLM_F32) LN_FWD(bf16, float);
    else ILVLM_FAIL(ILVLM_ERR_ARG, "layernorm_fwd: bad dtypes %d %d", x_dtype, y_dtype);
#undef LN_FWD
    ILVLM_LAUNCH_CHECK("layernorm_fwd");
    return ILVLM_OK;
}

extern "C" int ilvlm_layernorm_bwd_q8(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                                      const float* rstd, const float* gamma, const float* dres, float* dx_f32, void* dx_lp,
                                      int dx_lp_dtype, int act, const void* act_aux, float* dgamma, float* dbeta, long rows,
                                      int cols, int group, int skip, float* ws, int ws_blocks, void* dx8, const float* q_scale,
                                      float* q_amax, void* stream);
extern "C" int ilvlm_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                                   const float* rstd, const float* gamma, const float* dres, float* dx_f32, void* dx_lp,
                                   int dx_lp_dtype, int act, const void* act_aux, float* dgamma, float* dbeta, long rows,
                                   int cols, int group, int skip, float* ws, int ws_blocks, void* stream) {
    return ilvlm_layernorm_bwd_q8(dy, dy_dtype, x, x_dtype, mean, rstd, gamma, dres, dx_f32, dx_lp, dx_lp_dtype, act, act_aux, dgamma,
                                  dbeta, rows, cols, group, skip, ws, ws_blocks, nullptr, nullptr, nullptr, stream);
}
extern "C" int ilvlm_layernorm_bwd_q8(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                                      const float* rstd, const float* gamma, const float* dres, float* dx_f32, void* dx_lp,
                                      int dx_lp_dtype, int act, const void* act_aux, float* dgamma, float* dbeta, long rows,
                                      int cols, int group, int skip, float* ws, int ws_blocks, void* dx8, const float* q_scale,
                                      float* q_amax, void* stream) {
    ILVLM_REQUIRE(!(dx8 || q_amax) || (group == 0 && (!dx8 || q_scale)),
                  "layernorm_bwd_q8: the fp8 copy needs compact rows and a scale");
    ILVLM_REQUIRE(dx_lp || !act, "layernorm_bwd_q8: an activation derivative needs dx_lp");
    ILVLM_REQUIRE(dy && x && mean && rstd && gamma && dgamma && dbeta, "layernorm_bwd: null pointer");
    ILVLM_REQUIRE(dx_f32 || dx_lp || dx8, "layernorm_bwd: no output requested");
    ILVLM_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= MAXV * 256, "layernorm_bwd: bad cols %d", cols);
    ILVLM_REQUIRE(act == 0 || ((act == ILVLM_ACT_QUICKGELU_BWD || act == ILVLM_ACT_GELU_ERF_BWD) && act_aux && dx_lp),
                  "layernorm_bwd: bad activation arguments");
    hipStream_t s = (hipStream_t)stream;
    const bool defer = ws_blocks < 0;            // partial rows only: ilvlm_layernorm_bwd_reduce_batched adds them up later
    if (defer) ws_blocks = -ws_blocks;
    ILVLM_REQUIRE(ws == nullptr || ws_blocks > 0, "layernorm_bwd: ws_blocks must be non-zero");
    ILVLM_REQUIRE(!defer || ws, "layernorm_bwd: a deferred reduction needs the workspace");
    int blocks = ceil_div(rows, 4);
    const int cap = ws ? ws_blocks : 1024;
    if (blocks > cap) blocks = cap;
    dim3 grid(blocks), block(256);
    const int lp = dx_lp ? dx_lp_dtype : ILVLM_F32;
    const int nv = (cols + 255) / 256;   // register slots actually needed (1..4): fewer VGPRs -> more waves per SIMD
#define LN_BWD_NV(TDY, TX, TLP, NV)                                                                                   \
    hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TLP, NV>), grid, block, 0, s, (const TDY*)dy, (const TX*)x, mean, rstd,    \
                       gamma, dres, dx_f32, (TLP*)dx_lp, act, (const TLP*)act_aux, dgamma, dbeta, rows, cols, group, skip, ws, \
                       (unsigned char*)dx8, q_scale, q_amax)
#define LN_BWD(TDY, TX, TLP)                                                                                          \
    do {                                                                                                              \
        if (nv == 1) LN_BWD_NV(TDY, TX, TLP, 1);                                                                      \
        else if (nv == 2) LN_BWD_NV(TDY, TX, TLP, 2);                                                                 \
        else if (nv == 3) LN_BWD_NV(TDY, TX, TLP, 3);                                                                 \
        else LN_BWD_NV(TDY, TX, TLP, 4);                                                                              \
    } while (0)
    const int key = dy_dtype * 4 + x_dtype * 2 + lp;
    switch (key) {
        case 0: LN_BWD(float, float, float); break;
        case 1: LN_BWD(float, float, bf16); break;
        case 2: LN_BWD(float, bf16, float); break;
        case 3: LN_BWD(float, bf16, bf16); break;
        case 4: LN_BWD(bf16, float, float); break;
        case 5: LN_BWD(bf16, float, bf16); break;
        case 6: LN_BWD(bf16, bf16, float); break;
        case 7: LN_BWD(bf16, bf16, bf16); break;
        default: ILVLM_FAIL(ILVLM_ERR_ARG, "layernorm_bwd: bad dtypes");
    }
#undef LN_BWD
#undef LN_BWD_NV
    ILVLM_LAUNCH_CHECK("layernorm_bwd");
    if (ws && !defer) {
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(ceil_div(cols, 16), 2), dim3(256), 0, s, ws, blocks, cols, dgamma, dbeta);
        ILVLM_LAUNCH_CHECK("layernorm_bwd_reduce");
    }
    return ILVLM_OK;
}
