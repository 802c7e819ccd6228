// HBM-bound row / elementwise kernels of the CLIP+FDT step: embedding gather & scatter-add, patch gather,
// FDT token pooling, sparsemax / softmax over the codebook, L2 normalisation, temperature, InfoNCE, top-k
// accuracy, column sums, casts.  All fp32 arithmetic; coalesced / float4 accesses; wave = 64.
// Reference call sites are cited per kernel and in include/ilvlm_hip.h.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// token embedding (reference text_transformer.py:228-231); one wave per token row
// ------------------------------------------------------------------------------------------------
// Row layout of the token stream: dense = row b*L + l for every position; packed (seq_offs != nullptr) = only the
// first len_b = seq_offs[b+1] - seq_offs[b] positions of sequence b, at rows seq_offs[b] + l.  Positions past
// <|endoftext|> never reach the loss (causal attention, masked FDT scores, EOT pooling), so the training step runs
// the text tower on the packed rows.
__device__ __forceinline__ long token_row(const int* seq_offs, int b, int l, int L) {
    if (!seq_offs) return (long)b * L + l;
    return l < seq_offs[b + 1] - seq_offs[b] ? (long)seq_offs[b] + l : -1;
}

__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ table,
                                                        const float* __restrict__ pos, float* __restrict__ x, int B,
                                                        int L, int W, int vocab, const int* __restrict__ seq_offs) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);     // (b, l) pair
    if (i >= (long)B * L) return;
    const int b = i / L, l = i % L;
    const long row = token_row(seq_offs, b, l, L);
    if (row < 0) return;
    long t = tok[i];
    t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);   // clamp: never read out of the table
    const float* e = table + t * W;
    const float* p = pos + l * (long)W;
    float* o = x + row * W;
    for (int c = lane * 4; c < W; c += 256) *(f32x4*)(o + c) = *(const f32x4*)(e + c) + *(const f32x4*)(p + c);
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ dx,
                                                        float* __restrict__ dtable, int B, int L, int W, int vocab,
                                                        const int* __restrict__ seq_offs) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= (long)B * L) return;
    const long row = token_row(seq_offs, i / L, i % L, L);
    if (row < 0) return;
    long t = tok[i];
    if (t < 0 || t >= vocab) return;
    const float* g = dx + row * W;
    float* d = dtable + t * W;
    // 256 contiguous bytes per wave instruction.  Rows past <|endoftext|> carry an exactly-zero gradient (nothing
    // downstream reads them) and all share token id 0: skipping zero addends removes that 1000-way atomic pile-up
    // on table row 0 without changing the sum.
    for (int c = lane; c < W; c += 64) {
        const float v = g[c];
        if (v != 0.f) atomicAdd(d + c, v);
    }
}

// out[l, c] += sum_b x[b, l, c];  out0[c] += sum_b x[b, 0, c]
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                        float* __restrict__ out0, int B, int L, int W, int bchunk,
                                                        const int* __restrict__ seq_offs) {
    const int l = blockIdx.x, b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
    for (int c = threadIdx.x; c < W; c += 256) {
        float s = 0.f;
        for (int b = b0; b < b1; ++b) {
            const long row = token_row(seq_offs, b, l, L);
            if (row >= 0) s += x[row * W + c];
        }
        atomicAdd(out + (long)l * W + c, s);
        if (out0 && l == 0) atomicAdd(out0 + c, s);
    }
}

__global__ __launch_bounds__(256) void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                       float* __restrict__ tokens, int B, int L, int W) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * W) return;
    int b = i / W, c = i % W;
    tokens[(long)b * L * W + c] = cls[c] + pos[c];
}

template <class T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int C,
                                                       int res, int ps, int ld) {
    const int g = res / ps, K = C * ps * ps;
    const long total = (long)B * g * g * ld;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int k = i % ld;
        long p = i / ld;
        float v = 0.f;                      // columns K..ld-1 are zero padding (K-tile alignment of the patch GEMM)
        if (k < K) {
            int kx = k % ps, ky = (k / ps) % ps, c = k / (ps * ps);
            int px = p % g, py = (p / g) % g;
            long b = p / (g * g);
            v = img[((b * C + c) * res + py * ps + ky) * (long)res + px * ps + kx];
        }
        out[i] = from_f<T>(v);
    }
}

// four consecutive patch columns per thread (ps, K and ld multiples of 4): one 16-byte load, one 8- / 16-byte store, and the
// index arithmetic once per four elements -- the scalar form spends its time in integer divisions (160 us for 256 images,
// 1.4 TB/s)
template <class T>
__global__ __launch_bounds__(256) void patchify4_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int C,
                                                        int res, int ps, int ld) {
    const int g = res / ps, K = C * ps * ps, ld4 = ld >> 2, ps4 = ps >> 2;
    const long total = (long)B * g * g * ld4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k4 = (int)(i % ld4);
        const long p = i / ld4;
        const int k = k4 << 2;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < K) {
            const int kx = (k4 % ps4) << 2, ky = (k4 / ps4) % ps, c = k4 / (ps4 * ps);
            const int px = (int)(p % g), py = (int)((p / g) % g);
            const long b = p / (g * g);
            v = *(const f32x4*)(img + ((b * C + c) * res + py * ps + ky) * (long)res + px * ps + kx);
        }
        store4<T>(out + p * ld + k, v);
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                          float* __restrict__ y, int B, int L, int W,
                                                          const int* __restrict__ seq_offs) {
    const int b = blockIdx.x;
    const int len = seq_offs ? seq_offs[b + 1] - seq_offs[b] : L;
    long r = idx[b];
    r = r < 0 ? 0 : (r >= len ? len - 1 : r);
    const long row = (seq_offs ? (long)seq_offs[b] : (long)b * L) + r;
    for (int c = threadIdx.x; c < W; c += 256) y[(long)b * W + c] = x[row * W + c];
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ dy, const int64_t* __restrict__ idx,
                                                           float* __restrict__ dx, int B, int L, int W,
                                                           const int* __restrict__ seq_offs) {
    const int b = blockIdx.x;
    const int len = seq_offs ? seq_offs[b + 1] - seq_offs[b] : L;
    long r = idx[b];
    if (r < 0 || r >= len) return;
    const long row = (seq_offs ? (long)seq_offs[b] : (long)b * L) + r;
    for (int c = threadIdx.x; c < W; c += 256) dx[row * W + c] += dy[(long)b * W + c];
}

// ------------------------------------------------------------------------------------------------
// FDT pooling over tokens (reference clip_fdt.py:118-145): order of operations kept:
//   v = ((s / sqrt_d) * keep) / temperature
// ------------------------------------------------------------------------------------------------
// Packed rows (seq_offs): only the valid tokens have a score row; the masked positions behind them contribute the
// value the dense form gives them, (s * 0) / temperature = 0 -- to the maximum (argmax = first masked position, whose
// gradient is multiplied by the zero mask), to the sum, and to the mean's divisor T.
__global__ __launch_bounds__(256) void fdt_pool_fwd_kernel(const float* __restrict__ s, const float* __restrict__ mask,
                                                           float* __restrict__ pooled, int* __restrict__ argmax, int T, int C,
                                                           float sqrt_d, float temp, int pool, const int* __restrict__ seq_offs) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int len = seq_offs ? seq_offs[b + 1] - seq_offs[b] : T;
    const float* sb = s + (seq_offs ? (long)seq_offs[b] : (long)b * T) * C + c;
    float acc = pool == ILVLM_POOL_MAX ? -INFINITY : 0.f;
    int am = 0;
    for (int t = 0; t < len; ++t) {
        float keep = (seq_offs || mask == nullptr || mask[(long)b * T + t] == 0.f) ? 1.f : 0.f;
        float v = ((sb[(long)t * C] / sqrt_d) * keep) / temp;
        if (pool == ILVLM_POOL_MAX) {
            if (v > acc) { acc = v; am = t; }   // first maximum wins (torch.max semantics)
        } else acc += v;
    }
    if (len < T && pool == ILVLM_POOL_MAX && 0.f > acc) { acc = 0.f; am = len; }
    if (pool == ILVLM_POOL_MEAN) acc = acc / T;
    pooled[(long)b * C + c] = acc;
    if (argmax) argmax[(long)b * C + c] = am;
}

template <class T_>
__global__ __launch_bounds__(256) void fdt_pool_bwd_kernel(const float* __restrict__ dp, const int* __restrict__ argmax,
                                                           const float* __restrict__ mask, T_* __restrict__ ds, int T, int C,
                                                           float sqrt_d, float temp, int pool, const int* __restrict__ seq_offs) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float g = dp[(long)b * C + c] / temp / sqrt_d;
    if (pool == ILVLM_POOL_MEAN) g = g / T;
    const int am = pool == ILVLM_POOL_MAX ? argmax[(long)b * C + c] : -1;
    const int len = seq_offs ? seq_offs[b + 1] - seq_offs[b] : T;
    T_* db = ds + (seq_offs ? (long)seq_offs[b] : (long)b * T) * C + c;
    for (int t = 0; t < len; ++t) {
        float keep = (seq_offs || mask == nullptr || mask[(long)b * T + t] == 0.f) ? 1.f : 0.f;
        float v = (pool != ILVLM_POOL_MAX || t == am) ? g * keep : 0.f;
        db[(long)t * C] = from_f<T_>(v);
    }
}

// ------------------------------------------------------------------------------------------------
// sparsemax (reference sparsemax.py:22-71).  One 256-thread workgroup per row, the row (<= 8192 codes) in
// registers.  The reference sorts; here the threshold tau is found with Michelot's fixed-point iteration
// (tau <- (sum_{z>tau} z - 1) / |{z>tau}|, starting from the full set), which converges monotonically to the
// same support and tau in a handful of block reductions, no sort and no LDS traffic beyond the reductions.
// ------------------------------------------------------------------------------------------------
constexpr int SPV = 32;   // values per thread -> cols <= 8192

__global__ __launch_bounds__(256) void sparsemax_fwd_kernel(const float* __restrict__ z, float* __restrict__ out, int cols) {
    __shared__ float scratch[8];
    const float* zr = z + (long)blockIdx.x * cols;
    float v[SPV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SPV; ++i) {
        int c = i * 256 + threadIdx.x;
        v[i] = c < cols ? zr[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = block_max_256(mx, scratch);
#pragma unroll
    for (int i = 0; i < SPV; ++i) v[i] -= mx;   // translate by the max as the reference does
    float tau = -INFINITY;   // support = everything
    float kprev = -1.f;
    for (int it = 0; it < 64; ++it) {   // converges in a few iterations; bounded for safety
        float s = 0.f, k = 0.f;
#pragma unroll
        for (int i = 0; i < SPV; ++i)
            if (v[i] > tau) { s += v[i]; k += 1.f; }
        s = block_sum_256(s, scratch);
        k = block_sum_256(k, scratch + 4);
        tau = (s - 1.f) / k;
        if (k == kprev) break;   // uniform: support stopped shrinking
        kprev = k;
    }
    float* o = out + (long)blockIdx.x * cols;
#pragma unroll
    for (int i = 0; i < SPV; ++i) {
        int c = i * 256 + threadIdx.x;
        if (c < cols) o[c] = fmaxf(0.f, v[i] - tau);
    }
}

// dz = (g - mean_S g) on the support S = {out > 0}: what autograd through the reference's sort/cumsum yields
__global__ __launch_bounds__(256) void sparsemax_bwd_kernel(const float* __restrict__ out, const float* __restrict__ g,
                                                            float* __restrict__ dz, int cols) {
    __shared__ float scratch[8];
    const long base = (long)blockIdx.x * cols;
    float s = 0.f, k = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256)
        if (out[base + c] > 0.f) { s += g[base + c]; k += 1.f; }
    s = block_sum_256(s, scratch);
    k = block_sum_256(k, scratch + 4);
    const float mean = s / k;
    for (int c = threadIdx.x; c < cols; c += 256) dz[base + c] = out[base + c] > 0.f ? g[base + c] - mean : 0.f;
}

__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ z, float* __restrict__ out, int cols) {
    __shared__ float scratch[8];
    const long base = (long)blockIdx.x * cols;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, z[base + c]);
    mx = block_max_256(mx, scratch);
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += expf(z[base + c] - mx);
    s = block_sum_256(s, scratch + 4);
    for (int c = threadIdx.x; c < cols; c += 256) out[base + c] = expf(z[base + c] - mx) / s;
}
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ out, const float* __restrict__ g,
                                                          float* __restrict__ dz, int cols) {
    __shared__ float scratch[4];
    const long base = (long)blockIdx.x * cols;
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += out[base + c] * g[base + c];
    s = block_sum_256(s, scratch);
    for (int c = threadIdx.x; c < cols; c += 256) dz[base + c] = out[base + c] * (g[base + c] - s);
}

// att_func_type 'sigmoid' (reference clip_fdt.py:76, 149, 156-157): att_weight = sigmoid(z); the weighted codebook sum is
// divided by the row sum of the weights, i.e. att_ft = (att_weight / sum) @ sd.  w = the returned weights, wn = w / sum (the
// GEMM operand), rsum[row] = sum.
__global__ __launch_bounds__(256) void sigmoid_norm_fwd_kernel(const float* __restrict__ z, float* __restrict__ w,
                                                               float* __restrict__ wn, float* __restrict__ rsum, int cols) {
    __shared__ float scratch[4];
    const long base = (long)blockIdx.x * cols;
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float v = 1.0f / (1.0f + expf(-z[base + c]));
        w[base + c] = v;
        s += v;
    }
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) rsum[blockIdx.x] = s;
    for (int c = threadIdx.x; c < cols; c += 256) wn[base + c] = w[base + c] / s;      // own elements: no barrier needed
}
// g = d loss / d wn;  dw_i = (g_i - sum_j g_j wn_j) / S;  dz_i = dw_i w_i (1 - w_i)
__global__ __launch_bounds__(256) void sigmoid_norm_bwd_kernel(const float* __restrict__ w, const float* __restrict__ wn,
                                                               const float* __restrict__ rsum, const float* __restrict__ g,
                                                               float* __restrict__ dz, int cols) {
    __shared__ float scratch[4];
    const long base = (long)blockIdx.x * cols;
    float dot = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) dot += g[base + c] * wn[base + c];
    dot = block_sum_256(dot, scratch);
    const float inv = 1.0f / rsum[blockIdx.x];
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float v = w[base + c];
        dz[base + c] = (g[base + c] - dot) * inv * v * (1.0f - v);
    }
}

// ------------------------------------------------------------------------------------------------
// y = x / (||x|| + eps)  (reference clip_fdt.py:411-412, clip.py:133-134); one wave per row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ norm, int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += xr[c] * xr[c];
    const float n = sqrtf(wave_sum(s));
    if (lane == 0) norm[row] = n;
    const float d = n + eps;
    for (int c = lane; c < cols; c += 64) y[(long)row * cols + c] = xr[c] / d;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ norm,
                                                         const float* __restrict__ dy, float* __restrict__ dx, int rows,
                                                         int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * cols;
    const float* gr = dy + (long)row * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += xr[c] * gr[c];
    s = wave_sum(s);
    const float n = norm[row], d = n + eps;
    const float coef = n > 0.f ? s / (n * d * d) : 0.f;
    for (int c = lane; c < cols; c += 64) dx[(long)row * cols + c] = gr[c] / d - xr[c] * coef;
}

__global__ void logit_scale_fwd_kernel(const float* __restrict__ p, float* __restrict__ out, float mx) {
    out[0] = fminf(expf(p[0]), mx);
}
// dparam += (sum dli*li + sum dlt*lt) / scale_used * exp(param): the reference clamps exp().data, the gradient of
// exp() still multiplies by the UNCLAMPED exp(param) (autograd saved the pre-clamp output; clip_fdt.py:415-416)
__global__ __launch_bounds__(256) void logit_scale_bwd_kernel(const float* __restrict__ dli, const float* __restrict__ li,
                                                              const float* __restrict__ dlt, const float* __restrict__ lt,
                                                              long n, const float* __restrict__ p,
                                                              const float* __restrict__ scale_used, float* __restrict__ dp) {
    __shared__ float scratch[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += dli[i] * li[i] + dlt[i] * lt[i];
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) atomicAdd(dp, s / scale_used[0] * expf(p[0]));
}

// ------------------------------------------------------------------------------------------------
// InfoNCE (reference loss.py:37-47): one workgroup per row of each logit matrix (grid = 2B)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void infonce_fwd_kernel(const float* __restrict__ li, const float* __restrict__ lt,
                                                          int B, int Bg, int off, float* __restrict__ loss,
                                                          float* __restrict__ dli, float* __restrict__ dlt) {
    __shared__ float scratch[8];
    const int which = blockIdx.x / B, r = blockIdx.x % B;
    const float* row = (which ? lt : li) + (long)r * Bg;
    float* drow = (which ? dlt : dli) + (long)r * Bg;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < Bg; c += 256) mx = fmaxf(mx, row[c]);
    mx = block_max_256(mx, scratch);
    float s = 0.f;
    for (int c = threadIdx.x; c < Bg; c += 256) s += expf(row[c] - mx);
    s = block_sum_256(s, scratch + 4);
    const int label = off + r;
    const float w = 0.5f / B;
    for (int c = threadIdx.x; c < Bg; c += 256) drow[c] = (expf(row[c] - mx) / s - (c == label ? 1.f : 0.f)) * w;
    if (threadIdx.x == 0) atomicAdd(loss, (logf(s) + mx - row[label]) * w);
}

// top-1 / top-k hit counts: label is in the top-k iff fewer than k logits are strictly larger
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ logits, int B, int Bg, int off, int k,
                                                   float* __restrict__ out) {
    __shared__ float scratch[4];
    const int r = blockIdx.x;
    const float* row = logits + (long)r * Bg;
    const float ref = row[off + r];
    float cnt = 0.f;
    for (int c = threadIdx.x; c < Bg; c += 256) cnt += row[c] > ref ? 1.f : 0.f;
    cnt = block_sum_256(cnt, scratch);
    if (threadIdx.x == 0) {
        if (cnt < 1.f) atomicAdd(out, 100.f / B);
        if (cnt < (float)k) atomicAdd(out + 1, 100.f / B);
    }
}

// out[c] += sum_r x[r, c]
template <class T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long rows, int cols,
                                                     int ld, int rchunk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long r0 = (long)blockIdx.y * rchunk, r1 = min(rows, r0 + rchunk);
    float s = 0.f;
    for (long r = r0; r < r1; ++r) s += to_f<T>(x[r * ld + c]);
    atomicAdd(out + c, s);
}

template <class T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 1024;
    for (; i + 4 <= n; i += stride) store4<T>(dst + i, *(const f32x4*)(src + i));
    if (i < n && i + 4 > n)
        for (long j = i; j < n; ++j) dst[j] = from_f<T>(src[j]);
}

// dst (fp32) = src (bf16): bf16 gradient buckets back into the fp32 gradient arena after the all-reduce
__global__ __launch_bounds__(256) void widen_kernel(const bf16* __restrict__ src, float* __restrict__ dst, long n) {
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 1024;
    for (; i + 4 <= n; i += stride) *(f32x4*)(dst + i) = load4<bf16>(src + i);
    if (i < n && i + 4 > n)
        for (long j = i; j < n; ++j) dst[j] = (float)src[j];
}

__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, float* __restrict__ y, float a, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = a * x[i];
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += x[i];
}
__global__ __launch_bounds__(256) void scale_dev_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ a, long n) {
    const float s = a[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = s * x[i];
}
__global__ __launch_bounds__(256) void clamp_kernel(float* __restrict__ x, float lo, float hi, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = fminf(fmaxf(x[i], lo), hi);
}

// out[0] += sum x^2; x *= min(1, max_norm / (sqrt(ss[0]) + 1e-6)): clip_grad_norm_ over the flat gradient arena
// (prototype/utils/grad_clip.py:12-47).  BITWISE REPRODUCIBLE: under data parallelism every rank computes the clip coefficient
// of the (bit-identical) averaged arena on its own, so a sum whose order depends on workgroup arrival -- one float atomic per
// workgroup, the first form of this kernel -- lets the replicas drift apart by an ulp per step.  Two launches instead: a fixed
// grid writes one partial per workgroup (each thread adds its elements in index order, the workgroup reduces in a fixed tree),
// then ONE workgroup adds the partials in a fixed order and is the only writer of out[0].
constexpr int SUMSQ_PARTIALS = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ partials) {
    __shared__ float scratch[8];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i] * x[i];
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partials, int count, float* __restrict__ out) {
    __shared__ float scratch[8];
    float s = 0.f;
    for (int i = threadIdx.x; i < count; i += 256) s += partials[i];
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) out[0] += s;
}
__global__ __launch_bounds__(256) void clip_by_norm_kernel(float* __restrict__ x, long n, const float* __restrict__ ss, float max_norm) {
    const float coef = max_norm / (sqrtf(ss[0]) + 1e-6f);
    if (!(coef < 1.f)) return;               // the reference multiplies only when the norm exceeds the bound
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= coef;
}

inline int grid_1d(long n, int per_block, int cap = 4096) {
    long g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

#define S_ ((hipStream_t)stream)

static int embed_fwd_impl(const int64_t* tokens, const float* table, const float* pos, float* x, int B, int L, int W,
                          int vocab, const int* seq_offs, void* stream) {
    ILVLM_REQUIRE(tokens && table && pos && x && B > 0 && L > 0 && W > 0 && W % 4 == 0 && vocab > 0, "embed_fwd: bad args");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(ceil_div((long)B * L, 4)), dim3(256), 0, S_, tokens, table, pos, x, B, L, W, vocab, seq_offs);
    ILVLM_LAUNCH_CHECK("embed_fwd");
    return ILVLM_OK;
}
static int batch_sum_impl(const float* x, float* out, float* out0, int B, int L, int W, const int* seq_offs, void* stream) {
    ILVLM_REQUIRE(x && out && B > 0 && L > 0 && W > 0, "batch_sum: bad args");
    int bchunk = 32;
    hipLaunchKernelGGL(batch_sum_kernel, dim3(L, ceil_div(B, bchunk)), dim3(256), 0, S_, x, out, out0, B, L, W, bchunk, seq_offs);
    ILVLM_LAUNCH_CHECK("batch_sum");
    return ILVLM_OK;
}
static int embed_bwd_impl(const int64_t* tokens, const float* dx, float* dtable, float* dpos, int B, int L, int W, int vocab,
                          const int* seq_offs, void* stream) {
    ILVLM_REQUIRE(tokens && dx && dtable && B > 0 && L > 0 && W > 0 && vocab > 0, "embed_bwd: bad args");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(ceil_div((long)B * L, 4)), dim3(256), 0, S_, tokens, dx, dtable, B, L, W, vocab, seq_offs);
    ILVLM_LAUNCH_CHECK("embed_bwd");
    if (dpos) return batch_sum_impl(dx, dpos, nullptr, B, L, W, seq_offs, stream);
    return ILVLM_OK;
}
extern "C" int ilvlm_embed_fwd(const int64_t* tokens, const float* table, const float* pos, float* x, int B, int L, int W,
                               int vocab, void* stream) {
    return embed_fwd_impl(tokens, table, pos, x, B, L, W, vocab, nullptr, stream);
}
extern "C" int ilvlm_embed_bwd(const int64_t* tokens, const float* dx, float* dtable, float* dpos, int B, int L, int W,
                               int vocab, void* stream) {
    return embed_bwd_impl(tokens, dx, dtable, dpos, B, L, W, vocab, nullptr, stream);
}
extern "C" int ilvlm_embed_packed_fwd(const int64_t* tokens, const int32_t* seq_offs, const float* table, const float* pos,
                                      float* x, int B, int L, int W, int vocab, void* stream) {
    ILVLM_REQUIRE(seq_offs, "embed_packed_fwd: null seq_offs");
    return embed_fwd_impl(tokens, table, pos, x, B, L, W, vocab, seq_offs, stream);
}
extern "C" int ilvlm_embed_packed_bwd(const int64_t* tokens, const int32_t* seq_offs, const float* dx, float* dtable,
                                      float* dpos, int B, int L, int W, int vocab, void* stream) {
    ILVLM_REQUIRE(seq_offs, "embed_packed_bwd: null seq_offs");
    return embed_bwd_impl(tokens, dx, dtable, dpos, B, L, W, vocab, seq_offs, stream);
}
extern "C" int ilvlm_batch_sum(const float* x, float* out, float* out0, int B, int L, int W, void* stream) {
    return batch_sum_impl(x, out, out0, B, L, W, nullptr, stream);
}
extern "C" int ilvlm_cls_rows(const float* cls, const float* pos, float* tokens, int B, int L, int W, void* stream) {
    ILVLM_REQUIRE(cls && pos && tokens && B > 0 && L > 0 && W > 0, "cls_rows: bad args");
    hipLaunchKernelGGL(cls_rows_kernel, dim3(ceil_div((long)B * W, 256)), dim3(256), 0, S_, cls, pos, tokens, B, L, W);
    ILVLM_LAUNCH_CHECK("cls_rows");
    return ILVLM_OK;
}
extern "C" int ilvlm_patchify(const float* images, void* patches, int dtype, int B, int C, int res, int ps, int ld,
                              void* stream) {
    ILVLM_REQUIRE(images && patches && B > 0 && C > 0 && ps > 0 && res >= ps, "patchify: bad args");
    ILVLM_REQUIRE(ld >= C * ps * ps, "patchify: ld=%d smaller than the patch length %d", ld, C * ps * ps);
    int g = res / ps;
    long total = (long)B * g * g * ld;
    if (ps % 4 == 0 && ld % 4 == 0 && res % 4 == 0 && ((uintptr_t)images % 16) == 0 && ((uintptr_t)patches % 16) == 0) {
        int grid4 = grid_1d(total / 4, 256, 16384);
        if (dtype == ILVLM_BF16) hipLaunchKernelGGL(patchify4_kernel<bf16>, dim3(grid4), dim3(256), 0, S_, images, (bf16*)patches, B, C, res, ps, ld);
        else if (dtype == ILVLM_F32) hipLaunchKernelGGL(patchify4_kernel<float>, dim3(grid4), dim3(256), 0, S_, images, (float*)patches, B, C, res, ps, ld);
        else ILVLM_FAIL(ILVLM_ERR_ARG, "patchify: bad dtype %d", dtype);
        ILVLM_LAUNCH_CHECK("patchify");
        return ILVLM_OK;
    }
    int grid = grid_1d(total, 256, 8192);
    if (dtype == ILVLM_BF16) hipLaunchKernelGGL(patchify_kernel<bf16>, dim3(grid), dim3(256), 0, S_, images, (bf16*)patches, B, C, res, ps, ld);
    else if (dtype == ILVLM_F32) hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid), dim3(256), 0, S_, images, (float*)patches, B, C, res, ps, ld);
    else ILVLM_FAIL(ILVLM_ERR_ARG, "patchify: bad dtype %d", dtype);
    ILVLM_LAUNCH_CHECK("patchify");
    return ILVLM_OK;
}
extern "C" int ilvlm_gather_rows(const float* x, const int64_t* idx, float* y, int B, int L, int W, void* stream) {
    ILVLM_REQUIRE(x && idx && y && B > 0 && L > 0 && W > 0, "gather_rows: bad args");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, S_, x, idx, y, B, L, W, (const int*)nullptr);
    ILVLM_LAUNCH_CHECK("gather_rows");
    return ILVLM_OK;
}
extern "C" int ilvlm_gather_packed_rows(const float* x, const int64_t* idx, const int32_t* seq_offs, float* y, int B, int W,
                                        void* stream) {
    ILVLM_REQUIRE(x && idx && seq_offs && y && B > 0 && W > 0, "gather_packed_rows: bad args");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, S_, x, idx, y, B, 0, W, seq_offs);
    ILVLM_LAUNCH_CHECK("gather_packed_rows");
    return ILVLM_OK;
}
extern "C" int ilvlm_scatter_rows(const float* dy, const int64_t* idx, float* dx, int B, int L, int W, void* stream) {
    ILVLM_REQUIRE(dy && idx && dx && B > 0 && L > 0 && W > 0, "scatter_rows: bad args");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(B), dim3(256), 0, S_, dy, idx, dx, B, L, W, (const int*)nullptr);
    ILVLM_LAUNCH_CHECK("scatter_rows");
    return ILVLM_OK;
}
extern "C" int ilvlm_scatter_packed_rows(const float* dy, const int64_t* idx, const int32_t* seq_offs, float* dx, int B, int W,
                                         void* stream) {
    ILVLM_REQUIRE(dy && idx && seq_offs && dx && B > 0 && W > 0, "scatter_packed_rows: bad args");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(B), dim3(256), 0, S_, dy, idx, dx, B, 0, W, seq_offs);
    ILVLM_LAUNCH_CHECK("scatter_packed_rows");
    return ILVLM_OK;
}

// ---- input pipeline: uint8 image batch (as decoded / cropped by the loader's workers) -> normalised fp32 NCHW on the device.
// ToTensor (/255) + Normalize(mean, std) of prototype/data/imagenet_dataloader.py:13-14,59-68 and the two augmentations of
// MOCOV2_single whose effect is a pure function of the pixels once their coin is tossed: RandomHorizontalFlip (flag bit 0)
// and RandomGrayscale (flag bit 1; PIL's ITU-R 601-2 luma (19595 R + 38470 G + 7471 B + 32768) >> 16, replicated).
struct ImgNorm { float mean[3], std[3]; };
__global__ __launch_bounds__(256) void image_u8_kernel(const unsigned char* __restrict__ src, const unsigned char* __restrict__ flags,
                                                       float* __restrict__ dst, int H, int W, int nhwc, ImgNorm nm) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int f = flags ? flags[b] : 0;
    const int xs = (f & 1) ? W - 1 - x : x;
    const long plane = (long)H * W;
    int px[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
        px[ch] = nhwc ? src[(((long)b * H + y) * W + xs) * 3 + ch] : src[((long)b * 3 + ch) * plane + (long)y * W + xs];
    if (f & 2) {
        const int l = (px[0] * 19595 + px[1] * 38470 + px[2] * 7471 + 0x8000) >> 16;
        px[0] = px[1] = px[2] = l;
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
        dst[((long)b * 3 + ch) * plane + (long)y * W + x] =
            __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[ch], 255.0f), nm.mean[ch]), nm.std[ch]);     // float32 ToTensor / Normalize, division by division
}

extern "C" int ilvlm_image_u8_normalize(const unsigned char* src, int nhwc, const unsigned char* flags, float* dst, int B, int H,
                                        int W, const float* mean3, const float* std3, void* stream) {
    ILVLM_REQUIRE(src && dst && mean3 && std3 && B > 0 && H > 0 && W > 0, "image_u8_normalize: bad args");
    ILVLM_REQUIRE(B <= 65535 && H <= 65535, "image_u8_normalize: batch / height too large for one launch");
    ImgNorm nm;
    for (int i = 0; i < 3; ++i) {
        ILVLM_REQUIRE(std3[i] > 0.f, "image_u8_normalize: std must be positive");
        nm.mean[i] = mean3[i];
        nm.std[i] = std3[i];
    }
    hipLaunchKernelGGL(image_u8_kernel, dim3(ceil_div(W, 256), H, B), dim3(256), 0, S_, src, flags, dst, H, W, nhwc ? 1 : 0, nm);
    ILVLM_LAUNCH_CHECK("image_u8_normalize");
    return ILVLM_OK;
}

// ---- fused scores + max-pool: initial value and decode of the packed (key << 32 | 0x7fffffff - token) words
__global__ __launch_bounds__(256) void fdt_pack_init_kernel(unsigned long long* __restrict__ w, int C, int T,
                                                            const int* __restrict__ seq_offs) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    unsigned long long v = 0;                   // below the key of every float, -inf included
    if (seq_offs) {
        const int len = seq_offs[b + 1] - seq_offs[b];
        // masked positions of a short caption contribute exactly 0; token index = len marks "no token"
        if (len < T) v = ((unsigned long long)0x80000000u << 32) | (unsigned)(0x7fffffff - len);
    }
    w[(long)b * C + c] = v;
}
__global__ __launch_bounds__(256) void fdt_pack_decode_kernel(const unsigned long long* __restrict__ w, float* __restrict__ pooled,
                                                              int* __restrict__ argmax, long n, float sqrt_d, float temp) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long v = w[i];
    const unsigned key = (unsigned)(v >> 32);
    const unsigned u = (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key;
    pooled[i] = (__uint_as_float(u) / sqrt_d) / temp;      // the same two roundings as fdt_pool_fwd_kernel applies per token
    argmax[i] = 0x7fffffff - (int)(unsigned)(v & 0xffffffffu);
}

extern "C" int ilvlm_fdt_score_pool_fwd(const void* q, const void* sd, unsigned long long* packed_ws, float* pooled,
                                        int* argmax, long rows, int B, int T, int C, int d, float sqrt_d, float temperature,
                                        const int32_t* seq_offs, const int32_t* row_seq, void* stream) {
    ILVLM_REQUIRE(q && sd && packed_ws && pooled && argmax, "fdt_score_pool_fwd: null pointer");
    ILVLM_REQUIRE(rows > 0 && B > 0 && T > 0 && C > 0 && d > 0 && d % 64 == 0, "fdt_score_pool_fwd: bad shape (d %% 64 == 0)");
    ILVLM_REQUIRE(sqrt_d > 0.f && temperature > 0.f, "fdt_score_pool_fwd: needs sqrt_d > 0 and temperature > 0");
    ILVLM_REQUIRE((seq_offs != nullptr) == (row_seq != nullptr), "fdt_score_pool_fwd: seq_offs and row_seq go together");
    ILVLM_REQUIRE(seq_offs || rows == (long)B * T, "fdt_score_pool_fwd: dense layout needs rows == B * T");
    hipLaunchKernelGGL(fdt_pack_init_kernel, dim3(ceil_div(C, 256), B), dim3(256), 0, S_, packed_ws, C, T, seq_offs);
    ILVLM_LAUNCH_CHECK("fdt_score_pool_fwd(init)");
    ilvlm_gemm_epilogue ep = {};
    ep.alpha = 1.0f;
    ep.out_dtype = ILVLM_F32;
    ep.pool_out = packed_ws;
    ep.pool_seq = row_seq;
    ep.pool_offs = seq_offs;
    ep.pool_group = T;
    // C is never written by the pool epilogue; the scratch pointer only satisfies the argument checks
    int rc = ilvlm_gemm(ILVLM_BF16, 0, 0, (int)rows, C, d, q, d, sd, d, packed_ws, C, &ep, 1, stream);
    if (rc) return rc;
    const long n = (long)B * C;
    hipLaunchKernelGGL(fdt_pack_decode_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, S_, packed_ws, pooled, argmax, n, sqrt_d,
                       temperature);
    ILVLM_LAUNCH_CHECK("fdt_score_pool_fwd(decode)");
    return ILVLM_OK;
}

static int fdt_pool_fwd_impl(const float* scores, const float* pad_mask, float* pooled, int* argmax, int B, int T, int C,
                             float sqrt_d, float temperature, int pool, const int* seq_offs, void* stream) {
    ILVLM_REQUIRE(scores && pooled && B > 0 && T > 0 && C > 0, "fdt_pool_fwd: bad args");
    ILVLM_REQUIRE(pool >= 0 && pool <= 2 && (pool != ILVLM_POOL_MAX || argmax), "fdt_pool_fwd: bad pool / missing argmax");
    ILVLM_REQUIRE(sqrt_d > 0.f && temperature != 0.f, "fdt_pool_fwd: bad scale");
    hipLaunchKernelGGL(fdt_pool_fwd_kernel, dim3(ceil_div(C, 256), B), dim3(256), 0, S_, scores, pad_mask, pooled, argmax, T,
                       C, sqrt_d, temperature, pool, seq_offs);
    ILVLM_LAUNCH_CHECK("fdt_pool_fwd");
    return ILVLM_OK;
}
static int fdt_pool_bwd_impl(const float* dpooled, const int* argmax, const float* pad_mask, void* dscores, int dtype, int B,
                             int T, int C, float sqrt_d, float temperature, int pool, const int* seq_offs, void* stream) {
    ILVLM_REQUIRE(dpooled && dscores && B > 0 && T > 0 && C > 0, "fdt_pool_bwd: bad args");
    ILVLM_REQUIRE(pool >= 0 && pool <= 2 && (pool != ILVLM_POOL_MAX || argmax), "fdt_pool_bwd: bad pool / missing argmax");
    dim3 grid(ceil_div(C, 256), B);
    if (dtype == ILVLM_BF16) hipLaunchKernelGGL(fdt_pool_bwd_kernel<bf16>, grid, dim3(256), 0, S_, dpooled, argmax, pad_mask, (bf16*)dscores, T, C, sqrt_d, temperature, pool, seq_offs);
    else if (dtype == ILVLM_F32) hipLaunchKernelGGL(fdt_pool_bwd_kernel<float>, grid, dim3(256), 0, S_, dpooled, argmax, pad_mask, (float*)dscores, T, C, sqrt_d, temperature, pool, seq_offs);
    else ILVLM_FAIL(ILVLM_ERR_ARG, "fdt_pool_bwd: bad dtype %d", dtype);
    ILVLM_LAUNCH_CHECK("fdt_pool_bwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_fdt_pool_fwd(const float* scores, const float* pad_mask, float* pooled, int* argmax, int B, int T,
                                  int C, float sqrt_d, float temperature, int pool, void* stream) {
    return fdt_pool_fwd_impl(scores, pad_mask, pooled, argmax, B, T, C, sqrt_d, temperature, pool, nullptr, stream);
}
extern "C" int ilvlm_fdt_pool_bwd(const float* dpooled, const int* argmax, const float* pad_mask, void* dscores, int dtype,
                                  int B, int T, int C, float sqrt_d, float temperature, int pool, void* stream) {
    return fdt_pool_bwd_impl(dpooled, argmax, pad_mask, dscores, dtype, B, T, C, sqrt_d, temperature, pool, nullptr, stream);
}
extern "C" int ilvlm_fdt_pool_packed_fwd(const float* scores, const int32_t* seq_offs, float* pooled, int* argmax, int B, int T,
                                         int C, float sqrt_d, float temperature, int pool, void* stream) {
    ILVLM_REQUIRE(seq_offs, "fdt_pool_packed_fwd: null seq_offs");
    return fdt_pool_fwd_impl(scores, nullptr, pooled, argmax, B, T, C, sqrt_d, temperature, pool, seq_offs, stream);
}
extern "C" int ilvlm_fdt_pool_packed_bwd(const float* dpooled, const int* argmax, const int32_t* seq_offs, void* dscores,
                                         int dtype, int B, int T, int C, float sqrt_d, float temperature, int pool,
                                         void* stream) {
    ILVLM_REQUIRE(seq_offs, "fdt_pool_packed_bwd: null seq_offs");
    return fdt_pool_bwd_impl(dpooled, argmax, nullptr, dscores, dtype, B, T, C, sqrt_d, temperature, pool, seq_offs, stream);
}

extern "C" int ilvlm_sparsemax_fwd(const float* z, float* out, int rows, int cols, void* stream) {
    ILVLM_REQUIRE(z && out && rows > 0 && cols > 0 && cols <= SPV * 256, "sparsemax_fwd: cols=%d must be in (0, %d]", cols, SPV * 256);
    hipLaunchKernelGGL(sparsemax_fwd_kernel, dim3(rows), dim3(256), 0, S_, z, out, cols);
    ILVLM_LAUNCH_CHECK("sparsemax_fwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_sparsemax_bwd(const float* out, const float* g, float* dz, int rows, int cols, void* stream) {
    ILVLM_REQUIRE(out && g && dz && rows > 0 && cols > 0, "sparsemax_bwd: bad args");
    hipLaunchKernelGGL(sparsemax_bwd_kernel, dim3(rows), dim3(256), 0, S_, out, g, dz, cols);
    ILVLM_LAUNCH_CHECK("sparsemax_bwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_softmax_fwd(const float* z, float* out, int rows, int cols, void* stream) {
    ILVLM_REQUIRE(z && out && rows > 0 && cols > 0, "softmax_fwd: bad args");
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(rows), dim3(256), 0, S_, z, out, cols);
    ILVLM_LAUNCH_CHECK("softmax_fwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_softmax_bwd(const float* out, const float* g, float* dz, int rows, int cols, void* stream) {
    ILVLM_REQUIRE(out && g && dz && rows > 0 && cols > 0, "softmax_bwd: bad args");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(rows), dim3(256), 0, S_, out, g, dz, cols);
    ILVLM_LAUNCH_CHECK("softmax_bwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_sigmoid_norm_fwd(const float* z, float* w, float* wn, float* rowsum, int rows, int cols, void* stream) {
    ILVLM_REQUIRE(z && w && wn && rowsum && rows > 0 && cols > 0, "sigmoid_norm_fwd: bad args");
    hipLaunchKernelGGL(sigmoid_norm_fwd_kernel, dim3(rows), dim3(256), 0, S_, z, w, wn, rowsum, cols);
    ILVLM_LAUNCH_CHECK("sigmoid_norm_fwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_sigmoid_norm_bwd(const float* w, const float* wn, const float* rowsum, const float* g, float* dz, int rows,
                                      int cols, void* stream) {
    ILVLM_REQUIRE(w && wn && rowsum && g && dz && rows > 0 && cols > 0, "sigmoid_norm_bwd: bad args");
    hipLaunchKernelGGL(sigmoid_norm_bwd_kernel, dim3(rows), dim3(256), 0, S_, w, wn, rowsum, g, dz, cols);
    ILVLM_LAUNCH_CHECK("sigmoid_norm_bwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_l2norm_fwd(const float* x, float* y, float* norm, int rows, int cols, float eps, void* stream) {
    ILVLM_REQUIRE(x && y && norm && rows > 0 && cols > 0, "l2norm_fwd: bad args");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, S_, x, y, norm, rows, cols, eps);
    ILVLM_LAUNCH_CHECK("l2norm_fwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_l2norm_bwd(const float* x, const float* norm, const float* dy, float* dx, int rows, int cols, float eps,
                                void* stream) {
    ILVLM_REQUIRE(x && norm && dy && dx && rows > 0 && cols > 0, "l2norm_bwd: bad args");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, S_, x, norm, dy, dx, rows, cols, eps);
    ILVLM_LAUNCH_CHECK("l2norm_bwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_logit_scale_fwd(const float* logit_scale, float* out, float max_scale, void* stream) {
    ILVLM_REQUIRE(logit_scale && out, "logit_scale_fwd: null pointer");
    hipLaunchKernelGGL(logit_scale_fwd_kernel, dim3(1), dim3(1), 0, S_, logit_scale, out, max_scale);
    ILVLM_LAUNCH_CHECK("logit_scale_fwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_logit_scale_bwd(const float* dli, const float* li, const float* dlt, const float* lt, long n,
                                     const float* logit_scale, const float* scale_used, float* dparam, void* stream) {
    ILVLM_REQUIRE(dli && li && dlt && lt && logit_scale && scale_used && dparam && n > 0, "logit_scale_bwd: bad args");
    hipLaunchKernelGGL(logit_scale_bwd_kernel, dim3(grid_1d(n, 1024, 256)), dim3(256), 0, S_, dli, li, dlt, lt, n, logit_scale,
                       scale_used, dparam);
    ILVLM_LAUNCH_CHECK("logit_scale_bwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_infonce_fwd(const float* logits_i, const float* logits_t, int B, int Bg, int label_offset, float* loss,
                                 float* dlogits_i, float* dlogits_t, void* stream) {
    ILVLM_REQUIRE(logits_i && logits_t && loss && dlogits_i && dlogits_t, "infonce_fwd: null pointer");
    ILVLM_REQUIRE(B > 0 && Bg >= B && label_offset >= 0 && label_offset + B <= Bg, "infonce_fwd: bad shape B=%d Bg=%d off=%d", B, Bg, label_offset);
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), S_);
    if (e != hipSuccess) ILVLM_FAIL((int)e, "infonce_fwd: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(infonce_fwd_kernel, dim3(2 * B), dim3(256), 0, S_, logits_i, logits_t, B, Bg, label_offset, loss,
                       dlogits_i, dlogits_t);
    ILVLM_LAUNCH_CHECK("infonce_fwd");
    return ILVLM_OK;
}
extern "C" int ilvlm_topk_accuracy(const float* logits, int B, int Bg, int label_offset, int k, float* out, void* stream) {
    ILVLM_REQUIRE(logits && out && B > 0 && Bg >= B && label_offset >= 0 && label_offset + B <= Bg && k >= 1, "topk_accuracy: bad args");
    hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(float), S_);
    if (e != hipSuccess) ILVLM_FAIL((int)e, "topk_accuracy: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(256), 0, S_, logits, B, Bg, label_offset, k, out);
    ILVLM_LAUNCH_CHECK("topk_accuracy");
    return ILVLM_OK;
}
extern "C" int ilvlm_colsum(const void* x, int dtype, float* out, long rows, int cols, int ld, void* stream) {
    ILVLM_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols, "colsum: bad args");
    int rchunk = 256;
    dim3 grid(ceil_div(cols, 256), ceil_div(rows, rchunk));
    if (dtype == ILVLM_BF16) hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, S_, (const bf16*)x, out, rows, cols, ld, rchunk);
    else if (dtype == ILVLM_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, S_, (const float*)x, out, rows, cols, ld, rchunk);
    else ILVLM_FAIL(ILVLM_ERR_ARG, "colsum: bad dtype %d", dtype);
    ILVLM_LAUNCH_CHECK("colsum");
    return ILVLM_OK;
}
extern "C" int ilvlm_cast_f32(const float* src, void* dst, int dst_dtype, long n, void* stream) {
    ILVLM_REQUIRE(src && dst && n > 0, "cast_f32: bad args");
    ILVLM_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "cast_f32: 16-byte alignment required");
    int grid = grid_1d(n, 1024, 4096);
    if (dst_dtype == ILVLM_BF16) hipLaunchKernelGGL(cast_kernel<bf16>, dim3(grid), dim3(256), 0, S_, src, (bf16*)dst, n);
    else if (dst_dtype == ILVLM_F32) hipLaunchKernelGGL(cast_kernel<float>, dim3(grid), dim3(256), 0, S_, src, (float*)dst, n);
    else ILVLM_FAIL(ILVLM_ERR_ARG, "cast_f32: bad dtype %d", dst_dtype);
    ILVLM_LAUNCH_CHECK("cast_f32");
    return ILVLM_OK;
}
extern "C" int ilvlm_cast_to_f32(const void* src, int src_dtype, float* dst, long n, void* stream) {
    ILVLM_REQUIRE(src && dst && n > 0, "cast_to_f32: bad args");
    ILVLM_REQUIRE(src_dtype == ILVLM_BF16, "cast_to_f32: source must be bf16 (got %d)", src_dtype);
    ILVLM_REQUIRE(((uintptr_t)src % 8) == 0 && ((uintptr_t)dst % 16) == 0, "cast_to_f32: 8 / 16-byte alignment required");
    hipLaunchKernelGGL(widen_kernel, dim3(grid_1d(n, 1024, 4096)), dim3(256), 0, S_, (const bf16*)src, dst, n);
    ILVLM_LAUNCH_CHECK("cast_to_f32");
    return ILVLM_OK;
}
extern "C" int ilvlm_scale(const float* x, float* y, float a, long n, void* stream) {
    ILVLM_REQUIRE(x && y && n > 0, "scale: bad args");
    hipLaunchKernelGGL(scale_kernel, dim3(grid_1d(n, 256, 4096)), dim3(256), 0, S_, x, y, a, n);
    ILVLM_LAUNCH_CHECK("scale");
    return ILVLM_OK;
}

extern "C" int ilvlm_add_inplace(float* y, const float* x, long n, void* stream) {
    ILVLM_REQUIRE(x && y && n > 0, "add_inplace: bad args");
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_1d(n, 256, 4096)), dim3(256), 0, S_, y, x, n);
    ILVLM_LAUNCH_CHECK("add_inplace");
    return ILVLM_OK;
}
extern "C" int ilvlm_scale_dev(const float* x, float* y, const float* a, long n, void* stream) {
    ILVLM_REQUIRE(x && y && a && n > 0, "scale_dev: bad args");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(grid_1d(n, 256, 4096)), dim3(256), 0, S_, x, y, a, n);
    ILVLM_LAUNCH_CHECK("scale_dev");
    return ILVLM_OK;
}
extern "C" int ilvlm_sumsq_partials(void) { return SUMSQ_PARTIALS; }
extern "C" int ilvlm_sumsq(const float* x, long n, float* out, float* partials, void* stream) {
    ILVLM_REQUIRE(x && out && partials && n > 0, "sumsq: bad args (partials: workspace of ilvlm_sumsq_partials() floats)");
    const int blocks = grid_1d(n, 1024, SUMSQ_PARTIALS);          // depends on n alone: the summation order is a function of n
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, S_, x, n, partials);
    ILVLM_LAUNCH_CHECK("sumsq");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, S_, partials, blocks, out);
    ILVLM_LAUNCH_CHECK("sumsq");
    return ILVLM_OK;
}
extern "C" int ilvlm_clip_by_norm(float* x, long n, const float* sumsq, float max_norm, void* stream) {
    ILVLM_REQUIRE(x && sumsq && n > 0 && max_norm > 0.f, "clip_by_norm: bad args");
    hipLaunchKernelGGL(clip_by_norm_kernel, dim3(grid_1d(n, 1024, 4096)), dim3(256), 0, S_, x, n, sumsq, max_norm);
    ILVLM_LAUNCH_CHECK("clip_by_norm");
    return ILVLM_OK;
}
extern "C" int ilvlm_clamp(float* x, float lo, float hi, long n, void* stream) {
    ILVLM_REQUIRE(x && n > 0 && lo <= hi, "clamp: bad args");
    hipLaunchKernelGGL(clamp_kernel, dim3(grid_1d(n, 256, 4096)), dim3(256), 0, S_, x, lo, hi, n);
    ILVLM_LAUNCH_CHECK("clamp");
    return ILVLM_OK;
}
