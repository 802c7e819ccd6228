// Fused multi-tensor AdamW over a flat fp32 parameter arena (decoupled weight decay, amsgrad off): one launch
// for all ~320 tensors, 28 B/param of HBM traffic (+2 B for the bf16 shadow the GEMMs consume).
// torch.optim.AdamW semantics as selected by the reference (prototype/optimizer/__init__.py:3,18-26;
// hyper-parameters example/clip_fdt/config_cc3m.yaml:34-41):
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
#include "common.h"

namespace {

constexpr int CHUNK = 4096;   // elements per workgroup-chunk (16 per thread)

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ shadow,
                                                    const int64_t* __restrict__ coff, const int32_t* __restrict__ ccnt,
                                                    const int32_t* __restrict__ cgrp, int n_chunks, ilvlm_adamw_hyper h,
                                                    float bc1, float bc2_sqrt) {
    for (int ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const int grp = cgrp[ch];
        if (!h.active[grp]) continue;
        const long off = coff[ch];
        const int n = ccnt[ch];
        const float lr = h.lr[grp], wd = h.weight_decay[grp];
        const float decay = 1.f - lr * wd, step_size = lr / bc1;
        // chunk offsets are multiples of 64 elements: 16-byte accesses, four elements per lane, then the tail
        const int n4 = n & ~3;
        for (int i = threadIdx.x * 4; i < n4; i += 1024) {
            const long k = off + i;
            const f32x4 gk = *(const f32x4*)(g + k);
            f32x4 pk = *(const f32x4*)(p + k), mk = *(const f32x4*)(m + k), vk = *(const f32x4*)(v + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pk[j] *= decay;
                mk[j] = h.beta1 * mk[j] + (1.f - h.beta1) * gk[j];
                vk[j] = h.beta2 * vk[j] + (1.f - h.beta2) * gk[j] * gk[j];
                pk[j] -= step_size * mk[j] / (sqrtf(vk[j]) / bc2_sqrt + h.eps);
            }
            *(f32x4*)(m + k) = mk;
            *(f32x4*)(v + k) = vk;
            *(f32x4*)(p + k) = pk;
            if (shadow) store4<bf16>(shadow + k, pk);
        }
        for (int i = n4 + threadIdx.x; i < n; i += 256) {
            const long k = off + i;
            float gk = g[k], pk = p[k] * decay;
            float mk = h.beta1 * m[k] + (1.f - h.beta1) * gk;
            float vk = h.beta2 * v[k] + (1.f - h.beta2) * gk * gk;
            m[k] = mk;
            v[k] = vk;
            pk -= step_size * mk / (sqrtf(vk) / bc2_sqrt + h.eps);
            p[k] = pk;
            if (shadow) shadow[k] = (bf16)pk;
        }
    }
}

// AdamW over the 64 x 64 tiles of the GEMM weights that the streaming kernel reads in MFMA-fragment order
// (ilvlm_pack_weights): the update of a tile, then -- from the bf16 values still in LDS -- the tile's part of the forward
// image (B operand of x W^T) and of the input-gradient image (of dY W), plus the row-major bf16 shadow.  The separate
// re-pack launch (0.15 ms per step: it re-read the shadow of 123 M parameters) and its place on the critical path between
// the optimizer and the next forward are gone.  Same arithmetic per element as adamw_kernel.
// table: 6 ints per tile {arena offset / 64, rows, cols, r0, c0, group}.
__global__ __launch_bounds__(256) void adamw_pack_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, bf16* __restrict__ shadow, bf16* __restrict__ fwd,
                                                         bf16* __restrict__ bwd, const int32_t* __restrict__ table, ilvlm_adamw_hyper h,
                                                         float bc1, float bc2_sqrt) {
    __shared__ __attribute__((aligned(16))) bf16 tile[64][64 + 8];
    const int* e = table + (long)blockIdx.x * 6;
    const int grp = e[5];
    if (!h.active[grp]) return;                      // frozen: parameters, shadow and images stay as they are
    const long off = (long)e[0] * 64;
    const int rows = e[1], cols = e[2], r0 = e[3], c0 = e[4];
    const float lr = h.lr[grp], wd = h.weight_decay[grp];
    const float decay = 1.f - lr * wd, step_size = lr / bc1;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (t >> 4) + 16 * i, c = (t & 15) * 4;
        const long k = off + (long)(r0 + r) * cols + c0 + c;
        const f32x4 gk = *(const f32x4*)(g + k);
        f32x4 pk = *(const f32x4*)(p + k), mk = *(const f32x4*)(m + k), vk = *(const f32x4*)(v + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pk[j] *= decay;
            mk[j] = h.beta1 * mk[j] + (1.f - h.beta1) * gk[j];
            vk[j] = h.beta2 * vk[j] + (1.f - h.beta2) * gk[j] * gk[j];
            pk[j] -= step_size * mk[j] / (sqrtf(vk[j]) / bc2_sqrt + h.eps);
        }
        *(f32x4*)(m + k) = mk;
        *(f32x4*)(v + k) = vk;
        *(f32x4*)(p + k) = pk;
        const bf16x4 lp = {(bf16)pk[0], (bf16)pk[1], (bf16)pk[2], (bf16)pk[3]};
        if (shadow) *(bf16x4*)(shadow + k) = lp;
        *(bf16x4*)&tile[r][c] = lp;
    }
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {                 // as pack_weights_kernel (gemm.hip): 16-byte chunks of both images
        const int c = t + 256 * hh, blk = c >> 6, l = c & 63, hi = blk >> 1, lo = blk & 1;
        const int n = 16 * hi + (l & 15), k = 32 * lo + 8 * (l >> 4);
        {
            const long b = (long)((r0 >> 4) + hi) * (cols >> 5) + (c0 >> 5) + lo;
            *(bf16x8*)(fwd + off + b * 512 + l * 8) = *(const bf16x8*)&tile[n][k];
        }
        {
            bf16x8 x;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = tile[k + j][n];
            const long b = (long)((c0 >> 4) + hi) * (rows >> 5) + (r0 >> 5) + lo;
            *(bf16x8*)(bwd + off + b * 512 + l * 8) = x;
        }
    }
}

}  // namespace

extern "C" int ilvlm_adamw_step_packed(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                                       void* packed_fwd, void* packed_bwd, const int32_t* tile_table, int n_tiles,
                                       const ilvlm_adamw_hyper* hyper, void* stream) {
    ILVLM_REQUIRE(params && grads && exp_avg && exp_avg_sq && packed_fwd && packed_bwd && tile_table && hyper,
                  "adamw_step_packed: null pointer");
    ILVLM_REQUIRE(n_tiles > 0 && hyper->step >= 1, "adamw_step_packed: bad n_tiles / step");
    const double bc1 = 1.0 - pow((double)hyper->beta1, (double)hyper->step);
    const double bc2 = 1.0 - pow((double)hyper->beta2, (double)hyper->step);
    hipLaunchKernelGGL(adamw_pack_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       (bf16*)shadow_bf16, (bf16*)packed_fwd, (bf16*)packed_bwd, tile_table, *hyper, (float)bc1, (float)sqrt(bc2));
    ILVLM_LAUNCH_CHECK("adamw_step_packed");
    return ILVLM_OK;
}

extern "C" int ilvlm_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                                const int64_t* chunk_offset, const int32_t* chunk_count, const int32_t* chunk_group,
                                int n_chunks, const ilvlm_adamw_hyper* hyper, void* stream) {
    ILVLM_REQUIRE(params && grads && exp_avg && exp_avg_sq && chunk_offset && chunk_count && chunk_group && hyper,
                  "adamw_step: null pointer");
    ILVLM_REQUIRE(n_chunks > 0 && hyper->step >= 1, "adamw_step: bad n_chunks / step");
    const double bc1 = 1.0 - pow((double)hyper->beta1, (double)hyper->step);
    const double bc2 = 1.0 - pow((double)hyper->beta2, (double)hyper->step);
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       (bf16*)shadow_bf16, chunk_offset, chunk_count, chunk_group, n_chunks, *hyper, (float)bc1,
                       (float)sqrt(bc2));
    ILVLM_LAUNCH_CHECK("adamw_step");
    return ILVLM_OK;
}
