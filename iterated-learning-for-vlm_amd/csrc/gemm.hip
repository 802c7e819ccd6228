// GEMM with fused epilogues for gfx950.
//   bf16 path : 128x128x64 tile, 4 waves (2x2) each 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles; operands are DMA'd
//               into swizzled LDS images (buffer_load_dwordx4 .. lds); K-contiguous operands are read with
//               ds_read_b128, K-strided ("transposed") ones with ds_read_b64_tr_b16, so dgrad / wgrad need no
//               transposed copies of weights or activations.  XCD-aware block remap.  A register-staged,
//               predicated kernel of the same tile takes the shapes the DMA path cannot (K % 64 != 0).
//   fp32 path : 64x64x16 tile on v_mfma_f32_16x16x4_f32 (bit-exact fp32 FMA chain) for the fp32
//               parity mode and the small precision-critical products (att@sd, logits).
// Replaces F.linear / matmul call sites listed in include/ilvlm_hip.h.
#include <atomic>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace {

struct EpiArgs {
    ilvlm_gemm_epilogue e;
    float* Cf;     // fp32 view of C
    bf16* Cb;      // bf16 view of C
    int ldc;
    int M, N;
    int vec_ok;    // N%4==0 && ldc%4==0 and all pointers 16B aligned
    int vec8_ok;   // additionally ldc%8==0 and C / aux 16-byte, the fp8 copy 8-byte aligned: the 8-column (16-byte bf16 store) epilogue
    int tile_group;   // store-type launches walk tiles in column groups of this width inside row bands (L2 blocking); 0 = off
    int tile_bands;   // number of row bands (8 = about one per XCD)
    int plain_acc;    // accumulate launches: this launch is the only writer of C and split_k == 1 -> load-add-store, no atomics
};

// Handles 4 consecutive columns [n, n+4) of output row m.  AuxT = compute dtype.
// fp8 copy of 4 stored values (or fewer at a ragged edge) + running max|value| for the amax
__device__ __forceinline__ void out8_store(const ilvlm_gemm_epilogue& e, long off, f32x4 v, int nvalid, float& amax) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nvalid) amax = fmaxf(amax, fabsf(v[i]));
    if (!e.out8) return;
    const float s = e.out8_scale ? e.out8_scale[0] : 1.f;
    const unsigned w = fp8_pack4_fmt(e.out8_fmt, v[0] * s, v[1] * s, v[2] * s, v[3] * s);
    unsigned char* p = (unsigned char*)e.out8 + off;
    if (nvalid == 4 && (off & 3) == 0) *(unsigned*)p = w;
    else
        for (int i = 0; i < nvalid; ++i) p[i] = (unsigned char)(w >> (8 * i));
}

__device__ __forceinline__ f32x4 fma4(f32x4 v, float alpha, f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(v[i], alpha, b[i]);
    return r;
}

template <class AuxT>
__device__ __forceinline__ void epilogue4(const EpiArgs& a, int m, int n, f32x4 v, float alpha, float* amax8 = nullptr) {
    if (m >= a.M || n >= a.N) return;
    const ilvlm_gemm_epilogue& e = a.e;
    long orow = map_row(m, e.out_group, e.out_skip);
    long off = orow * (long)a.ldc + n;
    const bool full = a.vec_ok && (n + 4 <= a.N);
    if (e.accumulate) {
        v *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n + i < a.N) atomicAdd(a.Cf + off + i, v[i]);
        return;
    }
    // alpha * acc + bias with ONE rounding (fma4), as the whole-tile epilogues compute it: a result must not depend on whether
    // its tile was whole, i.e. on the tile shape of the kernel that produced it
    f32x4 b = {0, 0, 0, 0};
    if (e.bias) {
        if (full) b = *(const f32x4*)(e.bias + n);
        else
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n + i < a.N) b[i] = e.bias[n + i];
    }
    v = fma4(v, alpha, b);
    if (full) {
        if (e.rowbias) v += *(const f32x4*)(e.rowbias + (long)(e.out_skip + m % e.out_group) * a.N + n);
        if (e.act) {
            AuxT* aux = (AuxT*)e.aux + off;
            if (e.act == ILVLM_ACT_QUICKGELU || e.act == ILVLM_ACT_GELU_ERF) {
                store4<AuxT>(aux, v);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = e.act == ILVLM_ACT_QUICKGELU ? quick_gelu(v[i]) : gelu_erf(v[i]);
            } else {
                f32x4 u = load4<AuxT>(aux);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] *= e.act == ILVLM_ACT_QUICKGELU_BWD ? quick_gelu_grad(u[i]) : gelu_erf_grad(u[i]);
            }
        }
        if (e.residual) v += *(const f32x4*)(e.residual + off);
        if (!a.Cf) {}                      // only the fp8 copy (and aux) of the result is kept
        else if (e.out_dtype == ILVLM_F32) store4<float>(a.Cf + off, v);
        else store4<bf16>(a.Cb + off, v);
        if (amax8) out8_store(e, off, v, 4, *amax8);
        return;
    }
    f32x4 fin = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (n + i >= a.N) break;
        float x = v[i];
        if (e.rowbias) x += e.rowbias[(long)(e.out_skip + m % e.out_group) * a.N + n + i];
        if (e.act) {
            AuxT* aux = (AuxT*)e.aux + off + i;
            if (e.act == ILVLM_ACT_QUICKGELU) { *aux = from_f<AuxT>(x); x = quick_gelu(x); }
            else if (e.act == ILVLM_ACT_GELU_ERF) { *aux = from_f<AuxT>(x); x = gelu_erf(x); }
            else if (e.act == ILVLM_ACT_QUICKGELU_BWD) x *= quick_gelu_grad(to_f<AuxT>(*aux));
            else x *= gelu_erf_grad(to_f<AuxT>(*aux));
        }
        if (e.residual) x += e.residual[off + i];
        if (!a.Cf) {}
        else if (e.out_dtype == ILVLM_F32) a.Cf[off + i] = x;
        else a.Cb[off + i] = (bf16)x;
        fin[i] = x;
    }
    if (amax8) out8_store(e, off, fin, min(4, a.N - n), *amax8);
}

// XCD-aware bijective remap (blocks b and b+8 share an XCD; give each XCD a contiguous tile range)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// =====================================================================================
// bf16 kernel
// =====================================================================================
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDK = BK + 8;    // K-contiguous image: [128 rows][72]   (144-byte rows)
constexpr int LDT = 128 + 8;   // K-strided   image: [64 k][136]       (272-byte rows)
constexpr int TILE_ELEMS = 128 * LDK;   // 9216 >= 64*136 = 8704
constexpr int GEMM_LDS_BYTES = 4 * TILE_ELEMS * 2;   // A,B x 2 buffers = 73728

struct Stage {   // one operand tile's worth of global loads held in registers: 4 x 16 bytes per thread
    bf16x8 v[4];
};

// Load an operand tile (rows [r0, r0+128) x k [k0, k0+64)) into registers.
// TR = false: source is [R, K] with K contiguous.  TR = true: source is [K, R] with R contiguous.
template <bool TR>
__device__ __forceinline__ void stage_load(Stage& s, const bf16* __restrict__ src, int ld, int r0, int k0, int R, int Kend,
                                           int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + 256 * i;
        int row, col, rlim, clim;
        long goff;
        if (!TR) {
            row = r0 + (c >> 3);           // operand row
            col = k0 + (c & 7) * 8;        // k
            rlim = R; clim = Kend;
            goff = (long)row * ld + col;
        } else {
            row = k0 + (c >> 4);           // k
            col = r0 + (c & 15) * 8;       // operand row (contiguous)
            rlim = Kend; clim = R;
            goff = (long)row * ld + col;
        }
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < rlim) {
            if (col + 8 <= clim) {
                v = *(const bf16x8*)(src + goff);
            } else {
                for (int j = 0; j < 8; ++j)
                    if (col + j < clim) v[j] = src[goff + j];
            }
        }
        s.v[i] = v;
    }
}

template <bool TR>
__device__ __forceinline__ void stage_store(const Stage& s, bf16* lds, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + 256 * i;
        int off = !TR ? (c >> 3) * LDK + (c & 7) * 8 : (c >> 4) * LDT + (c & 15) * 8;
        *(bf16x8*)(lds + off) = s.v[i];
    }
}

// fragment of 16 operand rows starting at row r16 for the 32-deep k-substep starting at k32.
// lane l holds operand[r16 + (l&15)][k32 + 8*(l>>4) + j], j = 0..7.
template <bool TR>
__device__ __forceinline__ bf16x8 load_frag(const bf16* lds, int r16, int k32, int lane) {
    if (!TR) {
        return *(const bf16x8*)(lds + (r16 + (lane & 15)) * LDK + k32 + 8 * (lane >> 4));
    } else {
        // ds_read_b64_tr_b16: within a 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
        // 4x16 block; lane i receives column i of the 4 rows.  Two blocks (k 0-3 and 4-7 of this lane group's 8 k).
        int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g = lane >> 4;
        const bf16* base = lds + (k32 + 8 * g + q) * LDT + r16 + 4 * p;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * LDT));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
        return u.v;
    }
}

// SWAP = true : acc tile holds C^T fragments -> each lane owns 4 consecutive columns n (vector epilogue)
// SWAP = false: each lane owns 4 consecutive rows m, lanes 0..15 span 16 contiguous n (atomic accumulate shape)
template <bool TA, bool TB, bool SWAP>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ B,
                                                        int ldb, int K, int tiles_m, int tiles_n, int split_k,
                                                        EpiArgs ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* smem = (bf16*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = tiles_m * tiles_n * split_k;
    int wg = xcd_remap(blockIdx.x, nwg);
    const int z = wg % split_k; wg /= split_k;
    const int tn = wg % tiles_n, tm = wg / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int nt_total = (K + BK - 1) / BK;
    const int per = (nt_total + split_k - 1) / split_k;
    const int t_begin = z * per;
    const int t_end = min(nt_total, t_begin + per);
    if (t_begin >= t_end) return;   // uniform per block
    const int Kend = min(K, t_end * BK);

    bf16* As[2] = {smem, smem + 2 * TILE_ELEMS};
    bf16* Bs[2] = {smem + TILE_ELEMS, smem + 3 * TILE_ELEMS};

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};

    Stage sa, sb;
    stage_load<TA>(sa, A, lda, m0, t_begin * BK, ep.M, Kend, tid);
    stage_load<TB>(sb, B, ldb, n0, t_begin * BK, ep.N, Kend, tid);
    stage_store<TA>(sa, As[0], tid);
    stage_store<TB>(sb, Bs[0], tid);
    __syncthreads();

    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        const bool more = (t + 1 < t_end);
        if (more) {
            stage_load<TA>(sa, A, lda, m0, (t + 1) * BK, ep.M, Kend, tid);
            stage_load<TB>(sb, B, ldb, n0, (t + 1) * BK, ep.N, Kend, tid);
        }
        const bf16* as = As[cur];
        const bf16* bs = Bs[cur];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = load_frag<TA>(as, wm * 64 + i * 16, ks * 32, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = load_frag<TB>(bs, wn * 64 + j * 16, ks * 32, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) {
            stage_store<TA>(sa, As[cur ^ 1], tid);
            stage_store<TB>(sb, Bs[cur ^ 1], tid);
        }
        __syncthreads();
    }

    float alpha = ep.e.alpha;
    if (ep.e.alpha_ptr) alpha *= *ep.e.alpha_ptr;
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (SWAP) {
                // D[n_local = 4g + r][m_local = c]
                int m = m0 + wm * 64 + i * 16 + c;
                int n = n0 + wn * 64 + j * 16 + 4 * g;
                epilogue4<bf16>(ep, m, n, acc[i][j], alpha);
            } else {
                // D[m_local = 4g + r][n_local = c] : only used with accumulate (scalar atomics)
                int n = n0 + wn * 64 + j * 16 + c;
                if (n < ep.N) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int m = m0 + wm * 64 + i * 16 + 4 * g + r;
                        if (m < ep.M) {
                            long off = map_row(m, ep.e.out_group, ep.e.out_skip) * (long)ep.ldc + n;
                            atomicAdd(ep.Cf + off, acc[i][j][r] * alpha);
                        }
                    }
                }
            }
        }
}


// =====================================================================================
// LDS images of the direct-to-LDS kernels.  A DMA wave instruction writes 64 lanes x 16 B = 1 KiB linearly, so the
// bank-conflict swizzle is applied to the per-lane SOURCE address and undone by the same XOR on the fragment read:
//   K-contiguous operand : [rows][64 k] 128-byte rows, 16-byte chunk c of row r stored at slot c ^ (r & 7)
//                          ([rows][32 k] 64-byte rows: slot c ^ ((r >> 2) & 3))
//   K-strided operand    : [k][rows] rows*2-byte rows, chunk c of k-row r stored at slot (c & ~15) | ((c & 15) ^ swz16(r)),
//                          swz16(r) = ((r & 3) << 2) | ((r >> 2) & 3)  (conflict-free for ds_read_b64_tr_b16)
// Requirements (checked on the host): K % 64 == 0 unless both operands are K-strided (their k-rows past K are beyond the
// buffer extent and read as zeros); a K-strided operand needs rows % 8 == 0.
// =====================================================================================
__device__ __forceinline__ int swz16(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

template <bool TR, int ROWS, int BKT = 64>
__device__ __forceinline__ bf16x8 p8_frag(const unsigned char* tile, int r16, int k32, int lane) {
    if (!TR) {
        const int row = r16 + (lane & 15), chunk = (k32 >> 3) + (lane >> 4);
        if (BKT == 64) return *(const bf16x8*)(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
        return *(const bf16x8*)(tile + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
    } else {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        constexpr int RB = ROWS * 2;                      // bytes per k-row
        const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g = lane >> 4;
        const int kr = k32 + 8 * g + q, c = (r16 >> 3) + (p >> 1), half = 8 * (p & 1);
        const int c_lo = (c & ~15) | ((c & 15) ^ swz16(kr)), c_hi = (c & ~15) | ((c & 15) ^ swz16(kr + 4));
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + RB * kr + 16 * c_lo + half));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + RB * (kr + 4) + 16 * c_hi + half));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
        return u.v;
    }
}

// Workgroup barrier of the pipelined kernels: raw s_barrier preceded by s_waitcnt lgkmcnt(0), both fenced for the compiler.
// The raw intrinsic is neither a memory barrier for the compiler (IntrNoMem) nor does it wait for the wave's LDS traffic,
// and MFMAs are free to sink below it together with the s_waitcnt of the fragments they consume.  A wave could thus cross
// the end-of-K-tile barrier with fragment ds_reads still queued, a faster wave issued the next tile's DMA, and when the
// LDS pipeline was busy (other workgroups of the CU transposing their epilogue through LDS) the DMA data overtook those
// reads: about one launch in ten came out with a 32-row x 128-column slice short of one K-tile -- found by a full-size
// packed-vs-dense comparison, invisible at test sizes.  Draining the LDS counter before the barrier closes the window
// (`benchmarks/gemm_determinism.py`: 0 of 400 launches differ); it costs nothing measurable because the fragments are
// needed by the MFMAs in front of the barrier anyway.
#define ILVLM_WG_BARRIER()                                      \
    do {                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
        __builtin_amdgcn_s_barrier();                           \
        asm volatile("" ::: "memory");                          \
    } while (0)

// ILVLM_GEMM_ABLATE (diagnostic builds only, `make ablate`): 1 = no fragment reads / MFMA, 2 = no operand DMA,
// 3 = no epilogue, 4 = no MFMA (fragment reads kept), 5 = no fragment reads (MFMA kept; phased kernel only) -- what each
// phase of the direct-to-LDS kernels costs with the others left in place
#ifndef ILVLM_GEMM_ABLATE
#define ILVLM_GEMM_ABLATE 0
#endif
#ifdef ILVLM_GEMM_STAMPS
// diagnostic build only: per-wave cycle sums of the main-loop phases (never compiled into the product library)
__device__ unsigned long long g_stamps[4096 * 8 * 6];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define STAMP(v) unsigned long long v = stamp()
#define STAMP_ADD(acc, a, b) acc += (b) - (a)
#else
#define STAMP(v)
#define STAMP_ADD(acc, a, b)
#endif

// =====================================================================================
// bf16 "dma" path: as the direct-to-LDS kernels above, plus
//   * operands addressed through buffer descriptors: per-lane byte offsets are computed ONCE, the tile / K-tile offset is
//     a scalar (soffset), so a load in the main loop is `s_mov m0; buffer_load_dwordx4 .. offen lds` with no VALU, and
//     rows past the end of the matrix read as zeros (hardware range check) instead of being clamped;
//   * a two-phase epilogue: every load (bias, residual, saved pre-activation) is issued before the first store, because
//     vmcnt counts stores too and interleaving them serialises on store latency (in-kernel stamps: epilogue 14.7k cycles
//     vs 27k for a 12-K-tile main loop before this change);
//   * static priority for the second half of the waves of an 8-wave workgroup (they lose arbitration otherwise and the
//     older half idles at the barrier).
// Tile configs (BM x BN, waves, stages): 128x128 / 4 / 1 (several workgroups per CU), 256x256 / 8 / 2 and
// 256x128 / 8 / 3 (one workgroup per CU, K-tile t+1 / t+2 in flight during the MFMAs of tile t).
// =====================================================================================
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource types exist in the device pass only
typedef int pk_i32x4 __attribute__((ext_vector_type(4)));

// buffer descriptor as four SGPR words for the inline-asm loads (raw buffer, 32-bit range-checked offsets: what lies
// past num_bytes reads as zero)
__device__ __forceinline__ pk_i32x4 pk_rsrc(const void* base, long num_bytes) {
    const unsigned long long a = (unsigned long long)base;
    pk_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    r[2] = __builtin_amdgcn_readfirstlane(num_bytes > 0x7fffffffL ? 0x7fffffff : (int)num_bytes);
    r[3] = 0x00020000;
    return r;
}

template <bool TR, int ROWS, int NTHREADS, int BKT>
struct DmaOperand {
    static constexpr int NLOAD = ROWS * BKT * 2 / (NTHREADS * 16);   // loads per thread per K-tile
    // K-contiguous operands: load j of a wave is 64 / (BKT / 8) rows below load j - 1 and the chunk swizzle depends on
    // the row only modulo 8, so one per-lane offset plus a scalar step serves all loads (register budget of the
    // 256x128 tile); K-strided operands keep one offset per load
    static constexpr int NVOFF = TR ? NLOAD : 1;
    pk_i32x4 rs;
    int voff[NVOFF];
    int jstep;
    int tile_off;    // byte offset of this workgroup's tile at k = 0
    int k_step;      // byte offset added per K-tile

    __device__ __forceinline__ void init(const bf16* base, int ld, int r0, int R, int K, int wave, int lane) {
        const long elems = !TR ? (long)(R - 1) * ld + K : (long)(K - 1) * ld + R;
        rs = pk_rsrc(base, elems * 2);
        jstep = !TR ? (BKT == 64 ? 8 : 16) * ld * 2 : 0;
#pragma unroll
        for (int j = 0; j < NVOFF; ++j) {
            const int s = (wave * NLOAD + j) * 64 + lane;
            if (!TR) {
                // BKT = 64: 128-byte rows, chunk ^ (row & 7); BKT = 32: 64-byte rows, chunk ^ ((row >> 2) & 3)
                const int row = BKT == 64 ? s >> 3 : s >> 2;
                const int chunk = BKT == 64 ? (s & 7) ^ (row & 7) : (s & 3) ^ ((row >> 2) & 3);
                voff[j] = (row * ld + chunk * 8) * 2;
            } else {
                constexpr int CPR = ROWS / 8;
                const int kr = s / CPR, slot = s % CPR;
                const int chunk = (slot & ~15) | ((slot & 15) ^ swz16(kr));
                voff[j] = (kr * ld + chunk * 8) * 2;
            }
        }
        tile_off = !TR ? r0 * ld * 2 : r0 * 2;
        k_step = !TR ? BKT * 2 : BKT * ld * 2;
    }
    // The loads are inline asm on purpose.  hipcc puts an `s_waitcnt vmcnt(0)` in front of the first ds_read that follows an
    // LDS-DMA *builtin* (it must assume the read aliases the pending LDS write), which drains a prefetched K-tile right after
    // it was issued: every multi-stage form of this kernel measured in rounds 1-2 was silently single-stage.  In asm the
    // compiler sees no memory dependence; every wait of the main loop is placed by hand (s_waitcnt vmcnt + barrier in front
    // of the reads of a stage, ILVLM_WG_BARRIER).  M0 (the LDS destination) is written and read inside the statement; the
    // instruction between an M0 write and the load is the wait state that pair needs.
    __device__ __forceinline__ void issue(int t, unsigned char* tile, int wave, int extra = 0) const {
        static_assert(NLOAD == 4 || NLOAD == 8, "four or eight 1 KiB pieces per wave and operand (128 / 256 operand rows, 256 threads)");
        const int soff = tile_off + t * k_step + extra;
        const unsigned lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)tile + wave * (NLOAD * 1024);
        const pk_i32x4 r4 = rs;
        if constexpr (TR) {
#pragma unroll
            for (int h = 0; h < NLOAD / 4; ++h)
                asm volatile(
                    "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
                    "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
                    "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v2], %[rs], %[so] offen lds\n\t"
                    "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v3], %[rs], %[so] offen lds"
                    :
                    : [v0] "v"(voff[TR ? 4 * h : 0]), [v1] "v"(voff[TR ? 4 * h + 1 : 0]), [v2] "v"(voff[TR ? 4 * h + 2 : 0]),
                      [v3] "v"(voff[TR ? 4 * h + 3 : 0]), [rs] "s"(r4), [so] "s"(soff), [lds] "s"(lds + h * 4096)
                    : "memory", "scc");
        } else {
#pragma unroll
            for (int h = 0; h < NLOAD / 4; ++h) {
                int tmp;
                asm volatile(
                    "s_mov_b32 m0, %[lds]\n\ts_mov_b32 %[t], %[so]\n\tbuffer_load_dwordx4 %[v0], %[rs], %[t] offen lds\n\t"
                    "s_add_u32 m0, m0, 0x400\n\ts_add_u32 %[t], %[t], %[js]\n\tbuffer_load_dwordx4 %[v0], %[rs], %[t] offen lds\n\t"
                    "s_add_u32 m0, m0, 0x400\n\ts_add_u32 %[t], %[t], %[js]\n\tbuffer_load_dwordx4 %[v0], %[rs], %[t] offen lds\n\t"
                    "s_add_u32 m0, m0, 0x400\n\ts_add_u32 %[t], %[t], %[js]\n\tbuffer_load_dwordx4 %[v0], %[rs], %[t] offen lds"
                    : [t] "=&s"(tmp)
                    : [v0] "v"(voff[0]), [rs] "s"(r4), [so] "s"(soff + h * 4 * jstep), [js] "s"(jstep), [lds] "s"(lds + h * 4096)
                    : "memory", "scc");
            }
        }
    }
};

// K-strided fp8 operand (weight gradients in fp8 mode: A = dY [tokens, out] e5m2, B = X [tokens, in] e4m3, the token index is
// the reduction).  K-tile = 128 k-rows x 128 bytes (= 128 operand rows): a DMA piece is 8 k-rows of 128 contiguous bytes.
// Fragments come from ds_read_b64_tr_b8, whose lane map was measured (benchmarks/micro/tr8_probe.hip): within a group of
// 16 lanes, lane s supplies the address of 8 contiguous bytes of block row s >> 1, columns 8 (s & 1) .. +7, and lane i
// receives column i of the 8 rows -- with block rows = k and columns = operand rows that is exactly the fp8 16x16x32 MFMA
// operand (lane l: row l & 15, k = 8 (l >> 4) + j), one read per fragment.  The 16-byte chunk c of k-row r is stored at
// slot c ^ ((r >> 1) & 7): the 16 k-rows a 32-lane half reads then cover all 64 banks once.
template <int ROWS, int NTHREADS>
struct DmaOperandTr8 {
    static_assert(ROWS == 128, "128-byte k-rows");
    static constexpr int KT = 128;
    static constexpr int NLOAD = KT * ROWS / (NTHREADS * 16);
    pk_i32x4 rs;
    int voff[NLOAD];
    int tile_off, k_step;
    __device__ __forceinline__ void init(const bf16* base, int ld, int r0, int R, int K, int wave, int lane) {
        rs = pk_rsrc(base, (long)(K - 1) * ld + R);
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) {
            const int s = (wave * NLOAD + j) * 64 + lane;
            const int kr = s >> 3, slot = s & 7;
            voff[j] = kr * ld + ((slot ^ ((kr >> 1) & 7)) << 4);
        }
        tile_off = r0;
        k_step = KT * ld;
    }
    __device__ __forceinline__ void issue(int t, unsigned char* tile, int wave, int extra = 0) const {     // asm: see DmaOperand
        static_assert(NLOAD == 4, "four 1 KiB pieces per wave and operand");
        const int soff = tile_off + t * k_step + extra;
        const unsigned lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)tile + wave * (NLOAD * 1024);
        asm volatile(
            "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v2], %[rs], %[so] offen lds\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v3], %[rs], %[so] offen lds"
            :
            : [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]), [v3] "v"(voff[3]), [rs] "s"(rs), [so] "s"(soff), [lds] "s"(lds)
            : "memory", "scc");
    }
};

__device__ __forceinline__ long tr8_frag(const unsigned char* tile, int r16, int k32, int lane) {
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2i lds_v2i;
    const int i = lane & 15, g = lane >> 4;
    const int kr = k32 + 8 * g + (i >> 1);
    const int chunk = (r16 >> 4) ^ ((kr >> 1) & 7);
    union { v2i v; long l; } u;
    u.v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i*)(tile + 128 * kr + 16 * chunk + 8 * (i & 1)));
    return u.l;
}
// The same fragment for the wide fp8 tile, addressed so that ONE per-lane register serves every row tile and k-step: the swizzle
// ((kr >> 1) & 7 = (4 g + (i >> 2)) & 7) does not depend on the k-step (32 k-rows = 0 mod 8), the chunk bits (4..6) are disjoint
// from the lane's other address bits when the tile is 128-byte aligned, hence
//     address(row tile rt, k-step ks) = (addr0 ^ (rt << 4)) + 4096 ks,    addr0 = tile + 128 (8 g + (i >> 1)) + 8 (i & 1) + 16 swz
// -- one XOR per row tile and an instruction offset, where tr8_frag's address arithmetic is loop-invariant per (rt, ks) and hipcc
// hoists all 48 of them out of the K loop (they were what spilled the 256-register kernel).
__device__ __forceinline__ unsigned tr8_addr0(const unsigned char* tile, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int swz = (4 * g + (i >> 2)) & 7;
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)tile + 128 * (8 * g + (i >> 1)) + 8 * (i & 1) + 16 * swz;
}
__device__ __forceinline__ long tr8_frag_w(unsigned addr0, int rt, int ks) {
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2i lds_v2i;
    union { v2i v; long l; } u;
    u.v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i*)(unsigned long long)((addr0 ^ (unsigned)(rt << 4)) + 4096u * ks));
    return u.l;
}
#endif

// Tile epilogue for SWAP fragments (lane owns row (l&15) and 4 consecutive columns 4*(l>>4).. of each 16x16 tile).
// Written that way a wave instruction touches 16 rows x 64 (fp32) / 32 (bf16) bytes.  The vector-memory request rate is
// what bounds this kernel, so the fragments are first transposed through a wave-private 8 KiB
// slice of the (now idle) operand LDS: afterwards lane l owns row (l>>4) of a 4-row group and columns 4*(l&15)..+3, and
// every load / store instruction of the epilogue covers 4 rows x 256 (fp32) / 128 (bf16) contiguous bytes -- a quarter
// of the requests.  Two passes of 32 rows; all loads of a half-pass are issued before its first store.
// Epilogue modes.  The epilogue runs once per tile, so its instruction footprint matters as much as its instruction
// count: with every variant inlined behind run-time branches and all loops unrolled the 256x256 kernel was 196 KB of
// code against a 64 KB instruction cache, and a workgroup that owns its CU stalled on instruction fetch for ~27 us per
// tile (K-scaling, benchmarks/p8_scaling.py).  The variant is therefore a template parameter picked by one switch, and
// the row-group loop is a real loop: the executed path of a tile is a few KB.
enum { EPI_GENERIC = 0, EPI_PLAIN, EPI_RES, EPI_QGELU, EPI_GELU, EPI_QGELU_BWD, EPI_GELU_BWD, EPI_POOLMAX };
#ifndef ILVLM_EPI_WIDE
#define ILVLM_EPI_WIDE 1             // 16-byte bf16 stores in the tile epilogue (see epilogue_pass)
#endif


// one pass (32 rows: row tiles 2P, 2P+1) of epilogue_tile; the pass index is a template parameter so that the accumulator
// array is only ever indexed with constants (a run-time pass loop sends all of it through scratch memory)
// RT = row tiles per pass: 2 (32 rows, 8 KiB of LDS per wave) or 1 (16 rows, 4 KiB: the persistent kernel, whose operand
// LDS is being refilled while the epilogue runs)
__device__ __forceinline__ void out8_store8(const ilvlm_gemm_epilogue& e, long off, f32x4 v0, f32x4 v1, float& amax);

// WIDE (bf16 outputs, with or without an fp8 copy): after the transpose a lane owns EIGHT consecutive columns of a row of an 8-row
// group (two ds_read_b128), so every global store is 16 bytes per lane -- 8 rows x 128 B per wave instruction, half the store
// instructions of the 4-column form.  In-kernel stamps put this epilogue at 10 B/cycle/CU whether or not the rest of the chip
// stores at the same time: it is bound by the number of store instructions (cdna_hip_programming.md T21), not by HBM.
// `n` is the lane's first column in the mapping in use; bias[1] is used by WIDE only (columns n + 4 .. n + 7).
template <int MODE, int TI, int TJ, int P, int RT = 2, bool WIDE = false>
__device__ __forceinline__ void epilogue_pass(const EpiArgs& ep, f32x4 (&acc)[TI][TJ], int m_base, int n, int lane, float alpha,
                                              unsigned char* wlds, const f32x4 (&bias2)[2], float& amax8) {
    const f32x4 bias = bias2[0];
    if constexpr (P < TI / RT) {
        const ilvlm_gemm_epilogue& e = ep.e;
        const int g = lane >> 4, c = lane & 15;
        // fragments of row tiles 2P, 2P+1 -> [32 rows][64 fp32] image, 16-byte chunk index XOR (row & 15)
#pragma unroll
        for (int ii = 0; ii < RT; ++ii)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int row = ii * 16 + c;
                *(f32x4*)(wlds + row * 256 + (((4 * j + g) ^ (row & 15)) << 4)) = acc[RT * P + ii][j];
            }
        // lanes read what OTHER lanes wrote: per thread the stores above and the loads below touch different addresses, so
        // the compiler is free to reorder them unless told otherwise (a release fence alone lets the loads move up).  The
        // LDS queue itself is in order.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if constexpr (MODE == EPI_POOLMAX) {
            // token max-pool of the codebook scores: lane = column; the 32 rows are walked in order, a running (value, token)
            // maximum per sequence is merged into pool_out by one 64-bit atomic max per sequence and column
            const int col = n - 4 * c + lane;
            int cur = -1;
            unsigned long long best = 0;
#pragma unroll 1
            for (int r = 0; r < 16 * RT; ++r) {
                const int m = m_base + P * (16 * RT) + r;
                if (m >= ep.M) break;                                  // wave-uniform
                const float v = alpha * *(const float*)(wlds + r * 256 + ((((lane >> 2) ^ (r & 15)) << 4) | ((lane & 3) << 2)));
                int sq, tok;
                if (e.pool_seq) { sq = e.pool_seq[m]; tok = m - e.pool_offs[sq]; }
                else { sq = m / e.pool_group; tok = m - sq * e.pool_group; }
                if (sq != cur) {
                    if (cur >= 0 && col < ep.N) atomicMax(e.pool_out + (long)cur * ep.N + col, best);
                    cur = sq;
                    best = 0;
                }
                const unsigned u = __float_as_uint(v);
                const unsigned key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);    // order-preserving float -> uint
                const unsigned long long cand = ((unsigned long long)key << 32) | (unsigned)(0x7fffffff - tok);
                best = cand > best ? cand : best;
            }
            if (cur >= 0 && col < ep.N) atomicMax(e.pool_out + (long)cur * ep.N + col, best);
        } else if constexpr (MODE == EPI_GENERIC) {
            // ragged tile, unaligned operands or a rare epilogue (row-indexed bias): one bounds-checked call per 4-row group
#pragma unroll 1
            for (int hk = 0; hk < 4 * RT; ++hk) {
                const int row = 4 * hk + g;
                const f32x4 v = *(const f32x4*)(wlds + row * 256 + ((c ^ (row & 15)) << 4));
                epilogue4<bf16>(ep, m_base + P * (16 * RT) + row, n, v, alpha, (e.out8 || e.out8_amax) ? &amax8 : nullptr);
            }
        } else if constexpr (WIDE) {
            static_assert(MODE != EPI_RES, "the residual epilogue writes fp32");
            const int g8 = lane >> 3, c8 = lane & 7;
            constexpr bool BWD = MODE == EPI_QGELU_BWD || MODE == EPI_GELU_BWD;
            constexpr bool FWD_ACT = MODE == EPI_QGELU || MODE == EPI_GELU;
            // the 64 x 64 wave tiles belong to kernels capped at 128 VGPRs (four workgroups per CU): the second half of the bias
            // is fetched per pass (an L1 hit) instead of living in four more registers across the passes -- held there it was
            // spilled to scratch and reloaded per pass anyway
            f32x4 bias_hi = bias2[1];
            if constexpr (TI <= 4) bias_hi = e.bias ? *(const f32x4*)(e.bias + n + 4) : (f32x4){0, 0, 0, 0};
#pragma unroll 1
            for (int h = 0; h < RT; ++h) {           // rows 16h .. 16h+15 of the pass: 2 instructions x 8 rows
                bf16x8 pre[2];
                long off[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {        // both loads of the half-pass before its first store (vmcnt counts both)
                    const int m = m_base + P * (16 * RT) + h * 16 + 8 * k + g8;
                    off[k] = map_row(m, e.out_group, e.out_skip) * (long)ep.ldc + n;
                    if constexpr (BWD) pre[k] = *(const bf16x8*)((const bf16*)e.aux + off[k]);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int row = h * 16 + 8 * k + g8;
                    bf16x8 o, u;
                    f32x4 vq[2];
                    // one 4-column half at a time, finished (activation, rounding) before the other is read: the 128-VGPR
                    // kernels (four workgroups per CU) have no room for eight activation chains in flight
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 v = *(const f32x4*)(wlds + row * 256 + (((2 * c8 + q) ^ (row & 15)) << 4));
                        v = fma4(v, alpha, q == 0 ? bias : bias_hi);
                        if constexpr (FWD_ACT) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                u[4 * q + j] = (bf16)v[j];
                                v[j] = MODE == EPI_QGELU ? quick_gelu(v[j]) : gelu_erf(v[j]);
                            }
                        }
                        if constexpr (BWD) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                v[j] *= MODE == EPI_QGELU_BWD ? quick_gelu_grad((float)pre[k][4 * q + j]) : gelu_erf_grad((float)pre[k][4 * q + j]);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[4 * q + j] = (bf16)v[j];
                        vq[q] = v;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (FWD_ACT) *(bf16x8*)((bf16*)e.aux + off[k]) = u;
                    if (ep.Cb) *(bf16x8*)(ep.Cb + off[k]) = o;                    // (null: only the fp8 copy of the result is kept)
                    if (e.out8 || e.out8_amax) out8_store8(e, off[k], vq[0], vq[1], amax8);
                }
            }
        } else {
#pragma unroll 1
            for (int h = 0; h < RT; ++h) {           // rows 16h .. 16h+15 of the pass: 4 instructions x 4 rows
                f32x4 pre[4];
                long off[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {        // every load of the half-pass before its first store (vmcnt counts both)
                    const int m = m_base + P * (16 * RT) + h * 16 + 4 * k + g;
                    off[k] = map_row(m, e.out_group, e.out_skip) * (long)ep.ldc + n;
                    if constexpr (MODE == EPI_RES) pre[k] = *(const f32x4*)(e.residual + off[k]);
                    if constexpr (MODE == EPI_QGELU_BWD || MODE == EPI_GELU_BWD) pre[k] = load4<bf16>((const bf16*)e.aux + off[k]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = h * 16 + 4 * k + g;
                    f32x4 v = *(const f32x4*)(wlds + row * 256 + ((c ^ (row & 15)) << 4));
                    v = fma4(v, alpha, bias);
                    if constexpr (MODE == EPI_QGELU || MODE == EPI_GELU) {
                        store4<bf16>((bf16*)e.aux + off[k], v);
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = MODE == EPI_QGELU ? quick_gelu(v[q]) : gelu_erf(v[q]);
                    }
                    if constexpr (MODE == EPI_QGELU_BWD || MODE == EPI_GELU_BWD) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            v[q] *= MODE == EPI_QGELU_BWD ? quick_gelu_grad(pre[k][q]) : gelu_erf_grad(pre[k][q]);
                    }
                    if constexpr (MODE == EPI_RES) v += pre[k];
                    if (!ep.Cf) {}
                    else if (e.out_dtype == ILVLM_F32) store4<float>(ep.Cf + off[k], v);
                    else store4<bf16>(ep.Cb + off[k], v);
                    if (e.out8 || e.out8_amax) out8_store(e, off[k], v, 4, amax8);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();             // the next pass overwrites the image
        epilogue_pass<MODE, TI, TJ, P + 1, RT, WIDE>(ep, acc, m_base, n, lane, alpha, wlds, bias2, amax8);
    }
}

// launch-uniform part of the choice of the 8-column (16-byte bf16 store) epilogue; the per-tile part is "the tile is whole"
__host__ __device__ __forceinline__ bool epi_wide_cfg(const ilvlm_gemm_epilogue& e) {
    return e.out_dtype == ILVLM_BF16 && !e.residual && !e.rowbias && !e.accumulate && !e.pool_out;
}
// fp8 copy of 8 stored values (one 8-byte store) + running max|value| for the amax: the 8-column form of out8_store
__device__ __forceinline__ void out8_store8(const ilvlm_gemm_epilogue& e, long off, f32x4 v0, f32x4 v1, float& amax) {
#pragma unroll
    for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fmaxf(fabsf(v0[i]), fabsf(v1[i])));
    if (!e.out8) return;
    const float s = e.out8_scale ? e.out8_scale[0] : 1.f;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 w;
    w[0] = fp8_pack4_fmt(e.out8_fmt, v0[0] * s, v0[1] * s, v0[2] * s, v0[3] * s);
    w[1] = fp8_pack4_fmt(e.out8_fmt, v1[0] * s, v1[1] * s, v1[2] * s, v1[3] * s);
    *(u32x2*)((unsigned char*)e.out8 + off) = w;
}

// bias_in: this lane's four bias values already in registers (the persistent kernel loads them ahead of the tile's main loop:
// a load issued HERE would have to be waited for with everything older in the queue, i.e. with the next tile's operand prefetch)
// POOL = false: the caller never has a pool epilogue (the streaming kernels); its code is left out
template <int TI, int TJ, int RT = 2, bool POOL = true, bool ALLOW_WIDE = true>
__device__ __forceinline__ void epilogue_tile(const EpiArgs& ep, f32x4 (&acc)[TI][TJ], int m_base, int n_base, int lane,
                                              float alpha, unsigned char* wlds, const f32x4* bias_in = nullptr) {
    const ilvlm_gemm_epilogue& e = ep.e;
    static_assert(TJ == 4 && (TI % RT) == 0, "64-column wave tiles, RT row tiles per pass");
    const bool whole = ep.vec_ok && !e.accumulate && m_base + TI * 16 <= ep.M && n_base + TJ * 16 <= ep.N;
    int mode = EPI_GENERIC;
    if (POOL && e.pool_out) mode = EPI_POOLMAX;
    else if (whole && !e.rowbias) {
        if (e.act == ILVLM_ACT_NONE) mode = e.residual ? EPI_RES : EPI_PLAIN;
        else if (!e.residual) mode = EPI_QGELU + (e.act - ILVLM_ACT_QUICKGELU);
    }
    // bf16 outputs of whole tiles: 8 columns per lane and 16-byte stores (ILVLM_EPI_WIDE=0 at build time for the A/B)
    const bool wide = ILVLM_EPI_WIDE && ALLOW_WIDE && ep.vec8_ok && epi_wide_cfg(e) && mode != EPI_GENERIC && mode != EPI_POOLMAX && mode != EPI_RES;
    const int n = wide ? n_base + 8 * (lane & 7) : n_base + 4 * (lane & 15);          // this lane's columns after the transpose
    f32x4 bias[2] = {(f32x4){0, 0, 0, 0}, (f32x4){0, 0, 0, 0}};
    if (bias_in) {                                   // preloaded by the caller in the mapping epi_wide_cfg() selects
        bias[0] = bias_in[0];
        if (wide) bias[1] = bias_in[1];
    } else if (mode != EPI_GENERIC && mode != EPI_POOLMAX && e.bias) {
        bias[0] = *(const f32x4*)(e.bias + n);
        if (wide && TI > 4) bias[1] = *(const f32x4*)(e.bias + n + 4);
    }
    float amax8 = 0.f;
#define ILVLM_EPI_CASE(M)                                                                                             \
    case M:                                                                                                           \
        if constexpr (ILVLM_EPI_WIDE && ALLOW_WIDE) {                                                                 \
            if (wide) { epilogue_pass<M, TI, TJ, 0, RT, true>(ep, acc, m_base, n, lane, alpha, wlds, bias, amax8); break; } \
        }                                                                                                             \
        epilogue_pass<M, TI, TJ, 0, RT, false>(ep, acc, m_base, n, lane, alpha, wlds, bias, amax8);                   \
        break
    switch (mode) {
        ILVLM_EPI_CASE(EPI_PLAIN);
        ILVLM_EPI_CASE(EPI_QGELU);
        ILVLM_EPI_CASE(EPI_GELU);
        ILVLM_EPI_CASE(EPI_QGELU_BWD);
        ILVLM_EPI_CASE(EPI_GELU_BWD);
        case EPI_RES: epilogue_pass<EPI_RES, TI, TJ, 0, RT>(ep, acc, m_base, n, lane, alpha, wlds, bias, amax8); break;
        case EPI_POOLMAX:
            if constexpr (POOL) epilogue_pass<EPI_POOLMAX, TI, TJ, 0, RT>(ep, acc, m_base, n, lane, alpha, wlds, bias, amax8);
            break;
        default: epilogue_pass<EPI_GENERIC, TI, TJ, 0, RT>(ep, acc, m_base, n, lane, alpha, wlds, bias, amax8); break;
    }
#undef ILVLM_EPI_CASE
    if (e.out8_amax) {          // one conditional atomic per wave and tile
        amax8 = wave_max(amax8);
        if (lane == 0) fp8_amax_raise(e.out8_amax, amax8);
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
// accumulate (split-K) epilogue: fragments (C^T orientation: lane owns row (l & 15), 4 consecutive columns) are
// transposed through a wave-private 8 KiB LDS image so that every atomic wave-instruction covers 256 contiguous bytes
// of one output row (the shape global float atomics run at full rate with)
// PLAIN: the caller is the only writer of the tile in this launch (slab reducer below): load-add-store instead of atomics
template <int TI, int TJ, int P = 0, bool PLAIN = false>
__device__ __forceinline__ void epilogue_acc_tile(const EpiArgs& ep, f32x4 (&acc)[TI][TJ], int m_base, int n_base, int lane,
                                                  float alpha, unsigned char* wlds) {
    static_assert(TJ == 4 && TI % 2 == 0, "64-column wave tile");
    if constexpr (P < TI / 2) {
        const int g = lane >> 4, c = lane & 15;
        const int n = n_base + lane;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int row = ii * 16 + c;
                *(f32x4*)(wlds + row * 256 + (((4 * j + g) ^ (row & 15)) << 4)) = acc[2 * P + ii][j];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (PLAIN && ep.vec_ok && ep.e.out_group == 0 && m_base + P * 32 + 32 <= ep.M && n_base + 64 <= ep.N) {
            // single writer, whole 32 x 64 slice: 16-byte load-add-store, 4 rows x 256 B per wave instruction -- 16 vector-memory
            // instructions per pass instead of the 64 of the one-dword-per-lane form (the epilogue is bound by their number)
#pragma unroll 2
            for (int k = 0; k < 8; ++k) {
                const int row = 4 * k + g;
                float* dst = ep.Cf + (long)(m_base + P * 32 + row) * ep.ldc + n_base + 4 * c;
                const f32x4 v = *(const f32x4*)(wlds + row * 256 + ((c ^ (row & 15)) << 4));
                *(f32x4*)dst = *(const f32x4*)dst + v * alpha;
            }
        } else if (m_base + P * 32 < ep.M) {
#pragma unroll 4
            for (int r = 0; r < 32; ++r) {
                const int m = m_base + P * 32 + r;
                const float v = *(const float*)(wlds + r * 256 + ((((lane >> 2) ^ (r & 15)) << 4) | ((lane & 3) << 2)));
                if (m < ep.M && n < ep.N) {
                    float* dst = ep.Cf + map_row(m, ep.e.out_group, ep.e.out_skip) * (long)ep.ldc + n;
                    if constexpr (PLAIN) *dst += v * alpha;
                    else atomicAdd(dst, v * alpha);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        epilogue_acc_tile<TI, TJ, P + 1, PLAIN>(ep, acc, m_base, n_base, lane, alpha, wlds);
    }
}
#endif

#ifndef ILVLM_PKP_WIDE
#define ILVLM_PKP_WIDE 0
#endif
#ifndef ILVLM_PK_PRIO
#define ILVLM_PK_PRIO 1               // wave priority of the streaming kernel (see gemm_bf16_pk_kernel)
#endif
#ifndef ILVLM_FP8_SCALED_MFMA
#define ILVLM_FP8_SCALED_MFMA 1      // K-contiguous fp8 operands (forward, input gradient): -9 % GEMM time, fp8 step +3.9 %
#endif
#ifndef ILVLM_FP8_SCALED_WGRAD
#define ILVLM_FP8_SCALED_WGRAD 0     // K-strided fp8 operands (weight gradient): slower, see the main loop
#endif
// FP8 = 0: bf16 operands.  FP8 = 1 / 2: fp8 operands (OCP e4m3; 2: the A operand is e5m2, for gradients) addressed as if two
// fp8 elements were one bf16 element -- the host passes K / 2, lda / 2, ldb / 2 -- so tiles, DMA and the LDS images are
// byte for byte those of the bf16 kernel with a K-tile of 128 instead of 64; only the MFMA differs: every 16-byte
// fragment read feeds two v_mfma_f32_16x16x32_fp8 (8 bytes = 8 k each; both operands split their bytes the same way, so
// the products pair up whatever order k is visited in).  Half the operand bytes per FLOP of the bf16 kernel, which is
// what bounds it.  K-contiguous operands only.
// The body is shared by the single-problem kernel (gemm_bf16_dma_kernel) and the grouped weight-gradient kernel
// (wgrad_group_kernel): `wg_linear` is the workgroup's index within ITS problem, already passed through xcd_remap.
template <bool TA, bool TB, bool SWAP, int DBM, int DBN, int WM, int WN, int NSTAGE, int BKT, int FP8 = 0>
__device__ __forceinline__ void dma_gemm_body(const bf16* __restrict__ A, int lda, const bf16* __restrict__ B, int ldb, int K,
                                              int tiles_m, int tiles_n, int split_k, const EpiArgs& ep, int wg_linear) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = 64 * WM * WN;
    constexpr int TI = DBM / WM / 16, TJ = DBN / WN / 16;      // 16x16 tiles per wave
    constexpr int A_BYTES = DBM * BKT * 2, STAGE = (DBM + DBN) * BKT * 2;
    // FP8 = 3: K-strided fp8 operands (fp8 weight gradients), K-tile = 128 k-rows, 128 x 128 tile.  FP8 = 4 (round 4): the same
    // operands on a 256 x 128 tile (128 x 64 per wave) and the block-scaled MFMA: the A operand is TWO 128-row sub-operands in
    // the proven LDS layout (rows m0.. and m0 + 128.., 16 KiB each; the waves of row half wm read only theirs), so a K-tile is
    // 48 KiB for twice the products of the 128 x 128 tile's 32 KiB -- at the fp8 MFMA rate the 128 x 128 tile is bound by its
    // operand traffic (DMA into LDS + transposing reads out of it), not by the matrix pipe.
    constexpr bool TR8 = FP8 == 3 || FP8 == 4;
    constexpr bool TR8W = FP8 == 4;
    static_assert(!TR8 || (TA && TB && !SWAP && DBN == 128 && BKT == 64 && DBM == (TR8W ? 256 : 128)), "fp8 weight-gradient form");
    static_assert(!TR8W || (WM == 2 && WN == 2 && NSTAGE == 1), "wide fp8 weight-gradient tile: 2 x 2 waves, single stage");
    constexpr int KTILE = TR8 ? 128 : BKT;
    // 256-row K-strided A operands (the wide weight-gradient tiles, bf16 and fp8) are held as TWO 128-row sub-operands in the
    // LDS layout of the 128 x 128 kernel -- the second is the first, 128 rows further along the k-rows -- and the waves of row
    // half wm read only theirs: no new layout, no new bank-conflict analysis
    constexpr bool SPLIT_A = TA && TB && !SWAP && DBM == 256;
    constexpr int A_ROWS = SPLIT_A ? 128 : DBM;
    typedef typename std::conditional<TR8, DmaOperandTr8<128, NT>, DmaOperand<TA, A_ROWS, NT, BKT>>::type OpA;
    typedef typename std::conditional<TR8, DmaOperandTr8<DBN, NT>, DmaOperand<TB, DBN, NT, BKT>>::type OpB;
    constexpr int LOADS = OpA::NLOAD * (SPLIT_A ? 2 : 1) + OpB::NLOAD;
    constexpr int A_EXTRA = TR8 ? 128 : 256;          // byte offset of operand row 128 inside a k-row (fp8 / bf16)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int wg = wg_linear;
    const int tn = wg % tiles_n; wg /= tiles_n;
    // xcd_remap hands each XCD a contiguous range of wg.  Split-K (weight-gradient) launches order it K-slice major,
    // so one XCD's L2 sees one K-slice of both operands (each operand row is fetched by ~one XCD instead of all 8);
    // the other launches keep a tile's K-slices adjacent.
    int z, tm;
    if (!SWAP) { tm = wg % tiles_m; z = wg / tiles_m; }
    else { z = wg % split_k; tm = wg / split_k; }
    int tn2 = tn;
    if (SWAP && ep.tile_group > 0 && split_k == 1) {
        // L2 blocking of the tile walk.  xcd_remap hands an XCD a contiguous range of this linear order, i.e. a band of tile
        // rows; walked column-fastest, the band sweeps ALL of B once per tile row, and B (4.7 MB at 3072 x 768) does not fit the
        // XCD's 4 MiB L2 next to the band's A rows -- 13 % of the operand requests missed L2 (TCC counters).  So: row bands of
        // ceil(tiles_m / bands) tile rows, inside a band column groups of tile_group tiles, inside a group row-major: a band's
        // A rows (2.4 MB) and a group's B columns (0.8 MB at 4 tiles) stay resident.  fc forward 84.8 -> 78.3 us, FDT scores
        // 79.4 -> 71.7 us, +0.6...1.1 % of the step (bf16 and fp8).  Only for K <= 1024: beyond, a band's rows exceed L2 anyway.
        const int idx = tm * tiles_n + tn, Hb = (tiles_m + ep.tile_bands - 1) / ep.tile_bands, G = ep.tile_group;
        const int band = idx / (Hb * tiles_n), hb = min(Hb, tiles_m - band * Hb);
        const int r = idx - band * Hb * tiles_n, full = tiles_n / G;
        int g, gw, rr;
        if (r < full * hb * G) { g = r / (hb * G); gw = G; rr = r - g * hb * G; }
        else { g = full; gw = tiles_n - full * G; rr = r - full * hb * G; }
        tm = band * Hb + rr / gw;
        tn2 = g * G + rr % gw;
    }
    const int m0 = tm * DBM, n0 = tn2 * DBN;
    const int nt_total = (K + KTILE - 1) / KTILE;      // a partial last tile only with two K-strided operands (host check)
    const int per = (nt_total + split_k - 1) / split_k;
    const int t_begin = z * per, t_end = min(nt_total, t_begin + per);
    if (t_begin >= t_end) return;

    OpA opa; OpB opb;
    opa.init(A, lda, m0, ep.M, K, wave, lane);
    opb.init(B, ldb, n0, ep.N, K, wave, lane);

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    const bool rowsum = !SWAP && !TR8W && ep.e.a_rowsum != nullptr && tn == 0 && wn == 0;
    f32x4 accb[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) accb[i] = (f32x4){0, 0, 0, 0};
    const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};

    if (WM * WN == 8 && NSTAGE > 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);

    // prologue: NSTAGE-1 tiles in flight
#pragma unroll
    for (int d = 0; d < NSTAGE - 1; ++d)
        if (t_begin + d < t_end) {
            opa.issue(t_begin + d, smem_raw + d * STAGE, wave);
            if constexpr (SPLIT_A) opa.issue(t_begin + d, smem_raw + d * STAGE + A_BYTES / 2, wave, A_EXTRA);
            opb.issue(t_begin + d, smem_raw + d * STAGE + A_BYTES, wave);
        }
    int st = 0;
#ifdef ILVLM_GEMM_STAMPS
    unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_comp = 0;
    STAMP(t_start);
#endif
    for (int t = t_begin; t < t_end; ++t) {
#ifdef ILVLM_GEMM_STAMPS
        unsigned long long q3 = 0;
#endif
        if (NSTAGE == 1) {
            STAMP(q0);
#if ILVLM_GEMM_ABLATE != 2
            opa.issue(t, smem_raw, wave);
            if constexpr (SPLIT_A) opa.issue(t, smem_raw + A_BYTES / 2, wave, A_EXTRA);      // rows m0 + 128 ..
            opb.issue(t, smem_raw + A_BYTES, wave);
#endif
            STAMP(q1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(q2);
            ILVLM_WG_BARRIER();
#ifdef ILVLM_GEMM_STAMPS
            q3 = stamp();
#endif
            STAMP_ADD(c_issue, q0, q1); STAMP_ADD(c_wait, q1, q2); STAMP_ADD(c_bar, q2, q3);
        } else {
            STAMP(r0);
            if (NSTAGE == 3 && t + 1 < t_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(r1);
            ILVLM_WG_BARRIER();
            STAMP(r2);
            STAMP_ADD(c_wait, r0, r1); STAMP_ADD(c_bar, r1, r2);
            // NSTAGE == 3: the two waves of a SIMD issue their DMA at opposite ends of the K-tile, so one wave's load
            // issue (~100 cycles per instruction) overlaps the other's MFMAs instead of both stalling together
            if (!(NSTAGE == 3 && wave >= 4)) {
                const int tn_ = t + NSTAGE - 1;
                if (tn_ < t_end) {
                    const int sn = st == 0 ? NSTAGE - 1 : st - 1;     // (st + NSTAGE - 1) % NSTAGE
                    opa.issue(tn_, smem_raw + sn * STAGE, wave);
                    if constexpr (SPLIT_A) opa.issue(tn_, smem_raw + sn * STAGE + A_BYTES / 2, wave, A_EXTRA);
                    opb.issue(tn_, smem_raw + sn * STAGE + A_BYTES, wave);
                }
            }
#ifdef ILVLM_GEMM_STAMPS
            q3 = stamp();
            STAMP_ADD(c_issue, r2, q3);
#endif
        }
        const unsigned char* as = smem_raw + st * STAGE;
        const unsigned char* bs = as + A_BYTES;
#if ILVLM_GEMM_ABLATE == 1
        if (K < 0)
#endif
        if constexpr (TR8 && (ILVLM_FP8_SCALED_WGRAD || TR8W)) {
            // scaled MFMA over the whole 128-row K-tile (see the K-contiguous form below): lane (g, byte 8 s + j) holds
            // k = 32 s + 8 g + j of its operand row, for both operands.  On the 128 x 128 tile this measured SLOWER than the
            // non-scaled form (weight gradients 470 vs 437 us per block pair: four transposing reads per fragment in front of
            // four MFMAs), so there it is compiled out by default; the wide tile (FP8 = 4) runs eight row tiles against the four
            // B fragments a wave holds.
            typedef int v8i __attribute__((ext_vector_type(8)));
            union F8 { long l[4]; v8i v; };
            const unsigned char* asw = TR8W ? as + wm * 16384 : as;        // this wave's 128-row A sub-operand
            const int arow0 = TR8W ? 0 : wm * (TI * 16);
            const unsigned a_addr0 = tr8_addr0(asw, lane), b_addr0 = tr8_addr0(bs, lane);
            F8 fb8[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fb8[j].l[ks] = tr8_frag_w(b_addr0, wn * TJ + j, ks);
            F8 ones8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ones8.l[ks] = 0x3838383838383838L;       // e4m3 1.0
            // (the wide tile carries no row sums: 32 more accumulator registers do not fit beside its 128, and a wave-uniform
            // `if (rowsum)` around a fifth MFMA turned the accumulators into phi webs that spilled ~100 registers; the host gives
            // the first 128 output columns -- the tiles that carry the bias gradient -- to the 128 x 128 kernel instead)
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                F8 fa8;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa8.l[ks] = tr8_frag_w(a_addr0, (arow0 >> 4) + i, ks);
#pragma unroll
                for (int j = 0; j < TJ; ++j)      // first operand e4m3 (x), second e5m2 (dy)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb8[j].v, fa8.v, acc[i][j], 0, 1, 0, 0x7f7f7f7f, 0,
                                                                                 0x7f7f7f7f);
                if constexpr (!TR8W) {
                    if (rowsum)
                        accb[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones8.v, fa8.v, accb[i], 0, 1, 0, 0x7f7f7f7f, 0,
                                                                                   0x7f7f7f7f);
                }
            }
        } else if constexpr (TR8) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                long fa8[TI], fb8[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) fa8[i] = tr8_frag(as, wm * (TI * 16) + i * 16, ks * 32, lane);
#pragma unroll
                for (int j = 0; j < TJ; ++j) fb8[j] = tr8_frag(bs, wn * (TJ * 16) + j * 16, ks * 32, lane);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(fb8[j], fa8[i], acc[i][j], 0, 0, 0);
                if (rowsum) {     // e4m3 1.0 = 0x38
#pragma unroll
                    for (int i = 0; i < TI; ++i)
                        accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(0x3838383838383838L, fa8[i], accb[i], 0, 0, 0);
                }
            }
        } else if constexpr ((FP8 == 1 || FP8 == 2) && ILVLM_FP8_SCALED_MFMA) {
            // The block-scaled MFMA with all scales 2^0 as a plain fp8 MFMA: v_mfma_scale_f32_16x16x128_f8f6f4 takes the whole
            // 128-byte K-tile of a fragment row per instruction (32 bytes per lane) and issues at twice the rate of the four
            // non-scaled 16x16x32 fp8 instructions it replaces.  Both operands split their k bytes over (lane >> 4, byte)
            // the same way -- the two 16-byte reads the non-scaled form does, concatenated -- so the products pair up.
            typedef int v8i __attribute__((ext_vector_type(8)));
            static_assert(BKT == 64, "one instruction per 128-byte row");
            union F8 { struct { bf16x8 lo, hi; } h; v8i v; };
            F8 fb8[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                fb8[j].h.lo = p8_frag<TB, DBN, BKT>(bs, wn * (TJ * 16) + j * 16, 0, lane);
                fb8[j].h.hi = p8_frag<TB, DBN, BKT>(bs, wn * (TJ * 16) + j * 16, 32, lane);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                F8 fa8;
                fa8.h.lo = p8_frag<TA, DBM, BKT>(as, wm * (TI * 16) + i * 16, 0, lane);
                fa8.h.hi = p8_frag<TA, DBM, BKT>(as, wm * (TI * 16) + i * 16, 32, lane);
#pragma unroll
                for (int j = 0; j < TJ; ++j)      // cbsz / blgp: 0 = e4m3, 1 = e5m2; scale bytes 0x7f = 2^0
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb8[j].v, fa8.v, acc[i][j], 0, FP8 == 2 ? 1 : 0, 0,
                                                                                 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        } else
#pragma unroll
        for (int ks = 0; ks < BKT / 32; ++ks) {
            bf16x8 fa[TI], fb[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
                fa[i] = p8_frag<TA, A_ROWS, BKT>(SPLIT_A ? as + wm * (A_BYTES / 2) : as, (SPLIT_A ? 0 : wm * (TI * 16)) + i * 16, ks * 32, lane);
#pragma unroll
            for (int j = 0; j < TJ; ++j) fb[j] = p8_frag<TB, DBN, BKT>(bs, wn * (TJ * 16) + j * 16, ks * 32, lane);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {    // C^T fragments: lane owns row (l & 15), 4 consecutive columns
                    if constexpr (FP8 == 0 || FP8 == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    } else {
                        union { bf16x8 v; long l[2]; } ua, ub;
                        ua.v = fa[i]; ub.v = fb[j];
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            if constexpr (FP8 == 1)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(ub.l[hh], ua.l[hh], acc[i][j], 0, 0, 0);
                            else
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(ub.l[hh], ua.l[hh], acc[i][j], 0, 0, 0);
                        }
                    }
                }
            if (rowsum) {                         // every register of lane c = row sum of operand row c
#pragma unroll
                for (int i = 0; i < TI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], accb[i], 0, 0, 0);
            }
        }
#ifdef ILVLM_GEMM_STAMPS
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[TI - 1][TJ - 1]));
        STAMP(q4);
#endif
        if (NSTAGE == 3 && wave >= 4) {
            const int tn_ = t + 2;
            if (tn_ < t_end) {
                const int sn = st == 0 ? 2 : st - 1;
                opa.issue(tn_, smem_raw + sn * STAGE, wave);
                if constexpr (SPLIT_A) opa.issue(tn_, smem_raw + sn * STAGE + A_BYTES / 2, wave, A_EXTRA);
                opb.issue(tn_, smem_raw + sn * STAGE + A_BYTES, wave);
            }
        }
        if (NSTAGE == 1) ILVLM_WG_BARRIER();
        else st = st == NSTAGE - 1 ? 0 : st + 1;
#ifdef ILVLM_GEMM_STAMPS
        STAMP(q5);
        if (NSTAGE == 1) { STAMP_ADD(c_comp, q3, q4); STAMP_ADD(c_bar, q4, q5); }
        else { STAMP_ADD(c_comp, q3, q4); STAMP_ADD(c_issue, q4, q5); }
#endif
    }
#ifdef ILVLM_GEMM_STAMPS
    STAMP(t_loop_end);
#endif
    if (WM * WN == 8 && NSTAGE > 1 && wave >= 4) __builtin_amdgcn_s_setprio(0);

    float alpha = ep.e.alpha;
    if (ep.e.alpha_ptr) alpha *= *ep.e.alpha_ptr;
    if (ep.e.alpha_ptr2) alpha *= *ep.e.alpha_ptr2;
    const int mw = m0 + wm * (TI * 16), nw = n0 + wn * (TJ * 16);
    // (256-register kernels: the epilogue's per-lane address arithmetic must not be hoisted above the K loop and held across it --
    // it spilled 67 registers of the wide fp8 tile; an opaque copy of the lane index pins it here.  As gemm_bf16_pkp_kernel.)
    int lane_e = lane;
    if constexpr (DBM * DBN > 128 * 128) asm volatile("" : "+v"(lane_e));
#if ILVLM_GEMM_ABLATE == 3
    if (acc[0][0][0] != 12345.678f) return;
#endif
    // every wave must be done reading the operand tiles before any wave overwrites them with its fragments
    __syncthreads();
    if (SWAP) {
        epilogue_tile<TI, TJ>(ep, acc, mw, nw, lane, alpha, smem_raw + wave * 8192);
    } else {
        // split-K accumulate.  Default: fp32 atomics in whole 256-byte row segments (epilogue_acc_tile).  With a caller-provided
        // workspace (ep.e.splitk_ws): every K-slice stores its tile as a slab with plain 16-byte stores, takes a ticket, and
        // the workgroup that draws the last ticket of the tile adds the slabs up in slice order and is the only one to
        // touch C -- the in-launch combine of cdna_hip_programming.md ("one agent-scope release + one agent-scope acquire
        // per tile episode"): 28 MB of atomics at 1.3 TB/s become plain stores and loads, and the sum no longer depends on
        // arrival order (bit-reproducible weight gradients).
        if (ep.e.splitk_ws && split_k > 1) {
            // Hand-off form (cdna_hip_programming.md, guideline 16, R1): the slabs are stored WRITE-THROUGH (sc1) and read with
            // sc1 loads, so neither a release nor an acquire fence is needed -- the fenced form wrote back / invalidated whole
            // L2s 432 times per launch and cost the step 8.6 %.
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const int tile_id = tm * tiles_n + tn;
            const long nslab = (long)tiles_m * tiles_n * split_k;
            __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(ep.e.splitk_ws, 0, (int)(nslab * (DBM * DBN * 4)), 0x00020000);
            const int slab0 = tile_id * split_k * (DBM * DBN * 4);          // byte offsets (workspace < 2 GiB, host check)
            const int mine = slab0 + z * (DBM * DBN * 4);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    union { f32x4 f; u32x4 u; } x;
                    x.f = acc[i][j];
                    __builtin_amdgcn_raw_buffer_store_b128(x.u, ws, mine + ((i * TJ + j) * NT + tid) * 16, 0, 16);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave drains its stores
            __syncthreads();
            if (tid == 0) {
                const int old = __hip_atomic_fetch_add(ep.e.splitk_cnt + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == split_k - 1;
                if (last) __hip_atomic_store(ep.e.splitk_cnt + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
                *(volatile int*)smem_raw = last;        // broadcast through the one LDS array (the operand tiles are dead)
            }
            __syncthreads();
            const int last = *(volatile int*)smem_raw;
            __syncthreads();                            // wave 0's transpose slice starts at the flag word
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // compiler ordering only: loads stay below the ticket
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
                for (int zz = 0; zz < split_k; ++zz) {  // fixed order, own slab included; EVERY load of a slab is sc1
                    const int sl = slab0 + zz * (DBM * DBN * 4);
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < TJ; ++j) {
                            union { f32x4 f; u32x4 u; } x;
                            x.u = __builtin_amdgcn_raw_buffer_load_b128(ws, sl + ((i * TJ + j) * NT + tid) * 16, 0, 16);
                            acc[i][j] += x.f;
                        }
                }
                epilogue_acc_tile<TI, TJ, 0, true>(ep, acc, mw, nw, lane_e, alpha, smem_raw + wave * 8192);
            }       // (the bias gradient of every slice still goes out below, as atomics on M floats)
        } else if (ep.plain_acc && split_k == 1) {
            // the only writer of this tile (grouped weight gradients at one K-slice): no atomics
            epilogue_acc_tile<TI, TJ, 0, true>(ep, acc, mw, nw, lane_e, alpha, smem_raw + wave * 8192);
        } else {
            epilogue_acc_tile<TI, TJ>(ep, acc, mw, nw, lane_e, alpha, smem_raw + wave * 8192);
        }
        if (rowsum && lane_e < 16) {
            // fp8 operands: the row sums are sums of quantised values, de-quantised by the A operand's scale alone
            const float ra = (FP8 != 0 && ep.e.alpha_ptr) ? *ep.e.alpha_ptr : 1.f;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int m = mw + i * 16 + lane_e;
                if (m < ep.M) atomicAdd(ep.e.a_rowsum + m, accb[i][0] * ra);
            }
        }
    }
#ifdef ILVLM_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(t_end_);
    if (lane == 0 && blockIdx.x < 4096) {
        unsigned long long* o = g_stamps + ((long)blockIdx.x * 8 + wave) * 6;
        o[0] = c_wait; o[1] = c_bar; o[2] = c_issue; o[3] = c_comp; o[4] = t_loop_end - t_start; o[5] = t_end_ - t_loop_end;
    }
#endif
#endif   // __HIP_DEVICE_COMPILE__
}

template <bool TA, bool TB, bool SWAP, int DBM, int DBN, int WM, int WN, int NSTAGE, int BKT, int FP8 = 0>
__global__ __launch_bounds__(64 * WM * WN, (DBM * DBN > 128 * 128 && WM * WN == 4) ? 2 : (WM * WN == 4 ? (SWAP ? 4 : 3) : (NSTAGE == 1 ? (SWAP ? 4 : 3) : 2))) void gemm_bf16_dma_kernel(const bf16* __restrict__ A, int lda,
                                                                     const bf16* __restrict__ B, int ldb, int K, int tiles_m,
                                                                     int tiles_n, int split_k, EpiArgs ep) {
    dma_gemm_body<TA, TB, SWAP, DBM, DBN, WM, WN, NSTAGE, BKT, FP8>(A, lda, B, ldb, K, tiles_m, tiles_n, split_k, ep,
                                                                    xcd_remap(blockIdx.x, tiles_m * tiles_n * split_k));
}

// Grouped weight gradients: the (up to ILVLM_WGRAD_GROUP_MAX) products gW_p[N_p, K_p] += dY_p^T X_p of one transformer block --
// same contraction length (the block's token rows), different outputs -- as ONE launch over the concatenated tile lists.
// Why: launched one by one each product needs split-K slices to fill the chip (16..144 tiles of 128 x 128 against 512 workgroup
// slots), and the slices meet in fp32 atomics on gW: 28 MB per ViT-B/32 fc product at the 1.3 TB/s atomics run at, 20 % of
// that kernel's time, all workgroups reaching their epilogue together.  Grouped, the four products of a ViT-B/32 block are 432
// tiles: one K-slice each fills the chip, every tile has one writer (plain load-add-store), and four launch tails become one.
struct GroupProblem {
    const bf16* A;       // dY [rows, N] (K-strided: the contraction runs over rows)
    const bf16* B;       // X  [rows, K]
    int lda, ldb, K, tiles_m, tiles_n, split_k, wg_begin, pad_;
    EpiArgs ep;
};
struct GroupArgs {
    GroupProblem p[ILVLM_WGRAD_GROUP_MAX];
    int count, total;
};

template <int NSTAGE, int FP8, int TM = 128>
__global__ __launch_bounds__(256, TM == 256 ? 2 : (NSTAGE == 1 ? 3 : 2)) void wgrad_group_kernel(GroupArgs g) {
    const int wg = xcd_remap(blockIdx.x, g.total);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < ILVLM_WGRAD_GROUP_MAX; ++i)
        if (i < g.count && wg >= g.p[i].wg_begin) pi = i;
    const GroupProblem& P = g.p[pi];
    dma_gemm_body<true, true, false, TM, 128, 2, 2, NSTAGE, 64, FP8>(P.A, P.lda, P.B, P.ldb, P.K, P.tiles_m, P.tiles_n, P.split_k,
                                                                    P.ep, wg - P.wg_begin);
}

// =====================================================================================
// bf16 "streaming" kernel (round 3): store-type GEMMs C[M,N] = A[M,K] . Bop[N,K]^T whose B operand is a WEIGHT, i.e. can be
// kept pre-packed in MFMA fragment order (ilvlm_gemm_pack_b): the forward products and input gradients of the towers.
//
// What bounded the direct-to-LDS kernel above (benchmarks/micro/l2_paths.hip):
//   * the vector-memory path of a CU delivers 64 B/clk from L2 by LDS-DMA and by VGPR loads alike (130-139 GB/s per CU,
//     34 TB/s chip) -- the 61 GB/s per CU that kernel reaches is its issue -> wait -> compute serialisation (a K-tile period
//     is load latency PLUS compute), not a ceiling of the DMA path;
//   * it cannot double-buffer: 2 x 32 KiB x 4 workgroups exceeds the 160 KiB of LDS, and with fewer workgroups the bytes in
//     flight per CU stay the same;
//   * and the multi-stage variants of rounds 1-2 never actually pipelined: hipcc puts an `s_waitcnt vmcnt(0)` in front of the
//     first ds_read that follows an LDS-DMA *builtin* (it must assume the read aliases the pending LDS write), so the
//     prefetched tile was drained right after it was issued.
// Design:
//   * only A (activations) goes through LDS: 128 rows x 64 k = 16 KiB per stage, TWO stages = 32 KiB per workgroup, so four
//     workgroups per CU double-buffer inside the same LDS budget that gave the old kernel one stage;
//   * B goes global -> VGPR directly: the packed copy makes every wave load one contiguous KiB (fragment-shaped loads from
//     the row-major weight run at 38 GB/s per CU, a quarter of the rate), a wave owns 64 output columns and nobody else
//     in the workgroup needs them, and the registers double-buffer B (K-tile t+1 lands while t is multiplied);
//   * a wave owns 128 rows x 64 columns (8 x 4 accumulator tiles, 2 waves per SIMD): half the LDS fragment reads per
//     FLOP of the 64 x 64 wave tiles; WN waves side by side: workgroup tile 128 x (64 WN);
//   * every load of the main loop is inline asm and every wait is placed by hand: ONE counted `s_waitcnt vmcnt` + ONE barrier
//     per K-tile, both at the top of the step; A runs two K-tiles ahead through a THREE-stage ring, B one K-tile ahead in its
//     second register set, and the loads are issued one by one between the groups of four MFMAs (schedule, hazards and the
//     measurements behind it: at the main loop below).  The first form of this kernel -- two stages, everything one K-tile
//     ahead, the 16 loads issued as a block in front of the MFMAs -- is in the git history: 858 us against 801 us per
//     ViT + text block pair of forward + input-gradient launches.
// Same MFMA sequence per output element as the direct-to-LDS kernel: results are bit-identical.
// =====================================================================================
#if defined(__HIP_DEVICE_COMPILE__)
// NP LDS-DMA pieces (1 KiB each: 8 operand rows x 128 B) of one wave: piece j lands at lds + 1024 j and reads 8 rows
// further down (soffset + j * jstep).  M0 is written and read inside the one statement; the s_add between an M0 write and
// the load is the wait state that pair needs.
// `ok` == 0 skips the loads INSIDE the statement (a scalar branch): the compiler sees straight-line code, which keeps the
// accumulators of the loop in place (a branch around the statement gave phi webs, copies and spills).
template <int NP>
__device__ __forceinline__ void pk_dma(pk_i32x4 rs, int voff, int soff, int jstep, unsigned lds, int ok) {
    int t;
#define PK_DMA_NEXT "s_add_u32 m0, m0, 0x400\n\ts_add_u32 %[t], %[t], %[js]\n\tbuffer_load_dwordx4 %[vo], %[rs], %[t] offen lds\n\t"
    if constexpr (NP == 8) {
        asm volatile(
            "s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lpk_dma_skip%=\n\t"
            "s_mov_b32 m0, %[lds]\n\ts_mov_b32 %[t], %[soff]\n\tbuffer_load_dwordx4 %[vo], %[rs], %[t] offen lds\n\t"
            PK_DMA_NEXT PK_DMA_NEXT PK_DMA_NEXT PK_DMA_NEXT PK_DMA_NEXT PK_DMA_NEXT PK_DMA_NEXT
            ".Lpk_dma_skip%=:"
            : [t] "=&s"(t)
            : [vo] "v"(voff), [rs] "s"(rs), [soff] "s"(soff), [js] "s"(jstep), [lds] "s"(lds), [ok] "s"(ok)
            : "memory", "scc");
    } else {
        static_assert(NP == 4, "8 (two waves) or 4 (four waves) pieces per wave");
        asm volatile(
            "s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lpk_dma_skip%=\n\t"
            "s_mov_b32 m0, %[lds]\n\ts_mov_b32 %[t], %[soff]\n\tbuffer_load_dwordx4 %[vo], %[rs], %[t] offen lds\n\t"
            PK_DMA_NEXT PK_DMA_NEXT PK_DMA_NEXT
            ".Lpk_dma_skip%=:"
            : [t] "=&s"(t)
            : [vo] "v"(voff), [rs] "s"(rs), [soff] "s"(soff), [js] "s"(jstep), [lds] "s"(lds), [ok] "s"(ok)
            : "memory", "scc");
    }
#undef PK_DMA_NEXT
}

// the eight B fragments of a wave's K-tile (4 column tiles x 2 k-steps) from the packed copy: whole-KiB loads, the k-step
// in the instruction offset.  Early-clobber outputs: the first load may write before the last has read its operands.
// The registers are NOT valid behind this statement -- only behind the caller's s_waitcnt + pk_landed().
__device__ __forceinline__ void pk_load_b(bf16x8 (&b)[4][2], pk_i32x4 rs, int voff, int s0, int s1, int s2, int s3, int ok) {
    asm volatile(
        "s_cmp_eq_u32 %14, 0\n\ts_cbranch_scc1 .Lpk_b_skip%=\n\t"
        "buffer_load_dwordx4 %0, %8, %9, %10 offen\n\t"
        "buffer_load_dwordx4 %1, %8, %9, %10 offen offset:1024\n\t"
        "buffer_load_dwordx4 %2, %8, %9, %11 offen\n\t"
        "buffer_load_dwordx4 %3, %8, %9, %11 offen offset:1024\n\t"
        "buffer_load_dwordx4 %4, %8, %9, %12 offen\n\t"
        "buffer_load_dwordx4 %5, %8, %9, %12 offen offset:1024\n\t"
        "buffer_load_dwordx4 %6, %8, %9, %13 offen\n\t"
        "buffer_load_dwordx4 %7, %8, %9, %13 offen offset:1024\n\t"
        ".Lpk_b_skip%=:"
        : "=&v"(b[0][0]), "=&v"(b[0][1]), "=&v"(b[1][0]), "=&v"(b[1][1]), "=&v"(b[2][0]), "=&v"(b[2][1]), "=&v"(b[3][0]),
          "=&v"(b[3][1])
        : "v"(voff), "s"(rs), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(ok)
        : "memory", "scc");
}
// the compiler sees the registers (re)defined HERE, so no consumer can be scheduled above the wait in front of it
__device__ __forceinline__ void pk_landed(bf16x8 (&b)[4][2]) {
    asm volatile("" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[3][0]),
                 "+v"(b[3][1]));
}

// single-load forms of pk_dma / pk_load_b for the interleaved main loop (one load behind every group of four MFMAs)
__device__ __forceinline__ void pk_dma1(pk_i32x4 rs, int voff, int soff, unsigned lds, int ok) {
    asm volatile(
        "s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lpk_dma1_skip%=\n\t"
        "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[vo], %[rs], %[so] offen lds\n\t"
        ".Lpk_dma1_skip%=:"
        :
        : [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff), [lds] "s"(lds), [ok] "s"(ok)
        : "scc");
}
// NP pieces one by one (the prologue of the tile heights whose NP is neither 4 nor 8)
template <int NP>
__device__ __forceinline__ void pk_dma_n(pk_i32x4 rs, int voff, int soff, int jstep, unsigned lds, int ok) {
    if constexpr (NP == 4 || NP == 8) {
        pk_dma<NP>(rs, voff, soff, jstep, lds, ok);
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) pk_dma1(rs, voff, soff + j * jstep, lds + j * 1024, ok);
        asm volatile("" ::: "memory");
    }
}
template <int OFS>
__device__ __forceinline__ void pk_load_b1(bf16x8& b, pk_i32x4 rs, int voff, int soff, int ok) {
    asm volatile(
        "s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lpk_b1_skip%=\n\t"
        "buffer_load_dwordx4 %[b], %[vo], %[rs], %[so] offen offset:%c[ofs]\n\t"
        ".Lpk_b1_skip%=:"
        : [b] "=&v"(b)
        : [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff), [ok] "s"(ok), [ofs] "i"(OFS)
        : "scc");
}

// four fp32 (one lane's bias values) by an asm load the compiler does not count; `ok` == 0 leaves b as it is
__device__ __forceinline__ void pk_load_bias(f32x4& b, pk_i32x4 rs, int voff, int soff, int ok) {
    asm volatile(
        "s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lpk_bias_skip%=\n\t"
        "buffer_load_dwordx4 %[b], %[vo], %[rs], %[so] offen\n\t"
        ".Lpk_bias_skip%=:"
        : [b] "+v"(b)
        : [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff), [ok] "s"(ok)
        : "scc");
}

// interleaved step body: the 16 (NP = 8) / 12 (NP = 4) loads of the NEXT K-tile are issued one by one behind the groups of
// four MFMAs of the current one, so that a wave's own matrix pipe has work queued while a load instruction issues (a 1 KiB
// load holds the wave's instruction stream for ~50-100 cycles; issued as one block in front of the MFMAs they cost a wave
// ~800 cycles per K-tile without a single MFMA -- in-kernel stamps)
// TI = row tiles of 16 per wave (tile height 16 TI: 128, 96 or 64): 2 TI groups of four MFMAs per K-tile carry the 8 + NP loads
// -- one per group from the front while they fit (TI = 8: the schedule measured in round 3), two in the leading groups of the
// short tiles.  Issue order = wait order: the eight B loads of K-tile t + 1 first (needed at the top of the next step), then the
// A pieces of K-tile t + 2, which the next step's counted vmcnt leaves in flight.
template <int NP, int TI = 8>
__device__ __forceinline__ void pk_compute_il(f32x4 (&acc)[TI][4], const unsigned char* as, const bf16x8 (&b)[4][2], int lane,
                                              bf16x8 (&bn)[4][2], pk_i32x4 rsa, int a_voff, int a_soff, int a_jstep, unsigned lds,
                                              pk_i32x4 rsb, int b_voff, int s0, int s1, int s2, int s3, int ok_b, int ok_a) {
    constexpr int G = 2 * TI, NL = 8 + NP;
    static_assert(NL <= 2 * G, "at most two loads per group of four MFMAs");
    // the A fragment of group g + 1 is requested before the MFMAs of group g (the load statements between them are volatile
    // asm: the compiler keeps their order, so it cannot hoist the reads itself)
    bf16x8 fa = p8_frag<false, 128, 64>(as, 0, 0, lane);
    auto issue = [&](int idx) __attribute__((always_inline)) {
        if (idx < 8) {
            const int so = (idx >> 1) == 0 ? s0 : (idx >> 1) == 1 ? s1 : (idx >> 1) == 2 ? s2 : s3;    // column tile idx >> 1, k-step idx & 1
            if (idx & 1) pk_load_b1<1024>(bn[idx >> 1][1], rsb, b_voff, so, ok_b);
            else pk_load_b1<0>(bn[idx >> 1][0], rsb, b_voff, so, ok_b);
        } else {
            pk_dma1(rsa, a_voff, a_soff + (idx - 8) * a_jstep, lds + (idx - 8) * 1024, ok_a);
        }
    };
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int ks = g / TI, i = g % TI;
        bf16x8 fn = fa;
        if (g + 1 < G) fn = p8_frag<false, 128, 64>(as, ((g + 1) % TI) * 16, ((g + 1) / TI) * 32, lane);
        __builtin_amdgcn_sched_barrier(0);          // keeps the read above this group's MFMAs (hipcc sinks it to reuse the register)
        constexpr int DBL = NL > G ? NL - G : 0;    // leading groups that carry two loads
        const int first = g < DBL ? 2 * g : g + DBL;
        if (first < NL) issue(first);
        if (g < DBL) issue(first + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][ks], fa, acc[i][j], 0, 0, 0);
        fa = fn;
    }
}

// fp8 operands on the streaming structure (round 4).  Two fp8 elements are addressed as one bf16 element (the host passes
// K / 2 and lda / 2), so the A ring, the DMA and the B loads are byte for byte those of the bf16 kernel with a K-tile of 128
// fp8 values: a lane's two 16-byte A reads of a row tile (the bf16 kernel's two k-steps) concatenate into the 32-byte operand of
// ONE v_mfma_scale_f32_16x16x128_f8f6f4 (all block scales 2^0), and the packed B copy (ilvlm_gemm_pack_b8) stores a lane's two
// 16-byte halves of a column tile 1 KiB apart, so b[j][0] / b[j][1] are that operand's halves as well -- both operands split
// their k bytes the same way, the products pair up.  Per K-tile and wave: 32 scaled MFMAs (1024 cycles, as the bf16 kernel's
// 64) beside the same 12 loads -- the direct-to-LDS fp8 kernel issues 8 loads per 16 scaled MFMAs, twice as many per MFMA cycle.
// A_E5M2: the A operand holds e5m2 (gradients).  8 groups of four MFMAs; the 12 loads ride two per group in the first four.
template <int NP, bool A_E5M2>
__device__ __forceinline__ void pk_compute_il8(f32x4 (&acc)[8][4], const unsigned char* as, const bf16x8 (&b)[4][2], int lane,
                                               bf16x8 (&bn)[4][2], pk_i32x4 rsa, int a_voff, int a_soff, int a_jstep, unsigned lds,
                                               pk_i32x4 rsb, int b_voff, int s0, int s1, int s2, int s3, int ok_b, int ok_a) {
    typedef int v8i __attribute__((ext_vector_type(8)));
    union F8 { struct { bf16x8 lo, hi; } h; v8i v; };
    constexpr int G = 8, NL = 8 + NP;
    static_assert(NL <= 2 * G, "at most two loads per group");
    auto issue = [&](int idx) __attribute__((always_inline)) {
        if (idx < 8) {
            const int so = (idx >> 1) == 0 ? s0 : (idx >> 1) == 1 ? s1 : (idx >> 1) == 2 ? s2 : s3;
            if (idx & 1) pk_load_b1<1024>(bn[idx >> 1][1], rsb, b_voff, so, ok_b);
            else pk_load_b1<0>(bn[idx >> 1][0], rsb, b_voff, so, ok_b);
        } else {
            pk_dma1(rsa, a_voff, a_soff + (idx - 8) * a_jstep, lds + (idx - 8) * 1024, ok_a);
        }
    };
    F8 fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { fb[j].h.lo = b[j][0]; fb[j].h.hi = b[j][1]; }
    F8 fa;
    fa.h.lo = p8_frag<false, 128, 64>(as, 0, 0, lane);
    fa.h.hi = p8_frag<false, 128, 64>(as, 0, 32, lane);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        F8 fn = fa;
        if (g + 1 < G) {
            fn.h.lo = p8_frag<false, 128, 64>(as, (g + 1) * 16, 0, lane);
            fn.h.hi = p8_frag<false, 128, 64>(as, (g + 1) * 16, 32, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int DBL = NL > G ? NL - G : 0;
        const int first = g < DBL ? 2 * g : g + DBL;
        if (first < NL) issue(first);
        if (g < DBL) issue(first + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)      // cbsz / blgp: 0 = e4m3, 1 = e5m2 (first operand = B = weights, second = A); scale bytes 0x7f = 2^0
            acc[g][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j].v, fa.v, acc[g][j], 0, A_E5M2 ? 1 : 0, 0, 0x7f7f7f7f, 0,
                                                                         0x7f7f7f7f);
        fa = fn;
    }
}

#endif

// SK: compiled with the store-type split-K hand-off (a separate instantiation: its slab reduction raises the register
// allocation from ~200 to 256 per lane, which at two waves per SIMD would leave no room for a co-resident wave of another
// kernel -- and the step runs four streams).
// TI (round 4): row tiles per wave = tile height 16 TI.  At per-GPU batch 256 five of the eight store-type products of a block
// have N = 768 / 512: 300 / 178 tiles of 128 rows for 512 workgroup slots -- ONE round however few the tiles, so the launch takes
// what one tile takes.  96- / 64-row tiles (402 / 354 tiles: still one round) make that tile 25 / 50 % shorter; where several
// rounds are needed anyway the shorter tile pays when it does not add one (launch_pk_auto's cost model).  Same MFMA sequence per
// output element: bit-identical results.
template <int WN, bool SK, int TI = 8, int F8 = 0>       // F8: 0 bf16 operands, 1 e4m3 x e4m3, 2 e5m2 (A) x e4m3 (pk_compute_il8)
__global__ __launch_bounds__(64 * WN, 2) void gemm_bf16_pk_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ Bp,
                                                                  int K, int tiles_m, int tiles_n, int split_k, EpiArgs ep) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!SK || TI == 8, "the slab split-K form is built for 128-row tiles");
    static_assert(F8 == 0 || (TI == 8 && !SK), "fp8 operands: 128-row tiles, no K split");
    constexpr int BMT = 16 * TI;
    constexpr int NT = 64 * WN, NP = BMT * 64 * 2 / (NT * 16);      // DMA pieces per wave and K-tile
    static_assert(NP * NT * 16 == BMT * 64 * 2 && NP >= 1, "the A stage must split into whole 1 KiB pieces per wave");
    constexpr int STAGE = BMT * 64 * 2;
    // The forward / input-gradient chain is what the step waits for; the weight gradients that share its SIMDs (one streaming
    // wave + one weight-gradient wave is the common pairing at 221 + 151 VGPRs) are not.  Wave priority 1: 17.005 -> 16.94 ms,
    // same box, two pairs (priority 3: the same).  -DILVLM_PK_PRIO=0 for the A/B.
    __builtin_amdgcn_s_setprio(ILVLM_PK_PRIO);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the K-slices of a tile are neighbours in the linear order: they run at the same time, on the same XCD
    int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n * split_k);
    const int zs = wg % split_k;
    wg /= split_k;
    const int tile_id = wg;
    int tn = wg % tiles_n, tm = wg / tiles_n;
    if (ep.tile_group > 0) {            // L2 blocking of the tile walk, as in gemm_bf16_dma_kernel
        const int idx = wg, Hb = (tiles_m + ep.tile_bands - 1) / ep.tile_bands, G = ep.tile_group;
        const int band = idx / (Hb * tiles_n), hb = min(Hb, tiles_m - band * Hb);
        const int r = idx - band * Hb * tiles_n, full = tiles_n / G;
        int g, gw, rr;
        if (r < full * hb * G) { g = r / (hb * G); gw = G; rr = r - g * hb * G; }
        else { g = full; gw = tiles_n - full * G; rr = r - full * hb * G; }
        tm = band * Hb + rr / gw;
        tn = g * G + rr % gw;
    }
    const int m0 = tm * BMT, n0 = tn * (64 * WN);
    // this workgroup's K-tiles [t0, t0 + nt): an equal share (the host makes split_k divide into non-empty slices)
    const int nt_all = K >> 6, per = (nt_all + split_k - 1) / split_k;
    const int t0 = zs * per, nt = min(per, nt_all - t0);

    // A: per-lane source offset of piece 0 (rows 8 (wave NP) + (lane >> 3), 16-byte chunk (lane & 7) ^ (row & 7)); pieces
    // step 8 rows, which leaves the swizzle unchanged
    const pk_i32x4 rsa = pk_rsrc(A, ((long)(ep.M - 1) * lda + K) * 2);
    const int arow = wave * NP * 8 + (lane >> 3);
    const int a_voff = (arow * lda + (((lane & 7) ^ (arow & 7)) << 3)) * 2;
    const int a_jstep = 8 * lda * 2;
    int a_soff = m0 * lda * 2 + t0 * 128;                            // + 128 bytes per K-tile
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem_raw + wave * NP * 1024;
    // B: this wave's 4 column tiles of 16; block (n / 16, k / 32) is KiB number (n / 16) * (K / 32) + k / 32
    const pk_i32x4 rsb = pk_rsrc(Bp, (long)ep.N * K * 2);
    const int b_voff = lane * 16;
    const int kb = (K >> 5) << 10;                                   // bytes per column tile
    int bs0 = ((n0 >> 4) + wave * 4) * kb + t0 * 2048, bs1 = bs0 + kb, bs2 = bs1 + kb, bs3 = bs2 + kb;     // + 2 KiB per K-tile

    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    bf16x8 b0[4][2], b1[4][2];

#ifdef ILVLM_GEMM_STAMPS
    unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_comp = 0;
    STAMP(t_start);
#define PK_KEEP() asm volatile("" ::"v"(acc[0][0]), "v"(acc[TI - 1][3]))
#else
#define PK_KEEP()
#endif
    {
        // Prefetch distance 2 for A.  In-kernel stamps of the two-stage form: with the loads interleaved into the MFMA
        // stream a wave waited 1267 cycles per K-step at vmcnt (37 % of the loop) -- the last loads of a step are issued too
        // late to land by the next, and the A operand is a cold HBM stream (each row block is read once per launch).  So A
        // gets a third stage: step t multiplies stage t % 3 while A(t+1) is already in flight or landed and A(t+2) is
        // being issued.  B stays one K-tile ahead in its second register set (weights: L2 / Infinity-Cache resident) and
        // is issued in the FIRST half of the step.
        //   step t:  vmcnt(NP)  [all but the NP youngest = the pieces of A(t+1); vmcnt(0) on the last step]; barrier
        //            groups 0-7: B(t+1) -> other register set      groups 8..: A(t+2) -> stage (t+2) % 3
        //   RAW: a wave's own A(t) pieces were retired by the vmcnt of step t-1 or t, everyone else's by a barrier behind
        //        it.  WAR: stage (t+2) % 3 = (t-1) % 3 was last read in step t-1; its refill is issued behind the barrier
        //        of step t, which every wave passes after those reads returned (lgkmcnt(0) in front of the barrier).
        int st_cur = 0;                                       // stage of K-tile t
#define PK_STEP3(BCUR, BNXT, T)                                                                         \
    do {                                                                                                \
        const int more1 = __builtin_amdgcn_readfirstlane((T) + 1 < nt ? 1 : 0);                         \
        const int more2 = __builtin_amdgcn_readfirstlane((T) + 2 < nt ? 1 : 0);                         \
        STAMP(p0);                                                                                      \
        asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lpk_w0%=\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lpk_w1%=\n\t" \
                     ".Lpk_w0%=:\n\ts_waitcnt vmcnt(0)\n\t.Lpk_w1%=:" ::"s"(more1), "n"(NP) : "memory", "scc");             \
        pk_landed(BCUR);                                                                                \
        STAMP(p1);                                                                                      \
        ILVLM_WG_BARRIER();                                                                             \
        STAMP(p2);                                                                                      \
        a_soff += 128; bs0 += 2048; bs1 += 2048; bs2 += 2048; bs3 += 2048;                             \
        const int st_nn = st_cur == 0 ? 2 : st_cur - 1;       /* (st_cur + 2) % 3 */                    \
        if constexpr (F8 == 0)                                                                          \
            pk_compute_il<NP, TI>(acc, smem_raw + st_cur * STAGE, BCUR, lane, BNXT, rsa, a_voff, a_soff + 128, a_jstep,    \
                                  lds0 + st_nn * STAGE, rsb, b_voff, bs0, bs1, bs2, bs3, more1, more2);  \
        else                                                                                            \
            pk_compute_il8<NP, F8 == 2>(acc, smem_raw + st_cur * STAGE, BCUR, lane, BNXT, rsa, a_voff, a_soff + 128, a_jstep, \
                                        lds0 + st_nn * STAGE, rsb, b_voff, bs0, bs1, bs2, bs3, more1, more2); \
        PK_KEEP();                                                                                      \
        st_cur = st_cur == 2 ? 0 : st_cur + 1;                                                          \
        STAMP(p4);                                                                                      \
        STAMP_ADD(c_wait, p0, p1); STAMP_ADD(c_bar, p1, p2); STAMP_ADD(c_comp, p2, p4);                 \
    } while (0)
        // prologue in wait order: A(0), B(0), then A(1) (which the first step's counted vmcnt leaves in flight)
        pk_dma_n<NP>(rsa, a_voff, a_soff, a_jstep, lds0, 1);
        pk_load_b(b0, rsb, b_voff, bs0, bs1, bs2, bs3, 1);
        pk_dma_n<NP>(rsa, a_voff, a_soff + 128, a_jstep, lds0 + STAGE, __builtin_amdgcn_readfirstlane(nt > 1 ? 1 : 0));
        const int pairs = nt >> 1;
        for (int tp = 0; tp < pairs; ++tp) {
            PK_STEP3(b0, b1, 2 * tp);
            PK_STEP3(b1, b0, 2 * tp + 1);
        }
        if (nt & 1) PK_STEP3(b0, b1, nt - 1);
#undef PK_STEP3
    }
#undef PK_KEEP
#ifdef ILVLM_GEMM_STAMPS
    STAMP(t_loop_end);
#endif

    float alpha = ep.e.alpha;
    if (ep.e.alpha_ptr) alpha *= *ep.e.alpha_ptr;
    if (ep.e.alpha_ptr2) alpha *= *ep.e.alpha_ptr2;
    // every wave is done reading the operand stages before any wave transposes its fragments through the same LDS
    __syncthreads();
    if constexpr (SK) if (split_k > 1) {
        // Store-type split-K (round 3).  The deep-K / narrow-N products of the step (down-projection forward, up- and
        // in-projection input gradients: 600 / 356 tiles of 128x128 for 1024 slots, 24...48 K-tiles each) leave about ONE
        // wave per SIMD, so every wave's issue -> multiply -> wait chain is exposed (36 % MFMA duty in the in-kernel
        // stamps).  Splitting K doubles the waves in flight and halves the chain.  Hand-off without a scheduling
        // dependency and without atomics on C: a workgroup that finishes its K-slice draws a ticket; all but the last
        // arriver PUBLISH their partial tile (fp32 fragments, write-through stores, then a counter); the last arriver
        // waits for the publishers -- all of them are past their K-loop already, whatever the dispatch order --, adds the
        // slabs IN SLICE ORDER (its own registers in their place: the sum does not depend on who arrived when, results are
        // bit-reproducible) and alone runs the epilogue.  Protocol: cdna_hip_programming.md guideline 16 R1 (sc1 stores,
        // every storing wave drains vmcnt, barrier, relaxed agent-scope counter; the consumer polls relaxed and reads with
        // sc1 loads only).  cnt[tile] = tickets, cnt[ntiles + tile] = published slabs; the last arriver zeroes both.
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int SLAB = 128 * 64 * WN * 4;
        const int ntiles = tiles_m * tiles_n;
        __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(ep.e.splitk_ws, 0, ntiles * split_k * SLAB, 0x00020000);
        const int slab0 = tile_id * split_k * SLAB;
        int* cnt = ep.e.splitk_cnt;
        if (tid == 0) *(volatile int*)smem_raw = __hip_atomic_fetch_add(cnt + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int ticket = *(volatile int*)smem_raw;
        __syncthreads();                         // wave 0's transpose slice starts at that word
        if (ticket != split_k - 1) {
            const int mine = slab0 + zs * SLAB + wave * (8 * 4 * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    union { f32x4 f; u32x4 u; } x;
                    x.f = acc[i][j];
                    __builtin_amdgcn_raw_buffer_store_b128(x.u, ws, mine + (i * 4 + j) * 1024 + lane * 16, 0, 16);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt + ntiles + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid == 0) {
            while (__hip_atomic_load(cnt + ntiles + tile_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < split_k - 1)
                __builtin_amdgcn_s_sleep(4);
            __hip_atomic_store(cnt + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);             // next launch
            __hip_atomic_store(cnt + ntiles + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // compiler ordering only: the loads stay below the poll
        if (split_k == 2) {                       // a + b == b + a: one slab to add, no order to keep
            const int sl = slab0 + (1 - zs) * SLAB + wave * (8 * 4 * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    union { f32x4 f; u32x4 u; } x;
                    x.u = __builtin_amdgcn_raw_buffer_load_b128(ws, sl + (i * 4 + j) * 1024 + lane * 16, 0, 16);
                    acc[i][j] += x.f;
                    if (j == 3 && (i & 1)) asm volatile("" ::: "memory");      // at most 8 loads in flight: 32 would spill
                }
        } else {
            // three or more slices: the sum must not depend on which slice arrived last.  The last arriver parks its own
            // fragments in its slab too (each wave re-reads only what it wrote itself) and adds ALL slabs in slice order
            // -- no second register set for the 128 accumulators.
            const int mine = slab0 + zs * SLAB + wave * (8 * 4 * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    union { f32x4 f; u32x4 u; } x;
                    x.f = acc[i][j];
                    __builtin_amdgcn_raw_buffer_store_b128(x.u, ws, mine + (i * 4 + j) * 1024 + lane * 16, 0, 16);
                    acc[i][j] = (f32x4){0, 0, 0, 0};
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int z = 0; z < split_k; ++z) {
                const int sl = slab0 + z * SLAB + wave * (8 * 4 * 1024);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        union { f32x4 f; u32x4 u; } x;
                        x.u = __builtin_amdgcn_raw_buffer_load_b128(ws, sl + (i * 4 + j) * 1024 + lane * 16, 0, 16);
                        acc[i][j] += x.f;
                        if (j == 3 && (i & 1)) asm volatile("" ::: "memory");
                    }
            }
        }
    }
    epilogue_tile<TI, 4>(ep, acc, m0, n0 + wave * 64, lane, alpha, smem_raw + wave * 8192);
#ifdef ILVLM_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(t_end_);
    if (lane == 0 && blockIdx.x < 4096) {
        unsigned long long* o = g_stamps + ((long)blockIdx.x * 8 + wave) * 6;
        o[0] = c_wait; o[1] = c_bar; o[2] = c_issue; o[3] = c_comp; o[4] = t_loop_end - t_start; o[5] = t_end_ - t_loop_end;
    }
#endif
#endif
}

// =====================================================================================
// Persistent form of the streaming kernel (round 4).  What the one-tile-per-workgroup form above loses on the K = 512...768
// products of the step is not its main loop (0.42 of the MFMA peak at 4096^3, 0.38-0.44 on ViT-L/14) but everything around it:
// a workgroup's first operands arrive ~2 us after its launch with nothing to multiply meanwhile, its stores keep the slot
// until they are acknowledged, and the dispatcher refills the slot only then -- a third of a slot's time at 8-12 K-tiles per
// tile (in-kernel stamps).  Here a workgroup walks SEVERAL tiles (grid = 2 per CU, tile i of block b is
// linear position b + i * gridDim.x of the same XCD-aware, L2-blocked tile walk), and the operand stream does not stop at a
// tile boundary: the K-tiles of a workgroup's tiles form ONE sequence v = 0, 1, 2, ... through the three-stage A ring and
// the two B register sets, so the last two steps of a tile issue the first loads of the next one (B(0)', A(0)', A(1)') in
// the load slots that the one-tile form leaves empty, and the epilogue -- LDS transpose, bias / activation / residual,
// stores -- runs while those loads are in flight.
//   vmcnt bookkeeping across the epilogue.  Stores count in vmcnt too, so a counted wait behind the epilogue would either
//   wait for the stores' acknowledgements or depend on loads and stores retiring in ONE order.  Neither: the operands of the
//   next tile's first step (A(v+1), B(v+1)) are waited for IN FRONT of the epilogue -- `vmcnt(NP)`, which leaves only the
//   pieces of A(v+2) in flight; they were issued one to two steps earlier, so this wait is short -- and the first step of
//   the next tile skips its wait.  Its second step waits `vmcnt(NP)` as every step does: whatever the retirement order
//   between loads and stores, at most NP outstanding operations means that every load older than the NP youngest ones
//   (the pieces of A(v+3)) has landed, because loads retire in order among themselves.
//   LDS: EPI_SEP = 1: the epilogue transposes through 8 KiB per wave of its own BEHIND the ring (80 KiB per workgroup, two
//   workgroups fill the CU's 160 KiB); no barrier is needed around it and waves drift into the next tile on their own.
//   EPI_SEP = 2: 64 KiB -- two waves through the free stage, two through 16 KiB behind the ring (32-row passes, one barrier).
//   EPI_SEP = 0: through the stage that the tile's last step has just read (4 KiB per wave at four waves: 16-row passes),
//   48 KiB per workgroup, one barrier in front of the epilogue; the next step's top barrier orders the transposes
//   against the DMA that refills the stage.
// Even K-tile counts only (the B register sets swap roles every step; an odd count would need the loop body twice); every
// shape of the towers has one.  Same MFMA sequence per output element: bit-identical to the one-tile form.
// =====================================================================================
#if defined(__HIP_DEVICE_COMPILE__)
// linear position p of the tile walk -> tile coordinates (as in gemm_bf16_pk_kernel: XCD remap, then row bands x column groups)
template <class EP>
__device__ __forceinline__ void pk_tile_of(int p, int tiles_m, int tiles_n, const EP& ep, int& tm, int& tn) {
    const int wg = xcd_remap(p, tiles_m * tiles_n);
    tn = wg % tiles_n;
    tm = wg / tiles_n;
    if (ep.tile_group > 0) {
        const int idx = wg, Hb = (tiles_m + ep.tile_bands - 1) / ep.tile_bands, G = ep.tile_group;
        const int band = idx / (Hb * tiles_n), hb = min(Hb, tiles_m - band * Hb);
        const int r = idx - band * Hb * tiles_n, full = tiles_n / G;
        int g, gw, rr;
        if (r < full * hb * G) { g = r / (hb * G); gw = G; rr = r - g * hb * G; }
        else { g = full; gw = tiles_n - full * G; rr = r - full * hb * G; }
        tm = band * Hb + rr / gw;
        tn = g * G + rr % gw;
    }
}
#endif

// One by-value struct = the kernarg segment's layout: the epilogue's arguments (some 50 dwords) are RE-READ from the kernarg
// segment per tile (scalar loads through a pointer the compiler cannot see through) instead of living in SGPRs across the
// main loop -- held there they took the kernel to 140 spilled SGPRs (v_writelane / v_readlane in the loop, three more VGPRs,
// and with those VGPR spills).
struct PkpArgs {
    const bf16* A;
    const bf16* Bp;
    int lda, K, tiles_m, tiles_n;
    int stagger;         // shader cycles by which the second half of the grid starts its first tile late (see the kernel)
    int pad_;
    EpiArgs ep;
};

template <int WN, int EPI_SEP>
__global__ __launch_bounds__(64 * WN, 2) void gemm_bf16_pkp_kernel(PkpArgs args) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = 64 * WN, NP = 128 * 64 * 2 / (NT * 16);
    constexpr int STAGE = 128 * 64 * 2;
    constexpr int EPI_RT = (EPI_SEP || WN == 2) ? 2 : 1;            // row tiles per epilogue pass: 8 KiB or 4 KiB per wave
    static_assert(EPI_SEP != 2 || WN == 4, "the 64 KiB form splits four waves over the free stage and 16 KiB behind the ring");
    typedef __attribute__((address_space(4))) const PkpArgs* KernargPtr;
    const KernargPtr kargs = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const bf16* const A = args.A;
    const bf16* const Bp = args.Bp;
    const int lda = args.lda, K = args.K, tiles_m = args.tiles_m, tiles_n = args.tiles_n;
    struct { int M, N, tile_group, tile_bands; } ep = {args.ep.M, args.ep.N, args.ep.tile_group, args.ep.tile_bands};
    __builtin_amdgcn_s_setprio(ILVLM_PK_PRIO);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = tiles_m * tiles_n, stride = gridDim.x;
    const int nt = K >> 6;                                           // even, >= 2 (host check)

    const pk_i32x4 rsa = pk_rsrc(A, ((long)(ep.M - 1) * lda + K) * 2);
    const int arow = wave * NP * 8 + (lane >> 3);
    const int a_voff = (arow * lda + (((lane & 7) ^ (arow & 7)) << 3)) * 2;
    const int a_jstep = 8 * lda * 2;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem_raw + wave * NP * 1024;
    const pk_i32x4 rsb = pk_rsrc(Bp, (long)ep.N * K * 2);
    const int b_voff = lane * 16;
    const int kb = (K >> 5) << 10;                                   // bytes per column tile of the packed B

    // (tile coordinates come out of integer divisions, which run on the vector ALU: without the readfirstlane every one of
    // these wave-uniform values would occupy a VGPR across the main loop)
    int p = blockIdx.x, tm, tn;
    pk_tile_of(p, tiles_m, tiles_n, ep, tm, tn);
    int m0 = __builtin_amdgcn_readfirstlane(tm * 128), n0 = __builtin_amdgcn_readfirstlane(tn * (64 * WN));
    int a_cur = __builtin_amdgcn_readfirstlane(m0 * lda * 2);                        // byte offset of K-tile 0 of the tile's A rows
    int b_cur = __builtin_amdgcn_readfirstlane(((n0 >> 4) + wave * 4) * kb);         // ... of the wave's first B column tile

    f32x4 acc[8][4];
    bf16x8 b0[4][2], b1[4][2];
    float alpha = args.ep.e.alpha;
    if (args.ep.e.alpha_ptr) alpha *= *args.ep.e.alpha_ptr;
    if (args.ep.e.alpha_ptr2) alpha *= *args.ep.e.alpha_ptr2;
    // consumed HERE: hipcc waits for a load in front of its first use and does not see the asm loads -- first used in the
    // epilogue, these two scalars would put an `s_waitcnt vmcnt(0)` there, which drains the next tile's prefetch
    asm volatile("" : "+v"(alpha));
    alpha = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, alpha)));      // wave-uniform: an SGPR
    // the bias of a tile (this lane's four columns after the epilogue's transpose) is loaded by an asm statement at the TOP
    // of the tile, for the same reason; it is older than every load of the tile's main loop, whose counted waits retire it
    const int has_bias = __builtin_amdgcn_readfirstlane(args.ep.e.bias ? 1 : 0);
    const pk_i32x4 rsbias = pk_rsrc(args.ep.e.bias, (long)ep.N * 4);
    f32x4 bias_v[2] = {(f32x4){0, 0, 0, 0}, (f32x4){0, 0, 0, 0}};
    // (the 8-column epilogue takes this kernel past its register budget -- 12 spilled VGPRs, reloaded behind a vmcnt(0) -- so it
    // keeps the 4-column form unless built with -DILVLM_PKP_WIDE=1)
    const int wide_cfg = __builtin_amdgcn_readfirstlane((ILVLM_EPI_WIDE != 0 && ILVLM_PKP_WIDE != 0 && args.ep.vec8_ok && epi_wide_cfg(args.ep.e)) ? 1 : 0);

#ifdef ILVLM_GEMM_STAMPS
    unsigned long long c_wait = 0, c_bar = 0, c_pre = 0, c_comp = 0, c_epi = 0;
    STAMP(t_start);
#define PKP_KEEP() asm volatile("" ::"v"(acc[0][0]), "v"(acc[7][3]))
#else
#define PKP_KEEP()
#endif
    // prologue in wait order: A(0), B(0), A(1)
    pk_dma<NP>(rsa, a_voff, a_cur, a_jstep, lds0, 1);
    pk_load_b(b0, rsb, b_voff, b_cur, b_cur + kb, b_cur + 2 * kb, b_cur + 3 * kb, 1);
    pk_dma<NP>(rsa, a_voff, a_cur + 128, a_jstep, lds0 + STAGE, 1);
    // Stagger.  The two workgroups of a CU start together and do equal work, so they run in lock-step: both multiply at the
    // same time -- the two waves of a SIMD share its matrix pipe, a K-tile takes each ~2000 cycles for 1024 of MFMA -- and then
    // both store at the same time, every CU of the chip with them: an HBM write burst with the matrix pipes idle (in-kernel
    // stamps of the up-projection forward: 24 k cycles of multiply phase and 26 k of epilogue per tile for 12 k of MFMA).
    // The second half of the grid -- by dispatch order the second workgroup of each CU; a speed assumption only -- therefore
    // starts its first tile late by about one main loop: from then on one workgroup of a CU multiplies while the other
    // stores, and chip-wide the stores are spread over the whole launch.  Its prologue loads are in flight meanwhile.
    if (args.stagger > 0 && (int)blockIdx.x * 2 >= (int)gridDim.x) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)args.stagger) __builtin_amdgcn_s_sleep(16);
    }
    int st_cur = 0;                 // ring stage of the K-tile being multiplied; runs on across tiles
    int later = 0;                  // 1 from the second tile on: its first step's operands were waited for in front of the epilogue
    int has_next;
    // step T of the current tile: multiply K-tile T; issue B of the next K-tile of the SEQUENCE and A of the one after it --
    // of this tile while it has them, of the next tile otherwise
#define PKP_STEP(BCUR, BNXT, T)                                                                                      \
    do {                                                                                                             \
        const int tB = (T) + 1, tA = (T) + 2;                                                                        \
        const int okB = __builtin_amdgcn_readfirstlane((tB < nt || has_next) ? 1 : 0);                               \
        const int okA = __builtin_amdgcn_readfirstlane((tA < nt || has_next) ? 1 : 0);                               \
        const int sB = __builtin_amdgcn_readfirstlane(tB < nt ? b_cur + tB * 2048 : b_nxt);                          \
        const int sA = __builtin_amdgcn_readfirstlane(tA < nt ? a_cur + tA * 128 : a_nxt + (tA - nt) * 128);         \
        /* 2: no wait (first step of a later tile); 1: leave the NP pieces of the next A in flight; 0: the last step of all */ \
        const int wcode = __builtin_amdgcn_readfirstlane(((T) == 0 && later) ? 2 : okB);                             \
        STAMP(p0);                                                                                                   \
        asm volatile("s_cmp_eq_u32 %0, 2\n\ts_cbranch_scc1 .Lpkp_w2%=\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lpkp_w0%=\n\t"   \
                     "s_waitcnt vmcnt(%1)\n\ts_branch .Lpkp_w2%=\n\t.Lpkp_w0%=:\n\ts_waitcnt vmcnt(0)\n\t.Lpkp_w2%=:"          \
                     ::"s"(wcode), "n"(NP) : "memory", "scc");                                                       \
        pk_landed(BCUR);                                                                                             \
        STAMP(p1);                                                                                                   \
        ILVLM_WG_BARRIER();                                                                                          \
        STAMP(p2);                                                                                                   \
        const int st_nn = st_cur == 0 ? 2 : st_cur - 1;       /* (st_cur + 2) % 3 */                                 \
        pk_compute_il<NP>(acc, smem_raw + st_cur * STAGE, BCUR, lane, BNXT, rsa, a_voff, sA, a_jstep,                \
                          lds0 + st_nn * STAGE, rsb, b_voff, sB, sB + kb, sB + 2 * kb, sB + 3 * kb, okB, okA);       \
        PKP_KEEP();                                                                                                  \
        st_cur = st_cur == 2 ? 0 : st_cur + 1;                                                                       \
        STAMP(p4);                                                                                                   \
        STAMP_ADD(c_wait, p0, p1); STAMP_ADD(c_bar, p1, p2); STAMP_ADD(c_comp, p2, p4);                              \
    } while (0)
    do {
        const int pn = p + stride;
        has_next = __builtin_amdgcn_readfirstlane(pn < total ? 1 : 0);
        int tmn, tnn;
        pk_tile_of(has_next ? pn : p, tiles_m, tiles_n, ep, tmn, tnn);
        const int m0n = __builtin_amdgcn_readfirstlane(tmn * 128), n0n = __builtin_amdgcn_readfirstlane(tnn * (64 * WN));
        const int a_nxt = __builtin_amdgcn_readfirstlane(m0n * lda * 2);
        const int b_nxt = __builtin_amdgcn_readfirstlane(((n0n >> 4) + wave * 4) * kb);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
        {
            int lane_b = lane;                       // (opaque: the offset is recomputed per tile instead of living in a VGPR)
            asm volatile("" : "+v"(lane_b));
            const int hb = __builtin_amdgcn_readfirstlane(has_bias);
            const int wc = __builtin_amdgcn_readfirstlane(wide_cfg);
            const int bvo = wc ? (wave * 64 + 8 * (lane_b & 7)) * 4 : (wave * 64 + 4 * (lane_b & 15)) * 4;
            pk_load_bias(bias_v[0], rsbias, bvo, __builtin_amdgcn_readfirstlane(n0 * 4), hb);
            if constexpr (ILVLM_PKP_WIDE != 0) pk_load_bias(bias_v[1], rsbias, bvo, __builtin_amdgcn_readfirstlane(n0 * 4 + 16), hb & wc);
        }
        const int pairs = nt >> 1;
        for (int tp = 0; tp < pairs; ++tp) {
            PKP_STEP(b0, b1, 2 * tp);
            PKP_STEP(b1, b0, 2 * tp + 1);
        }
        STAMP(e0);
        // the next tile's first operands, in front of the epilogue's stores (see the header): A(v+1), B(v+1) landed, the
        // pieces of A(v+2) stay in flight.  On the last tile the last step has drained everything.
        asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lpkp_e%=\n\ts_waitcnt vmcnt(%1)\n\t.Lpkp_e%=:" ::"s"(__builtin_amdgcn_readfirstlane(has_next)), "n"(NP) : "memory", "scc");
        pk_landed(b0);
        asm volatile("" : "+v"(bias_v[0]));             // retired by the waits of the main loop (nt >= 2)
        if constexpr (ILVLM_PKP_WIDE != 0) asm volatile("" : "+v"(bias_v[1]));
        unsigned char* wlds;
        if constexpr (EPI_SEP == 1) {
            wlds = smem_raw + 3 * STAGE + wave * 8192;
        } else if constexpr (EPI_SEP == 2) {
            // 64 KiB per workgroup: waves 0, 1 transpose through the stage the last step has read, waves 2, 3 through 16 KiB
            // behind the ring -- 8 KiB each, the 32-row passes of the one-tile kernel
            ILVLM_WG_BARRIER();
            const int st_free = st_cur == 0 ? 2 : st_cur - 1;
            wlds = wave < 2 ? smem_raw + st_free * STAGE + wave * 8192 : smem_raw + 3 * STAGE + (wave - 2) * 8192;
        } else {
            // the stage the last step has read: every wave is done with it behind this barrier
            ILVLM_WG_BARRIER();
            const int st_free = st_cur == 0 ? 2 : st_cur - 1;
            wlds = smem_raw + st_free * STAGE + wave * (STAGE / WN);
        }
        STAMP(e1);
        {
            KernargPtr kq = kargs;
            asm volatile("" : "+s"(kq));             // opaque per tile: the loads below cannot be hoisted out of the tile loop
            const EpiArgs epl = kq->ep;
            // likewise the lane index: everything the epilogue derives from it (LDS transpose addresses, column offsets) would
            // otherwise be hoisted out of the tile loop and held in VGPRs across the main loop (7 spilled VGPRs, reloaded with a
            // vmcnt(0) at the top of every tile)
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            epilogue_tile<8, 4, EPI_RT, false, ILVLM_PKP_WIDE != 0>(epl, acc, m0, n0 + wave * 64, lane_e, alpha, wlds, bias_v);
        }
        STAMP(e2);
        STAMP_ADD(c_pre, e0, e1); STAMP_ADD(c_epi, e1, e2);
        p = pn; m0 = m0n; n0 = n0n; a_cur = a_nxt; b_cur = b_nxt;
        later = 1;
    } while (has_next);
#undef PKP_STEP
#undef PKP_KEEP
#ifdef ILVLM_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(t_end_);
    if (lane == 0 && blockIdx.x < 4096) {
        unsigned long long* o = g_stamps + ((long)blockIdx.x * 8 + wave) * 6;
        o[0] = c_wait; o[1] = c_bar; o[2] = c_pre; o[3] = c_comp; o[4] = t_end_ - t_start - c_epi; o[5] = c_epi;
    }
#endif
#endif
}

// packed copy of one B operand (ilvlm_gemm_pack_b): one thread per 16-byte chunk
__global__ __launch_bounds__(256) void pack_b_kernel(const bf16* __restrict__ B, int ldb, int trans, int N, int K, bf16* __restrict__ out) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= (long)N * K / 8) return;
    const int l = (int)(c & 63);
    const long blk = c >> 6;
    const int kblocks = K >> 5;
    const int n = (int)(blk / kblocks) * 16 + (l & 15), k0 = (int)(blk % kblocks) * 32 + 8 * (l >> 4);
    bf16x8 v;
    if (!trans) v = *(const bf16x8*)(B + (long)n * ldb + k0);
    else
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = B[(long)(k0 + j) * ldb + n];
    *(bf16x8*)(out + c * 8) = v;
}

// packed copies of every GEMM weight of a bf16 arena (ilvlm_pack_weights): one workgroup per 64 x 64 tile of a weight
// W [rows, cols]; fwd = image of Bop[n][k] = W[n][k], bwd = image of Bop[n'][k'] = W[k'][n'] (the input gradient dY W)
__global__ __launch_bounds__(256) void pack_weights_kernel(const bf16* __restrict__ S, bf16* __restrict__ fwd, bf16* __restrict__ bwd,
                                                           const int* __restrict__ table) {
    __shared__ __attribute__((aligned(16))) bf16 tile[64][64 + 8];
    const int* e = table + (long)blockIdx.x * 5;
    const long off = (long)e[0] * 64;
    const int rows = e[1], cols = e[2], r0 = e[3], c0 = e[4];
    const int t = threadIdx.x;
    {
        const int r = t >> 2, cq = (t & 3) * 16;
        const bf16* src = S + off + (long)(r0 + r) * cols + c0 + cq;
        *(bf16x8*)&tile[r][cq] = *(const bf16x8*)src;
        *(bf16x8*)&tile[r][cq + 8] = *(const bf16x8*)(src + 8);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = t + 256 * h, blk = c >> 6, l = c & 63, hi = blk >> 1, lo = blk & 1;
        // forward image: column tile hi (W rows), k-step lo (W columns)
        {
            const int n = 16 * hi + (l & 15), k = 32 * lo + 8 * (l >> 4);
            const long b = (long)((r0 >> 4) + hi) * (cols >> 5) + (c0 >> 5) + lo;
            *(bf16x8*)(fwd + off + b * 512 + l * 8) = *(const bf16x8*)&tile[n][k];
        }
        // backward image: column tile hi (W columns), k-step lo (W rows)
        {
            const int n = 16 * hi + (l & 15), k = 32 * lo + 8 * (l >> 4);
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[k + j][n];
            const long b = (long)((c0 >> 4) + hi) * (rows >> 5) + (r0 >> 5) + lo;
            *(bf16x8*)(bwd + off + b * 512 + l * 8) = v;
        }
    }
}

// =====================================================================================
// fp32 kernel: 64x64x16 tile, 4 waves (2x2), each 32x32 = 2x2 tiles of 16x16x4
// =====================================================================================
constexpr int FM = 64, FN = 64, FK = 16, FLD = 68;

// LDS image is [k][rows] (rows contiguous, padded): fragment lane l reads [kk + (l>>4)][r16 + (l&15)]
template <bool TR>
__device__ __forceinline__ void f32_stage(const float* __restrict__ src, int ld, int r0, int k0, int R, int Kend, float* lds,
                                          int tid, bool vec) {
    if (!TR) {
        // source [R, K]: thread -> row tid>>2, 4 consecutive k
        int row = r0 + (tid >> 2), kc = (tid & 3) * 4, k = k0 + kc;
        f32x4 v = {0, 0, 0, 0};
        if (row < R) {
            const float* p = src + (long)row * ld + k;
            if (vec && k + 4 <= Kend) v = *(const f32x4*)p;
            else
                for (int j = 0; j < 4; ++j)
                    if (k + j < Kend) v[j] = p[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[(kc + j) * FLD + (tid >> 2)] = v[j];
    } else {
        // source [K, R]: thread -> k tid>>4, 4 consecutive rows
        int k = k0 + (tid >> 4), rc = (tid & 15) * 4, row = r0 + rc;
        f32x4 v = {0, 0, 0, 0};
        if (k < Kend) {
            const float* p = src + (long)k * ld + row;
            if (vec && row + 4 <= R) v = *(const f32x4*)p;
            else
                for (int j = 0; j < 4; ++j)
                    if (row + j < R) v[j] = p[j];
        }
        *(f32x4*)(lds + (tid >> 4) * FLD + rc) = v;
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                       int ldb, int K, int tiles_m, int tiles_n, int split_k, EpiArgs ep,
                                                       int vec_a, int vec_b) {
    __shared__ __attribute__((aligned(16))) float As[FK * FLD];
    __shared__ __attribute__((aligned(16))) float Bs[FK * FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int wg = blockIdx.x;
    const int z = wg % split_k; wg /= split_k;
    const int tn = wg % tiles_n, tm = wg / tiles_n;
    const int m0 = tm * FM, n0 = tn * FN;
    const int nt_total = (K + FK - 1) / FK;
    const int per = (nt_total + split_k - 1) / split_k;
    const int t_begin = z * per, t_end = min(nt_total, t_begin + per);
    if (t_begin >= t_end) return;
    const int Kend = min(K, t_end * FK);

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};

    for (int t = t_begin; t < t_end; ++t) {
        f32_stage<TA>(A, lda, m0, t * FK, ep.M, Kend, As, tid, vec_a);
        f32_stage<TB>(B, ldb, n0, t * FK, ep.N, Kend, Bs, tid, vec_b);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < FK; kk += 4) {
            float fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = As[(kk + (lane >> 4)) * FLD + wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = Bs[(kk + (lane >> 4)) * FLD + wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa[i], acc[i][j], 0, 0, 0);   // C^T fragments
        }
        __syncthreads();
    }
    float alpha = ep.e.alpha;
    if (ep.e.alpha_ptr) alpha *= *ep.e.alpha_ptr;
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            epilogue4<float>(ep, m0 + wm * 32 + i * 16 + c, n0 + wn * 32 + j * 16 + 4 * g, acc[i][j], alpha);
}

template <bool TA, bool TB, bool SWAP>
int launch_bf16(const bf16* A, int lda, const bf16* B, int ldb, int K, int tm, int tn, int split_k, const EpiArgs& ep,
                hipStream_t s) {
    auto kern = gemm_bf16_kernel<TA, TB, SWAP>;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [&] { attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES); });
    if (attr_err != hipSuccess) ILVLM_FAIL((int)attr_err, "gemm_bf16: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    hipLaunchKernelGGL(kern, dim3(tm * tn * split_k), dim3(256), GEMM_LDS_BYTES, s, A, lda, B, ldb, K, tm, tn, split_k, ep);
    ILVLM_LAUNCH_CHECK("gemm_bf16");
    return ILVLM_OK;
}

template <bool TA, bool TB, bool SWAP, int DBM, int DBN, int WM, int WN, int NSTAGE, int BKT = 64, int FP8 = 0>
int launch_dma(const bf16* A, int lda, const bf16* B, int ldb, int K, int M, int N, int split_k, const EpiArgs& ep, hipStream_t s) {
    auto kern = gemm_bf16_dma_kernel<TA, TB, SWAP, DBM, DBN, WM, WN, NSTAGE, BKT, FP8>;
    // operand ring; the SWAP epilogue transposes through 8 KiB per wave of the same allocation
    constexpr int ring = NSTAGE * (DBM + DBN) * BKT * 2, epi = WM * WN * 8192;
    constexpr int bytes = ring > epi ? ring : epi;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [&] { attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
    if (attr_err != hipSuccess) ILVLM_FAIL((int)attr_err, "gemm_bf16_dma: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    const int tm = ceil_div(M, DBM), tn = ceil_div(N, DBN);
    hipLaunchKernelGGL(kern, dim3(tm * tn * split_k), dim3(64 * WM * WN), bytes, s, A, lda, B, ldb, K, tm, tn, split_k, ep);
    ILVLM_LAUNCH_CHECK("gemm_bf16_dma");
    return ILVLM_OK;
}

template <int WN, int TI>
int launch_pk_ti(const bf16* A, int lda, const bf16* Bp, int K, int M, int N, int split_k, const EpiArgs& ep, hipStream_t s) {
    constexpr int ring = 3 * (16 * TI) * 64 * 2, epi = WN * 8192;
    constexpr int bytes = ring > epi ? ring : epi;
    const int tm = ceil_div(M, 16 * TI), tn = ceil_div(N, 64 * WN);
    if constexpr (TI == 8) {
        if (split_k > 1) {
            hipLaunchKernelGGL((gemm_bf16_pk_kernel<WN, true, 8>), dim3(tm * tn * split_k), dim3(64 * WN), bytes, s, A, lda, Bp, K, tm, tn,
                               split_k, ep);
            ILVLM_LAUNCH_CHECK("gemm_bf16_pk");
            return ILVLM_OK;
        }
    }
    hipLaunchKernelGGL((gemm_bf16_pk_kernel<WN, false, TI>), dim3(tm * tn), dim3(64 * WN), bytes, s, A, lda, Bp, K, tm, tn, 1, ep);
    ILVLM_LAUNCH_CHECK("gemm_bf16_pk");
    return ILVLM_OK;
}

// fp8 operands (K = number of fp8 values per row; the kernel is told K / 2 "bf16 elements")
template <int WN, int F8>
int launch_pk8(const void* A8, int lda, const void* Bp8, int K, int M, int N, const EpiArgs& ep, hipStream_t s) {
    constexpr int ring = 3 * 128 * 64 * 2, epi = WN * 8192;
    constexpr int bytes = ring > epi ? ring : epi;
    const int tm = ceil_div(M, 128), tn = ceil_div(N, 64 * WN);
    hipLaunchKernelGGL((gemm_bf16_pk_kernel<WN, false, 8, F8>), dim3(tm * tn), dim3(64 * WN), bytes, s, (const bf16*)A8, lda / 2,
                       (const bf16*)Bp8, K / 2, tm, tn, 1, ep);
    ILVLM_LAUNCH_CHECK("gemm_fp8_pk");
    return ILVLM_OK;
}

std::atomic<int> g_concurrent{0};     // ilvlm_gemm_set_concurrent: the caller keeps several GEMM streams in flight
std::atomic<int> g_wgrad_tile{-1};    // -1 = ILVLM_WGRAD_TILE (default 128); 128, 256 (two-stage 256 x 128) or 257 (single-stage 256 x 128)
std::atomic<int> g_pk_ti{-1};         // -1 = ILVLM_PK_TI (default 8); 8, 6, 4 force a tile height, 0 = the cost model (tests, A/B)

// Tile height of a streaming launch (round 4).  A launch takes rounds x (time of one tile): the tiles of a round run side by
// side on the 2 x CUs workgroup slots.  One tile of 16 TI rows costs, in units of a 128-row K-tile step, nt x t(TI) for its
// K-loop -- t(TI) = (750 + 256 TI) / 2798: the MFMA work scales with the height, the eight B loads and the barrier of a step do
// not (in-kernel stamps: 2.8 k cycles per step pair at TI = 8, of which 2 k are MFMA) -- plus E x TI / 8 for the epilogue (its
// stores scale with the rows; E ~ 4 steps for a bf16 output, twice that for two outputs or fp32) plus a fixed 0.7 for the first
// operands.  The height with the lowest rounds x tile cost wins; ties go to the taller tile (fewer B bytes per FLOP).
template <int WN>
int launch_pk(const bf16* A, int lda, const bf16* Bp, int K, int M, int N, int split_k, const EpiArgs& ep, hipStream_t s) {
    // Default: 128 rows everywhere.  The cost model below (ILVLM_PK_TI=0, or ilvlm_gemm_set_tile_rows(0) after a forced height)
    // is right about launches that own the chip -- alone, cold caches, it takes 4-5 % off a block pair's forward and
    // input-gradient launches (N = 768 products with 96 rows: -7...-11 %; the text tower's K = 512 products with 64 rows: -16 %)
    // -- and wrong about the step: there the slots such a launch leaves empty are filled by the other tower and by the
    // weight-gradient streams, and shorter tiles only add B-operand traffic and workgroups: 16.67-16.85 ms with 128 rows,
    // 16.95-17.14 ms with the model's choice, 16.83 with 96 rows everywhere, 17.77 with 64 (same box,
    // profiles/round4/step_ab_tile_rows.txt).  Kept as a tested option for single-stream use.
    static const int ti_env = getenv("ILVLM_PK_TI") ? atoi(getenv("ILVLM_PK_TI")) : 8;
    static int slots = 0;
    if (slots == 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = 2 * cus;
    }
    int ti = g_pk_ti.load(std::memory_order_relaxed);
    if (ti < 0) ti = ti_env;
    if (split_k > 1) ti = 8;
    if (ti != 8 && ti != 6 && ti != 4) {
        const int tn = ceil_div(N, 64 * WN), nt = K / 64;
        const double E = (ep.e.out_dtype == ILVLM_F32 || ep.e.act == ILVLM_ACT_QUICKGELU || ep.e.act == ILVLM_ACT_GELU_ERF) ? 8.0 : 4.0;
        double best = 1e30;
        ti = 8;
        for (int cand = 8; cand >= 4; cand -= 2) {
            if (WN == 2 && cand == 6) continue;                          // (two-wave form: whole pieces per wave need TI even; kept to 8 / 4)
            const long tiles = (long)ceil_div(M, 16 * cand) * tn;
            const double rounds = (double)((tiles + slots - 1) / slots);
            const double cost = rounds * (nt * (750.0 + 256.0 * cand) / 2798.0 + E * cand / 8.0 + 0.7);
            if (cost < best - 1e-9) { best = cost; ti = cand; }
        }
    }
    if (ti == 6 && WN == 4) return launch_pk_ti<WN, 6>(A, lda, Bp, K, M, N, split_k, ep, s);
    if (ti == 4) return launch_pk_ti<WN, 4>(A, lda, Bp, K, M, N, split_k, ep, s);
    return launch_pk_ti<WN, 8>(A, lda, Bp, K, M, N, split_k, ep, s);
}

// persistent streaming kernel: `slots` workgroups (default two per CU) walk the tiles; a grid smaller than the tile count
// must be a multiple of 8 so that a workgroup's tiles stay in its XCD's range of the walk
std::atomic<int> g_pkp_slots{0};      // 0 = two per CU (ilvlm_gemm_set_persistent_slots: tests force the multi-tile path at small sizes)
std::atomic<int> g_pkp_epi_sep{-1};   // -1 = ILVLM_PKP_EPI_SEP / built-in default
std::atomic<int> g_pkp_stagger{-1};   // -1 = ILVLM_PKP_STAGGER / built-in default; cycles per K-tile

template <int WN, int EPI_SEP>
int launch_pkp(const bf16* A, int lda, const bf16* Bp, int K, int M, int N, const EpiArgs& ep, hipStream_t s) {
    constexpr int bytes = 3 * 128 * 64 * 2 + (EPI_SEP == 1 ? WN * 8192 : EPI_SEP == 2 ? 16384 : 0);
    auto kern = gemm_bf16_pkp_kernel<WN, EPI_SEP>;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    static int cus = 0;
    std::call_once(once, [&] {
        attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        int dev = 0;
        if (attr_err == hipSuccess) attr_err = hipGetDevice(&dev);
        if (attr_err == hipSuccess) attr_err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    });
    if (attr_err != hipSuccess) ILVLM_FAIL((int)attr_err, "gemm_bf16_pkp: setup: %s", hipGetErrorString(attr_err));
    const int tm = ceil_div(M, 128), tn = ceil_div(N, 64 * WN), total = tm * tn;
    static const int slots_env = getenv("ILVLM_PKP_SLOTS") ? atoi(getenv("ILVLM_PKP_SLOTS")) : 0;
    int slots = g_pkp_slots.load(std::memory_order_relaxed);
    if (slots <= 0) slots = slots_env > 0 ? slots_env : 2 * cus;
    int grid = total;
    if (total > slots) grid = slots >= 8 ? (slots & ~7) : 8;
    if (grid > total) grid = total;
    PkpArgs args;
    args.A = A; args.Bp = Bp; args.lda = lda; args.K = K; args.tiles_m = tm; args.tiles_n = tn; args.ep = ep;
    args.pad_ = 0;
    // stagger of the second workgroup of each CU: ILVLM_PKP_STAGGER cycles per K-tile of a tile (only where CUs hold two)
    static const int stagger_env = getenv("ILVLM_PKP_STAGGER") ? atoi(getenv("ILVLM_PKP_STAGGER")) : 0;
    const int stag = g_pkp_stagger.load(std::memory_order_relaxed);
    args.stagger = grid > cus ? (stag >= 0 ? stag : stagger_env) * (K / 64) : 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WN), bytes, s, args);
    ILVLM_LAUNCH_CHECK("gemm_bf16_pkp");
    return ILVLM_OK;
}

template <bool TA, bool TB>
int launch_f32(const float* A, int lda, const float* B, int ldb, int K, int tm, int tn, int split_k, const EpiArgs& ep,
               hipStream_t s) {
    // float4 staging only where every row start is 16-byte aligned; odd leading dimensions take scalar loads
    const int vec_a = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0), vec_b = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
    hipLaunchKernelGGL((gemm_f32_kernel<TA, TB>), dim3(tm * tn * split_k), dim3(256), 0, s, A, lda, B, ldb, K, tm, tn,
                       split_k, ep, vec_a, vec_b);
    ILVLM_LAUNCH_CHECK("gemm_f32");
    return ILVLM_OK;
}

inline bool aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

// bf16 kernel selection (a tuning / test hook, ilvlm_gemm_set_variant):
//   15 (default) = the streaming kernel (weights pre-packed in fragment order, gemm_bf16_pk_kernel) wherever the caller
//                  offers a packed copy of B, the direct-to-LDS 128x128 kernel everywhere else;
//   16           = as 15, but the streaming kernel for EVERY eligible shape with a packed B (tests: short K-loops);
//   17           = as 16, plus store-type split-K (2..4 slices) wherever the caller offers a slab workspace (tests; opt-in
//                  for production through ILVLM_PK_SPLITK -- measured slower on the step's shapes);
//   18           = as 16 with the PERSISTENT streaming kernel (gemm_bf16_pkp_kernel) wherever the K-tile count is even;
//   19           = as 15 (K >= 512 on the streaming kernels) without the persistent form (the A/B reference of round 4);
//    5           = always the single-stage direct-to-LDS 128x128 kernel (both operands through LDS; the A/B reference);
//    0           = the register-staged general kernel only.
// (Round 3 re-ran round 2's 256x256 phased 8-wave kernel on the inline-asm DMA, i.e. for the first time with loads that stay in
// flight across its barriers: 1180 TFLOP/s at 4096^3 against 1057 for the streaming kernel and 913 for the 128x128 one -- and
// 7..20 % SLOWER than the streaming kernel on every shape of the ViT-B/32 step (12 K-tiles, 150..600 tiles for 256 CUs) and of
// ViT-L/14 (516 tiles = 2.02 rounds); profiles/round3/gemm_bench_phased_asm_dma.txt, gemm_bench_vitl14_phased.txt.  It is the
// commit before this one in the git history.)
// The tilings and pipelines that rounds 1 and 2 measured and lost with (64x128, 256x128 3-stage, 256x256 phased, stream-K,
// persistent, 2-/3-deep rings, 128-deep K-tiles) live in the git history; DESIGN.md section 6 has their numbers.
std::atomic<int> g_gemm_variant{15};

}  // namespace

extern "C" int ilvlm_gemm(int compute_dtype, int trans_a, int trans_b, int M, int N, int K, const void* A, int lda,
                          const void* B, int ldb, void* C, int ldc, const ilvlm_gemm_epilogue* epi, int split_k,
                          void* stream) {
    ILVLM_REQUIRE(A && B && epi, "gemm: null pointer");
    // C == NULL: the result is kept only as its fp8 copy (out8; plus aux for the activation epilogues) -- fp8 mode, where
    // the consumer GEMMs read the copy and the bf16 tensor would be written for nobody
    ILVLM_REQUIRE(C || (epi->out8 && !epi->accumulate && !epi->pool_out && compute_dtype != ILVLM_F32),
                  "gemm: null output (allowed only together with an fp8 output copy)");
    ILVLM_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
    ILVLM_REQUIRE(compute_dtype == ILVLM_F32 || compute_dtype == ILVLM_BF16 || compute_dtype == ILVLM_FP8 ||
                      compute_dtype == ILVLM_FP8_BF8A, "gemm: bad compute dtype %d", compute_dtype);
    ILVLM_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? N : K) && ldc >= N, "gemm: leading dimension too small");
    ILVLM_REQUIRE(split_k >= 1, "gemm: split_k must be >= 1");
    ILVLM_REQUIRE(split_k == 1 || epi->accumulate, "gemm: split_k > 1 needs accumulate");
    ILVLM_REQUIRE(!(epi->accumulate && (epi->bias || epi->rowbias || epi->residual || epi->act)),
                  "gemm: accumulate excludes the other epilogue terms");
    ILVLM_REQUIRE(!(epi->accumulate && epi->out_dtype != ILVLM_F32), "gemm: accumulate needs fp32 output");
    ILVLM_REQUIRE(!(compute_dtype == ILVLM_F32 && epi->out_dtype != ILVLM_F32), "gemm: fp32 compute writes fp32");
    const bool fp8 = compute_dtype == ILVLM_FP8 || compute_dtype == ILVLM_FP8_BF8A;
    ILVLM_REQUIRE(!(epi->out8 || epi->out8_amax) || ((fp8 || compute_dtype == ILVLM_BF16) && !epi->accumulate && !epi->pool_out &&
                                                     epi->out_group == 0 && (epi->out8_fmt == 0 || epi->out8_fmt == 1)),
                  "gemm: the fp8 output copy needs bf16 / fp8 compute, no accumulate / pool epilogue, compact rows");
    const bool fp8_wgrad = compute_dtype == ILVLM_FP8_BF8A && trans_a && trans_b && epi->accumulate;
    ILVLM_REQUIRE(!fp8 || fp8_wgrad || (!trans_a && !trans_b && !epi->accumulate && !epi->a_rowsum && K % 128 == 0),
                  "gemm fp8: (0,0) layout without accumulate and K %% 128 == 0, or the weight-gradient form (1,1) with accumulate "
                  "and an e5m2 A operand");
    ILVLM_REQUIRE(!fp8 || (!epi->pool_out && lda % 16 == 0 && ldb % 16 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0),
                  "gemm fp8: 16-byte aligned rows, no pool epilogue");
    ILVLM_REQUIRE(!fp8_wgrad || (M % 16 == 0 && N % 16 == 0), "gemm fp8 weight gradient: M and N must be multiples of 16");
    ILVLM_REQUIRE(!(epi->act && !epi->aux), "gemm: activation needs aux");
    ILVLM_REQUIRE(!(epi->rowbias && epi->out_group <= 0), "gemm: rowbias needs out_group");
    ILVLM_REQUIRE(epi->act >= 0 && epi->act <= ILVLM_ACT_GELU_ERF_BWD, "gemm: bad act %d", epi->act);
    ILVLM_REQUIRE(!(epi->a_rowsum && !(epi->accumulate && (compute_dtype == ILVLM_BF16 || fp8_wgrad))),
                  "gemm: a_rowsum needs accumulate and bf16 compute (or the fp8 weight-gradient form)");
    ILVLM_REQUIRE(!epi->pool_out || (compute_dtype == ILVLM_BF16 && !epi->accumulate && !epi->bias && !epi->rowbias &&
                                     !epi->residual && !epi->act && epi->out_group == 0 &&
                                     ((epi->pool_seq && epi->pool_offs) || epi->pool_group > 0)),
                  "gemm: the pool epilogue needs bf16 compute, no other epilogue term, and pool_seq + pool_offs or pool_group");
    hipStream_t s = (hipStream_t)stream;
    EpiArgs ep;
    ep.e = *epi;
    ep.Cf = (float*)C;
    ep.Cb = (bf16*)C;
    ep.ldc = ldc;
    ep.M = M;
    ep.N = N;
    ep.plain_acc = 0;
    static const int tile_group_env = getenv("ILVLM_GEMM_TILE_GROUP") ? atoi(getenv("ILVLM_GEMM_TILE_GROUP")) : 4;
    static const int tile_kmax_env = getenv("ILVLM_GEMM_TILE_KMAX") ? atoi(getenv("ILVLM_GEMM_TILE_KMAX")) : 1024;
    static const int tile_bands_env = getenv("ILVLM_GEMM_TILE_BANDS") ? atoi(getenv("ILVLM_GEMM_TILE_BANDS")) : 8;
    ep.tile_group = (tile_group_env > 0 && K <= tile_kmax_env && ceil_div(N, 128) > tile_group_env) ? tile_group_env : 0;
    ep.tile_bands = tile_bands_env;
    // slab split-K (epilogue fields splitk_*): offered by the caller, taken only by the 128x128 weight-gradient kernels and
    // only when every K-slice is non-empty (each must draw a ticket), the split is small enough for one workgroup to add
    // the slabs up, and the workspace holds tiles x split slabs
    void* const slab_ws = epi->splitk_ws;
    ep.e.splitk_ws = nullptr;
    auto slab_setup = [&](int k_tile, int tile_m = 128) {
        if (!slab_ws || !epi->splitk_cnt || !epi->accumulate || split_k <= 1) return;
        static const int max_split = getenv("ILVLM_SLAB_MAX_SPLIT") ? atoi(getenv("ILVLM_SLAB_MAX_SPLIT")) : 8;
        const int nt_ = ceil_div(K, k_tile);
        int sk = split_k < nt_ ? split_k : nt_;
        while (sk > 1 && (long)(sk - 1) * ceil_div(nt_, sk) >= nt_) --sk;
        const long tiles = (long)ceil_div(M, tile_m) * ceil_div(N, 128), slab = (long)tile_m * 128 * 4;
        if (sk > 1 && sk <= max_split && tiles <= epi->splitk_cnt_len && tiles * sk * slab <= epi->splitk_ws_bytes &&
            tiles * sk * slab < (1L << 31)) {
            split_k = sk;
            ep.e.splitk_ws = slab_ws;
        }
    };
    size_t caln = epi->out_dtype == ILVLM_F32 ? 16 : 8;
    size_t auxaln = compute_dtype == ILVLM_F32 ? 16 : 8;
    ep.vec8_ok = (ldc % 8 == 0) && aligned(C, 16) && (!epi->aux || aligned(epi->aux, 16)) && (!epi->out8 || aligned(epi->out8, 8)) &&
                 (!epi->bias || aligned(epi->bias, 16));
    if (fp8) {      // two fp8 elements are addressed as one bf16 element (gemm_bf16_dma_kernel, FP8)
        ep.vec_ok = (N % 4 == 0) && (ldc % 4 == 0) && aligned(C, caln) && (!epi->bias || aligned(epi->bias, 16)) &&
                    (!epi->rowbias || aligned(epi->rowbias, 16)) && (!epi->residual || aligned(epi->residual, 16)) &&
                    (!epi->aux || aligned(epi->aux, auxaln));
        if (fp8_wgrad) {    // K-strided fp8 operands: byte addressing, K-tiles of 128 reduction rows
            int nt = ceil_div(K, 128);
            if (split_k > nt) split_k = nt;
            slab_setup(128);
            return launch_dma<true, true, false, 128, 128, 2, 2, 1, 64, 3>((const bf16*)A, lda, (const bf16*)B, ldb, K, M, N, split_k, ep, s);
        }
        // streaming form: the caller offers the e4m3 B operand in fragment order (ilvlm_gemm_pack_b8 / the weight quantiser)
        static const int pk8_env = getenv("ILVLM_FP8_PK") ? atoi(getenv("ILVLM_FP8_PK")) : 1;
        const int variant8 = g_gemm_variant.load(std::memory_order_relaxed);
        static const int pk8_min_k = getenv("ILVLM_FP8_PK_MIN_K") ? atoi(getenv("ILVLM_FP8_PK_MIN_K")) : 0;
        if (pk8_env && variant8 >= 15 && K >= pk8_min_k && epi->b_packed && N % 16 == 0 && (long)N * K < (1L << 31) && ((long)(M - 1) * lda + K) < (1L << 31) &&
            aligned(epi->b_packed, 16)) {
            ep.tile_group = 0;
            const bool wide4 = N % 256 == 0;
            if (compute_dtype == ILVLM_FP8)
                return wide4 ? launch_pk8<4, 1>(A, lda, epi->b_packed, K, M, N, ep, s) : launch_pk8<2, 1>(A, lda, epi->b_packed, K, M, N, ep, s);
            return wide4 ? launch_pk8<4, 2>(A, lda, epi->b_packed, K, M, N, ep, s) : launch_pk8<2, 2>(A, lda, epi->b_packed, K, M, N, ep, s);
        }
        if (compute_dtype == ILVLM_FP8)
            return launch_dma<false, false, true, 128, 128, 2, 2, 1, 64, 1>((const bf16*)A, lda / 2, (const bf16*)B, ldb / 2, K / 2, M, N,
                                                                           1, ep, s);
        return launch_dma<false, false, true, 128, 128, 2, 2, 1, 64, 2>((const bf16*)A, lda / 2, (const bf16*)B, ldb / 2, K / 2, M, N, 1,
                                                                       ep, s);
    }
    ep.vec_ok = (N % 4 == 0) && (ldc % 4 == 0) && aligned(C, caln) && (!epi->bias || aligned(epi->bias, 16)) &&
                (!epi->rowbias || aligned(epi->rowbias, 16)) && (!epi->residual || aligned(epi->residual, 16)) &&
                (!epi->aux || aligned(epi->aux, auxaln));
    if (compute_dtype == ILVLM_BF16) {
        ILVLM_REQUIRE(aligned(A, 16) && aligned(B, 16) && lda % 8 == 0 && ldb % 8 == 0,
                      "gemm bf16: operands need 16-byte alignment and lda/ldb %% 8 == 0 (lda=%d ldb=%d)", lda, ldb);
        int tm = ceil_div(M, BM), tn = ceil_div(N, BN);
        int nt = ceil_div(K, BK);
        if (split_k > nt) split_k = nt;
        const bf16* a = (const bf16*)A;
        const bf16* b = (const bf16*)B;
        const bool swap = !epi->accumulate;
        const int variant = g_gemm_variant.load(std::memory_order_relaxed);
        // K-contiguous operands need whole K-tiles (a partial one would run into the next row); when both operands are
        // K-strided (weight gradients: the contraction runs over token rows) the k-rows past K lie beyond the buffer
        // descriptors' extent and read as zeros, so any K works -- the packed text rows need exactly that.
        const bool fast = (variant != 0 || epi->pool_out) && (K % BK == 0 || (trans_a && trans_b)) && (!trans_a || (M % 8 == 0 && M >= 8)) &&
                          (!trans_b || (N % 8 == 0 && N >= 8));
        // streaming kernel: the caller offers B in fragment order (weights); A must be K-contiguous, whole K-tiles
        // Per-shape choice (benchmarks/gemm_bench.py, profiles/round3/gemm_bench.txt): alone and cold-cache the streaming kernel
        // wins +5..11 % on the K >= 768 shapes of ViT-B/32 and loses 2..8 % on three of the K = 512 shapes of the text tower
        // (8 K-tiles: its two-deep prologue weighs more); inside the step the K = 512 shapes are 0.3 % better on it too (fewer
        // kernels competing for LDS), hence the threshold of 512.
        static const int pk_min_k = getenv("ILVLM_PK_MIN_K") ? atoi(getenv("ILVLM_PK_MIN_K")) : 512;
        // (the kernel addresses A with 32-bit byte offsets through a buffer descriptor: the operand must end below 2 GiB)
        if (((variant >= 16 && variant != 19) || ((variant == 15 || variant == 19) && K >= pk_min_k)) && epi->b_packed && swap && !trans_a && K % 64 == 0 && N % 16 == 0 && !epi->pool_out &&
            (long)N * K * 2 < (1L << 31) && ((long)(M - 1) * lda + K) * 2 < (1L << 31) && aligned(epi->b_packed, 16)) {
            static const int pk_wn = getenv("ILVLM_PK_WN") ? atoi(getenv("ILVLM_PK_WN")) : 4;
            const int wn = (pk_wn == 4 && N % 256 == 0) ? 4 : 2;
            const int tn_pk = ceil_div(N, 64 * wn);
            ep.tile_group = (tile_group_env > 0 && K <= tile_kmax_env && tn_pk > tile_group_env) ? tile_group_env : 0;
            // store-type split-K: when the caller offers a slab workspace and the launch would leave the chip short of waves
            // (tiles x waves below ~2 per SIMD) while the K-loop is long enough to cut (>= 6 K-tiles per slice)
            int sk = 1;
            // (measured SLOWER on every shape of the step -- the slab exchange costs ~20 us per tile pair and the extra waves
            // only add contention: these shapes are load-latency bound, not wave-starved -- so it is opt-in:
            // ILVLM_PK_SPLITK=<max slices>; the tests force it.  DESIGN.md section 6.)
            static const int pk_sk_max = getenv("ILVLM_PK_SPLITK") ? atoi(getenv("ILVLM_PK_SPLITK")) : 1;
            static const int pk_sk_target = getenv("ILVLM_PK_SPLITK_WAVES") ? atoi(getenv("ILVLM_PK_SPLITK_WAVES")) : 2048;
            const int sk_cap = variant == 17 ? 4 : pk_sk_max;
            if (slab_ws && epi->splitk_cnt && sk_cap > 1 && !epi->pool_out) {
                const long tiles = (long)ceil_div(M, 128) * tn_pk, nt_ = K / 64;
                int want = (int)((pk_sk_target + tiles * wn - 1) / (tiles * wn));       // slices to reach the wave target
                if (want > sk_cap) want = sk_cap;
                while (want > 1 && nt_ / want < 6) --want;
                while (want > 1 && (long)(want - 1) * ceil_div(nt_, want) >= nt_) --want;           // every slice non-empty
                const long slab = 128L * 64 * wn * 4;
                if (want > 1 && 2 * tiles <= epi->splitk_cnt_len && tiles * want * slab <= epi->splitk_ws_bytes &&
                    tiles * want * slab < (1L << 31)) {
                    sk = want;
                    ep.e.splitk_ws = slab_ws;
                }
            }
            // persistent form: even K-tile counts, no K split
            // Opt-in (ILVLM_PKP=1; selector 18 in the tests): measured round 4, same box -- alone +1..5 % on the launches with more
            // tiles than workgroup slots, but 16.75 -> 16.9 ms in the step with 64 KiB of LDS and 17.2 ms with 80 KiB (fewer
            // co-resident weight-gradient workgroups), and no start stagger helps (profiles/round4/).  The in-kernel stamps say why
            // the one-tile form loses little: a tile's first operands cost ~2 k of its ~55 k cycles.
            static const int pkp_env = getenv("ILVLM_PKP") ? atoi(getenv("ILVLM_PKP")) : 0;
            static const int pkp_sep_env = getenv("ILVLM_PKP_EPI_SEP") ? atoi(getenv("ILVLM_PKP_EPI_SEP")) : 2;
            const bool persistent = (variant == 18 || (variant == 15 && pkp_env)) && sk == 1 && (K / 64) % 2 == 0 && K >= 128;
            if (persistent) {
                const int sep_sel = g_pkp_epi_sep.load(std::memory_order_relaxed);
                const int sep = sep_sel < 0 ? pkp_sep_env : sep_sel;
                const bf16* bp = (const bf16*)epi->b_packed;
                if (wn == 4) {
                    if (sep == 1) return launch_pkp<4, 1>(a, lda, bp, K, M, N, ep, s);
                    if (sep == 2) return launch_pkp<4, 2>(a, lda, bp, K, M, N, ep, s);
                    return launch_pkp<4, 0>(a, lda, bp, K, M, N, ep, s);
                }
                return sep == 1 ? launch_pkp<2, 1>(a, lda, bp, K, M, N, ep, s) : launch_pkp<2, 0>(a, lda, bp, K, M, N, ep, s);
            }
            if (wn == 4) return launch_pk<4>(a, lda, (const bf16*)epi->b_packed, K, M, N, sk, ep, s);
            return launch_pk<2>(a, lda, (const bf16*)epi->b_packed, K, M, N, sk, ep, s);
        }
        if (fast) {
            // both operands through LDS: 128x128x64 tile, 4 waves, one stage, 4 (store) / 3 (accumulate) workgroups per CU
            // weight gradients on 256 x 128 workgroup tiles, 128 x 64 per wave (round 4; ILVLM_WGRAD_TILE=128 / 256 / 257,
            // ilvlm_gemm_set_wgrad_tile): 48 KiB of operands per K-tile for twice the MFMAs of the 128 x 128 tile's 32 KiB -- a quarter
            // fewer LDS-DMA pieces and transposing fragment reads per MFMA; 96 KiB (two stages) or 48 KiB (one) of LDS
            // Default: 128 x 128 two-stage for a caller that runs one GEMM at a time (best alone: 517 against 757 us per block
            // pair), the single-stage 256 x 128 tile when the caller has declared several GEMM streams in flight
            // (ilvlm_gemm_set_concurrent: the engine's towers + weight-gradient companions) -- there the wide tile's fewer
            // vector-memory and LDS instructions per MFMA leave more of the CU to the other streams: step 16.79 -> 16.56 ms
            // (ViT-B/32, three same-box pairs), 95.75 -> 94.63 ms (ViT-L/14); serial towers 21.16 -> 23.60 ms, hence the hint.
            static const int wg_tile_env = getenv("ILVLM_WGRAD_TILE") ? atoi(getenv("ILVLM_WGRAD_TILE")) : -1;
            static const int wg_tile_mul = getenv("ILVLM_WGRAD_TILE_SPLIT_MUL") ? atoi(getenv("ILVLM_WGRAD_TILE_SPLIT_MUL")) : 1;
            const int wg_sel = g_wgrad_tile.load(std::memory_order_relaxed);
            const int wg_tile = wg_sel >= 0 ? wg_sel : (wg_tile_env >= 0 ? wg_tile_env : (g_concurrent.load(std::memory_order_relaxed) ? 257 : 128));
            if (!swap && trans_a && trans_b && variant >= 15 && (wg_tile == 256 || wg_tile == 257) && M % 256 == 0) {
                split_k *= wg_tile_mul;
                if (split_k > nt) split_k = nt;
                slab_setup(64, 256);
                if (wg_tile == 257) return launch_dma<true, true, false, 256, 128, 2, 2, 1>(a, lda, b, ldb, K, M, N, split_k, ep, s);
                return launch_dma<true, true, false, 256, 128, 2, 2, 2>(a, lda, b, ldb, K, M, N, split_k, ep, s);
            }
            if (!swap) slab_setup(64);
            // weight gradients (both operands K-strided, accumulate): two-stage operand ring -- K-tile t+1 is in flight while
            // t is multiplied (64 KiB of LDS, two workgroups per CU).  Round 2 measured this form 3 % SLOWER; that build
            // was not pipelined at all (compiler-inserted vmcnt(0) behind the DMA builtin, see DmaOperand::issue).
            static const int wgrad_stages = getenv("ILVLM_WGRAD_STAGES") ? atoi(getenv("ILVLM_WGRAD_STAGES")) : 2;
            // (Issuing the next K-tile's DMA pieces one by one between the MFMA groups, as the streaming kernel does, LOSES here:
            // 585 against 515 us per block pair -- with two stages the loads need the whole compute phase to land, and a piece
            // issued late is waited for at the top of the next step; profiles/round3/gemm_bench_wgrad_interleaved_issue.txt.
            // Giving the interleaved loads a whole step of slack -- THREE stages of 32-deep K-tiles, 48 KiB, three workgroups
            // per CU -- loses too: 547 us per block pair alone, 17.19 against 16.88 ms in the step (twice the barriers per
            // MFMA); profiles/round3/gemm_bench_wgrad_3stage_bk32.txt.  Both forms are in the git history.)
            if (!swap && trans_a && trans_b && wgrad_stages == 2 && variant >= 15)
                return launch_dma<true, true, false, 128, 128, 2, 2, 2>(a, lda, b, ldb, K, M, N, split_k, ep, s);
#define ILVLM_DMA(TA, TB)                                                                                            \
    return swap ? launch_dma<TA, TB, true, 128, 128, 2, 2, 1>(a, lda, b, ldb, K, M, N, split_k, ep, s)               \
                : launch_dma<TA, TB, false, 128, 128, 2, 2, 1>(a, lda, b, ldb, K, M, N, split_k, ep, s)
            if (!trans_a && !trans_b) { ILVLM_DMA(false, false); }
            if (!trans_a && trans_b) { ILVLM_DMA(false, true); }
            if (trans_a && !trans_b) { ILVLM_DMA(true, false); }
            ILVLM_DMA(true, true);
#undef ILVLM_DMA
        }
        ILVLM_REQUIRE(epi->a_rowsum == nullptr, "gemm: a_rowsum needs the direct-to-LDS path (K %% 64 == 0, M %% 8 == 0)");
        ILVLM_REQUIRE(epi->pool_out == nullptr, "gemm: the pool epilogue needs the direct-to-LDS path (K %% 64 == 0)");
        ILVLM_REQUIRE(!epi->out8 && !epi->out8_amax, "gemm: the fp8 output copy needs the direct-to-LDS path (K %% 64 == 0)");
#define ILVLM_DISPATCH(TA, TB)                                                                       \
    return swap ? launch_bf16<TA, TB, true>(a, lda, b, ldb, K, tm, tn, split_k, ep, s)               \
                : launch_bf16<TA, TB, false>(a, lda, b, ldb, K, tm, tn, split_k, ep, s)
        if (!trans_a && !trans_b) { ILVLM_DISPATCH(false, false); }
        if (!trans_a && trans_b) { ILVLM_DISPATCH(false, true); }
        if (trans_a && !trans_b) { ILVLM_DISPATCH(true, false); }
        ILVLM_DISPATCH(true, true);
#undef ILVLM_DISPATCH
    }
    ILVLM_REQUIRE(aligned(A, 4) && aligned(B, 4) && aligned(C, 4), "gemm f32: operands need 4-byte alignment");
    int tm = ceil_div(M, FM), tn = ceil_div(N, FN);
    int nt = ceil_div(K, FK);
    if (split_k > nt) split_k = nt;
    const float* a = (const float*)A;
    const float* b = (const float*)B;
    if (!trans_a && !trans_b) return launch_f32<false, false>(a, lda, b, ldb, K, tm, tn, split_k, ep, s);
    if (!trans_a && trans_b) return launch_f32<false, true>(a, lda, b, ldb, K, tm, tn, split_k, ep, s);
    if (trans_a && !trans_b) return launch_f32<true, false>(a, lda, b, ldb, K, tm, tn, split_k, ep, s);
    return launch_f32<true, true>(a, lda, b, ldb, K, tm, tn, split_k, ep, s);
}

// regime hint (see the weight-gradient dispatch in ilvlm_gemm)
extern "C" int ilvlm_gemm_set_concurrent(int concurrent) {
    g_concurrent.store(concurrent ? 1 : 0, std::memory_order_relaxed);
    return ILVLM_OK;
}

extern "C" int ilvlm_gemm_get_concurrent(void) { return g_concurrent.load(std::memory_order_relaxed); }

// tuning / test hook: workgroup tile of the bf16 weight-gradient kernel (-1 = default / ILVLM_WGRAD_TILE, 128, 256 = 256 x 128
// two-stage, 257 = 256 x 128 single-stage)
extern "C" int ilvlm_gemm_set_wgrad_tile(int rows) {
    ILVLM_REQUIRE(rows == -1 || rows == 128 || rows == 256 || rows == 257, "gemm_set_wgrad_tile: -1, 128, 256 or 257");
    g_wgrad_tile.store(rows, std::memory_order_relaxed);
    return ILVLM_OK;
}

// tuning hook for the benchmarks/tests: selects the bf16 kernel variant (see g_gemm_variant)
extern "C" int ilvlm_gemm_set_variant(int variant) {
    ILVLM_REQUIRE(variant == 0 || variant == 5 || (variant >= 15 && variant <= 19), "gemm_set_variant: 0, 5 or 15..19");
    g_gemm_variant.store(variant, std::memory_order_relaxed);
    return ILVLM_OK;
}

// tuning / test hook of the persistent streaming kernel: workgroups per launch (0 = two per CU; tests pass 8 so that small
// problems walk several tiles per workgroup), where its epilogue transposes (-1 = default, 1 = own LDS behind the operand
// ring, 0 = through the ring's free stage) and the start delay of the second workgroup of each CU in shader cycles per K-tile
// of a tile (-1 = default / ILVLM_PKP_STAGGER, 0 = none)
extern "C" int ilvlm_gemm_set_persistent(int slots, int epi_sep, int stagger) {
    ILVLM_REQUIRE(slots >= 0 && slots <= 65536 && epi_sep >= -1 && epi_sep <= 2 && stagger >= -1 && stagger <= 100000,
                  "gemm_set_persistent: slots >= 0, epi_sep in -1..2, stagger in -1..100000");
    g_pkp_slots.store(slots, std::memory_order_relaxed);
    g_pkp_epi_sep.store(epi_sep, std::memory_order_relaxed);
    g_pkp_stagger.store(stagger, std::memory_order_relaxed);
    return ILVLM_OK;
}

// tuning / test hook of the streaming kernel: tile height in rows (128, 96, 64; 0 = the per-launch cost model; -1 = back to the
// default, 128 rows unless ILVLM_PK_TI says otherwise)
extern "C" int ilvlm_gemm_set_tile_rows(int rows) {
    ILVLM_REQUIRE(rows == -1 || rows == 0 || rows == 128 || rows == 96 || rows == 64,
                  "gemm_set_tile_rows: -1 (default), 0 (cost model), 128, 96 or 64");
    if (rows < 0) { g_pk_ti.store(-1, std::memory_order_relaxed); return ILVLM_OK; }
    g_pk_ti.store(rows / 16, std::memory_order_relaxed);
    return ILVLM_OK;
}

extern "C" int ilvlm_gemm_pack_b(int trans_b, int N, int K, const void* B, int ldb, void* packed, void* stream) {
    ILVLM_REQUIRE(B && packed && N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "gemm_pack_b: N %% 16 == 0 and K %% 32 == 0 (N=%d K=%d)", N, K);
    ILVLM_REQUIRE(ldb >= (trans_b ? N : K) && ldb % 8 == 0 && aligned(B, 16) && aligned(packed, 16), "gemm_pack_b: alignment / ldb");
    const long chunks = (long)N * K / 8;
    hipLaunchKernelGGL(pack_b_kernel, dim3((int)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)B, ldb,
                       trans_b, N, K, (bf16*)packed);
    ILVLM_LAUNCH_CHECK("gemm_pack_b");
    return ILVLM_OK;
}

// packed copy of an fp8 B operand (ilvlm_gemm_pack_b8): one thread per 16-byte chunk.  Block (n / 16, k / 128) = 16 rows x 128
// bytes = 2 KiB; lane l = (n & 15) + 16 g owns bytes [16 g, 16 g + 16) and [64 + 16 g, ...) of its row: the first halves of
// the 64 lanes form the first KiB of the block, the second halves the second
__global__ __launch_bounds__(256) void pack_b8_kernel(const unsigned char* __restrict__ B, int ldb, int N, int K, unsigned char* __restrict__ out) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= (long)N * K / 16) return;
    const int l = (int)(c & 63), half = (int)((c >> 6) & 1);
    const long blk = c >> 7;
    const int kblocks = K >> 7;
    const int n = (int)(blk / kblocks) * 16 + (l & 15), k0 = (int)(blk % kblocks) * 128 + 64 * half + 16 * (l >> 4);
    *(uint4*)(out + c * 16) = *(const uint4*)(B + (long)n * ldb + k0);
}

extern "C" int ilvlm_gemm_pack_b8(int N, int K, const void* B8, int ldb, void* packed, void* stream) {
    ILVLM_REQUIRE(B8 && packed && N > 0 && K > 0 && N % 16 == 0 && K % 128 == 0, "gemm_pack_b8: N %% 16 == 0 and K %% 128 == 0 (N=%d K=%d)", N, K);
    ILVLM_REQUIRE(ldb >= K && ldb % 16 == 0 && aligned(B8, 16) && aligned(packed, 16), "gemm_pack_b8: alignment / ldb");
    const long chunks = (long)N * K / 16;
    hipLaunchKernelGGL(pack_b8_kernel, dim3((int)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)B8, ldb, N, K,
                       (unsigned char*)packed);
    ILVLM_LAUNCH_CHECK("gemm_pack_b8");
    return ILVLM_OK;
}

extern "C" int ilvlm_pack_weights(const void* arena_bf16, void* fwd, void* bwd, const int32_t* table, int n_tiles, void* stream) {
    ILVLM_REQUIRE(arena_bf16 && fwd && bwd && table && n_tiles > 0, "pack_weights: bad arguments");
    ILVLM_REQUIRE(aligned(arena_bf16, 16) && aligned(fwd, 16) && aligned(bwd, 16), "pack_weights: 16-byte aligned arenas");
    hipLaunchKernelGGL(pack_weights_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, (const bf16*)arena_bf16, (bf16*)fwd,
                       (bf16*)bwd, table);
    ILVLM_LAUNCH_CHECK("pack_weights");
    return ILVLM_OK;
}

extern "C" int ilvlm_wgrad_group(int compute_dtype, const ilvlm_wgrad_problem* problems, int count, long rows, int split_target,
                                 void* stream) {
    ILVLM_REQUIRE(problems && count >= 1 && count <= ILVLM_WGRAD_GROUP_MAX, "wgrad_group: 1..%d problems", ILVLM_WGRAD_GROUP_MAX);
    ILVLM_REQUIRE(compute_dtype == ILVLM_BF16 || compute_dtype == ILVLM_FP8_BF8A, "wgrad_group: bf16 or fp8 (e5m2 dy) operands");
    ILVLM_REQUIRE(rows > 0 && rows < (1L << 31) && split_target != 0, "wgrad_group: bad rows / split_target");
    // split_target < 0: "spread" -- the caller knows nothing else runs beside this launch (the last block of a tower's backward):
    // K-slices for |split_target| slots even on the wide bf16 tile, which otherwise runs one slice per tile
    const bool spread = split_target < 0;
    if (spread) split_target = -split_target;
    const bool f8 = compute_dtype == ILVLM_FP8_BF8A;
    const int ktile = f8 ? 128 : 64, K = (int)rows;
    // fp8: 256 x 128 tiles on the block-scaled MFMA when every output has whole 256-row tiles and more than one tile column
    // (ILVLM_FP8_WGRAD_TILE=128: off).  The wide kernel has no registers left for the bias gradient's row sums, so the launch is
    // split by output COLUMNS: columns [128, k) of every product on the wide tile, columns [0, 128) -- the tile column that
    // carries the row sums -- on the 128 x 128 kernel; two launches per block, each with its own K-slices.
    static const int f8_tile_env = getenv("ILVLM_FP8_WGRAD_TILE") ? atoi(getenv("ILVLM_FP8_WGRAD_TILE")) : 256;
    const int f8_tile_sel = g_wgrad_tile.load(std::memory_order_relaxed);            // ilvlm_gemm_set_wgrad_tile: tests, A/B
    // (measured, same box: fp8 step at per-GPU batch 512 -- 25600 / 22600 token rows -- 23.27 -> 22.74 ms; at batch 256 13.14 ->
    // 13.27 ms: half the K-tiles per tile, so the K-slices that refill the chip weigh more.  Hence a row threshold; an explicit
    // selector overrides it.)
    static const long f8_wide_min_rows = getenv("ILVLM_FP8_WGRAD_WIDE_MIN_ROWS") ? atol(getenv("ILVLM_FP8_WGRAD_WIDE_MIN_ROWS")) : 16384;
    bool wide = f8 && (f8_tile_sel < 0 ? (f8_tile_env == 256 && rows >= f8_wide_min_rows) : f8_tile_sel >= 256);
    for (int i = 0; i < count && wide; ++i) wide = problems[i].n > 0 && problems[i].n % 256 == 0 && problems[i].k > 128 && problems[i].k % 128 == 0;
    for (int i = 0; i < count; ++i) {
        const ilvlm_wgrad_problem& q = problems[i];
        ILVLM_REQUIRE(q.dy && q.x && q.gw && q.n > 0 && q.k > 0, "wgrad_group: problem %d: null pointer / bad shape", i);
        ILVLM_REQUIRE(q.n % (f8 ? 16 : 8) == 0 && q.k % (f8 ? 16 : 8) == 0, "wgrad_group: problem %d: n, k must be multiples of %d",
                      i, f8 ? 16 : 8);
        ILVLM_REQUIRE(aligned(q.dy, 16) && aligned(q.x, 16) && aligned(q.gw, 4), "wgrad_group: problem %d: operand alignment", i);
        ILVLM_REQUIRE(!f8 || (q.inv_g && q.inv_x), "wgrad_group: problem %d: fp8 operands need their scales", i);
        // both operands are addressed through 32-bit byte offsets of a buffer descriptor: they must end below 2 GiB
        ILVLM_REQUIRE(((rows - 1) * (long)q.n + q.n) * (f8 ? 1 : 2) < (1L << 31) && ((rows - 1) * (long)q.k + q.k) * (f8 ? 1 : 2) < (1L << 31),
                      "wgrad_group: problem %d: an operand of %ld rows exceeds 2 GiB", i, rows);
    }
    hipStream_t s = (hipStream_t)stream;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    // ILVLM_WGRAD_GROUP_LDS: LDS bytes requested per workgroup beyond what the operand ring needs -- a way to cap the
    // workgroups per CU of this (long-running) launch so that the input-gradient chain on the other stream keeps its share
    static const int lds_env = getenv("ILVLM_WGRAD_GROUP_LDS") ? atoi(getenv("ILVLM_WGRAD_GROUP_LDS")) : 0;
    std::call_once(once, [&] {
        attr_err = hipFuncSetAttribute((const void*)wgrad_group_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr_err == hipSuccess)
            attr_err = hipFuncSetAttribute((const void*)wgrad_group_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr_err == hipSuccess)
            attr_err = hipFuncSetAttribute((const void*)wgrad_group_kernel<1, 4, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr_err == hipSuccess)
            attr_err = hipFuncSetAttribute((const void*)wgrad_group_kernel<1, 0, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr_err != hipSuccess) ILVLM_FAIL((int)attr_err, "wgrad_group: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    const int nt = ceil_div(K, ktile);
    // one launch over the output columns [c0, c1) of every product (c1 < 0: to the product's last column) on tm x 128 tiles
    auto launch = [&](int c0, int c1, int tm, bool with_rowsum) -> int {
        long tiles = 0;
        for (int i = 0; i < count; ++i) {
            const int kc = (c1 < 0 ? problems[i].k : c1) - c0;
            tiles += (long)ceil_div(problems[i].n, tm) * ceil_div(kc, 128);
        }
        // K-slices: the count that minimises  rounds of workgroups x (K-tiles per slice + epilogue), in K-tile units -- a single
        // writer adds its tile with plain loads and stores (~8 K-tiles' worth), K-slices meet in fp32 atomics (~25: they run at
        // 1.3 TB/s and every workgroup of a round reaches them together).  ViT-B/32 block: 432 tiles -> 1 slice; text block:
        // 192 tiles -> 2; ViT-L/14 block: 768 tiles -> 2 (three full rounds instead of one and a half).  ops.wgrad_group_split.
        long cap = K >= 256 ? K / 256 : 1;
        if (cap > 16) cap = 16;
        if (cap > nt) cap = nt;
        long split = 1;
        double best = 1e30;
        for (long sp = 1; sp <= cap; ++sp) {
            const double rounds = (double)((tiles * sp + split_target - 1) / split_target);
            const double c = rounds * ((double)((nt + sp - 1) / sp) + (sp == 1 ? 8.0 : 25.0));
            if (c < best) { best = c; split = sp; }
        }
        GroupArgs g = {};
        g.count = count;
        int total = 0;
        const int esz = f8 ? 1 : 2;
        for (int i = 0; i < count; ++i) {
            const ilvlm_wgrad_problem& q = problems[i];
            const int kc = (c1 < 0 ? q.k : c1) - c0;
            GroupProblem& P = g.p[i];
            P.A = (const bf16*)q.dy;
            P.B = (const bf16*)((const unsigned char*)q.x + (long)c0 * esz);
            P.lda = q.n;
            P.ldb = q.k;
            P.K = K;
            P.tiles_m = ceil_div(q.n, tm);
            P.tiles_n = ceil_div(kc, 128);
            P.split_k = (int)split;
            P.wg_begin = total;
            total += P.tiles_m * P.tiles_n * (int)split;
            P.ep.e.alpha = 1.0f;
            P.ep.e.out_dtype = ILVLM_F32;
            P.ep.e.accumulate = 1;
            P.ep.e.a_rowsum = with_rowsum ? q.gb : nullptr;
            P.ep.e.alpha_ptr = f8 ? q.inv_g : nullptr;
            P.ep.e.alpha_ptr2 = f8 ? q.inv_x : nullptr;
            P.ep.Cf = q.gw + c0;
            P.ep.Cb = (bf16*)(q.gw + c0);
            P.ep.ldc = q.k;
            P.ep.M = q.n;
            P.ep.N = kc;
            P.ep.vec_ok = (q.k % 4 == 0) && (c0 % 4 == 0) && aligned(q.gw, 16);
            P.ep.vec8_ok = 0;
            P.ep.plain_acc = split == 1;
        }
        g.total = total;
        const int lds8 = lds_env > 32768 ? (lds_env > 160 * 1024 ? 160 * 1024 : lds_env) : 32768;
        const int lds8w = lds_env > 49152 ? (lds_env > 160 * 1024 ? 160 * 1024 : lds_env) : 49152;
        const int lds16 = lds_env > 65536 ? (lds_env > 160 * 1024 ? 160 * 1024 : lds_env) : 65536;
        if (tm == 256 && !f8) hipLaunchKernelGGL((wgrad_group_kernel<1, 0, 256>), dim3(total), dim3(256), lds8w, s, g);
        else if (tm == 256) hipLaunchKernelGGL((wgrad_group_kernel<1, 4, 256>), dim3(total), dim3(256), lds8w, s, g);
        else if (f8) hipLaunchKernelGGL((wgrad_group_kernel<1, 3>), dim3(total), dim3(256), lds8, s, g);
        else hipLaunchKernelGGL((wgrad_group_kernel<2, 0>), dim3(total), dim3(256), lds16, s, g);
        ILVLM_LAUNCH_CHECK("wgrad_group");
        return ILVLM_OK;
    };
    if (wide) {
        int rc = launch(128, -1, 256, false);
        if (rc != ILVLM_OK) return rc;
        return launch(0, 128, 128, true);
    }
    // bf16 under the regime hint (several GEMM streams in flight): the single-stage 256 x 128 tile, as ilvlm_gemm's weight gradients
    bool wide16 = !f8 && (f8_tile_sel >= 0 ? f8_tile_sel == 257 : g_concurrent.load(std::memory_order_relaxed) != 0);
    for (int i = 0; i < count && wide16; ++i) wide16 = problems[i].n % 256 == 0;
    if (wide16) {
        // one K-slice wherever the cost model allows it: every tile then has a single writer (16-byte load-add-store instead of
        // atomics).  Measured in the step, same box: slot targets 32 ... 160 tie (16.1-16.4 ms), 256 +0.1 ms, 512 +0.2 ms
        static const int wide_slots = getenv("ILVLM_WGRAD_GROUP_SLOTS_WIDE") ? atoi(getenv("ILVLM_WGRAD_GROUP_SLOTS_WIDE")) : 128;
        if (!spread && split_target > wide_slots) split_target = wide_slots;
        return launch(0, -1, 256, true);
    }
    return launch(0, -1, 128, true);
}

#ifdef ILVLM_GEMM_STAMPS
extern "C" int ilvlm_debug_read_stamps(unsigned long long* host_out, int n) {
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
    return (int)e;
}
#endif
