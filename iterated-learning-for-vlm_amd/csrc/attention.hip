// Self-attention core for short sequences (head_dim 64): one workgroup per (batch, head).
//   L <= 128 (ViT-B/32: 50, text: 77): one WAVE per 16-row tile.  Forward: the wave owns a 16-query tile, scores are
//   computed transposed (S^T = K Q^T) so that softmax statistics are in-lane + two cross-lane steps and the
//   probabilities are already in MFMA B-operand layout for P V -- they never touch LDS; K and V are staged once per
//   head, Q fragments come straight from global memory.  Backward: pass A (wave = query tile) gives dQ, pass B
//   (wave = key tile) gives dK and dV with register accumulators, so no cross-wave reduction and one barrier in the
//   whole kernel; P and dS are recomputed from the saved log-sum-exp in both passes (the MFMAs are not the bottleneck).
//   41 KB of LDS at L = 77 (was 95 KB): 3 workgroups x 5 waves per CU instead of 1 x 4.
//   128 < L <= 288 (ViT-B/16 197, ViT-L/14 257 tokens): the forward is the same kernel with 8 waves walking the
//   10..18 tiles; the backward keeps Q / dO resident and sweeps the keys in blocks of 32 with dQ in registers (a
//   wave-per-tile backward would hold up to 18 tiles of dS per wave and spills).
// bf16 path: all products on v_mfma_f32_16x16x32_bf16, K-strided operands fetched with ds_read_b64_tr_b16 (no
// transposed copies), fp32 softmax, only the row log-sum-exp is saved for backward.  fp32 path: plain VALU kernel used
// by the fp32 parity mode.
// Replaces the bmm/baddbmm -> softmax -> bmm sequence of F.multi_head_attention_forward reached from reference
// image_encoder/base_transformer.py:45-48 and text_encoder/base_transformer.py:45-48 (causal mask
// text_transformer.py:147-153).  The reference's head-averaged attention weights are discarded by its callers
// (base_transformer.py:59, text_transformer.py:234 unless return_att) and are not produced.
#include <mutex>

#include "common.h"

namespace {

constexpr int HD = 64;
constexpr int LDH = HD + 8;   // [rows][72] bf16 images of Q, K, V, dO (144-byte rows)

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// operand stored [rows][k] (k contiguous): lane l gets [r16 + (l&15)][k32 + 8*(l>>4) + j]
__device__ __forceinline__ bf16x8 frag_k(const bf16* base, int ld, int r16, int k32, int lane) {
    return *(const bf16x8*)(base + (r16 + (lane & 15)) * ld + k32 + 8 * (lane >> 4));
}
// operand stored [k][cols] (cols contiguous): lane l gets [k32 + 8*(l>>4) + j][c16 + (l&15)]
__device__ __forceinline__ bf16x8 frag_tr(const bf16* base, int ld, int k32, int c16, int lane) {
    int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g = lane >> 4;
    const bf16* a = base + (k32 + 8 * g + q) * ld + c16 + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * ld));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

// Same, with a permuted contraction order: the lane's first four values are rows r_lo + 4*(l>>4) + 0..3, the last four
// rows r_hi + 4*(l>>4) + 0..3 of column c16 + (l&15).  That is the order in which a lane holds two 16x16 MFMA result
// tiles (rows 4*(l>>4) + r), so a result in registers can be the other operand without a layout change.
__device__ __forceinline__ bf16x8 frag_tr2(const bf16* base, int ld, int r_lo, int r_hi, int c16, int lane) {
    int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g = lane >> 4;
    const bf16* a = base + (r_lo + 4 * g + q) * ld + c16 + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + (r_hi - r_lo) * ld));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}
__device__ __forceinline__ bf16x8 pack8(bf16x4 a, bf16x4 b) {
    bf16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return v;
}
__device__ __forceinline__ float group4_max(float v) {   // over the four 16-lane groups of a wave (same l & 15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group4_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// fp8 copy of a kernel's bf16 output for the fp8 GEMM that consumes it (fp8 mode: saves the separate quantise pass over
// the tensor).  out8 has the layout of the bf16 output; amax is raised to max|value| (non-null = the feature is on; out8 may
// still be null: observe only).  The copy quantises the bf16-rounded value, exactly what the separate pass would read.
struct AttnQ8 {
    unsigned char* out8;
    const float* scale;
    float* amax;
};
template <int FMT>
__device__ __forceinline__ void q8_emit(unsigned char* out8, long off, bf16x4 v, float qs, float& qm) {
    const float a = (float)v[0], b = (float)v[1], c = (float)v[2], d = (float)v[3];
    qm = fmaxf(qm, fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))));
    if (out8) *(unsigned*)(out8 + off) = fp8_pack4<FMT>(a * qs, b * qs, c * qs, d * qs);
}

// [rows < L][64] slice of the packed rows -> [ROWS][72] LDS image, zero padded; NTH threads
template <int ROWS, int NTH>
__device__ __forceinline__ void stage_rows(bf16* dst, const bf16* src, long row_stride, int L, float scale, int tid) {
    for (int c = tid; c < ROWS * 8; c += NTH) {
        int r = c >> 3, ch = c & 7;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < L) {
            v = *(const bf16x8*)(src + r * row_stride + ch * 8);
            if (scale != 1.0f)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] * scale);
        }
        *(bf16x8*)(dst + r * LDH + ch * 8) = v;
    }
}
// The same staging split in two: every global load of a workgroup is issued before the first one is consumed.  Written as
// one loop (load, convert, store) the compiler waits for each load right after issuing it, and a workgroup paid 6-8
// memory round trips back to back before its first MFMA (17 us lifetime for ~2 us of arithmetic at L = 50).
template <int ROWS, int NTH>
struct RowRegs {
    static constexpr int ITER = (ROWS * 8 + NTH - 1) / NTH;
    bf16x8 v[ITER];
    // range-checked buffer loads: rows >= L lie beyond the descriptor's extent and read as zeros -- no branch around the
    // load for the compiler to sink the consumer into
    __device__ __forceinline__ void load(const bf16* src, long row_stride, int L, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int rsb = (int)row_stride * 2;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, L > 0 ? (L - 1) * rsb + 128 : 0, 0x00020000);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int c = tid + it * NTH, r = c >> 3, ch = c & 7;
            union { u32x4 u; bf16x8 h; } x;
            x.u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, c < ROWS * 8 ? r * rsb + ch * 16 : 0x7ffffff0, 0, 0);
            v[it] = x.h;
        }
#endif
    }
    __device__ __forceinline__ void store(bf16* dst, float scale, int tid) const {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int c = tid + it * NTH, r = c >> 3, ch = c & 7;
            if (c < ROWS * 8) {
                bf16x8 w = v[it];
                if (scale != 1.0f)
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = (bf16)((float)w[j] * scale);
                *(bf16x8*)(dst + r * LDH + ch * 8) = w;
            }
        }
    }
};

// one operand fragment straight from global memory: 8 contiguous head-dim values of row `row` (zero beyond L)
__device__ __forceinline__ bf16x8 frag_global(const bf16* src, long row_stride, int row, int L, int k32, int lane) {
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < L) v = *(const bf16x8*)(src + row * row_stride + k32 + 8 * (lane >> 4));
    return v;
}

// ---------------------------------------------------------------------------------------------
// L <= 128: one wave per 16-row tile (NT = ceil(L / 16) waves per workgroup)
// ---------------------------------------------------------------------------------------------
// CAUSAL is a template parameter: with the mask a run-time flag every key tile sits behind a branch (kt < nkt), and the
// compiler can neither hoist the next tile's LDS fragment reads above it nor interleave the tiles' MFMA chains -- the
// non-causal (vision) launches paid for the text tower's tile skipping
template <int NT, int NW, bool CAUSAL>
__global__ __launch_bounds__(64 * NW) void attn_fwd_wave(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                         float* __restrict__ lse, int Lmax, int H, int,
                                                         const int* __restrict__ seq_offs, AttnQ8 q8) {
    constexpr bool causal = CAUSAL;
    constexpr int NTE = (NT + 1) & ~1, ROWS = NTE * 16, NTH = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const float qs = q8.out8 ? q8.scale[0] : 1.f;
    float qm = 0.f;
    bf16* Ks = (bf16*)smem_raw;          // [ROWS][72]
    bf16* Vs = Ks + ROWS * LDH;          // [ROWS][72]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = HD * H;
    const long rs = 3L * E;
    // dense layout: sequence b = rows [b*Lmax, (b+1)*Lmax); packed layout: rows [seq_offs[b], seq_offs[b+1])
    const long row0 = seq_offs ? seq_offs[b] : (long)b * Lmax;
    const int L = seq_offs ? seq_offs[b + 1] - seq_offs[b] : Lmax;
    const bf16* base = qkv + row0 * rs + h * HD;
    const int g = lane >> 4, c16 = lane & 15;
    // the Q tile as B operand (column = query), pre-scaled by 1/sqrt(64) (exact in bf16); the first tile's fragments are
    // requested before the staging loop
    bf16x8 qb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qb[ks] = frag_global(base, rs, wave * 16 + c16, L, ks * 32, lane);
    {
        RowRegs<ROWS, NTH> rk, rv;
        rk.load(base + E, rs, L, tid);
        rv.load(base + 2 * E, rs, L, tid);
        rk.store(Ks, 1.0f, tid);
        rv.store(Vs, 1.0f, tid);
    }
    __syncthreads();
    // NW == NT: one tile per wave; longer sequences (NW = 8 < NT): the wave walks tiles wave, wave + 8, ...
    for (int tile = wave; tile < NT; tile += NW) {
    if (tile * 16 >= L) break;                         // a shorter (packed) sequence: no rows here (after the only barrier)
    const int q = tile * 16 + c16;
    if (tile != wave) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qb[ks] = frag_global(base, rs, q, L, ks * 32, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) qb[ks][j] = (bf16)((float)qb[ks][j] * 0.125f);

    const int nkt = causal ? min(NT, tile + 1) : NT;   // key tiles this query tile can see (wave-uniform)
    f32x4 s[NT];                                       // s[kt][r] = S[q][key = 16 kt + 4 g + r]
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        if (kt < nkt) {
            f32x4 a = {0, 0, 0, 0};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(Ks, LDH, kt * 16, 0, lane), qb[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(Ks, LDH, kt * 16, 32, lane), qb[1], a, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g + r;
                const bool ok = key < L && (!causal || key <= q);
                a[r] = ok ? a[r] : -INFINITY;
                m = fmaxf(m, a[r]);
            }
            s[kt] = a;
        }
    }
    m = group4_max(m);
    float t = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[kt][r] - m);
                s[kt][r] = e;
                t += e;
            }
        }
    }
    t = group4_sum(t);
    const float inv = 1.0f / t;
    if (g == 0 && q < L) lse[((long)b * H + h) * Lmax + q] = m + __logf(t);
    bf16x4 pk[NTE];
#pragma unroll
    for (int kt = 0; kt < NTE; ++kt) {
        pk[kt] = (bf16x4){0, 0, 0, 0};
        if (kt < NT && kt < nkt) pk[kt] = (bf16x4){(bf16)(s[kt][0] * inv), (bf16)(s[kt][1] * inv), (bf16)(s[kt][2] * inv), (bf16)(s[kt][3] * inv)};
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {        // O^T tile: D[d][q] = sum_key V[key][d] P[q][key]
        f32x4 o = {0, 0, 0, 0};
#pragma unroll
        for (int kp = 0; kp < NTE / 2; ++kp)
            if (2 * kp < nkt)
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr2(Vs, LDH, kp * 32, kp * 32 + 16, dt * 16, lane),
                                                            pack8(pk[2 * kp], pk[2 * kp + 1]), o, 0, 0, 0);
        if (q < L) {
            bf16x4 ov = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
            const long off = (row0 + q) * E + h * HD + dt * 16 + 4 * g;
            *(bf16x4*)(out + off) = ov;
            if (q8.amax) q8_emit<0>(q8.out8, off, ov, qs, qm);
        }
    }
    }
    if (q8.amax) {                 // every wave reaches this point (the tile loop only breaks wave-uniformly)
        qm = wave_max(qm);
        if (lane == 0) fp8_amax_raise(q8.amax, qm);
    }
}

#ifdef ILVLM_ATTN_STAMPS
// diagnostic build only (benchmarks/attn_stamps.py): per-wave cycle stamps of the backward kernel's phases
__device__ unsigned long long g_attn_stamps[4096 * 8 * 6];
__device__ __forceinline__ unsigned long long attn_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define ASTAMP(v) unsigned long long v = attn_stamp()
#else
#define ASTAMP(v)
#endif
// (the backward keeps the mask a run-time flag: specialised, the causal instantiation measured 42.6 instead of 36.8 us on the
// packed text rows and the non-causal one gained nothing inside the step)
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_wave(const bf16* __restrict__ dout, const bf16* __restrict__ qkv,
                                                         const bf16* __restrict__ outp, const float* __restrict__ lse,
                                                         bf16* __restrict__ dqkv, int Lmax, int H, int causal,
                                                         const int* __restrict__ seq_offs, AttnQ8 q8) {
    constexpr int NTE = (NT + 1) & ~1, ROWS = NTE * 16, NTH = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const float qs = q8.out8 ? q8.scale[0] : 1.f;      // e5m2 copy of dqkv (dqkv itself may then be null)
    float qm = 0.f;
    ASTAMP(ts0);
    bf16* Qs = (bf16*)smem_raw;          // [ROWS][72], pre-scaled by 1/8
    bf16* Ks = Qs + ROWS * LDH;
    bf16* dOs = Ks + ROWS * LDH;
    float* delta = (float*)(dOs + ROWS * LDH);   // [ROWS] rowsum(dO * O)
    float* lses = delta + ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = HD * H;
    const long rs = 3L * E;
    const long row0 = seq_offs ? seq_offs[b] : (long)b * Lmax;
    const int L = seq_offs ? seq_offs[b + 1] - seq_offs[b] : Lmax;
    const bf16* base = qkv + row0 * rs + h * HD;
    const bf16* vbase = base + 2 * E;
    // V is only ever an MFMA operand with the head dimension contiguous: its fragments come straight from global memory,
    // requested before the staging loop so their latency hides behind it and the barrier
    // (up to 4 tiles = 32 registers; beyond that the occupancy lost costs more than the latency hidden, so longer sequences
    // fetch the fragments where they are used)
    // (re-measured with the batched staging loads: no prefetch 76 us, prefetch at every length 62 us, 5 waves per SIMD
    // forced 64 us, against 60 us for this form at L = 50)
    constexpr bool PREV = NT <= 4;
    bf16x8 vf[PREV ? NT : 1][2];
    if (PREV) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            vf[PREV ? kt : 0][0] = frag_global(vbase, rs, kt * 16 + (lane & 15), L, 0, lane);
            vf[PREV ? kt : 0][1] = frag_global(vbase, rs, kt * 16 + (lane & 15), L, 32, lane);
        }
    }
    const bf16* dob = dout + row0 * E + h * HD;
    const bf16* ob = outp + row0 * E + h * HD;
    {
        RowRegs<ROWS, NTH> rq, rk, rd, ro;        // all loads in flight before the first is used
        rq.load(base, rs, L, tid);
        rk.load(base + E, rs, L, tid);
        rd.load(dob, E, L, tid);
        ro.load(ob, E, L, tid);
        float lv[(ROWS + NTH - 1) / NTH];
#pragma unroll
        for (int i = 0; i < (ROWS + NTH - 1) / NTH; ++i) {
            const int r = tid + i * NTH;
            lv[i] = r < L ? lse[((long)b * H + h) * Lmax + r] : 0.f;
        }
        rq.store(Qs, 0.125f, tid);
        rk.store(Ks, 1.0f, tid);
        rd.store(dOs, 1.0f, tid);
        constexpr int ITER = RowRegs<ROWS, NTH>::ITER;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {     // uniform trip count: every lane takes part in the shuffles
            const int c = tid + it * NTH;
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) part += (float)rd.v[it][j] * (float)ro.v[it][j];     // zero where the row is padding
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            if (c < ROWS * 8 && (c & 7) == 0) delta[c >> 3] = part;
        }
#pragma unroll
        for (int i = 0; i < (ROWS + NTH - 1) / NTH; ++i) {
            const int r = tid + i * NTH;
            if (r < ROWS) lses[r] = lv[i];
        }
    }
    ASTAMP(ts1);
    __syncthreads();
    ASTAMP(ts2);
    const int g = lane >> 4, c16 = lane & 15;
    for (int tile = wave; tile < NT; tile += NW) {
        if (tile * 16 >= L) break;                     // shorter (packed) sequence: no rows in this tile
        // ---- pass A: dQ of query tile `tile`.  Transposed products: lane (g, c16 = q) holds keys 16 kt + 4 g + r.
        const int q = tile * 16 + c16;
        const bf16x8 qb0 = frag_k(Qs, LDH, tile * 16, 0, lane), qb1 = frag_k(Qs, LDH, tile * 16, 32, lane);
        const bf16x8 db0 = frag_k(dOs, LDH, tile * 16, 0, lane), db1 = frag_k(dOs, LDH, tile * 16, 32, lane);
        const float lq = lses[q], dl = delta[q];
        const int nkt = causal ? min(NT, tile + 1) : NT;
        bf16x4 ds[NTE];
#pragma unroll
        for (int kt = 0; kt < NTE; ++kt) {
            ds[kt] = (bf16x4){0, 0, 0, 0};
            if (kt < NT && kt < nkt) {
                f32x4 st = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(Ks, LDH, kt * 16, 0, lane), qb0, st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(Ks, LDH, kt * 16, 32, lane), qb1, st, 0, 0, 0);
                const int krow = kt * 16 + c16;
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PREV ? vf[PREV ? kt : 0][0] : frag_global(vbase, rs, krow, L, 0, lane), db0, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(PREV ? vf[PREV ? kt : 0][1] : frag_global(vbase, rs, krow, L, 32, lane), db1, dp, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * g + r;
                    const bool ok = q < L && key < L && (!causal || key <= q);
                    const float p = ok ? __expf(st[r] - lq) : 0.f;
                    ds[kt][r] = (bf16)(p * (dp[r] - dl));
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {    // dQ^T tile: D[d][q] = sum_key K[key][d] dS[q][key]
            f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int kp = 0; kp < NTE / 2; ++kp)
                if (2 * kp < nkt)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr2(Ks, LDH, kp * 32, kp * 32 + 16, dt * 16, lane),
                                                                  pack8(ds[2 * kp], ds[2 * kp + 1]), acc, 0, 0, 0);
            if (q < L) {
                bf16x4 ov = {(bf16)(acc[0] * 0.125f), (bf16)(acc[1] * 0.125f), (bf16)(acc[2] * 0.125f), (bf16)(acc[3] * 0.125f)};
                const long off = (row0 + q) * rs + h * HD + dt * 16 + 4 * g;
                if (dqkv) *(bf16x4*)(dqkv + off) = ov;
                if (q8.amax) q8_emit<1>(q8.out8, off, ov, qs, qm);
            }
        }
    }
    ASTAMP(ts3);
    for (int tile = wave; tile < NT; tile += NW) {
        if (tile * 16 >= L) break;
        // ---- pass B: dK, dV of key tile `tile`.  Plain products: lane (g, c16 = key) holds queries 16 qt + 4 g + r.
        const int key = tile * 16 + c16;
        const bf16x8 kb0 = frag_k(Ks, LDH, tile * 16, 0, lane), kb1 = frag_k(Ks, LDH, tile * 16, 32, lane);
        bf16x8 vb0, vb1;                                // this wave's own key tile
        if (PREV) {                                     // wave-uniform select, no register indexing
            vb0 = vf[0][0]; vb1 = vf[0][1];
#pragma unroll
            for (int kt = 1; kt < NT; ++kt)
                if (kt == tile) { vb0 = vf[PREV ? kt : 0][0]; vb1 = vf[PREV ? kt : 0][1]; }
        } else {
            vb0 = frag_global(vbase, rs, key, L, 0, lane); vb1 = frag_global(vbase, rs, key, L, 32, lane);
        }
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0, 0, 0, 0}; dv[dt] = (f32x4){0, 0, 0, 0}; }
        const int qt0 = causal ? tile : 0;           // query tiles below the key tile are fully masked
#pragma unroll
        for (int kp = 0; kp < NTE / 2; ++kp) {
            if (2 * kp + 1 >= qt0) {
                bf16x4 pp[2], pd[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int qt = 2 * kp + hf;
                    pp[hf] = (bf16x4){0, 0, 0, 0};
                    pd[hf] = (bf16x4){0, 0, 0, 0};
                    if (qt < NT && qt >= qt0) {
                        f32x4 sc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
                        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(Qs, LDH, qt * 16, 0, lane), kb0, sc, 0, 0, 0);
                        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(Qs, LDH, qt * 16, 32, lane), kb1, sc, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(dOs, LDH, qt * 16, 0, lane), vb0, dp, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(dOs, LDH, qt * 16, 32, lane), vb1, dp, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int qq = qt * 16 + 4 * g + r;
                            const bool ok = qq < L && key < L && (!causal || key <= qq);
                            const float p = ok ? __expf(sc[r] - lses[qq]) : 0.f;
                            pp[hf][r] = (bf16)p;
                            pd[hf][r] = (bf16)(p * (dp[r] - delta[qq]));
                        }
                    }
                }
                const bf16x8 bp = pack8(pp[0], pp[1]), bd = pack8(pd[0], pd[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {   // D[d][key] += sum_q X[q][d] Y[q][key]
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr2(dOs, LDH, kp * 32, kp * 32 + 16, dt * 16, lane), bp, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr2(Qs, LDH, kp * 32, kp * 32 + 16, dt * 16, lane), bd, dk[dt], 0, 0, 0);
                }
            }
        }
        if (key < L) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x4 kv = {(bf16)dk[dt][0], (bf16)dk[dt][1], (bf16)dk[dt][2], (bf16)dk[dt][3]};
                bf16x4 vv = {(bf16)dv[dt][0], (bf16)dv[dt][1], (bf16)dv[dt][2], (bf16)dv[dt][3]};
                const long off = (row0 + key) * rs + h * HD + dt * 16 + 4 * g;
                if (dqkv) {
                    *(bf16x4*)(dqkv + off + E) = kv;
                    *(bf16x4*)(dqkv + off + 2 * E) = vv;
                }
                if (q8.amax) {
                    q8_emit<1>(q8.out8, off + E, kv, qs, qm);
                    q8_emit<1>(q8.out8, off + 2 * E, vv, qs, qm);
                }
            }
        }
    }
    if (q8.amax) {
        qm = wave_max(qm);
        if (lane == 0) fp8_amax_raise(q8.amax, qm);
    }
#ifdef ILVLM_ATTN_STAMPS
    ASTAMP(ts4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASTAMP(ts5);
    if (lane == 0 && blockIdx.x < 4096 && wave < 8) {
        unsigned long long* o = g_attn_stamps + ((long)blockIdx.x * 8 + wave) * 6;
        o[0] = ts1 - ts0; o[1] = ts2 - ts1; o[2] = ts3 - ts2; o[3] = ts4 - ts3; o[4] = ts5 - ts4; o[5] = ts0;
    }
#endif
}

// load [L][64] slice (column offset coff of the packed qkv / out rows) into an [LP][72] LDS image, zero padded
template <int LP>
__device__ __forceinline__ void load_rows(bf16* dst, const bf16* src, long row_stride, int L, float scale, int tid) {
    for (int c = tid; c < LP * 8; c += 256) {
        int r = c >> 3, ch = c & 7;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < L) {
            v = *(const bf16x8*)(src + r * row_stride + ch * 8);
            if (scale != 1.0f)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] * scale);
        }
        *(bf16x8*)(dst + r * LDH + ch * 8) = v;
    }
}

// Backward for 128 < L <= 288 (ViT-L/14: 257 tokens): the L x L probabilities no longer fit in LDS, so keys are swept in
// blocks of 32.  Q, dO (whole sequence) stay resident; per key block: P and dS for all query rows (each wave its own
// 16-row query tiles), dQ accumulated in registers across blocks, dV / dK of the block finished and stored.
template <int LP>
__global__ __launch_bounds__(256) void attn_bwd_tiled_bf16(const bf16* __restrict__ dout, const bf16* __restrict__ qkv,
                                                           const bf16* __restrict__ outp, const float* __restrict__ lse,
                                                           bf16* __restrict__ dqkv, int L, int H, int causal, AttnQ8 q8) {
    // fp8 mode: e5m2 copy of dqkv emitted with the stores (dqkv itself may then be null), amax raised once per wave
    const float qs = q8.out8 ? q8.scale[0] : 1.f;
    float qm = 0.f;
    constexpr int NQ = LP / 16, KB = 32, LDB = KB + 8, NA = (NQ + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* Qs = (bf16*)smem_raw;            // [LP][72]
    bf16* dOs = Qs + LP * LDH;             // [LP][72]
    bf16* Ks = dOs + LP * LDH;             // [32][72]
    bf16* Vs = Ks + KB * LDH;              // [32][72]
    bf16* Ps = Vs + KB * LDH;              // [LP][40]
    bf16* dSs = Ps + LP * LDB;             // [LP][40]
    float* delta = (float*)(dSs + LP * LDB);
    float* lses = delta + LP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = HD * H;
    const long rs = 3L * E;
    const bf16* base = qkv + (long)b * L * rs + h * HD;
    const bf16* dob = dout + (long)b * L * E + h * HD;
    const bf16* ob = outp + (long)b * L * E + h * HD;
    {
        // One workgroup per CU (138 KB of LDS at L = 257): nothing hides a load but the workgroup itself.  All Q / dO / O rows
        // are requested before the first is consumed (RowRegs: 6-9 x 16 bytes per thread and tensor; the loop form paid one
        // memory round trip per iteration, 27 in a row), the first key block's K / V right behind them.
        RowRegs<LP, 256> rq, rd, ro;
        rq.load(base, rs, L, tid);
        rd.load(dob, E, L, tid);
        ro.load(ob, E, L, tid);
        constexpr int NL = (LP + 255) / 256;
        float lv[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int r = tid + i * 256;
            lv[i] = r < L ? lse[((long)b * H + h) * L + r] : 0.f;
        }
        rq.store(Qs, 0.125f, tid);
        rd.store(dOs, 1.0f, tid);
        constexpr int ITER = RowRegs<LP, 256>::ITER;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int c = tid + it * 256;
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) part += (float)rd.v[it][j] * (float)ro.v[it][j];
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            if (c < LP * 8 && (c & 7) == 0) delta[c >> 3] = part;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int r = tid + i * 256;
            if (r < LP) lses[r] = lv[i];
        }
    }

    const int g = lane >> 4, c16 = lane & 15;
    f32x4 dq[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int d = 0; d < 4; ++d) dq[a][d] = (f32x4){0, 0, 0, 0};

    // K / V rows of a key block: 32 rows x 8 chunks = 256 chunks each, one per thread; block kb + 1 is requested while block
    // kb is computed on and parked in registers until the block's last barrier
    const int kr = tid >> 3, kch = tid & 7;
    bf16x8 kv_n = {0, 0, 0, 0, 0, 0, 0, 0}, vv_n = {0, 0, 0, 0, 0, 0, 0, 0};
    if (kr < L) {
        kv_n = *(const bf16x8*)(base + E + (long)kr * rs + kch * 8);
        vv_n = *(const bf16x8*)(base + 2 * E + (long)kr * rs + kch * 8);
    }
    for (int kb = 0; kb < LP / KB; ++kb) {
        *(bf16x8*)(Ks + kr * LDH + kch * 8) = kv_n;
        *(bf16x8*)(Vs + kr * LDH + kch * 8) = vv_n;
        {
            const int key = (kb + 1) * KB + kr;
            kv_n = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            vv_n = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (kb + 1 < LP / KB && key < L) {
                kv_n = *(const bf16x8*)(base + E + (long)key * rs + kch * 8);
                vv_n = *(const bf16x8*)(base + 2 * E + (long)key * rs + kch * 8);
            }
        }
        __syncthreads();
        // phase 1: P and dS of this key block for the wave's query tiles
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int qt = wave + 4 * a;
            if (qt < NQ) {
                bf16x8 qa0 = frag_k(Qs, LDH, qt * 16, 0, lane), qa1 = frag_k(Qs, LDH, qt * 16, 32, lane);
                bf16x8 da0 = frag_k(dOs, LDH, qt * 16, 0, lane), da1 = frag_k(dOs, LDH, qt * 16, 32, lane);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    f32x4 sc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
                    sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, frag_k(Ks, LDH, kt * 16, 0, lane), sc, 0, 0, 0);
                    sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, frag_k(Ks, LDH, kt * 16, 32, lane), sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da0, frag_k(Vs, LDH, kt * 16, 0, lane), dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da1, frag_k(Vs, LDH, kt * 16, 32, lane), dp, 0, 0, 0);
                    const int key = kb * KB + kt * 16 + c16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = qt * 16 + 4 * g + r;
                        const bool ok = q < L && key < L && (!causal || key <= q);
                        const float p = ok ? __expf(sc[r] - lses[q]) : 0.f;
                        Ps[q * LDB + kt * 16 + c16] = (bf16)p;
                        dSs[q * LDB + kt * 16 + c16] = (bf16)(p * (dp[r] - delta[q]));
                    }
                }
            }
        }
        __syncthreads();
        // phase 1b: dQ[q][d] += sum_{key in block} dS[q][key] K[key][d]   (D[d][q], one 32-deep k-step)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int qt = wave + 4 * a;
            if (qt < NQ) {
                const bf16x8 dsf = frag_k(dSs, LDB, qt * 16, 0, lane);
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    dq[a][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(Ks, LDH, 0, d * 16, lane), dsf, dq[a][d], 0, 0, 0);
            }
        }
        // phase 2: dV / dK of the block: 2 (which) x 2 (key tiles) x 4 (d tiles) output tiles, reduction over all queries
        for (int job = wave; job < 16; job += 4) {
            const int which = job >> 3, kt = (job >> 2) & 1, dt = job & 3;
            const bf16* X = which == 0 ? dOs : Qs;
            const bf16* Y = which == 0 ? Ps : dSs;
            f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < LP / 32; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(X, LDH, ks * 32, dt * 16, lane),
                                                              frag_tr(Y, LDB, ks * 32, kt * 16, lane), acc, 0, 0, 0);
            const int key = kb * KB + kt * 16 + c16;
            if (key < L) {
                bf16x4 ov = {(bf16)acc[0], (bf16)acc[1], (bf16)acc[2], (bf16)acc[3]};
                const long off = ((long)b * L + key) * rs + (which == 0 ? 2 * E : E) + h * HD + dt * 16 + 4 * g;
                if (dqkv) *(bf16x4*)(dqkv + off) = ov;
                if (q8.amax) q8_emit<1>(q8.out8, off, ov, qs, qm);
            }
        }
        __syncthreads();   // K/V/P/dS of this block are dead: the next block may overwrite them
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int qt = wave + 4 * a, q = qt * 16 + c16;
        if (qt < NQ && q < L) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 ov = {(bf16)(dq[a][d][0] * 0.125f), (bf16)(dq[a][d][1] * 0.125f), (bf16)(dq[a][d][2] * 0.125f),
                             (bf16)(dq[a][d][3] * 0.125f)};
                const long off = ((long)b * L + q) * rs + h * HD + d * 16 + 4 * g;
                if (dqkv) *(bf16x4*)(dqkv + off) = ov;
                if (q8.amax) q8_emit<1>(q8.out8, off, ov, qs, qm);
            }
        }
    }
    if (q8.amax) {                 // every wave reaches this point
        qm = wave_max(qm);
        if (lane == 0) fp8_amax_raise(q8.amax, qm);
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 VALU kernels (parity mode; LDS-resident for L <= 96 forward / 80 backward, global-memory forms beyond)
// ---------------------------------------------------------------------------------------------
constexpr int LDF = HD + 1;

__device__ __forceinline__ void load_rows_f32(float* dst, const float* src, long row_stride, int L, float scale, int tid) {
    for (int i = tid; i < L * HD; i += 256) {
        int r = i >> 6, d = i & 63;
        dst[r * LDF + d] = src[r * row_stride + d] * scale;
    }
}

__global__ __launch_bounds__(256) void attn_fwd_f32(const float* __restrict__ qkv, float* __restrict__ out,
                                                    float* __restrict__ lse, int Lmax, int H, int causal,
                                                    const int* __restrict__ seq_offs) {
    const long row0 = seq_offs ? seq_offs[blockIdx.x / H] : (long)(blockIdx.x / H) * Lmax;
    const int L = seq_offs ? seq_offs[blockIdx.x / H + 1] - seq_offs[blockIdx.x / H] : Lmax;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* Qs = (float*)smem_raw;
    float* Ks = Qs + L * LDF;
    float* Vs = Ks + L * LDF;
    float* S = Vs + L * LDF;   // [L][L+1]
    const int LS = L + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = HD * H;
    const long rs = 3L * E;
    const float* base = qkv + row0 * rs + h * HD;
    load_rows_f32(Qs, base, rs, L, 0.125f, tid);
    load_rows_f32(Ks, base + E, rs, L, 1.0f, tid);
    load_rows_f32(Vs, base + 2 * E, rs, L, 1.0f, tid);
    __syncthreads();
    for (int i = tid; i < L * L; i += 256) {
        int q = i / L, k = i % L;
        float s = 0.f;
#pragma unroll 8
        for (int d = 0; d < HD; ++d) s = fmaf(Qs[q * LDF + d], Ks[k * LDF + d], s);
        S[q * LS + k] = (causal && k > q) ? -INFINITY : s;
    }
    __syncthreads();
    for (int q = wave; q < L; q += 4) {
        float m = -INFINITY;
        for (int k = lane; k < L; k += 64) m = fmaxf(m, S[q * LS + k]);
        m = wave_max(m);
        float t = 0.f;
        for (int k = lane; k < L; k += 64) t += __expf(S[q * LS + k] - m);
        t = wave_sum(t);
        for (int k = lane; k < L; k += 64) S[q * LS + k] = __expf(S[q * LS + k] - m) / t;
        if (lane == 0) lse[((long)b * H + h) * Lmax + q] = m + __logf(t);
    }
    __syncthreads();
    for (int i = tid; i < L * HD; i += 256) {
        int q = i >> 6, d = i & 63;
        float o = 0.f;
        for (int k = 0; k < L; ++k) o = fmaf(S[q * LS + k], Vs[k * LDF + d], o);
        out[(row0 + q) * E + h * HD + d] = o;
    }
}

__global__ __launch_bounds__(256) void attn_bwd_f32(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                    const float* __restrict__ outp, const float* __restrict__ lse,
                                                    float* __restrict__ dqkv, int Lmax, int H, int causal,
                                                    const int* __restrict__ seq_offs) {
    const long row0 = seq_offs ? seq_offs[blockIdx.x / H] : (long)(blockIdx.x / H) * Lmax;
    const int L = seq_offs ? seq_offs[blockIdx.x / H + 1] - seq_offs[blockIdx.x / H] : Lmax;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* Qs = (float*)smem_raw;
    float* Ks = Qs + L * LDF;
    float* Vs = Ks + L * LDF;
    float* dOs = Vs + L * LDF;
    float* P = dOs + L * LDF;      // [L][L+1]
    const int LS = L + 1;
    float* dS = P + L * LS;        // [L][L+1]
    float* delta = dS + L * LS;    // [L]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = HD * H;
    const long rs = 3L * E;
    const float* base = qkv + row0 * rs + h * HD;
    load_rows_f32(Qs, base, rs, L, 0.125f, tid);
    load_rows_f32(Ks, base + E, rs, L, 1.0f, tid);
    load_rows_f32(Vs, base + 2 * E, rs, L, 1.0f, tid);
    const float* dob = dout + row0 * E + h * HD;
    const float* ob = outp + row0 * E + h * HD;
    load_rows_f32(dOs, dob, E, L, 1.0f, tid);
    for (int q = wave; q < L; q += 4) {
        float t = dob[(long)q * E + lane] * ob[(long)q * E + lane];
        t = wave_sum(t);
        if (lane == 0) delta[q] = t;
    }
    __syncthreads();
    for (int i = tid; i < L * L; i += 256) {
        int q = i / L, k = i % L;
        float s = 0.f, dp = 0.f;
#pragma unroll 8
        for (int d = 0; d < HD; ++d) {
            s = fmaf(Qs[q * LDF + d], Ks[k * LDF + d], s);
            dp = fmaf(dOs[q * LDF + d], Vs[k * LDF + d], dp);
        }
        float p = (causal && k > q) ? 0.f : __expf(s - lse[((long)b * H + h) * Lmax + q]);
        P[q * LS + k] = p;
        dS[q * LS + k] = p * (dp - delta[q]);
    }
    __syncthreads();
    for (int i = tid; i < L * HD; i += 256) {
        int r = i >> 6, d = i & 63;
        float dv = 0.f, dk = 0.f, dq = 0.f;
        for (int j = 0; j < L; ++j) {
            dv = fmaf(P[j * LS + r], dOs[j * LDF + d], dv);
            dk = fmaf(dS[j * LS + r], Qs[j * LDF + d], dk);
            dq = fmaf(dS[r * LS + j], Ks[j * LDF + d], dq);
        }
        float* o = dqkv + (row0 + r) * rs + h * HD + d;
        o[0] = dq * 0.125f;
        o[E] = dk;
        o[2 * E] = dv;
    }
}


// fp32 kernels for sequences that do not fit the LDS-resident forms above (ViT-B/16: 197, ViT-L/14: 257 tokens; parity mode
// only, so clarity over speed): one wave per query row (forward, dQ) or per key row (dK, dV); K / V / Q rows come straight
// from global memory (L2-resident at these sizes), probabilities are recomputed from the saved log-sum-exp.
constexpr int LMAX_F32 = 320;

__global__ __launch_bounds__(256) void attn_fwd_f32_long(const float* __restrict__ qkv, float* __restrict__ out,
                                                         float* __restrict__ lse, int L, int H, int causal) {
    __shared__ float qs[4][HD];
    __shared__ float ps[4][LMAX_F32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H, q = blockIdx.y * 4 + wave;
    if (q >= L) return;                                   // whole wave
    const int E = HD * H;
    const long rs = 3L * E;
    const float* base = qkv + (long)b * L * rs + h * HD;
    qs[wave][lane] = base[(long)q * rs + lane] * 0.125f;
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY;
    for (int k = lane; k < L; k += 64) {
        float sc = 0.f;
        const float* kr = base + E + (long)k * rs;
        for (int d = 0; d < HD; ++d) sc = fmaf(qs[wave][d], kr[d], sc);
        if (causal && k > q) sc = -INFINITY;
        ps[wave][k] = sc;
        m = fmaxf(m, sc);
    }
    m = wave_max(m);
    float t = 0.f;
    for (int k = lane; k < L; k += 64) t += __expf(ps[wave][k] - m);
    t = wave_sum(t);
    for (int k = lane; k < L; k += 64) ps[wave][k] = __expf(ps[wave][k] - m) / t;
    if (lane == 0) lse[((long)b * H + h) * L + q] = m + __logf(t);
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
    for (int k = 0; k < L; ++k) o = fmaf(ps[wave][k], base[2 * E + (long)k * rs + lane], o);
    out[((long)b * L + q) * E + h * HD + lane] = o;
}

// role 0 (blockIdx.z == 0): wave per query row -> dQ; role 1: wave per key row -> dK, dV
__global__ __launch_bounds__(256) void attn_bwd_f32_long(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                         const float* __restrict__ outp, const float* __restrict__ lse,
                                                         float* __restrict__ dqkv, int L, int H, int causal) {
    __shared__ float r0[4][HD], r1[4][HD];
    __shared__ float w0[4][LMAX_F32], w1[4][LMAX_F32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H, row = blockIdx.y * 4 + wave;
    if (row >= L) return;
    const int E = HD * H;
    const long rs = 3L * E;
    const float* base = qkv + (long)b * L * rs + h * HD;
    const float* dob = dout + (long)b * L * E + h * HD;
    const float* ob = outp + (long)b * L * E + h * HD;
    const float* ls = lse + ((long)b * H + h) * L;
    float* o = dqkv + ((long)b * L + row) * rs + h * HD + lane;
    if (blockIdx.z == 0) {
        const int q = row;
        r0[wave][lane] = base[(long)q * rs + lane] * 0.125f;            // scaled q
        r1[wave][lane] = dob[(long)q * E + lane];                       // dO row
        const float delta = wave_sum(dob[(long)q * E + lane] * ob[(long)q * E + lane]);
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < L; k += 64) {
            float sc = 0.f, dp = 0.f;
            const float *kr = base + E + (long)k * rs, *vr = base + 2 * E + (long)k * rs;
            for (int d = 0; d < HD; ++d) { sc = fmaf(r0[wave][d], kr[d], sc); dp = fmaf(r1[wave][d], vr[d], dp); }
            const float p = (causal && k > q) ? 0.f : __expf(sc - ls[q]);
            w0[wave][k] = p * (dp - delta);                             // dS[q][k]
        }
        __builtin_amdgcn_wave_barrier();
        float dq = 0.f;
        for (int k = 0; k < L; ++k) dq = fmaf(w0[wave][k], base[E + (long)k * rs + lane], dq);
        o[0] = dq * 0.125f;
    } else {
        const int k = row;
        r0[wave][lane] = base[E + (long)k * rs + lane];                 // K row
        r1[wave][lane] = base[2 * E + (long)k * rs + lane];             // V row
        __builtin_amdgcn_wave_barrier();
        for (int q = lane; q < L; q += 64) {
            float sc = 0.f, dp = 0.f, delta = 0.f;
            const float *qr = base + (long)q * rs, *dr = dob + (long)q * E, *orow = ob + (long)q * E;
            for (int d = 0; d < HD; ++d) {
                sc = fmaf(qr[d] * 0.125f, r0[wave][d], sc);
                dp = fmaf(dr[d], r1[wave][d], dp);
                delta = fmaf(dr[d], orow[d], delta);
            }
            const float p = (causal && k > q) ? 0.f : __expf(sc - ls[q]);
            w0[wave][q] = p;
            w1[wave][q] = p * (dp - delta);
        }
        __builtin_amdgcn_wave_barrier();
        float dv = 0.f, dk = 0.f;
        for (int q = 0; q < L; ++q) {
            dv = fmaf(w0[wave][q], dob[(long)q * E + lane], dv);
            dk = fmaf(w1[wave][q], base[(long)q * rs + lane] * 0.125f, dk);
        }
        o[E] = dk;
        o[2 * E] = dv;
    }
}

template <class K>
int set_lds(K kern, int bytes, const char* name) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) ILVLM_FAIL((int)e, "%s: hipFuncSetAttribute(%d): %s", name, bytes, hipGetErrorString(e));
    return ILVLM_OK;
}

template <int NT, int NW = NT>
int launch_fwd_wave(const bf16* qkv, bf16* out, float* lse, int B, int L, int H, int causal, const int* seq_offs,
                    hipStream_t s, AttnQ8 q8) {
    constexpr int ROWS = ((NT + 1) & ~1) * 16;
    constexpr int bytes = 2 * ROWS * LDH * 2;
    if (bytes > 64 * 1024) {
        // once per instantiation, thread-safe: forward runs on the main thread, backward on autograd's worker
        static std::once_flag once;
        static int attr_rc = ILVLM_OK;
        std::call_once(once, [&] {
            attr_rc = set_lds(attn_fwd_wave<NT, NW, true>, bytes, "attention_fwd");
            if (!attr_rc) attr_rc = set_lds(attn_fwd_wave<NT, NW, false>, bytes, "attention_fwd");
        });
        if (attr_rc) return attr_rc;
    }
    if (causal) hipLaunchKernelGGL((attn_fwd_wave<NT, NW, true>), dim3(B * H), dim3(64 * NW), bytes, s, qkv, out, lse, L, H, 1, seq_offs, q8);
    else hipLaunchKernelGGL((attn_fwd_wave<NT, NW, false>), dim3(B * H), dim3(64 * NW), bytes, s, qkv, out, lse, L, H, 0, seq_offs, q8);
    ILVLM_LAUNCH_CHECK("attention_fwd");
    return ILVLM_OK;
}
template <int NT, int NW = NT>
int launch_bwd_wave(const bf16* dout, const bf16* qkv, const bf16* out, const float* lse, bf16* dqkv, int B, int L, int H,
                    int causal, const int* seq_offs, hipStream_t s, AttnQ8 q8) {
    constexpr int ROWS = ((NT + 1) & ~1) * 16;
    constexpr int bytes = 3 * ROWS * LDH * 2 + 2 * ROWS * 4;    // 57 KB at 128 tokens, 127 KB at 288
    if (bytes > 64 * 1024) {
        static std::once_flag once;
        static int attr_rc = ILVLM_OK;
        std::call_once(once, [&] { attr_rc = set_lds(attn_bwd_wave<NT, NW>, bytes, "attention_bwd"); });
        if (attr_rc) return attr_rc;
    }
    hipLaunchKernelGGL((attn_bwd_wave<NT, NW>), dim3(B * H), dim3(64 * NW), bytes, s, dout, qkv, out, lse, dqkv, L, H, causal,
                       seq_offs, q8);
    ILVLM_LAUNCH_CHECK("attention_bwd");
    return ILVLM_OK;
}

template <int LP>
int launch_bwd_tiled_bf16(const bf16* dout, const bf16* qkv, const bf16* out, const float* lse, bf16* dqkv, int B, int L, int H,
                          int causal, hipStream_t s, AttnQ8 q8) {
    constexpr int bytes = (2 * LP * LDH + 2 * 32 * LDH + 2 * LP * 40) * 2 + 2 * LP * 4;
    static std::once_flag once;
    static int attr_rc = ILVLM_OK;
    std::call_once(once, [&] { attr_rc = set_lds(attn_bwd_tiled_bf16<LP>, bytes, "attention_bwd_tiled"); });
    if (attr_rc) return attr_rc;
    hipLaunchKernelGGL((attn_bwd_tiled_bf16<LP>), dim3(B * H), dim3(256), bytes, s, dout, qkv, out, lse, dqkv, L, H, causal, q8);
    ILVLM_LAUNCH_CHECK("attention_bwd_tiled");
    return ILVLM_OK;
}

}  // namespace

// Lcap: longest sequence of the launch (tile count / LDS size); L: row stride of lse and, for the dense layout
// (seq_offs == nullptr), the length of every sequence
static int attention_fwd_impl(const void* qkv, void* out, float* lse, int dtype, int B, int L, int Lcap, int H, int causal,
                              const int* seq_offs, void* stream, AttnQ8 q8 = AttnQ8{nullptr, nullptr, nullptr}) {
    ILVLM_REQUIRE(qkv && out && lse, "attention_fwd: null pointer");
    ILVLM_REQUIRE(!q8.amax || (dtype == ILVLM_BF16 && (!q8.out8 || q8.scale)), "attention_fwd: the fp8 copy needs bf16 and a scale");
    ILVLM_REQUIRE(B > 0 && L > 0 && H > 0 && Lcap > 0 && Lcap <= L, "attention_fwd: bad shape B=%d L=%d Lcap=%d H=%d", B, L, Lcap, H);
    ILVLM_REQUIRE(!seq_offs || Lcap <= (dtype == ILVLM_BF16 ? 128 : 96), "attention_fwd: packed rows support sequences up to 128 (bf16) / 96 (f32) tokens");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ILVLM_BF16) {
        ILVLM_REQUIRE(L <= 288, "attention_fwd(bf16): L=%d > 288 not supported", L);
        const bf16* q = (const bf16*)qkv;
        bf16* o = (bf16*)out;
        switch ((Lcap + 15) / 16) {
            case 1: return launch_fwd_wave<1>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 2: return launch_fwd_wave<2>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 3: return launch_fwd_wave<3>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 4: return launch_fwd_wave<4>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 5: return launch_fwd_wave<5>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 6: return launch_fwd_wave<6>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 7: return launch_fwd_wave<7>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 8: return launch_fwd_wave<8>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 9: case 10: return launch_fwd_wave<10, 8>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 11: case 12: return launch_fwd_wave<12, 8>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 13: case 14: return launch_fwd_wave<14, 8>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 15: case 16: return launch_fwd_wave<16, 8>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            case 17: case 18: return launch_fwd_wave<18, 8>(q, o, lse, B, L, H, causal, seq_offs, s, q8);
            default: break;
        }
        ILVLM_FAIL(ILVLM_ERR_ARG, "attention_fwd(bf16): no kernel for L=%d", L);
    }
    ILVLM_REQUIRE(dtype == ILVLM_F32, "attention_fwd: bad dtype %d", dtype);
    if (Lcap > 96) {            // long sequences (ViT-B/16, ViT-L/14 in fp32 parity mode): global-memory kernel
        ILVLM_REQUIRE(!seq_offs && L <= LMAX_F32, "attention_fwd(f32): L=%d > %d (or packed rows > 96) not supported", L, LMAX_F32);
        hipLaunchKernelGGL(attn_fwd_f32_long, dim3(B * H, ceil_div(L, 4)), dim3(256), 0, s, (const float*)qkv, (float*)out, lse, L,
                           H, causal);
        ILVLM_LAUNCH_CHECK("attention_fwd_f32_long");
        return ILVLM_OK;
    }
    int bytes = (3 * Lcap * LDF + Lcap * (Lcap + 1)) * 4;
    static std::once_flag once_f;
    static int attr_rc_f = ILVLM_OK;
    std::call_once(once_f, [&] { attr_rc_f = set_lds(attn_fwd_f32, 160 * 1024, "attention_fwd_f32"); });
    if (attr_rc_f) return attr_rc_f;
    hipLaunchKernelGGL(attn_fwd_f32, dim3(B * H), dim3(256), bytes, s, (const float*)qkv, (float*)out, lse, L, H, causal, seq_offs);
    ILVLM_LAUNCH_CHECK("attention_fwd_f32");
    return ILVLM_OK;
}
extern "C" int ilvlm_attention_fwd(const void* qkv, void* out, float* lse, int dtype, int B, int L, int H, int causal,
                                   void* stream) {
    return attention_fwd_impl(qkv, out, lse, dtype, B, L, L, H, causal, nullptr, stream);
}
extern "C" int ilvlm_attention_packed_fwd(const void* qkv, void* out, float* lse, int dtype, int B, int L, int Lcap, int H,
                                          int causal, const int32_t* seq_offs, void* stream) {
    ILVLM_REQUIRE(seq_offs, "attention_packed_fwd: null seq_offs");
    return attention_fwd_impl(qkv, out, lse, dtype, B, L, Lcap, H, causal, seq_offs, stream);
}

static int attention_bwd_impl(const void* dout, const void* qkv, const void* out, const float* lse, void* dqkv, int dtype,
                              int B, int L, int Lcap, int H, int causal, const int* seq_offs, void* stream,
                              AttnQ8 q8 = AttnQ8{nullptr, nullptr, nullptr}) {
    ILVLM_REQUIRE(dout && qkv && out && lse && (dqkv || q8.out8), "attention_bwd: null pointer");
    ILVLM_REQUIRE(!q8.amax || (dtype == ILVLM_BF16 && (!q8.out8 || q8.scale)), "attention_bwd: the fp8 copy needs bf16 and a scale");
    ILVLM_REQUIRE(B > 0 && L > 0 && H > 0 && Lcap > 0 && Lcap <= L, "attention_bwd: bad shape B=%d L=%d Lcap=%d H=%d", B, L, Lcap, H);
    ILVLM_REQUIRE(!seq_offs || Lcap <= (dtype == ILVLM_BF16 ? 128 : 80), "attention_bwd: packed rows support sequences up to 128 (bf16) / 80 (f32) tokens");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ILVLM_BF16) {
        ILVLM_REQUIRE(L <= 288, "attention_bwd(bf16): L=%d > 288 not supported", L);
        const bf16 *d = (const bf16*)dout, *q = (const bf16*)qkv, *o = (const bf16*)out;
        bf16* dq = (bf16*)dqkv;
        switch ((Lcap + 15) / 16) {
            case 1: return launch_bwd_wave<1>(d, q, o, lse, dq, B, L, H, causal, seq_offs, s, q8);
            case 2: return launch_bwd_wave<2>(d, q, o, lse, dq, B, L, H, causal, seq_offs, s, q8);
            case 3: return launch_bwd_wave<3>(d, q, o, lse, dq, B, L, H, causal, seq_offs, s, q8);
            case 4: return launch_bwd_wave<4>(d, q, o, lse, dq, B, L, H, causal, seq_offs, s, q8);
            case 5: return launch_bwd_wave<5>(d, q, o, lse, dq, B, L, H, causal, seq_offs, s, q8);
            case 6: return launch_bwd_wave<6>(d, q, o, lse, dq, B, L, H, causal, seq_offs, s, q8);
            case 7: return launch_bwd_wave<7>(d, q, o, lse, dq, B, L, H, causal, seq_offs, s, q8);
            case 8: return launch_bwd_wave<8>(d, q, o, lse, dq, B, L, H, causal, seq_offs, s, q8);
            // longer sequences: the wave-per-tile backward would hold 14-18 tiles of dS per wave and spills at 256 VGPRs
            // (425 us against 325 us for the key-block kernel below at L = 257); the forward does profit (190 -> 117 us)
            default: break;
        }
        if (L <= 192) return launch_bwd_tiled_bf16<192>(d, q, o, lse, dq, B, L, H, causal, s, q8);
        if (L <= 224) return launch_bwd_tiled_bf16<224>(d, q, o, lse, dq, B, L, H, causal, s, q8);
        return launch_bwd_tiled_bf16<288>(d, q, o, lse, dq, B, L, H, causal, s, q8);
    }
    ILVLM_REQUIRE(dtype == ILVLM_F32, "attention_bwd: bad dtype %d", dtype);
    if (Lcap > 80) {
        ILVLM_REQUIRE(!seq_offs && L <= LMAX_F32, "attention_bwd(f32): L=%d > %d (or packed rows > 80) not supported", L, LMAX_F32);
        hipLaunchKernelGGL(attn_bwd_f32_long, dim3(B * H, ceil_div(L, 4), 2), dim3(256), 0, s, (const float*)dout,
                           (const float*)qkv, (const float*)out, lse, (float*)dqkv, L, H, causal);
        ILVLM_LAUNCH_CHECK("attention_bwd_f32_long");
        return ILVLM_OK;
    }
    int bytes = (4 * Lcap * LDF + 2 * Lcap * (Lcap + 1) + Lcap) * 4;
    static std::once_flag once_b;
    static int attr_rc_b = ILVLM_OK;
    std::call_once(once_b, [&] { attr_rc_b = set_lds(attn_bwd_f32, 160 * 1024, "attention_bwd_f32"); });
    if (attr_rc_b) return attr_rc_b;
    hipLaunchKernelGGL(attn_bwd_f32, dim3(B * H), dim3(256), bytes, s, (const float*)dout, (const float*)qkv,
                       (const float*)out, lse, (float*)dqkv, L, H, causal, seq_offs);
    ILVLM_LAUNCH_CHECK("attention_bwd_f32");
    return ILVLM_OK;
}
extern "C" int ilvlm_attention_bwd(const void* dout, const void* qkv, const void* out, const float* lse, void* dqkv,
                                   int dtype, int B, int L, int H, int causal, void* stream) {
    return attention_bwd_impl(dout, qkv, out, lse, dqkv, dtype, B, L, L, H, causal, nullptr, stream);
}
extern "C" int ilvlm_attention_packed_bwd(const void* dout, const void* qkv, const void* out, const float* lse, void* dqkv,
                                          int dtype, int B, int L, int Lcap, int H, int causal, const int32_t* seq_offs,
                                          void* stream) {
    ILVLM_REQUIRE(seq_offs, "attention_packed_bwd: null seq_offs");
    return attention_bwd_impl(dout, qkv, out, lse, dqkv, dtype, B, L, Lcap, H, causal, seq_offs, stream);
}

// fp8 mode: the attention kernels also emit the fp8 copy their consumer GEMM reads (forward: e4m3 of `out` for the output
// projection; backward: e5m2 of `dqkv` for the in-projection's input / weight gradients) and raise the slot's amax, saving a
// quantise pass over the tensor.  seq_offs nullable (dense rows, then Lcap = L).  Backward: dqkv may be NULL when only the fp8
// copy is consumed; sequences up to 128 tokens.
extern "C" int ilvlm_attention_fwd_q8(const void* qkv, void* out, float* lse, int dtype, int B, int L, int Lcap, int H, int causal,
                                      const int32_t* seq_offs, void* out8, const float* q_scale, float* q_amax, void* stream) {
    ILVLM_REQUIRE(q_amax, "attention_fwd_q8: null amax");
    return attention_fwd_impl(qkv, out, lse, dtype, B, L, seq_offs ? Lcap : L, H, causal, seq_offs, stream,
                              AttnQ8{(unsigned char*)out8, q_scale, q_amax});
}
extern "C" int ilvlm_attention_bwd_q8(const void* dout, const void* qkv, const void* out, const float* lse, void* dqkv, int dtype,
                                      int B, int L, int Lcap, int H, int causal, const int32_t* seq_offs, void* dqkv8,
                                      const float* q_scale, float* q_amax, void* stream) {
    ILVLM_REQUIRE(q_amax, "attention_bwd_q8: null amax");
    return attention_bwd_impl(dout, qkv, out, lse, dqkv, dtype, B, L, seq_offs ? Lcap : L, H, causal, seq_offs, stream,
                              AttnQ8{(unsigned char*)dqkv8, q_scale, q_amax});
}

#ifdef ILVLM_ATTN_STAMPS
extern "C" int ilvlm_debug_read_attn_stamps(unsigned long long* host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * n);
}
#endif
