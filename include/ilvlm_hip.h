/* libilvlm_hip.so -- C ABI of the MI355X (gfx950) kernels behind the CLIP / CLIP+FDT contrastive
 * training step of hellomuffin/iterated-learning-for-vlm.
 *
 * The reference has no FFI layer: its hot path is a sequence of ATen ops reached through torch.nn
 * (SURVEY.md section 2.2).  Each entry point below replaces one such op sequence; the reference
 * call site it stands for is cited as file:line relative to the reference root.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (PyTorch's
 *    caching allocator); the library never allocates, frees or retains device memory.
 *  - every function returns 0 on success, ILVLM_ERR_ARG (<0) for a rejected argument or a positive
 *    hipError_t; the message is in the thread-local ilvlm_last_error().  No exceptions, no exit().
 *  - launches are asynchronous on `stream` (a hipStream_t passed as void*); re-entrant (forward runs on
 *    the main thread, backward on autograd's worker thread).  The only process-wide mutable state is the
 *    kernel-selection knob of ilvlm_gemm_set_variant (one atomic int, a tuning / test hook) and the
 *    once-only hipFuncSetAttribute of each kernel (std::call_once).
 *  - dtype arguments: ILVLM_F32 or ILVLM_BF16.  "T" below means "the dtype argument of that call".
 *  - all matrices are row-major with the row stride given where it can differ from the width.
 *  - gradient outputs documented as "+=" are ACCUMULATED (fp32 atomics) into zero-initialised or
 *    previously accumulated buffers, which is autograd's .grad semantics.
 */
#ifndef ILVLM_HIP_H
#define ILVLM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILVLM_VERSION 300 /* round 3 */

enum { ILVLM_OK = 0, ILVLM_ERR_ARG = -1 };
enum { ILVLM_F32 = 0, ILVLM_BF16 = 1,
       ILVLM_FP8 = 2,      /* GEMM compute dtype: A and B are OCP fp8 e4m3 bytes                                       */
       ILVLM_FP8_BF8A = 3  /* GEMM compute dtype: A is OCP fp8 e5m2 (gradients), B is e4m3                             */ };
enum {
    ILVLM_ACT_NONE = 0,
    ILVLM_ACT_QUICKGELU = 1,     /* out = x*sigmoid(1.702x); pre-activation stored to aux  (base_transformer.py:24-26) */
    ILVLM_ACT_GELU_ERF = 2,      /* out = exact-erf GELU;     pre-activation stored to aux  (clip_fdt.py:89)          */
    ILVLM_ACT_QUICKGELU_BWD = 3, /* out = acc * quickgelu'(aux)                                                        */
    ILVLM_ACT_GELU_ERF_BWD = 4   /* out = acc * gelu_erf'(aux)                                                         */
};
enum { ILVLM_POOL_MAX = 0, ILVLM_POOL_MEAN = 1, ILVLM_POOL_SUM = 2 };

int ilvlm_version(void);
const char* ilvlm_last_error(void);

/* ---- GEMM with fused epilogue --------------------------------------------------------------
 * C[m,n] = epilogue( sum_k A(m,k) * B(n,k) )
 *   trans_a = 0: A stored [M,K] (lda >= K)      trans_a = 1: A stored [K,M] (lda >= M)
 *   trans_b = 0: B stored [N,K] (ldb >= K)      trans_b = 1: B stored [K,N] (ldb >= N)
 * so  nn.Linear forward  = (0,0) with B = weight[out,in]   (F.linear at base_transformer.py:35-41,
 *     clip_fdt.py:86-92; in/out projections inside nn.MultiheadAttention, base_transformer.py:45-48)
 *     its input gradient  = (0,1) with A = dY, B = weight
 *     its weight gradient = (1,1) with A = dY, B = X, accumulate = 1.
 * compute_dtype ILVLM_BF16: A,B bf16 -> v_mfma_f32_16x16x32_bf16, fp32 accumulate.
 * compute_dtype ILVLM_F32 : A,B f32  -> v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
 * compute_dtype ILVLM_FP8 / ILVLM_FP8_BF8A: A,B fp8 bytes (OCP e4m3; _BF8A: A is e5m2) -> v_mfma_f32_16x16x32_fp8_*,
 *   fp32 accumulate; (0,0) layout with K % 128 == 0, or -- ILVLM_FP8_BF8A only -- the weight-gradient form (1,1) with
 *   accumulate = 1 (A = dY [K, M] e5m2, B = X [K, N] e4m3, any K, M and N multiples of 16; a_rowsum is de-quantised by
 *   alpha_ptr alone); lda / ldb in elements (= bytes); the de-quantisation scales go through alpha_ptr (A) / alpha_ptr2
 *   (B); aux (activation epilogues) is bf16, out is bf16 or fp32 (BASELINE configs[4]).
 * Epilogue, in this order: acc *= alpha * (alpha_ptr ? *alpha_ptr : 1); += bias[n];
 *   += rowbias[(out_skip + m % out_group) * N + n]; activation (see ILVLM_ACT_*; aux is [M,N] with
 *   stride ldc, dtype = compute_dtype); += residual (fp32, laid out like C); store as out_dtype
 *   (accumulate = 1: fp32 atomic add, no other epilogue term except alpha).
 * out_group > 0 maps output row m to (m / out_group) * (out_group + out_skip) + out_skip + m % out_group
 *   (patch tokens written behind the class token, visual_transformer.py:56-63).
 * split_k >= 1 partitions K over that many workgroups per tile (requires accumulate = 1 if > 1).
 * Alignment: bf16 operands need 16-byte aligned bases and lda/ldb multiples of 8; fp32 operands may have any
 *   leading dimension (float4 staging when 16-byte aligned rows, scalar loads otherwise).
 */
typedef struct ilvlm_gemm_epilogue {
    const float* bias;      /* [N] or NULL */
    const float* rowbias;   /* [(out_group+out_skip), N] or NULL */
    const float* residual;  /* fp32, same layout/stride as C, or NULL */
    void* aux;              /* [M,N] stride ldc, compute dtype; written (ACT fwd) or read (ACT bwd) */
    const float* alpha_ptr; /* device scalar or NULL */
    float alpha;            /* host scalar (use 1.0f) */
    int act;                /* ILVLM_ACT_* */
    int out_dtype;          /* ILVLM_F32 / ILVLM_BF16 (bf16 only with compute_dtype bf16) */
    int accumulate;         /* 0 store, 1 fp32 atomic += */
    int out_group, out_skip;
    float* a_rowsum;        /* optional [M]: += sum_k A(m,k) (bias gradient fused into the weight-gradient GEMM:
                               A = dY^T); needs accumulate = 1, bf16 compute, K % 64 == 0, M % 8 == 0 */
    /* token max-pool epilogue (FDT codebook scores, clip_fdt.py:113-145): when pool_out != NULL nothing is stored to C;
     * instead pool_out[seq(m) * N + n] = max over the rows m of one sequence of (sortable key of alpha * acc) << 32 |
     * (0x7fffffff - token index) by 64-bit atomic max, so the [M, N] score matrix never reaches memory.  seq(m) =
     * pool_seq[m] and token = m - pool_offs[seq] (packed text rows), or m / pool_group and m % pool_group.  bf16 compute,
     * direct-to-LDS kernels only; used through ilvlm_fdt_score_pool_fwd. */
    unsigned long long* pool_out;
    const int32_t* pool_seq;
    const int32_t* pool_offs;
    int pool_group;
    const float* alpha_ptr2; /* second device scalar multiplied into alpha (fp8: the two de-quantisation scales) or NULL */
    /* fp8 copy of the stored output for the next fp8 GEMM (fp8 mode): out8[m, n] (bytes, same ldc) = fp8(value * out8_scale[0])
     * in e4m3 (out8_fmt 0) or e5m2 (1); out8_amax[0] is raised to max|value| (either may be NULL).  Direct-to-LDS bf16 /
     * fp8 kernels only, not with accumulate or the pool epilogue, out_group == 0.  With out8 given, C may be NULL: only the
     * copy (and aux) is kept -- the bf16 tensor would be written for nobody when every consumer reads the fp8 copy. */
    void* out8;
    const float* out8_scale;
    float* out8_amax;
    int out8_fmt;
    /* slab split-K for accumulate = 1, split_k > 1 (optional; all four NULL / 0 = fp32 atomics): every K-slice stores its
     * 128x128 tile into the workspace, the workgroup drawing the tile's last ticket adds the slabs in slice order and is the
     * only writer of C -- no atomics on C, and a sum that does not depend on arrival order.  splitk_ws: device buffer of
     * splitk_ws_bytes >= tiles * split_k * 64 KiB; splitk_cnt: int32[splitk_cnt_len >= tiles], ZERO before first use (the
     * kernel leaves it zero); one launch at a time per workspace (i.e. one workspace per stream).  Taken by the 128x128
     * direct-to-LDS weight-gradient kernels (bf16 and fp8) for split_k <= 8; otherwise silently the atomic form. */
    void* splitk_ws;
    long splitk_ws_bytes;
    int32_t* splitk_cnt;
    int splitk_cnt_len;
    /* optional copy of the B operand in MFMA-fragment order (ilvlm_gemm_pack_b / ilvlm_pack_weights), N x K bf16 elements:
     * store-type bf16 GEMMs with a K-contiguous A operand (trans_a = 0, accumulate = 0, no pool epilogue, K % 64 == 0,
     * N % 16 == 0) then run the streaming kernel -- A through a two-deep LDS ring, B straight from the packed copy into
     * registers in whole 1 KiB wave loads -- instead of the direct-to-LDS 128x128 kernel; bit-identical results
     * (same MFMA sequence per output element).  B / ldb are ignored by that kernel but must still be valid. */
    const void* b_packed;
} ilvlm_gemm_epilogue;

int ilvlm_gemm(int compute_dtype, int trans_a, int trans_b, int M, int N, int K, const void* A, int lda,
               const void* B, int ldb, void* C, int ldc, const ilvlm_gemm_epilogue* epi, int split_k, void* stream);
/* bf16 kernel selection (tuning / tests; process-wide atomic): 15 (default) the streaming kernel where the epilogue offers
 * b_packed and K >= 512 (ILVLM_PK_MIN_K), the two-stage direct-to-LDS kernel for weight gradients and the
 * single-stage direct-to-LDS 128x128 kernel elsewhere; 16 as 15 but the streaming kernel for every eligible shape (tests);
 * 17 as 16 plus store-type split-K with an in-launch slab reduction where splitk_ws / splitk_cnt are offered (tests);
 * 18 / 19: see ilvlm_gemm_set_persistent;
 * 5 always the single-stage direct-to-LDS 128x128 kernel (the A/B reference); 0 the register-staged general kernel only.  Shapes the direct-to-LDS kernels cannot take (K % 64 != 0, ragged
 * K-strided operands) always use the general kernel. */
int ilvlm_gemm_set_variant(int variant);
/* The persistent streaming kernel (opt-in: ILVLM_PKP=1, for b_packed products with an even number of 64-deep K-tiles; selector
 * 18 forces it for every eligible shape, 19 = selector 15 without it): `slots` workgroups per launch, each walking several output
 * tiles with the next tile's operands in flight under the current tile's epilogue (0 = two per CU); epi_sep: where the
 * epilogue transposes, 1 = LDS of its own behind the operand ring (80 KiB per workgroup), 0 = the ring's free stage (48 KiB),
 * 2 = half and half (64 KiB),
 * -1 = default (ILVLM_PKP_EPI_SEP); stagger: the second half of the grid (by dispatch order the second workgroup of each CU)
 * starts its first tile this many shader cycles per K-tile late, so that the two workgroups of a CU alternate between
 * multiplying and storing instead of doing both in lock-step (-1 = default / ILVLM_PKP_STAGGER, 0 = none; results do not
 * depend on it).  Tuning / test hook, process-wide atomics. */
int ilvlm_gemm_set_persistent(int slots, int epi_sep, int stagger);
/* Tile height of the streaming kernel (round 4): 128 (default), 96 or 64 rows per workgroup tile; 0 = chosen per launch by a
 * rounds x tile-time cost model; -1 = back to the default (ILVLM_PK_TI = 8 | 6 | 4 | 0 sets it process-wide).  A launch whose
 * 128-row tiles number fewer than the workgroup slots takes as long as ONE tile takes, and shorter tiles that still fit one
 * round shorten it -- alone; inside the multi-stream step the empty slots are filled anyway and the shorter tiles lose
 * (DESIGN.md section 6, round 4), hence the default.  Results do not depend on the height (same MFMA sequence per output
 * element).  Tuning / test hook, process-wide atomic. */
int ilvlm_gemm_set_tile_rows(int rows);
/* Workgroup tile of the bf16 weight-gradient kernel ((1,1) layout, accumulate): 128 = 128 x 128 (four waves of 64 x 64, two-stage
 * operand ring; the default), 256 = 256 x 128 (four waves of 128 x 64: a quarter fewer LDS-DMA pieces and transposing fragment
 * reads per MFMA; two stages, 96 KiB of LDS), 257 = the same tile single-stage (48 KiB); taken when M % 256 == 0.  -1 = default
 * (ILVLM_WGRAD_TILE).  Same K-slices, same order of K-tiles per output element: results equal the 128 x 128 tile's bit for bit at
 * one K-slice or with the slab workspace.  The grouped fp8 weight gradients (ilvlm_wgrad_group, ILVLM_FP8_BF8A) follow the same
 * selector: 128 = the 128 x 128 tile on the non-scaled fp8 MFMA, >= 256 = 256 x 128 tiles on the block-scaled MFMA for the output
 * columns from 128 up (their default: ILVLM_FP8_WGRAD_TILE=256).  Tuning / test hook, process-wide atomic. */
int ilvlm_gemm_set_wgrad_tile(int rows);
/* Regime hint, process-wide: concurrent != 0 declares that the caller keeps several GEMM streams in flight (the engine's two
 * towers with their weight-gradient companion streams).  ilvlm_gemm then picks, where two kernels compute the same result, the
 * one measured best INSIDE such a step rather than alone -- today the bf16 weight gradients: single-stage 256 x 128 tiles
 * (slower alone, 1.2-1.4 % faster steps) instead of two-stage 128 x 128 tiles.  Default 0.  Results do not depend on it beyond
 * fp32 summation order between K-slices (bit-identical at one K-slice or through the slab workspace). */
int ilvlm_gemm_set_concurrent(int concurrent);
int ilvlm_gemm_get_concurrent(void);
/* B operand of ilvlm_gemm in MFMA-fragment order (the `b_packed` epilogue field).  With Bop[n][k] = B[n * ldb + k]
 * (trans_b = 0) or B[k * ldb + n] (trans_b = 1): the 16 x 32 block (n / 16, k / 32) is one contiguous KiB, block index
 * (n / 16) * (K / 32) + k / 32, and lane l of a wave owns its bytes [16 l, 16 l + 16): Bop[16 (n/16) + (l & 15)][32 (k/32) +
 * 8 (l >> 4) + j], j = 0..7 -- the operand register image of v_mfma_f32_16x16x32_bf16.  N % 16 == 0, K % 32 == 0. */
int ilvlm_gemm_pack_b(int trans_b, int N, int K, const void* B, int ldb, void* packed, void* stream);
/* fp8 (e4m3) B operand, row-major B8[n * ldb + k], in the fragment order of the streaming kernel's fp8 form (b_packed with
 * compute dtype ILVLM_FP8 / ILVLM_FP8_BF8A, (0,0) layout): the 16 x 128 block (n / 16, k / 128) is 2 contiguous KiB, block index
 * (n / 16) * (K / 128) + k / 128; lane l owns bytes [16 l, 16 l + 16) of the first KiB = B8[16 (n/16) + (l & 15)][128 (k/128) +
 * 16 (l >> 4) + j] and the same bytes of the second KiB = ... [128 (k/128) + 64 + 16 (l >> 4) + j], j = 0..15 -- together the
 * 32-byte operand of v_mfma_scale_f32_16x16x128_f8f6f4.  N % 16 == 0, K % 128 == 0.  Bit-identical results to the row-major call. */
int ilvlm_gemm_pack_b8(int N, int K, const void* B8, int ldb, void* packed, void* stream);
/* The same for every GEMM weight of a flat bf16 parameter arena in one launch (nn.Linear weights [out, in] of the residual
 * attention blocks; base_transformer.py:35-48): table = n_tiles x {arena offset / 64, rows, cols, r0, c0} int32, one entry
 * per 64 x 64 tile of a weight (rows, cols and the offset multiples of 64).  fwd (same offsets) receives the trans_b = 0
 * image of W (the forward product X W^T), bwd the trans_b = 1 image (the input gradient dY W). */
int ilvlm_pack_weights(const void* arena_bf16, void* fwd, void* bwd, const int32_t* table, int n_tiles, void* stream);
/* Grouped weight gradients: gw_p[n_p, k_p] += dy_p^T x_p (and gb_p[n_p] += column sums of dy_p) for `count` nn.Linear layers
 * that share their token rows -- the four linears of one residual attention block (autograd of F.linear at
 * image_encoder/base_transformer.py:35-41,45-48; text twin text_encoder/base_transformer.py:33-48) -- as ONE launch over the
 * concatenated 128 x 128 tile lists.  dy_p: [rows, n_p], x_p: [rows, k_p], row-major and compact, compute dtype
 * ILVLM_BF16, or ILVLM_FP8_BF8A (dy e5m2, x e4m3, inv_g / inv_x = their de-quantisation scales on the device).
 * split_target: workgroup slots of the chip for this kernel (512 = two per CU); the number of K-slices (1..16, >= 256 rows
 * each) minimises ceil(tiles * slices / slots) * (ceil(K-tiles / slices) + e), e = 8 K-tiles' worth for the plain epilogue of
 * one slice and 25 for the atomic one.  At one K-slice every tile has a single writer and is added with plain loads and
 * stores; the caller guarantees that no OTHER launch accumulates into the same gw concurrently (launches on one stream are
 * fine).  n_p, k_p multiples of 8 (fp8: 16).  Under ilvlm_gemm_set_concurrent(1) bf16 problems whose n_p are multiples of 256 run
 * the single-stage 256 x 128 tile at ONE K-slice per tile (slot target capped at ILVLM_WGRAD_GROUP_SLOTS_WIDE = 128); a NEGATIVE
 * split_target lifts that cap ("spread": the caller knows that nothing else runs beside this launch -- the last block of a
 * tower's backward -- and wants the chip filled).  fp8 problems with >= 16384 rows run 256 x 128 tiles on the block-scaled MFMA
 * for the output columns from 128 up and the 128 x 128 tile for the first tile column (two launches). */
#define ILVLM_WGRAD_GROUP_MAX 4
typedef struct ilvlm_wgrad_problem {
    const void* dy;
    const void* x;
    float* gw;          /* [n, k] fp32, accumulated */
    float* gb;          /* [n] fp32, accumulated; or NULL */
    int n, k;
    const float* inv_g; /* fp8 only */
    const float* inv_x; /* fp8 only */
} ilvlm_wgrad_problem;
int ilvlm_wgrad_group(int compute_dtype, const ilvlm_wgrad_problem* problems, int count, long rows, int split_target,
                      void* stream);

/* ---- LayerNorm (nn.LayerNorm eps 1e-5 affine; base_transformer.py:10-18, clip_fdt.py:86-92) ----
 * y[r,:] = (x[R,:] - mean) * rstd * gamma + beta, R = map(r) when in_group > 0 (row remap as above:
 * the FDT image query reads patch tokens x[:,1:,:], visual_transformer.py:72).  mean/rstd: [rows]. */
int ilvlm_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                        float* mean, float* rstd, long rows, int cols, float eps, int in_group, int in_skip,
                        void* stream);
/* the same, additionally emitting an OCP e4m3 copy y8 = fp8(y * q_scale[0]) of the output (the next GEMM's fp8 operand) and
 * raising q_amax[0] to max|y| (either may be NULL); compact rows only (fp8 mode, BASELINE configs[4]) */
int ilvlm_layernorm_fwd_q8(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                           float* mean, float* rstd, long rows, int cols, float eps, int in_group, int in_skip, void* y8,
                           const float* q_scale, float* q_amax, void* stream);
/* dx = LN'(dy) [+ dres]; writes dx_f32 (fp32) and/or dx_lp (dtype dx_lp_dtype, optionally multiplied by
 * act'(act_aux) for act in {ILVLM_ACT_QUICKGELU_BWD, ILVLM_ACT_GELU_ERF_BWD}); dgamma/dbeta "+=".
 * x, dres, dx_f32, dx_lp use the remapped rows when group > 0; dy, mean, rstd, act_aux are compact. */
int ilvlm_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                        const float* rstd, const float* gamma, const float* dres, float* dx_f32, void* dx_lp,
                        int dx_lp_dtype, int act, const void* act_aux, float* dgamma, float* dbeta, long rows,
                        int cols, int group, int skip, float* ws, int ws_blocks, void* stream);
/* ws (2 * |ws_blocks| * cols floats) selects the two-stage dgamma / dbeta reduction; ws_blocks < 0 DEFERS its second stage:
 * the launch only leaves its partial rows in ws, and one ilvlm_layernorm_bwd_reduce_batched call later adds the partials of
 * n_slots such launches (slot z at ws + z * slot_stride, same rows / |ws_blocks| / cols for all) into grad_ptrs[2z]
 * (dgamma) and grad_ptrs[2z+1] (dbeta), a DEVICE array of 2 * n_slots pointers -- one launch per tower instead of one
 * per LayerNorm (50 per step). */
/* ilvlm_layernorm_bwd that also emits an OCP e5m2 copy dx8 = fp8(dx_lp * q_scale[0]) of the low-precision gradient and
 * raises q_amax[0] to max|dx_lp| (either may be NULL); compact rows only */
int ilvlm_layernorm_bwd_q8(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                           const float* gamma, const float* dres, float* dx_f32, void* dx_lp, int dx_lp_dtype, int act,
                           const void* act_aux, float* dgamma, float* dbeta, long rows, int cols, int group, int skip, float* ws,
                           int ws_blocks, void* dx8, const float* q_scale, float* q_amax, void* stream);
int ilvlm_layernorm_bwd_reduce_batched(const float* ws, long slot_stride, int n_slots, long rows, int ws_blocks, int cols,
                                       float* const* grad_ptrs, void* stream);
/* ws: optional workspace of 2 * ws_blocks * cols floats: per-workgroup dgamma/dbeta partials are written there and
 * summed by a second tiny kernel (deterministic, no atomic contention); ws == NULL falls back to fp32 atomics. */

/* ---- multi-head self attention core, head_dim 64 (F.multi_head_attention_forward math path reached
 * from base_transformer.py:45-48 / text_encoder/base_transformer.py:45-48; additive causal mask
 * text_transformer.py:147-153).  qkv: T [B*L, 3*64*H] rows = (b, l), columns = [q | k | v] each split
 * into H heads of 64 (bf16: L <= 288; fp32 parity kernels: L <= 96 forward, 80 backward).  out: T [B*L, 64*H].
 * lse: fp32 [B, H, L] row log-sum-exp saved for backward
 * (the reference materialises the [B*H, L, L] probabilities and their head mean; neither is needed). */
int ilvlm_attention_fwd(const void* qkv, void* out, float* lse, int dtype, int B, int L, int H, int causal,
                        void* stream);
int ilvlm_attention_bwd(const void* dout, const void* qkv, const void* out, const float* lse, void* dqkv,
                        int dtype, int B, int L, int H, int causal, void* stream);

/* ---- token embedding + positional embedding (text_transformer.py:228-231) ---- */
int ilvlm_embed_fwd(const int64_t* tokens, const float* table, const float* pos, float* x, int B, int L, int W,
                    int vocab, void* stream);
/* dtable[tokens[b,l],:] += dx[b,l,:];  dpos[l,:] += sum_b dx[b,l,:] */
int ilvlm_embed_bwd(const int64_t* tokens, const float* dx, float* dtable, float* dpos, int B, int L, int W,
                    int vocab, void* stream);

/* ---- ViT patch embedding helpers (nn.Conv2d k=s=patch as an im2col GEMM, visual_transformer.py:56-63) ----
 * patches[(b*g*g + py*g + px), c*ps*ps + ky*ps + kx] = images[b,c,py*ps+ky,px*ps+kx]  (T out); rows have stride ld and
 * columns C*ps*ps .. ld-1 are written as zeros (ViT-L/14: 588 -> 640 so the patch GEMM's K is a multiple of 64) */
int ilvlm_patchify(const float* images, void* patches, int dtype, int B, int C, int res, int ps, int ld, void* stream);
/* tokens[b,0,:] = cls + pos[0,:]   (tokens: fp32 [B, L, W]) */
int ilvlm_cls_rows(const float* cls, const float* pos, float* tokens, int B, int L, int W, void* stream);
/* out[l,:] += sum_b x[b,l,:] (positional-embedding grad); out0[:] += sum_b x[b,0,:] if out0 (class embedding grad) */
int ilvlm_batch_sum(const float* x, float* out, float* out0, int B, int L, int W, void* stream);
/* gather / scatter of one row per batch element: y[b,:] = x[b, idx[b], :]; dx[b, idx[b], :] += dy[b,:] */
int ilvlm_gather_rows(const float* x, const int64_t* idx, float* y, int B, int L, int W, void* stream);
int ilvlm_scatter_rows(const float* dy, const int64_t* idx, float* dx, int B, int L, int W, void* stream);

/* ---- FDT codebook attention (Query_model.forward, clip_fdt.py:113-145) ----
 * scores: fp32 [B*T, C] = q @ sd^T.  v = ((s / sqrt_d) * (pad_mask[b,t] == 0)) / temperature;
 * pooled[b,c] = max_t | mean_t | sum_t v;  argmax: int32 [B,C] (max only). */
int ilvlm_fdt_pool_fwd(const float* scores, const float* pad_mask, float* pooled, int* argmax, int B, int T, int C,
                       float sqrt_d, float temperature, int pool, void* stream);
/* dscores[b,t,c] (dtype T) = dpooled[b,c] * d v / d s, routed to the argmax token for max pooling */
/* Fused codebook scores + scale + token max-pool with argmax (clip_fdt.py:113-145 with pool_type 'max'):
 * pooled[b,c] = max_t ((q[row(b,t)] . sd[c]) / sqrt_d) / temperature over the tokens of sequence b, argmax[b,c] = that t.
 * The [rows, C] score matrix is never materialised: the bf16 MFMA GEMM's epilogue reduces each 32-row group per sequence
 * and merges by 64-bit atomic max into `packed_ws` ([B, C] uint64, caller-owned scratch), which a decode pass turns into
 * pooled / argmax.  Dense layout (seq_offs == NULL): rows = B * T, no pad mask (image side).  Packed text rows: seq_offs
 * [B+1] and row_seq [rows] (sequence of every row); captions shorter than T also pool over their masked positions, which
 * contribute exactly 0 (argmax = length marks "no token").  Requires temperature > 0, q / sd bf16, d % 64 == 0. */
int ilvlm_fdt_score_pool_fwd(const void* q, const void* sd, unsigned long long* packed_ws, float* pooled, int* argmax,
                             long rows, int B, int T, int C, int d, float sqrt_d, float temperature,
                             const int32_t* seq_offs, const int32_t* row_seq, void* stream);
int ilvlm_fdt_pool_bwd(const float* dpooled, const int* argmax, const float* pad_mask, void* dscores, int dtype,
                       int B, int T, int C, float sqrt_d, float temperature, int pool, void* stream);

/* ---- row-wise simplex maps over C codes (sparsemax.py:22-71; nn.Softmax clip_fdt.py:73-78) ---- */
int ilvlm_sparsemax_fwd(const float* z, float* out, int rows, int cols, void* stream);
int ilvlm_sparsemax_bwd(const float* out, const float* g, float* dz, int rows, int cols, void* stream);
int ilvlm_softmax_fwd(const float* z, float* out, int rows, int cols, void* stream);
int ilvlm_softmax_bwd(const float* out, const float* g, float* dz, int rows, int cols, void* stream);
/* att_func_type 'sigmoid' (nn.Sigmoid clip_fdt.py:76-78 and the division of the weighted sum by the weights' row sum,
 * clip_fdt.py:156-157): w = sigmoid(z) (the returned attention weights), wn = w / rowsum (the operand of the weighted
 * codebook sum), rowsum[row] = sum_c w.  Backward: g = d loss / d wn -> dz. */
int ilvlm_sigmoid_norm_fwd(const float* z, float* w, float* wn, float* rowsum, int rows, int cols, void* stream);
int ilvlm_sigmoid_norm_bwd(const float* w, const float* wn, const float* rowsum, const float* g, float* dz, int rows, int cols,
                           void* stream);

/* ---- y = x / (||x|| + eps) (clip_fdt.py:411-412; clip.py:133-134) ---- */
int ilvlm_l2norm_fwd(const float* x, float* y, float* norm, int rows, int cols, float eps, void* stream);
int ilvlm_l2norm_bwd(const float* x, const float* norm, const float* dy, float* dx, int rows, int cols, float eps,
                     void* stream);

/* ---- temperature: out[0] = min(exp(logit_scale[0]), max_scale) (clip_fdt.py:415-416) ---- */
int ilvlm_logit_scale_fwd(const float* logit_scale, float* out, float max_scale, void* stream);
/* dparam[0] += (sum dli*li + sum dlt*lt) / scale_used * exp(param)   (d/d param of logits = scale * cos) */
int ilvlm_logit_scale_bwd(const float* dli, const float* li, const float* dlt, const float* lt, long n,
                          const float* logit_scale, const float* scale_used, float* dparam, void* stream);

/* ---- InfoNCE (ClipInfoCELoss.forward, loss.py:37-47): labels[r] = label_offset + r.
 * loss[0] = (CE(logits_i) + CE(logits_t)) / 2 (mean over B rows);
 * dlogits_* = d loss / d logits_* = (softmax - onehot) / (2B). */
int ilvlm_infonce_fwd(const float* logits_i, const float* logits_t, int B, int Bg, int label_offset, float* loss,
                      float* dlogits_i, float* dlogits_t, void* stream);
/* accuracy() of misc.py:464-477: out[0] = 100/B * #{rows whose label is in the top-1}, out[1] same for top-k */
int ilvlm_topk_accuracy(const float* logits, int B, int Bg, int label_offset, int k, float* out, void* stream);

/* ---- fp8 path (BASELINE.json configs[4]; no counterpart in the reference, whose nn.Linear calls are fp32:
 * image_encoder/base_transformer.py:35-48).  Per-tensor delayed scaling: q = saturate(x * scale[slot]) in OCP e4m3
 * (fmt 0, activations and weights) or e5m2 (fmt 1, gradients); every quantising call also raises amax[slot] to max|x| of
 * what it saw; ilvlm_fp8_scale_update turns the amax history of each slot into next step's scale (fmt_max / max(history))
 * and its inverse (the GEMM's alpha_ptr / alpha_ptr2).  dst == NULL observes amax without quantising. */
int ilvlm_fp8_quantize(const void* src, int src_dtype, void* dst, long n, const float* scale, float* amax, int fmt, void* stream);
/* all GEMM weights of the towers in one launch: tile_table holds 6 ints per 64 x 64 tile {arena offset / 64, rows, cols,
 * scale slot, tile row, tile col}; w8 receives the e4m3 weight in its own [rows, cols] layout (forward operand), w8t its
 * transpose [cols, rows] (input-gradient operand), both at the tensor's offset in an arena-shaped byte buffer. */
int ilvlm_fp8_quantize_weights(const float* params, void* w8, void* w8t, const int32_t* tile_table, int n_tiles,
                               const float* scale, float* amax, void* stream);
/* as above, plus both copies in the fragment order the streaming GEMM reads with fp8 operands (w8p: forward operand, w8tp:
 * input-gradient operand; ilvlm_gemm_epilogue.b_packed, layout of ilvlm_gemm_pack_b8), each at the tensor's arena offset.
 * Every matrix of the table needs rows % 128 == 0 and cols % 128 == 0. */
int ilvlm_fp8_quantize_weights_packed(const float* params, void* w8, void* w8t, void* w8p, void* w8tp, const int32_t* tile_table,
                                      int n_tiles, const float* scale, float* amax, void* stream);
int ilvlm_fp8_scale_update(float* amax_cur, float* hist, float* scale, float* inv_scale, const float* fmt_max, int n_slots,
                           int hist_len, int pos, void* stream);

/* ---- input pipeline: uint8 image batch -> normalised fp32 NCHW on the device.
 * dst[b,c,y,x] = (src(b,c,y,x') / 255 - mean[c]) / std[c], i.e. transforms.ToTensor + transforms.Normalize of
 * prototype/data/imagenet_dataloader.py:13-14 (the tail of every augmentation list, :59-68), with the two MOCOV2_single
 * augmentations that are pure pixel functions once their coin is tossed: flags[b] bit 0 = horizontal flip (x' = W-1-x),
 * bit 1 = grayscale (PIL "L" luma, replicated to 3 channels).  src is [B,H,W,3] (nhwc = 1) or [B,3,H,W] uint8; flags may
 * be NULL; mean3 / std3 are HOST arrays of 3 floats.  The multiplication by 1/std rounds once more than a division: results
 * equal the float CPU pipeline to 1 ulp (tests/test_input_pipeline_gpu.py). */
int ilvlm_image_u8_normalize(const unsigned char* src, int nhwc, const unsigned char* flags, float* dst, int B, int H, int W,
                             const float* mean3, const float* std3, void* stream);

/* ---- input pipeline, the random augmentations: MOCOV2_single of prototype/data/imagenet_dataloader.py:59-68
 * (RandomResizedCrop, ColorJitter @ 0.8, RandomGrayscale, GaussianBlur @ 0.5, RandomHorizontalFlip, ToTensor, Normalize) from
 * DECODED uint8 images of any size.  The random draws are made on the host (what torchvision's get_params return, one
 * ilvlm_augment_params per sample); the pixel work runs on the device in PIL's own arithmetic -- the crop box resampled to
 * out_size x out_size as Image.resize(BILINEAR) does it (8-bit fixed point, an 8-bit image between the two passes), the four
 * colour operations in the drawn order with PIL's integer luma / blend / HSV arithmetic, luma replacement, the box-blur passes
 * behind ImageFilter.GaussianBlur(radius = sigma) (prototype/data/transforms.py:82-91), flip, float32 ToTensor / Normalize --
 * so that, given the same draws, dst equals what the reference's loader workers produce bit for bit.  src: the images back to back as [H][W][3] uint8, image b at byte
 * src_offsets[b] with src_hw[2b], src_hw[2b+1] = its height, width (device arrays; the crop box must lie inside the image);
 * dst: fp32 [B,3,out_size,out_size]; scratch: ilvlm_image_augment_scratch_floats(B, out_size, max_crop_rows) floats, where
 * max_crop_rows >= every crop_h; mean3 / std3: HOST arrays.  tests/test_augment_cpu.py (restatement = PIL, exactly) and
 * tests/test_input_pipeline_gpu.py (kernels = restatement, exactly). */
typedef struct ilvlm_augment_params {
    int crop_top, crop_left, crop_h, crop_w; /* RandomResizedCrop.get_params */
    int jitter;                              /* ColorJitter applied (RandomApply, p = 0.8) */
    int jitter_order;                        /* its four operations in application order, 2 bits each from bit 0:
                                                0 brightness, 1 contrast, 2 saturation, 3 hue (torch.randperm(4)) */
    float brightness, contrast, saturation;  /* factors, U(0.6, 1.4) for the shipped 0.4 */
    float hue;                               /* U(-0.1, 0.1) */
    int grayscale;                           /* RandomGrayscale (p = 0.2) */
    float blur_sigma;                        /* > 0: GaussianBlur with this sigma (p = 0.5, U(0.1, 2.0)); 0: none */
    int flip;                                /* RandomHorizontalFlip */
    int pad_;
} ilvlm_augment_params;
long ilvlm_image_augment_scratch_floats(int B, int out_size, int max_crop_rows);
int ilvlm_image_augment(const unsigned char* src, const long* src_offsets, const int32_t* src_hw, const ilvlm_augment_params* params,
                        float* dst, float* scratch, int B, int out_size, int max_crop_rows, const float* mean3, const float* std3,
                        void* stream);

/* ---- small utilities ---- */
/* out[c] += sum_r x[r,c]  (bias gradients) */
int ilvlm_colsum(const void* x, int dtype, float* out, long rows, int cols, int ld, void* stream);
/* dst (dtype) = src (fp32): bf16 shadow of the fp32 master weights */
int ilvlm_cast_f32(const float* src, void* dst, int dst_dtype, long n, void* stream);
/* dst (fp32) = src (bf16): a bf16 gradient bucket back into the fp32 gradient arena after the data-parallel mean
 * (the reference reduces fp32 buckets inside torch DDP, prototype/utils/torch_ddp_dist.py:65) */
int ilvlm_cast_to_f32(const void* src, int src_dtype, float* dst, long n, void* stream);
/* y = a * x (fp32, in place allowed) */
int ilvlm_scale(const float* x, float* y, float a, long n, void* stream);
/* y = a[0] * x with the scalar on the device (upstream gradient of the loss, loss.py:46 / train_solver.py:420) */
int ilvlm_scale_dev(const float* x, float* y, const float* a, long n, void* stream);
/* y += x (own-slice gradient of the gathered embeddings, clip_fdt.py:182-188) */
int ilvlm_add_inplace(float* y, const float* x, long n, void* stream);
/* x = min(max(x, lo), hi) in place (logit_scale.data.clamp_, train_solver.py:381-382, 397-398) */
int ilvlm_clamp(float* x, float lo, float hi, long n, void* stream);
/* Gradient-norm clipping over the flat gradient arena (grad_clip.type 'norm': clip_grad_norm_, prototype/utils/grad_clip.py:12-47,
 * called at train_solver.py:403-405): out[0] += sum x^2 (zero it first), then x *= max_norm / (sqrt(sumsq[0]) + 1e-6) when that
 * factor is below 1; the norm never leaves the device.  The sum is bitwise reproducible (its order is a function of n alone:
 * fixed grid, one partial per workgroup into `partials` -- a caller workspace of ilvlm_sumsq_partials() floats --, added up in
 * index order by one workgroup), so data-parallel replicas that clip the same averaged gradients stay bit-identical, as
 * torch's clip_grad_norm_ keeps them. */
int ilvlm_sumsq_partials(void);
int ilvlm_sumsq(const float* x, long n, float* out, float* partials, void* stream);
int ilvlm_clip_by_norm(float* x, long n, const float* sumsq, float max_norm, void* stream);

/* ---- fused multi-tensor AdamW (torch.optim.AdamW semantics, optimizer/__init__.py:3,18-26).
 * The parameters live in one flat fp32 arena; `chunk_*` arrays (device) describe n_chunks pieces:
 * element offset (a multiple of 4: the kernel moves 16 bytes per lane), element count and param-group id of each; lr/wd
 * are per group (<= 16 groups).
 * active[group] == 0 skips the group entirely (parameters without gradient are not touched).
 * shadow (optional, bf16) receives the updated parameters. step is 1-based. */
typedef struct ilvlm_adamw_hyper {
    float lr[16];
    float weight_decay[16];
    int active[16];
    float beta1, beta2, eps;
    int step;
} ilvlm_adamw_hyper;
int ilvlm_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                     const int64_t* chunk_offset, const int32_t* chunk_count, const int32_t* chunk_group,
                     int n_chunks, const ilvlm_adamw_hyper* hyper, void* stream);
/* The same update over the GEMM weights that the streaming kernel reads in MFMA-fragment order, tile by tile (64 x 64), writing
 * -- besides parameters, moments and the row-major bf16 shadow -- the tile's part of both fragment-order images
 * (ilvlm_pack_weights' fwd / bwd buffers): the optimizer keeps the packed copies current itself and the per-step re-pack launch
 * disappears.  tile_table: n_tiles x 6 int32 {arena offset / 64, rows, cols, r0, c0, param group}; such weights must NOT also
 * appear in the chunk table of ilvlm_adamw_step.  A tile whose group is inactive is left untouched (images included). */
int ilvlm_adamw_step_packed(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                            void* packed_fwd, void* packed_bwd, const int32_t* tile_table, int n_tiles,
                            const ilvlm_adamw_hyper* hyper, void* stream);

/* ---- composite: one residual attention block per call (image_encoder/base_transformer.py:29-62, text twin
 * text_encoder/base_transformer.py:29-59): x_mid = x + out_proj(attn(in_proj(ln_1 x))), x_out = x_mid +
 * c_proj(QuickGELU(c_fc(ln_2 x_mid))).  The calls only sequence the kernels above, in the order the host engine issues
 * them one by one; they exist to cut host time (48 calls instead of ~800 per step).  The caller owns all memory:
 * `saved` (ilvlm_block_saved_bytes) receives the activations the backward needs, `scratch` (ilvlm_block_scratch_bytes)
 * holds the backward temporaries and must stay untouched until the weight-gradient stream has been joined.
 * dtype: ILVLM_BF16 (GEMM weights = bf16 shadow, activations bf16, fp32 residual stream) or ILVLM_F32.
 * Gradient pointers accumulate (+=); NULL marks a frozen weight / bias (LayerNorm parameters must be trainable).
 * seq_offs != NULL: packed text rows (below), rows = total valid tokens; else rows = B * L.
 * wgrad_stream (nullable): weight-gradient GEMMs are issued there, ordered after their operands by events. */
typedef struct ilvlm_block {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    const void *in_w, *out_w, *fc_w, *proj_w;      /* [3E,E] [E,E] [4E,E] [E,4E], compute dtype */
    const float *in_b, *out_b, *fc_b, *proj_b;
    float *g_ln1_w, *g_ln1_b, *g_ln2_w, *g_ln2_b, *g_in_w, *g_in_b, *g_out_w, *g_out_b, *g_fc_w, *g_fc_b, *g_proj_w,
        *g_proj_b;
    int E, H, causal, dtype;
    /* fp8 mode (BASELINE configs[4]); fp8 = 0 and NULL pointers otherwise.  fp8 = 1: bf16 GEMMs, only the amaxes of the
     * tensors that would be quantised are recorded (first step: no scale history yet); fp8 = 2: the four forward GEMMs take
     * e4m3 activations x e4m3 weights and the four input-gradient GEMMs e5m2 gradients x transposed e4m3 weights, weight
     * gradients stay bf16; fp8 = 3: the weight gradients too (e5m2 gradient copies x the e4m3 activation copies the forward
     * call left in `saved`; forward and backward must agree on fp8 >= 2).  f8_scale / f8_inv / f8_amax: 12 floats each, in the order h1, att, h2, g (GEMM inputs),
     * in_w, out_w, fc_w, proj_w, d(x_out), du, d(x_mid), dqkv (gradients). */
    const void *in_w8, *out_w8, *fc_w8, *proj_w8;       /* e4m3, the weights' own [out, in] layout */
    const void *in_w8t, *out_w8t, *fc_w8t, *proj_w8t;   /* e4m3, transposed [in, out] */
    const float* f8_scale;
    const float* f8_inv;
    float* f8_amax;
    int fp8;
    /* slab split-K workspace for the block's weight-gradient GEMMs (see ilvlm_gemm_epilogue.splitk_*); NULL = atomics.  One
     * workspace per weight-gradient stream. */
    void* splitk_ws;
    long splitk_ws_bytes;
    int32_t* splitk_cnt;
    int splitk_cnt_len;
    /* optional fragment-order copies of the four GEMM weights (ilvlm_pack_weights), bf16 mode: *_wp = forward image (the
     * b_packed operand of x W^T), *_wpt = input-gradient image (of dY W).  With them the eight store-type GEMMs of the block
     * run the streaming kernel (ilvlm_gemm_epilogue.b_packed); NULL = the direct-to-LDS kernel on the row-major weights. */
    const void *in_wp, *out_wp, *fc_wp, *proj_wp;
    const void *in_wpt, *out_wpt, *fc_wpt, *proj_wpt;
    /* the same for fp8 >= 2: fragment-order copies of *_w8 / *_w8t (ilvlm_fp8_quantize_weights_packed); NULL = direct-to-LDS */
    const void *in_w8p, *out_w8p, *fc_w8p, *proj_w8p;
    const void *in_w8tp, *out_w8tp, *fc_w8tp, *proj_w8tp;
} ilvlm_block;
long ilvlm_block_saved_bytes(const ilvlm_block* b, long rows, int B, int L);
long ilvlm_block_scratch_bytes(const ilvlm_block* b, long rows);
int ilvlm_block_fwd(const ilvlm_block* b, const float* x_in, float* x_out, void* saved, long rows, int B, int L, int Lcap,
                    const int32_t* seq_offs, void* stream);
/* dx_f32 / dx_lp: gradient of the block output (fp32, and its compute-dtype copy in bf16 mode); din_*: the same for the
 * block input.  ln_ws: 2 * ln_ws_blocks * E floats (LayerNorm second stage); ln_ws_blocks < 0 defers the second stage of
 * both LayerNorms (ilvlm_layernorm_bwd_reduce_batched): ln_ws then holds two slots of 2 * |ln_ws_blocks| * E floats, ln_2's
 * partials first, ln_1's second.  wgrad_target: workgroups a split-K weight-gradient launch should reach (384 fills the
 * chip). */
int ilvlm_block_bwd(const ilvlm_block* b, const float* x_in, const void* saved, const float* dx_f32, const void* dx_lp,
                    float* din_f32, void* din_lp, void* scratch, float* ln_ws, int ln_ws_blocks, long rows, int B, int L,
                    int Lcap, const int32_t* seq_offs, int wgrad_target, void* stream, void* wgrad_stream, const void* dx8,
                    void* din8, const float* din8_scale, float* din8_amax);

/* ---- composite: a whole tower per call.  The caller allocates, for the n_blocks blocks of a tower (all of one width E):
 *   xs      [n_blocks, rows, E] fp32: xs[i] = output of block i (the input of block i is xs[i - 1], x0 for block 0);
 *   saved   n_blocks x saved_stride bytes (>= ilvlm_block_saved_bytes of every block);
 *   d_f32   [n_blocks, rows, E] fp32: d_f32[i] receives the gradient of block i's INPUT; dtop_* is the gradient of the last
 *           block's output; per_block[i].din_lp / din8: where the compute-dtype / e5m2 copies of d_f32[i] go (NULL: not produced,
 *           exactly as the din_lp / din8 arguments of ilvlm_block_bwd), with the e5m2 copy's scale and amax slots;
 *           per_block[i].ln_ws / ln_ws_blocks: as in ilvlm_block_bwd;
 *   scratch n_blocks x scratch_stride bytes, untouched until the weight-gradient stream has been joined.
 * The functions walk the blocks (forward 0 .. n-1, backward n-1 .. 0) through ilvlm_block_fwd / ilvlm_block_bwd: the same
 * kernels in the same order on the same streams, one call per tower instead of one per block (host time).  `done` (nullable)
 * runs on the calling thread after block i's backward has been enqueued -- the hook of the data-parallel gradient reducer
 * (reference prototype/utils/torch_ddp_dist.py:52-67 starts a bucket's all-reduce when its gradients are ready). */
typedef struct ilvlm_tower_grad {
    void* din_lp;
    void* din8;
    const float* din8_scale;
    float* din8_amax;
    float* ln_ws;
    int ln_ws_blocks;
    int pad_;
} ilvlm_tower_grad;
typedef void (*ilvlm_block_done_fn)(int block, void* user);
int ilvlm_tower_fwd(const ilvlm_block* blocks, int n_blocks, const float* x0, float* xs, void* saved, long saved_stride, long rows,
                    int B, int L, int Lcap, const int32_t* seq_offs, void* stream);
int ilvlm_tower_bwd(const ilvlm_block* blocks, int n_blocks, const ilvlm_tower_grad* per_block, const float* x0, const float* xs,
                    const void* saved, long saved_stride, const float* dtop_f32, const void* dtop_lp, float* d_f32, void* scratch,
                    long scratch_stride, long rows, int B, int L, int Lcap, const int32_t* seq_offs, int wgrad_target, void* stream,
                    void* wgrad_stream, ilvlm_block_done_fn done, void* user);
/* fp8 mode (b->fp8 >= 2), all four nullable.  With b->fp8 == 3 and all four weight gradients requested, dx_lp may be NULL
 * when dx8 is given and din_lp may be NULL when din8 is given (every consumer then reads the e5m2 copy): dx8 = e5m2 copy of dx_lp if the producer already made one (the previous call's
 * din8), else the call quantises dx_lp itself; din8 [rows, E] bytes receives the e5m2 copy of din_lp, quantised with
 * din8_scale[0] (the d(x_out) slot of the block that will consume it), din8_amax[0] raised to max|din_lp|. */

/* ---- packed text rows.  Positions behind <|endoftext|> never reach the loss: attention is causal
 * (text_transformer.py:147-153), the FDT scores of masked tokens are multiplied by zero (clip_fdt.py:118-123) and the
 * pooled feature is read at the EOT position (text_transformer.py:248).  The training step therefore may run the text
 * tower on the valid tokens only: sequence b owns rows [seq_offs[b], seq_offs[b+1]) of every [rows, W] text tensor
 * (seq_offs: int32[B+1] on the device, seq_offs[0] = 0, lengths 1..L).  `tokens` keeps the reference's [B][L] layout.
 * These entry points are the packed forms of the functions above; results on the valid rows are the same. */
int ilvlm_embed_packed_fwd(const int64_t* tokens, const int32_t* seq_offs, const float* table, const float* pos, float* x,
                           int B, int L, int W, int vocab, void* stream);
int ilvlm_embed_packed_bwd(const int64_t* tokens, const int32_t* seq_offs, const float* dx, float* dtable, float* dpos,
                           int B, int L, int W, int vocab, void* stream);
/* L = row stride of lse ([B][H][L]) = context length; Lcap = longest sequence of this batch (<= 128 bf16, 80 f32) */
int ilvlm_attention_packed_fwd(const void* qkv, void* out, float* lse, int dtype, int B, int L, int Lcap, int H, int causal,
                               const int32_t* seq_offs, void* stream);
int ilvlm_attention_packed_bwd(const void* dout, const void* qkv, const void* out, const float* lse, void* dqkv, int dtype,
                               int B, int L, int Lcap, int H, int causal, const int32_t* seq_offs, void* stream);
/* fp8 mode: the same kernels also emit the fp8 copy their consumer GEMM reads and raise q_amax[0] to max|value| (the copy
 * quantises the bf16-rounded value with q_scale[0]; out8 / dqkv8 NULL = observe the amax only).  Forward: out8 [rows, E]
 * e4m3 of `out`.  Backward: dqkv8 [rows, 3E] e5m2 of `dqkv`; dqkv itself may be NULL when every consumer reads the copy
 * (every sequence length the bf16 kernels take, up to 288 tokens).  bf16 only.  seq_offs NULL = dense rows (Lcap ignored). */
int ilvlm_attention_fwd_q8(const void* qkv, void* out, float* lse, int dtype, int B, int L, int Lcap, int H, int causal,
                           const int32_t* seq_offs, void* out8, const float* q_scale, float* q_amax, void* stream);
int ilvlm_attention_bwd_q8(const void* dout, const void* qkv, const void* out, const float* lse, void* dqkv, int dtype, int B,
                           int L, int Lcap, int H, int causal, const int32_t* seq_offs, void* dqkv8, const float* q_scale,
                           float* q_amax, void* stream);
/* idx[b] = position inside sequence b (EOT pooling, text_transformer.py:248) */
int ilvlm_gather_packed_rows(const float* x, const int64_t* idx, const int32_t* seq_offs, float* y, int B, int W, void* stream);
int ilvlm_scatter_packed_rows(const float* dy, const int64_t* idx, const int32_t* seq_offs, float* dx, int B, int W,
                              void* stream);
/* scores [rows][C] on packed rows; the T - len masked positions count as the zeros the dense form gives them */
int ilvlm_fdt_pool_packed_fwd(const float* scores, const int32_t* seq_offs, float* pooled, int* argmax, int B, int T, int C,
                              float sqrt_d, float temperature, int pool, void* stream);
int ilvlm_fdt_pool_packed_bwd(const float* dpooled, const int* argmax, const int32_t* seq_offs, void* dscores, int dtype,
                              int B, int T, int C, float sqrt_d, float temperature, int pool, void* stream);

/* ---- host-side byte-level BPE tokenizer (no GPU work).  Replaces SimpleTokenizer.encode / bpe
 * (prototype/model/utils/text_utils/simple_tokenizer.py:63-135) and TextTransformer.tokenize framing
 * (text_encoder/text_transformer.py:155-202) that the reference runs in Python inside every forward().
 * create: `merges_text` is the DECOMPRESSED bpe_simple_vocab_16e6.txt (header line + one merge per line); the
 *         handle is thread-safe (calls serialise on an internal mutex) and owns no device memory.
 * encode: texts[n] NUL-terminated UTF-8 -> tokens[n][ctx] (int64: <|startoftext|> ids <|endoftext|> 0...; over-long
 *         captions keep sot + the first ctx-2 ids + eot), pad_mask[n][ctx] (0 valid / -inf pad), lengths[n].
 *         Captions with bytes outside printable ASCII / ASCII white space, or with '&' (HTML entities), need the
 *         reference's Unicode cleaning: they are reported in fallback[i] = 1 with their rows untouched, for the caller's
 *         full-Unicode tokenizer. */
int ilvlm_tokenizer_create(const char* merges_text, long nbytes, void** handle);
int ilvlm_tokenizer_encode(void* handle, const char* const* texts, int n, int context_length, long long* tokens,
                           float* pad_mask, int* lengths, unsigned char* fallback);
int ilvlm_tokenizer_destroy(void* handle);

/* ---- self tests of the MFMA / LDS-transpose fragment maps (used by tests only) ---- */
int ilvlm_selftest_fragments(float* out /* [5][64][8] */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ILVLM_HIP_H */
