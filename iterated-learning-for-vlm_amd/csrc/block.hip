// Composite entry points: one residual attention block (reference image_encoder/base_transformer.py:29-62, text twin
// text_encoder/base_transformer.py:29-59) forward or backward per C call.  They only sequence the kernels of this
// library (LayerNorm, GEMM, attention) exactly as the Python engine does one by one; the point is host time: a step
// issues ~600 launches, and at ~15-20 us of interpreter + ctypes work each the host needs 12-16 ms per step against 19 ms
// of GPU work.  24 blocks x (13 forward + 21 backward) launches become 48 calls.
#include <hip/hip_runtime.h>
#include <math.h>

#include "common.h"

namespace {

inline size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
inline int esz(int dtype) { return dtype == ILVLM_BF16 ? 2 : 4; }

// fp8 == 3 with all four weight matrices trainable: forward, input-gradient and weight-gradient GEMMs all read fp8 copies,
// and the bf16 tensors nobody else reads are not written (a frozen weight keeps the fp8 == 2 behaviour for the whole block:
// its bias gradient is a column sum over the bf16 gradient)
inline bool fp8_all(const ilvlm_block* b) {
    return b->fp8 == 3 && b->g_in_w && b->g_out_w && b->g_fc_w && b->g_proj_w;
}

struct Saved {   // byte offsets into the saved-activation workspace of one block
    size_t x_mid, h1, qkv, att, h2, u, g, mean1, rstd1, mean2, rstd2, lse, h1_8, att8, h2_8, g8, total;
    Saved(const ilvlm_block* b, long rows, int B, int L) {
        const size_t E = b->E, es = esz(b->dtype), r = rows;
        size_t o = 0;
        x_mid = o; o += al(r * E * 4);
        const bool lp_copies = !fp8_all(b);      // fully-fp8 blocks keep h1, h2 and g only as their e4m3 copies
        h1 = o; if (lp_copies) o += al(r * E * es);
        qkv = o; o += al(r * 3 * E * es);
        att = o; o += al(r * E * es);
        h2 = o; if (lp_copies) o += al(r * E * es);
        u = o; o += al(r * 4 * E * es);
        g = o; if (lp_copies) o += al(r * 4 * E * es);
        mean1 = o; o += al(r * 4);
        rstd1 = o; o += al(r * 4);
        mean2 = o; o += al(r * 4);
        rstd2 = o; o += al(r * 4);
        lse = o; o += al((size_t)B * b->H * L * 4);
        // e4m3 copies of the four GEMM inputs; each keeps its own region because the backward pass reads them again as the
        // X operand of the fp8 weight gradients (7 E bytes per row)
        h1_8 = att8 = h2_8 = g8 = o;
        if (b->fp8 >= 2) {
            h1_8 = o; o += al(r * E);
            att8 = o; o += al(r * E);
            h2_8 = o; o += al(r * E);
            g8 = o; o += al(r * 4 * E);
        }
        total = o;
    }
};

struct Scratch {   // backward temporaries
    size_t du, dh2, dmid, dmid_lp, da, dqkv, dh1, dout8, du8, dmid8, dqkv8, total;
    Scratch(const ilvlm_block* b, long rows) {
        const size_t E = b->E, es = esz(b->dtype), r = rows;
        size_t o = 0;
        const bool lp_copies = !fp8_all(b);      // fully-fp8 blocks: du, the bf16 copy of d(x_mid) and dqkv exist in e5m2 only
        du = o; if (lp_copies) o += al(r * 4 * E * es);
        dh2 = o; o += al(r * E * es);
        dmid = o; o += al(r * E * 4);
        dmid_lp = o; if (lp_copies) o += al(r * E * es);
        da = o; o += al(r * E * es);
        dqkv = o; o += al(r * 3 * E * es);      // (kept: sequences above 128 tokens take the bf16 attention backward)
        dh1 = o; o += al(r * E * es);
        // e5m2 copies of the four gradients: separate regions, the weight-gradient stream reads each one while the main
        // stream is already producing the next (9 E bytes per row)
        dout8 = du8 = dmid8 = dqkv8 = o;
        if (b->fp8 >= 2) {
            dout8 = o; o += al(r * E);
            du8 = o; o += al(r * 4 * E);
            dmid8 = o; o += al(r * E);
            dqkv8 = o; o += al(r * 3 * E);
        }
        total = o;
    }
};

int check_block(const ilvlm_block* b, const char* who) {
    ILVLM_REQUIRE(b, "%s: null block descriptor", who);
    ILVLM_REQUIRE(b->dtype == ILVLM_BF16 || b->dtype == ILVLM_F32, "%s: bad dtype %d", who, b->dtype);
    ILVLM_REQUIRE(b->E > 0 && b->H > 0 && b->E == 64 * b->H, "%s: width %d must be 64 x heads (%d)", who, b->E, b->H);
    ILVLM_REQUIRE(b->ln1_w && b->ln1_b && b->ln2_w && b->ln2_b && b->in_w && b->in_b && b->out_w && b->out_b && b->fc_w &&
                      b->fc_b && b->proj_w && b->proj_b, "%s: null parameter pointer", who);
    ILVLM_REQUIRE(b->fp8 >= 0 && b->fp8 <= 3, "%s: bad fp8 mode %d", who, b->fp8);
    ILVLM_REQUIRE(!b->fp8 || b->dtype == ILVLM_BF16, "%s: fp8 mode needs bf16 storage", who);
    ILVLM_REQUIRE(b->fp8 < 2 || (b->in_w8 && b->out_w8 && b->fc_w8 && b->proj_w8 && b->in_w8t && b->out_w8t && b->fc_w8t &&
                                  b->proj_w8t && b->E % 128 == 0), "%s: fp8 mode needs the fp8 weights and E %% 128 == 0", who);
    return ILVLM_OK;
}

// split-K of a weight-gradient GEMM: enough workgroups to fill the chip (ops.wgrad_split)
int wgrad_split(long out_rows, long out_cols, long k, int tile, int target) {
    const long tiles = ((out_rows + tile - 1) / tile) * ((out_cols + tile - 1) / tile);
    long s = (long)nearbyint((double)target / (double)tiles);      // round-half-even, as Python's round()
    s = s < 1 ? 1 : (s > 16 ? 16 : s);
    const long cap = k >= 256 ? k / 256 : 1;
    s = s < cap ? s : cap;
    return (int)(s < 1 ? 1 : s);
}

#define TRY(call)            \
    do {                     \
        int rc__ = (call);   \
        if (rc__) return rc__; \
    } while (0)

enum { F8_H1 = 0, F8_ATT, F8_H2, F8_G, F8_IN_W, F8_OUT_W, F8_FC_W, F8_PROJ_W, F8_DOUT, F8_DU, F8_DMID, F8_DQKV };

// fp8 copy of x (n elements) for slot `slot`; fp8 == 1 only records the amax.  Returns the quantised buffer or null.
int f8_quant(const ilvlm_block* b, const void* x, long n, int slot, int e5m2, void* q8, hipStream_t s, const void** out) {
    *out = nullptr;
    if (!b->fp8) return ILVLM_OK;
    ILVLM_REQUIRE(b->f8_scale && b->f8_inv && b->f8_amax, "block: fp8 mode without scale arrays");
    if (b->fp8 == 1) return ilvlm_fp8_quantize(x, ILVLM_BF16, nullptr, n, nullptr, b->f8_amax + slot, e5m2, s);
    int rc = ilvlm_fp8_quantize(x, ILVLM_BF16, q8, n, b->f8_scale + slot, b->f8_amax + slot, e5m2, s);
    if (rc == ILVLM_OK) *out = q8;
    return rc;
}

// forward linear: y = x W^T (+ epilogue); fp8 operands when x8 is given
int linear_fwd(const ilvlm_block* b, const void* x, const void* x8, int slot_a, const void* W, const void* W8, int slot_w,
               void* y, long M, int N, int K, ilvlm_gemm_epilogue ep, hipStream_t s, const void* Wp = nullptr, const void* W8p = nullptr) {
    ep.b_packed = (!x8 && b->dtype == ILVLM_BF16) ? Wp : nullptr;      // fragment-order weights: the streaming kernel
    if (x8) {
        ep.b_packed = W8p;
        ep.alpha_ptr = b->f8_inv + slot_a;
        ep.alpha_ptr2 = b->f8_inv + slot_w;
        return ilvlm_gemm(ILVLM_FP8, 0, 0, (int)M, N, K, x8, K, W8, K, y, N, &ep, 1, s);
    }
    return ilvlm_gemm(b->dtype, 0, 0, (int)M, N, K, x, K, W, K, y, N, &ep, 1, s);
}

// Weight gradients of one block collected for ONE grouped launch (ilvlm_wgrad_group) instead of four split-K launches:
// linear_bwd appends its product here when `defer` is given and the product qualifies; flush_wgrad launches the list on
// the weight-gradient stream behind an event on the input-gradient stream.  ILVLM_WGRAD_GROUP=0 switches it off.
struct WgradBatch {
    ilvlm_wgrad_problem p[ILVLM_WGRAD_GROUP_MAX];
    int n = 0;
    int dtype = -1;
    long rows = 0;
};

// wg waits for everything enqueued on s so far.  The events come from a small per-thread ring that is never destroyed: an
// event recorded while a stream is being captured into a hipGraph must outlive the capture (destroying it right after the
// wait, as this code did, crashed hipStreamEndCapture), and a ring also saves a create / destroy pair per call.  Re-recording
// an event that an earlier wait still refers to is safe: a wait binds to the record that preceded it.
int order_after(hipStream_t s, hipStream_t wg) {
    constexpr int RING = 64;
    static thread_local hipEvent_t ring[RING];
    static thread_local int filled = 0, next = 0;
    if (filled < RING) {
        hipError_t e = hipEventCreateWithFlags(&ring[filled], hipEventDisableTiming);
        if (e != hipSuccess) ILVLM_FAIL((int)e, "block_bwd: hipEventCreate: %s", hipGetErrorString(e));
        next = filled++;
    } else {
        next = (next + 1) % RING;
    }
    hipError_t e = hipEventRecord(ring[next], s);
    if (e == hipSuccess) e = hipStreamWaitEvent(wg, ring[next], 0);
    if (e != hipSuccess) ILVLM_FAIL((int)e, "block_bwd: stream ordering: %s", hipGetErrorString(e));
    return ILVLM_OK;
}

// set by ilvlm_tower_bwd around the LAST block it processes (block 0): nothing of the tower's chain is left to run beside that
// block's weight gradients, so they are cut into K-slices to fill the chip instead of running one slice per tile (ilvlm_wgrad_group:
// a negative slot target = spread)
static thread_local int t_tail_block = 0;

int flush_wgrad(WgradBatch* wb, hipStream_t s, hipStream_t wg) {
    if (!wb || wb->n == 0) return ILVLM_OK;
    static const int slots_env = getenv("ILVLM_WGRAD_GROUP_SLOTS") ? atoi(getenv("ILVLM_WGRAD_GROUP_SLOTS")) : 512;
    static const int tail_spread = getenv("ILVLM_WGRAD_TAIL_SPREAD") ? atoi(getenv("ILVLM_WGRAD_TAIL_SPREAD")) : 1;
    const int slots = (t_tail_block && tail_spread) ? -slots_env : slots_env;
    hipStream_t ws = s;
    if (wg && wg != s) {
        TRY(order_after(s, wg));
        ws = wg;
    }
    const int n = wb->n;
    wb->n = 0;
    return ilvlm_wgrad_group(wb->dtype, wb->p, n, wb->rows, slots, ws);
}

// dy [M,N], x [M,K], W [N,K] (compute dtype): accumulates dW (and db) on wg (or s when wg is null), writes dx on s.
// dy8 / W8T / inv_*: the input gradient on fp8 operands (e5m2 dy, transposed e4m3 weight) when dy8 is given; x8 / inv_x: the
// weight gradient on fp8 operands too (e5m2 dy^T, the e4m3 activation copy the forward pass kept), bias gradient as its row sums.
int linear_bwd(const ilvlm_block* b, int dtype, const void* dy, const void* x, const void* W, float* gW, float* gb, void* dx, long M, int N, int K,
               int dx_act, const void* dx_aux, int wgrad_target, hipStream_t s, hipStream_t wg, const void* dy8 = nullptr,
               const void* W8T = nullptr, const float* inv_g = nullptr, const float* inv_w = nullptr, void* dx8 = nullptr,
               const float* dx8_scale = nullptr, float* dx8_amax = nullptr, const void* x8 = nullptr,
               const float* inv_x = nullptr, const void* Wpt = nullptr, WgradBatch* defer = nullptr, const void* W8Tp = nullptr) {
    const bool fuse_b = gb && gW && dtype == ILVLM_BF16 && N % 8 == 0 && N >= 8;
    const bool f8w = dy8 && x8;
    const int wdt = f8w ? ILVLM_FP8_BF8A : dtype;
    if (defer && gW && dtype == ILVLM_BF16 && (fuse_b || !gb) && N % (f8w ? 16 : 8) == 0 && K % (f8w ? 16 : 8) == 0 &&
        defer->n < ILVLM_WGRAD_GROUP_MAX && (defer->n == 0 || (defer->dtype == wdt && defer->rows == M))) {
        ilvlm_wgrad_problem& q = defer->p[defer->n++];
        q.dy = f8w ? dy8 : dy;
        q.x = f8w ? x8 : x;
        q.gw = gW;
        q.gb = gb;
        q.n = N;
        q.k = K;
        q.inv_g = f8w ? inv_g : nullptr;
        q.inv_x = f8w ? inv_x : nullptr;
        defer->dtype = wdt;
        defer->rows = M;
    } else if (gW || gb) {
        hipStream_t ws = s;
        if (wg && wg != s) {     // weight gradients leave the dgrad chain (engine._linear_bwd)
            TRY(order_after(s, wg));
            ws = wg;
        }
        if (gW) {
            ilvlm_gemm_epilogue ep = {};
            ep.alpha = 1.0f;
            ep.out_dtype = ILVLM_F32;
            ep.accumulate = 1;
            ep.a_rowsum = fuse_b ? gb : nullptr;
            ep.splitk_ws = b->splitk_ws;
            ep.splitk_ws_bytes = b->splitk_ws_bytes;
            ep.splitk_cnt = b->splitk_cnt;
            ep.splitk_cnt_len = b->splitk_cnt_len;
            const int split = wgrad_split(N, K, M, dtype == ILVLM_BF16 ? 128 : 64, wgrad_target);
            if (dy8 && x8) {
                ep.alpha_ptr = inv_g;
                ep.alpha_ptr2 = inv_x;
                TRY(ilvlm_gemm(ILVLM_FP8_BF8A, 1, 1, N, K, (int)M, dy8, N, x8, K, gW, K, &ep, split, ws));
            } else {
                TRY(ilvlm_gemm(dtype, 1, 1, N, K, (int)M, dy, N, x, K, gW, K, &ep, split, ws));
            }
        }
        if (gb && !fuse_b) TRY(ilvlm_colsum(dy, dtype, gb, M, N, N, ws));
    }
    ilvlm_gemm_epilogue ep = {};
    ep.alpha = 1.0f;
    ep.out_dtype = dtype;
    ep.act = dx_act;
    ep.aux = (void*)dx_aux;
    ep.out8 = dx8;                  // e5m2 copy (and amax) of dx for the next input-gradient GEMM
    ep.out8_scale = dx8_scale;
    ep.out8_amax = dx8_amax;
    ep.out8_fmt = 1;
    if (dy8) {      // dx[m,k] = sum_n dy8[m,n] W8T[k,n]
        ep.alpha_ptr = inv_g;
        ep.alpha_ptr2 = inv_w;
        ep.b_packed = W8Tp;
        return ilvlm_gemm(ILVLM_FP8_BF8A, 0, 0, (int)M, K, N, dy8, N, W8T, N, dx, K, &ep, 1, s);
    }
    ep.b_packed = dtype == ILVLM_BF16 ? Wpt : nullptr;     // fragment-order image of W for dx = dy W
    return ilvlm_gemm(dtype, 0, 1, (int)M, K, N, dy, N, W, K, dx, K, &ep, 1, s);
}

}  // namespace

extern "C" long ilvlm_block_saved_bytes(const ilvlm_block* b, long rows, int B, int L) {
    if (!b || rows <= 0 || B <= 0 || L <= 0) return -1;
    return (long)Saved(b, rows, B, L).total;
}
extern "C" long ilvlm_block_scratch_bytes(const ilvlm_block* b, long rows) {
    if (!b || rows <= 0) return -1;
    return (long)Scratch(b, rows).total;
}

extern "C" int ilvlm_block_fwd(const ilvlm_block* b, const float* x_in, float* x_out, void* saved, long rows, int B, int L,
                               int Lcap, const int32_t* seq_offs, void* stream) {
    TRY(check_block(b, "block_fwd"));
    ILVLM_REQUIRE(x_in && x_out && saved && rows > 0 && B > 0 && L > 0, "block_fwd: bad arguments");
    ILVLM_REQUIRE(seq_offs || rows == (long)B * L, "block_fwd: rows=%ld is not B*L=%ld (dense layout)", rows, (long)B * L);
    const Saved o(b, rows, B, L);
    char* w = (char*)saved;
    const int E = b->E, T = b->dtype;
    float *mean1 = (float*)(w + o.mean1), *rstd1 = (float*)(w + o.rstd1), *mean2 = (float*)(w + o.mean2),
          *rstd2 = (float*)(w + o.rstd2), *lse = (float*)(w + o.lse), *x_mid = (float*)(w + o.x_mid);
    void *h1 = w + o.h1, *qkv = w + o.qkv, *att = w + o.att, *h2 = w + o.h2, *u = w + o.u, *g = w + o.g;
    // x_mid = x_in + out_proj(attn(in_proj(ln_1(x_in))))
    // fp8 mode: the fp8 copy of each GEMM input is emitted by its producer where that is a kernel of this library with the
    // values in registers (LayerNorm, the fc GEMM's QuickGELU epilogue); the attention output takes a quantise pass
    hipStream_t s = (hipStream_t)stream;
    // fp8 == 3 (weight gradients on fp8 operands too): every consumer of ln_1 / ln_2's output and of the activation g reads
    // the e4m3 copy, so the bf16 tensors are not written at all
    const bool f8on = b->fp8 >= 2, f8obs = b->fp8 != 0, f8only = fp8_all(b);
    void *h1_8 = w + o.h1_8, *att8 = w + o.att8, *h2_8 = w + o.h2_8, *g8 = w + o.g8;
    const float* sc = b->f8_scale;
    float* am = b->f8_amax;
    const void* x8;
    if (f8only) h1 = h2 = g = nullptr;
    TRY(ilvlm_layernorm_fwd_q8(x_in, ILVLM_F32, b->ln1_w, b->ln1_b, h1, T, mean1, rstd1, rows, E, 1e-5f, 0, 0, f8on ? h1_8 : nullptr,
                               f8on ? sc + F8_H1 : nullptr, f8obs ? am + F8_H1 : nullptr, stream));
    ilvlm_gemm_epilogue ep = {};
    ep.alpha = 1.0f;
    ep.out_dtype = T;
    ep.bias = b->in_b;
    TRY(linear_fwd(b, h1, f8on ? h1_8 : nullptr, F8_H1, b->in_w, b->in_w8, F8_IN_W, qkv, rows, 3 * E, E, ep, s, b->in_wp, b->in_w8p));
    if (f8on) {                  // the attention kernel emits the e4m3 copy of its output itself
        TRY(ilvlm_attention_fwd_q8(qkv, att, lse, T, B, L, Lcap, b->H, b->causal, seq_offs, att8, sc + F8_ATT, am + F8_ATT, stream));
        x8 = att8;
    } else {
        if (seq_offs) TRY(ilvlm_attention_packed_fwd(qkv, att, lse, T, B, L, Lcap, b->H, b->causal, seq_offs, stream));
        else TRY(ilvlm_attention_fwd(qkv, att, lse, T, B, L, b->H, b->causal, stream));
        TRY(f8_quant(b, att, rows * E, F8_ATT, 0, att8, s, &x8));
    }
    ep = {};
    ep.alpha = 1.0f;
    ep.out_dtype = ILVLM_F32;
    ep.bias = b->out_b;
    ep.residual = x_in;
    TRY(linear_fwd(b, att, x8, F8_ATT, b->out_w, b->out_w8, F8_OUT_W, x_mid, rows, E, E, ep, s, b->out_wp, b->out_w8p));
    // x_out = x_mid + c_proj(quickgelu(c_fc(ln_2(x_mid))))
    TRY(ilvlm_layernorm_fwd_q8(x_mid, ILVLM_F32, b->ln2_w, b->ln2_b, h2, T, mean2, rstd2, rows, E, 1e-5f, 0, 0, f8on ? h2_8 : nullptr,
                               f8on ? sc + F8_H2 : nullptr, f8obs ? am + F8_H2 : nullptr, stream));
    ep = {};
    ep.alpha = 1.0f;
    ep.out_dtype = T;
    ep.bias = b->fc_b;
    ep.aux = u;
    ep.act = ILVLM_ACT_QUICKGELU;
    if (f8obs) {                                    // the activation's fp8 copy (and amax) straight from the epilogue
        ep.out8 = f8on ? g8 : nullptr;
        ep.out8_scale = f8on ? sc + F8_G : nullptr;
        ep.out8_amax = am + F8_G;
        ep.out8_fmt = 0;
    }
    TRY(linear_fwd(b, h2, f8on ? h2_8 : nullptr, F8_H2, b->fc_w, b->fc_w8, F8_FC_W, g, rows, 4 * E, E, ep, s, b->fc_wp, b->fc_w8p));
    ep = {};
    ep.alpha = 1.0f;
    ep.out_dtype = ILVLM_F32;
    ep.bias = b->proj_b;
    ep.residual = x_mid;
    return linear_fwd(b, g, f8on ? g8 : nullptr, F8_G, b->proj_w, b->proj_w8, F8_PROJ_W, x_out, rows, E, 4 * E, ep, s, b->proj_wp, b->proj_w8p);
}

extern "C" int ilvlm_block_bwd(const ilvlm_block* b, const float* x_in, const void* saved, const float* dx_f32,
                               const void* dx_lp, float* din_f32, void* din_lp, void* scratch, float* ln_ws,
                               int ln_ws_blocks, long rows, int B, int L, int Lcap, const int32_t* seq_offs,
                               int wgrad_target, void* stream, void* wgrad_stream, const void* dx8, void* din8,
                               const float* din8_scale, float* din8_amax) {
    TRY(check_block(b, "block_bwd"));
    ILVLM_REQUIRE(x_in && saved && dx_f32 && din_f32 && scratch && rows > 0 && B > 0 && L > 0, "block_bwd: bad arguments");
    const int E = b->E, T = b->dtype;
    const bool lp = T != ILVLM_F32;
    ILVLM_REQUIRE(!lp || ((dx_lp || (dx8 && fp8_all(b))) && (din_lp || (din8 && fp8_all(b)))),
                  "block_bwd: the low-precision gradient copies are required in bf16 mode (fp8 == 3: or their e5m2 copies)");
    ILVLM_REQUIRE(wgrad_target > 0, "block_bwd: wgrad_target must be positive");
    const Saved o(b, rows, B, L);
    const Scratch c(b, rows);
    const char* w = (const char*)saved;
    char* t = (char*)scratch;
    hipStream_t s = (hipStream_t)stream, wg = (hipStream_t)wgrad_stream;
    const float *mean1 = (const float*)(w + o.mean1), *rstd1 = (const float*)(w + o.rstd1),
                *mean2 = (const float*)(w + o.mean2), *rstd2 = (const float*)(w + o.rstd2),
                *lse = (const float*)(w + o.lse), *x_mid = (const float*)(w + o.x_mid);
    const void *h1 = w + o.h1, *qkv = w + o.qkv, *att = w + o.att, *h2 = w + o.h2, *u = w + o.u, *g = w + o.g;
    void *du = t + c.du, *dh2 = t + c.dh2, *da = t + c.da, *dqkv = t + c.dqkv, *dh1 = t + c.dh1;
    const int Lq = seq_offs ? Lcap : L;            // longest sequence of the launch
    float* dmid = (float*)(t + c.dmid);
    void* dmid_lp = lp ? (void*)(t + c.dmid_lp) : nullptr;
    // deferred LayerNorm reductions: slot 0 = ln_2, slot 1 = ln_1 (each 2 * |ln_ws_blocks| * E floats)
    float* ln_ws1 = ln_ws_blocks < 0 ? ln_ws + 2L * (-ln_ws_blocks) * E : ln_ws;
    // MLP
    const void* dy = lp ? dx_lp : (const void*)dx_f32;
    // fp8 mode: e5m2 copies of the gradients entering the four input-gradient GEMMs.  d(x_out) arrives from the caller when
    // the previous call (the next block) emitted it (din8); du comes out of the proj input-gradient GEMM's epilogue, d(x_mid)
    // out of ln_2's backward, dqkv takes a quantise pass; ln_1's backward emits the next call's d(x_out) into din8.
    // fp8 == 3: the four weight gradients take fp8 operands as well (the e5m2 gradient copies x the e4m3 activation copies
    // kept by the forward pass)
    const bool f8on = b->fp8 >= 2, f8obs = b->fp8 != 0, f8wg = fp8_all(b);
    void *dout8 = t + c.dout8, *du8 = t + c.du8, *dmid8 = t + c.dmid8, *dqkv8 = t + c.dqkv8;
    const void *h1_8 = f8wg ? w + o.h1_8 : nullptr, *att8 = f8wg ? w + o.att8 : nullptr, *h2_8 = f8wg ? w + o.h2_8 : nullptr,
               *g8a = f8wg ? w + o.g8 : nullptr;
    const float* sc = b->f8_scale;
    const float* inv = b->f8_inv;
    float* am = b->f8_amax;
    const void* g8 = nullptr;
    ILVLM_REQUIRE(!(dx8 || din8) || f8on, "block_bwd: fp8 gradient copies only in active fp8 mode");
    ILVLM_REQUIRE(!din8 || din8_scale, "block_bwd: din8 needs its scale");
    if (dx8) g8 = dx8;
    else TRY(f8_quant(b, dy, rows * E, F8_DOUT, 1, dout8, s, &g8));
    // with fp8 weight gradients every consumer of du, d(x_mid)'s bf16 copy, dqkv and (when the caller takes din8) din_lp reads
    // the e5m2 copy: the bf16 tensors are not written
    if (f8wg) du = nullptr;
    // Grouped weight gradients: the four products of the block as ONE launch (ilvlm_wgrad_group) once the last dY (dqkv)
    // exists; the operands of all four stay alive in `saved` / `scratch` / the caller's dx until the caller joins the
    // weight-gradient stream.  Measured (profiles/round3/step_ab_grouped*.txt): alone, the grouped launch is 17-21 % faster
    // than the four split-K launches (ViT-B/32 block 316 -> 262 us, text block 192 -> 152 us, ViT-L/14 block 1061 -> 882 us:
    // no atomics at one K-slice, one tail instead of four).  Inside the step the bf16 form LOSES -- 16.8 -> 17.1..17.4 ms
    // (ViT-B/32 + FDT), 98.2 -> 101.9 ms (ViT-L/14), whatever the K-slices, with or without companion streams, with the
    // launch capped to one workgroup per CU -- two towers already keep every CU supplied, so tails and atomics were never
    // exposed, while one launch of 432 workgroups that live 260 us each holds the LDS of the whole chip against the
    // input-gradient chain.  The fp8 form (single-stage, 32 KiB of LDS, three workgroups per CU) WINS: 13.31 -> 12.99 ms.
    // Round 4: on the SINGLE-STAGE 256 x 128 tile (which ilvlm_wgrad_group takes for bf16 when the caller has declared several
    // GEMM streams in flight and every output has whole 256-row tiles) the grouped bf16 launch wins too -- one K-slice, a single
    // writer per tile, no atomics: 16.71 -> 16.53 ms at 512 slots, 16.2-16.3 ms at <= 160 (ViT-B/32 + FDT), 95.8 -> 91.7 ms
    // (ViT-L/14); profiles/round4/step_ab_wgrad_grouped_wide.txt.
    // Hence: ILVLM_WGRAD_GROUP unset = fp8 weight gradients, and bf16 ones under the concurrency hint at widths that are
    // multiples of 256; 1 = always; 0 = never.
    static const int group_env = getenv("ILVLM_WGRAD_GROUP") ? atoi(getenv("ILVLM_WGRAD_GROUP")) : -1;
    WgradBatch batch;
    const bool group16 = !f8wg && E % 256 == 0 && ilvlm_gemm_get_concurrent() != 0;
    WgradBatch* wb = (lp && (group_env > 0 || (group_env < 0 && (f8wg || group16)))) ? &batch : nullptr;
    // proj: du = (dy W_proj) * quickgelu'(u), with its e5m2 copy from the epilogue
    TRY(linear_bwd(b, T, dy, g, b->proj_w, b->g_proj_w, b->g_proj_b, du, rows, E, 4 * E, ILVLM_ACT_QUICKGELU_BWD, u, wgrad_target, s, wg,
                   g8, b->proj_w8t, inv + F8_DOUT, inv + F8_PROJ_W, f8on ? du8 : nullptr, f8on ? sc + F8_DU : nullptr,
                   f8obs ? am + F8_DU : nullptr, g8a, inv + F8_G, b->proj_wpt, wb, b->proj_w8tp));
    TRY(linear_bwd(b, T, du, h2, b->fc_w, b->g_fc_w, b->g_fc_b, dh2, rows, 4 * E, E, 0, nullptr, wgrad_target, s, wg,
                   f8on ? du8 : nullptr, b->fc_w8t, inv + F8_DU, inv + F8_FC_W, nullptr, nullptr, nullptr, h2_8, inv + F8_H2, b->fc_wpt, wb, b->fc_w8tp));
    ILVLM_REQUIRE(b->g_ln1_w && b->g_ln1_b && b->g_ln2_w && b->g_ln2_b, "block_bwd: frozen LayerNorm parameters are not supported");
    if (f8wg) dmid_lp = nullptr;
    TRY(ilvlm_layernorm_bwd_q8(dh2, T, x_mid, ILVLM_F32, mean2, rstd2, b->ln2_w, dx_f32, dmid, dmid_lp, T, 0, nullptr, b->g_ln2_w,
                               b->g_ln2_b, rows, E, 0, 0, ln_ws, ln_ws_blocks, f8on ? dmid8 : nullptr, f8on ? sc + F8_DMID : nullptr,
                               f8obs ? am + F8_DMID : nullptr, s));
    // attention
    dy = lp ? (const void*)dmid_lp : (const void*)dmid;
    TRY(linear_bwd(b, T, dy, att, b->out_w, b->g_out_w, b->g_out_b, da, rows, E, E, 0, nullptr, wgrad_target, s, wg,
                   f8on ? dmid8 : nullptr, b->out_w8t, inv + F8_DMID, inv + F8_OUT_W, nullptr, nullptr, nullptr, att8, inv + F8_ATT,
                   b->out_wpt, wb, b->out_w8tp));
    if (f8on) {                  // the attention backward kernels emit the e5m2 copy of dqkv themselves (all sequence lengths)
        if (f8wg) dqkv = nullptr;
        TRY(ilvlm_attention_bwd_q8(da, qkv, att, lse, dqkv, T, B, L, Lcap, b->H, b->causal, seq_offs, dqkv8, sc + F8_DQKV,
                                   am + F8_DQKV, s));
        g8 = dqkv8;
    } else {
        if (seq_offs) TRY(ilvlm_attention_packed_bwd(da, qkv, att, lse, dqkv, T, B, L, Lcap, b->H, b->causal, seq_offs, s));
        else TRY(ilvlm_attention_bwd(da, qkv, att, lse, dqkv, T, B, L, b->H, b->causal, s));
        TRY(f8_quant(b, dqkv, rows * 3 * E, F8_DQKV, 1, dqkv8, s, &g8));
    }
    // the in-projection's weight gradient joins the batch and the grouped launch leaves BEFORE its input-gradient GEMM is
    // enqueued: everything the four products read is final once the attention backward above has run
    if (wb) {
        const bool f8w = g8 && h1_8;
        const int N = 3 * E, K = E;
        if (b->g_in_w && (b->g_in_b == nullptr || N % 8 == 0) && wb->n < ILVLM_WGRAD_GROUP_MAX &&
            (wb->n == 0 || (wb->dtype == (f8w ? ILVLM_FP8_BF8A : T) && wb->rows == rows))) {
            ilvlm_wgrad_problem& q = wb->p[wb->n++];
            q.dy = f8w ? g8 : (const void*)dqkv;
            q.x = f8w ? h1_8 : h1;
            q.gw = b->g_in_w;
            q.gb = b->g_in_b;
            q.n = N;
            q.k = K;
            q.inv_g = f8w ? inv + F8_DQKV : nullptr;
            q.inv_x = f8w ? inv + F8_H1 : nullptr;
            wb->dtype = f8w ? ILVLM_FP8_BF8A : T;
            wb->rows = rows;
            TRY(flush_wgrad(wb, s, wg));
            TRY(linear_bwd(b, T, dqkv, h1, b->in_w, nullptr, nullptr, dh1, rows, 3 * E, E, 0, nullptr, wgrad_target, s, wg, g8, b->in_w8t,
                           inv + F8_DQKV, inv + F8_IN_W, nullptr, nullptr, nullptr, h1_8, inv + F8_H1, b->in_wpt, nullptr, b->in_w8tp));
        } else {
            TRY(flush_wgrad(wb, s, wg));
            TRY(linear_bwd(b, T, dqkv, h1, b->in_w, b->g_in_w, b->g_in_b, dh1, rows, 3 * E, E, 0, nullptr, wgrad_target, s, wg, g8, b->in_w8t,
                           inv + F8_DQKV, inv + F8_IN_W, nullptr, nullptr, nullptr, h1_8, inv + F8_H1, b->in_wpt, nullptr, b->in_w8tp));
        }
    } else {
        TRY(linear_bwd(b, T, dqkv, h1, b->in_w, b->g_in_w, b->g_in_b, dh1, rows, 3 * E, E, 0, nullptr, wgrad_target, s, wg, g8, b->in_w8t,
                       inv + F8_DQKV, inv + F8_IN_W, nullptr, nullptr, nullptr, h1_8, inv + F8_H1, b->in_wpt, nullptr, b->in_w8tp));
    }
    return ilvlm_layernorm_bwd_q8(dh1, T, x_in, ILVLM_F32, mean1, rstd1, b->ln1_w, dmid, din_f32,
                                  lp ? din_lp : nullptr, T, 0,
                                  nullptr, b->g_ln1_w, b->g_ln1_b, rows, E, 0, 0, ln_ws1, ln_ws_blocks, din8, din8_scale, din8_amax, s);
}


// ---- composite: a whole tower (n_blocks residual attention blocks) per call -----------------------------------------------
// The block calls above cut ~800 Python-level operations per step to 48; what remains of the host's 7-10 ms per step is the
// interpreter work AROUND each of those 48 calls (descriptor lookup, two or three allocations, argument marshalling, the
// bookkeeping of the weight-gradient stream).  Here a tower is ONE call: the caller allocates the activations, gradients and
// scratch of all blocks as a few big buffers, and these functions walk the blocks, calling the block entry points above with
// pointers into them -- the same kernels in the same order on the same streams.  `done` (nullable) is called on the host
// after block i's backward has been enqueued: the data-parallel reducer starts that block's gradient mean from it.
extern "C" int ilvlm_tower_fwd(const ilvlm_block* blocks, int n_blocks, const float* x0, float* xs, void* saved, long saved_stride,
                               long rows, int B, int L, int Lcap, const int32_t* seq_offs, void* stream) {
    ILVLM_REQUIRE(blocks && n_blocks > 0 && x0 && xs && saved && rows > 0, "tower_fwd: bad arguments");
    const int E = blocks[0].E;
    for (int i = 0; i < n_blocks; ++i) {
        ILVLM_REQUIRE(blocks[i].E == E, "tower_fwd: block %d has width %d, block 0 has %d", i, blocks[i].E, E);
        const long need = ilvlm_block_saved_bytes(&blocks[i], rows, B, L);
        ILVLM_REQUIRE(need > 0 && need <= saved_stride, "tower_fwd: block %d needs %ld bytes of saved activations, stride is %ld", i, need,
                      saved_stride);
        const float* x_in = i ? xs + (size_t)(i - 1) * rows * E : x0;
        TRY(ilvlm_block_fwd(&blocks[i], x_in, xs + (size_t)i * rows * E, (char*)saved + (size_t)i * saved_stride, rows, B, L, Lcap, seq_offs,
                            stream));
    }
    return ILVLM_OK;
}

extern "C" int ilvlm_tower_bwd(const ilvlm_block* blocks, int n_blocks, const ilvlm_tower_grad* per_block, const float* x0,
                               const float* xs, const void* saved, long saved_stride, const float* dtop_f32, const void* dtop_lp,
                               float* d_f32, void* scratch, long scratch_stride, long rows, int B, int L, int Lcap,
                               const int32_t* seq_offs, int wgrad_target, void* stream, void* wgrad_stream,
                               ilvlm_block_done_fn done, void* user) {
    ILVLM_REQUIRE(blocks && per_block && n_blocks > 0 && x0 && xs && saved && dtop_f32 && d_f32 && scratch && rows > 0,
                  "tower_bwd: bad arguments");
    const int E = blocks[0].E;
    for (int i = n_blocks - 1; i >= 0; --i) {
        ILVLM_REQUIRE(blocks[i].E == E, "tower_bwd: block %d has width %d, block 0 has %d", i, blocks[i].E, E);
        ILVLM_REQUIRE(ilvlm_block_saved_bytes(&blocks[i], rows, B, L) <= saved_stride &&
                          ilvlm_block_scratch_bytes(&blocks[i], rows) <= scratch_stride,
                      "tower_bwd: block %d does not fit the saved / scratch strides", i);
        const bool top = i == n_blocks - 1;
        const ilvlm_tower_grad& g = per_block[i];
        const float* x_in = i ? xs + (size_t)(i - 1) * rows * E : x0;
        // gradient of block i's output: the caller's for the last block, else what block i + 1 wrote for its input
        const float* dx_f32 = top ? dtop_f32 : d_f32 + (size_t)(i + 1) * rows * E;
        const void* dx_lp = top ? dtop_lp : per_block[i + 1].din_lp;
        const void* dx8 = top ? nullptr : per_block[i + 1].din8;
        t_tail_block = i == 0 && n_blocks > 1;
        const int rc_blk = ilvlm_block_bwd(&blocks[i], x_in, (const char*)saved + (size_t)i * saved_stride, dx_f32, dx_lp,
                                           d_f32 + (size_t)i * rows * E, g.din_lp, (char*)scratch + (size_t)i * scratch_stride, g.ln_ws,
                                           g.ln_ws_blocks, rows, B, L, Lcap, seq_offs, wgrad_target, stream, wgrad_stream, dx8, g.din8,
                                           g.din8_scale, g.din8_amax);
        t_tail_block = 0;
        if (rc_blk) return rc_blk;
        if (done) done(i, user);
    }
    return ILVLM_OK;
}
