// On-device tail of the reference's training augmentation: MOCOV2_single of
// prototype/data/imagenet_dataloader.py:59-68 --
//     RandomResizedCrop(224, scale=(0.2, 1.)), RandomApply([ColorJitter(0.4, 0.4, 0.4, 0.1)], p=0.8), RandomGrayscale(p=0.2),
//     RandomApply([GaussianBlur([.1, 2.])], p=0.5), RandomHorizontalFlip(), ToTensor(), Normalize
// -- from decoded uint8 images.  At 15-20 k pairs/s per GPU (x 8 GPUs per host) these per-pixel passes are what the loader
// workers cannot keep up with; shard reading and JPEG decode stay with the loader
// (prototype/data/datasets/clip_dataset_wsd.py:158-240).  The RANDOM DRAWS stay on the host too -- a dozen numbers per sample
// (ilvlm_augment_params: what torchvision's get_params return) -- the pixel work runs here, per sample:
//   1. the crop box resized to OUT x OUT as PIL's Image.resize(BILINEAR) does it (Resample.c; torchvision crops first, so the
//      taps are clipped to the crop box): a triangle filter whose support grows with the down-scaling factor, its weights
//      computed in double and rounded to 22-bit fixed point, a horizontal pass into an 8-bit image, then the vertical pass;
//   2. the four ColorJitter operations in the drawn order, each as torchvision's F.adjust_* defines it on a PIL image:
//      ImageEnhance blends with black / the rounded mean of the luma image / the luma image (Image.blend truncates to uint8),
//      hue through PIL's 8-bit HSV with numpy's uint8 wrap; then RandomGrayscale's luma replacement (PIL "L");
//   3. the reference's GaussianBlur = ImageFilter.GaussianBlur(radius = sigma) (prototype/data/transforms.py:82-91), i.e.
//      PIL's approximation by box blurs (BoxBlur.c): a fractional box radius from sigma, three horizontal passes and three
//      vertical ones in 32-bit fixed point with an 8-bit image after each, edges extended; horizontal flip; ToTensor's
//      float32 / 255 and Normalize's (x - mean) / std as float32 divisions, written as fp32 NCHW.
// One workgroup per image for 2-3: the contrast step needs the image's mean luma (a workgroup reduction) and the passes meet
// at workgroup barriers, the 600 KB working image staying in L2.
// Every stage repeats PIL's own arithmetic operation by operation (no fused multiply-adds where the C code has none), so
// given the same draws the output equals what the reference's loader workers produce BIT FOR BIT: tests/augment_ref.py restates
// the same arithmetic with numpy, tests/test_augment_cpu.py holds that restatement to PIL exactly (every stage and the whole
// chain), tests/test_input_pipeline_gpu.py holds these kernels to the restatement exactly.
#include "common.h"

namespace {

constexpr int AUG_THREADS = 512;

__device__ __forceinline__ float clamp255(float x) { return fminf(fmaxf(x, 0.f), 255.f); }
__device__ __forceinline__ float round255(float x) { return rintf(clamp255(x)); }
__device__ __forceinline__ float trunc255(float x) { return floorf(clamp255(x)); }           // Image.blend: (UINT8) cast
// PIL "L": (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16 on integers
__device__ __forceinline__ float luma_pil(float r, float g, float b) {
    return (float)(((int)r * 19595 + (int)g * 38470 + (int)b * 7471 + 0x8000) >> 16);
}

// PIL's precompute_coeffs for the bilinear (triangle) filter + normalize_coeffs_8bpc: taps [xmin, xmin + n) of output index o in
// coordinates of the crop, weight of tap x as 22-bit fixed point.  The double operations of Resample.c in their order, unfused
// (fp contract off: the host code PIL is compiled to has no fused multiply-add, and one flipped rounding of a weight shows).
constexpr int RS_BITS = 32 - 8 - 2;
struct Taps { int xmin, n; double center, ss, ww; };
__device__ __forceinline__ double tap_weight(const Taps& t, int x) {
#pragma clang fp contract(off)
    double w = ((double)(x + t.xmin) - t.center + 0.5) * t.ss;
    w = w < 0.0 ? -w : w;
    return w < 1.0 ? 1.0 - w : 0.0;
}
__device__ __forceinline__ Taps taps_of(int o, int in_size, int out_size) {
#pragma clang fp contract(off)
    const double scale = (double)in_size / (double)out_size;
    const double fs = scale < 1.0 ? 1.0 : scale;        // antialias when down-scaling; support = 1.0 * filterscale
    Taps t;
    t.center = ((double)o + 0.5) * scale;
    t.ss = 1.0 / fs;
    int xmin = (int)(t.center - fs + 0.5), xmax = (int)(t.center + fs + 0.5);
    xmin = xmin < 0 ? 0 : xmin;
    xmax = xmax > in_size ? in_size : xmax;
    t.xmin = xmin;
    t.n = xmax - xmin;
    double ww = 0.0;
    for (int x = 0; x < t.n; ++x) ww += tap_weight(t, x);
    t.ww = ww;
    return t;
}
__device__ __forceinline__ int tap_coef(const Taps& t, int x) {
#pragma clang fp contract(off)
    double w = tap_weight(t, x);
    if (t.ww != 0.0) w = w / t.ww;
    return (int)(0.5 + w * (double)(1 << RS_BITS));     // weights are non-negative
}
__device__ __forceinline__ float rs_clip8(int v) { return (float)min(max(v >> RS_BITS, 0), 255); }

// pass 1: every row of the crop box resampled horizontally: tmp[b][y][ox][c], y in [0, crop_h), 8-bit values (held as floats)
__global__ __launch_bounds__(256) void aug_resize_h_kernel(const unsigned char* __restrict__ src, const long* __restrict__ src_off,
                                                           const int* __restrict__ src_hw, const ilvlm_augment_params* __restrict__ prm,
                                                           float* __restrict__ tmp, long tmp_stride, int OUT) {
    const int b = blockIdx.y;
    const ilvlm_augment_params p = prm[b];
    const int W = src_hw[2 * b + 1];
    const unsigned char* img = src + src_off[b] + ((long)p.crop_top * W + p.crop_left) * 3;
    float* t = tmp + (long)b * tmp_stride;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.crop_h * OUT; i += gridDim.x * 256) {
        const int y = i / OUT, ox = i - y * OUT;
        const Taps tp = taps_of(ox, p.crop_w, OUT);
        int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
        for (int x = 0; x < tp.n; ++x) {
            const int k = tap_coef(tp, x);
            const unsigned char* q = img + ((long)y * W + tp.xmin + x) * 3;
            a0 += k * q[0]; a1 += k * q[1]; a2 += k * q[2];
        }
        float* o = t + (long)i * 3;
        o[0] = rs_clip8(a0); o[1] = rs_clip8(a1); o[2] = rs_clip8(a2);
    }
}

// pass 2: vertical resample of the 8-bit intermediate -> work[b][oy][ox][c], 8-bit values
__global__ __launch_bounds__(256) void aug_resize_v_kernel(const ilvlm_augment_params* __restrict__ prm, const float* __restrict__ tmp,
                                                           long tmp_stride, float* __restrict__ work, int OUT) {
    const int b = blockIdx.y;
    const ilvlm_augment_params p = prm[b];
    const float* t = tmp + (long)b * tmp_stride;
    float* w_ = work + (long)b * OUT * OUT * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < OUT * OUT; i += gridDim.x * 256) {
        const int oy = i / OUT, ox = i - oy * OUT;
        const Taps tp = taps_of(oy, p.crop_h, OUT);
        int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
        for (int y = 0; y < tp.n; ++y) {
            const int k = tap_coef(tp, y);
            const float* q = t + ((long)(tp.xmin + y) * OUT + ox) * 3;
            a0 += k * (int)q[0]; a1 += k * (int)q[1]; a2 += k * (int)q[2];
        }
        float* o = w_ + (long)i * 3;
        o[0] = rs_clip8(a0); o[1] = rs_clip8(a1); o[2] = rs_clip8(a2);
    }
}

// _gaussian_blur_radius of BoxBlur.c: the fractional box radius whose three passes approximate a Gaussian of this sigma
// (float variables, the square root in double, as the C code)
__device__ __forceinline__ float box_radius_of(float sigma) {
#pragma clang fp contract(off)
    const float sigma2 = sigma * sigma / 3.f;
    const float L = (float)sqrt(12.0 * (double)sigma2 + 1.0);
    const float l = (float)floor(((double)L - 1.0) / 2.0);
    float a = (2.f * l + 1.f) * (l * (l + 1.f) - 3.f * sigma2);
    a = __fdiv_rn(a, 6.f * (sigma2 - (l + 1.f) * (l + 1.f)));
    return l + a;
}

__device__ __forceinline__ double aug_block_sum_d(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < AUG_THREADS / 64; ++i) s += sh[i];
    return s;
}

// PIL Image.blend(degenerate, image, f) on 8-bit data, as ImageEnhance.*.enhance calls it (Blend.c): single precision
// d + f * (a - d) with the product and the sum rounded separately (no fused multiply-add: the host code PIL is compiled to has
// none), clipped when extrapolating, then the (UINT8) cast = truncation
__device__ __forceinline__ float blend_pil(float d, float a, float f) { return trunc255(__fadd_rn(d, __fmul_rn(f, a - d))); }

// torchvision F_pil.adjust_hue: PIL's rgb2hsv_row (8-bit H and S, V = max), h += uint8(hue * 255) with numpy's wrap, PIL's
// hsv2rgb -- with the mix of float and double arithmetic of Convert.c, which decides the 8-bit results
__device__ __forceinline__ void adjust_hue(float& r, float& g, float& b, float hue) {
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    int h8 = 0, s8 = 0;
    if (mx != mn) {
        const float cr = mx - mn;
        const float s = __fdiv_rn(cr, mx);
        const float rc = __fdiv_rn(mx - r, cr), gc = __fdiv_rn(mx - g, cr), bc = __fdiv_rn(mx - b, cr);
        float h;
        if (r == mx) h = __fsub_rn(bc, gc);
        else if (g == mx) h = (float)(2.0 + (double)rc - (double)bc);
        else h = (float)(4.0 + (double)gc - (double)rc);
        h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
        h8 = min(max((int)((double)h * 255.0), 0), 255);
        s8 = min(max((int)((double)s * 255.0), 0), 255);
    }
    h8 = (h8 + (int)(hue * 255.f)) & 255;                 // np.uint8 addition wraps
    if (s8 == 0) { r = g = b = mx; return; }
    const double hd = (double)(float)h8 * 6.0 / 255.0;
    const double fi = floor(hd);
    const double f = (double)(float)(hd - fi), fs = (double)(float)((double)(float)s8 / 255.0), v = (double)mx;
    // C round(): half away from zero; the arguments are non-negative
    const float pq = (float)fmin(fmax(floor(v * (1.0 - fs) + 0.5), 0.0), 255.0);
    const float qq = (float)fmin(fmax(floor(v * (1.0 - fs * f) + 0.5), 0.0), 255.0);
    const float tq = (float)fmin(fmax(floor(v * (1.0 - fs * (1.0 - f)) + 0.5), 0.0), 255.0);
    switch (((int)fi) % 6) {
        case 0: r = mx; g = tq; b = pq; break;
        case 1: r = qq; g = mx; b = pq; break;
        case 2: r = pq; g = mx; b = tq; break;
        case 3: r = pq; g = qq; b = mx; break;
        case 4: r = tq; g = pq; b = mx; break;
        default: r = mx; g = pq; b = qq; break;
    }
}

// colour jitter, grayscale, blur, flip, normalise: one workgroup per image; work [OUT][OUT][3] in place, blur passes alternate with tmp
__global__ __launch_bounds__(AUG_THREADS) void aug_color_kernel(const ilvlm_augment_params* __restrict__ prm, float* __restrict__ work,
                                                                float* __restrict__ tmp, long tmp_stride, float* __restrict__ dst, int OUT,
                                                                float m0, float m1, float m2, float sd0, float sd1, float sd2) {
    __shared__ double shd[AUG_THREADS / 64];
    const int b = blockIdx.x, npix = OUT * OUT;
    const ilvlm_augment_params p = prm[b];
    float* w = work + (long)b * npix * 3;
    float* t = tmp + (long)b * tmp_stride;
    if (p.jitter) {
        for (int k = 0; k < 4; ++k) {
            const int op = (p.jitter_order >> (2 * k)) & 3;      // 0 brightness, 1 contrast, 2 saturation, 3 hue
            float mean = 0.f;
            if (op == 1) {                                        // ImageEnhance.Contrast: int(mean of the L image + 0.5)
                // (per-thread sums stay below 2^24, where fp32 adds of integers are exact; the cross-thread sum is double)
                float s = 0.f;
                for (int i = threadIdx.x; i < npix; i += AUG_THREADS) s += luma_pil(w[3 * i], w[3 * i + 1], w[3 * i + 2]);
                mean = (float)floor(aug_block_sum_d((double)s, shd) / (double)npix + 0.5);
            }
            for (int i = threadIdx.x; i < npix; i += AUG_THREADS) {
                float r = w[3 * i], g = w[3 * i + 1], bl = w[3 * i + 2];
                if (op == 0) {                                    // ImageEnhance.Brightness: blend(black, image, f)
                    r = blend_pil(0.f, r, p.brightness); g = blend_pil(0.f, g, p.brightness); bl = blend_pil(0.f, bl, p.brightness);
                } else if (op == 1) {                             // ImageEnhance.Contrast: blend(mean, image, f)
                    r = blend_pil(mean, r, p.contrast); g = blend_pil(mean, g, p.contrast); bl = blend_pil(mean, bl, p.contrast);
                } else if (op == 2) {                             // ImageEnhance.Color: blend(luma image, image, f)
                    const float l = luma_pil(r, g, bl);
                    r = blend_pil(l, r, p.saturation); g = blend_pil(l, g, p.saturation); bl = blend_pil(l, bl, p.saturation);
                } else {
                    adjust_hue(r, g, bl, p.hue);
                }
                w[3 * i] = r; w[3 * i + 1] = g; w[3 * i + 2] = bl;
            }
            __syncthreads();
        }
    }
    if (p.grayscale) {
        for (int i = threadIdx.x; i < npix; i += AUG_THREADS) {
            const float l = luma_pil(w[3 * i], w[3 * i + 1], w[3 * i + 2]);
            w[3 * i] = l; w[3 * i + 1] = l; w[3 * i + 2] = l;
        }
    }
    __threadfence_block();
    __syncthreads();
    const float fr = p.blur_sigma > 0.f ? box_radius_of(p.blur_sigma) : 0.f;
    if (fr != 0.f) {
        // ImagingBoxBlur: three passes along x, then three along y; a pass = ImagingLineBoxBlur: the 2 r + 1 window with weight ww,
        // the two pixels just beyond it with fw, edge pixels repeated, 32-bit unsigned arithmetic, (bulk + 2^23) >> 24
        const int r = (int)fr;
        const unsigned ww = (unsigned)__fdiv_rn(16777216.f, __fadd_rn(__fmul_rn(fr, 2.f), 1.f));
        const unsigned fw = ((1u << 24) - (unsigned)(r * 2 + 1) * ww) / 2u;
        for (int pass = 0; pass < 6; ++pass) {
            const float* in = (pass & 1) ? t : w;         // w -> t -> w -> t -> w -> t -> w: the sixth pass ends in `work`
            float* out = (pass & 1) ? w : t;
            const bool along_x = pass < 3;
            for (int i = threadIdx.x; i < npix; i += AUG_THREADS) {
                const int y = i / OUT, x = i - y * OUT;
                const int pos = along_x ? x : y;
                const long base = along_x ? (long)y * OUT * 3 : (long)x * 3, step = along_x ? 3 : (long)OUT * 3;
                unsigned a0 = 0, a1 = 0, a2 = 0;
                for (int d = -r; d <= r; ++d) {
                    const float* q = in + base + (long)min(max(pos + d, 0), OUT - 1) * step;
                    a0 += (unsigned)q[0]; a1 += (unsigned)q[1]; a2 += (unsigned)q[2];
                }
                const float* ql = in + base + (long)max(pos - r - 1, 0) * step;
                const float* qr = in + base + (long)min(pos + r + 1, OUT - 1) * step;
                float* o = out + (long)i * 3;
                o[0] = (float)((a0 * ww + ((unsigned)ql[0] + (unsigned)qr[0]) * fw + (1u << 23)) >> 24);
                o[1] = (float)((a1 * ww + ((unsigned)ql[1] + (unsigned)qr[1]) * fw + (1u << 23)) >> 24);
                o[2] = (float)((a2 * ww + ((unsigned)ql[2] + (unsigned)qr[2]) * fw + (1u << 23)) >> 24);
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    float* d = dst + (long)b * 3 * npix;
    for (int i = threadIdx.x; i < npix; i += AUG_THREADS) {
        const int y = i / OUT, x = i - y * OUT;
        const int xs = p.flip ? OUT - 1 - x : x;
        const float* q = w + ((long)y * OUT + xs) * 3;
        // ToTensor: float32 value / 255; Normalize: (x - mean) / std -- float32 divisions, as torchvision's tensor ops
        d[i] = __fdiv_rn(__fsub_rn(__fdiv_rn(q[0], 255.f), m0), sd0);
        d[npix + i] = __fdiv_rn(__fsub_rn(__fdiv_rn(q[1], 255.f), m1), sd1);
        d[2 * npix + i] = __fdiv_rn(__fsub_rn(__fdiv_rn(q[2], 255.f), m2), sd2);
    }
}

}  // namespace

extern "C" long ilvlm_image_augment_scratch_floats(int B, int out_size, int max_crop_rows) {
    if (B <= 0 || out_size <= 0 || max_crop_rows <= 0) return -1;
    const long rows = max_crop_rows > out_size ? max_crop_rows : out_size;      // the blur reuses the row scratch
    return (long)B * (rows * out_size * 3 + (long)out_size * out_size * 3);
}

extern "C" int ilvlm_image_augment(const unsigned char* src, const long* src_offsets, const int32_t* src_hw,
                                   const ilvlm_augment_params* params, float* dst, float* scratch, int B, int out_size,
                                   int max_crop_rows, const float* mean3, const float* std3, void* stream) {
    ILVLM_REQUIRE(src && src_offsets && src_hw && params && dst && scratch && mean3 && std3, "image_augment: null pointer");
    ILVLM_REQUIRE(B > 0 && out_size > 0 && out_size <= 1024 && max_crop_rows > 0, "image_augment: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const long rows = max_crop_rows > out_size ? max_crop_rows : out_size;
    const long tmp_stride = rows * out_size * 3;
    float* tmp = scratch;                                                       // [B][rows][OUT][3]
    float* work = scratch + (long)B * tmp_stride;                               // [B][OUT][OUT][3]
    hipLaunchKernelGGL(aug_resize_h_kernel, dim3(64, B), dim3(256), 0, s, src, src_offsets, src_hw, params, tmp, tmp_stride, out_size);
    ILVLM_LAUNCH_CHECK("image_augment (horizontal resample)");
    hipLaunchKernelGGL(aug_resize_v_kernel, dim3(32, B), dim3(256), 0, s, params, tmp, tmp_stride, work, out_size);
    ILVLM_LAUNCH_CHECK("image_augment (vertical resample)");
    hipLaunchKernelGGL(aug_color_kernel, dim3(B), dim3(AUG_THREADS), 0, s, params, work, tmp, tmp_stride, dst, out_size, mean3[0], mean3[1],
                       mean3[2], std3[0], std3[1], std3[2]);
    ILVLM_LAUNCH_CHECK("image_augment (colour / blur / normalise)");
    return ILVLM_OK;
}
