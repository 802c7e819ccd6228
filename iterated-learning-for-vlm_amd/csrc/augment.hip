// On-device tail of the reference's training augmentation: MOCOV2_single of
// prototype/data/imagenet_dataloader.py:59-68 --
//     RandomResizedCrop(224, scale=(0.2, 1.)), RandomApply([ColorJitter(0.4, 0.4, 0.4, 0.1)], p=0.8), RandomGrayscale(p=0.2),
//     RandomApply([GaussianBlur([.1, 2.])], p=0.5), RandomHorizontalFlip(), ToTensor(), Normalize
// -- from decoded uint8 images.  At 15-20 k pairs/s per GPU (x 8 GPUs per host) these per-pixel passes are what the loader
// workers cannot keep up with; shard reading and JPEG decode stay with the loader
// (prototype/data/datasets/clip_dataset_wsd.py:158-240).  The RANDOM DRAWS stay on the host too -- a dozen numbers per sample
// (ilvlm_augment_params: what torchvision's get_params return) -- the pixel work runs here, per sample:
//   1. the crop box resized to OUT x OUT: bilinear WITH the antialiasing PIL's resize applies (a triangle filter whose support
//      grows with the down-scaling factor, taps clipped to the crop box as torchvision crops first), separable: horizontal
//      pass into a float scratch, vertical pass, rounded to 0..255 as PIL's uint8 result is;
//   2. the four ColorJitter operations in the drawn order, each as torchvision's F.adjust_* defines it on a PIL image:
//      ImageEnhance blends with black / the rounded mean of the luma image / the luma image (Image.blend truncates to uint8),
//      hue through PIL's 8-bit HSV with numpy's uint8 wrap; then RandomGrayscale's luma replacement (PIL "L");
//   3. Gaussian blur with the drawn sigma (separable, radius ceil(3 sigma), taps outside the image dropped and the rest
//      renormalised), horizontal flip, / 255, (x - mean) / std, written as fp32 NCHW.
// One workgroup per image for 2-3: the contrast step needs the image's mean luma (a workgroup reduction) and the passes meet
// at workgroup barriers, the 600 KB working image staying in L2.
// Differences from PIL that remain, stated rather than hidden: PIL resamples in 8-bit fixed point with a uint8 image between
// the two passes, and ImageFilter.GaussianBlur approximates the Gaussian by repeated box blurs; here both are fp32 with one
// rounding.  tests/test_input_pipeline_gpu.py restates exactly the arithmetic above on the CPU (fp32) and bounds the
// difference; the distribution of augmented images is the reference's.
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

// PIL's precompute_coeffs for the bilinear (triangle) filter: taps [lo, hi) of output index o, in coordinates of the crop
struct Taps { int lo, hi; float center, ss, norm; };
__device__ __forceinline__ Taps taps_of(int o, int in_size, int out_size) {
    const float scale = (float)in_size / (float)out_size;
    const float fs = fmaxf(scale, 1.f);                 // antialias when down-scaling
    Taps t;
    t.center = ((float)o + 0.5f) * scale;
    t.ss = 1.f / fs;
    t.lo = max((int)(t.center - fs + 0.5f), 0);
    t.hi = min((int)(t.center + fs + 0.5f), in_size);
    float n = 0.f;
    for (int x = t.lo; x < t.hi; ++x) n += fmaxf(0.f, 1.f - fabsf(((float)x - t.center + 0.5f) * t.ss));
    t.norm = n > 0.f ? 1.f / n : 0.f;
    return t;
}
__device__ __forceinline__ float tap_w(const Taps& t, int x) { return fmaxf(0.f, 1.f - fabsf(((float)x - t.center + 0.5f) * t.ss)) * t.norm; }

// pass 1: every row of the crop box resampled horizontally: tmp[b][y][ox][c], y in [0, crop_h)
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
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int x = tp.lo; x < tp.hi; ++x) {
            const float w = tap_w(tp, x);
            const unsigned char* q = img + ((long)y * W + x) * 3;
            a0 += w * q[0]; a1 += w * q[1]; a2 += w * q[2];
        }
        float* o = t + (long)i * 3;
        o[0] = a0; o[1] = a1; o[2] = a2;
    }
}

// pass 2: vertical resample -> work[b][oy][ox][c], rounded to 0..255
__global__ __launch_bounds__(256) void aug_resize_v_kernel(const ilvlm_augment_params* __restrict__ prm, const float* __restrict__ tmp,
                                                           long tmp_stride, float* __restrict__ work, int OUT) {
    const int b = blockIdx.y;
    const ilvlm_augment_params p = prm[b];
    const float* t = tmp + (long)b * tmp_stride;
    float* w_ = work + (long)b * OUT * OUT * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < OUT * OUT; i += gridDim.x * 256) {
        const int oy = i / OUT, ox = i - oy * OUT;
        const Taps tp = taps_of(oy, p.crop_h, OUT);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int y = tp.lo; y < tp.hi; ++y) {
            const float w = tap_w(tp, y);
            const float* q = t + ((long)y * OUT + ox) * 3;
            a0 += w * q[0]; a1 += w * q[1]; a2 += w * q[2];
        }
        float* o = w_ + (long)i * 3;
        o[0] = round255(a0); o[1] = round255(a1); o[2] = round255(a2);
    }
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

// colour jitter, grayscale, blur, flip, normalise: one workgroup per image; work [OUT][OUT][3] in place, blur through tmp
__global__ __launch_bounds__(AUG_THREADS) void aug_color_kernel(const ilvlm_augment_params* __restrict__ prm, float* __restrict__ work,
                                                                float* __restrict__ tmp, long tmp_stride, float* __restrict__ dst, int OUT,
                                                                float m0, float m1, float m2, float is0, float is1, float is2) {
    __shared__ double shd[AUG_THREADS / 64];
    __shared__ float gk[32];
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
    if (p.blur_sigma > 0.f) {
        const int rad = min((int)ceilf(3.f * p.blur_sigma), 15);
        if ((int)threadIdx.x <= rad) gk[threadIdx.x] = expf(-0.5f * (float)(threadIdx.x * threadIdx.x) / (p.blur_sigma * p.blur_sigma));
        __syncthreads();
        // horizontal into tmp, vertical back into work
        for (int pass = 0; pass < 2; ++pass) {
            const float* in = pass == 0 ? w : t;
            float* out = pass == 0 ? t : w;
            for (int i = threadIdx.x; i < npix; i += AUG_THREADS) {
                const int y = i / OUT, x = i - y * OUT;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, n = 0.f;
                for (int d = -rad; d <= rad; ++d) {
                    const int xx = pass == 0 ? x + d : x, yy = pass == 0 ? y : y + d;
                    if (xx < 0 || xx >= OUT || yy < 0 || yy >= OUT) continue;
                    const float wgt = gk[d < 0 ? -d : d];
                    const float* q = in + ((long)yy * OUT + xx) * 3;
                    a0 += wgt * q[0]; a1 += wgt * q[1]; a2 += wgt * q[2]; n += wgt;
                }
                const float inv = 1.f / n;
                float* o = out + (long)i * 3;
                if (pass == 0) { o[0] = a0 * inv; o[1] = a1 * inv; o[2] = a2 * inv; }
                else { o[0] = round255(a0 * inv); o[1] = round255(a1 * inv); o[2] = round255(a2 * inv); }
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
        d[i] = (q[0] * (1.f / 255.f) - m0) * is0;
        d[npix + i] = (q[1] * (1.f / 255.f) - m1) * is1;
        d[2 * npix + i] = (q[2] * (1.f / 255.f) - m2) * is2;
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
                       mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
    ILVLM_LAUNCH_CHECK("image_augment (colour / blur / normalise)");
    return ILVLM_OK;
}
