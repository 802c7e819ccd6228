"""CPU restatement of the augmentation arithmetic the device kernels implement (csrc/augment.hip): MOCOV2_single of the
reference (prototype/data/imagenet_dataloader.py:59-68) after decode, in PIL's own arithmetic -- 8-bit fixed-point resampling
(Resample.c), Image.blend / "L" / HSV for the colour operations, the box-blur passes behind ImageFilter.GaussianBlur
(BoxBlur.c), then ToTensor / Normalize in float32.  Test infrastructure: tests/test_augment_cpu.py pins it against PIL (the
library torchvision's transforms call for PIL images, i.e. what the reference's loader workers run) -- EXACTLY, every stage;
tests/test_input_pipeline_gpu.py compares the kernels with it, bit for bit."""
import numpy as np

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


PRECISION_BITS = 32 - 8 - 2


def _coeffs(in_size, out_size):
    """PIL's precompute_coeffs (bilinear = triangle filter, support 1, widened by the down-scaling factor) followed by
    normalize_coeffs_8bpc: per output index the first tap and the taps' weights as 22-bit fixed point.  Every operation is the
    C double operation of Resample.c, in its order (the kernels repeat them with unfused intrinsics)."""
    scale = float(in_size) / float(out_size)
    fs = max(scale, 1.0)
    ss = 1.0 / fs
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - fs + 0.5), 0)
        xmax = min(int(center + fs + 0.5), in_size)
        k, ww = [], 0.0
        for x in range(xmax - xmin):
            w = (x + xmin - center + 0.5) * ss
            w = -w if w < 0.0 else w
            w = 1.0 - w if w < 1.0 else 0.0
            k.append(w)
            ww += w
        ki = [int(0.5 + (w / ww if ww != 0.0 else w) * (1 << PRECISION_BITS)) for w in k]       # weights are >= 0: (int)(0.5 + ...)
        out.append((xmin, np.array(ki, dtype=np.int64)))
    return out


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255)


def resize_pil(img, out):
    """Image.resize((out, out), BILINEAR) of a uint8 [h, w, 3] image: horizontal pass into a uint8 image, then vertical"""
    a = img.astype(np.int64)
    half = 1 << (PRECISION_BITS - 1)
    tmp = np.empty((a.shape[0], out, 3), dtype=np.int64)
    for ox, (xmin, k) in enumerate(_coeffs(a.shape[1], out)):
        tmp[:, ox] = _clip8(half + np.tensordot(a[:, xmin:xmin + len(k)], k, axes=(1, 0)))
    res = np.empty((out, out, 3), dtype=np.int64)
    for oy, (ymin, k) in enumerate(_coeffs(a.shape[0], out)):
        res[oy] = _clip8(half + np.tensordot(k, tmp[ymin:ymin + len(k)], axes=(0, 0)))
    return res


def box_radius(sigma, passes=3):
    """_gaussian_blur_radius of BoxBlur.c (float variables, the square root in double)"""
    f = np.float32
    sigma = f(sigma)
    sigma2 = f(sigma * sigma / f(passes))
    L = f(np.sqrt(12.0 * np.float64(sigma2) + 1.0))
    l = f(np.floor((np.float64(L) - 1.0) / 2.0))
    a = f(f(f(2) * l + f(1)) * f(f(l * f(l + f(1))) - f(f(3) * sigma2)))
    a = f(a / f(f(6) * f(sigma2 - f(f(l + f(1)) * f(l + f(1))))))
    return f(l + a)


def _box_blur(a, fr):
    """ImagingLineBoxBlur along axis 0: the 2 r + 1 window with weight ww, the two pixels beyond it with fw, edges extended,
    32-bit unsigned arithmetic, (bulk + 2^23) >> 24"""
    n, r = a.shape[0], int(fr)
    ww = int(np.float32(16777216.0) / np.float32(np.float32(fr) * np.float32(2) + np.float32(1)))
    fw = (((1 << 24) - (r * 2 + 1) * ww) & 0xffffffff) // 2
    idx = np.arange(n)
    acc = sum(a[np.clip(idx + d, 0, n - 1)] for d in range(-r, r + 1))
    far = a[np.clip(idx - r - 1, 0, n - 1)] + a[np.clip(idx + r + 1, 0, n - 1)]
    return ((((acc * ww + far * fw) & 0xffffffff) + (1 << 23)) & 0xffffffff) >> 24


def gaussian_blur_pil(a, sigma, passes=3):
    """ImageFilter.GaussianBlur(radius=sigma): three horizontal box blurs, then three vertical (ImagingBoxBlur)"""
    fr = box_radius(sigma, passes)
    a = a.astype(np.int64)
    if fr != 0:
        a = a.transpose(1, 0, 2)
        for _ in range(passes):
            a = _box_blur(a, fr)
        a = a.transpose(1, 0, 2)
        for _ in range(passes):
            a = _box_blur(a, fr)
    return a


F32 = np.float32


def _luma(a):
    """PIL "L": (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16"""
    a = a.astype(np.int64)
    return ((a[..., 0] * 19595 + a[..., 1] * 38470 + a[..., 2] * 7471 + 0x8000) >> 16).astype(np.float64)


def _blend(d, a, f):
    """PIL Image.blend(degenerate, image, f) on uint8 data, as ImageEnhance.*.enhance calls it: single-precision
    d + f * (a - d), clipped when extrapolating, then the (UINT8) cast, i.e. truncation"""
    f = F32(f)
    t = (d.astype(F32) + (f * (a.astype(F32) - d.astype(F32))).astype(F32)).astype(F32)
    return np.floor(np.clip(t, 0, 255)).astype(np.float64)


def _rgb2hsv(a):
    """PIL's rgb2hsv_row (Convert.c), with its mix of float and double arithmetic: 8-bit H and S, V = max"""
    r, g, b = [a[..., i].astype(np.int64) for i in range(3)]
    mx, mn = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = np.where(mx != mn, mx - mn, 1).astype(F32)
    s = (cr / np.where(mx > 0, mx, 1).astype(F32)).astype(F32)
    rc, gc, bc = [((mx - c).astype(F32) / cr).astype(F32) for c in (r, g, b)]
    d = np.float64
    h = np.where(r == mx, (bc - gc).astype(F32),
                 np.where(g == mx, (2.0 + rc.astype(d) - bc.astype(d)).astype(F32), (4.0 + gc.astype(d) - rc.astype(d)).astype(F32)))
    h = np.fmod(h.astype(d) / 6.0 + 1.0, 1.0).astype(F32)
    uh = np.clip((h.astype(d) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(d) * 255.0).astype(np.int64), 0, 255)
    flat = mx == mn
    return np.where(flat, 0, uh), np.where(flat, 0, us), mx


def _hsv2rgb(h, s, v):
    """PIL's hsv2rgb (Convert.c)"""
    d = np.float64
    hd = h.astype(F32).astype(d) * 6.0 / 255.0
    i = np.floor(hd)
    f = (hd - i).astype(F32).astype(d)
    fs = (s.astype(F32).astype(d) / 255.0).astype(F32).astype(d)
    vf = v.astype(d)
    rnd = lambda x: np.clip(np.floor(x + 0.5), 0, 255)           # C round() on non-negative values
    p, q, t = rnd(vf * (1.0 - fs)), rnd(vf * (1.0 - fs * f)), rnd(vf * (1.0 - fs * (1.0 - f)))
    k = i.astype(np.int64) % 6
    R = np.choose(k, [vf, q, p, p, t, vf]); G = np.choose(k, [t, vf, vf, q, p, p]); B = np.choose(k, [p, p, t, vf, vf, q])
    return np.where((s == 0)[..., None], vf[..., None].repeat(3, -1), np.stack([R, G, B], -1))


def _hue(a, hue):
    """torchvision F_pil.adjust_hue: 8-bit HSV, h += uint8(hue * 255) with numpy's wrap, back to RGB"""
    h, s, v = _rgb2hsv(a)
    return _hsv2rgb((h + int(hue * 255.0)) & 255, s, v)


def cpu_augment(img, p, OUT, mean=MEAN, std=STD):
    crop = img[p.crop_top:p.crop_top + p.crop_h, p.crop_left:p.crop_left + p.crop_w]
    a = resize_pil(crop, OUT).astype(np.float64)
    if p.jitter:
        for k in range(4):
            op = (p.jitter_order >> (2 * k)) & 3
            if op == 0:                                   # ImageEnhance.Brightness: blend with black
                a = _blend(np.zeros_like(a), a, p.brightness)
            elif op == 1:                                 # ImageEnhance.Contrast: blend with int(mean of the L image + 0.5)
                a = _blend(np.full_like(a, np.floor(_luma(a).mean() + 0.5)), a, p.contrast)
            elif op == 2:                                 # ImageEnhance.Color: blend with the L image
                a = _blend(_luma(a)[..., None].repeat(3, -1), a, p.saturation)
            else:
                a = _hue(a, p.hue)
    if p.grayscale:
        a = _luma(a)[..., None].repeat(3, -1)
    if p.blur_sigma > 0:
        a = gaussian_blur_pil(a, p.blur_sigma).astype(np.float64)
    if p.flip:
        a = a[:, ::-1]
    # ToTensor: uint8 -> float32 / 255; Normalize: (x - mean) / std, all float32 (torchvision's tensor ops)
    t = (a.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)).astype(np.float32)
    m, sd = np.array(mean, dtype=np.float32).reshape(3, 1, 1), np.array(std, dtype=np.float32).reshape(3, 1, 1)
    return ((t - m).astype(np.float32) / sd).astype(np.float32)
