"""Float CPU restatement of the augmentation arithmetic the device kernels implement (csrc/augment.hip): MOCOV2_single of the
reference (prototype/data/imagenet_dataloader.py:59-68) after decode.  Test infrastructure: tests/test_augment_cpu.py pins it
against PIL (the library torchvision's transforms call for PIL images, i.e. what the reference's loader workers run);
tests/test_input_pipeline_gpu.py compares the kernels with it."""
import numpy as np

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


# truncation, PIL's "L" luma and 8-bit HSV, a separable Gaussian with renormalised borders, flip, ToTensor, Normalize
def _coeffs(in_size, out_size):
    scale = in_size / out_size
    fs = max(scale, 1.0)
    out = []
    for o in range(out_size):
        center = (o + 0.5) * scale
        lo = max(int(center - fs + 0.5), 0)
        hi = min(int(center + fs + 0.5), in_size)
        w = np.maximum(0.0, 1.0 - np.abs((np.arange(lo, hi) - center + 0.5) / fs))
        out.append((lo, hi, w / w.sum()))
    return out


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
    crop = img[p.crop_top:p.crop_top + p.crop_h, p.crop_left:p.crop_left + p.crop_w].astype(np.float64)
    tmp = np.stack([np.tensordot(w, crop[:, lo:hi], axes=(0, 1)) for lo, hi, w in _coeffs(p.crop_w, OUT)], 1)       # [h, OUT, 3]
    a = np.stack([np.tensordot(w, tmp[lo:hi], axes=(0, 0)) for lo, hi, w in _coeffs(p.crop_h, OUT)], 0)              # [OUT, OUT, 3]
    a = np.rint(np.clip(a, 0, 255))
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
        rad = min(int(np.ceil(3.0 * np.float32(p.blur_sigma))), 15)
        g = np.exp(-0.5 * np.arange(-rad, rad + 1) ** 2 / np.float64(np.float32(p.blur_sigma)) ** 2)
        for axis in (1, 0):
            acc, nrm = np.zeros_like(a), np.zeros(a.shape[:2] + (1,))
            for d, wgt in zip(range(-rad, rad + 1), g):
                sl_dst = [slice(None)] * 3
                sl_src = [slice(None)] * 3
                sl_dst[axis] = slice(max(0, -d), OUT - max(0, d))
                sl_src[axis] = slice(max(0, d), OUT - max(0, -d))
                acc[tuple(sl_dst)] += wgt * a[tuple(sl_src)]
                nrm[tuple(sl_dst[:2]) + (slice(None),)] += wgt
            a = acc / nrm
        a = np.rint(np.clip(a, 0, 255))
    if p.flip:
        a = a[:, ::-1]
    t = a.transpose(2, 0, 1) / 255.0
    return ((t - np.array(mean).reshape(3, 1, 1)) / np.array(std).reshape(3, 1, 1)).astype(np.float32)
