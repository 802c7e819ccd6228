"""Pins tests/augment_ref.py -- the restatement the device augmentation kernels are compared with -- against PIL, the
library torchvision's transforms call on PIL images (torchvision itself is not installed here; its F_pil functions are thin:
resized_crop = img.crop(box).resize(size, BILINEAR); adjust_brightness / contrast / saturation = ImageEnhance.*.enhance(f);
adjust_hue = the 8-bit HSV round trip with a uint8 wrap; RandomGrayscale = convert("L") replicated; the reference's own
GaussianBlur = ImageFilter.GaussianBlur(radius=sigma), prototype/data/transforms.py:82-91; hflip = FLIP_LEFT_RIGHT).
EVERY stage must agree EXACTLY: the resize is PIL's 8-bit fixed-point resampling with a uint8 image between its two passes, the
colour operations PIL's integer / float32 arithmetic, the blur the three box-blur passes per direction PIL approximates the
Gaussian with; ToTensor / Normalize are float32 divisions as torchvision's tensor ops."""
import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance, ImageFilter  # noqa: E402

import augment_ref as R  # noqa: E402


class P:      # the fields of ilvlm_augment_params that cpu_augment reads
    def __init__(self, **kw):
        self.crop_top = self.crop_left = 0
        self.crop_h = self.crop_w = 0
        self.jitter, self.jitter_order = 0, 0 | (1 << 2) | (2 << 4) | (3 << 6)
        self.brightness = self.contrast = self.saturation = 1.0
        self.hue, self.grayscale, self.blur_sigma, self.flip = 0.0, 0, 0.0, 0
        self.__dict__.update(kw)


def _img(H, W, seed):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.stack([128 + 100 * np.sin(xx / 17.0 + c) * np.cos(yy / 23.0 - c) for c in range(3)], -1)
    return np.clip(base + rng.randn(H, W, 3) * 25, 0, 255).astype(np.uint8)


def _denorm(t):
    """normalised CHW float -> HWC 0..255 (the restatement's last step undone)"""
    a = t.astype(np.float64) * np.array(R.STD).reshape(3, 1, 1) + np.array(R.MEAN).reshape(3, 1, 1)
    return np.rint(a.transpose(1, 2, 0) * 255.0)


def _hue_pil(img, hue):
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h = (np_h.astype(np.int64) + int(hue * 255)).astype(np.uint8)      # np.uint8 arithmetic with wrap
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


@pytest.mark.parametrize("H,W,box", [(300, 400, (20, 30, 200, 260)), (64, 48, (4, 2, 40, 40)), (900, 1100, (10, 50, 880, 1000)),
                                     (224, 224, (0, 0, 224, 224)), (500, 375, (0, 0, 500, 375)), (333, 500, (100, 7, 37, 411)),
                                     (700, 700, (1, 1, 223, 225))])
def test_resized_crop_matches_pil_exactly(H, W, box):
    im = _img(H, W, 1)
    top, left, h, w = box
    want = np.asarray(Image.fromarray(im).crop((left, top, left + w, top + h)).resize((224, 224), Image.BILINEAR)).astype(np.float64)
    got = _denorm(R.cpu_augment(im, P(crop_top=top, crop_left=left, crop_h=h, crop_w=w), 224))
    assert np.array_equal(got, want), np.abs(got - want).max()


def test_resize_matches_pil_exactly_on_random_sizes():
    rng = np.random.RandomState(7)
    for it in range(25):
        h, w = int(rng.randint(8, 640)), int(rng.randint(8, 640))
        im = rng.randint(0, 256, (h, w, 3)).astype(np.uint8) if it % 2 else _img(h, w, it)
        out = 224 if it % 5 else 96
        want = np.asarray(Image.fromarray(im).resize((out, out), Image.BILINEAR))
        assert np.array_equal(R.resize_pil(im, out), want), (h, w, out)


@pytest.mark.parametrize("op,factor", [(0, 0.6), (0, 1.4), (1, 0.61), (1, 1.39), (2, 0.7), (2, 1.33), (3, -0.1), (3, 0.07), (3, 0.0)])
def test_colour_operations_match_pil_exactly(op, factor):
    im = _img(224, 224, 2)
    pil = Image.fromarray(im)
    if op == 0:
        want = ImageEnhance.Brightness(pil).enhance(factor)
    elif op == 1:
        want = ImageEnhance.Contrast(pil).enhance(factor)
    elif op == 2:
        want = ImageEnhance.Color(pil).enhance(factor)
    else:
        want = _hue_pil(pil, factor)
    p = P(crop_h=224, crop_w=224, jitter=1, jitter_order=op | (op << 2) * 0)
    # apply ONLY this operation: the other three slots get neutral factors (hue 0 is not neutral in 8-bit HSV: order them last
    # and compare with PIL applying the same sequence)
    neutral = {0: "brightness", 1: "contrast", 2: "saturation"}
    order = [op] + [k for k in (0, 1, 2) if k != op]
    if op != 3:
        order = [op] + [k for k in (0, 1, 2) if k != op]
        setattr(p, neutral[op], factor)
        p.jitter_order = order[0] | (order[1] << 2) | (order[2] << 4) | (3 << 6)
        want = _hue_pil(want, 0.0)                      # the fourth slot: hue with factor 0, as torchvision would still run it
    else:
        p.hue = factor
        p.jitter_order = 3 | (0 << 2) | (1 << 4) | (2 << 6)
    got = _denorm(R.cpu_augment(im, p, 224))
    assert np.array_equal(got, np.asarray(want).astype(np.float64)), np.abs(got - np.asarray(want)).max()


def test_full_jitter_chain_grayscale_and_flip_match_pil_exactly():
    im = _img(224, 224, 3)
    p = P(crop_h=224, crop_w=224, jitter=1, jitter_order=2 | (0 << 2) | (3 << 4) | (1 << 6), brightness=1.27, contrast=0.71,
          saturation=1.18, hue=-0.06, grayscale=1, flip=1)
    pil = Image.fromarray(im)
    pil = ImageEnhance.Color(pil).enhance(p.saturation)
    pil = ImageEnhance.Brightness(pil).enhance(p.brightness)
    pil = _hue_pil(pil, p.hue)
    pil = ImageEnhance.Contrast(pil).enhance(p.contrast)
    pil = pil.convert("L").convert("RGB")               # RandomGrayscale(num_output_channels = 3)
    pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
    got = _denorm(R.cpu_augment(im, p, 224))
    assert np.array_equal(got, np.asarray(pil).astype(np.float64))


@pytest.mark.parametrize("sigma", [0.1, 0.2887, 0.29, 0.7, 1.0, 1.5, 2.0, 0.4567, 1.9321])
def test_gaussian_blur_matches_pil_exactly(sigma):
    """ImageFilter.GaussianBlur(radius=sigma) (prototype/data/transforms.py:82-91): PIL's box radius from sigma, three box passes per
    direction with an 8-bit image after each"""
    im = _img(224, 224, 4)
    want = np.asarray(Image.fromarray(im).filter(ImageFilter.GaussianBlur(radius=sigma))).astype(np.float64)
    got = _denorm(R.cpu_augment(im, P(crop_h=224, crop_w=224, blur_sigma=sigma), 224))
    assert np.array_equal(got, want), np.abs(got - want).max()


def test_whole_chain_matches_the_pil_pipeline_exactly_and_normalises_in_float32():
    """crop + resize, jitter chain, blur, flip against the same PIL calls in MOCOV2_single's order; the final tensor against
    torch's float32 ToTensor / Normalize arithmetic"""
    import torch
    im = _img(480, 640, 9)
    p = P(crop_top=31, crop_left=77, crop_h=300, crop_w=411, jitter=1, jitter_order=1 | (3 << 2) | (0 << 4) | (2 << 6),
          brightness=0.83, contrast=1.21, saturation=0.66, hue=0.04, grayscale=0, blur_sigma=1.37, flip=1)
    pil = Image.fromarray(im).crop((77, 31, 77 + 411, 31 + 300)).resize((224, 224), Image.BILINEAR)
    pil = ImageEnhance.Contrast(pil).enhance(p.contrast)
    pil = _hue_pil(pil, p.hue)
    pil = ImageEnhance.Brightness(pil).enhance(p.brightness)
    pil = ImageEnhance.Color(pil).enhance(p.saturation)
    pil = pil.filter(ImageFilter.GaussianBlur(radius=p.blur_sigma)).transpose(Image.FLIP_LEFT_RIGHT)
    got = R.cpu_augment(im, p, 224)
    assert np.array_equal(_denorm(got), np.asarray(pil).astype(np.float64))
    t = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).float().div(255)
    t = t.sub(torch.tensor(R.MEAN).view(3, 1, 1)).div(torch.tensor(R.STD).view(3, 1, 1))
    assert np.array_equal(got, t.numpy())
