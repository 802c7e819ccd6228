"""Input pipeline on the device: uint8 H2D + ToTensor + Normalize (+ the flip / grayscale coin results of
MOCOV2_single) against the float CPU pipeline the reference runs in its loader workers
(prototype/data/imagenet_dataloader.py:13-14, 59-68: ..., RandomGrayscale, ..., RandomHorizontalFlip, ToTensor, Normalize)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def cpu_pipeline(u8_nhwc, flags):
    """what torchvision does per sample, restated on tensors: PIL 'L' conversion for grayscale (integer luma, 3 equal
    channels), horizontal flip, ToTensor (HWC uint8 -> CHW float / 255), Normalize"""
    out = []
    for img, f in zip(u8_nhwc, flags):
        a = img.to(torch.int64)
        if f & 2:
            l = (a[..., 0] * 19595 + a[..., 1] * 38470 + a[..., 2] * 7471 + 0x8000) >> 16
            a = torch.stack([l, l, l], -1)
        if f & 1:
            a = torch.flip(a, dims=[1])
        t = a.permute(2, 0, 1).to(torch.float32) / 255.0
        t = (t - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
        out.append(t)
    return torch.stack(out)


@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
@pytest.mark.parametrize("B,H,W", [(5, 224, 224), (3, 37, 61)])
def test_uint8_normalise_matches_float_cpu_pipeline(layout, B, H, W):
    from ilvlm_amd import ops
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    u8[0] = 255; u8[1, :, :, 1] = 0
    flags = torch.tensor([(i * 7) % 4 for i in range(B)], dtype=torch.uint8)
    want = cpu_pipeline(u8, flags.tolist())
    src = u8 if layout == "nhwc" else u8.permute(0, 3, 1, 2).contiguous()
    got = ops.image_u8_normalize(src.cuda(), flags=flags.cuda()).cpu()
    # ToTensor's / 255 and Normalize's / std as float32 divisions, as torch computes them: bit-identical
    assert torch.equal(got, want)
    plain = ops.image_u8_normalize(src.cuda()).cpu()
    assert torch.equal(plain, cpu_pipeline(u8, [0] * B))


def test_prefetcher_takes_uint8_batches():
    """uint8 batches (with and without flags) through DevicePrefetcher come out as the normalised float batch the model takes"""
    from ilvlm_amd.solver import DevicePrefetcher
    g = torch.Generator().manual_seed(4)
    batches = []
    for i in range(3):
        u8 = torch.randint(0, 256, (4, 32, 32, 3), generator=g, dtype=torch.uint8)
        flags = torch.tensor([0, 1, 2, 3], dtype=torch.uint8)
        tok = torch.zeros(4, 8, dtype=torch.int64); pad = torch.zeros(4, 8)
        batches.append(((u8, flags) if i % 2 == 0 else u8, (tok, pad)))
    outs = list(DevicePrefetcher(batches, tokenize=None, device="cuda"))
    assert len(outs) == 3
    for (img, text), (src, _) in zip(outs, batches):
        u8, flags = src if isinstance(src, tuple) else (src, torch.zeros(4, dtype=torch.uint8))
        torch.cuda.synchronize()
        assert img.dtype == torch.float32 and img.shape == (4, 3, 32, 32) and img.is_cuda
        assert torch.equal(img.cpu(), cpu_pipeline(u8, flags.tolist()))


from augment_ref import cpu_augment  # noqa: E402  (tests/augment_ref.py: PIL's arithmetic restated, pinned against PIL on the CPU)


def test_mocov2_augmentations_on_the_device_equal_the_pil_arithmetic_bit_for_bit():
    """Every operation of MOCOV2_single after decode, with the random draws fixed: images of several sizes (up- and
    down-scaling by the crop, a 4.6x reduction among them), all four colour operations in three different orders, grayscale,
    a narrow and the widest blur, flips.  The kernels repeat PIL's arithmetic (8-bit fixed-point resampling, Image.blend, 8-bit
    HSV, box-blur passes) and torchvision's float32 ToTensor / Normalize: the output equals the restatement -- which
    tests/test_augment_cpu.py holds to PIL itself -- BIT FOR BIT."""
    import random
    from ilvlm_amd import ops, lib as L
    OUT = 224
    rng = np.random.RandomState(7)
    sizes = [(300, 400), (48, 64), (1100, 900), (224, 224), (500, 333), (260, 260)]
    imgs = []
    for (H, W) in sizes:          # smooth structure + noise, so that resampling and blur have something to act on
        yy, xx = np.mgrid[0:H, 0:W]
        base = np.stack([128 + 100 * np.sin(xx / 17.0 + c) * np.cos(yy / 23.0 - c) for c in range(3)], -1)
        imgs.append(np.clip(base + rng.randn(H, W, 3) * 20, 0, 255).astype(np.uint8))
    params = ops.mocov2_params(sizes, random.Random(3))
    # fix the draws so that every branch is exercised somewhere
    orders = [0 | (1 << 2) | (2 << 4) | (3 << 6), 3 | (2 << 2) | (1 << 4) | (0 << 6), 1 | (3 << 2) | (0 << 4) | (2 << 6)]
    for i, p in enumerate(params):
        p.jitter, p.jitter_order = int(i != 3), orders[i % 3]
        p.grayscale, p.flip = int(i == 4), int(i % 2)
        p.blur_sigma = [0.0, 0.1, 2.0, 0.7, 0.0, 1.3][i]
    params[2].crop_top, params[2].crop_left, params[2].crop_h, params[2].crop_w = 30, 10, 1030, 880        # 4.6x down
    params[1].crop_top, params[1].crop_left, params[1].crop_h, params[1].crop_w = 4, 8, 40, 50              # 5x up
    flat = np.concatenate([im.reshape(-1) for im in imgs])
    offs = np.cumsum([0] + [im.size for im in imgs[:-1]]).astype(np.int64)
    hw = np.array(sizes, dtype=np.int32)
    got = ops.image_augment(torch.from_numpy(flat).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(hw).cuda(), params, OUT)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    for i, (im, p) in enumerate(zip(imgs, params)):
        want = cpu_augment(im, p, OUT)
        assert np.array_equal(got[i], want), "image %d %s crop %dx%d: %d values differ, largest difference %.3e" % (
            i, sizes[i], p.crop_h, p.crop_w, int((got[i] != want).sum()), float(np.abs(got[i] - want).max()))
    assert got.shape == (len(sizes), 3, OUT, OUT) and np.isfinite(got).all()


def test_prefetcher_augments_decoded_images_on_the_device():
    """a loader that stops after decode hands lists of uint8 [H,W,3] images of different sizes to DevicePrefetcher: the batch
    comes out as the augmented, normalised [B,3,224,224] float tensor -- equal to the restatement run with the draws the
    prefetcher made (same seed), and different from batch to batch"""
    import random
    from ilvlm_amd import ops
    from ilvlm_amd.solver import DevicePrefetcher
    rng = np.random.RandomState(11)
    sizes = [(240, 320), (100, 90), (400, 260), (224, 224)]
    imgs = [torch.from_numpy(rng.randint(0, 256, (h, w, 3)).astype(np.uint8)) for h, w in sizes]
    tok = torch.zeros(4, 8, dtype=torch.int64); pad = torch.zeros(4, 8)
    outs = list(DevicePrefetcher([(imgs, (tok, pad)), (imgs, (tok, pad))], tokenize=None, device="cuda", augment_seed=5))
    torch.cuda.synchronize()
    assert len(outs) == 2
    for b, (img, _) in enumerate(outs):
        assert img.shape == (4, 3, 224, 224) and img.dtype == torch.float32 and img.is_cuda
        params = ops.mocov2_params(sizes, random.Random((5 << 32) ^ b))
        for i in range(4):
            want = cpu_augment(imgs[i].numpy(), params[i], 224)
            assert np.array_equal(img[i].cpu().numpy(), want), (b, i)
    assert not torch.equal(outs[0][0], outs[1][0])
