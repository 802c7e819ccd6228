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
    # one rounding apart at most: the kernel multiplies by 1/std where Normalize divides by std
    assert float((got - want).abs().max()) <= 4e-7 * float(want.abs().max())
    plain = ops.image_u8_normalize(src.cuda()).cpu()
    assert float((plain - cpu_pipeline(u8, [0] * B)).abs().max()) <= 4e-7 * float(want.abs().max())


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
        assert float((img.cpu() - cpu_pipeline(u8, flags.tolist())).abs().max()) < 1e-5
