"""Deterministic parameter / input generation shared by make_golden.py (which feeds the
values to the REFERENCE) and by the tests (which feed the same values to the oracle and the
HIP path).  numpy RandomState is frozen across numpy versions, so fixtures only need to
store seeds, not weights.  Biases and LayerNorm affine terms are deliberately non-trivial
(the reference's default init has zero biases, which would hide bias bugs)."""
import zlib

import numpy as np

SOT, EOT, VOCAB = 49407, 49408, 49409


def _rs(name, seed):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def det_param(name, shape, seed, logit_scale=None):
    """Value for parameter `name` of `shape` (tuple)."""
    shape = tuple(shape)
    n = _rs(name, seed).standard_normal(size=shape).astype(np.float32)
    if "logit_scale" in name:
        v = np.log(1 / 0.07) if logit_scale is None else logit_scale
        return np.full(shape, v, dtype=np.float32)
    if len(shape) == 1:
        if name.endswith(".weight"):          # LayerNorm gain
            return (1.0 + 0.1 * n).astype(np.float32)
        if "bias" in name:
            return (0.05 * n).astype(np.float32)
        return (n * shape[0] ** -0.5).astype(np.float32)   # class_embedding
    if name.endswith("positional_embedding"):
        return (0.01 * n).astype(np.float32)
    if name.endswith("token_embedding.weight"):
        return (0.02 * n).astype(np.float32)
    if name == "space_dict":
        return n
    if name.endswith("visual.proj") or name == "visual.proj":
        return (n * shape[0] ** -0.5).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    return (n * fan_in ** -0.5).astype(np.float32)


def det_state(shapes, seed, logit_scale=None):
    """shapes: ordered {name: shape}.  Returns {name: np.ndarray}."""
    return {k: det_param(k, s, seed, logit_scale) for k, s in shapes.items()}


def det_images(batch, res, seed):
    return _rs("images", seed).standard_normal(size=(batch, 3, res, res)).astype(np.float32)


def det_tokens(batch, ctx, seed, min_len=8):
    """SURVEY.md section 8(d) synthetic text: [SOT, U{0..49405} x (n-2), EOT, 0 ...]; returns
    (tokens int64 [B,ctx], pad_mask float32 [B,ctx] with 0 valid / -inf pad) as
    TextTransformer.tokenize does (text_transformer.py:182-194)."""
    rs = _rs("tokens", seed)
    toks = np.zeros((batch, ctx), dtype=np.int64)
    mask = np.full((batch, ctx), -np.inf, dtype=np.float32)
    for b in range(batch):
        n = int(rs.randint(min(min_len, ctx), ctx + 1))
        if b == 0:
            n = ctx          # always exercise the no-padding row
        toks[b, 0] = SOT
        toks[b, 1:n - 1] = rs.randint(0, 49406, size=n - 2)
        toks[b, n - 1] = EOT
        mask[b, :n] = 0.0
    return toks, mask


def probe_index(name, numel, k=64):
    """Fixed sample of flat indices used to pin large tensors (gradients) cheaply."""
    k = min(k, numel)
    return np.sort(_rs("probe:" + name, 7).choice(numel, size=k, replace=False)).astype(np.int64)


def probe(name, arr, k=64):
    """[sum, abs-sum, k sampled entries] of a tensor, float64."""
    a = np.asarray(arr, dtype=np.float64).reshape(-1)
    idx = probe_index(name, a.size, k)
    return np.concatenate([[a.sum(), np.abs(a).sum()], a[idx]])
