"""CPU oracle for the CLIP / CLIP+FDT contrastive training step.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, plain-PyTorch (CPU, fp32) restatement of the reference
algorithm, written as explicit functional ops over a flat dict of tensors that uses the
reference's parameter names.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the product path
(`iterated-learning-for-vlm_amd/`) never does and fails loudly without its HIP library.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here against
fixtures under `tests/golden/` that were produced by importing the unmodified reference in
the build container (`tests/golden/make_golden.py`); the reference has no tests or golden
vectors of its own for this path.

Reference citations are relative to /root/reference.  Layout note: the reference runs its
transformers sequence-first ([L, N, E]); the oracle is batch-first ([N, L, E]) which is
the same arithmetic per token.
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-5

# --------------------------------------------------------------------------------------
# operand-rounding control.  By default Q is the identity and this file is the fp32 restatement.
# Tests that allow the bf16 HIP path more than the 1e-2 of north_star must show that the excess is
# the dtype's and not a kernel's: `with rounding(bf16_ste): ...` re-runs the SAME oracle with every
# matrix-product operand (weights, GEMM / attention inputs, softmax probabilities, the stored word
# features) rounded to bf16 while sums stay fp32 -- the storage points of the bf16 mode listed in
# DESIGN.md section 3 -- and the test compares the two errors.
# --------------------------------------------------------------------------------------
_ROUND = None


def Q(t):
    return t if _ROUND is None else _ROUND(t)


def bf16_ste(t):
    """round-to-nearest-even to bf16, gradient passed straight through"""
    return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


class rounding(object):
    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global _ROUND
        self.prev, _ROUND = _ROUND, self.fn

    def __exit__(self, *a):
        global _ROUND
        _ROUND = self.prev


# --------------------------------------------------------------------------------------
# elementwise / row ops
# --------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps=LN_EPS):
    """nn.LayerNorm over the last dim (image_encoder/base_transformer.py:10-18)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def quick_gelu(x):
    """x * sigmoid(1.702 x)  (image_encoder/base_transformer.py:24-26)."""
    return x * torch.sigmoid(1.702 * x)


def gelu_erf(x):
    """nn.GELU() default = exact erf form (clip_fdt.py:89)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x, w, b=None):
    y = Q(x) @ Q(w).t()
    return y if b is None else y + b


# --------------------------------------------------------------------------------------
# fp8 control (BASELINE configs[4]; the reference has no fp8 path -- this emulates what the fp8 MODE of the product
# computes, so that tests can tell the dtype's error from a kernel's).  Inside `with fp8_blocks():` the four linears of
# every residual attention block (in_proj, out_proj, c_fc, c_proj: exactly the GEMMs the product runs on fp8 operands) take
# OCP e4m3 activations and weights in the forward pass and e5m2 gradients in both backward products, each tensor with
# its own scale = format max / amax (the steady state of the product's delayed scaling when the batch repeats); products
# exact, sums fp32.  Combine with rounding(bf16_ste) for the bf16 storage of everything else.
# --------------------------------------------------------------------------------------
_FP8 = False


def _q8(t, dtype, fmax):
    amax = float(t.detach().abs().max())
    if amax == 0.0 or not math.isfinite(amax):
        return t
    s = fmax / amax
    return (t * s).clamp(-fmax, fmax).to(dtype).to(t.dtype) / s


class _Fp8Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x8, w8 = _q8(x, torch.float8_e4m3fn, 448.0), _q8(w, torch.float8_e4m3fn, 448.0)
        ctx.save_for_backward(x8, w8)
        return x8 @ w8.t()

    @staticmethod
    def backward(ctx, dy):
        x8, w8 = ctx.saved_tensors
        g8 = _q8(dy, torch.float8_e5m2, 57344.0)
        dx = g8 @ w8
        dw = g8.reshape(-1, g8.shape[-1]).t() @ x8.reshape(-1, x8.shape[-1])
        return dx, dw


class fp8_blocks(object):
    def __enter__(self):
        global _FP8
        self.prev, _FP8 = _FP8, True

    def __exit__(self, *a):
        global _FP8
        _FP8 = self.prev


def block_linear(x, w, b=None):
    """the four GEMMs of a residual attention block: fp8 operands under fp8_blocks(), else linear()"""
    if not _FP8:
        return linear(x, w, b)
    y = _Fp8Linear.apply(x, w)
    return y if b is None else y + b


# --------------------------------------------------------------------------------------
# transformer
# --------------------------------------------------------------------------------------
def causal_mask(L, dtype=torch.float32):
    """Additive -inf strictly-upper-triangular mask (text_transformer.py:147-153)."""
    m = torch.full((L, L), float("-inf"), dtype=dtype)
    return torch.triu(m, diagonal=1)


def mha(x, in_w, in_b, out_w, out_b, heads, causal):
    """nn.MultiheadAttention(x, x, x, need_weights=True, attn_mask=mask) output only.

    torch.nn.functional.multi_head_attention_forward math path: q is pre-scaled by
    sqrt(1/head_dim), scores = q k^T (+ mask), softmax, @ v, out_proj
    (called at image_encoder/base_transformer.py:45-48, text_encoder/base_transformer.py:45-48).
    """
    B, L, E = x.shape
    hd = E // heads
    qkv = Q(block_linear(x, in_w, in_b))
    q, k, v = qkv.split(E, dim=-1)
    q = q.reshape(B, L, heads, hd).transpose(1, 2) * math.sqrt(1.0 / hd)
    k = k.reshape(B, L, heads, hd).transpose(1, 2)
    v = v.reshape(B, L, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + causal_mask(L, s.dtype)
    p = torch.softmax(s, dim=-1)
    o = (Q(p) @ v).transpose(1, 2).reshape(B, L, E)
    return block_linear(o, out_w, out_b)


def resblock(x, p, pre, heads, causal):
    """ResidualAttentionBlock.forward (image_encoder/base_transformer.py:50-62)."""
    h = layer_norm(x, p[pre + "ln_1.weight"], p[pre + "ln_1.bias"])
    x = x + mha(h, p[pre + "attn.in_proj_weight"], p[pre + "attn.in_proj_bias"],
                p[pre + "attn.out_proj.weight"], p[pre + "attn.out_proj.bias"], heads, causal)
    h = layer_norm(x, p[pre + "ln_2.weight"], p[pre + "ln_2.bias"])
    h = quick_gelu(block_linear(h, p[pre + "mlp.c_fc.weight"], p[pre + "mlp.c_fc.bias"]))
    return x + block_linear(h, p[pre + "mlp.c_proj.weight"], p[pre + "mlp.c_proj.bias"])


def n_layers(p, pre):
    n = 0
    while (pre + "transformer.resblocks.%d.ln_1.weight" % n) in p:
        n += 1
    return n


def vit_forward(images, p, heads, pre="visual."):
    """VisualTransformer.forward (image_encoder/visual_transformer.py:55-91).

    Returns (projected cls [B,D], dense patch tokens BEFORE ln_post [B,P,W], ln_post(cls) [B,W]).
    """
    w = p[pre + "conv1.weight"]
    patch = w.shape[-1]
    x = F.conv2d(Q(images), Q(w), stride=patch)                # [B,W,g,g]
    B, W = x.shape[0], x.shape[1]
    x = x.reshape(B, W, -1).permute(0, 2, 1)                   # [B,P,W]
    cls = p[pre + "class_embedding"].reshape(1, 1, W).expand(B, 1, W)
    x = torch.cat([cls, x], dim=1) + p[pre + "positional_embedding"]
    x = layer_norm(x, p[pre + "ln_pre.weight"], p[pre + "ln_pre.bias"])
    for i in range(n_layers(p, pre)):
        x = resblock(x, p, pre + "transformer.resblocks.%d." % i, heads, causal=False)
    dense = x[:, 1:, :]
    feat = layer_norm(x[:, 0, :], p[pre + "ln_post.weight"], p[pre + "ln_post.bias"])
    return feat @ p[pre + "proj"], dense, feat


def text_forward(tokens, p, heads, pre="encode_text."):
    """TextTransformer.forward, 'Transformer' branch (text_encoder/text_transformer.py:211-338).

    tokens: int64 [B, ctx].  Returns (projected eot [B,D], ln_final word tokens [B,ctx,W],
    eot feature before projection [B,W]).
    """
    x = p[pre + "token_embedding.weight"][tokens] + p[pre + "positional_embedding"]
    for i in range(n_layers(p, pre)):
        x = resblock(x, p, pre + "transformer.resblocks.%d." % i, heads, causal=True)
    x = layer_norm(x, p[pre + "ln_final.weight"], p[pre + "ln_final.bias"])
    words = Q(x)                       # the word features are stored in the compute dtype; the pooled head stays fp32
    feat = x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)]
    out = feat @ p[pre + "text_projection.weight"].t() + p[pre + "text_projection.bias"]
    return out, words, feat


# --------------------------------------------------------------------------------------
# FDT
# --------------------------------------------------------------------------------------
def sparsemax(z):
    """Sparsemax over the last dim of a 2-D tensor (prototype/model/sparsemax.py:22-71)."""
    n = z.shape[-1]
    z = z - z.max(dim=-1, keepdim=True)[0]
    zs = torch.sort(z, dim=-1, descending=True)[0]
    rng = torch.arange(1, n + 1, dtype=z.dtype).reshape(1, -1)
    bound = 1 + rng * zs
    csum = torch.cumsum(zs, dim=-1)
    is_gt = (bound > csum).to(z.dtype)
    k = (is_gt * rng).max(dim=-1, keepdim=True)[0]
    tau = ((is_gt * zs).sum(dim=-1, keepdim=True) - 1) / k
    return torch.clamp(z - tau, min=0)


def q_map(ft, p, pre):
    """Query_model.q_map: LN -> Linear -> GELU(erf) -> LN -> Linear (clip_fdt.py:86-92)."""
    h = layer_norm(ft, p[pre + "q_map.0.weight"], p[pre + "q_map.0.bias"])
    h = gelu_erf(linear(h, p[pre + "q_map.1.weight"], p[pre + "q_map.1.bias"]))
    h = layer_norm(h, p[pre + "q_map.3.weight"], p[pre + "q_map.3.bias"])
    return linear(h, p[pre + "q_map.4.weight"], p[pre + "q_map.4.bias"])


def query_model(ft, sd, p, pre, temperature, att_func, pool, mask=None):
    """Query_model.forward (clip_fdt.py:96-161).

    mask: [B,T] additive pad mask (0 valid / -inf pad).  Padded tokens are MULTIPLIED by 0
    (clip_fdt.py:122-127), they are not excluded from the pooling.
    Returns dict(q, pooled, att_w, att_ft).
    """
    q = Q(q_map(ft, p, pre))
    dot = q @ Q(sd).t()
    dot = dot / math.sqrt(sd.shape[1])
    if mask is not None:
        dot = dot * ((mask == 0) * 1).unsqueeze(-1)
    dot = dot / temperature
    if pool == "sum":
        pooled = dot.sum(1)
    elif pool == "mean":
        pooled = dot.mean(1)
    else:
        pooled = dot.max(1)[0]
    if att_func == "softmax":
        att_w = torch.softmax(pooled, dim=-1)
    elif att_func == "sparsemax":
        att_w = sparsemax(pooled)
    else:
        att_w = torch.sigmoid(pooled)
    att_ft = att_w @ sd
    if att_func == "sigmoid":
        att_ft = att_ft / att_w.sum(dim=-1, keepdim=True)
    return dict(q=q, pooled=pooled, att_w=att_w, att_ft=att_ft)


# --------------------------------------------------------------------------------------
# whole-model forwards.  `gather` maps a local [B,D] matrix to the rank-major [W*B,D]
# global matrix (identity for one rank), as AllGather does (clip_fdt.py:164-188).
# --------------------------------------------------------------------------------------
def clip_fdt_forward(p, images, tokens, pad_mask, cfg, gather=lambda t: t):
    """Clip_FDT.forward (clip_fdt.py:390-428).  cfg keys: v_heads, t_heads, temperature,
    att_func, pool."""
    _, patch_ft, _ = vit_forward(images, p, cfg["v_heads"])
    _, word_ft, _ = text_forward(tokens, p, cfg["t_heads"])
    sd = p["space_dict"]
    qi = query_model(patch_ft, sd, p, "img_query_model.", cfg["temperature"], cfg["att_func"], cfg["pool"])
    qt = query_model(word_ft, sd, p, "txt_query_model.", cfg["temperature"], cfg["att_func"], cfg["pool"],
                     mask=pad_mask)
    img = qi["att_ft"] / (qi["att_ft"].norm(dim=-1, keepdim=True) + 1e-10)
    txt = qt["att_ft"] / (qt["att_ft"].norm(dim=-1, keepdim=True) + 1e-10)
    scale = torch.clamp(p["logit_scale"].exp().detach(), max=100) + (p["logit_scale"].exp() - p["logit_scale"].exp().detach())
    # ^ value clamped to <=100 in .data, gradient of exp() untouched (clip_fdt.py:415-416)
    g_img, g_txt = gather(img), gather(txt)
    logits_i = img @ g_txt.t() * scale
    logits_t = txt @ g_img.t() * scale
    return dict(logits_i=logits_i, logits_t=logits_t, img=img, txt=txt, patch_ft=patch_ft, word_ft=word_ft,
                img_q=qi, txt_q=qt)


def clip_forward(p, images, tokens, cfg, gather=lambda t: t):
    """CLIP.forward (prototype/model/clip.py:125-149): image features normalised WITHOUT
    eps, text with +1e-10; the scale multiplies the features before the matmul."""
    img, _, _ = vit_forward(images, p, cfg["v_heads"])
    txt, _, _ = text_forward(tokens, p, cfg["t_heads"])
    img = img / img.norm(dim=-1, keepdim=True)
    txt = txt / (txt.norm(dim=-1, keepdim=True) + 1e-10)
    scale = torch.clamp(p["logit_scale"].exp().detach(), max=100) + (p["logit_scale"].exp() - p["logit_scale"].exp().detach())
    g_img, g_txt = gather(img), gather(txt)
    logits_i = scale * img @ g_txt.t()
    logits_t = scale * txt @ g_img.t()
    return dict(logits_i=logits_i, logits_t=logits_t, img=img, txt=txt)


def info_nce(logits_i, logits_t, rank=0):
    """ClipInfoCELoss.forward (prototype/loss_functions/loss.py:37-47)."""
    bs, l_bs = logits_i.shape
    labels = torch.arange(bs) if l_bs == bs else rank * bs + torch.arange(bs)
    loss = (F.cross_entropy(logits_i, labels) + F.cross_entropy(logits_t, labels)) / 2
    return loss, labels


def accuracy(output, target, topk=(1,)):
    """prototype/utils/misc.py:464-477."""
    maxk = max(topk)
    pred = output.topk(maxk, 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0, keepdim=True) * (100.0 / target.size(0)) for k in topk]


def simulate_ranks(forward, p, per_rank_inputs):
    """One-process model of the W-rank step (SURVEY.md section 3.1 'effective gradient
    scaling'): every rank's local loss is divided by W (train_solver.py:420), gathered
    feature gradients are summed over ranks (clip_fdt.py:182-188) and parameter gradients
    are averaged by DDP (torch_ddp_dist.py:65).  Returns (list of per-rank outputs, list of
    per-rank local losses (already /W), total whose autograd gradient is W x the
    post-DDP gradient)."""
    W = len(per_rank_inputs)
    outs = [None] * W
    # pass 1: local features of every rank (autograd graph shared through p)
    feats = []
    for r, inp in enumerate(per_rank_inputs):
        o = forward(p, *inp, gather=lambda t: t)
        feats.append((o["img"], o["txt"]))
    g_img = torch.cat([f[0] for f in feats], 0)
    g_txt = torch.cat([f[1] for f in feats], 0)
    losses = []
    for r, inp in enumerate(per_rank_inputs):
        which = {}

        def gather(t, _r=r, _which=which):
            # first call is img, second is txt (same order as the forwards above)
            idx = len(_which)
            _which[idx] = True
            return g_img if idx == 0 else g_txt
        o = forward(p, *inp, gather=gather)
        outs[r] = o
        loss, labels = info_nce(o["logits_i"], o["logits_t"], rank=r)
        o["labels"] = labels
        losses.append(loss / W)
    return outs, losses, sum(losses)


# --------------------------------------------------------------------------------------
# optimiser side
# --------------------------------------------------------------------------------------
def cosine_lr(step, base_lr, warmup_lr, warmup_steps, max_iter, min_lr=0.0, reset_steps=0):
    """CosineLRScheduler._get_new_lr with the periodic re-warm-up
    (prototype/lr_scheduler/scheduler.py:68-94, 239-255).  Returns the lr of a group whose
    initial lr equals base_lr."""
    ratio = (step - warmup_steps) / (max_iter - warmup_steps)
    target = min_lr + (warmup_lr - min_lr) * (1 + math.cos(math.pi * ratio)) / 2
    scale = target / base_lr
    if warmup_steps >= 2:
        if step < warmup_steps:
            t = (warmup_lr - base_lr) / (warmup_steps - 1) * (step - 1) + base_lr
            return t / base_lr * base_lr
        if reset_steps > 0 and step % reset_steps < warmup_steps:
            s = step % reset_steps
            t = (warmup_lr - base_lr) / (warmup_steps - 1) * (s - 1) + base_lr
            return scale * (t / warmup_lr) * base_lr
    return scale * base_lr


def adamw_step(p, g, m, v, step, lr, beta1, beta2, eps, wd):
    """torch.optim.AdamW single-tensor update (decoupled weight decay, amsgrad=False), as
    selected by prototype/optimizer/__init__.py:3,18-26.  In place; step is 1-based."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def param_group_of(name, shape, is_ln_weight, is_bias_of_module):
    """Group index (0..9) a parameter lands in under param_group_all with the shipped pconfig
    (prototype/utils/misc.py:285-461; example/clip_fdt/config_cc3m.yaml:43-55):
    0 normal, 1 bn_w, 2 bn_b, 3 conv_b, 4 linear_b, 5 ln_w, 6 ln_b, 7 code_trs,
    8 logit_scale, 9 bias."""
    if is_ln_weight:
        return 5
    if is_bias_of_module:
        return 9
    if "logit_scale" in name:
        return 8
    return 0
