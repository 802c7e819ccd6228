"""ViT image encoder container (reference prototype/model/image_encoder/visual_transformer.py:6-150).
conv1 is frozen whenever train() is called, as in the reference (:40-52)."""
import torch
from torch import nn

from .base_transformer import Transformer, LayerNorm, init_blocks


class VisualTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, embed_dim, checkpoint=False, dropout=0,
                 emb_dropout=0):
        super().__init__()
        if dropout or emb_dropout:
            raise NotImplementedError("dropout is 0 in every shipped config; not implemented on the HIP path")
        if input_resolution % patch_size:
            raise ValueError("input_resolution must be a multiple of patch_size")
        self.input_resolution, self.patch_size, self.output_dim = input_resolution, patch_size, embed_dim
        self.freeze_conv1 = True
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, embed_dim))
        nn.init.normal_(self.positional_embedding, std=0.01)
        init_blocks(self.transformer)

    def forward(self, x, return_dense=False, return_raw_feature=False, return_att=False):
        """Inference-only call with the reference signature (visual_transformer.py:55-91): projected class feature
        [, dense patch tokens before ln_post][, ln_post(cls)].  Training goes through the owning model's forward."""
        if return_att:
            raise NotImplementedError("attention maps are not produced on the HIP path")
        owner = getattr(self, "_owner", lambda: None)()
        if owner is None:
            raise RuntimeError("VisualTransformer runs inside a CLIP / Clip_FDT model (its engine owns the kernels)")
        with torch.no_grad():
            e = owner._eng
            e.prepare()
            xv, _ = e.vision_fwd(x, False)
            B = x.shape[0]
            Lv = xv.shape[0] // B
            proj, feat, _ = e.vision_pooled(xv, B, Lv, False)
        ret = [proj]
        if return_dense:
            ret.append(xv.view(B, Lv, -1)[:, 1:, :])
        if return_raw_feature:
            ret.append(feat)
        return ret[0] if len(ret) == 1 else tuple(ret)

    def train(self, mode=True):
        super().train(mode)
        if self.freeze_conv1:
            self.conv1.eval()
            for p in self.conv1.parameters():
                p.requires_grad = False
        return self


def _vit(defaults, kwargs):
    d = dict(defaults)
    d.update(kwargs)
    return VisualTransformer(**d)


def visual_transformer_B32(**kwargs):
    return _vit(dict(layers=12, heads=12, input_resolution=224, patch_size=32, width=768, checkpoint=False), kwargs)


def visual_transformer_B16(**kwargs):
    return _vit(dict(layers=12, heads=12, input_resolution=224, patch_size=16, width=768, checkpoint=False), kwargs)


def visual_transformer_L14(**kwargs):
    return _vit(dict(layers=24, heads=16, input_resolution=224, patch_size=14, width=1024, checkpoint=False), kwargs)
