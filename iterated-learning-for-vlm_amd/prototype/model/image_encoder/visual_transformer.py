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
