"""Parameter containers with the reference's module tree and names
(prototype/model/image_encoder/base_transformer.py:10-89).  They hold weights only: the arithmetic of a
block (LN -> MHA -> +res -> LN -> c_fc -> QuickGELU -> c_proj -> +res) is executed by engine.Engine.block_fwd /
block_bwd on HIP kernels, never by these modules' own forward."""
from collections import OrderedDict

from torch import nn


class LayerNorm(nn.LayerNorm):
    pass


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x); parameter-free marker (the kernel is the fused GEMM epilogue)."""


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head):
        super().__init__()
        if d_model % n_head or d_model // n_head != 64:
            raise ValueError("head_dim must be 64 (got d_model=%d heads=%d)" % (d_model, n_head))
        self.attn = nn.MultiheadAttention(d_model, n_head)      # in_proj_{weight,bias}, out_proj.{weight,bias}
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)


class Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


def init_blocks(transformer):
    """CLIP block init (visual_transformer.py:28-38, text_transformer.py:128-141)."""
    proj_std = (transformer.width ** -0.5) * ((2 * transformer.layers) ** -0.5)
    attn_std = transformer.width ** -0.5
    fc_std = (2 * transformer.width) ** -0.5
    for blk in transformer.resblocks:
        nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
        nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
        nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
        nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)
