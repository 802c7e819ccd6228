"""Baseline CLIP on the MI355X engine (mirror of reference prototype/model/clip.py:46-149, factory :190-197):
pooled projected features, image features normalised without eps, text with +1e-10, gathered logits."""
import numpy as np
import torch
from torch import nn

from .base import ContrastiveBase, default_precision
from .image_encoder.visual_transformer import visual_transformer_B32
from .text_encoder.text_transformer import text_transformers
from ... import ops


class CLIP(ContrastiveBase):
    def __init__(self, image_encode, text_encode, use_allgather, precision=None):
        super().__init__()
        self.use_allgather = use_allgather
        self.visual = image_encode
        self.encode_text = text_encode
        self.logit_scale = nn.Parameter(torch.ones([1]))
        nn.init.constant_(self.logit_scale, np.log(1 / 0.07))
        v, t = self.visual, self.encode_text
        self._init_engine(dict(
            precision=precision or default_precision(), fdt=False,
            res=v.input_resolution, patch=v.patch_size, v_layers=v.transformer.layers, v_heads=v.transformer.heads,
            ctx=t.context_length, t_layers=t.transformer.layers, t_heads=t.transformer.heads))

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def unused_parameter_names(self):
        return []

    def forward(self, images, texts):
        return self._run(images, texts)

    def _forward_impl(self, images, tokens, pad_mask, save, seq=None):
        e = self._eng
        B = images.shape[0]
        main, side = torch.cuda.current_stream(), e.side_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):              # text tower concurrently with the vision tower
            xt, st = e.text_fwd(tokens, save, seq)
            Lt = tokens.shape[1]
            txt, _, spt = e.text_pooled(xt, tokens, B, Lt, save, seq)
        xv, sv = e.vision_fwd(images, save)
        Lv = xv.shape[0] // B
        img, _, spv = e.vision_pooled(xv, B, Lv, save)
        main.wait_stream(side)
        li, lt, sh = e.head_fwd(img, txt, 0.0, 1e-10, save)
        saved = dict(vision=sv, text=st, pv=spv, pt=spt, head=sh, B=B, Lv=Lv, Lt=Lt, xv=xv.shape, xt=xt.shape) if save else None
        return li, lt, saved

    def _backward_impl(self, s, dli, dlt):
        e = self._eng
        main, side = torch.cuda.current_stream(), e.side_stream
        d_img, d_txt = e.head_bwd(s["head"], dli, dlt)
        d_txt.record_stream(side)
        lp = e.T != torch.float32
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dxt = torch.zeros(s["xt"], dtype=torch.float32, device=dli.device)
            e.text_pooled_bwd(s["pt"], d_txt, dxt, s["B"], s["Lt"])
            dxt_lp = None
            if lp:
                dxt_lp = torch.empty(s["xt"], dtype=e.T, device=dli.device)
                ops.cast_f32(dxt, dxt_lp)
            e.text_bwd(s["text"], dxt, dxt_lp)
            e.join_wgrad()
            self._sync("text_done")
        dxv = torch.zeros(s["xv"], dtype=torch.float32, device=dli.device)
        e.vision_pooled_bwd(s["pv"], d_img, dxv, s["B"], s["Lv"])
        dxv_lp = None
        if lp:
            dxv_lp = torch.empty(s["xv"], dtype=e.T, device=dli.device)
            ops.cast_f32(dxv, dxv_lp)
        e.vision_bwd(s["vision"], dxv, dxv_lp)
        e.join_wgrad()
        main.wait_stream(side)
        self._sync("all_done")

    @torch.no_grad()
    def encode_image(self, image, return_dense=False, return_att=False):
        e = self._eng
        e.prepare()
        xv, _ = e.vision_fwd(image, False)
        B = image.shape[0]
        Lv = xv.shape[0] // B
        proj, _, _ = e.vision_pooled(xv, B, Lv, False)
        return (proj, xv.view(B, Lv, -1)[:, 1:, :]) if return_dense else proj


def clip_vitb32(**kwargs):
    extra = {k: v for k, v in kwargs.items() if k not in ("image_encode", "text_encode", "clip")}
    return CLIP(visual_transformer_B32(**kwargs["image_encode"]), text_transformers(**kwargs["text_encode"]),
                **kwargs["clip"], **extra)
