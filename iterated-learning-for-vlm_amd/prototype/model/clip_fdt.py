"""CLIP + FDT (finite discrete tokens) model on the MI355X engine.

Mirror of reference prototype/model/clip_fdt.py: class and attribute names (Query_model :54-161, Clip_FDT :193-428),
factory names (:443-460), state_dict keys and the iterated-learning helpers (reset_text_encoder :256-261,
freeze/unfreeze :285-321).  forward(images, texts) -> ((logits_per_image, logits_per_text), (space_dict, space_dict)).
"""
import numpy as np
import torch
from torch import nn

from .base import ContrastiveBase, default_precision
from .image_encoder.visual_transformer import visual_transformer_B32, visual_transformer_B16, visual_transformer_L14
from .text_encoder.text_transformer import text_transformers, text_transformers_L


def weight_reset(m):
    """Same dispatch as reference clip_fdt.py:40-48: PyTorch-default reset_parameters() for Conv/Linear/LayerNorm."""
    if isinstance(m, (nn.Conv2d, nn.Linear, nn.ConvTranspose2d)):
        m.reset_parameters()
    elif isinstance(m, nn.BatchNorm2d):
        m.reset_parameters()
        m.running_mean.zero_()
        m.running_var.fill_(1)
    elif isinstance(m, nn.LayerNorm):
        m.reset_parameters()


class Query_model(nn.Module):
    """Parameter container of the token -> codebook query head; executed by Engine.qmap_* / fdt_*."""

    def __init__(self, ft_dim, sd_dim, temperature=1, att_func_type="softmax", pool_type="sum"):
        super().__init__()
        assert att_func_type in ["softmax", "sigmoid", "sparsemax"]
        assert pool_type in ["mean", "max", "sum"]
        self.att_func_type, self.pool_type = att_func_type, pool_type
        self.att_dim = sd_dim
        self.temperature = temperature
        self.q_map = nn.Sequential(nn.LayerNorm(ft_dim), nn.Linear(ft_dim, sd_dim), nn.GELU(), nn.LayerNorm(sd_dim),
                                   nn.Linear(sd_dim, sd_dim))


class Clip_FDT(ContrastiveBase):
    def __init__(self, image_encode, text_encode, use_allgather, sd_num, sd_dim, raw_img_ft_dim, raw_txt_ft_dim,
                 att_func_type, pool_type, sd_temperature, precision=None):
        super().__init__()
        self.use_allgather = use_allgather
        self.visual = image_encode
        self.encode_text = text_encode
        self.space_dict = nn.Parameter(torch.randn(sd_num, sd_dim))
        self.img_query_model = Query_model(raw_img_ft_dim, sd_dim, sd_temperature, att_func_type, pool_type)
        self.txt_query_model = Query_model(raw_txt_ft_dim, sd_dim, sd_temperature, att_func_type, pool_type)
        self.logit_scale = nn.Parameter(torch.ones([1]))
        self.logit_scale_sd = nn.Parameter(torch.ones([1]))
        nn.init.constant_(self.logit_scale, np.log(1 / 0.07))
        nn.init.constant_(self.logit_scale_sd, np.log(1 / 0.07))
        self.stored_vision_encoder_weight = None
        self.weight_always_freeze = []
        v, t = self.visual, self.encode_text
        self._init_engine(dict(
            precision=precision or default_precision(), fdt=True, att_func=att_func_type, pool=pool_type,
            res=v.input_resolution, patch=v.patch_size, v_layers=v.transformer.layers, v_heads=v.transformer.heads,
            ctx=t.context_length, t_layers=t.transformer.layers, t_heads=t.transformer.heads))

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def unused_parameter_names(self):
        """Parameters the FDT loss never reaches (grad stays None in the reference, so AdamW never touches them):
        the pooled-feature heads and logit_scale_sd."""
        return ["logit_scale_sd", "visual.proj", "visual.ln_post.weight", "visual.ln_post.bias",
                "encode_text.text_projection.weight", "encode_text.text_projection.bias"]

    # ---------------------------------------------------------------- hot path
    def forward(self, images, texts):
        li, lt = self._run(images, texts)
        return (li, lt), (self.space_dict, self.space_dict)

    def _mark(self, name):
        """diagnostic (bench.py --phase-times): an event on the current stream at a phase boundary of the step"""
        marks = getattr(self, "_phase_marks", None)
        if marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((name, ev))

    def _forward_impl(self, images, tokens, pad_mask, save, seq=None):
        e = self._eng
        B = images.shape[0]
        self._mark("step_begin")
        main, side = torch.cuda.current_stream(), e.side_stream
        side.wait_stream(main)                     # parameters / shadow / inputs produced on the main stream
        with torch.cuda.stream(side):              # text tower + text query head (on the valid tokens only with seq)
            xt, st = e.text_fwd(tokens, save, seq)
            Lt, Wt = tokens.shape[1], xt.shape[1]
            words, sw = e.text_words(xt, save)
            qt, sqt = e.qmap_fwd(words, "txt_query_model.", xt.shape[0], Wt, 0, 0, save)
            _, ftt, sft = e.fdt_fwd(qt, B, Lt, pad_mask, self.txt_query_model.temperature, save, seq)
        xv, sv = e.vision_fwd(images, save)        # vision tower + image query head, concurrently on the main stream
        Lv, W = xv.shape[0] // B, xv.shape[1]
        qi, sqi = e.qmap_fwd(xv, "img_query_model.", B * (Lv - 1), W, Lv - 1, 1, save)
        _, fti, sfi = e.fdt_fwd(qi, B, Lv - 1, None, self.img_query_model.temperature, save)
        main.wait_stream(side)
        self._mark("towers_fwd_done")
        li, lt, sh = e.head_fwd(fti, ftt, 1e-10, 1e-10, save)
        self._mark("head_fwd_done")
        saved = dict(vision=sv, text=st, words=sw, qi=sqi, fi=sfi, qt=sqt, ft=sft, head=sh, B=B, Lv=Lv, W=W) if save else None
        return li, lt, saved

    def _backward_impl(self, s, dli, dlt):
        e = self._eng
        main, side = torch.cuda.current_stream(), e.side_stream
        self._mark("loss_done")
        d_fti, d_ftt = e.head_bwd(s["head"], dli, dlt)
        self._mark("head_bwd_done")
        d_ftt.record_stream(side)                  # allocated on the main stream, consumed on the side stream
        side.wait_stream(main)
        with torch.cuda.stream(side):              # text side
            dqt = e.fdt_bwd(s["ft"], d_ftt)
            dwords = e.qmap_bwd(s["qt"], "txt_query_model.", dqt)
            dxt, dxt_lp = e.text_words_bwd(s["words"], dwords)
            e.text_bwd(s["text"], dxt, dxt_lp)
            e.join_wgrad()
            self._sync("text_done")                # gradient all-reduce of the text ranges waits on this stream
        dqi = e.fdt_bwd(s["fi"], d_fti)            # image side, concurrently
        B, Lv, W = s["B"], s["Lv"], s["W"]
        # token-stream gradient entering the last block: the query head writes every patch row (remapped rows), the
        # class-token rows get no gradient under FDT
        dxv = torch.empty((B * Lv, W), dtype=torch.float32, device=dli.device)
        dxv.view(B, Lv, W)[:, 0].zero_()
        dxv_lp = None
        if e.T != torch.float32:
            dxv_lp = torch.empty((B * Lv, W), dtype=e.T, device=dli.device)
            dxv_lp.view(B, Lv, W)[:, 0].zero_()
        e.qmap_bwd(s["qi"], "img_query_model.", dqi, dxv, dxv_lp)
        e.vision_bwd(s["vision"], dxv, dxv_lp)
        e.join_wgrad()
        main.wait_stream(side)
        self._mark("towers_bwd_done")
        self._sync("all_done")

    # ---------------------------------------------------------------- evaluation-time API (no gradient)
    @torch.no_grad()
    def encode_image(self, image):
        """(projected cls [B,D], dense patch tokens before ln_post [B,P,W], ln_post(cls) [B,W])."""
        e = self._eng
        e.prepare()
        xv, _ = e.vision_fwd(image, False)
        B = image.shape[0]
        Lv = xv.shape[0] // B
        proj, feat, _ = e.vision_pooled(xv, B, Lv, False)
        return proj, xv.view(B, Lv, -1)[:, 1:, :], feat

    @torch.no_grad()
    def _query(self, side, ft, B, Tn, ftdim, group, skip, mask, return_token_att):
        e = self._eng
        qm = getattr(self, side[:-1])
        q, _ = e.qmap_fwd(ft, side, B * Tn, ftdim, group, skip, False)
        att_w, att_ft, _ = e.fdt_fwd(q, B, Tn, mask, qm.temperature, False)
        if return_token_att:
            sd = e.Wf["space_dict"]
            scores = torch.empty((B * Tn, sd.shape[0]), dtype=torch.float32, device=q.device)
            from ... import ops
            ops.gemm(q, e._mat("space_dict"), scores)
            scores = scores.view(B, Tn, -1)
            if mask is not None:
                scores = scores * ((mask == 0) * 1).unsqueeze(-1)
            return scores, att_ft, self.space_dict
        return att_w, att_ft, self.space_dict

    @torch.no_grad()
    def extract_img_sd_ft(self, images, return_token_att=False):
        e = self._eng
        e.prepare()
        xv, _ = e.vision_fwd(images, False)
        B = images.shape[0]
        Lv = xv.shape[0] // B
        return self._query("img_query_model.", xv, B, Lv - 1, xv.shape[1], Lv - 1, 1, None, return_token_att)

    @torch.no_grad()
    def extract_txt_sd_ft(self, texts, return_token_att=False, raw_text=True):
        e = self._eng
        e.prepare()
        tokens, pad_mask = self._text_inputs(texts, e.arena.P.device)
        xt, _ = e.text_fwd(tokens, False)
        words, _ = e.text_words(xt, False)
        B, Lt = tokens.shape
        return self._query("txt_query_model.", words, B, Lt, xt.shape[1], 0, 0, pad_mask, return_token_att)

    @torch.no_grad()
    def extract_patch_ft(self, images):
        e = self._eng
        e.prepare()
        xv, _ = e.vision_fwd(images, False)
        B = images.shape[0]
        Lv = xv.shape[0] // B
        q, _ = e.qmap_fwd(xv, "img_query_model.", B * (Lv - 1), xv.shape[1], Lv - 1, 1, False)
        return q.view(B, Lv - 1, -1)

    @torch.no_grad()
    def extract_word_ft(self, texts):
        e = self._eng
        e.prepare()
        tokens, pad_mask = self._text_inputs(texts, e.arena.P.device)
        xt, _ = e.text_fwd(tokens, False)
        words, _ = e.text_words(xt, False)
        B, Lt = tokens.shape
        q, _ = e.qmap_fwd(words, "txt_query_model.", B * Lt, xt.shape[1], 0, 0, False)
        return q.view(B, Lt, -1), pad_mask

    # ---------------------------------------------------------------- iterated learning / freezing helpers
    def find_always_freeze_weight(self):
        self.weight_always_freeze = [n for n, p in self.named_parameters() if not p.requires_grad]
        print("always freeze weight", self.weight_always_freeze)

    def reset_text_encoder(self, seed):
        """Re-initialise every Linear / LayerNorm of the text encoder and the text query head with PyTorch's default
        init under torch.manual_seed(seed); embeddings and attn.in_proj_* are left alone (reference :256-261)."""
        torch.manual_seed(seed)
        self.encode_text.apply(weight_reset)
        self.txt_query_model.apply(weight_reset)

    def reset_vision_encoder(self):
        self.visual.apply(weight_reset)
        self.img_query_model.apply(weight_reset)

    def swap_vision_encoder(self):
        cur = ({k: v.clone() for k, v in self.visual.state_dict().items()},
               {k: v.clone() for k, v in self.img_query_model.state_dict().items()})
        if self.stored_vision_encoder_weight is None:
            self.reset_vision_encoder()
        else:
            self.visual.load_state_dict(self.stored_vision_encoder_weight[0])
            self.img_query_model.load_state_dict(self.stored_vision_encoder_weight[1])
        self.stored_vision_encoder_weight = cur

    def reset_codebook(self):
        with torch.no_grad():
            self.space_dict.copy_(torch.randn(self.space_dict.shape))

    def unfreeze_weights(self, module_names, freeze_codebook=False):
        for encoder_name in module_names:
            for name, param in getattr(self, encoder_name).named_parameters():
                if name not in self.weight_always_freeze:     # encoder-relative names, as in the reference (:285-290)
                    param.requires_grad = True
        self.logit_scale.requires_grad = True
        self.logit_scale_sd.requires_grad = True
        self.space_dict.requires_grad = not freeze_codebook

    def unfreeze_all_parameters(self):
        for p in self.parameters():
            p.requires_grad = True

    def freeze_weights(self, module_names, freeze_codebook=False):
        for encoder_name in module_names:
            for p in getattr(self, encoder_name).parameters():
                p.requires_grad = False
        self.logit_scale.requires_grad = False
        self.logit_scale_sd.requires_grad = False
        self.space_dict.requires_grad = not freeze_codebook

    def freeze_unfreeze_vision_weights(self, unfreeze, freeze_codebook):
        names = ["visual", "img_query_model"]
        (self.unfreeze_weights if unfreeze else self.freeze_weights)(names, freeze_codebook)

    def freeze_unfreeze_text_weights(self, unfreeze, freeze_codebook):
        names = ["encode_text", "txt_query_model"]
        (self.unfreeze_weights if unfreeze else self.freeze_weights)(names, freeze_codebook)


def _build(vit, txt, kwargs):
    extra = {k: v for k, v in kwargs.items() if k not in ("image_encode", "text_encode", "fdt")}
    return Clip_FDT(vit(**kwargs["image_encode"]), txt(**kwargs["text_encode"]), **kwargs["fdt"], **extra)


def clip_fdt_vitb32(**kwargs):
    return _build(visual_transformer_B32, text_transformers, kwargs)


def clip_fdt_vitb16(**kwargs):
    return _build(visual_transformer_B16, text_transformers, kwargs)


def clip_fdt_vitL14(**kwargs):
    """ViT-L/14 + 768-wide text tower + FDT (BASELINE.json config 4; the reference has both towers,
    visual_transformer.py:134-150 and text_transformer.py:356-368, but no FDT factory for them)."""
    return _build(visual_transformer_L14, text_transformers_L, kwargs)
