"""ClipInfoCELoss on the fused HIP InfoNCE kernel (reference prototype/loss_functions/loss.py:24-47):
labels = rank*B + arange(B) when the logit matrix is not square, loss = (CE_i + CE_t) / 2 with mean reduction."""
import torch
from torch.nn.modules.loss import _Loss

from ... import ops
from .. import linklink as link


class _InfoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_i, logits_t, label_offset):
        li, lt = logits_i.contiguous(), logits_t.contiguous()
        loss = torch.empty(1, device=li.device, dtype=torch.float32)
        dli, dlt = torch.empty_like(li), torch.empty_like(lt)
        ops.infonce_fwd(li, lt, label_offset, loss, dli, dlt)
        ctx.save_for_backward(dli, dlt)
        return loss.view(())

    @staticmethod
    def backward(ctx, grad_out):
        dli, dlt = ctx.saved_tensors
        g = grad_out.reshape(1).to(torch.float32).contiguous()     # device scalar (e.g. 1 / world_size)
        return ops.scale_dev(dli, g), ops.scale_dev(dlt, g), None


class ClipInfoCELoss(_Loss):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._labels = {}       # (batch, offset, device) -> label vector: built once, not with two ATen kernels per step

    def forward(self, logits_per_image, logits_per_text):
        bs, l_bs = logits_per_image.shape
        offset = 0 if l_bs == bs else link.get_rank() * bs
        key = (bs, offset, logits_per_image.device)
        labels = self._labels.get(key)
        if labels is None:
            labels = self._labels[key] = offset + torch.arange(bs, dtype=torch.long, device=logits_per_image.device)
        loss = _InfoNCE.apply(logits_per_image, logits_per_text, offset)
        return loss, labels
