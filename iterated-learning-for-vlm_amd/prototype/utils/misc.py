"""Solver-side helpers with the reference's names and behaviour (prototype/utils/misc.py):
parse_config :64-69, AverageMeter :22-56, param_group_all :285-461, accuracy :464-477, load_state_model :490-506."""
import copy
import logging
import os
from collections import defaultdict

import numpy as np
import torch
import yaml

from .. import linklink as link


class EasyDict(dict):
    """Minimal attribute dictionary (the reference depends on the third-party `easydict`)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(EasyDict(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x for x in v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def parse_config(config_file):
    with open(config_file) as f:
        return EasyDict(yaml.load(f, Loader=yaml.FullLoader))


def get_logger(name, level=logging.INFO):
    logger = logging.getLogger(name)
    logger.setLevel(level)
    return logger


class _RankFilter(logging.Filter):
    def filter(self, record):
        return link.get_rank() == 0


def create_logger(log_file, level=logging.INFO):
    """log to file + stderr, non-zero ranks filtered (reference misc.py:91-127)."""
    root = logging.getLogger()
    root.setLevel(level)
    fmt = logging.Formatter("[%(asctime)s][%(filename)15s][line:%(lineno)4d][%(levelname)8s] %(message)s")
    for h in (logging.FileHandler(log_file), logging.StreamHandler()):
        h.setFormatter(fmt)
        h.addFilter(_RankFilter())
        root.addHandler(h)
    return root


def makedir(path):
    if link.get_rank() == 0 and not os.path.exists(path):
        os.makedirs(path, exist_ok=True)
    link.barrier()


def count_params(model):
    total = sum(p.numel() for p in model.parameters())
    get_logger(__name__).info("total param: %.3fM" % (total / 1e6))
    return total


class AverageMeter(object):
    """Sliding-window (length > 0) or cumulative average; reduce_update all-reduces (SUM) then reads the value."""

    def __init__(self, length=0):
        self.length = length
        self.reset()

    def reset(self):
        self.history, self.count, self.sum = [], 0, 0.0
        self.val = self.avg = 0.0

    def reduce_update(self, tensor, num=1):
        if link.get_world_size() > 1 and torch.distributed.is_initialized():
            link.allreduce(tensor)
        self.update(tensor.item(), num=num)

    def update(self, val, num=1):
        if self.length > 0:
            assert num == 1
            self.history.append(val)
            if len(self.history) > self.length:
                del self.history[0]
            self.val = self.history[-1]
            self.avg = float(np.mean(self.history))
        else:
            self.val = val
            self.sum += val * num
            self.count += num
            self.avg = self.sum / self.count


def accuracy(output, target, topk=(1,)):
    """precision@k in percent of the local batch, on the HIP top-k kernel.  `target` must be the contiguous labels
    offset + arange(B) that ClipInfoCELoss returns (the only way the solvers call it)."""
    from ... import ops
    B = target.shape[0]
    offset = int(target[0]) if not target.is_cuda else None
    if offset is None:
        offset = link.get_rank() * B if output.shape[1] != B else 0
    res = []
    for k in topk:
        out = torch.empty(2, device=output.device, dtype=torch.float32)
        ops.topk_accuracy(output.detach().contiguous(), offset, int(k), out)
        res.append(out[1:2] if k > 1 else out[0:1])
    return res


# ---------------------------------------------------------------------------------------------------
_ALWAYS = ["bn_w", "bn_b", "conv_b", "linear_b", "ln_w", "ln_b", "code_trs", "space_dict"]
_OPTIONAL = ["conv_dw_w", "conv_dw_b", "conv_dense_w", "conv_dense_b", "linear_w", "logit_scale", "bias"]


def param_group_all(model, config, default_config={}):
    """Partition the parameters into optimizer groups exactly as the reference does: group 0 'normal', then one
    group per special kind (always-present kinds first, then the optional kinds named in `config`, in the reference's
    order; an empty 'space_dict' group is dropped).  Module-type rules: biases of Conv/Linear/BatchNorm/LayerNorm go
    to 'bias' when that kind is configured, LayerNorm/BatchNorm gains to ln_w / bn_w; name rules: 'logit_scale',
    'space_dict', 'code_trs'.  Parameters with requires_grad == False never enter the 'normal' group."""
    kinds = _ALWAYS + [k for k in _OPTIONAL if k in config]
    groups = {k: [] for k in kinds}
    taken = set()
    type2num = defaultdict(int)

    def put(kind, module_name, attr, param, tag):
        groups[kind].append(param)
        taken.add(module_name + "." + attr if module_name else attr)
        type2num[tag] += 1

    for name, m in model.named_modules():
        cls = m.__class__.__name__
        if isinstance(m, torch.nn.Conv2d):
            depthwise = m.groups == m.in_channels
            if m.bias is not None:
                kind = ("bias" if "bias" in groups else "conv_dw_b" if "conv_dw_b" in groups and depthwise else
                        "conv_dense_b" if "conv_dense_b" in groups and m.groups == 1 else "conv_b")
                put(kind, name, "bias", m.bias, cls + ".bias")
            if "conv_dw_w" in groups and depthwise:
                put("conv_dw_w", name, "weight", m.weight, cls + ".weight(dw)")
            elif "conv_dense_w" in groups and m.groups == 1:
                put("conv_dense_w", name, "weight", m.weight, cls + ".weight(dense)")
        elif isinstance(m, torch.nn.Linear):
            if m.bias is not None:
                put("bias" if "bias" in groups else "linear_b", name, "bias", m.bias, cls + ".bias")
            if "linear_w" in groups:
                put("linear_w", name, "weight", m.weight, cls + ".weight")
        elif isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            if m.weight is not None:
                put("bn_w", name, "weight", m.weight, cls + ".weight")
            if m.bias is not None:
                put("bias" if "bias" in groups else "bn_b", name, "bias", m.bias, cls + ".bias")
        elif isinstance(m, torch.nn.LayerNorm):
            if m.weight is not None:
                put("ln_w", name, "weight", m.weight, cls + ".weight")
            if m.bias is not None:
                put("bias" if "bias" in groups else "ln_b", name, "bias", m.bias, cls + ".bias")

    normal = []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if "code_trs" in config and "code_trs" in name and name.endswith((".w", ".b", "w_1", "w_2")):
            groups["code_trs"].append(p); taken.add(name); type2num["code_trs"] += 1
        if "space_dict" in config and "space_dict" in name:
            groups["space_dict"].append(p); taken.add(name); type2num["space_dict"] += 1
        if "logit_scale" in config and "logit_scale" in name:
            groups["logit_scale"].append(p); taken.add(name)
        if name not in taken:
            normal.append(p)

    logger = get_logger(__name__)
    param_groups = [{"params": normal, **default_config}]
    for kind in kinds:
        if kind == "space_dict" and not groups[kind]:
            continue
        cfg = copy.deepcopy(default_config)
        if kind in config:
            cfg.update(config[kind])
        param_groups.append({"params": groups[kind], **cfg})
        logger.info("%s: %d params %s" % (kind, len(groups[kind]), cfg))
    return param_groups, type2num


def load_state_model(model, state):
    """non-strict load with warnings (reference misc.py:490-506)"""
    logger = get_logger(__name__)
    missing, unexpected = model.load_state_dict(state, strict=False)
    for k in missing:
        logger.warning("missing key: %s" % k)
    for k in unexpected:
        logger.warning("unexpected key: %s" % k)


def load_state_optimizer(optimizer, state):
    optimizer.load_state_dict(state)
