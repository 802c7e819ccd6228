"""Harness that imports the UNMODIFIED reference (/root/reference) on CPU.

Used ONLY by tests/golden/make_golden.py in the build container to produce the
committed fixtures.  Never imported by the product, tests, smoke or bench; the
reference does not exist on the GPU box.

Shims (none on the arithmetic path):
  * stub modules for absent third-party imports (easydict, ftfy, timm)
  * Tensor.cuda / Module.cuda -> identity (the reference hard-codes .cuda())
  * a gloo process group so the reference AllGather works
"""
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("ILVLM_REFERENCE", "/root/reference")
BPE = os.path.join(REF, "prototype/model/text_encoder/bpe_simple_vocab_16e6.txt.gz")


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        dict.__setitem__(self, k, v)
        object.__setattr__(self, k, v)

    __setitem__ = __setattr__


def install(rank=0, world_size=1, port=29511, init_pg=True):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    stubs = {
        "easydict": dict(EasyDict=EasyDict),
        "ftfy": dict(fix_text=lambda t: t),
        "timm": {},
        "timm.models": {},
        "timm.models.layers": dict(DropPath=nn.Identity, to_2tuple=lambda x: (x, x),
                                   trunc_normal_=nn.init.trunc_normal_),
    }
    for name, attrs in stubs.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world_size),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if init_pg and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world_size)


def build(model_type, kwargs):
    """model_entry from the reference with pre-tokenised text input."""
    from prototype.model import model_entry
    model = model_entry(dict(type=model_type, kwargs=kwargs))
    model.encode_text.tokenize = lambda text, **kw: text
    return model
