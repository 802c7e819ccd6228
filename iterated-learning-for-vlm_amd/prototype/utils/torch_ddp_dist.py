"""Process-group setup and the data-parallel wrapper (reference prototype/utils/torch_ddp_dist.py:9-67).

One process per GPU, torchrun env contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), backend
'nccl' (= RCCL over xGMI on MI355X).  convert_to_ddp_model returns a wrapper exposing `.module` like torch DDP, but
gradient averaging is done on the flat gradient arena: each transformer block's range is reduced on a side stream as
soon as that block's backward is complete (the two towers announce their blocks from their own streams), the rest of
the text side when the text backward ends and whatever is left at the end; unused and frozen parameters ride along as
zeros (the reference needs find_unused_parameters=True for them).  The gradient mean is taken in fp32, as the reference's
DDP does; ILVLM_GRAD_BUCKET=bf16 opts into bf16 buckets (half the bytes on the wire, one bf16 rounding of every averaged
gradient; tests/test_comm_gloo.py bounds the effect on a two-rank trajectory)."""
import os
import random

import numpy as np
import torch
import torch.distributed as distributed
from torch import nn

from ... import comm


def get_world_size():
    return int(os.environ.get("WORLD_SIZE", 1))


def get_local_rank():
    return int(os.environ.get("LOCAL_RANK", 0))


def get_rank():
    return int(os.environ.get("RANK", 0))


def is_main_process():
    return get_rank() == 0


def set_random_seed(seed=0):
    """seed 0 on every rank, as the reference does (:21-27)"""
    random.seed(seed)
    np.random.seed(seed)
    torch.random.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def init_ddp(backend="nccl"):
    local_rank, world_size, rank = get_local_rank(), get_world_size(), get_rank()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", random.randint(6000, 60000)))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    distributed.init_process_group(backend=backend, init_method="tcp://%s:%d" % (addr, port), world_size=world_size,
                                   rank=rank)


class NativeDDP(nn.Module):
    """Data-parallel wrapper over the flat gradient arena.  The engine announces "<block prefix>" when a transformer
    block's backward is complete, "text_done" when the text side is finished and "all_done" at the end of backward; each
    announcement all-reduces (mean) the arena ranges that became final, on a side stream, while backward continues."""

    def __init__(self, module):
        super().__init__()
        self.module = module
        module._eng.prepare()                      # adopt the parameters into the arena now
        arena = module._eng.arena
        if comm._active(comm.world()[1]):
            distributed.broadcast(arena.P, 0)      # ONE flattened broadcast (reference: one per state_dict tensor)
        bucket = os.environ.get("ILVLM_GRAD_BUCKET", "fp32")       # bf16 buckets are opt-in (not the reference's arithmetic)
        if module._eng.precision == "fp32":
            bucket = "fp32"
        arena.reducer = comm.GradReducer(arena.G, bucket=bucket)
        self._done = []                            # ranges already reduced in this backward
        object.__setattr__(module, "_grad_sync", self._on_sync)

    def _reduce(self, b, e):
        if e > b:
            eng = self.module._eng
            # the range's weight gradients come from the companion of the announcing stream: the communication stream waits
            # for both, so the tower's own stream never has to join its companion in the middle of backward
            wg = eng._wg.get(torch.cuda.current_stream().cuda_stream) if torch.cuda.is_available() else None
            eng.arena.reducer.reduce_range(b, e, also_wait=wg)
            self._done.append((b, e))

    def _on_sync(self, what):
        a = self.module._eng.arena
        if what.endswith("."):                     # one transformer block
            self._reduce(*a.range_of(what))
        elif what == "text_done":                  # everything of the text side that is not a block
            t0, t1 = a.range_of("encode_text.")
            q0, q1 = a.range_of("txt_query_model.")
            for b, e in self._gaps([(t0, t1), (q0, q1)]):
                self._reduce(b, e)
        elif what == "all_done":
            for b, e in self._gaps([(0, a.total)]):
                self._reduce(b, e)
            self._done = []

    def _gaps(self, spans):
        """parts of `spans` not covered by ranges reduced earlier in this backward"""
        out = []
        done = sorted(self._done)
        for b, e in spans:
            cur = b
            for db, de in done:
                if de <= cur or db >= e:
                    continue
                if db > cur:
                    out.append((cur, db))
                cur = max(cur, de)
            if cur < e:
                out.append((cur, e))
        return out

    def forward(self, *args, **kwargs):
        self._done = []        # a backward that raised half-way must not leave ranges marked as reduced for the next step
        return self.module(*args, **kwargs)


def convert_to_ddp_model(model, local_rank=None, find_unused_parameters=True):
    return NativeDDP(model)
