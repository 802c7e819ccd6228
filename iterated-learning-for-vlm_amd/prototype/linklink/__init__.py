"""Mirror of reference prototype/linklink/__init__.py:13-40: env-based rank helpers and thin
torch.distributed aliases (RCCL on MI355X: backend string 'nccl')."""
import os

import torch
import torch.distributed as dist

allreduce = dist.all_reduce
allgather = dist.all_gather
broadcast = dist.broadcast
synchronize = torch.cuda.synchronize
init_process_group = dist.init_process_group


def get_rank():
    return int(os.environ.get("RANK", 0))


def get_world_size():
    return int(os.environ.get("WORLD_SIZE", 1))


def get_local_rank():
    return int(os.environ.get("LOCAL_RANK", 0))


def barrier():
    """The reference all-reduces one int and copies it to the host (linklink/__init__.py:30-34); collectives on
    one HIP stream are already ordered, so this is only a rendezvous for host-side effects (checkpoint files)."""
    if get_world_size() > 1 and dist.is_initialized():
        dist.barrier()


def finalize():
    pass
