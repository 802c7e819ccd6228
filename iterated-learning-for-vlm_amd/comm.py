"""Data-parallel exchange steps of the contrastive path over torch.distributed (backend 'nccl' = RCCL over
xGMI on MI355X; 'gloo' in the CPU tests).  Device-agnostic plumbing: no arithmetic besides the collectives.

Reference behaviour restated (file:line relative to the reference root):
  AllGather.forward   prototype/model/clip_fdt.py:166-178 -- dist.all_gather of [B,D] twice per step (image, text)
  AllGather.backward  prototype/model/clip_fdt.py:180-188 -- all_reduce(SUM) of the full [W,B,D] gradient, then [rank]
  DDP gradient mean   prototype/utils/torch_ddp_dist.py:52-67
Here: ONE fused all-gather of [2,B,D] forward, ONE reduce-scatter of [W,2,B,D] backward (mathematically the
all-reduce + slice of the reference at 1/W of the traffic), and chunked all-reduce(mean) of the flat gradient
arena on a side stream so the text-tower gradients travel while the vision tower is still in backward.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def gather_pair(img, txt):
    """[B,D] x2 -> rank-major [W*B,D] x2 (own rows included)."""
    rank, W = world()
    if W == 1:
        return img, txt
    B, D = img.shape
    send = torch.stack([img, txt], 0).contiguous()                  # [2,B,D]
    recv = torch.empty((W, 2, B, D), dtype=img.dtype, device=img.device)
    dist.all_gather_into_tensor(recv.view(W * 2 * B, D), send.view(2 * B, D))
    return recv[:, 0].reshape(W * B, D).contiguous(), recv[:, 1].reshape(W * B, D).contiguous()


def reduce_gathered(dg_img, dg_txt, B):
    """Gradients of the gathered matrices [W*B,D] x2 -> this rank's [B,D] slices summed over ranks."""
    rank, W = world()
    if W == 1:
        return dg_img, dg_txt
    D = dg_img.shape[1]
    send = torch.stack([dg_img.view(W, B, D), dg_txt.view(W, B, D)], 1).contiguous()   # [W,2,B,D]
    if dist.get_backend() == "nccl":
        out = torch.empty((2, B, D), dtype=send.dtype, device=send.device)
        dist.reduce_scatter_tensor(out.view(2 * B, D), send.view(W * 2 * B, D), op=dist.ReduceOp.SUM)
        return out[0], out[1]
    dist.all_reduce(send, op=dist.ReduceOp.SUM)       # gloo has no reduce-scatter: reference formulation
    return send[rank, 0].contiguous(), send[rank, 1].contiguous()


class GradReducer:
    """Mean all-reduce of ranges of a flat gradient buffer on a dedicated communication stream."""

    def __init__(self, flat):
        self.flat = flat
        self.rank, self.W = world()
        self.cuda = flat.is_cuda
        self.stream = torch.cuda.Stream(device=flat.device) if self.cuda else None
        self.pending = False

    def reduce_range(self, begin, end, chunk_elems=32 * 1024 * 1024):
        """Enqueue all-reduce(mean) of flat[begin:end]; the producer stream's work so far is waited for."""
        if self.W == 1 or end <= begin:
            return
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            ctx = torch.cuda.stream(self.stream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            for a in range(begin, end, chunk_elems):
                piece = self.flat[a:min(end, a + chunk_elems)]
                if dist.get_backend() == "nccl":
                    dist.all_reduce(piece, op=dist.ReduceOp.AVG)
                else:
                    dist.all_reduce(piece, op=dist.ReduceOp.SUM)
                    piece.div_(self.W)
        self.pending = True

    def wait(self):
        """Make the current stream wait for every enqueued reduction (call before the optimizer reads gradients)."""
        if self.pending and self.cuda:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        self.pending = False
