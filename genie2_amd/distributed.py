"""The N>1 path of bench.py / multi-GPU sampling with torch.distributed
(RCCL over xGMI on GPUs, gloo on CPU for tests).  Structures are independent:
ranks never exchange data on the denoising path; these helpers cover what is
left -- the static task split, the timing reduction and the trivial result
gather (24.6 KB per rank at N=256, batch=8)."""
import torch
import torch.distributed as td

from .multiprocessor import split_tasks


def rank_tasks(tasks, rank=None, world=None):
    rank = td.get_rank() if rank is None else rank
    world = td.get_world_size() if world is None else world
    return split_tasks(tasks, world)[rank]


def max_over_ranks(seconds, device='cpu'):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def gather_coordinates(trans):
    """all_gather of the final C-alpha coordinates [B,N,3] -> [world*B,N,3]."""
    out = [torch.empty_like(trans) for _ in range(td.get_world_size())]
    td.all_gather(out, trans.contiguous())
    return torch.cat(out, dim=0)
