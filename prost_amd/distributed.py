"""Multi-GPU batches: independent problems sharded one (or more) per rank, with a GLOBAL stopping
criterion -- the four residual sums (sum diff_p^2, sum z^2, sum diff_d^2, sum w^2) are all-reduced
so every rank takes the same step-size / stopping decisions (the reference itself is
single-GPU).  No data-path collective: iterates never leave their GPU.
"""
import numpy as np


def shard(num_problems, rank, world):
    """problem ids owned by `rank`: contiguous blocks, sizes differing by at most one."""
    base, extra = divmod(num_problems, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def problem_seed(problem_id, base_seed=42):
    """BASELINE config 5: seeds 42..49 for the eight 4096^2 problems"""
    return base_seed + problem_id


def init_native_comm(dist, device):
    """Create the RCCL communicator owned by the native solver (prost_hip_comm_create): rank 0
    makes the ncclUniqueId, torch.distributed broadcasts its 128 bytes."""
    import torch

    from . import _capi
    ident = torch.zeros(128, dtype=torch.float64, device=device)
    if dist.get_rank() == 0:
        ident.copy_(torch.from_numpy(_capi.comm_unique_id()))
    dist.broadcast(ident, src=0)
    _capi.comm_init(ident.cpu().numpy(), dist.get_rank(), dist.get_world_size())


def allreduce_hook(dist):
    """host-side hook with the same contract (sum 4 doubles in place) for CPU/gloo runs"""
    import torch

    def hook(v4):
        t = torch.from_numpy(np.array(v4, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()
    return hook
