"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

The C library shards every directed pair's source cloud into `world_size` slices (e3d_icp_set_shard) and calls back
into the host language to sum its small f64 buffers -- the 6(n-1) x 6(n-1) normal equations + cost once per LM pass,
and the per-pair correspondence counts once per outer iteration.  This module provides that callback.
"""
import numpy as np


def shard_slice(n, rank, world):
    """[begin, end) of rank's slice of n items -- same formula as the C library (e3d_icp.hip, align_meshes)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def make_allreduce(group=None, device=None):
    """Returns f(np.ndarray float64) that sums the array in place across the process group."""
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    if backend == "nccl" and device is None:
        # one process per GPU: the staging tensor must live on THIS rank's GPU ("cuda" alone is cuda:0 unless the caller called
        # torch.cuda.set_device; every rank on cuda:0 makes RCCL fail with a duplicate-GPU error)
        device = _rank_device()

    def allreduce(arr):
        if backend == "nccl":
            t = torch.from_numpy(arr).to(device)
            dist.all_reduce(t, group=group)
            arr[:] = t.cpu().numpy()
        else:
            t = torch.from_numpy(arr)          # shares memory with arr
            dist.all_reduce(t, group=group)
    return allreduce


def _rank_device():
    """The GPU of this rank: torch's current device if the caller selected one, else LOCAL_RANK."""
    import os
    import torch
    cur = torch.cuda.current_device()
    local = int(os.environ.get("LOCAL_RANK", cur))
    if cur == 0 and local != 0 and local < torch.cuda.device_count():
        cur = local                      # nobody called torch.cuda.set_device: follow the launcher's LOCAL_RANK
    return torch.device("cuda", cur)


def attach(icp, group=None, device=None):
    """Configure a PointToPlaneICP handle for the current process group."""
    import torch.distributed as dist
    # a group of one rank takes the unsharded path (a callback with a world of 1 is the library's "tap": every reduction of a
    # single-rank run would make a host round trip through it for nothing -- ADVICE round 5)
    world = dist.get_world_size(group)
    icp.set_shard(dist.get_rank(group), world, make_allreduce(group, device) if world > 1 else None)
    return icp


# ---- path (B): image sharding ---------------------------------------------------------------------------------------------
class _DeviceView:
    """Zero-copy view of a device buffer of the C library for torch (`torch.as_tensor` reads __cuda_array_interface__)."""

    def __init__(self, ptr, count, dtype):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4" if dtype == 0 else "<i4",
                                         "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, count, dtype, device=None):
    import torch
    return torch.as_tensor(_DeviceView(ptr, count, dtype), device=device if device is not None else _rank_device())


def make_allreduce_device(group=None, device=None):
    """Returns f(ptr, count, dtype) that sums a device buffer of the library in place across the process group: RCCL straight
    from HBM with the "nccl" backend (the descriptor exchange of the colour update is the one bandwidth-relevant collective
    of path (B): K*N f32 + N i32 per point scale over xGMI); staged through the host for "gloo"."""
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)

    def allreduce(ptr, count, dtype):
        if count == 0:
            return
        t = device_tensor(ptr, count, dtype, device)
        if backend == "nccl":
            dist.all_reduce(t, group=group)
        else:
            c = t.cpu()
            dist.all_reduce(c, group=group)
            t.copy_(c)
        torch.cuda.synchronize()
    return allreduce


def image_owner(image_id, world):
    """Rank that owns an image -- same rule as the C library (e3d_reg_image_owner)."""
    return image_id % world


def attach_reg(problem, group=None, device=None):
    """Configure a RegProblem for the current process group (call before set_image)."""
    import torch.distributed as dist
    problem.set_shard(dist.get_rank(group), dist.get_world_size(group), make_allreduce(group, device), make_allreduce_device(group, device))
    return problem
