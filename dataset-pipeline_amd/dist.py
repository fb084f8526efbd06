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

    def allreduce(arr):
        if backend == "nccl":
            t = torch.from_numpy(arr).to(device if device is not None else "cuda")
            dist.all_reduce(t, group=group)
            arr[:] = t.cpu().numpy()
        else:
            t = torch.from_numpy(arr)          # shares memory with arr
            dist.all_reduce(t, group=group)
    return allreduce


def attach(icp, group=None, device=None):
    """Configure a PointToPlaneICP handle for the current process group."""
    import torch.distributed as dist
    icp.set_shard(dist.get_rank(group), dist.get_world_size(group), make_allreduce(group, device))
    return icp
