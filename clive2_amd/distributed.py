"""Multi-GPU sample split: every rank renders its own samples of the replicated
scene; the accumulators -- pure sums (renderer.py:269-273) -- are combined by ONE sum all-reduce
of the packed planar buffer [8][H*W] float32 (66 MB at 1080p).  On GPUs the process group is
`nccl` (= RCCL over xGMI); the same code path runs with `gloo` on CPUs for tests.
"""
import os

import numpy as np


def rank_info():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def samples_for_rank(total_samples, rank, world_size):
    """Split `total_samples` over ranks; the first `total % world` ranks take one extra."""
    base, extra = divmod(int(total_samples), int(world_size))
    return base + (1 if rank < extra else 0)


def allreduce_packed_host(packed, group=None):
    """Sum a packed accumulator array over the group through host memory (gloo, or any backend
    that accepts CPU tensors).  Returns a new float32 numpy array."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(packed, dtype=np.float32).copy())
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy()


def radiance_from_packed(packed, height, width):
    """`summed_image / summed_sample_weights` (renderer.py:295-297) from the packed planar buffer."""
    a = np.asarray(packed, dtype=np.float32).reshape(8, height, width)
    img = np.moveaxis(a[0:3], 0, -1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.nan_to_num(img / a[3][..., None], neginf=0, posinf=0)
