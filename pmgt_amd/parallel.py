"""Data-parallel plumbing for the one exchange step of PMGT pre-training.

The reference gets all of this implicitly from PyTorch-Lightning (`pl.Trainer(gpus=N)` -> DDP +
DistributedSampler, pmgt/base_trainer.py:309-322).  Here it is explicit and tiny: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests), ONE
all-reduce of the flat gradient buffer per optimizer step, one parameter broadcast at start, and a
strided shard of one seeded permutation per epoch.
"""
from __future__ import annotations

import numpy as np
import torch


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def allreduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """In-place average of a flat buffer over all ranks (DDP gradient semantics)."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return flat
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:                       # gloo has no AVG
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(ws)
    return flat


def broadcast_(flat: torch.Tensor, src: int = 0) -> torch.Tensor:
    import torch.distributed as dist
    if world()[1] > 1:
        dist.broadcast(flat, src=src)
    return flat


def shard_indices(n: int, rank: int, world_size: int, seed: int = 0, epoch: int = 0, shuffle: bool = True,
                  drop_last: bool = False) -> np.ndarray:
    """torch.utils.data.DistributedSampler semantics (what PL injects for the train loader): one
    permutation seeded by seed + epoch, padded by wrap-around to a multiple of world_size, rank takes
    indices rank, rank + W, rank + 2W, ..."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).numpy()
    else:
        idx = np.arange(n)
    if drop_last and n % world_size:
        idx = idx[: n - n % world_size]
    else:
        total = -(-len(idx) // world_size) * world_size
        pad = total - len(idx)
        if pad:
            reps = -(-pad // max(len(idx), 1))
            idx = np.concatenate([idx, np.tile(idx, reps)[:pad]])
    return idx[rank::world_size]
