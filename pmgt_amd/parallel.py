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


_AVG_OK = None


def backend_averages(device) -> bool:
    """Whether the process group reduces with ReduceOp.AVG.  gloo has none; on "nccl" (= RCCL on ROCm) it is tried ONCE with a one-element
    collective -- every rank reaches this at the same point of its first exchange -- and a build that refuses the op (PyTorch raises before
    anything is sent) falls back to SUM and a division, like gloo.  RCCL has never run this code in the dev loop: the driver's multi-GPU
    tier must not die on a capability assumption."""
    global _AVG_OK
    import torch.distributed as dist
    if dist.get_backend() != "nccl":
        return False
    if _AVG_OK is None:
        try:
            dist.all_reduce(torch.ones(1, device=device), op=dist.ReduceOp.AVG)
            _AVG_OK = True
        except (RuntimeError, ValueError, NotImplementedError):
            _AVG_OK = False
    return _AVG_OK


def allreduce_mean_(flat: torch.Tensor, force: bool = False) -> torch.Tensor:
    """In-place average of a flat buffer over all ranks (DDP gradient semantics).  force: issue the collective on a one-rank group too."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1 and not (force and dist.is_initialized()):
        return flat
    if backend_averages(flat.device):
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:                       # gloo has no AVG
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(ws)
    return flat


class BucketedAllReduce:
    """The overlapped form of the exchange (DDP's bucketed overlap, pmgt/base_trainer.py:309-322):
    `bucket_ready(offset, numel)` is the engine's gradient-ready hook -- it is called in stream order as soon as a
    contiguous range of the flat gradient buffer is final (NFR head, layers L-1 .. 0, embeddings) and starts an
    ASYNCHRONOUS all-reduce of that slice.  On RCCL the collective runs on the process group's own stream, ordered after
    the launches enqueued so far, next to the rest of the backward pass; `wait()` orders the caller's stream after all of
    them (no host block on RCCL) and returns the number of elements exchanged so the caller can check coverage."""

    def __init__(self, flat: torch.Tensor, boundaries=()):
        """boundaries: offsets into the flat buffer at which a collective is issued.  () = one collective per engine bucket
        (NFR head, each layer, embeddings: 0.8 - 3 MB each at L4 / d256).  The engine reports ranges in DESCENDING offset
        order; adjacent ranges are coalesced and the coalesced range is sent as soon as its low end reaches a boundary (or
        offset 0), so `boundaries = (offset of layer 0,)` gives TWO collectives -- NFR head + encoder layers, started while
        the embedding backward still runs, then the embeddings -- latency-sized messages merged into bandwidth-sized ones."""
        self.flat = flat
        self.enabled = True
        self.boundaries = frozenset(int(b) for b in boundaries)
        self._pending = []
        self._elems = 0
        self._held = None            # (offset, numel) coalesced, not yet sent
        self.sent = []               # (offset, numel) of the collectives of the current step, in issue order

    def _send(self, offset: int, numel: int):
        import torch.distributed as dist
        sl = self.flat[offset: offset + numel]
        avg = backend_averages(sl.device)
        work = dist.all_reduce(sl, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True)
        self._pending.append((work, sl, avg))
        self._elems += numel
        self.sent.append((offset, numel))

    def bucket_ready(self, offset: int, numel: int):
        if not self.enabled:
            return
        if not self.boundaries:
            self._send(offset, numel)
            return
        if self._held is not None and offset + numel == self._held[0]:
            self._held = (offset, numel + self._held[1])              # grows downwards
        else:
            if self._held is not None:                                # not adjacent: what is held goes out as it is
                self._send(*self._held)
            self._held = (offset, numel)
        if self._held[0] == 0 or self._held[0] in self.boundaries:
            self._send(*self._held)
            self._held = None

    def wait(self) -> int:
        ws = world()[1]
        if self._held is not None:
            self._send(*self._held)
            self._held = None
        self.last_sent, self.sent = self.sent, []
        for work, sl, avg in self._pending:
            work.wait()
            if not avg:
                sl.div_(ws)
        n, self._elems = self._elems, 0
        self._pending.clear()
        return n


def gather_predictions(preds: np.ndarray, labels: np.ndarray):
    """Validation across ranks (the reference logs a per-rank AUC, pmgt/pmgt/trainer.py:182-195 without
    sync_dist; here every rank gets the predictions / labels of ALL shards, rank order, so one AUC is reported)."""
    import torch.distributed as dist
    if world()[1] == 1:
        return preds, labels
    parts = [None] * world()[1]
    dist.all_gather_object(parts, (np.asarray(preds), np.asarray(labels)))
    return np.concatenate([p for p, _ in parts]), np.concatenate([l for _, l in parts])


def broadcast_(flat: torch.Tensor, src: int = 0, force: bool = False) -> torch.Tensor:
    import torch.distributed as dist
    if world()[1] > 1 or (force and dist.is_initialized()):
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
