"""Host side of the hot path with the reference's names and semantics (pmgt/pmgt/datasets.py):
`get_input_tensor`, `PMGTDataset`, `pmgt_collate_fn` — backed by the C++ sampler
(pmgt_amd/csrc/sampler.cpp) instead of networkx + numpy.

Randomness: like the reference, sampling consumes ONE sequential legacy-numpy-compatible stream per
sampler (seed it with `seed()`, the analogue of np.random.seed in pmgt/utils/base.py:37); drawing
items in dataset order reproduces the reference run with num_workers=0 bit for bit.
`BatchSampler` is the throughput path: multi-threaded, per-target counter-derived seeds, writes
straight into pinned host buffers.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from .graph import CSRGraph

MODE_TRAIN, MODE_EVAL, MODE_INFERENCE = 0, 1, 2


def _p(a):
    return C.c_void_p(0 if a is None else a.ctypes.data)


class MCNSampler:
    """Handle on the native sampler for one (graph, hop sizes, context length) configuration."""

    def __init__(self, graph: CSRGraph, max_ctx_neigh: int = 5, hop_sampling_sizes: Sequence[int] = (16, 8, 4),
                 max_total_samples: int = 10, min_neg_samples: int = 5):
        self.lib = _lib.sampler()
        self.graph = graph
        self.S = max_ctx_neigh + 1
        self.hops = tuple(int(h) for h in hop_sampling_sizes)
        hops = np.asarray(list(hop_sampling_sizes), dtype=np.int32)
        self.h = self.lib.pmgt_sampler_create(graph.n_nodes, _p(graph.indptr), _p(graph.indices), _p(graph.weights),
                                              _p(hops), len(hops), max_ctx_neigh, max_total_samples, min_neg_samples)
        if not self.h:
            raise ValueError(self.lib.pmgt_sampler_last_error().decode())

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.pmgt_sampler_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise ValueError(self.lib.pmgt_sampler_last_error().decode())
        return rc

    def seed(self, seed: int):
        self.lib.pmgt_sampler_seed(self.h, int(seed) & 0xFFFFFFFF)

    def max_pairs(self, mode: int) -> int:
        return self.lib.pmgt_sampler_max_pairs(self.h, mode)

    def context(self, target: int) -> Tuple[np.ndarray, np.ndarray]:
        ids = np.empty(self.S, dtype=np.int64)
        mask = np.empty(self.S, dtype=np.float32)
        self._check(self.lib.pmgt_sampler_context(self.h, int(target), _p(ids), _p(mask)))
        return ids, mask

    def alloc(self, n: int, mode: int, pinned: bool = False):
        mp = max(self.max_pairs(mode), 1)
        mk = lambda shape, dt: (torch.empty(shape, dtype=dt).pin_memory() if pinned else torch.empty(shape, dtype=dt))
        return dict(tgt_ids=mk((n, self.S), torch.int64), tgt_mask=mk((n, self.S), torch.float32),
                    pair_ids=mk((n * mp, self.S), torch.int64), pair_mask=mk((n * mp, self.S), torch.float32),
                    num_pairs=mk((n,), torch.int64), labels=mk((n * mp,), torch.float32))

    def batch(self, targets: np.ndarray, mode: int = MODE_TRAIN, out=None, threads: int = 0, base_seed: int = 0,
              counter: int = 0, counter_stride: int = 1):
        """Sample one collated batch.  threads == 0: the sequential reference-order stream;
        threads >= 1: per-target seeded streams on that many host threads (target i: stream counter + counter_stride * i)."""
        targets = np.ascontiguousarray(targets, dtype=np.int64)
        n = len(targets)
        buf = out if out is not None else self.alloc(n, mode)
        tp = lambda k: C.c_void_p(buf[k].data_ptr())
        args = (tp("tgt_ids"), tp("tgt_mask"), tp("pair_ids"), tp("pair_mask"), tp("num_pairs"), tp("labels"))
        if threads <= 0:
            tot = self._check(self.lib.pmgt_sampler_batch(self.h, _p(targets), n, mode, *args))
        else:
            tot = self._check(self.lib.pmgt_sampler_batch_mt(self.h, _p(targets), n, mode, base_seed, counter, counter_stride, threads, *args))
        tgt = {"node_ids": buf["tgt_ids"][:n], "attention_mask": buf["tgt_mask"][:n]}
        if mode == MODE_INFERENCE:
            return tgt
        pair = {"node_ids": buf["pair_ids"][:tot], "attention_mask": buf["pair_mask"][:tot]}
        return tgt, pair, buf["num_pairs"][:n], buf["labels"][:tot]


_DEFAULT_SEED = 0


def set_seed(seed: int):
    """np.random.seed analogue (pmgt/utils/base.py:37) for the samplers `get_input_tensor` creates behind a graph:
    re-seeds the ones that exist and seeds the ones created later."""
    global _DEFAULT_SEED
    _DEFAULT_SEED = int(seed)
    for g in list(_GRAPHS_WITH_SAMPLERS):
        for smp in g._pmgt_samplers.values():
            smp.seed(seed)


_GRAPHS_WITH_SAMPLERS: list = []


def _sampler_for(graph: CSRGraph, hop_sampling_sizes: Sequence[int], max_num_ctx_neigh: int) -> MCNSampler:
    cache = graph.__dict__.setdefault("_pmgt_samplers", {})
    key = (tuple(int(h) for h in hop_sampling_sizes), int(max_num_ctx_neigh))
    smp = cache.get(key)
    if smp is None:
        smp = cache[key] = MCNSampler(graph, max_num_ctx_neigh, key[0])
        smp.seed(_DEFAULT_SEED)
        if not any(g is graph for g in _GRAPHS_WITH_SAMPLERS):
            _GRAPHS_WITH_SAMPLERS.append(graph)
    return smp


def get_input_tensor(graph, target_node: int, hop_sampling_sizes: Optional[Sequence[int]] = None,
                     max_num_ctx_neigh: Optional[int] = None) -> Tuple[torch.LongTensor, torch.FloatTensor]:
    """pmgt/pmgt/datasets.py:64-79 -- (LongTensor [S] = [target] + context, FloatTensor [S] mask), same four arguments
    (second caller: pmgt/pmgt_ncf/datasets.py:62).  `graph` is a CSRGraph; the native sampler of that (graph, hop sizes,
    context length) is created on first use and kept on the graph, so successive calls continue ONE sequential stream as
    the reference's global np.random does (seed it with `set_seed`).  An MCNSampler may be passed instead of the graph
    (then the last two arguments are optional and, if given, must agree with it)."""
    if isinstance(graph, MCNSampler):
        sampler = graph
        if max_num_ctx_neigh is not None and max_num_ctx_neigh + 1 != sampler.S:
            raise ValueError(f"max_num_ctx_neigh={max_num_ctx_neigh} does not match the sampler's context length {sampler.S - 1}")
        if hop_sampling_sizes is not None and tuple(hop_sampling_sizes) != tuple(sampler.hops):
            raise ValueError(f"hop_sampling_sizes={list(hop_sampling_sizes)} do not match the sampler's {list(sampler.hops)}")
    else:
        if hop_sampling_sizes is None or max_num_ctx_neigh is None:
            raise TypeError("get_input_tensor(graph, target_node, hop_sampling_sizes, max_num_ctx_neigh)")
        sampler = _sampler_for(graph, hop_sampling_sizes, max_num_ctx_neigh)
    ids, mask = sampler.context(target_node)
    assert len(ids) == sampler.S, f"# of context nodes must be {sampler.S - 1}"
    return torch.from_numpy(ids), torch.from_numpy(mask)


class PMGTDataset(torch.utils.data.Dataset):
    """Same constructor and item layout as the reference's PMGTDataset (pmgt/pmgt/datasets.py:82-183);
    `graph` is a CSRGraph.  Items: ((ids, mask), (pair_ids, pair_mask), labels), or ((ids, mask),) in
    inference mode."""

    def __init__(self, graph: CSRGraph, node_ids: Optional[np.ndarray] = None, max_ctx_neigh: int = 5,
                 hop_sampling_sizes: List[int] = [16, 8, 4], max_total_samples: int = 10, min_neg_samples: int = 5,
                 is_training: bool = True, is_inference: bool = False) -> None:
        super().__init__()
        self.graph = graph
        self.node_ids = node_ids if node_ids is not None else np.arange(start=2, stop=len(graph) + 2)
        self.max_num_ctx_neigh = max_ctx_neigh
        self.hop_sampling_sizes = hop_sampling_sizes
        self.max_total_samples = max_total_samples
        self.min_neg_samples = min_neg_samples
        self.is_training = is_training
        self.is_inference = is_inference
        self.sampler = MCNSampler(graph, max_ctx_neigh, hop_sampling_sizes, max_total_samples, min_neg_samples)

    @property
    def mode(self) -> int:
        return MODE_INFERENCE if self.is_inference else (MODE_TRAIN if self.is_training else MODE_EVAL)

    def seed(self, seed: int):
        self.sampler.seed(seed)

    def __len__(self) -> int:
        return len(self.node_ids)

    def __getitem__(self, idx: int):
        res = self.sampler.batch(np.array([self.node_ids[idx]]), self.mode)
        if self.is_inference:
            return ((res["node_ids"][0], res["attention_mask"][0]),)
        tgt, pair, _, labels = res
        return (tgt["node_ids"][0], tgt["attention_mask"][0]), (pair["node_ids"], pair["attention_mask"]), labels


def pmgt_collate_fn(batch: Iterable[Tuple[torch.Tensor, ...]]) -> Union[Dict[str, torch.Tensor], tuple]:
    """pmgt/pmgt/datasets.py:186-208."""
    batch = list(batch)
    target_inputs = {"node_ids": torch.stack([b[0][0] for b in batch]),
                     "attention_mask": torch.stack([b[0][1] for b in batch])}
    if len(batch[0]) == 1:
        return target_inputs
    pair_inputs = {"node_ids": torch.cat([b[1][0] for b in batch]),
                   "attention_mask": torch.cat([b[1][1] for b in batch])}
    num_pairs = torch.LongTensor([len(b[1][0]) for b in batch])
    labels = torch.cat([b[2] for b in batch])
    return target_inputs, pair_inputs, num_pairs, labels


def train_valid_split(n_nodes: int, valid_size: float, seed: int):
    """train_test_split(arange(2, N+2), test_size=valid_size, random_state=seed) of
    pmgt/pmgt/trainer.py:45-52, without sklearn."""
    n_test = int(np.ceil(valid_size * n_nodes))
    tr = np.empty(n_nodes - n_test, dtype=np.int64)
    va = np.empty(n_test, dtype=np.int64)
    rc = _lib.sampler().pmgt_train_valid_split(n_nodes, float(valid_size), int(seed) & 0xFFFFFFFF, _p(tr), _p(va))
    assert rc == n_test
    return tr, va
