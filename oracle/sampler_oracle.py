"""CPU oracle for the host side of the PMGT pre-training hot path (MCNSampling + batch assembly).

TEST INFRASTRUCTURE ONLY — see the header of `oracle/pmgt_oracle.py` for who may import this.

It restates pmgt/pmgt/datasets.py of the reference on a plain ordered-adjacency graph (no
networkx) and draws from numpy's process-global legacy `np.random` stream exactly where the
reference does, so for the same `np.random.seed` it returns bit-identical index tensors.
Parity status: PINNED by `tests/golden/sampler_*.npz` (generated from the reference).

The C++ sampler (`pmgt_amd/csrc/sampler.cpp`) re-implements the same draws without numpy
(own MT19937 + legacy `random_sample` / `choice` / `randint` / `permutation` algorithms,
SURVEY.md Appendix C) and is tested against this file and against the golden vectors.
"""
from __future__ import annotations

from collections import Counter
from typing import List, Sequence, Tuple

import numpy as np


class OrderedGraph:
    """Undirected weighted graph with node ids 2..N+1 whose neighbour lists keep insertion order,
    i.e. what `nx.Graph.add_edge` builds when scanning an ordered edge list."""

    def __init__(self, n_nodes: int, edges: Sequence[Tuple[int, int]], weights: Sequence[float]):
        self.n_nodes = int(n_nodes)
        self.adj: List[List[int]] = [[] for _ in range(n_nodes + 2)]
        self.w: List[List[float]] = [[] for _ in range(n_nodes + 2)]
        pos = {}
        for (u, v), wt in zip(edges, weights):
            u, v, wt = int(u), int(v), float(wt)
            if (u, v) in pos:            # nx.Graph.add_edge on an existing edge only updates data
                iu, iv = pos[(u, v)]
                self.w[u][iu] = wt
                self.w[v][iv] = wt
                continue
            self.adj[u].append(v)
            self.w[u].append(wt)
            if u != v:
                self.adj[v].append(u)
                self.w[v].append(wt)
            pos[(u, v)] = (len(self.adj[u]) - 1, len(self.adj[v]) - 1)
            pos[(v, u)] = (len(self.adj[v]) - 1, len(self.adj[u]) - 1)
        self._nbr_sets = [set(a) for a in self.adj]

    def csr(self):
        indptr = np.zeros(self.n_nodes + 3, dtype=np.int64)
        for i, a in enumerate(self.adj):
            indptr[i + 1] = indptr[i] + len(a)
        indices = np.array([v for a in self.adj for v in a], dtype=np.int64)
        wts = np.array([x for a in self.w for x in a], dtype=np.float64)
        return indptr, indices, wts


def softmax64(x: np.ndarray) -> np.ndarray:
    """scipy.special.softmax in float64: exp(x - max) / sum (pmgt/pmgt/datasets.py:27-29)."""
    e = np.exp(x - np.max(x))
    return e / np.sum(e)


def sample_context_neigh(g: OrderedGraph, target: int, hops: Sequence[int], max_ctx: int):
    """pmgt/pmgt/datasets.py:14-53 (_sample_context_neigh)."""
    depth = len(hops)
    scores = {}
    sampled = [[int(target)]] + [[] for _ in range(depth)]
    for k, size in enumerate(hops, start=1):
        for node in sampled[k - 1]:
            p = softmax64(np.asarray(g.w[node], dtype=np.float64))          # :27-29
            nbrs = np.asarray(g.adj[node], dtype=np.int64)
            sampled[k].extend(np.random.choice(nbrs, size=size, replace=True, p=p).tolist())  # :30-33
        for node, freq in Counter(sampled[k]).items():                      # :35-40
            if node == target:
                continue
            scores[node] = scores.get(node, 0) + freq * (depth - k + 1)
    if not scores:
        raise ValueError("target has no scored neighbour (pmgt/pmgt/datasets.py:42 raises)")
    ctx = [n for n, _ in sorted(scores.items(), key=lambda kv: kv[1], reverse=True)]  # stable, :42
    if len(ctx) < max_ctx:                                                  # :46-51
        num = len(ctx)
        ctx = ctx + [0] * (max_ctx - len(ctx))
    else:
        num = max_ctx
        ctx = ctx[:max_ctx]
    return ctx, num


def get_input_tensor(g: OrderedGraph, target: int, hops, max_ctx: int):
    """pmgt/pmgt/datasets.py:56-79 → (int64[S], float32[S])."""
    ctx, num = sample_context_neigh(g, target, hops, max_ctx)
    ids = np.array([int(target)] + ctx, dtype=np.int64)
    mask = np.zeros(max_ctx + 1, dtype=np.float32)
    mask[: num + 1] = 1
    return ids, mask


def dataset_getitem(g: OrderedGraph, target: int, max_ctx: int, hops=(16, 8, 4), max_total=10,
                    min_neg=5, is_training=True, is_inference=False):
    """PMGTDataset.__getitem__ (pmgt/pmgt/datasets.py:113-165) incl. _sample_neigh (:167-171)
    and _sample_neg (:173-180).  Draw order: target ctx → positives → their ctxs → negatives → ctxs."""
    tgt = get_input_tensor(g, target, hops, max_ctx)
    if is_inference:
        return (tgt,)
    k = (max_total - min_neg) if is_training else 1
    neigh = list(g.adj[target])
    pos = np.random.choice(neigh, min(k, len(neigh)), replace=False).tolist()
    pos_in = [get_input_tensor(g, n, hops, max_ctx) for n in pos]
    n_neg = max(min_neg, max_total - len(pos)) if is_training else 1
    neg = []
    for _ in range(n_neg):
        cand = np.random.randint(g.n_nodes) + 2
        while cand in g._nbr_sets[target]:
            cand = np.random.randint(g.n_nodes) + 2
        neg.append(cand)
    neg_in = [get_input_tensor(g, n, hops, max_ctx) for n in neg]
    ids = np.stack([a for a, _ in pos_in] + [a for a, _ in neg_in])
    msk = np.stack([b for _, b in pos_in] + [b for _, b in neg_in])
    labels = np.array([1.0] * len(pos_in) + [0.0] * len(neg_in), dtype=np.float32)
    return tgt, (ids, msk), labels


def collate(items):
    """pmgt_collate_fn (pmgt/pmgt/datasets.py:186-208) on numpy arrays."""
    tgt = {"node_ids": np.stack([b[0][0] for b in items]),
           "attention_mask": np.stack([b[0][1] for b in items])}
    if len(items[0]) == 1:
        return tgt
    pair = {"node_ids": np.concatenate([b[1][0] for b in items]),
            "attention_mask": np.concatenate([b[1][1] for b in items])}
    num_pairs = np.array([len(b[1][0]) for b in items], dtype=np.int64)
    labels = np.concatenate([b[2] for b in items])
    return tgt, pair, num_pairs, labels


def train_test_split_ids(n_nodes: int, valid_size: float, seed: int):
    """sklearn.model_selection.train_test_split(arange(2, N+2), test_size, random_state=seed)
    as used at pmgt/pmgt/trainer.py:45-52 (ShuffleSplit: permutation, first ceil(test*N) = valid)."""
    ids = np.arange(2, n_nodes + 2)
    n_test = int(np.ceil(valid_size * n_nodes))
    perm = np.random.RandomState(seed).permutation(n_nodes)
    return ids[perm[n_test:]], ids[perm[:n_test]]


def synth_graph(n_nodes: int, n_edges: int, seed: int):
    """Seeded synthetic item graph: ring (so no node is isolated; the reference raises on those,
    pmgt/pmgt/datasets.py:42) + random extra edges; weight = (ln c + 1)/(ln sqrt(deg_u deg_v) + 1)
    with co-review count c = 3 + Poisson(2) (notebooks/PMGT.ipynb cell 20).  Returns the ordered
    edge list (ids 2..N+1) and weights; deterministic via numpy's legacy RandomState."""
    rs = np.random.RandomState(seed)
    edges = [(i, (i + 1) % n_nodes) for i in range(n_nodes)]
    seen = set((min(a, b), max(a, b)) for a, b in edges)
    while len(edges) < n_edges:
        need = n_edges - len(edges)
        u = rs.randint(0, n_nodes, size=need * 2)
        v = rs.randint(0, n_nodes, size=need * 2)
        for a, b in zip(u.tolist(), v.tolist()):
            if a == b:
                continue
            key = (min(a, b), max(a, b))
            if key in seen:
                continue
            seen.add(key)
            edges.append((a, b))
            if len(edges) == n_edges:
                break
    e = np.array(edges, dtype=np.int64)
    deg = np.bincount(e.ravel(), minlength=n_nodes).astype(np.float64)
    c = 3 + rs.poisson(2.0, size=len(edges))
    w = (np.log(c) + 1.0) / (np.log(np.sqrt(deg[e[:, 0]] * deg[e[:, 1]])) + 1.0)
    return e + 2, w
