"""Item graph in CSR form with networkx-compatible neighbour ORDER (adjacency insertion order is part
of the sampling contract: it is the `a` array of np.random.choice)."""
from __future__ import annotations

import numpy as np


class CSRGraph:
    """Node ids 2..N+1 (0 = <pad>, 1 = <mask>; pmgt/pmgt/trainer.py:38-41). `indptr` has N+3 entries."""

    def __init__(self, n_nodes: int, indptr: np.ndarray, indices: np.ndarray, weights: np.ndarray):
        self.n_nodes = int(n_nodes)
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int64)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        assert self.indptr.shape[0] == self.n_nodes + 3

    def __len__(self):
        return self.n_nodes

    def degree(self, node: int) -> int:
        return int(self.indptr[node + 1] - self.indptr[node])

    def neighbors(self, node: int) -> np.ndarray:
        return self.indices[self.indptr[node]: self.indptr[node + 1]]

    def validate(self):
        """The reference crashes on isolated nodes (pmgt/pmgt/datasets.py:42); refuse them up front."""
        deg = np.diff(self.indptr)[2:]
        if (deg == 0).any():
            raise ValueError(f"{int((deg == 0).sum())} isolated node(s): PMGT sampling is undefined for them")

    @classmethod
    def from_edge_list(cls, n_nodes: int, edges: np.ndarray, weights: np.ndarray) -> "CSRGraph":
        """Same adjacency order as scanning `edges` with nx.Graph.add_edge (no duplicate / self edges)."""
        e = np.asarray(edges, dtype=np.int64)
        w = np.asarray(weights, dtype=np.float64)
        key = np.minimum(e[:, 0], e[:, 1]) * (n_nodes + 2) + np.maximum(e[:, 0], e[:, 1])
        if len(np.unique(key)) != len(key) or (e[:, 0] == e[:, 1]).any():
            raise ValueError("duplicate or self edges are not supported by the vectorised builder")
        src = np.stack([e[:, 0], e[:, 1]], 1).ravel()      # u0, v0, u1, v1, ...
        dst = np.stack([e[:, 1], e[:, 0]], 1).ravel()
        ww = np.repeat(w, 2)
        order = np.argsort(src, kind="stable")
        indptr = np.zeros(n_nodes + 3, dtype=np.int64)
        np.cumsum(np.bincount(src, minlength=n_nodes + 2), out=indptr[1:])
        return cls(n_nodes, indptr, dst[order], ww[order])

    @classmethod
    def from_networkx(cls, g) -> "CSRGraph":
        """From the relabelled nx.Graph the reference builds (ids 2..N+1, float `weight` per edge)."""
        n = g.number_of_nodes()
        indptr = np.zeros(n + 3, dtype=np.int64)
        idx, wts = [], []
        for v in range(2, n + 2):
            nb = g[v]
            idx.extend(nb.keys())
            wts.extend(d["weight"] for d in nb.values())
            indptr[v + 1] = len(idx)
        return cls(n, indptr, np.array(idx, dtype=np.int64), np.array(wts, dtype=np.float64))


def synthetic_graph(n_nodes: int, n_edges: int, seed: int = 0) -> CSRGraph:
    """Seeded synthetic item graph with the statistics of SURVEY.md §8(d): a ring (no isolated nodes)
    plus uniformly random extra edges; weight (ln c + 1) / (ln sqrt(deg_u deg_v) + 1) with co-review
    count c = 3 + Poisson(2) (notebooks/PMGT.ipynb cell 20)."""
    rs = np.random.RandomState(seed)
    ring = np.stack([np.arange(n_nodes), (np.arange(n_nodes) + 1) % n_nodes], 1)
    keys = set()
    lo, hi = np.minimum(ring[:, 0], ring[:, 1]), np.maximum(ring[:, 0], ring[:, 1])
    have = lo * n_nodes + hi
    extra = np.empty((0, 2), dtype=np.int64)
    need = n_edges - n_nodes
    while need > 0:
        u = rs.randint(0, n_nodes, size=int(need * 1.2) + 16)
        v = rs.randint(0, n_nodes, size=len(u))
        ok = u != v
        u, v = u[ok], v[ok]
        k = np.minimum(u, v) * n_nodes + np.maximum(u, v)
        _, first = np.unique(k, return_index=True)
        first.sort()
        u, v, k = u[first], v[first], k[first]
        new = ~np.isin(k, have)
        u, v, k = u[new][:need], v[new][:need], k[new][:need]
        extra = np.concatenate([extra, np.stack([u, v], 1)])
        have = np.concatenate([have, k])
        need = n_edges - n_nodes - len(extra)
    e = np.concatenate([ring, extra]).astype(np.int64)
    deg = np.bincount(e.ravel(), minlength=n_nodes).astype(np.float64)
    c = 3 + rs.poisson(2.0, size=len(e))
    w = (np.log(c) + 1.0) / (np.log(np.sqrt(deg[e[:, 0]] * deg[e[:, 1]])) + 1.0)
    return CSRGraph.from_edge_list(n_nodes, e + 2, w)


def synthetic_graph_regular(n_nodes: int, n_edges: int, seed: int = 0) -> CSRGraph:
    """Seeded synthetic item graph for MILLION-node workloads, built without a sort: a circulant graph -- every node u is joined to
    u +- o_j for n_edges / n_nodes seeded distinct offsets o_j (o_0 = 1: the ring) -- with the same weight formula as
    `synthetic_graph` (co-review count c = 3 + Poisson(2) per undirected edge).  Regular degree 2 n_edges / n_nodes instead of the
    Poisson-like degrees of the G(n, m) construction, same node and edge counts; seconds instead of two minutes at 10^6 / 2 x 10^7
    (`synthetic_graph` spends its time in four sorts of 2-4 x 10^7 keys).  Used by bench.py for the c4 / c5 shapes only."""
    k = n_edges // n_nodes
    assert k >= 1 and n_nodes > 4 * k, "regular construction: n_edges must be a small multiple of n_nodes"
    rs = np.random.RandomState(seed)
    offs = [1]
    while len(offs) < k:
        o = int(rs.randint(2, n_nodes // 2))
        if o not in offs:
            offs.append(o)
    offs = np.asarray(offs, dtype=np.int64)
    u = np.arange(n_nodes, dtype=np.int64)
    c = (3 + rs.poisson(2.0, size=(n_nodes, k))).astype(np.float64)      # c[u, j]: edge {u, u + o_j}
    wf = (np.log(c) + 1.0) / (np.log(2.0 * k) + 1.0)                     # sqrt(deg_u deg_v) = 2 k
    idx = np.empty((n_nodes, 2 * k), dtype=np.int64)
    w = np.empty((n_nodes, 2 * k), dtype=np.float64)
    for j in range(k):
        idx[:, 2 * j] = (u + offs[j]) % n_nodes + 2
        w[:, 2 * j] = wf[:, j]
        idx[:, 2 * j + 1] = (u - offs[j]) % n_nodes + 2
        w[:, 2 * j + 1] = np.roll(wf[:, j], offs[j])                     # the edge {u - o_j, u} as seen from u
    indptr = np.zeros(n_nodes + 3, dtype=np.int64)
    indptr[2:] = np.arange(n_nodes + 1, dtype=np.int64) * (2 * k)
    return CSRGraph(n_nodes, indptr, idx.ravel(), w.ravel())
