"""C++ MCNSampling (libpmgt_sampler.so) against numpy's legacy stream, the CPU oracle and the golden
vectors from the reference.  Index tensors must be BIT-EXACT.  CPU-only."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import sampler_oracle as so
from pmgt_amd import _lib
from pmgt_amd import datasets as pds
from pmgt_amd.datasets import (MODE_EVAL, MODE_INFERENCE, MODE_TRAIN, MCNSampler, PMGTDataset, get_input_tensor,
                               pmgt_collate_fn, train_valid_split)
from pmgt_amd.graph import CSRGraph, synthetic_graph
from tests import golden_util as gu


def csr(gname):
    n, edges, w = gu.graph(gname)
    return CSRGraph.from_edge_list(n, edges, w)


def test_legacy_stream_primitives_match_numpy():
    g = csr("A")
    s = MCNSampler(g, 5)
    for seed in (0, 1, 12345, 2 ** 32 - 1):
        s.seed(seed)
        np.random.seed(seed)
        a = [s.lib.pmgt_sampler_random_sample(s.h) for _ in range(700)]
        assert a == np.random.random_sample(700).tolist()
        for n in (1, 2, 7, 60, 7252, 10 ** 6, 2 ** 31 + 5):
            assert [s.lib.pmgt_sampler_randint(s.h, n) for _ in range(50)] == [int(np.random.randint(n)) for _ in range(50)]


def test_csr_matches_oracle_adjacency():
    for gname in "ABC":
        n, edges, w = gu.graph(gname)
        og = so.OrderedGraph(n, edges, w)
        g = CSRGraph.from_edge_list(n, edges, w)
        ip, ix, ww = og.csr()
        assert np.array_equal(ip, g.indptr) and np.array_equal(ix, g.indices) and np.array_equal(ww, g.weights)
        g.validate()


@pytest.mark.parametrize("gname", ["A", "B", "C"])
def test_sampler_bit_exact_vs_reference_golden(gname):
    gold = gu.load("sampler_" + gname)
    g = csr(gname)
    idx = gold["idx"]
    for S in (6, 16, 32):
        smp = MCNSampler(g, S - 1)
        for seed in (0, 1, 2):
            key = f"S{S}_seed{seed}_"
            smp.seed(seed)
            ctx = [smp.context(t) for t in range(2, 10)]
            assert np.array_equal(np.stack([c[0][1:] for c in ctx]), gold[key + "ctx"])
            assert np.array_equal(np.array([int(c[1].sum()) - 1 for c in ctx]), gold[key + "num_ctx"])
            for mode, nm in ((MODE_TRAIN, "train_"), (MODE_EVAL, "eval_")):
                smp.seed(seed)
                tgt, pair, num_pairs, labels = smp.batch(idx + 2, mode)
                assert np.array_equal(tgt["node_ids"].numpy(), gold[key + nm + "tgt_ids"])
                assert np.array_equal(tgt["attention_mask"].numpy(), gold[key + nm + "tgt_mask"])
                assert np.array_equal(pair["node_ids"].numpy(), gold[key + nm + "pair_ids"])
                assert np.array_equal(pair["attention_mask"].numpy(), gold[key + nm + "pair_mask"])
                assert np.array_equal(num_pairs.numpy(), gold[key + nm + "num_pairs"])
                assert np.array_equal(labels.numpy(), gold[key + nm + "labels"])
            smp.seed(seed)
            inf = smp.batch(idx + 2, MODE_INFERENCE)
            assert np.array_equal(inf["node_ids"].numpy(), gold[key + "inf_ids"])
            assert np.array_equal(inf["attention_mask"].numpy(), gold[key + "inf_mask"])


def test_dataset_surface_matches_reference_layout():
    """PMGTDataset / get_input_tensor / pmgt_collate_fn reproduce the reference's item tuples."""
    gold = gu.load("sampler_A")
    g = csr("A")
    ds = PMGTDataset(g, np.arange(2, 62), max_ctx_neigh=15)
    ds.seed(1)
    coll = pmgt_collate_fn([ds[int(i)] for i in gold["idx"]])
    assert np.array_equal(coll[0]["node_ids"].numpy(), gold["S16_seed1_train_tgt_ids"])
    assert np.array_equal(coll[1]["node_ids"].numpy(), gold["S16_seed1_train_pair_ids"])
    assert np.array_equal(coll[2].numpy(), gold["S16_seed1_train_num_pairs"])
    assert np.array_equal(coll[3].numpy(), gold["S16_seed1_train_labels"])
    assert coll[0]["node_ids"].dtype == torch.int64 and coll[0]["attention_mask"].dtype == torch.float32
    inf = PMGTDataset(g, max_ctx_neigh=15, is_training=False, is_inference=True)
    inf.seed(1)
    c2 = pmgt_collate_fn([inf[int(i)] for i in gold["idx"]])
    assert np.array_equal(c2["node_ids"].numpy(), gold["S16_seed1_inf_ids"])
    smp = MCNSampler(g, 15)
    smp.seed(0)
    ids, mask = get_input_tensor(smp, 2)
    assert ids[0] == 2 and ids.shape == (16,) and mask.shape == (16,)


def test_get_input_tensor_reference_call_shape_matches_reference_items():
    """get_input_tensor(graph, target, hop_sampling_sizes, max_num_ctx_neigh) -- the reference's own signature
    (pmgt/pmgt/datasets.py:64-69) as its second caller uses it (pmgt/pmgt_ncf/datasets.py:62): fixture G8 holds the item
    tensors the REFERENCE produced with np.random.seed(sseed) and successive 4-argument calls on one graph."""
    for name, (gname, _cfg, S, B, users, *_rest) in gu.NCF_CASES.items():
        sseed = gu.NCF_CASES[name][8]
        gold = gu.load(name)
        g = csr(gname)
        n = gu.GRAPHS[gname]["n"]
        pds.set_seed(sseed)                                # np.random.seed(sseed)
        items = np.random.RandomState(sseed + 50).choice(n, B, replace=False)
        pairs = [get_input_tensor(g, int(i) + 2, [16, 8, 4], S - 1) for i in items]
        assert all(p[0].dtype == torch.int64 and p[1].dtype == torch.float32 for p in pairs)
        assert np.array_equal(torch.stack([p[0] for p in pairs]).numpy(), gold["item_ids"])
        assert np.array_equal(torch.stack([p[1] for p in pairs]).numpy(), gold["item_mask"])
        # one stream per graph: a re-seed replays it, and a second configuration gets its own sampler
        pds.set_seed(sseed)
        again = get_input_tensor(g, int(items[0]) + 2, [16, 8, 4], S - 1)
        assert torch.equal(again[0], pairs[0][0])
        other = get_input_tensor(g, int(items[0]) + 2, [4, 2], 7)
        assert other[0].shape == (8,) and len(g._pmgt_samplers) == 2
    with pytest.raises(TypeError):
        get_input_tensor(g, 2)
    with pytest.raises(ValueError):
        get_input_tensor(MCNSampler(g, 15), 2, [16, 8, 4], 7)


def test_sampler_vs_oracle_on_larger_graph():
    """A 3 000-node graph with heavier degrees: C++ sampler == numpy oracle for whole training items."""
    g = synthetic_graph(3000, 40000, seed=5)
    og_edges = []
    n = g.n_nodes
    # rebuild the oracle graph from the CSR (insertion order is what the CSR stores)
    og = so.OrderedGraph.__new__(so.OrderedGraph)
    og.n_nodes = n
    og.adj = [g.indices[g.indptr[v]:g.indptr[v + 1]].tolist() for v in range(n + 2)]
    og.w = [g.weights[g.indptr[v]:g.indptr[v + 1]].tolist() for v in range(n + 2)]
    og._nbr_sets = [set(a) for a in og.adj]
    smp = MCNSampler(g, 31)
    targets = np.array([2, 17, 999, 3001, 1500], dtype=np.int64)
    smp.seed(7)
    tgt, pair, num_pairs, labels = smp.batch(targets, MODE_TRAIN)
    np.random.seed(7)
    ref = so.collate([so.dataset_getitem(og, int(t), 31) for t in targets])
    assert np.array_equal(tgt["node_ids"].numpy(), ref[0]["node_ids"])
    assert np.array_equal(pair["node_ids"].numpy(), ref[1]["node_ids"])
    assert np.array_equal(pair["attention_mask"].numpy(), ref[1]["attention_mask"])
    assert np.array_equal(labels.numpy(), ref[3])


def test_threaded_sampler_is_thread_count_invariant_and_valid():
    g = synthetic_graph(2000, 20000, seed=3)
    smp = MCNSampler(g, 15)
    targets = np.arange(2, 2 + 64, dtype=np.int64)
    a = smp.batch(targets, MODE_TRAIN, threads=1, base_seed=42, counter=100)
    b = smp.batch(targets, MODE_TRAIN, threads=4, base_seed=42, counter=100)
    for x, y in ((a[0]["node_ids"], b[0]["node_ids"]), (a[1]["node_ids"], b[1]["node_ids"]), (a[2], b[2]), (a[3], b[3])):
        assert torch.equal(x, y)
    c = smp.batch(targets, MODE_TRAIN, threads=4, base_seed=42, counter=164)
    assert not torch.equal(a[1]["node_ids"], c[1]["node_ids"])
    tgt, pair, num_pairs, labels = a
    assert torch.equal(tgt["node_ids"][:, 0], torch.from_numpy(targets))
    assert int(num_pairs.sum()) == pair["node_ids"].shape[0] == 640
    # positives are neighbours, negatives are not; context ids are valid nodes or padding
    off = 0
    for i, t in enumerate(targets):
        nb = set(g.neighbors(int(t)).tolist())
        for j in range(int(num_pairs[i])):
            node = int(pair["node_ids"][off + j, 0])
            assert (node in nb) == bool(labels[off + j] == 1)
        off += int(num_pairs[i])
    ids = pair["node_ids"].numpy()
    assert ((ids == 0) | ((ids >= 2) & (ids < g.n_nodes + 2))).all()
    assert ((pair["attention_mask"].numpy() == 0) == (ids == 0)).all()


def test_sampler_errors():
    g = csr("A")
    smp = MCNSampler(g, 5)
    with pytest.raises(ValueError):
        smp.context(1)            # <mask> id is not a node
    with pytest.raises(ValueError):
        smp.context(g.n_nodes + 2)
    iso = CSRGraph(3, np.array([0, 0, 0, 1, 2, 2]), np.array([3, 2]), np.array([1.0, 1.0]))
    with pytest.raises(ValueError):
        iso.validate()
    with pytest.raises(ValueError):
        MCNSampler(iso, 5).context(4)      # isolated node: the reference raises at datasets.py:42


def test_train_valid_split_matches_sklearn_golden():
    gold = gu.load("split")
    for n, vs, seed in ((301, 0.2, 0), (7252, 0.2, 0), (10834, 0.1, 3)):
        tr, va = train_valid_split(n, vs, seed)
        k = f"n{n}_v{vs}_s{seed}_"
        assert list(gold[k + "sizes"]) == [len(tr), len(va)]
        assert hashlib.sha256(tr.tobytes()).digest() == gold[k + "train_sha"].tobytes()
        assert hashlib.sha256(va.tobytes()).digest() == gold[k + "valid_sha"].tobytes()


def test_capi_symbols_exported():
    """Both shared objects load and export every symbol include/pmgt_capi.h and include/pmgt_ops.h declare (no compute here)."""
    import ctypes
    import os
    import re
    from pmgt_amd import _build
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    decl = lambda f: set(re.findall(r"\b(pmgt_[a-z0-9_]+)\s*\(", open(os.path.join(inc, f)).read()))
    pub, ops = decl("pmgt_capi.h"), decl("pmgt_ops.h")
    assert pub == set(_lib.HIP_SYMBOLS) | set(_lib.SAMPLER_SYMBOLS), pub ^ (set(_lib.HIP_SYMBOLS) | set(_lib.SAMPLER_SYMBOLS))
    assert ops == set(_lib.OPS_SYMBOLS), ops ^ set(_lib.OPS_SYMBOLS)
    assert not [s for s in pub | ops if "debug" in s]            # no process-global switches: options are per engine
    hip = ctypes.CDLL(_build.hip_lib_path())
    smp = ctypes.CDLL(_build.sampler_lib_path())
    for s in _lib.HIP_SYMBOLS + _lib.OPS_SYMBOLS:
        assert hasattr(hip, s), s
    for s in _lib.SAMPLER_SYMBOLS:
        assert hasattr(smp, s), s
    assert _lib.hip().pmgt_abi_version() == 4      # 4: n modalities (pmgt_config.feat_sizes[], pmgt_tensors.tables[], feats arrays)


def test_strided_counters_reproduce_the_single_process_streams():
    """evaluate(distributed=True): rank r samples items r, r + W, ... of the node list with counter = r + W * offset and
    counter_stride = W, i.e. every item draws from the stream of its GLOBAL index -- the rows a single process draws."""
    from pmgt_amd.datasets import MODE_EVAL
    smp = MCNSampler(csr("A"), 15)
    nodes = np.arange(2, 2 + 37)
    one = smp.batch(nodes, MODE_EVAL, threads=3, base_seed=11, counter=5)
    W = 3
    for r in range(W):
        part = smp.batch(nodes[r::W], MODE_EVAL, threads=2, base_seed=11, counter=5 + r, counter_stride=W)
        assert torch.equal(part[0]["node_ids"], one[0]["node_ids"][r::W])
        assert torch.equal(part[0]["attention_mask"], one[0]["attention_mask"][r::W])
        ends = np.cumsum(one[2].numpy())
        rows = np.concatenate([np.arange(e - n, e) for e, n in zip(ends[r::W], one[2].numpy()[r::W])])
        assert torch.equal(part[1]["node_ids"], one[1]["node_ids"][rows]) and torch.equal(part[3], one[3][rows])
