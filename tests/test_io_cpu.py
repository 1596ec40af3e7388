"""On-disk formats: dataset directory -> CSR in the reference's adjacency order,
embedding remap.  CPU only; the reference side of each comparison is networkx / sklearn themselves
(what pmgt/pmgt/trainer.py:30-41 and pmgt/pmgt/utils.py:15-40 call)."""
import os

import numpy as np
import pytest

from oracle import sampler_oracle as so
from pmgt_amd import io as pio
from pmgt_amd.datasets import MCNSampler

nx = pytest.importorskip("networkx")


def labelled_graph(n=40, e=120, seed=5):
    """nx.Graph over shuffled STRING labels, edges inserted in a label order unrelated to the encoder's order."""
    rs = np.random.RandomState(seed)
    labels = np.array(["item_%03d" % i for i in rs.permutation(n)])
    edges, w = so.synth_graph(n, e, seed)
    g = nx.Graph()
    g.add_nodes_from(labels[rs.permutation(n)])
    for (u, v), wt in zip(edges.tolist(), w.tolist()):
        g.add_edge(labels[u - 2], labels[v - 2], weight=wt)
    return g, np.sort(labels)


def test_dataset_dir_roundtrip_keeps_reference_adjacency_order(tmp_path):
    g, classes = labelled_graph()
    n = g.number_of_nodes()
    vis = np.random.RandomState(1).standard_normal((n + 2, 12)).astype(np.float32)
    txt = np.random.RandomState(2).standard_normal((n + 2, 8)).astype(np.float32)
    pio.save_dataset_dir(str(tmp_path), g, classes, vis, txt)
    csr, cls2 = pio.load_graph(str(tmp_path))
    assert list(cls2) == list(classes) and len(csr) == n
    # what the reference does (pmgt/pmgt/trainer.py:37-41)
    ref = nx.relabel_nodes(g, {label: i + 2 for i, label in enumerate(classes)})
    reordered = 0
    for v in range(2, n + 2):
        assert list(csr.neighbors(v)) == list(ref[v].keys())
        np.testing.assert_array_equal(csr.weights[csr.indptr[v]: csr.indptr[v + 1]], [d["weight"] for d in ref[v].values()])
        orig = [int(np.searchsorted(classes, u)) + 2 for u in g[classes[v - 2]].keys()]
        reordered += orig != list(csr.neighbors(v))
    assert reordered > 0        # relabel_nodes really permutes neighbour lists: restating it naively would break sampling parity
    v2, t2 = pio.load_features(str(tmp_path), n)
    np.testing.assert_array_equal(v2, vis)
    np.testing.assert_array_equal(t2, txt)
    # the C++ sampler on the loaded CSR draws what the sampler oracle draws on the relabelled networkx graph
    og = so.OrderedGraph(n, [], [])
    for v in range(2, n + 2):
        og.adj[v] = list(ref[v].keys())
        og.w[v] = [d["weight"] for d in ref[v].values()]
    og._nbr_sets = [set(a) for a in og.adj]
    smp = MCNSampler(csr, max_ctx_neigh=7)
    smp.seed(3)
    np.random.seed(3)
    for tgt in (2, 9, n + 1):
        ids, mask = smp.context(tgt)                     # get_input_tensor layout: [target] + context, mask of 1s
        exp, enum = so.sample_context_neigh(og, tgt, [16, 8, 4], 7)
        assert [int(x) for x in ids] == [tgt] + [int(x) for x in exp] and int(mask.sum()) == enum + 1


def test_isolated_nodes_are_refused(tmp_path):
    g, classes = labelled_graph()
    g.add_node("zzz_isolated")
    classes = np.sort(np.append(classes, "zzz_isolated"))
    pio.save_dataset_dir(str(tmp_path), g, classes, np.zeros((len(classes) + 2, 4), np.float32), np.zeros((len(classes) + 2, 4), np.float32))
    with pytest.raises(ValueError, match="isolated"):
        pio.load_graph(str(tmp_path))


def test_load_node_init_emb_remap_and_normalise(tmp_path):
    import joblib
    from sklearn.preprocessing import LabelEncoder
    from sklearn.preprocessing import normalize as sk_normalize
    node_enc, item_enc = LabelEncoder(), LabelEncoder()
    node_enc.classes_ = np.array(["a", "c", "d", "f"])
    item_enc.classes_ = np.array(["a", "b", "c", "d", "e"])
    emb = np.random.RandomState(0).standard_normal((4, 6)).astype(np.float32)
    emb[2] = 0.0                                        # a zero row stays zero under sklearn's normalize
    joblib.dump(node_enc, tmp_path / "node_encoder")
    joblib.dump(item_enc, tmp_path / "item_encoder")
    pio.save_embeddings(str(tmp_path / "emb.npy"), emb)
    np.random.seed(11)
    got = pio.load_node_init_emb(str(tmp_path / "item_encoder"), str(tmp_path / "node_encoder"), str(tmp_path / "emb.npy"))
    np.random.seed(11)
    exp = np.empty((5, 6), np.float32)
    exp[0], exp[2], exp[3] = emb[0], emb[1], emb[2]
    exp[1] = np.random.normal(size=6)
    exp[4] = np.random.normal(size=6)
    np.testing.assert_allclose(got, sk_normalize(exp), rtol=1e-6, atol=1e-7)
    raw = pio.load_node_init_emb(str(tmp_path / "item_encoder"), str(tmp_path / "node_encoder"), str(tmp_path / "emb.npy"), normalize=False)
    np.testing.assert_array_equal(raw[[0, 2, 3]], emb[[0, 1, 2]])
