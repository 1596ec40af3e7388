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


def test_lightning_shaped_checkpoint_loads_without_lightning(tmp_path):
    """A checkpoint as the reference's run writes it (pmgt/base_trainer.py:291-298 ModelCheckpoint + save_hyperparameters,
    :146 `self.net`): state_dict under `net.`, hyper_parameters as a pytorch_lightning AttributeDict, callbacks, optimizer
    states.  pytorch_lightning is NOT installed here, and torch's weights_only loader refuses the AttributeDict global: the
    file must still load, as plain data, without executing anything from it."""
    import pickle
    import sys
    import types

    import torch

    from pmgt_amd import io as pio

    # build the checkpoint with a stand-in for pytorch_lightning.utilities.parsing.AttributeDict, then make it unimportable
    names = ["pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing"]
    mods = {n: types.ModuleType(n) for n in names}

    class AttributeDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    class Boom:                                      # an arbitrary object whose unpickling would run code
        def __reduce__(self):
            return (exec, ("raise SystemExit('code from the checkpoint ran')",))

    AttributeDict.__module__ = names[2]
    AttributeDict.__qualname__ = "AttributeDict"
    mods[names[2]].AttributeDict = AttributeDict
    sys.modules.update(mods)
    try:
        sd = {"net.bert.embeddings.position_embeddings.weight": torch.arange(12.0).view(3, 4),
              "net.bert.embeddings.position_ids": torch.arange(3)[None],
              "net.nfr_loss.projections.0.bias": torch.ones(5)}
        ck = {"epoch": 3, "global_step": 120, "pytorch-lightning_version": "1.5.4", "state_dict": sd,
              "hyper_parameters": AttributeDict(lr=1e-4, hidden_size=32, dataset_name="TG", nested=AttributeDict(a=1)),
              "callbacks": {"ModelCheckpoint": {"best_model_score": torch.tensor(0.81), "best_model_path": "/x/y.ckpt",
                                                "monitor": "val/auc"}, "evil": Boom()},
              "optimizer_states": [{"state": {0: {"step": 120, "exp_avg": torch.zeros(3, 4)}},
                                    "param_groups": [{"lr": 1e-4, "betas": (0.9, 0.999), "params": [0]}]}],
              "lr_schedulers": []}
        path = tmp_path / "last.ckpt"
        torch.save(ck, path)
    finally:
        for n in names:
            sys.modules.pop(n, None)
    with pytest.raises(pickle.UnpicklingError):
        torch.load(path, map_location="cpu", weights_only=True)             # why the plain safe loader is not enough
    got = pio.read_checkpoint(str(path))
    assert got["epoch"] == 3 and got["pytorch-lightning_version"] == "1.5.4"
    assert torch.equal(got["state_dict"]["net.bert.embeddings.position_embeddings.weight"], sd["net.bert.embeddings.position_embeddings.weight"])
    assert got["state_dict"]["net.bert.embeddings.position_ids"].dtype == torch.int64
    assert dict(got["hyper_parameters"])["hidden_size"] == 32 and dict(got["hyper_parameters"]["nested"]) == {"a": 1}
    assert float(got["callbacks"]["ModelCheckpoint"]["best_model_score"]) == pytest.approx(0.81)
    assert got["optimizer_states"][0]["param_groups"][0]["betas"] == (0.9, 0.999)
    assert type(got["callbacks"]["evil"]).__name__ == "_Opaque"              # inert placeholder: exec() was never resolved

    class Sink:                                        # load_checkpoint strips the `net.` prefix and hands over the weights
        def load_state_dict(self, state, strict=True):
            self.state = state
            return "ok"
    sink = Sink()
    assert pio.load_checkpoint(sink, str(path)) == "ok"
    assert set(sink.state) == {k[4:] for k in sd}
