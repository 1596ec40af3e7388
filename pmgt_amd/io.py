"""On-disk formats either side of the hot path: the dataset directory the reference's
trainer reads (`node_encoder`, `graph.gpickle`, `visual_init_emb.npy`, `textual_init_emb.npy`;
pmgt/pmgt/trainer.py:30-71,108-135), Lightning-style checkpoints with the reference's key names
(`net.` prefix, pmgt/base_trainer.py:99-110,291-298), the exported `[N, d]` embedding file and its
node -> item remap (pmgt/pmgt/utils.py:15-40).  Host-side glue only: nothing here is on the device path."""
from __future__ import annotations

import os
import pickle
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .graph import CSRGraph


def _load_encoder(path: str):
    import joblib
    return joblib.load(path)


def relabel_like_reference(graph, classes) -> "object":
    """mapping = {label: i + 2}; nx.relabel_nodes(graph, mapping) (pmgt/pmgt/trainer.py:37-41).  The copy that
    networkx builds re-inserts edges in `graph.edges()` order, which CHANGES the neighbour order of every node —
    and the neighbour order is the `a` array of the sampler's np.random.choice — so the relabel is done by
    networkx itself rather than restated."""
    import networkx as nx
    mapping = {label: i + 2 for i, label in enumerate(classes)}
    return nx.relabel_nodes(graph, mapping)


def load_graph(data_dir: str) -> Tuple[CSRGraph, np.ndarray]:
    """`node_encoder` (joblib'd sklearn LabelEncoder) + `graph.gpickle` (pickled nx.Graph, float `weight` per
    edge) -> CSR with ids 2..N+1 in the reference's adjacency order, and the encoder's classes."""
    enc = _load_encoder(os.path.join(data_dir, "node_encoder"))
    with open(os.path.join(data_dir, "graph.gpickle"), "rb") as f:        # nx.read_gpickle == pickle.load
        g = pickle.load(f)
    classes = np.asarray(enc.classes_)
    if g.number_of_nodes() != len(classes):
        raise ValueError(f"graph has {g.number_of_nodes()} nodes but node_encoder has {len(classes)} classes")
    csr = CSRGraph.from_networkx(relabel_like_reference(g, classes))
    csr.validate()
    return csr, classes


def load_features(data_dir: str, n_nodes: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """`visual_init_emb.npy`, `textual_init_emb.npy`: [N+2, F_m] with rows 0 (<pad>) and 1 (<mask>)
    (notebooks/PMGT.ipynb cell 30; loaded at pmgt/pmgt/trainer.py:114-116)."""
    vis = np.load(os.path.join(data_dir, "visual_init_emb.npy"))
    txt = np.load(os.path.join(data_dir, "textual_init_emb.npy"))
    if vis.shape[0] != txt.shape[0] or (n_nodes is not None and vis.shape[0] != n_nodes + 2):
        raise ValueError(f"feature tables have {vis.shape[0]} / {txt.shape[0]} rows, expected {None if n_nodes is None else n_nodes + 2}")
    return vis, txt


def save_dataset_dir(data_dir: str, graph, classes, visual: np.ndarray, textual: np.ndarray):
    """Writer for the same layout (used by the tests and to stage synthetic datasets)."""
    import joblib
    from sklearn.preprocessing import LabelEncoder
    os.makedirs(data_dir, exist_ok=True)
    enc = LabelEncoder()
    enc.classes_ = np.asarray(classes)
    joblib.dump(enc, os.path.join(data_dir, "node_encoder"))
    with open(os.path.join(data_dir, "graph.gpickle"), "wb") as f:
        pickle.dump(graph, f, pickle.HIGHEST_PROTOCOL)
    np.save(os.path.join(data_dir, "visual_init_emb.npy"), visual)
    np.save(os.path.join(data_dir, "textual_init_emb.npy"), textual)


# ---- checkpoints ---------------------------------------------------------------------------------------------
def to_reference_state_dict(model, prefix: str = "net.") -> Dict[str, torch.Tensor]:
    """state_dict of a `pmgt_amd.models.PMGT` under the keys a Lightning checkpoint of the reference holds
    (`net.bert.…`, `net.nfr_loss.projections.…`, `net.feat_embeddings.{0,1}.weight`, position/role id buffers)."""
    return {prefix + k: v.detach().to("cpu", copy=True) for k, v in model.state_dict().items()}


def save_checkpoint(model, path: str, prefix: str = "net.", **extra):
    torch.save({"state_dict": to_reference_state_dict(model, prefix), **extra}, path)


class _Opaque(dict):
    """Stand-in for any class a checkpoint pickles that is not plain data (Lightning's AttributeDict / AttrDict
    hyper-parameters, callback state objects, ...): keeps items and attributes, runs none of the original code."""

    def __init__(self, *args, **kwargs):
        dict.__init__(self)

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):
            self.__dict__.update(state[1])

    def __reduce__(self):                       # never re-pickled as the original class
        return (dict, (dict(self),))


class _TolerantUnpickler(pickle.Unpickler):
    """Restricted unpickler for checkpoints written by the reference's Lightning run (pmgt/base_trainer.py:291-298:
    ModelCheckpoint -> {"state_dict", "hyper_parameters", "callbacks", "optimizer_states", ...}).  torch's
    weights_only loader refuses such a file because `hyper_parameters` is an AttributeDict / AttrDict; this one resolves
    ONLY tensor / container reconstruction globals and maps every other global to an inert placeholder, so a real checkpoint
    loads without pytorch_lightning installed and without executing code from the file."""

    _ALLOWED = {
        ("collections", "OrderedDict"), ("collections", "defaultdict"), ("copyreg", "_reconstructor"), ("builtins", "dict"),
        ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "object"),
        ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"),
        ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range"),
        ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
        ("torch._utils", "_rebuild_parameter_with_state"), ("torch", "Size"), ("torch", "device"), ("torch", "dtype"),
        ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
        ("numpy", "dtype"), ("numpy", "ndarray"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        if module == "torch" and (name.endswith("Storage") or isinstance(getattr(torch, name, None), torch.dtype)):
            return getattr(torch, name)
        return _Opaque


class _TolerantPickle:            # the `pickle_module` interface torch.load asks for
    Unpickler = _TolerantUnpickler
    __name__ = "pmgt_amd.io._TolerantPickle"

    @staticmethod
    def load(f, **kw):
        return _TolerantUnpickler(f, **kw).load()


def read_checkpoint(path) -> dict:
    """A checkpoint file as plain data.  First torch's weights_only loader (enough for checkpoints written by
    `save_checkpoint`); a file it refuses because it pickles non-tensor classes -- a real Lightning checkpoint of the
    reference -- goes through the restricted tolerant unpickler above (still no code from the file is executed)."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_TolerantPickle)


def load_checkpoint(model, path_or_dict, prefix: str = "net.", strict: bool = True):
    """Accepts a Lightning checkpoint ({"state_dict": {"net.…": …}, "hyper_parameters": …, "callbacks": …,
    "optimizer_states": …}), a bare state_dict with or without the prefix, or a path to either; copies the weights into the
    engine's flat buffer and re-uploads the tables."""
    ck = read_checkpoint(path_or_dict) if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
    sd = ck.get("state_dict", ck)
    if any(k.startswith(prefix) for k in sd):
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    return model.load_state_dict(sd, strict=strict)


# ---- exported embeddings ------------------------------------------------------------------------------------------
def save_embeddings(path: str, emb: np.ndarray):
    """[N, d] fp32 in node-id order (pmgt/base_trainer.py:403-405: np.save(args.inference_result_path, predictions))."""
    np.save(path, np.ascontiguousarray(emb, dtype=np.float32))


def load_node_init_emb(item_encoder_path: str, node_encoder_path: str, node_init_emb_path: str, normalize: bool = True) -> np.ndarray:
    """pmgt/pmgt/utils.py:15-40: rows of the exported node embeddings re-indexed by the downstream item encoder;
    items absent from the graph get `np.random.normal` rows (global numpy stream, as in the reference); rows are
    L2-normalised (sklearn `normalize`: zero rows stay zero)."""
    item_encoder = _load_encoder(item_encoder_path)
    node_encoder = _load_encoder(node_encoder_path)
    node_init_emb = np.load(node_init_emb_path)
    item2idx = {item: i for i, item in enumerate(node_encoder.classes_)}
    out = np.empty((len(item_encoder.classes_), node_init_emb.shape[1]), dtype=node_init_emb.dtype)
    for i, item in enumerate(item_encoder.classes_):
        if item in item2idx:
            out[i] = node_init_emb[item2idx[item]]
        else:
            out[i] = np.random.normal(size=node_init_emb.shape[1])
    if normalize:
        nrm = np.sqrt((out.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
        nrm[nrm == 0.0] = 1.0
        out = (out / nrm).astype(out.dtype)
    return out
