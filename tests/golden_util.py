"""Helpers shared by the parity tests: rebuild the deterministic inputs that the golden fixtures
were generated from (tests/golden/make_golden.py) and unpack stored reference outputs."""
import os

import numpy as np
import torch

from oracle import pmgt_oracle as po
from oracle import sampler_oracle as so

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STRIDE = 37

GRAPHS = {"A": dict(n=60, e=200, seed=1), "B": dict(n=40, e=44, seed=2), "C": dict(n=300, e=1500, seed=3),
          "VG": dict(n=7252, e=88606, seed=4)}      # BASELINE.json configs[1]: the VG item graph's size (synthetic edges)

MODEL_CASES = {
    "m1": ("A", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.5)),
    "m1_beta1": ("A", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=1.0)),
    "m1_beta0": ("A", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.0)),
    "m1_pad": ("B", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.5)),
    "m2": ("A", dict(hidden_size=128, num_attention_heads=4, num_hidden_layers=2, intermediate_size=128, beta=0.5)),
    "m3": ("C", dict(hidden_size=256, num_attention_heads=8, num_hidden_layers=4, intermediate_size=256, beta=0.5)),
    "m4": ("C", dict(hidden_size=128, num_attention_heads=2, num_hidden_layers=1, intermediate_size=512, beta=0.3)),
    # the reference's two entry configurations: CLI defaults (train.py:225-272) and the author's script (scripts/run_pmgt.sh:18-25)
    "e_cli": ("C", dict(hidden_size=128, num_attention_heads=1, num_hidden_layers=5, intermediate_size=128, beta=0.5)),
    "e_script": ("C", dict(hidden_size=32, num_attention_heads=1, num_hidden_layers=3, intermediate_size=128, beta=1.0)),
    # three, one and four modalities (the reference's modules are generic over len(feat_hidden_sizes); its trainer builds two)
    "f3": ("C", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.5,
                     feat_hidden_sizes=[1536, 768, 256])),
    "f1": ("A", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.5,
                     feat_hidden_sizes=[768])),
    "f4": ("A", dict(hidden_size=128, num_attention_heads=4, num_hidden_layers=1, intermediate_size=128, beta=0.5,
                     feat_hidden_sizes=[64, 128, 32, 256])),
}


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def graph(gname):
    spec = GRAPHS[gname]
    edges, w = so.synth_graph(spec["n"], spec["e"], spec["seed"])
    return spec["n"], edges, w


def model_case(name, dtype=torch.float32):
    """→ dict(cfg, params, tables, batch, n_nodes, gold)"""
    gname, cfgkw = MODEL_CASES[name]
    gold = load("model_" + name)
    n = GRAPHS[gname]["n"]
    cfg = po.default_cfg(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **cfgkw)
    params = po.synth_params(cfg, int(gold["pseed"]), dtype)
    tables = po.synth_tables(n, cfg["feat_hidden_sizes"], int(gold["tseed"]), dtype)
    batch = batch_from(gold, "b_")
    return dict(cfg=cfg, params=params, tables=tables, batch=batch, n_nodes=n, gold=gold, name=name)


def batch_from(gold, prefix):
    tgt = {"node_ids": torch.from_numpy(gold[prefix + "tgt_ids"]),
           "attention_mask": torch.from_numpy(gold[prefix + "tgt_mask"])}
    pair = {"node_ids": torch.from_numpy(gold[prefix + "pair_ids"]),
            "attention_mask": torch.from_numpy(gold[prefix + "pair_mask"])}
    return tgt, pair, torch.from_numpy(gold[prefix + "num_pairs"]), torch.from_numpy(gold[prefix + "labels"])


def nfr_inject(gold, ids, n_nodes, prefix="nfr_", suffix=""):
    r1 = torch.from_numpy(gold[prefix + "r1" + suffix])
    repl = torch.from_numpy(gold[prefix + "repl" + suffix])
    r2 = torch.from_numpy(gold[prefix + "r2" + suffix])
    return po.nfr_masking(ids, n_nodes, r1, repl, r2)


def check_stored(gold, key, arr, rtol, atol):
    """Compare `arr` with a fixture entry stored either in full or strided (+L2 norm)."""
    a = np.asarray(arr, dtype=np.float64)
    if key in gold.files:
        np.testing.assert_allclose(a, gold[key], rtol=rtol, atol=atol, err_msg=key)
    else:
        np.testing.assert_allclose(a.ravel()[::STRIDE], gold[key + "@strided"], rtol=rtol, atol=atol, err_msg=key)
        np.testing.assert_allclose(np.sqrt((a ** 2).sum()), float(gold[key + "@norm"]), rtol=max(rtol, 1e-5), err_msg=key)


# ---- second caller of the encoder boundary: PMGT_NCF (pmgt/pmgt_ncf/models.py:15-105) ------------------------
NCF_CASES = {
    # name: (graph, cfg kwargs, S, B, users, factor_num, num_layers, model, sampler seed, param seed, head seed)
    "ncf_mlp": ("A", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.5),
                16, 6, 9, 16, 3, "MLP", 5, 21, 31),
    "ncf_neumf": ("C", dict(hidden_size=128, num_attention_heads=4, num_hidden_layers=2, intermediate_size=128, beta=0.5),
                  16, 8, 13, 32, 3, "NeuMF-end", 6, 22, 32),
}


def ncf_head_shapes(user_num, item_num, factor_num, num_layers, model):
    """Names/shapes of the NCF head parameters in the reference's named_parameters() order."""
    shapes = [("mlp_user_embeddings.weight", (user_num, factor_num * 2 ** (num_layers - 1)))]
    for i in range(num_layers):
        n_in = factor_num * 2 ** (num_layers - i)
        shapes += [(f"mlp_layers.{i}.linear.weight", (n_in // 2, n_in)), (f"mlp_layers.{i}.linear.bias", (n_in // 2,))]
    if model == "NeuMF-end":
        shapes += [("gmf_user_embeddings.weight", (user_num, factor_num)), ("gmf_item_embeddings.weight", (item_num, factor_num)),
                   ("predict_layer.weight", (1, 2 * factor_num)), ("predict_layer.bias", (1,))]
    else:
        shapes += [("predict_layer.weight", (1, factor_num)), ("predict_layer.bias", (1,))]
    return shapes


def ncf_head_params(user_num, item_num, factor_num, num_layers, model, seed):
    rs = np.random.RandomState(seed)
    return {k: torch.from_numpy((rs.standard_normal(shp) * (0.02 if k.endswith("bias") else 0.1)).astype(np.float32))
            for k, shp in ncf_head_shapes(user_num, item_num, factor_num, num_layers, model)}


def ncf_case(name, dtype=torch.float32):
    gname, cfgkw, S, B, users, factor, nl, model, sseed, pseed, hseed = NCF_CASES[name]
    gold = load(name)
    n = GRAPHS[gname]["n"]
    cfg = po.default_cfg(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **cfgkw)
    params = {k: v for k, v in po.synth_params(cfg, pseed, dtype).items() if k.startswith("bert.")}
    tables = po.synth_tables(n, cfg["feat_hidden_sizes"], 77, dtype)
    head = ncf_head_params(users, n, factor, nl, model, hseed)
    item = {"node_ids": torch.from_numpy(gold["item_ids"]), "attention_mask": torch.from_numpy(gold["item_mask"])}
    return dict(cfg=cfg, params=params, tables=tables, head=head, n_nodes=n, gold=gold, user=torch.from_numpy(gold["user"]),
                item=item, labels=torch.from_numpy(gold["labels"]), users=users, factor=factor, num_layers=nl, model=model)


# ---- G9: 30-step loss curve over fresh batches ---------------------------------------------------------------
CURVES = {
    "curve_c": ("C", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.5)),
    # the benchmark configuration: VG-sized graph, L4 H8 d256 S32, B = 32, lr 1e-4 (BASELINE.json configs[1], scripts/run_pmgt.sh:11-13)
    "curve_c2": ("VG", dict(hidden_size=256, num_attention_heads=8, num_hidden_layers=4, intermediate_size=256, beta=0.5)),
}


def curve_case(name="curve_c", dtype=torch.float32):
    gold = load(name)
    gname, cfgkw = CURVES[name]
    n = GRAPHS[gname]["n"]
    cfg = po.default_cfg(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **cfgkw)
    params = po.synth_params(cfg, int(gold["pseed"]), dtype)
    tables = po.synth_tables(n, cfg["feat_hidden_sizes"], 77, dtype)
    lr = float(gold["lr"]) if "lr" in gold.files else 1e-3
    return dict(cfg=cfg, params=params, tables=tables, n_nodes=n, gold=gold, gname=gname, lr=lr)


def curve_batches(case):
    """The batches the reference drew (sequential numpy-legacy stream), regenerated by the C++ sampler."""
    from pmgt_amd.datasets import MODE_TRAIN, MCNSampler
    from pmgt_amd.graph import CSRGraph
    gold = case["gold"]
    n, edges, w = graph(case["gname"])
    S, B, steps = int(gold["S"]), int(gold["B"]), int(gold["steps"])
    smp = MCNSampler(CSRGraph.from_edge_list(n, edges, w), max_ctx_neigh=S - 1)
    smp.seed(int(gold["sseed"]))
    order = gold["order"]
    for step in range(steps):
        lo = (step * B) % (n - B)
        tgt, pair, num_pairs, labels = smp.batch(order[lo: lo + B] + 2, MODE_TRAIN, threads=0)
        yield step, ({k: v.clone() for k, v in tgt.items()}, {k: v.clone() for k, v in pair.items()}, num_pairs.clone(), labels.clone())
