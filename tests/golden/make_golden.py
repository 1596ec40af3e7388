"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Dev-container tool: needs /root/reference (read-only) and is never run on the GPU box.  It imports
the reference through `_ref_shim`, feeds it deterministic synthetic inputs (graphs, feature
tables and parameters all come from seeded numpy legacy streams via `oracle/` helpers, so the
fixtures store only the reference's OUTPUTS plus the small inputs that are not re-derivable), and
writes `*.npz` files.  Run:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_shim  # noqa: E402

_ref_shim.install()

import networkx as nx  # noqa: E402
from pmgt.optimizers import DenseSparseAdamW  # noqa: E402  (reference)
from pmgt.pmgt.configuration_pmgt import PMGTConfig  # noqa: E402
from pmgt.pmgt.datasets import PMGTDataset, _sample_context_neigh, pmgt_collate_fn  # noqa: E402
from pmgt.pmgt.models import PMGT  # noqa: E402

from oracle import pmgt_oracle as po  # noqa: E402
from oracle import sampler_oracle as so  # noqa: E402

BIG = 3000           # tensors above this many elements are stored strided (+ their L2 norm)
STRIDE = 37


def build_nx(n_nodes, edges, weights):
    g = nx.Graph()
    g.add_nodes_from(range(2, n_nodes + 2))
    for (u, v), w in zip(edges.tolist(), weights.tolist()):
        g.add_edge(u, v, weight=w)
    return g


GRAPHS = {
    "A": dict(n=60, e=200, seed=1),
    "B": dict(n=40, e=44, seed=2),      # sparse: contexts get padded
    "C": dict(n=300, e=1500, seed=3),
    "VG": dict(n=7252, e=88606, seed=4),   # BASELINE.json configs[1]: VG item graph size (synthetic edges: the data does not ship)
}


def graph(name):
    spec = GRAPHS[name]
    edges, w = so.synth_graph(spec["n"], spec["e"], spec["seed"])
    return spec["n"], edges, w, build_nx(spec["n"], edges, w)


def t2n(x):
    return x.detach().cpu().numpy()


def collated_to_np(prefix, coll, out):
    tgt, pair, num_pairs, labels = coll
    out[prefix + "tgt_ids"] = t2n(tgt["node_ids"])
    out[prefix + "tgt_mask"] = t2n(tgt["attention_mask"])
    out[prefix + "pair_ids"] = t2n(pair["node_ids"])
    out[prefix + "pair_mask"] = t2n(pair["attention_mask"])
    out[prefix + "num_pairs"] = t2n(num_pairs)
    out[prefix + "labels"] = t2n(labels)


# ------------------------------------------------------------------------------------------
# G1: sampler
# ------------------------------------------------------------------------------------------
def make_sampler():
    for gname in ("A", "B", "C"):
        n, edges, w, g = graph(gname)
        out = {"edges": edges, "weights": w, "n_nodes": np.int64(n)}
        for S in (6, 16, 32):
            for seed in (0, 1, 2):
                key = f"S{S}_seed{seed}_"
                # raw contexts
                np.random.seed(seed)
                ctxs, nums = [], []
                for t in range(2, 2 + 8):
                    c, k = _sample_context_neigh(g, t, [16, 8, 4], S - 1)
                    ctxs.append(c)
                    nums.append(k)
                out[key + "ctx"] = np.array(ctxs, dtype=np.int64)
                out[key + "num_ctx"] = np.array(nums, dtype=np.int64)
                idx = [0, 5, 7, n - 1, 3, 11]
                node_ids = np.arange(2, n + 2)
                # training items
                np.random.seed(seed)
                ds = PMGTDataset(g, node_ids, max_ctx_neigh=S - 1)
                collated_to_np(key + "train_", pmgt_collate_fn([ds[i] for i in idx]), out)
                # eval items
                np.random.seed(seed)
                ds = PMGTDataset(g, node_ids, max_ctx_neigh=S - 1, is_training=False)
                collated_to_np(key + "eval_", pmgt_collate_fn([ds[i] for i in idx]), out)
                # inference items
                np.random.seed(seed)
                ds = PMGTDataset(g, max_ctx_neigh=S - 1, is_training=False, is_inference=True)
                inf = pmgt_collate_fn([ds[i] for i in idx])
                out[key + "inf_ids"] = t2n(inf["node_ids"])
                out[key + "inf_mask"] = t2n(inf["attention_mask"])
        out["idx"] = np.array([0, 5, 7, n - 1, 3, 11], dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, f"sampler_{gname}.npz"), **out)
        print("sampler", gname, "ok")

    # G7: train_test_split partitions (pmgt/pmgt/trainer.py:45-52)
    from sklearn.model_selection import train_test_split
    out = {}
    for n, vs, seed in ((301, 0.2, 0), (7252, 0.2, 0), (10834, 0.1, 3)):
        tr, va = train_test_split(np.arange(2, n + 2), test_size=vs, random_state=seed)
        k = f"n{n}_v{vs}_s{seed}_"
        out[k + "train_head"] = tr[:64]
        out[k + "valid_head"] = va[:64]
        out[k + "train_sha"] = np.frombuffer(hashlib.sha256(tr.astype(np.int64).tobytes()).digest(), dtype=np.uint8)
        out[k + "valid_sha"] = np.frombuffer(hashlib.sha256(va.astype(np.int64).tobytes()).digest(), dtype=np.uint8)
        out[k + "sizes"] = np.array([len(tr), len(va)])
    np.savez_compressed(os.path.join(HERE, "split.npz"), **out)


# ------------------------------------------------------------------------------------------
# G2-G5: model
# ------------------------------------------------------------------------------------------
MODEL_CASES = {
    # name: (graph, cfg kwargs, S, B, sampler seed, param seed)
    "m1": ("A", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.5), 16, 4, 0, 11),
    "m1_beta1": ("A", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=1.0), 16, 4, 0, 11),
    "m1_beta0": ("A", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.0), 16, 4, 0, 11),
    "m1_pad": ("B", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.5), 16, 4, 1, 12),
    "m2": ("A", dict(hidden_size=128, num_attention_heads=4, num_hidden_layers=2, intermediate_size=128, beta=0.5), 16, 3, 2, 13),
    "m3": ("C", dict(hidden_size=256, num_attention_heads=8, num_hidden_layers=4, intermediate_size=256, beta=0.5), 32, 2, 3, 14),
    "m4": ("C", dict(hidden_size=128, num_attention_heads=2, num_hidden_layers=1, intermediate_size=512, beta=0.3), 6, 5, 4, 15),
    # the reference's own two entry configurations: CLI defaults (train.py:225-272: hidden 128, ONE head of size 128, 5 layers,
    # max_ctx_neigh 5) and the author's script (scripts/run_pmgt.sh:18-25: hidden 32, 3 layers, beta 1.0, the other defaults)
    "e_cli": ("C", dict(hidden_size=128, num_attention_heads=1, num_hidden_layers=5, intermediate_size=128, beta=0.5), 6, 6, 5, 16),
    "e_script": ("C", dict(hidden_size=32, num_attention_heads=1, num_hidden_layers=3, intermediate_size=128, beta=1.0), 6, 6, 6, 17),
    # modality counts other than the trainer's two (modeling_pmgt.py:163-173,195-201,549-569 are generic over len(feat_hidden_sizes))
    "f3": ("C", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.5,
                     feat_hidden_sizes=[1536, 768, 256]), 16, 4, 7, 18),
    "f1": ("A", dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.5,
                     feat_hidden_sizes=[768]), 16, 4, 8, 19),
    "f4": ("A", dict(hidden_size=128, num_attention_heads=4, num_hidden_layers=1, intermediate_size=128, beta=0.5,
                     feat_hidden_sizes=[64, 128, 32, 256]), 6, 5, 9, 20),
}


def ref_model(n, cfgkw, pseed, tseed=77):
    cfg = po.default_cfg(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **cfgkw)
    rcfg = PMGTConfig(**{k: v for k, v in cfg.items()})
    tables = po.synth_tables(n, cfg["feat_hidden_sizes"], tseed)
    model = PMGT(node_size=n, config=rcfg, feat_init_emb=[t.numpy() for t in tables])
    params = po.synth_params(cfg, pseed)
    sd = model.state_dict()
    for k, v in params.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k].copy_(v)
    names = [nme for nme, p in model.named_parameters() if p.requires_grad]
    assert names == [nme for nme, _ in po.param_shapes(cfg)], "param order differs from reference"
    return cfg, model, tables


def store(out, key, arr):
    a = np.ascontiguousarray(arr)
    if a.size > BIG:
        out[key + "@strided"] = a.ravel()[::STRIDE].copy()
        out[key + "@norm"] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
    else:
        out[key] = a


def draw_nfr(seed, ids, n):
    """Replay the three draws PMGT.forward makes (pmgt/pmgt/models.py:135-147) for torch seed."""
    torch.manual_seed(seed)
    B, S = ids.shape
    r1 = torch.rand(B, S - 1)
    m1 = (r1 < 0.02) * (ids[:, 1:] != 0)
    repl = torch.randint(2, n + 2, (int(m1.sum()),))
    r2 = torch.rand(B, S - 1)
    return r1, repl, r2


def make_model(only=None):
    for name, (gname, cfgkw, S, B, sseed, pseed) in MODEL_CASES.items():
        if only and name not in only:
            continue
        n, edges, w, g = graph(gname)
        cfg, model, tables = ref_model(n, cfgkw, pseed)
        out = {"S": np.int64(S), "B": np.int64(B), "graph": np.array(gname), "pseed": np.int64(pseed),
               "tseed": np.int64(77)}
        np.random.seed(sseed)
        ds = PMGTDataset(g, np.arange(2, n + 2), max_ctx_neigh=S - 1)
        idx = np.random.RandomState(sseed + 100).choice(n, B, replace=False)
        out["idx"] = idx
        batch = pmgt_collate_fn([ds[int(i)] for i in idx])
        collated_to_np("b_", batch, out)

        # ---- eval forward (G3) + module-level (G2)
        model.eval()
        with torch.no_grad():
            o = model(*batch, output_attentions=True, output_hidden_states=True)
        out["eval_loss"] = t2n(o.loss)
        out["eval_logits"] = t2n(o.prediction_logits)
        full = name == "m1"           # one case keeps every module output in full
        keep = (lambda k, a: out.__setitem__(k, a)) if full else (lambda k, a: store(out, k, a))
        keep("eval_last_hidden", t2n(o.last_hidden_state))
        for i, hs in enumerate(o.hidden_states):
            keep(f"eval_hidden_{i}", t2n(hs))
        for i, at in enumerate(o.attentions):
            keep(f"eval_attn_{i}", t2n(at))
        with torch.no_grad():
            inf = model(batch[0])                      # inference path: loss None → [0] is hidden
        out["inf_cls"] = t2n(inf[0][:, 0])

        # ---- train forward/backward with captured NFR draws (G4)
        model.train()
        model.zero_grad()
        nseed = 1000 + pseed
        torch.manual_seed(nseed)
        o = model(*batch)
        o.loss.backward()
        r1, repl, r2 = draw_nfr(nseed, batch[0]["node_ids"], n)
        out["nfr_r1"], out["nfr_repl"], out["nfr_r2"] = t2n(r1), t2n(repl), t2n(r2)
        out["train_loss"] = t2n(o.loss)
        out["train_logits"] = t2n(o.prediction_logits)
        # gsr part = eval loss on same batch (dropout 0) → nfr = loss - gsr
        for nme, p in model.named_parameters():
            if p.requires_grad:
                assert p.grad is not None, nme
                store(out, "grad/" + nme, t2n(p.grad))

        # ---- optimizer steps (G5): clip 5.0 + DenseSparseAdamW(lr 1e-3, wd 1e-2), same batch,
        #      NFR draws re-seeded per step.
        no_decay = ["bias", "LayerNorm.weight"]
        groups = [
            {"params": [p for nme, p in model.named_parameters() if p.requires_grad and not any(nd in nme for nd in no_decay)],
             "weight_decay": 1e-2, "lr": 1e-3},
            {"params": [p for nme, p in model.named_parameters() if p.requires_grad and any(nd in nme for nd in no_decay)],
             "weight_decay": 0.0, "lr": 1e-3},
        ]
        opt = DenseSparseAdamW(groups)
        nsteps = 6 if name != "m3" else 3
        losses, norms = [], []
        draws = []
        for step in range(nsteps):
            model.zero_grad()
            torch.manual_seed(2000 + step)
            o = model(*batch)
            o.loss.backward()
            r1, repl, r2 = draw_nfr(2000 + step, batch[0]["node_ids"], n)
            draws.append((t2n(r1), t2n(repl), t2n(r2)))
            tn = torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], 5.0)
            opt.step()
            losses.append(float(o.loss))
            norms.append(float(tn))
        out["opt_losses"] = np.array(losses, dtype=np.float64)
        out["opt_gradnorms"] = np.array(norms, dtype=np.float64)
        for s, (a, b, c) in enumerate(draws):
            out[f"opt_r1_{s}"], out[f"opt_repl_{s}"], out[f"opt_r2_{s}"] = a, b, c
        for nme, p in model.named_parameters():
            if p.requires_grad:
                store(out, "final/" + nme, t2n(p))
        np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **out)
        print("model", name, "eval", float(out["eval_loss"]), "train", float(out["train_loss"]), losses)

    if only:
        return
    # ---- G6: init statistics of a fresh reference model (pmgt/pmgt/modeling_pmgt.py:44-58)
    torch.manual_seed(0)
    cfg = po.default_cfg(hidden_size=128, num_attention_heads=4, num_hidden_layers=2, intermediate_size=128)
    tables = po.synth_tables(60, cfg["feat_hidden_sizes"], 77)
    model = PMGT(node_size=60, config=PMGTConfig(**cfg), feat_init_emb=[t.numpy() for t in tables])
    out = {}
    for nme, p in model.named_parameters():
        if p.requires_grad:
            out["mean/" + nme] = np.float64(p.mean().item())
            out["std/" + nme] = np.float64(p.std().item()) if p.numel() > 1 else np.float64(0)
            out["absmax/" + nme] = np.float64(p.abs().max().item())
    np.savez_compressed(os.path.join(HERE, "init_stats.npz"), **out)


# ------------------------------------------------------------------------------------------
# G8: PMGT_NCF — the encoder boundary driven by a second caller (pmgt/pmgt_ncf/models.py:15-105)
# ------------------------------------------------------------------------------------------
def make_ncf():
    from pmgt.pmgt.datasets import get_input_tensor
    from pmgt.pmgt_ncf.models import PMGT_NCF
    from tests import golden_util as gu

    for name, (gname, cfgkw, S, B, users, factor, nl, mdl, sseed, pseed, hseed) in gu.NCF_CASES.items():
        n, edges, w, g = graph(gname)
        cfg = po.default_cfg(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **cfgkw)
        tables = po.synth_tables(n, cfg["feat_hidden_sizes"], 77)
        model = PMGT_NCF(user_num=users, item_num=n, factor_num=factor, num_layers=nl, model=mdl, config=PMGTConfig(**cfg))
        sd = model.state_dict()
        for k, v in po.synth_params(cfg, pseed).items():
            if k.startswith("bert."):
                sd[k].copy_(v)
        head = gu.ncf_head_params(users, n, factor, nl, mdl, hseed)
        assert [k for k, p in model.named_parameters() if p.requires_grad and not k.startswith("bert.")] == list(head), "head order"
        for k, v in head.items():
            sd[k].copy_(v)
        for i, t in enumerate(tables):
            sd[f"feat_embeddings.{i}.weight"].copy_(t)
        np.random.seed(sseed)
        rs = np.random.RandomState(sseed + 50)
        items = rs.choice(n, B, replace=False)
        pairs = [get_input_tensor(g, int(i) + 2, [16, 8, 4], S - 1) for i in items]
        item = {"node_ids": torch.stack([p[0] for p in pairs]), "attention_mask": torch.stack([p[1] for p in pairs])}
        user = torch.from_numpy(rs.randint(0, users, B).astype(np.int64))
        labels = torch.from_numpy((rs.random_sample(B) < 0.5).astype(np.float32))
        out = {"item_ids": t2n(item["node_ids"]), "item_mask": t2n(item["attention_mask"]), "user": t2n(user), "labels": t2n(labels)}
        model.train()                      # dropout probabilities are 0: train == eval numerically
        model.zero_grad()
        logits = model(user, item)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, labels)
        loss.backward()
        out["logits"], out["loss"] = t2n(logits), t2n(loss)
        for nme, p in model.named_parameters():
            if p.requires_grad:
                assert p.grad is not None, nme
                store(out, "grad/" + nme, t2n(p.grad))
        assert all(p.grad is None for p in model.feat_embeddings.parameters())       # frozen tables (models.py:45-47)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print("ncf", name, "loss", float(loss))


# ------------------------------------------------------------------------------------------
# G9: a 30-step pre-training loss curve over FRESH batches (reference sampler + model + clip + DenseSparseAdamW)
# ------------------------------------------------------------------------------------------
CURVES = {
    # name: (graph, S, B, steps, sampler seed, param seed, lr, cfg kwargs)
    "curve_c": ("C", 16, 6, 30, 9, 41, 1e-3, dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=64, beta=0.5)),
    # the benchmark configuration (BASELINE.json configs[1] shapes: 7 252-node graph, L4 H8 d256 S32) at the author's batch size
    # and learning rate (scripts/run_pmgt.sh:11-13): 20 optimizer steps of the reference on fresh batches
    "curve_c2": ("VG", 32, 32, 20, 10, 42, 1e-4, dict(hidden_size=256, num_attention_heads=8, num_hidden_layers=4, intermediate_size=256, beta=0.5)),
}


def make_curve(name="curve_c"):
    gname, S, B, steps, sseed, pseed, lr, cfgkw = CURVES[name]
    n, edges, w, g = graph(gname)
    cfg, model, tables = ref_model(n, cfgkw, pseed)
    np.random.seed(sseed)                                   # the sampler's process-global stream (pmgt/utils/base.py:37)
    ds = PMGTDataset(g, np.arange(2, n + 2), max_ctx_neigh=S - 1)
    order = np.random.RandomState(sseed + 1).permutation(n)
    no_decay = ["bias", "LayerNorm.weight"]
    groups = [
        {"params": [p for nme, p in model.named_parameters() if p.requires_grad and not any(nd in nme for nd in no_decay)],
         "weight_decay": 1e-2, "lr": lr},
        {"params": [p for nme, p in model.named_parameters() if p.requires_grad and any(nd in nme for nd in no_decay)],
         "weight_decay": 0.0, "lr": lr},
    ]
    opt = DenseSparseAdamW(groups)
    model.train()
    out = {"S": np.int64(S), "B": np.int64(B), "steps": np.int64(steps), "sseed": np.int64(sseed), "pseed": np.int64(pseed),
           "order": order, "lr": np.float64(lr)}
    losses, norms = [], []
    for step in range(steps):
        idx = order[(step * B) % (n - B): (step * B) % (n - B) + B]
        batch = pmgt_collate_fn([ds[int(i)] for i in idx])
        model.zero_grad()
        torch.manual_seed(3000 + step)
        o = model(*batch)
        o.loss.backward()
        r1, repl, r2 = draw_nfr(3000 + step, batch[0]["node_ids"], n)
        out[f"r1_{step}"], out[f"repl_{step}"], out[f"r2_{step}"] = t2n(r1), t2n(repl), t2n(r2)
        tn = torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], 5.0)
        opt.step()
        losses.append(float(o.loss.detach()))
        norms.append(float(tn))
        if step in (0, steps - 1):
            out[f"tgt_ids_{step}"] = t2n(batch[0]["node_ids"])
        print("curve", name, step, losses[-1], norms[-1], flush=True)
    out["losses"] = np.array(losses, dtype=np.float64)
    out["gradnorms"] = np.array(norms, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("curve", losses[:3], "...", losses[-3:])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "curve":          # curve [name ...]
        for nm in (sys.argv[2:] or ["curve_c"]):
            make_curve(nm)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "model":          # model name [name ...]: only these cases
        make_model(set(sys.argv[2:]))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ncf":
        make_ncf()
        sys.exit(0)
    make_sampler()
    make_model()
    make_ncf()
    for nm in CURVES:
        make_curve(nm)
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print("total fixture bytes", tot)
