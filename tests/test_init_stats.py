"""a15: initial parameter distributions against fixture G6 (tests/golden/init_stats.npz = mean / std / absmax of every
parameter of a freshly constructed reference PMGT, d=128 L=2; pmgt/pmgt/modeling_pmgt.py:44-58 for `bert.*`,
torch's nn.Linear default for `nfr_loss.*` because PMGT never calls _init_weights on itself, pmgt/pmgt/models.py:31-54)."""
import math
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
D, L, H, I, N = 128, 2, 4, 128, 60


def golden():
    z = np.load(os.path.join(HERE, "golden", "init_stats.npz"))
    names = sorted({k.split("/", 1)[1] for k in z.files})
    return {n: {s: float(z[f"{s}/{n}"]) for s in ("mean", "std", "absmax")} for n in names}


def shapes():
    from pmgt_amd.configuration_pmgt import PMGTConfig
    cfg = PMGTConfig(hidden_size=D, num_hidden_layers=L, num_attention_heads=H, intermediate_size=I)
    sh = {"bert.embeddings.position_embeddings.weight": (100, D), "bert.embeddings.role_embeddings.weight": (2, D),
          "bert.embeddings.feat_linear.0.weight": (D, 1536), "bert.embeddings.feat_linear.0.bias": (D,),
          "bert.embeddings.feat_linear.1.weight": (D, 768), "bert.embeddings.feat_linear.1.bias": (D,),
          "bert.embeddings.attention.1.weight": (2, 2 * D), "bert.embeddings.attention.1.bias": (2,),
          "bert.embeddings.LayerNorm.weight": (D,), "bert.embeddings.LayerNorm.bias": (D,),
          "nfr_loss.projections.0.weight": (1536, D), "nfr_loss.projections.0.bias": (1536,),
          "nfr_loss.projections.1.weight": (768, D), "nfr_loss.projections.1.bias": (768,)}
    for l in range(L):
        p = f"bert.encoder.layer.{l}."
        for m in ("query", "key", "value", "ctx_attention"):
            sh[p + f"attention.self.{m}.weight"] = (D, D)
            sh[p + f"attention.self.{m}.bias"] = (D,)
        sh[p + "attention.output.dense.weight"] = (D, D); sh[p + "attention.output.dense.bias"] = (D,)
        sh[p + "attention.output.LayerNorm.weight"] = (D,); sh[p + "attention.output.LayerNorm.bias"] = (D,)
        sh[p + "intermediate.dense.weight"] = (I, D); sh[p + "intermediate.dense.bias"] = (I,)
        sh[p + "output.dense.weight"] = (D, I); sh[p + "output.dense.bias"] = (D,)
        sh[p + "output.LayerNorm.weight"] = (D,); sh[p + "output.LayerNorm.bias"] = (D,)
    return cfg, sh


def check_against_golden(named, gold):
    """Same distribution as the reference: constants are exact; random tensors agree in std within 6 standard errors of
    the sample std (two independent draws), in mean within 6 standard errors, and respect the same support."""
    assert set(named) == set(gold), set(named) ^ set(gold)
    bound = 1.0 / math.sqrt(D)
    for n, t in named.items():
        t = t.detach().float().cpu().flatten()
        g = gold[n]
        k = t.numel()
        if g["std"] == 0.0:                                   # biases 0, LayerNorm (1, 0): exact
            assert float(t.std(unbiased=False)) == 0.0 and float(t.mean()) == g["mean"], n
            continue
        se_std = g["std"] / math.sqrt(2 * k) * math.sqrt(2)    # difference of two sample stds
        se_mean = g["std"] / math.sqrt(k) * math.sqrt(2)
        uniform = n.startswith("nfr_loss.")
        if uniform:
            se_std *= 0.7                                      # kurtosis of U is lower; keep the normal bound (looser)
            assert float(t.abs().max()) <= bound + 1e-7 and g["absmax"] <= bound + 1e-7, n
            assert float(t.abs().max()) > 0.9 * bound, n       # fills the support (a N(0, 0.02) would not)
        else:
            assert 2.0 * g["std"] < float(t.abs().max()) < 7.0 * g["std"], n      # unbounded normal tails, std 0.02
        assert abs(float(t.std(unbiased=False)) - g["std"]) < 6 * se_std + 1e-9, (n, float(t.std()), g["std"])
        assert abs(float(t.mean()) - g["mean"]) < 6 * se_mean + 1e-9, (n, float(t.mean()), g["mean"])


def test_init_rule_matches_reference_statistics():
    """CPU: the rule `reference_init` applies per parameter (pmgt_amd.models.init_value), on the reference's shapes."""
    from pmgt_amd.models import init_value
    gold = golden()
    cfg, sh = shapes()
    assert set(sh) == set(gold)
    gen = torch.Generator().manual_seed(123)
    check_against_golden({n: init_value(n, s, cfg, gen) for n, s in sh.items()}, gold)
    assert abs(gold["bert.embeddings.feat_linear.0.weight"]["std"] - 0.02) < 1e-3             # the [probe] of SURVEY a15:
    assert abs(gold["nfr_loss.projections.0.weight"]["std"] - 1 / math.sqrt(3 * D)) < 1e-3    # std 0.020 vs 0.051


@pytest.mark.gpu
def test_engine_and_module_init_match_reference_statistics():
    from pmgt_amd.engine import Engine
    from pmgt_amd.models import PMGT, reference_init
    gold = golden()
    cfg, sh = shapes()
    eng = Engine(cfg, dtype="fp32")
    assert {e["name"]: tuple(e["shape"]) for e in eng.entries} == sh      # parameter inventory == the reference's named_parameters
    reference_init(eng, seed=7)
    check_against_golden(eng.named_views(), gold)
    model = PMGT(node_size=N, config=cfg, dtype="fp32")                  # constructor path: bert._init_weights + nn.Linear default
    named = {n: p for n, p in model.named_parameters() if p.requires_grad}
    check_against_golden(named, gold)
