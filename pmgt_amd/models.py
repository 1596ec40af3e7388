"""Initialisation and workload helpers for the PMGT engine.

`reference_init` reproduces the reference's initial parameter DISTRIBUTIONS:
everything under PMGTModel gets PMGTPretrainedModel._init_weights (pmgt/pmgt/modeling_pmgt.py:44-58:
Linear/Embedding weights N(0, initializer_range), biases 0, LayerNorm (1, 0)); PMGT itself never calls
it, so the NFR projections keep torch's nn.Linear default (kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in))
for weight and bias) — fixture tests/golden/init_stats.npz pins these statistics.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def reference_init(engine, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    std = engine.config.initializer_range
    for e in engine.entries:
        name, shape = e["name"], e["shape"]
        if name.startswith("nfr_loss."):
            fan_in = engine.config.hidden_size
            bound = 1.0 / math.sqrt(fan_in)
            v = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif name.endswith("LayerNorm.weight"):
            v = torch.ones(shape)
        elif name.endswith(".bias"):
            v = torch.zeros(shape)
        else:
            v = torch.randn(shape, generator=g) * std
        engine.view(name).copy_(v)


def synthetic_features(n_nodes: int, feat_sizes=(1536, 768), seed: int = 0):
    """fp32 [N+2, F] ~ N(0,1) with rows 0 (<pad>) and 1 (<mask>) zero (notebooks/PMGT.ipynb cell 30)."""
    rs = np.random.RandomState(seed)
    out = []
    for f in feat_sizes:
        a = rs.standard_normal((n_nodes + 2, f)).astype(np.float32)
        a[:2] = 0
        out.append(a)
    return out


def train_flops_per_node(d, I, L, S, Fv=1536, Ft=768, pairs=10):
    """Algorithmic FLOPs of one pre-training step per target node:
    forward per token 2*F*d + 8d + L(10 d^2 + 4 d I + 6 S d); training = 3x encoder terms + 2x the
    feature projection (frozen tables: no dgrad) + the NFR head (3 * M * 2 d F, M ~ 0.16 (S-1))."""
    F = Fv + Ft
    seqs = pairs + 2
    tok = seqs * S
    enc = L * (10 * d * d + 4 * d * I + 6 * S * d) + 8 * d
    proj = 2 * F * d
    nfr = 3 * 0.16 * (S - 1) * 2 * d * F
    return tok * (3 * enc + 2 * proj) + nfr
