#!/usr/bin/env python3
"""Soak run of the training loop a user of the reference would run (pmgt/base_trainer.py:309-322: fit = train steps + periodic validation):
C2 workload, B = 1 024, bf16, live sampler (graph replay per slot), `blocks` blocks of `steps` steps; after each block the validation pass
(loss/val, val/auc on held-out nodes) and the block's step time.  Checks along the way: finite loss, loss/val falling, val/auc rising above
chance, step time flat from block to block (no leak, no drift into a slower path).  python tools/soak.py [blocks=6] [steps=500]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import WORKLOADS, host_cpu_share  # noqa: E402
from pmgt_amd.configuration_pmgt import PMGTConfig  # noqa: E402
from pmgt_amd.datasets import MCNSampler  # noqa: E402
from pmgt_amd.engine import Engine  # noqa: E402
from pmgt_amd.graph import synthetic_graph  # noqa: E402
from pmgt_amd.models import reference_init, synthetic_features  # noqa: E402
from pmgt_amd.trainer import Trainer, evaluate  # noqa: E402

kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
blocks, steps = int(kv.get("blocks", 6)), int(kv.get("steps", 500))
nodes, edges, L, H, d, I, S = WORKLOADS["c2"]
B = 1024
cfg = PMGTConfig(hidden_size=d, num_hidden_layers=L, num_attention_heads=H, intermediate_size=I, hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, beta=0.5)
graph = synthetic_graph(nodes, edges, seed=0)
vis, txt = synthetic_features(nodes, seed=0)
eng = Engine(cfg, dtype="bf16", device="cuda:0", seed=1234)
reference_init(eng, seed=0)
eng.set_tables(vis, txt)
trainer = Trainer(eng, lr=3e-4, weight_decay=1e-2, max_grad_norm=5.0)
sampler = MCNSampler(graph, max_ctx_neigh=S - 1)
perm = np.random.RandomState(0).permutation(nodes) + 2
valid, train = perm[:512], perm[512:]            # (pmgt/pmgt/utils.py train_valid_split semantics: disjoint node id sets)
threads = max(4, int(host_cpu_share()) - 3)
rows = []
v0 = evaluate(eng, sampler, valid, batch_size=256, threads=threads)
print(json.dumps({"block": 0, **{k: round(v, 5) for k, v in v0.items()}}), flush=True)
mem0 = torch.cuda.memory_allocated()
for blk in range(1, blocks + 1):
    t0 = time.perf_counter()
    r = trainer.run_live(sampler, train, B, steps=steps, threads=threads, depth=3, graphs=True)
    wall = time.perf_counter() - t0
    loss = float(trainer.last_loss.item())
    v = evaluate(eng, sampler, valid, batch_size=256, threads=threads)
    row = {"block": blk, "steps_total": blk * steps, "train_loss_last": round(loss, 5), **{k: round(x, 5) for k, x in v.items()},
           "ms_per_step": r["ms_per_step"], "gpu_step_ms": r["gpu_step_ms"], "gpu_idle_ms_per_step": r["gpu_idle_ms_per_step"],
           "nodes_per_s": round(B / r["ms_per_step"] * 1e3, 1), "block_wall_s": round(wall, 2),
           "grad_norm": round(float(eng.grad_norm().item()), 5), "cuda_mem_mb": round(torch.cuda.memory_allocated() / 2 ** 20, 1)}
    rows.append(row)
    print(json.dumps(row), flush=True)
    assert np.isfinite(loss) and np.isfinite(v["loss/val"]) and torch.isfinite(eng.params).all()
ms = [r_["ms_per_step"] for r_ in rows]
ok = {"finite": True, "val_loss_fell": rows[-1]["loss/val"] < v0["loss/val"], "val_auc_above_chance": rows[-1]["val/auc"] > 0.6,
      "step_time_spread": round(max(ms) / min(ms), 4), "memory_flat": abs(torch.cuda.memory_allocated() - mem0) < 64 * 2 ** 20 or rows[-1]["cuda_mem_mb"] == rows[0]["cuda_mem_mb"]}
print(json.dumps({"summary": ok}))
sys.exit(0 if ok["val_loss_fell"] and ok["val_auc_above_chance"] and ok["step_time_spread"] < 1.1 else 1)
