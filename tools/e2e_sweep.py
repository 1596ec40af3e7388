#!/usr/bin/env python3
"""Live input pipeline (Trainer.run_live) against sampler threads and pipeline depth, several repetitions in ONE process on one box
(C2 workload, B = 1024): the rate relative to pre-staged steps, GPU idle time between steps, and the producer's time split.
    python tools/e2e_sweep.py [threads,threads,...] [depth,depth,...] [reps] [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import WORKLOADS, host_cpu_share, time_steps  # noqa: E402
from pmgt_amd.configuration_pmgt import PMGTConfig  # noqa: E402
from pmgt_amd.datasets import MODE_TRAIN, MCNSampler  # noqa: E402
from pmgt_amd.engine import Engine  # noqa: E402
from pmgt_amd.graph import synthetic_graph  # noqa: E402
from pmgt_amd.models import reference_init, synthetic_features  # noqa: E402
from pmgt_amd.parallel import shard_indices  # noqa: E402
from pmgt_amd.trainer import Trainer  # noqa: E402

threads = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,12,13,14").split(",")]
depths = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3,4").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
nodes, edges, L, H, d, I, S = WORKLOADS["c2"]
B = 1024
cfg = PMGTConfig(hidden_size=d, num_hidden_layers=L, num_attention_heads=H, intermediate_size=I, hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, beta=0.5)
graph = synthetic_graph(nodes, edges, seed=0)
vis, txt = synthetic_features(nodes, seed=0)
eng = Engine(cfg, dtype="bf16", device="cuda:0", seed=1234)
reference_init(eng, seed=0)
eng.set_tables(vis, txt)
trainer = Trainer(eng, lr=1e-4, weight_decay=1e-2, max_grad_norm=5.0)
sampler = MCNSampler(graph, max_ctx_neigh=S - 1)
shard = shard_indices(nodes, 0, 1, seed=0, epoch=0) + 2
cu = lambda dct: {k: v.cuda() for k, v in dct.items()}
staged = []
for i in range(8):
    tgt, pair, num_pairs, labels = sampler.batch(np.resize(shard[i * B:], B), MODE_TRAIN, threads=8, base_seed=0, counter=i * B)
    staged.append((cu(tgt), cu(pair), num_pairs.cuda(), labels.cuda()))
pre = time_steps(trainer, staged, steps=30, warmup=5) * 1e3
print(json.dumps({"cpu_share": host_cpu_share(), "prestaged_ms_per_step": round(pre, 3)}), flush=True)
for rep in range(reps):
    for dp in depths:
        for th in threads:
          for gr in (False, True):
            r = trainer.run_live(sampler, shard, B, steps=steps, threads=th, depth=dp, graphs=gr)
            print(json.dumps({"rep": rep, "threads": th, "depth": dp, "graphs": gr, "launch_ms": r["launch_thread_busy_ms_per_step"], "vs_prestaged": round(pre / r["ms_per_step"], 4), "ms_per_step": r["ms_per_step"],
                              "gpu_idle_ms_per_step": r["gpu_idle_ms_per_step"], "gpu_step_ms": r["gpu_step_ms"],
                              "producer": r["producer_ms_per_batch"]}), flush=True)
pre2 = time_steps(trainer, staged, steps=30, warmup=5) * 1e3
print(json.dumps({"prestaged_ms_per_step_after": round(pre2, 3)}))
