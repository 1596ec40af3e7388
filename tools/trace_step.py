#!/usr/bin/env python3
"""Which kernel families one training step dispatches to, and their per-phase times: launch trace (pmgt_launch_trace_count) + phase timers.
   python tools/trace_step.py --workload c2 --intermediate 1024 --batch 1024"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAMILIES = ["gemm_wsr", "gemm_wsr_lnb", "gemm_wsr512", "gemm_ws", "nt_big", "nt_big_gather", "nt_big_128", "nt_lnb", "nt_lnf", "nt_tile",
            "tn_big", "tn_big_gather", "tn_dma", "tn_dma_gather", "tn_tile", "attn_tiles_fwd", "attn_tiles_bwd", "qkvc_attn_fwd",
            "attn_bwd_wgrad", "f8_big", "f8_tile", "f8_wsr512", "gemm_rowln", "embed_tok8", "qkvc_attn_fwd_vc", "attn_bwd_wgrad_vc", "attn_bwd_wgrad_vc2", "nt_vc"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=7252)
    ap.add_argument("--edges", type=int, default=88606)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--intermediate", type=int, default=256)
    ap.add_argument("--seq", type=int, default=32)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--beta", type=float, default=0.5)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--engine-option", action="append", default=[])
    a = ap.parse_args()
    from pmgt_amd import _lib
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.datasets import MODE_TRAIN, MCNSampler
    from pmgt_amd.engine import Engine
    from pmgt_amd.graph import synthetic_graph
    from pmgt_amd.models import reference_init, synthetic_features, train_flops_per_node
    from pmgt_amd.trainer import Trainer
    graph = synthetic_graph(a.nodes, a.edges, seed=0)
    eng = Engine(PMGTConfig(hidden_size=a.hidden, num_hidden_layers=a.layers, num_attention_heads=a.heads, intermediate_size=a.intermediate,
                            hidden_dropout_prob=a.dropout, attention_probs_dropout_prob=a.dropout, beta=a.beta), dtype=a.dtype, seed=1)
    for k in a.engine_option:
        eng.set_option(k, 1)
    reference_init(eng, 0)
    eng.set_tables(*synthetic_features(a.nodes, seed=0))
    smp = MCNSampler(graph, a.seq - 1)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}
    staged = []
    for i in range(3):
        tg = np.resize(np.arange(2, a.nodes + 2)[(i * a.batch) % a.nodes:], a.batch)
        tgt, pair, num_pairs, labels = smp.batch(tg, MODE_TRAIN, threads=8, base_seed=0, counter=i * a.batch)
        staged.append((cu(tgt), cu(pair), num_pairs.cuda(), labels.cuda()))
    tr = Trainer(eng, lr=1e-4, max_grad_norm=5.0)
    for i in range(3):
        tr.train_step(staged[i % 3])
    torch.cuda.synchronize()
    H = _lib.hip()
    H.pmgt_launch_trace_reset()
    tr.train_step(staged[0])
    torch.cuda.synchronize()
    trace = {f: int(H.pmgt_launch_trace_count(f.encode())) for f in FAMILIES}
    trace = {k: v for k, v in trace.items() if v > 0}
    import time
    t0 = time.perf_counter()
    for i in range(a.steps):
        tr.train_step(staged[i % 3])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    eng.profile_begin()
    for i in range(3):
        tr.train_step(staged[i % 3])
    prof = eng.profile_end()
    phases = {k: (c // 3, round(v / 3, 4), round(v / c * 1e3, 1)) for k, (c, v) in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    fl = train_flops_per_node(a.hidden, a.intermediate, a.layers, a.seq)
    print(json.dumps({"ms_per_step": round(ms, 4), "nodes_per_s": round(a.batch / ms * 1e3, 1), "gflop_per_node": round(fl / 1e9, 3),
                      "mfma_util": round(a.batch / ms * 1e3 * fl / 2.5e15, 4), "loss": tr.last_loss.item(), "launch_trace": trace}))
    print("phase                          launches/step  ms/step  us/launch")
    for k, (c, v, u) in phases.items():
        print(f"{k:32s} {c:6d} {v:10.4f} {u:10.1f}")


if __name__ == "__main__":
    main()
