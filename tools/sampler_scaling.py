#!/usr/bin/env python3
"""Host MCNSampling throughput against the thread count (C2 graph, B = 1024 targets per call, train mode)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pmgt_amd.datasets import MODE_TRAIN, MCNSampler  # noqa: E402
from pmgt_amd.graph import synthetic_graph  # noqa: E402

g = synthetic_graph(7252, 88606, seed=0)
s = MCNSampler(g, max_ctx_neigh=31)
tg = np.random.RandomState(0).permutation(7252)[:1024] + 2
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 32, 64, 128]:
    s.batch(tg, MODE_TRAIN, threads=th, base_seed=0, counter=0)
    t0 = time.time()
    for r in range(3):
        s.batch(tg, MODE_TRAIN, threads=th, base_seed=0, counter=1024 * (r + 1))
    dt = (time.time() - t0) / 3
    print(f"{th:4d} threads  {1024 / dt:10.0f} nodes/s  {dt * 1e3:8.1f} ms per batch  {dt / 1024 * 1e6 * th:8.0f} us/node/thread", flush=True)
