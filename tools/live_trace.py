#!/usr/bin/env python3
"""Where the live input pipeline loses against pre-staged steps, from a rocprofv3 --kernel-trace --memory-copy-trace of bench.py:
steps are delimited by the optimizer kernel (adamw_kernel); for the pre-staged region (the bench's timed steps) and the live region
(the end_to_end pass) it reports, per step: wall time, the sum of kernel durations, the GPU idle time INSIDE the step (gaps between
consecutive kernels: the launch thread late) and the slowest kernels' average durations in both regions (kernels themselves slower:
clock / memory contention), plus where the H2D copies sit.
    python tools/live_trace.py <rocprof output dir> <n_prestaged_steps incl. warmup> <n_live_steps>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
n_pre, n_live = int(sys.argv[2]), int(sys.argv[3])
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
copies = []
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), int(r.get("Bytes", r.get("Size", 0)) or 0)))
steps = []
cur = []
for s, e, n in rows:
    cur.append((s, e, n))
    if "adamw_kernel" in n:
        steps.append(cur)
        cur = []
print(f"{len(rows)} kernel dispatches, {len(steps)} optimizer steps, {len(copies)} copies")


def region(st, label):
    if not st:
        return
    wall = [(b[-1][1] - a[-1][1]) / 1e6 for a, b in zip(st[:-1], st[1:])]
    busy = [sum(e - s for s, e, _ in b) / 1e6 for b in st[1:]]
    gaps = [sum(max(0, b[i][0] - b[i - 1][1]) for i in range(1, len(b))) / 1e6 for b in st[1:]]
    lead = [max(0, b[0][0] - a[-1][1]) / 1e6 for a, b in zip(st[:-1], st[1:])]
    m = lambda x: sum(x) / max(len(x), 1)
    print(f"{label}: {len(st)} steps: wall {m(wall):.3f} ms/step, kernel time {m(busy):.3f}, gaps inside the step {m(gaps):.3f}, "
          f"gap before the step's first kernel {m(lead):.3f}; worst step wall {max(wall):.3f}")
    per = defaultdict(lambda: [0, 0.0])
    for b in st:
        for s, e, n in b:
            per[n[:60]][0] += 1
            per[n[:60]][1] += (e - s) / 1e3
    return {k: v[1] / v[0] for k, v in per.items()}, {k: v[1] / len(st) for k, v in per.items()}


pre = steps[max(0, n_pre - 20):n_pre]
live = steps[-n_live:]
a = region(pre, "pre-staged")
b = region(live, "live")
if a and b:
    print("kernels by time per step (avg us per launch: pre-staged -> live):")
    for k, v in sorted(a[1].items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {k:60s} {a[0][k]:8.1f} -> {b[0].get(k, float('nan')):8.1f}")
if copies and live:
    t0, t1 = live[0][0][0], live[-1][-1][1]
    inside = [c for c in copies if t0 <= c[0] <= t1]
    print(f"copies during the live region: {len(inside)} ({sum(c[3] for c in inside) / 1e6:.1f} MB, {sum(c[1] - c[0] for c in inside) / 1e6:.3f} ms of copy time in total)")
