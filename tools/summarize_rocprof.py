#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace / PMC) per kernel name.

  summarize_rocprof.py stats <dir>            calls, total ms, avg us, share from *kernel_trace.csv
  summarize_rocprof.py pmc <dir> <COUNTER>    average counter value per dispatch from *counter_collection.csv
                                              (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
                                              under-reports wide coalesced reads by 2x, see MI355X_MICROARCH.md)
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\[clone.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def find(d, pat):
    return [p for p in glob.glob(os.path.join(d, "**", pat), recursive=True)]


def stats(d):
    files = find(d, "*kernel_trace.csv")
    if not files:
        print("no kernel_trace.csv under", d)
        return
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            n = short(row.get("Kernel_Name", "?"))
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            agg[n][0] += 1
            agg[n][1] += dur
    tot = sum(v[1] for v in agg.values())
    print(f"{'kernel':112s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'share':>7s}")
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:112s} {c:7d} {us / 1e3:10.3f} {us / c:10.2f} {us / tot:7.3f}")
    print(f"TOTAL kernel time {tot / 1e3:.3f} ms over {sum(v[0] for v in agg.values())} dispatches")


def pmc(d, counter):
    files = find(d, "*counter_collection.csv")
    if not files:
        print("no counter_collection.csv under", d)
        return
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            n = short(row.get("Kernel_Name", "?"))
            agg[n][0] += 1
            agg[n][1] += float(row["Counter_Value"])
    print(f"{'kernel':112s} {'dispatches':>10s} {'avg_' + counter:>16s}")
    for n, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:112s} {c:10d} {v / c:16.1f}")


def pmcseq(d, counter):
    """Every dispatch of the run in dispatch order: "<dispatch id>\t<counter value>\t<kernel>" (tools/make_traffic.py attributes them to
    the bench's phases by walking this list next to the engine's phase sequence)."""
    rows = []
    for f in find(d, "*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"]), short(row.get("Kernel_Name", "?"))))
    for did, v, n in sorted(rows):
        print(f"{did}\t{v:.1f}\t{n}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "pmcseq":
        pmcseq(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3])
