#!/usr/bin/env python3
"""profiles/rNN/README.md (per-kernel table) from a committed profile set.  Usage: tools/make_profile_readme.py profiles/r03/a"""
import json
import os
import re
import sys


def parse(path):
    rows = {}
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"^(.*?)\s+(\d+)\s+([0-9.]+)\s*$", line)
        if m:
            rows[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return rows


def main():
    pre = sys.argv[1]
    ks = {}
    steps = 13          # tools/gpu_profile.sh: --steps 10 --warmup 3
    for line in open(pre + "_kernel_stats.txt").read().splitlines()[1:]:
        m = re.match(r"^(.*?)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line)
        if m:
            ks[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    f, w = parse(pre + "_pmc_fetch_size.txt"), parse(pre + "_pmc_write_size.txt")
    busy, cu = parse(pre + "_pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt"), parse(pre + "_pmc_SQ_BUSY_CU_CYCLES.txt")
    valu, mf = parse(pre + "_pmc_SQ_INSTS_VALU.txt"), parse(pre + "_pmc_SQ_INSTS_MFMA.txt")
    tr = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(pre))), "traffic.json")))
    out = [
        "# %s -- one MI355X, C2 (7 252 nodes, L4 H8 d256 S32), bf16, B = 1 024, dropout 0.1\n" % os.path.dirname(pre),
        "Files: `a_kernel_stats.txt` (`rocprofv3 --kernel-trace --stats` of `bench.py --steps 10 --warmup 3`: 13 steps), `a_pmc_*.txt` (separate "
        "`--pmc` passes of 3 steps: FETCH_SIZE / WRITE_SIZE in KiB per launch; SQ counters), `a_bench_*.json` (bench lines: default C2, C3 "
        "graph, C4 / C5 shapes, C2 in fp8 mode, a 4-rank gloo rehearsal on one GPU), `a_kernel_stats_B32.txt` / `_B256.txt` (the "
        "reference's own batch sizes), `macro_tile_experiment.txt` (two kernel designs that were measured and not kept). Collected by "
        "`tools/gpu_profile.sh`; `profiles/traffic.json` comes from this set (`tools/make_traffic.py`), this file from `tools/make_profile_readme.py`.\n",
        "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md). Matrix pipe busy = "
        "SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES).\n",
        "| kernel | launches / step | avg us | ms / step | HBM MB / launch | TB/s | matrix pipe busy | VALU per MFMA |", "|---|---|---|---|---|---|---|---|"]
    for k, (calls, tms, avg) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:24]:
        if "pmgt" not in k:
            continue
        mb = (2 * f[k][1] + w.get(k, (0, 0))[1]) * 1024 / 1e6 if k in f else None
        b, c = busy.get(k, (0, 0))[1], cu.get(k, (0, 0))[1]
        name = re.sub(r"^_ZN4pmgt\d+", "", re.sub(r"\(.*", "", k).replace("pmgt::", ""))
        out.append(f"| `{name[:48]}` | {calls / steps:.1f} | {avg:.1f} | {tms / steps:.3f} | {'%.0f' % mb if mb else '-'} | "
                   f"{'%.2f' % (mb / avg) if mb else '-'} | {'%.2f' % (b / (4 * c)) if c else '-'} | "
                   f"{'%.1f' % (valu[k][1] / mf[k][1]) if k in mf and mf[k][1] > 0 else '-'} |")
    out.append(f"\nWhole step: {sum(v[1] for v in ks.values()) / steps:.2f} ms of kernel time under the profiler, **{tr['step_hbm_gb']} GB of HBM "
               f"traffic per step** by the counters (every kernel's dispatches x its average, per step).\n")
    open(os.path.join(os.path.dirname(pre), "README.md"), "w").write("\n".join(out))
    print("\n".join(out[4:16]))


if __name__ == "__main__":
    main()
