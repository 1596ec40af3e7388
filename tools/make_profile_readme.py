#!/usr/bin/env python3
"""profiles/rNN/README.md (per-kernel table) from a committed profile set, followed by the hand-written profiles/rNN/NOTES.md when there
is one.  Usage: tools/make_profile_readme.py profiles/r04/a"""
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
        "`--pmc` passes of 3 steps, and of 6 steps for `*_6steps.txt`: FETCH_SIZE / WRITE_SIZE in KiB per launch; SQ counters), `a_bench_*.json` (bench lines: default C2, C3 "
        "graph, C4 / C5 shapes, C2 in fp8 mode, a 4-rank gloo rehearsal on one GPU), `a_kernel_stats_B32.txt` / `_B256.txt` (the "
        "reference's own batch sizes). Collected by `tools/gpu_collect.sh` -> `tools/gpu_profile.sh`; `profiles/traffic.json` comes from this set "
        "(`tools/make_traffic.py`), the tables of this file from `tools/make_profile_readme.py`, the notes behind them from `NOTES.md`.\n",
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
    how = tr.get("step_hbm_gb_method", "every kernel's dispatches x its average, per step")
    out.append(f"\nWhole step: {sum(v[1] for v in ks.values()) / steps:.2f} ms of kernel time under the profiler, **{tr['step_hbm_gb']} GB of HBM "
               f"traffic per step** by the counters ({how}).\n")
    # ---- C4 shapes (10^6 nodes, L6 d512 S64, B = 256): same passes with BENCH_EXTRA="--workload c4 --batch 256"
    c4 = pre + "_kernel_stats_c4shapes_B256.txt"
    if os.path.exists(c4):
        ks4 = {}
        for line in open(c4).read().splitlines()[1:]:
            m = re.match(r"^(.*?)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line)
            if m:
                ks4[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
        q = pre + "_c4shapes_pmc_"
        f4, w4 = parse(q + "fetch.txt"), parse(q + "write.txt")
        b4, c4u = parse(q + "SQ_VALU_MFMA_BUSY_CYCLES.txt"), parse(q + "SQ_BUSY_CU_CYCLES.txt")
        v4, m4 = parse(q + "SQ_INSTS_VALU.txt"), parse(q + "SQ_INSTS_MFMA.txt")
        lc, la = parse(q + "SQ_LDS_BANK_CONFLICT.txt"), parse(q + "SQ_LDS_IDX_ACTIVE.txt")
        out += ["## C4 shapes (10^6 nodes, L6 H8 d512 I512 S64), bf16, B = 256\n",
                "`a_kernel_stats_c4shapes_B256.txt`, `a_c4shapes_pmc_*.txt` (same passes, `BENCH_EXTRA=\"--workload c4 --batch 256\"`). LDS busy = "
                "SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES; conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.\n",
                "| kernel | launches / step | avg us | ms / step | HBM MB / launch | TB/s | matrix pipe busy | VALU per MFMA | LDS busy | conflict share |",
                "|---|---|---|---|---|---|---|---|---|---|"]
        tot_gb = 0.0
        for k, (calls, tms, avg) in sorted(ks4.items(), key=lambda kv: -kv[1][1]):
            if k in f4:
                tot_gb += (2 * f4[k][1] + w4.get(k, (0, 0))[1]) * 1024 / 1e9 * f4[k][0] / 3.0       # counter passes: 3 steps
        for k, (calls, tms, avg) in sorted(ks4.items(), key=lambda kv: -kv[1][1])[:16]:
            if "pmgt" not in k:
                continue
            mb = (2 * f4[k][1] + w4.get(k, (0, 0))[1]) * 1024 / 1e6 if k in f4 else None
            b, c = b4.get(k, (0, 0))[1], c4u.get(k, (0, 0))[1]
            name = re.sub(r"^_ZN4pmgt\d+", "", re.sub(r"\(.*", "", k).replace("pmgt::", ""))
            out.append(f"| `{name[:48]}` | {calls / steps:.1f} | {avg:.1f} | {tms / steps:.3f} | {'%.0f' % mb if mb else '-'} | "
                       f"{'%.2f' % (mb / avg) if mb else '-'} | {'%.2f' % (b / (4 * c)) if c else '-'} | "
                       f"{'%.1f' % (v4[k][1] / m4[k][1]) if k in m4 and m4[k][1] > 0 else '-'} | "
                       f"{'%.2f' % (la[k][1] / c) if k in la and c else '-'} | {'%.2f' % (lc[k][1] / la[k][1]) if k in la and la[k][1] > 0 else '-'} |")
        out.append(f"\nWhole step: {sum(v[1] for v in ks4.values()) / steps:.2f} ms of kernel time under the profiler, {tot_gb:.1f} GB of HBM traffic per step.\n")
    notes = os.path.join(os.path.dirname(pre), "NOTES.md")
    if os.path.exists(notes):
        out.append(open(notes).read())
    open(os.path.join(os.path.dirname(pre), "README.md"), "w").write("\n".join(out))
    print("\n".join(out[4:16]))


if __name__ == "__main__":
    main()
