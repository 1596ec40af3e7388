#!/usr/bin/env python3
"""profiles/rNN/README.md (per-kernel tables) from a committed profile set, followed by the hand-written profiles/rNN/NOTES.md when there
is one.  Usage: tools/make_profile_readme.py profiles/r05/a

A set is what tools/gpu_collect.sh leaves: <pre>_kernel_stats.txt + <pre>_pmc_*.txt for the headline workload, and the same files behind a
second prefix for every other workload that was profiled (<pre>_c2_i1024_*, <pre>_c2_beta1_*, <pre>_c4_*)."""
import json
import os
import re
import sys

STEPS = 13          # tools/gpu_profile.sh: --steps 10 --warmup 3
PMC_STEPS = 3       # counter passes: --steps 2 --warmup 1


def parse(path):
    rows = {}
    if not os.path.exists(path):
        return rows
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"^(.*?)\s+(\d+)\s+([0-9.]+)\s*$", line)
        if m:
            rows[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return rows


def stats(path):
    ks = {}
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"^(.*?)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line)
        if m:
            ks[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    return ks


def table(pre, title, traffic_file, rows=22, lds=False):
    """One workload's table; pre = file prefix including the trailing underscore."""
    if not os.path.exists(pre + "kernel_stats.txt"):
        return []
    ks = stats(pre + "kernel_stats.txt")
    f, w = parse(pre + "pmc_fetch_size.txt"), parse(pre + "pmc_write_size.txt")
    busy, cu = parse(pre + "pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt"), parse(pre + "pmc_SQ_BUSY_CU_CYCLES.txt")
    valu, mf = parse(pre + "pmc_SQ_INSTS_VALU.txt"), parse(pre + "pmc_SQ_INSTS_MFMA.txt")
    lc, la = parse(pre + "pmc_SQ_LDS_BANK_CONFLICT.txt"), parse(pre + "pmc_SQ_LDS_IDX_ACTIVE.txt")
    base = os.path.basename(pre)
    out = [f"## {title}\n", f"`{base}kernel_stats.txt`, `{base}pmc_*.txt`.\n",
           "| kernel | launches / step | avg us | ms / step | HBM MB / launch | TB/s | matrix pipe busy | VALU per MFMA |" + (" LDS busy | conflict share |" if lds else ""),
           "|---|---|---|---|---|---|---|---|" + ("---|---|" if lds else "")]
    n = 0
    for k, (calls, tms, avg) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
        if "pmgt" not in k:
            continue
        n += 1
        if n > rows:
            break
        mb = (2 * f[k][1] + w.get(k, (0, 0))[1]) * 1024 / 1e6 if k in f else None
        b, c = busy.get(k, (0, 0))[1], cu.get(k, (0, 0))[1]
        name = re.sub(r"^_ZN4pmgt\d+", "", re.sub(r"\(.*", "", k).replace("pmgt::", ""))
        line = (f"| `{name[:48]}` | {calls / STEPS:.1f} | {avg:.1f} | {tms / STEPS:.3f} | {'%.0f' % mb if mb else '-'} | "
                f"{'%.2f' % (mb / avg) if mb else '-'} | {'%.2f' % (b / (4 * c)) if c else '-'} | "
                f"{'%.1f' % (valu[k][1] / mf[k][1]) if k in mf and mf[k][1] > 0 else '-'} |")
        if lds:
            line += (f" {'%.2f' % (la[k][1] / c) if k in la and c else '-'} | "
                     f"{'%.2f' % (lc[k][1] / la[k][1]) if k in la and la[k][1] > 0 else '-'} |")
        out.append(line)
    tail = f"\nWhole step: {sum(v[1] for v in ks.values()) / STEPS:.2f} ms of kernel time under the profiler"
    if traffic_file and os.path.exists(traffic_file):
        tr = json.load(open(traffic_file))
        tail += f", **{tr['step_hbm_gb']} GB of HBM traffic per step** by the counters ({tr.get('step_hbm_gb_method', '')}; `{os.path.relpath(traffic_file, os.path.dirname(os.path.dirname(pre)))}`)"
    out.append(tail + ".\n")
    return out


def phase_table(traffic_file, detail_file, title):
    """Per PHASE of the bench (profiles/traffic*.json: the counter runs' dispatches attributed to the engine's phase order, round 6): median
    launch time from the bench's own HIP events, PMC bytes of the median launch, the algorithmic bytes / flops of bench.phase_work."""
    if not (os.path.exists(traffic_file) and os.path.exists(detail_file)):
        return []
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    tr, d = json.load(open(traffic_file)), json.load(open(detail_file))
    if "per dispatch" not in tr.get("phase_attribution", ""):
        return []
    wl = d["config"]["workload"]
    g = lambda key: int(re.search(key + r"=(\d+)", wl).group(1))
    L, H, dd, I, S, B = g("L"), g("H"), g("d"), g("I"), g("S"), g("B")
    M = 12 * B * S
    vc = "beta 1," in wl
    out = [f"### {title}: per phase (`{os.path.basename(traffic_file)}`, `{os.path.basename(detail_file)}`)\n",
           "| phase | launches / step | median launch us | PMC MB (median launch) | algorithmic MB | PMC / algorithmic | TB/s (algorithmic) | TFLOP/s | matrix pipe busy |",
           "|---|---|---|---|---|---|---|---|---|"]
    for ph, p in d["phases"].items():
        t = tr["phases"].get(ph)
        w = bench.phase_work(ph, M, dd, I, S, H, 1536, 768, B * (S - 1), 2, vc)
        if t is None or w is None or p["median_launch_ms"] <= 0:
            continue
        fl, by = w
        if t["hbm_mb_per_launch"] < 0.5 * by / 1e6:       # the per-token model does not describe this launch (table mode: once per node; compacted rows)
            continue
        us = p["median_launch_ms"] * 1e3
        out.append(f"| `{ph}` | {p['launches_per_step']} | {us:.1f} | {t['hbm_mb_per_launch']:.0f} | {by / 1e6:.0f} | {t['hbm_mb_per_launch'] / (by / 1e6):.2f} | "
                   f"{by / 1e6 / us:.2f} | {fl / 1e6 / us:.0f} | {t.get('matrix_pipe_busy', '-')} |")
    out.append("")
    return out


def main():
    pre = sys.argv[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(pre)))
    d = os.path.dirname(pre)
    out = [
        "# %s -- one MI355X, bf16, dropout 0.1\n" % d,
        "Files: `a_kernel_stats.txt` (`rocprofv3 --kernel-trace --stats` of `bench.py --steps 10 --warmup 3`: 13 steps), `a_pmc_*.txt` (separate "
        "`--pmc` passes of 3 steps, and of 6 steps for `*_6steps.txt`: FETCH_SIZE / WRITE_SIZE in KiB per launch; SQ counters), the same behind "
        "`a_c2_i1024_` (C2 with I = 4d), `a_c2_beta1_` (C2 at beta = 1), `a_c4_` (C4 shapes, B = 256), `a_c4_i2048_`, `a_c4_b1024_` where collected; `*_pmc_{fetch,write}_seq.txt` = the same counter runs dispatch by dispatch and `*_detail.json` = a bench record with the phase order of a step (round 6: per-phase attribution, `tools/make_traffic.py`); `a_bench_*.json` (bench lines: the default "
        "line with its `workloads`, C3 graph, C2 in fp8 mode, the one-rank RCCL runs, a 4-rank gloo rehearsal on one GPU), `a_kernel_stats_B32.txt` / "
        "`_B256.txt` (the reference's own batch sizes). Collected by `tools/gpu_collect.sh <tag> lines | c2 | i4d | c4 | c4i | c4b | small` -> `tools/gpu_profile.sh`; "
        "`profiles/traffic*.json` come from these sets (`tools/make_traffic.py`), the tables of this file from `tools/make_profile_readme.py`, the "
        "notes behind them from `NOTES.md`; `trace_*.txt` and the A/B files are named in the notes.\n",
        "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md). Matrix pipe busy = "
        "SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES). Per-launch HBM bytes are what the counters saw DURING the launch: the 256-MB Infinity "
        "Cache moves write-backs of one launch into the next, so a writer can show fewer bytes than it produced.\n"]
    out += table(pre + "_", "C2 (7 252 nodes, L4 H8 d256 I256 S32), B = 1 024 -- the headline", os.path.join(root, "traffic.json"))
    out += phase_table(os.path.join(root, "traffic.json"), pre + "_detail.json", "C2")
    out += table(pre + "_c2_i1024_", "C2 with I = 4d = 1 024, B = 1 024 (bench `c2_i1024`)", os.path.join(root, "traffic_c2_i1024.json"), rows=14)
    out += phase_table(os.path.join(root, "traffic_c2_i1024.json"), pre + "_c2_i1024_detail.json", "C2, I = 4d")
    out += table(pre + "_c2_beta1_", "C2 at beta = 1 (bench `c2_beta1`: the vc_only kernels)", os.path.join(root, "traffic_c2_beta1.json"), rows=8)
    out += phase_table(os.path.join(root, "traffic_c2_beta1.json"), pre + "_c2_beta1_detail.json", "C2, beta = 1")
    out += table(pre + "_c4_", "C4 shapes (10^6 nodes, L6 H8 d512 I512 S64), B = 256", os.path.join(root, "traffic_c4.json"), rows=16, lds=True)
    out += phase_table(os.path.join(root, "traffic_c4.json"), pre + "_c4_detail.json", "C4 shapes")
    out += table(pre + "_c4_i2048_", "C4 shapes with I = 4d = 2 048, B = 256 (bench `c4_i2048`)", os.path.join(root, "traffic_c4_i2048.json"), rows=12)
    out += phase_table(os.path.join(root, "traffic_c4_i2048.json"), pre + "_c4_i2048_detail.json", "C4 shapes, I = 4d")
    out += table(pre + "_c4_b1024_", "C4 shapes, B = 1 024 (bench `c4_bf16_b1024`)", os.path.join(root, "traffic_c4_b1024.json"), rows=10)
    out += phase_table(os.path.join(root, "traffic_c4_b1024.json"), pre + "_c4_b1024_detail.json", "C4 shapes, B = 1 024")
    notes = os.path.join(d, "NOTES.md")
    if os.path.exists(notes):
        out.append(open(notes).read())
    open(os.path.join(d, "README.md"), "w").write("\n".join(out))
    print("\n".join(out[4:20]))


if __name__ == "__main__":
    main()
