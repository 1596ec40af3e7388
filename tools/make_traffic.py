#!/usr/bin/env python3
"""profiles/traffic.json from the committed PMC summaries of one profile set.

Usage: tools/make_traffic.py profiles/rNN/<tag> [kernel_sources_sha] [workload_key batch]   (reads <tag>_pmc_fetch_size.txt and
<tag>_pmc_write_size.txt; the fingerprint defaults to the current sources: run it on the tree the counters were measured on).  With a
workload key other than "c2" (c2_i1024, c4 ...) the result goes to profiles/traffic_<key>.json, which bench.py reads for that workload.

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (counters in KiB; gfx950 tallies 128-B read requests at 64 B,
/opt/skills/guides/MI355X_MICROARCH.md).  bench.py quotes the entry of its dominant phase as `roofline.traffic`."""
import json
import os
import re
import sys

PHASE_KERNELS = {           # bench.py phase name -> substring of the kernel that dominates it
    "bwd.dgrad_qkvc_lnb": "gemm_nt_big_kernel<256, 8, 1>",
    "fwd.gemm_ffn2": "gemm_nt_big_kernel<256, 8, 2>",          # (I = 4 d: the 256 x 256 tile with the LayerNorm epilogue)
    "fwd.gemm_ffn1": "gemm_ws2_kernel<8, 1>",
    "bwd.dgrad_ffn2": "gemm_ws2_kernel<8, 2>",
    "bwd.wgrad_ffn1": "gemm_tn_big_kernel<false>",
    "fwd.gemm_qkvc": "gemm_wsr512_kernel<0>",
    "fwd.attention": "attn_fwd_tiles_kernel",
    "bwd.dgrad_ffn1_lnb": "gemm_wsr_kernel<true>",
    "fwd.qkvc_attention": "qkvc_attn_fwd",
    "bwd.attention": "attn_bwd_tiles_kernel",
    "bwd.attention_wgrad": "attn_bwd_wgrad_kernel",
    "bwd.dgrad_qkvc": "gemm_nt_big_kernel<256, 8, 0>",
    "bwd.wgrad_qkvc": "gemm_tn_big_kernel",
    "bwd.layernorm": "ln_bwd_kernel",
}


def parse(path, counts=None):
    rows = {}
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"^(.*?)\s+(\d+)\s+([0-9.]+)\s*$", line)
        if m:
            rows[m.group(1).strip()] = float(m.group(3))
            if counts is not None:
                counts[m.group(1).strip()] = int(m.group(2))
    return rows


def phase_class(phase):
    """Which kernels a bench phase launches, as (must contain any of, must not contain any of); None = not attributed."""
    if phase.startswith("bwd.wgrad_"):
        return (("gemm_tn",), ())
    if phase.startswith(("fwd.gemm_", "bwd.dgrad_", "loss.gemm_")):
        return (("gemm_",), ("gemm_tn",))
    return {"fwd.qkvc_attention": (("qkvc_attn_fwd",), ()), "fwd.attention": (("attn_fwd",), ()),
            "bwd.attention_wgrad": (("attn_bwd_wgrad",), ()), "bwd.attention": (("attn_bwd",), ("attn_bwd_wgrad",)),
            "fwd.layernorm": (("ln_fwd",), ()), "bwd.layernorm": (("ln_bwd",), ()),
            "fwd.embed_mix": (("embed_mix_fwd", "embed_tok8_fwd"), ()), "bwd.embed_mix": (("embed_mix_bwd", "embed_tok8_bwd"), ())}.get(phase)


def read_seq(path):
    rows = []
    for line in open(path).read().splitlines():
        parts = line.split("\t")
        if len(parts) == 3:
            rows.append((parts[2], float(parts[1])))
    return rows


def steps_of(rows):
    """The dispatch list of a bench run cut into training steps: a step ends with its adamw_kernel; what precedes the first step's weight
    mirror (table casts, parameter init) is dropped.  The first step is the warm-up (first-touch page faults): not used."""
    steps, cur = [], []
    for name, v in rows:
        cur.append((name, v))
        if "adamw_kernel" in name:
            steps.append(cur)
            cur = []
    if steps:
        first = next((i for i, (n, _) in enumerate(steps[0]) if "mirror_kernel" in n), 0)
        steps[0] = steps[0][first:]
    return steps


def attribute(seq_path, phase_sequence):
    """{phase: [[counter value per launch, in launch order] per timed step]} -- every phase record takes the next dispatch of its class."""
    per_phase = {}
    classed = [c for c in (phase_class(p) for p in set(phase_sequence)) if c]
    in_class = lambda n, c: any(a in n for a in c[0]) and not any(b in n for b in c[1])
    for step in steps_of(read_seq(seq_path))[1:]:
        k, got = 0, {}
        for ph in phase_sequence:
            c = phase_class(ph)
            if c is None:
                continue
            while k < len(step) and not in_class(step[k][0], c):
                k += 1
            if k == len(step):
                raise RuntimeError(f"{seq_path}: no dispatch left for phase {ph}")
            got.setdefault(ph, []).append(step[k])
            k += 1
        left = [n for n, _ in step[k:] if any(in_class(n, c) for c in classed)]
        if left:
            raise RuntimeError(f"{seq_path}: {len(left)} classed dispatches not attributed to a phase (first: {left[0]})")
        for ph, lst in got.items():
            per_phase.setdefault(ph, []).append(lst)
    return per_phase


PMC_STEPS = 3        # tools/gpu_profile.sh runs the counter passes with --steps 2 --warmup 1 ...
PMC_STEPS_B = 6      # ... and once more with --steps 5 --warmup 1 (the *_6steps.txt summaries)


def main():
    prefix = sys.argv[1]
    nf, nw = {}, {}
    fetch, write = parse(prefix + "_pmc_fetch_size.txt", nf), parse(prefix + "_pmc_write_size.txt", nw)
    # the whole step = (counter totals of the six-step run - totals of the three-step run) / 3: set-up kernels of the process (table
    # casts, parameter init, the first mirror) are launched equally often in both runs and cancel, whatever their launch count
    six_f, six_w = prefix + "_pmc_fetch_size_6steps.txt", prefix + "_pmc_write_size_6steps.txt"
    if os.path.exists(six_f) and os.path.exists(six_w):
        nf6, nw6 = {}, {}
        fetch6, write6 = parse(six_f, nf6), parse(six_w, nw6)
        tot = lambda v, n, w: sum(w * v[k] * n[k] for k in v)
        step_kib = ((tot(fetch6, nf6, 2.0) + tot(write6, nw6, 1.0)) - (tot(fetch, nf, 2.0) + tot(write, nw, 1.0))) / (PMC_STEPS_B - PMC_STEPS)
        step_how = "difference of the six-step and the three-step counter runs / 3"
    else:       # older profile sets: dispatches per step = floor(launch count / steps of the run) for every kernel
        step_kib = sum(2.0 * fetch[k] * (nf[k] // PMC_STEPS) + write.get(k, 0.0) * (nw.get(k, 0) // PMC_STEPS) for k in fetch)
        step_how = "launch count // 3 dispatches of every kernel per step (three-step counter run)"
    phases = {}
    # exact attribution when the set holds the per-dispatch lists and the phase order of a step: a kernel that serves several phases with
    # different shapes (the d = 512 tiles: one weight-gradient kernel for four GEMM shapes) is then split by phase instead of averaged
    by_phase = None
    try:
        detail = json.load(open(prefix + "_detail.json"))
        seq = detail.get("phase_sequence")
        if seq:
            f_ph, w_ph = attribute(prefix + "_pmc_fetch_seq.txt", seq), attribute(prefix + "_pmc_write_seq.txt", seq)
            by_phase = {}
            for ph in f_ph:
                # per launch slot of the phase (launch order inside a step): mean over the timed steps of 2 x FETCH + WRITE
                n_l = len(f_ph[ph][0])
                mbs = []
                for j in range(n_l):
                    fs = [st[j][1] for st in f_ph[ph] if len(st) == n_l]
                    ws = [st[j][1] for st in w_ph[ph] if len(st) == n_l]
                    mbs.append((2.0 * sum(fs) / len(fs) + sum(ws) / len(ws)) * 1024.0 / 1e6)
                srt = sorted(mbs)
                med = srt[len(srt) // 2]
                kern = f_ph[ph][0][mbs.index(med)][0]        # the kernel of the median launch (the shortcut layer's small launch may take another)
                by_phase[ph] = {"kernel": kern, "launches_per_step": n_l, "hbm_mb_per_launch": round(med, 1),
                                "hbm_mb_by_launch": [round(x, 1) for x in mbs]}
    except (OSError, KeyError, ValueError, RuntimeError) as ex:
        print("per-phase attribution not available:", ex, file=sys.stderr)
    if by_phase is not None:
        phases = by_phase
    for ph, sub in ([] if by_phase is not None else PHASE_KERNELS.items()):
        kf = next((k for k in fetch if sub in k), None)
        kw = next((k for k in write if sub in k), None)
        if kf is None or kw is None:
            continue
        mb = (2.0 * fetch[kf] + write[kw]) * 1024.0 / 1e6
        phases[ph] = {"kernel": kf, "fetch_size_kib": round(fetch[kf], 1), "write_size_kib": round(write[kw], 1),
                      "hbm_mb_per_launch": round(mb, 1)}
        # matrix-pipe occupancy of the same kernel, when the SQ passes of the profile set exist:
        # SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), and VALU instructions per MFMA
        try:
            busy, cu = parse(prefix + "_pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt"), parse(prefix + "_pmc_SQ_BUSY_CU_CYCLES.txt")
            valu, mfma = parse(prefix + "_pmc_SQ_INSTS_VALU.txt"), parse(prefix + "_pmc_SQ_INSTS_MFMA.txt")
            kb = next((k for k in busy if sub in k), None)
            if kb is not None and cu.get(kb, 0) > 0:
                phases[ph]["matrix_pipe_busy"] = round(busy[kb] / (4.0 * cu[kb]), 4)
            kv = next((k for k in valu if sub in k), None)
            if kv is not None and mfma.get(kv, 0) > 0:
                phases[ph]["valu_per_mfma"] = round(valu[kv] / mfma[kv], 1)
        except OSError:
            pass
    if by_phase is not None:      # SQ counters stay per KERNEL (summaries by name): the matrix-pipe share of the kernel that served the phase
        try:
            busy, cu = parse(prefix + "_pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt"), parse(prefix + "_pmc_SQ_BUSY_CU_CYCLES.txt")
            valu, mfma = parse(prefix + "_pmc_SQ_INSTS_VALU.txt"), parse(prefix + "_pmc_SQ_INSTS_MFMA.txt")
            for ent in phases.values():
                k = ent["kernel"]
                if cu.get(k, 0) > 0 and k in busy:
                    ent["matrix_pipe_busy"] = round(busy[k] / (4.0 * cu[k]), 4)
                if mfma.get(k, 0) > 0 and k in valu:
                    ent["valu_per_mfma"] = round(valu[k] / mfma[k], 1)
        except OSError:
            pass
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_sources_sha
    wkey = sys.argv[3] if len(sys.argv) > 3 else "c2"
    wbatch = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    out = {
        "workload": wkey.split("_")[0], "workload_key": wkey, "batch": wbatch,
        # fingerprint of pmgt_amd/csrc at measurement time: bench.py nulls `traffic` when the kernels have changed since
        "kernel_sources_sha": sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else kernel_sources_sha(),
        "source": f"{prefix}_pmc_fetch_size.txt + {os.path.basename(prefix)}_pmc_write_size.txt "
                  "(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
        "correction": "HBM bytes = 2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE; counters in KiB",
        "step_hbm_gb": round(step_kib * 1024.0 / 1e9, 2),
        "step_hbm_gb_method": step_how,
        "phase_attribution": ("per dispatch: the launch order of the counter run walked next to the engine's phase order (pmgt_profile_sequence); "
                              "hbm_mb_per_launch = median over the phase's launches of a step (the shortcut layer's launches are smaller), "
                              "hbm_mb_by_launch = each of them" if by_phase is not None else "kernel average (one kernel name per phase)"),
        "phases": phases,
    }
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json" if wkey == "c2" else f"traffic_{wkey}.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(phases, indent=1))
    print("HBM traffic per training step: %.2f GB" % out["step_hbm_gb"])


if __name__ == "__main__":
    main()
