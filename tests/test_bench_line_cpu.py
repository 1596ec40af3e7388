"""bench.py's stdout contract (no GPU): ONE JSON line that a bounded reader can keep.

Round 5's line grew to 20 KB and the driver's record lost its head (the headline keys).  `bench.compact_line` builds the line from the
full record; these tests feed it records far larger than any run produces and hold it to the budget, the contract's keys and a JSON round trip.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config")


def fat_record(n_children=6, n_phases=40, prose=400):
    roof = {"kernel": "bwd.attention_wgrad", "bound": "hbm", "achieved": 3640.13, "peak": 8000.0, "unit": "GB/s", "frac": 0.45502,
            "traffic": 2049.1, "traffic_ratio": 1.018, "traffic_src": "profiles/traffic.json", "traffic_unit": "x" * prose,
            "avg_launch_ms": 0.55308, "algorithmic_mb_per_launch": 2013.266, "matrix_pipe_busy": 0.2513, "valu_per_mfma": 7.7}
    phases = {f"bwd.phase_{i}": {"launches_per_step": 4, "ms_per_step": 1.2345, "share": 0.1234} for i in range(n_phases)}
    top3 = [dict(roof, launches_per_step=4) for _ in range(3)]
    child = {"nodes_per_s": 11604.59, "ms_per_step": 22.0602, "batch": 256, "steps": 10, "train_gflop_per_node": 71.4,
             "mfma_util_vs_bf16_dense_peak": 0.26, "mfma_util_encoder": 0.248, "mfma_util_executed": 0.249, "roofline": roof,
             "roofline_top3": top3, "sustained_sclk_mhz": 2100.0, "loss_first": 1.9, "loss_last": 1.8, "setup_s": 3.4, "wall_s": 5.0,
             "phases_top": dict(list(phases.items())[:8])}
    rec = {
        "metric": "PMGT pre-train nodes/sec", "value": 112624.73, "unit": "target nodes/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 9.0921, "sustained_sclk_mhz": 2139.1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic (seeded G(n,m)+ring item graph, N(0,1) visual/textual features, random-init weights)",
        "config": {"workload": "c2: 7252 nodes / 88606 edges, L=4 H=8 d=256 I=256 S=32 (context=S tokens incl. target), B=1024 targets/GPU/step, "
                               "12 sequences/target, dropout 0.1, beta 0.5, clip 5.0, AdamW lr 1e-4 wd 1e-2",
                   "parallelism": "dp1", "global_batch": 1024, "seq_len": 32},
        "loss_first": 1.95726, "loss_last": 1.74765, "grad_norm_last": 0.17534, "setup_s": 0.9, "side_stream_reductions": False,
        "engine_options": [], "host_sampler": {"nodes_per_s": 170333.1, "threads": 13, "cpus": 256, "cpu_share": 16.0},
        "allreduce": None, "train_gflop_per_node": 5.38, "mfma_util_vs_bf16_dense_peak": 0.24238, "encoder_gflop_per_node": 4.45,
        "mfma_util_encoder": 0.20077, "executed_gflop_per_node": 4.06, "mfma_util_executed": 0.18276,
        "phases": phases, "roofline": roof, "roofline_top3": top3, "step_hbm_gb": 37.78,
        "batch_sweep": {str(b): {"nodes_per_s": 25586.6, "ms_per_step": 1.2507, "ms_per_step_passes": [1.25, 1.26, 1.25],
                                 "hipgraph_replay": {"nodes_per_s": 25441.7, "ms_per_step": 1.2578}} for b in (32, 256, 2048, 4096)},
        "end_to_end": {"nodes_per_s": 109794.9, "ms_per_step": 9.326, "vs_prestaged": 0.9749, "gpu_idle_ms_per_step": 0.012, "note": "y" * prose},
        "workloads": {f"child_{i}": dict(child) for i in range(n_children)},
        "cpu_baseline": {"value": 56.06, "unit": "target nodes/s", "cores": 16, "kind": "port",
                         "sample": "11 timed steps of B=8 targets (96 sequences x S=32) after 1 warm-up, fp32 torch CPU oracle, fwd+bwd+clip+AdamW, 143 ms/step"},
    }
    return rec


def check_line(line, want_roofline=True):
    assert "\n" not in line
    assert len(line.encode()) < bench.LINE_BUDGET, len(line)
    d = json.loads(line)
    for k in CONTRACT_KEYS:
        assert k in d, k
    assert "workload" in d["config"]
    if want_roofline:
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, k
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        cb = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in cb, k
    # the head of the line is the contract: a reader that keeps only the first kilobyte still has the headline
    head = line[:1024]
    for k in ("\"metric\"", "\"value\"", "\"unit\"", "\"n_gpus\"", "\"ms_per_step\"", "\"dtype\""):
        assert k in head, k
    return d


def test_compact_line_of_a_round5_sized_record_fits_and_keeps_the_contract():
    rec = fat_record()
    assert len(json.dumps(rec)) > 15000           # the record itself is the size that broke round 5
    d = check_line(bench.compact_line(rec, "/somewhere/bench_detail.json"))
    assert d["detail"] == "bench_detail.json"
    assert d["value"] == rec["value"] and d["ms_per_step"] == rec["ms_per_step"]
    assert "phases" not in d and "traffic_unit" not in d["roofline"]
    # the side measurements survive as short summaries
    assert set(d["workloads"]) == set(rec["workloads"])
    assert d["workloads"]["child_0"]["roofline_frac"] == 0.45502 and d["workloads"]["child_0"]["traffic"] == 2049.1
    assert d["end_to_end"]["vs_prestaged"] == 0.9749
    assert d["mfma_util_encoder"] == 0.20077


@pytest.mark.parametrize("n_children,n_phases,prose", [(40, 400, 5000), (200, 10, 100), (0, 0, 0)])
def test_compact_line_gives_up_side_measurements_before_the_budget(n_children, n_phases, prose):
    rec = fat_record(n_children, n_phases, prose)
    rec["workloads"]["broken"] = {"skipped": "child exited with 1: " + "trace " * 2000}
    rec["batch_sweep"]["8192"] = {"error": "RuntimeError(" + "z" * 3000 + ")"}
    check_line(bench.compact_line(rec, "bench_detail.json"))


def test_compact_line_of_a_multi_rank_record():
    rec = fat_record()
    rec.update({"n_gpus": 8, "cpu_baseline": None, "workloads": None, "batch_sweep": None, "exchange_backend": "nccl",
                "allreduce": {"buckets": 2, "mb": 12.2, "ms_per_step": 0.21, "bucket_mb": [9.8, 2.4], "policy": "two",
                              "overlapped_with_backward": True, "note": "n" * 300}})
    rec.pop("cpu_baseline")
    d = check_line(bench.compact_line(rec, None), want_roofline=False)
    assert d["n_gpus"] == 8 and d["allreduce"]["policy"] == "two" and "note" not in d["allreduce"]
    assert "detail" not in d


def test_write_detail_round_trips(tmp_path):
    rec = fat_record()
    p = bench.write_detail(rec, str(tmp_path / "d.json"))
    assert json.load(open(p)) == rec
    assert bench.write_detail(rec, str(tmp_path / "no_such_dir" / "d.json")) is None      # never a reason to lose the line


def test_traffic_key_names_the_committed_summaries():
    assert bench.traffic_key("c2", 0, 0.5) == "c2"
    assert bench.traffic_key("c2", 1024, 0.5) == "c2_i1024"
    assert bench.traffic_key("c2", 0, 1.0) == "c2_beta1"
    assert bench.traffic_key("c4", 2048, 0.5) == "c4_i2048"
    assert bench.traffic_key("c4", 0, 0.5, 256) == "c4" and bench.traffic_key("c4", 0, 0.5, 1024) == "c4_b1024"
    assert bench.traffic_key("c2", 0, 0.5, 1024) == "c2" and bench.traffic_key("c2", 0, 0.5, 256) == "c2_b256"


def test_rehearsal_line_at_world_8_is_one_short_line():
    """The launcher path at the driver's largest rank count, no GPU: 8 ranks rendezvous over gloo and rank 0 prints ONE line."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rehearse-launch"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["rank_sum"] == 28.0 and len(lines[0]) < bench.LINE_BUDGET


def test_committed_traffic_files_were_measured_on_these_kernels():
    """bench.py quotes `roofline.traffic` only while profiles/traffic*.json carry the fingerprint of pmgt_amd/csrc: the committed files are the ones
    of THIS tree (a kernel edit without a new counter run would silently turn every `traffic` in the line into null), they are attributed per phase
    launch, and there is one for every configuration the default line quotes."""
    sha = bench.kernel_sources_sha()
    want = {"c2": ("traffic.json", 1024), "c2_i1024": ("traffic_c2_i1024.json", 1024), "c2_beta1": ("traffic_c2_beta1.json", 1024),
            "c4": ("traffic_c4.json", 256), "c4_i2048": ("traffic_c4_i2048.json", 256), "c4_b1024": ("traffic_c4_b1024.json", 1024)}
    for key, (name, batch) in want.items():
        tr = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert tr["kernel_sources_sha"] == sha, (name, "re-run tools/gpu_collect.sh + tools/publish_profiles.sh on this tree")
        assert tr.get("workload_key", tr["workload"]) == key and tr["batch"] == batch
        assert "per dispatch" in tr["phase_attribution"] and tr["step_hbm_gb"] > 0
        dom = max(tr["phases"].values(), key=lambda e: e["hbm_mb_per_launch"] * e["launches_per_step"])
        assert len(dom["hbm_mb_by_launch"]) == dom["launches_per_step"] and dom["hbm_mb_per_launch"] > 100
    # every child of the default line has a file under the key bench.py derives for it
    for name, extra in bench.EXTRA_WORKLOADS:
        if "fp8" in extra:
            continue
        arg = lambda flag, dflt: extra[extra.index(flag) + 1] if flag in extra else dflt
        key = bench.traffic_key(arg("--workload", "c2"), int(arg("--intermediate", 0)), float(arg("--beta", 0.5)), int(arg("--batch", 1024)))
        assert key in want, (name, key)
