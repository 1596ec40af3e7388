"""Data-parallel path with the ENGINE in the loop (reference behaviour: PL DDP, pmgt/base_trainer.py:309-322).

* gradient-ready hook: the buckets the engine reports tile the flat gradient buffer exactly once, arrive in backward order
  and every slice is FINAL when it is reported (stream-ordered copy at hook time == buffer after the step);
* DP equivalence: two processes on cuda:0 over gloo, each with one half of a batch, bucketed exchange from the hook:
  the all-reduced gradient is bit-equal to the mean of the two half-batch gradients computed by ONE process (DDP's
  mean of per-rank means) and equal, up to fp32 summation order, to the single-rank gradient of the whole batch when both
  halves mask the same number of NFR rows; parameters stay bit-identical across ranks over 5 optimizer steps;
* bench.py's N = 2 control flow end to end (2 ranks on this GPU, gloo): one JSON line, both ranks exit 0.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import pmgt_oracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N, S, B = 400, 16, 8
CFG = dict(hidden_size=64, num_hidden_layers=3, num_attention_heads=4, intermediate_size=64, hidden_dropout_prob=0.0,
           attention_probs_dropout_prob=0.0, beta=0.5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_world():
    """Engine (fp32) + one batch of B targets + NFR injections with the SAME number of masked rows in each half."""
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.datasets import MODE_TRAIN, MCNSampler
    from pmgt_amd.engine import Engine
    from pmgt_amd.graph import synthetic_graph
    graph = synthetic_graph(N, 2400, seed=2)
    ocfg = po.default_cfg(**CFG)
    eng = Engine(PMGTConfig(**CFG), dtype="fp32", device="cuda:0", seed=0)
    eng.load_params(po.synth_params(ocfg, 3))
    tables = po.synth_tables(N, ocfg["feat_hidden_sizes"], 4)
    eng.set_tables(*[t.numpy() for t in tables])
    smp = MCNSampler(graph, S - 1)
    tgt, pair, num_pairs, labels = smp.batch(np.arange(2, 2 + B), MODE_TRAIN, threads=2, base_seed=5, counter=0)
    ids = tgt["node_ids"]
    masked = ids.clone()
    targets = torch.full_like(ids, -1)
    for b in range(B):                      # 3 masked (valid) positions per target: equal counts in every shard
        valid = (ids[b, 1:] != 0).nonzero().flatten()[:3] + 1
        assert len(valid) == 3
        targets[b, valid] = ids[b, valid]
        masked[b, valid] = 1
    return eng, (tgt, pair, num_pairs, labels), (masked, targets)


def shard(batch, inj, lo, hi):
    tgt, pair, num_pairs, labels = batch
    off = torch.cat([torch.zeros(1, dtype=torch.int64), num_pairs.cumsum(0)])
    p0, p1 = int(off[lo]), int(off[hi])
    cu = lambda d, a, b: {k: v[a:b].cuda() for k, v in d.items()}
    return ((cu(tgt, lo, hi), cu(pair, p0, p1), num_pairs[lo:hi].cuda(), labels[p0:p1].cuda()),
            (inj[0][lo:hi].cuda(), inj[1][lo:hi].cuda()))


def test_grad_ready_buckets_tile_the_buffer_and_are_final():
    eng, batch, inj = make_world()
    b, i = shard(batch, inj, 0, B)
    seen = []
    snaps = []

    def hook(off, numel):
        seen.append((off, numel))
        snaps.append((off, eng.grads[off: off + numel].clone()))        # stream-ordered: what an all-reduce started here would read

    eng.grads.fill_(float("nan"))
    eng.set_grad_ready_hook(hook)
    eng.pretrain_step(b, training=True, backward=True, nfr_inject=i, want_hidden=False)
    torch.cuda.synchronize()
    eng.set_grad_ready_hook(None)
    L = CFG["num_hidden_layers"]
    assert len(seen) == L + 2
    cover = sorted(seen)
    assert cover[0][0] == 0 and all(a[0] + a[1] == b_[0] for a, b_ in zip(cover, cover[1:])) and cover[-1][0] + cover[-1][1] == eng.n_params
    # backward order: NFR head, layers L-1 .. 0, embeddings
    names = [next(e["name"] for e in eng.entries if e["offset"] == off) for off, _ in seen]
    assert names[0].startswith("nfr_loss.") and names[-1].startswith("bert.embeddings.")
    assert [n.split(".")[3] for n in names[1:-1]] == [str(l) for l in range(L - 1, -1, -1)]
    assert torch.isfinite(eng.grads).all()
    for off, snap in snaps:
        assert torch.equal(snap, eng.grads[off: off + snap.numel()]), off
    # a hook that raises surfaces as a Python exception of the step, not as a crash inside the C callback
    eng.set_grad_ready_hook(lambda off, numel: 1 / 0)
    with pytest.raises(ZeroDivisionError):
        eng.pretrain_step(b, training=True, backward=True, nfr_inject=i, want_hidden=False)
    eng.set_grad_ready_hook(None)
    eng.pretrain_step(b, training=True, backward=True, nfr_inject=i, want_hidden=False)       # and the engine is still usable


def _dp_worker(rank, world, port, out_dir, buckets="layer"):
    import torch.distributed as dist
    from pmgt_amd.trainer import Trainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng, batch, inj = make_world()
        if rank == 1:
            eng.params.mul_(1.5)                 # replicas start different: the broadcast must fix that
        tr = Trainer(eng, lr=1e-3, weight_decay=1e-2, max_grad_norm=5.0, world_size=world, buckets=buckets)
        tr.broadcast_parameters()
        half = B // world
        b, i = shard(batch, inj, rank * half, (rank + 1) * half)
        eng.pretrain_step(b, training=True, backward=True, nfr_inject=i, want_hidden=False)    # the hook exchanges the buckets
        assert tr._exchange.wait() == eng.n_params
        n_coll = {"layer": CFG["num_hidden_layers"] + 2, "two": 2, "one": 1}[buckets]
        assert len(tr._exchange.last_sent) == n_coll, tr._exchange.last_sent
        torch.cuda.synchronize()
        torch.save(eng.grads.cpu(), os.path.join(out_dir, f"g{rank}.pt"))
        # 5 optimizer steps through the Trainer (overlapped exchange + clip + AdamW); dropout stays 0, NFR masks from the device RNG
        for step in range(5):
            tr.train_step(b)
        torch.cuda.synchronize()
        torch.save(eng.params.cpu(), os.path.join(out_dir, f"p{rank}.pt"))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("buckets", ["layer", "two", "one"])
def test_two_rank_gradients_equal_single_rank_and_replicas_stay_identical(tmp_path, buckets):
    import torch.multiprocessing as mp
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path), buckets), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    assert torch.equal(g0, g1)                                              # every rank holds the same reduced gradient
    eng, batch, inj = make_world()
    halves = []
    for r in range(2):
        b, i = shard(batch, inj, r * (B // 2), (r + 1) * (B // 2))
        eng.pretrain_step(b, training=True, backward=True, nfr_inject=i, want_hidden=False)
        torch.cuda.synchronize()
        halves.append(eng.grads.clone())
    mean_of_ranks = ((halves[0] + halves[1]) / 2).cpu()
    assert torch.equal(g0, mean_of_ranks)                                   # DDP: mean of the per-rank gradients, bit for bit
    b, i = shard(batch, inj, 0, B)
    eng.pretrain_step(b, training=True, backward=True, nfr_inject=i, want_hidden=False)
    torch.cuda.synchronize()
    whole = eng.grads.cpu()
    # equal shard sizes and equal masked-row counts: mean of shard means == mean over the whole batch (fp32 summation order aside)
    scale = whole.abs().max().item()
    assert (g0 - whole).abs().max().item() < 2e-5 * scale
    assert torch.nn.functional.cosine_similarity(g0.double(), whole.double(), dim=0).item() > 1 - 1e-9
    p0, p1 = torch.load(tmp_path / "p0.pt"), torch.load(tmp_path / "p1.pt")
    assert torch.equal(p0, p1) and torch.isfinite(p0).all()


@pytest.mark.parametrize("launch", ["driver", "self"])
def test_bench_two_ranks_on_one_gpu_over_gloo(tmp_path, launch):
    """launch = "driver": the exact launch line of the driver at N = 2 (torch.distributed.run, one process per rank);
    launch = "self": `python bench.py --gpus 2` alone, which starts the two ranks itself.  Both ranks on this GPU and gloo
    instead of RCCL: rendezvous, parameter broadcast, sharded staging, bucketed exchange, barriers, MAX over ranks, the
    symmetric phase pass, the all-reduce measurement and the barrier -> destroy teardown."""
    env = dict(os.environ, PMGT_BENCH_BACKEND="gloo", PMGT_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    detail = str(tmp_path / "detail.json")
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64", "--sampler-threads", "2",
            "--detail-out", detail]
    if launch == "driver":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail + ["--buckets", "layer"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 128 and out["scaling"] == "weak"
    assert np.isfinite(out["value"]) and out["value"] > 0 and np.isfinite(out["loss_last"])
    assert "cpu_baseline" not in out and "end_to_end" not in out       # the live pipeline is an N = 1 (or opt-in) pass
    ar = out["allreduce"]
    assert ar["buckets"] == (2 if launch == "driver" else 6) and ar["ms_per_step"] > 0      # default policy "two" (profiles/r05: one-rank RCCL runs); per-bucket on request
    assert abs(ar["mb"] - 4 * 3.06) < 0.5                                                   # the whole flat gradient buffer, once
    # the per-rank sampler rate against what the GPU consumes is in the line (a sampler-bound --end-to-end run is visible, not mysterious)
    assert out["host_sampler"]["vs_gpu_consumption"] > 0 and out["host_sampler"]["threads"] == 2
    full = json.load(open(detail))                                     # the full record: rank 0's, next to the line
    assert "phases" in full and full["value"] == out["value"] and full["allreduce"]["bucket_mb"]


def test_rccl_single_rank_group_runs_the_exchange_unchanged():
    """RCCL executes the exchange step on the hardware the dev loop has: a ONE-rank `nccl` process group (librccl load, communicator
    init, the ReduceOp.AVG probe, asynchronous all_reduce on the process group's stream issued from the engine's C callback, the
    stream-ordered wait in front of the optimizer, barrier, destroy) under Trainer(force_exchange=True) for all three bucket policies and
    the blocking form.  A one-rank average is the identity, so gradients / moments / parameters must equal the no-exchange run bit for
    bit over three optimizer steps.  Own process: the rendezvous environment and the communicator do not leak into the test session."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank.py")], env=env, capture_output=True, text=True,
                       timeout=420)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["backend"] == "nccl"
    assert line["policies"]["layer"]["collectives_per_step"] == 3 + 2          # NFR head, three layers, embeddings
    assert line["policies"]["two"]["collectives_per_step"] == 2 and line["policies"]["one"]["collectives_per_step"] == 1


def test_bench_force_exchange_prints_exactly_one_json_line():
    """`bench.py --force-exchange`: the N > 1 control flow of the bench on a one-rank RCCL group (init, broadcast, bucketed exchange, MAX over ranks,
    the all-reduce measurement, barrier -> destroy).  RCCL prints a version banner to the C-level stdout when its first communicator comes up; the
    bench keeps file descriptor 1 for its ONE JSON line (the driver parses stdout)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-exchange", "--steps", "3", "--warmup", "1", "--batch", "64", "--sampler-threads", "2",
           "--no-cpu-baseline", "--no-end-to-end", "--no-batch-sweep", "--no-extra-workloads"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["exchange_backend"] == "nccl" and out["allreduce"]["buckets"] == 2 and out["allreduce"]["ms_per_step"] > 0
    assert np.isfinite(out["value"]) and out["value"] > 0 and np.isfinite(out["loss_last"])


def test_bench_with_the_drivers_exact_command_prints_one_parsable_line_under_4_kb():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- the command BENCH_rNN.json records, every default side measurement on (phase pass,
    batch sweep, live pipeline, six child workloads, CPU baseline).  Round 5's line was 20 KB and the driver's record lost its head; the line is now
    bounded, starts with the contract's keys, carries `roofline` and `cpu_baseline`, and the full record sits in bench_detail.json."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    detail = os.path.join(ROOT, "bench_detail.json")
    if os.path.exists(detail):
        os.unlink(detail)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=840)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{\"metric\""), r.stdout[-2000:]
    assert len(lines[0].encode()) < 4096, len(lines[0])
    out = json.loads(lines[0])
    assert out["metric"] == "PMGT pre-train nodes/sec" and out["unit"] == "target nodes/s" and out["n_gpus"] == 1
    assert out["steps"] == 20 and out["warmup"] == 5 and out["dtype"] == "bf16" and out["vs_baseline"] is None and out["scaling"] == "weak"
    assert out["config"]["workload"].startswith("c2:") and "B=1024" in out["config"]["workload"]
    assert abs(out["value"] - 1024 / out["ms_per_step"] * 1e3) / out["value"] < 1e-3
    roof, cb = out["roofline"], out["cpu_baseline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["peak"] in (8000.0, 2500.0) and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert "traffic" in roof and cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert 0 < out["mfma_util_encoder"] < 1 and out["end_to_end"]["vs_prestaged"] > 0
    for name, w in out["workloads"].items():
        assert "skipped" in w or (w["nodes_per_s"] > 0 and w["roofline_kernel"]), (name, w)
    full = json.load(open(detail))
    assert full["value"] == out["value"] and "phases" in full and "roofline_top3" in full and "batch_sweep" in full
