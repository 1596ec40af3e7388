"""world_size-2 gloo tests of the data-parallel plumbing (CPU): the gradient exchange averages the
flat buffer, parameters are broadcast from rank 0, shards follow DistributedSampler semantics, and
the AUC helper matches sklearn."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pmgt_amd.parallel import BucketedAllReduce, allreduce_mean_, broadcast_, gather_predictions, shard_indices
from pmgt_amd.trainer import roc_auc_score


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # parameters: rank 0's values win
        params = torch.full((1000,), float(rank + 1))
        broadcast_(params)
        assert torch.all(params == 1.0)
        # gradients: average over ranks; identical result on every rank
        g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        allreduce_mean_(g)
        assert torch.allclose(g, torch.arange(1000, dtype=torch.float32) * 1.5)
        # a "training loop" on a quadratic: both replicas stay bit-identical
        w = params.clone()
        for step in range(5):
            shard = shard_indices(101, rank, world, seed=3, epoch=step)
            grad = w * 0 + float(shard.sum() % 7)       # rank-dependent local gradient
            allreduce_mean_(grad)
            w -= 0.1 * grad
        torch.save(w, os.path.join(out_dir, f"w{rank}.pt"))
        # the overlapped form: buckets reported in backward order (head, layers L-1..0, embeddings) tile the flat buffer;
        # after wait() it holds exactly what ONE flat all-reduce gives, on every rank
        flat = torch.arange(1000, dtype=torch.float32).mul_(0.37 * (rank + 1)).sin_()
        want = allreduce_mean_(flat.clone())
        ex = BucketedAllReduce(flat)
        bounds = [0, 130, 400, 670, 900, 1000]
        for lo, hi in reversed(list(zip(bounds[:-1], bounds[1:]))):
            ex.bucket_ready(lo, hi - lo)
        assert ex.wait() == 1000 and torch.equal(flat, want)
        ex.enabled = False                       # micro-batches before the last one of an accumulation window: no exchange
        ex.bucket_ready(0, 1000)
        assert ex.wait() == 0
        # validation gather: every rank sees all shards in rank order
        p, l = gather_predictions(np.full(3 + rank, 0.25 * (rank + 1), np.float32), np.full(3 + rank, float(rank), np.float32))
        assert p.tolist() == [0.25] * 3 + [0.5] * 4 and l.tolist() == [0.0] * 3 + [1.0] * 4
    finally:
        dist.destroy_process_group()


def test_gloo_world2_allreduce_broadcast(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = torch.load(tmp_path / "w0.pt"), torch.load(tmp_path / "w1.pt")
    assert torch.equal(w0, w1)


def test_shard_indices_match_distributed_sampler():
    from torch.utils.data import DistributedSampler
    ds = list(range(103))
    for world in (1, 2, 4, 8):
        for epoch in (0, 3):
            seen = []
            for rank in range(world):
                ref = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=5)
                ref.set_epoch(epoch)
                mine = shard_indices(len(ds), rank, world, seed=5, epoch=epoch)
                assert list(ref) == mine.tolist()
                seen.extend(mine.tolist())
            assert set(seen) == set(range(103))          # every node is visited each epoch


def test_auc_matches_sklearn():
    from sklearn.metrics import roc_auc_score as sk
    rs = np.random.RandomState(0)
    for _ in range(5):
        y = rs.randint(0, 2, 500)
        s = np.round(rs.rand(500), 2)       # plenty of ties
        assert abs(roc_auc_score(y, s) - sk(y, s)) < 1e-12
    with pytest.raises(ValueError):
        roc_auc_score(np.ones(5), rs.rand(5))


def test_bench_control_flow_is_symmetric_across_ranks():
    """bench.py at N > 1: every rank must execute the same sequence of GPU steps and collectives from init to destroy
    (round 1 let rank 0 alone run a profile pass while its peers destroyed the process group: a hang / abort on RCCL).
    Statically: nothing that launches steps or collectives sits under a rank-dependent condition, the teardown is
    barrier -> destroy, and the staging shard comes from shard_indices (DistributedSampler semantics)."""
    import ast
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    collective_or_step = ("train_step", "run_live", "profile_begin", "profile_end", "all_reduce", "barrier(", "broadcast",
                          "time_steps", "capture_step", "destroy_process_group")
    for node in ast.walk(main):
        if isinstance(node, ast.If) and "rank" in ast.unparse(node.test) and "world" not in ast.unparse(node.test):
            body = "\n".join(ast.unparse(b) for b in node.body + node.orelse)
            assert not any(k in body for k in collective_or_step), ast.unparse(node)[:300]
    tail = ast.unparse(main.body[-1])
    assert tail.index("dist.barrier()") < tail.index("dist.destroy_process_group()"), tail
    assert "shard_indices(nodes, rank, world" in src and "RandomState(0).permutation" not in src
    # the rank-0-only blocks are the print, the CPU baseline and the extra single-GPU workloads (child processes; N = 1 only)
    r0 = [ast.unparse(n) for n in ast.walk(main) if isinstance(n, ast.If) and ast.unparse(n.test).startswith("rank == 0")]
    assert all(("print(" in b or "cpu_baseline" in b or "extra_workloads" in b) for b in r0), r0


def test_bench_gpus_flag_starts_its_own_ranks_and_refuses_a_mismatch():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (torch.distributed.run as a child, before
    any GPU call) and reports n_gpus = 2; under a launcher whose WORLD_SIZE differs from --gpus it refuses to print a line.
    --rehearse-launch keeps the engine out of it (no GPU here): rendezvous, one collective, symmetric teardown."""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--rehearse-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 1.0
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--rehearse-launch"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 8" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bucket_policy_coalesces_descending_ranges_at_the_boundaries():
    """The engine reports gradient ranges in descending offset order (NFR head, layers L-1 .. 0, embeddings); with a boundary at
    layer 0's offset the exchange issues TWO collectives (head + layers | embeddings), with none one per range."""
    from pmgt_amd.parallel import BucketedAllReduce
    flat = torch.zeros(100)
    ranges = [(90, 10), (70, 20), (50, 20), (30, 20), (0, 30)]       # head, layers 2..0, embeddings

    def run(bounds):
        ex = BucketedAllReduce(flat, boundaries=bounds)
        sent = []
        ex._send = lambda off, n: sent.append((off, n))               # no process group here: record instead of all-reducing
        for off, n in ranges:
            ex.bucket_ready(off, n)
        assert ex._held is None
        return sent
    assert run(()) == ranges
    assert run((30,)) == [(30, 70), (0, 30)]
    assert run((70, 30)) == [(70, 30), (30, 40), (0, 30)]
    # a range that is not adjacent to what is held flushes it first (pmgt_encode_backward: no NFR head bucket)
    ex = BucketedAllReduce(flat, boundaries=(30,))
    sent = []
    ex._send = lambda off, n: sent.append((off, n))
    for off, n in [(70, 20), (30, 20), (0, 30)]:
        ex.bucket_ready(off, n)
    assert sent == [(70, 20), (30, 20), (0, 30)]


def test_bench_line_helpers():
    """Pure-host pieces of the bench line: the sustained shader clock from two probe samples (cycle counter against the 100 MHz wall counter, paired
    per XCC: the counters of different XCCs need not share an origin), the encoder-only flop count SURVEY 8(d) quotes, and the algorithmic bytes
    of the fused attention kernels in the beta = 1 form."""
    import torch
    import bench
    from pmgt_amd.models import encoder_flops_per_node, train_flops_per_node
    before = torch.tensor([[1_000, 50, 0, 1], [9_000_000, 50, 1, 1], [77, 60, 0, 1]], dtype=torch.int64)
    after = torch.tensor([[9_000_000 + 2_100_000, 1_050, 1, 1], [1_000 + 2_100_000, 1_050, 0, 1]], dtype=torch.int64)
    # 2.1e6 cycles over 1 000 ticks of the 100 MHz counter (10 us): cycles / (ticks / 100) us = MHz (a deliberately unphysical number)
    assert bench.sustained_sclk_mhz(before, after) == round(2_100_000 / (1_000 / 100.0), 1)
    assert bench.sustained_sclk_mhz(before, before) is None                       # no elapsed wall time: no number
    assert abs(encoder_flops_per_node(256, 256, 4, 32) / 1e9 - 4.457) < 1e-3 and abs(train_flops_per_node(256, 256, 4, 32) / 1e9 - 5.380) < 1e-3
    assert abs(train_flops_per_node(256, 1024, 4, 32) / 1e9 - 9.004) < 1e-3 and abs(train_flops_per_node(512, 2048, 6, 64) / 1e9 - 100.64) < 1e-2
    M, d = 393216, 256
    gen = bench.phase_work("bwd.attention_wgrad", M, d, 256, 32, 8, 1536, 768, 1024 * 31, 2)
    vc = bench.phase_work("bwd.attention_wgrad", M, d, 256, 32, 8, 1536, 768, 1024 * 31, 2, vc=True)
    assert gen[1] == M * 10 * d * 2 and vc[1] == M * 6 * d * 2 and vc[0] < gen[0]
