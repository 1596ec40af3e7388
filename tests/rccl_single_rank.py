"""Child process of tests/test_dp_gpu.py::test_rccl_single_rank_group_runs_the_exchange_unchanged: a ONE-rank `nccl` (= RCCL on ROCm)
process group on cuda:0 and the N > 1 code path of the trainer unchanged on it (reference: PL DDP, pmgt/base_trainer.py:309-322).
Prints one JSON line; every check that fails raises."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from pmgt_amd import parallel
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.datasets import MODE_TRAIN, MCNSampler
    from pmgt_amd.engine import Engine
    from pmgt_amd.graph import synthetic_graph
    from pmgt_amd.models import reference_init, synthetic_features
    from pmgt_amd.trainer import Trainer
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    avg = parallel.backend_averages(torch.device("cuda:0"))          # the probe collective: the first thing RCCL runs
    n, S, B, steps = 600, 32, 24, 3
    graph = synthetic_graph(n, 5000, seed=3)
    cfg = dict(hidden_size=256, num_hidden_layers=3, num_attention_heads=8, intermediate_size=256, hidden_dropout_prob=0.1,
               attention_probs_dropout_prob=0.1)           # the headline kernels (fused projection + attention, fused backward) are on the path
    feats = synthetic_features(n, seed=3)
    smp = MCNSampler(graph, S - 1)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}
    batches = []
    for i in range(steps):
        tgt, pair, num_pairs, labels = smp.batch(np.arange(2 + i * B, 2 + (i + 1) * B), MODE_TRAIN, threads=2, base_seed=9, counter=i * B)
        batches.append((cu(tgt), cu(pair), num_pairs.cuda(), labels.cuda()))

    def run(**kw):
        eng = Engine(PMGTConfig(**cfg), dtype="bf16", device="cuda:0", seed=11)
        reference_init(eng, 0)
        eng.set_tables(*feats)
        tr = Trainer(eng, lr=1e-3, max_grad_norm=5.0, **kw)
        tr.broadcast_parameters()
        sent = []
        for b in batches:
            tr.train_step(b)
            if tr._exchange is not None:
                sent.append(list(tr._exchange.last_sent))
        torch.cuda.synchronize()
        return eng, tr, sent

    base, _, _ = run()
    res = {"backend": dist.get_backend(), "reduce_op_avg": bool(avg), "policies": {}}
    for policy in ("layer", "two", "one"):
        eng, tr, sent = run(force_exchange=True, buckets=policy)
        assert tr._exchange is not None
        for s in sent:          # every step's collectives tile the flat gradient buffer exactly once
            cover = sorted(s)
            assert cover[0][0] == 0 and sum(nn for _, nn in cover) == eng.n_params, (policy, cover)
            assert all(cover[i][0] + cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1)), (policy, cover)
        # a one-rank average is the identity: gradients, moments and parameters equal the no-exchange run BIT FOR BIT
        assert torch.equal(eng.grads, base.grads), policy
        assert torch.equal(eng.params, base.params) and torch.equal(eng.exp_avg_sq, base.exp_avg_sq), policy
        res["policies"][policy] = {"collectives_per_step": len(sent[-1]), "mb": round(eng.n_params * 4 / 1e6, 3)}
    # the blocking form (overlap_allreduce = False): one all-reduce after the backward pass
    eng, tr, _ = run(force_exchange=True, overlap_allreduce=False)
    assert tr._exchange is None and torch.equal(eng.params, base.params)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
