"""bf16 bench loop printing the loss of every step (NaN hunt)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, ".")
from pmgt_amd.configuration_pmgt import PMGTConfig
from pmgt_amd.datasets import MODE_TRAIN, MCNSampler
from pmgt_amd.engine import Engine
from pmgt_amd.graph import synthetic_graph
from pmgt_amd.models import reference_init, synthetic_features
from pmgt_amd.trainer import Trainer
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
nodes, edges, L, H, d, I, S = 7252, 88606, 4, 8, 256, 256, 32
B = 1024
cfg = PMGTConfig(hidden_size=d, num_hidden_layers=L, num_attention_heads=H, intermediate_size=I, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, beta=0.5)
graph = synthetic_graph(nodes, edges, seed=0)
vis, txt = synthetic_features(nodes, seed=0)
eng = Engine(cfg, dtype=dtype, device="cuda:0", seed=1234)
reference_init(eng, seed=0)
eng.set_tables(vis, txt)
tr = Trainer(eng, lr=1e-4, weight_decay=1e-2, max_grad_norm=5.0)
sampler = MCNSampler(graph, max_ctx_neigh=S - 1)
perm = np.random.RandomState(0).permutation(nodes) + 2
staged = []
for i in range(8):
    tg = np.resize(perm[(i * B) % max(len(perm) - B, 1):], B)
    tgt, pair, num_pairs, labels = sampler.batch(tg, MODE_TRAIN, threads=16, base_seed=0, counter=i * B)
    cu = lambda dct: {k: v.to("cuda:0") for k, v in dct.items()}
    staged.append((cu(tgt), cu(pair), num_pairs.to("cuda:0"), labels.to("cuda:0")))
for i in range(nsteps):
    out = eng.pretrain_step(staged[i % 8], training=True, backward=True, want_hidden=False)
    gbad = int((~torch.isfinite(eng.grads)).sum())
    tr.last_loss = out["loss"]; tr.optimizer_step()
    pbad = int((~torch.isfinite(eng.params)).sum())
    print(i, [round(x, 5) for x in out["losses"].tolist()], "gnorm", round(eng.grad_norm().item(), 4), "bad grads", gbad, "bad params", pbad, flush=True)
    if i == 0 or gbad:
        for e in eng.entries:
            g = eng.view(e["name"], grad=True)
            if g.abs().max().item() > 1e3 or not torch.isfinite(g).all(): print("   HUGE", e["name"], g.abs().max().item(), int((g.abs() > 1e3).sum()), "of", g.numel(), (g.abs() > 1e3).nonzero()[:5].tolist())
    if gbad:
        for e in eng.entries:
            g = eng.view(e["name"], grad=True)
            nb = int((~torch.isfinite(g)).sum())
            if nb: print("   ", e["name"], nb, "of", g.numel())
        break
