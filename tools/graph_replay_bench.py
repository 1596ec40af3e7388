"""hipGraph replay vs eager launches of one training step (A/B tool, not a test): python tools/graph_replay_bench.py [B]."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmgt_amd.configuration_pmgt import PMGTConfig
from pmgt_amd.datasets import MODE_TRAIN, MCNSampler
from pmgt_amd.engine import Engine
from pmgt_amd.graph import synthetic_graph
from pmgt_amd.models import reference_init, synthetic_features
from pmgt_amd.trainer import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = PMGTConfig(hidden_size=256, num_hidden_layers=4, num_attention_heads=8, intermediate_size=256, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, beta=0.5)
g = synthetic_graph(7252, 88606, seed=0)
vis, txt = synthetic_features(7252, seed=0)
eng = Engine(cfg, dtype="bf16", device="cuda:0", seed=1)
reference_init(eng, seed=0); eng.set_tables(vis, txt)
tr = Trainer(eng, lr=1e-4, weight_decay=1e-2, max_grad_norm=5.0)
s = MCNSampler(g, max_ctx_neigh=31)
tgt, pair, npairs, labels = s.batch(np.arange(2, 2 + B), MODE_TRAIN, threads=8, base_seed=0, counter=0)
cu = lambda d: {k: v.cuda() for k, v in d.items()}
batch = (cu(tgt), cu(pair), npairs.cuda(), labels.cuda())
for _ in range(5): tr.train_step(batch)
torch.cuda.synchronize()
def timeit(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager ms/step", round(timeit(lambda: tr.train_step(batch)), 3))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3): tr.train_step(batch)
    st.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        loss = tr.train_step(batch)
torch.cuda.synchronize()
l0 = []
for _ in range(5):
    gr.replay(); torch.cuda.synchronize(); l0.append(loss.item())
print("graph losses", l0)
print("graph ms/step", round(timeit(lambda: gr.replay()), 3))
