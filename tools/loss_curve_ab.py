"""160 optimisation steps at B = 256 on the C2 graph, fast path against the plain path (fused forward / backward kernels, LayerNorm in the GEMM
epilogues (forward and backward), LayerNorm backward from the output, recomputed embedding sum / per-node segment sums, last-layer shortcut and
deferred reductions switched off).  PMGT_AB_SHAPE=c4: hidden 512, S = 64, 3 layers, B = 64 (the d = 512 kernels, full-row LayerNorm GEMM included): the two loss curves agree to 1e-5 for the first ~40 steps and
then drift apart as two bf16 trajectories do; mean of the last 20 steps 1.52357 vs 1.52357 (round 2, final tree)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from pmgt_amd import _lib
from pmgt_amd.configuration_pmgt import PMGTConfig
from pmgt_amd.datasets import MODE_TRAIN, MCNSampler
from pmgt_amd.graph import synthetic_graph
from pmgt_amd.models import synthetic_features, reference_init
from pmgt_amd.engine import Engine
graph = synthetic_graph(7252, 88606, seed=0)
vis, txt = synthetic_features(7252, seed=0)
C4 = os.environ.get("PMGT_AB_SHAPE", "c2") == "c4"
if C4:
    cfg = PMGTConfig(hidden_size=512, num_hidden_layers=3, num_attention_heads=8, intermediate_size=512, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, beta=0.5)
else:
    cfg = PMGTConfig(hidden_size=256, num_hidden_layers=4, num_attention_heads=8, intermediate_size=256, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, beta=0.5)
smp = MCNSampler(graph, max_ctx_neigh=63 if C4 else 31)
B = 64 if C4 else 256
batches = []
for c in range(8):
    tgt, pair, num_pairs, labels = smp.batch(np.arange(2 + c * B, 2 + (c + 1) * B), MODE_TRAIN, threads=4, base_seed=3, counter=c)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}
    batches.append((cu(tgt), cu(pair), num_pairs.cuda(), labels.cuda()))
def run(plain):
    eng = Engine(cfg, dtype=os.environ.get("PMGT_AB_DTYPE", "bf16"), seed=5)
    for k in ("no_fused_qkvc_attention", "no_fused_attention_bwd", "store_ln_input", "no_shortcut", "eager_reduce", "unfused_ln", "unfused_ln_bwd", "no_segment_sum"):
        eng.set_option(k, 1 if plain else 0)
    reference_init(eng, seed=0)
    eng.set_tables(vis, txt)
    losses = []
    for s in range(160):
        out = eng.pretrain_step(batches[s % 8], training=True, backward=True)
        eng.optimizer_step(lr=3e-4, weight_decay=1e-2, max_grad_norm=5.0)
        losses.append(out["loss"].item())
    return np.array(losses)
a = run(False); b = run(True)
for k in (0, 1, 10, 40, 80, 120, 159): print(k, "fast %.5f plain %.5f" % (a[k], b[k]))
print("mean last 20: fast %.5f plain %.5f; max |diff| over the run %.4f" % (a[-20:].mean(), b[-20:].mean(), np.abs(a - b).max()))
