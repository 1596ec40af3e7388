import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from tests import golden_util as gu
from tests.test_engine_gpu import make_engine, dev_batch, inject_for
from pmgt_amd import _lib
L = _lib.hip()
case = gu.model_case("m3")
batch = dev_batch(case["batch"]); inj, _ = inject_for(case)
def grads(dtype, off):
    L.pmgt_debug_disable_segment_sum(off)
    eng = make_engine(case, dtype=dtype)
    eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj)
    L.pmgt_debug_disable_segment_sum(0)
    return eng, eng.grads.clone()
eng, g32 = grads("fp32", 1)
_, gb_tok = grads("bf16", 1)
_, gb_node = grads("bf16", 0)
for k in ("bert.embeddings.feat_linear.0.weight", "bert.embeddings.feat_linear.1.weight", "bert.embeddings.attention.1.weight"):
    e = eng.entry(k); sl = slice(e["offset"], e["offset"] + e["numel"])
    r = lambda a, b: ((a - b).norm() / b.norm()).item()
    print(k, "per-token bf16 vs fp32: %.4f | per-node bf16 vs fp32: %.4f | mutual %.4f" % (r(gb_tok[sl], g32[sl]), r(gb_node[sl], g32[sl]), r(gb_node[sl], gb_tok[sl])))
print("all: tok %.4f node %.4f" % (((gb_tok - g32).norm() / g32.norm()).item(), ((gb_node - g32).norm() / g32.norm()).item()))
