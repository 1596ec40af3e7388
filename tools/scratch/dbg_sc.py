import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from tests import golden_util as gu
from tests.test_engine_gpu import make_engine, dev_batch, inject_for
case = gu.model_case("m3")
batch = dev_batch(case["batch"]); inj, _ = inject_for(case)
res = {}
for label, wh in (("full", True), ("fast", False)):
    eng = make_engine(case, dtype="bf16")
    out = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj, want_hidden=wh)
    res[label] = (out["loss"].item(), eng.grads.clone())
print("loss", res["full"][0], res["fast"][0])
for e in eng.entries:
    sl = slice(e["offset"], e["offset"] + e["numel"])
    a, b = res["fast"][1][sl], res["full"][1][sl]
    rel = ((a - b).norm() / (b.norm() + 1e-20)).item()
    if rel > 0.05 or not np.isfinite(rel): print("%-60s rel %.3g  |fast| %.3g |full| %.3g" % (e["name"], rel, a.norm().item(), b.norm().item()))
