"""Cycles per step inside gemm_wsr_kernel (K = N = 256, residual + LayerNorm; library built with PMGT_EXTRA_HIP_FLAGS=-DPMGT_W5_PROF)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _lib
L = _lib.ops()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
M, N, K = 393216, 256, 256
NS = 3
As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(NS)]
W = torch.randn(N, K, device="cuda").bfloat16()
Cs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
Rs = [torch.randn(M, N, device="cuda").bfloat16() for _ in range(NS)]
lnos = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
stats = torch.empty(M, 2, device="cuda")
g = torch.ones(N, device="cuda")
bias = torch.zeros(N, device="cuda")
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(i):
    _lib.check(L.pmgt_op_linear(1, P(As[i]), K, P(W), K, P(Cs[i]), N, M, N, K, P(bias), 0, None, N, P(Rs[i]), N, 0.1, 5, P(rng),
                                P(lnos[i]), P(stats), P(g), P(g), 1e-12, st))
for k in range(6): run(k % NS)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for k in range(9): run(k % NS)
b.record(); torch.cuda.synchronize()
print("us per launch", a.elapsed_time(b) / 9 * 1e3)
out = (C.c_ulonglong * 128)()
_lib.hip().pmgt_debug_w2_prof_read(out)
v = np.array(list(out), dtype=np.float64).reshape(16, 8)
for w in (0, 4, 7):
    n = max(v[w, 7], 1)
    print("GEMM wave", w, "steps", int(n), " ".join(f"{lbl}={v[w, k] / n:.0f}" for k, lbl in enumerate(("dma", "frags+mfma", "staging+waits", "barrier"))), "per step", f"{v[w, :4].sum() / n:.0f}")
for w in (8, 12, 15):
    n = max(v[w, 7], 1)
    print("epilogue wave", w, " ".join(f"{lbl}={v[w, k] / n:.0f}" for k, lbl in enumerate(("work", "barrier"))))
