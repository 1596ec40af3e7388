"""Cycles per step inside gemm_wsr512_kernel (library built with PMGT_EXTRA_HIP_FLAGS=-DPMGT_W5_PROF).  argv: mode (0 plain, 1 gelu, 2 gelu', 3 res) M N"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _lib
L = _lib.ops()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M, N = (int(v) for v in sys.argv[2:4]) if len(sys.argv) >= 4 else (196608, 2048)
K = 512
A = torch.randn(M, K, device="cuda").bfloat16()
W = torch.randn(N, K, device="cuda").bfloat16()
Cd = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
R = torch.randn(M, N, device="cuda").bfloat16()
X = torch.randn(M, N, device="cuda").bfloat16()
bias = torch.zeros(N, device="cuda")
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    epi = {0: 0, 1: 1, 2: 2, 3: 0}[mode]
    _lib.check(L.pmgt_op_linear(1, P(A), K, P(W), K, P(Cd), N, M, N, K, P(bias), epi, P(X) if mode in (1, 2) else None, N, P(R) if mode == 3 else None, N,
                                0.1 if mode == 3 else 0.0, 5, P(rng), None, None, None, None, 1e-12, st))
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): run()
b.record(); torch.cuda.synchronize()
print("mode", mode, "us per launch", a.elapsed_time(b) / 10 * 1e3)
if hasattr(_lib.hip(), "pmgt_debug_w5_prof_read"):
    out = (C.c_ulonglong * 96)()
    _lib.hip().pmgt_debug_w5_prof_read(out)
    v = np.array(list(out), dtype=np.float64).reshape(12, 8)
    for w in (0, 4, 7):
        n = max(v[w, 7], 1)
        print("GEMM wave", w, "steps", int(n), " ".join(f"{lbl}={v[w, k] / n:.0f}" for k, lbl in enumerate(("dma", "frags+mfma", "staging+waits", "barrier"))), "per step", f"{v[w, :4].sum() / n:.0f}")
    for w in (8, 11):
        n = max(v[w, 7], 1)
        print("epilogue wave", w, " ".join(f"{lbl}={v[w, k] / n:.0f}" for k, lbl in enumerate(("work", "barrier"))))
