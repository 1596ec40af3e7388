"""Role ablation of gemm_wsr_kernel<false> (K = N = 256 GEMM + residual + LayerNorm epilogue): builds of the library with one role's work
compiled out (-DPMGT_WSR_NO_EPI: the epilogue role only keeps the barriers; -DPMGT_WSR_NO_MFMA: the GEMM role only moves the A tiles), timed
in one process.  Usage: python tools/prof/wsr_ablate.py lib1.so lib2.so ... [rounds=N]   (M = 393 216 rows, dropout 0.1)"""
import ctypes as C, sys, os
import numpy as np, torch
paths = [a for a in sys.argv[1:] if not a.startswith("rounds=")]
rounds = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("rounds=")), 7)
libs = [C.CDLL(os.path.abspath(p)) for p in paths]
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
M, N, K = 393216, 256, 256
sets = [dict(A=torch.randn(M, K, device="cuda").bfloat16(), R=torch.randn(M, N, device="cuda").bfloat16()) for _ in range(3)]
W = (torch.randn(N, K, device="cuda") * 0.06).bfloat16()
bias = torch.randn(N, device="cuda")
gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
Cd = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
lno = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
stats = torch.empty(M, 2, device="cuda")
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
vp, i, i64, f, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
for L in libs:
    L.pmgt_op_linear.argtypes = [i, vp, i64, vp, i64, vp, i64, i, i, i, vp, i, vp, i64, vp, i64, f, u32, vp, vp, vp, vp, vp, f, u32, vp]
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(k, s):
    rc = libs[k].pmgt_op_linear(1, P(s["A"]), K, P(W), K, P(Cd), N, M, N, K, P(bias), 0, None, N, P(s["R"]), N, 0.1, 33, P(rng), P(lno), P(stats), P(gam), P(bet), 1e-12, 0, st())
    assert rc == 0, rc
times = [[] for _ in libs]
for r in range(rounds):
    order = list(range(len(libs)))
    if r % 2: order.reverse()
    for k in order:
        for s in sets: run(k, s)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(4):
            for s in sets: run(k, s)
        ev[1].record(); torch.cuda.synchronize()
        times[k].append(ev[0].elapsed_time(ev[1]) / 12 * 1e3)
for k, p in enumerate(paths):
    print("%-40s median %.1f us/launch (min %.1f, max %.1f)" % (os.path.basename(p), np.median(times[k]), min(times[k]), max(times[k])))
