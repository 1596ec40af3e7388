"""Timing of the full-row LayerNorm GEMM (gemm_rowln512_kernel) through pmgt_op_linear at the C4 shape, several builds of the library in ONE
process.  Usage: python tools/prof/rl_ablate.py lib1.so [lib2.so ...] [M=196608] [K=512] [nores] [unfused]"""
import ctypes as C, sys, os
import numpy as np, torch
paths = [a for a in sys.argv[1:] if a.endswith(".so")]
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
M, K, N = int(kv.get("M", 196608)), int(kv.get("K", 512)), 512
libs = [C.CDLL(os.path.abspath(p)) for p in paths]
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(3)]
W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
bias = torch.randn(N, device="cuda")
R = None if "nores" in sys.argv else torch.randn(M, N, device="cuda").bfloat16()
gam, bet = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda") * 0.1
Cd = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
Y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
stats = torch.empty(M, 2, device="cuda")
rng = torch.tensor([5, 9], dtype=torch.int64, device="cuda")
vp, i, f, u32, i64 = C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_int64
for L in libs:
    L.pmgt_op_linear.argtypes = [i, vp, i64, vp, i64, vp, i64, i, i, i, vp, i, vp, i64, vp, i64, f, u32, vp, vp, vp, vp, vp, f, u32, vp]
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
opts = (1 << 13) if "unfused" in sys.argv else 0      # (OPT_UNFUSED_LN: see include/pmgt_ops.h)
def run(k, A):
    rc = libs[k].pmgt_op_linear(1, P(A), K, P(W), K, P(Cd), N, M, N, K, P(bias), 0, None, 0, P(R), N, 0.1, 33, P(rng), P(Y), P(stats), P(gam), P(bet), 1e-12, opts, st())
    assert rc == 0, rc
times = [[] for _ in libs]
for r in range(7):
    order = list(range(len(libs)))
    if r % 2: order.reverse()
    for k in order:
        for A in As: run(k, A)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(4):
            for A in As: run(k, A)
        ev[1].record(); torch.cuda.synchronize()
        times[k].append(ev[0].elapsed_time(ev[1]) / 12 * 1e3)
for k, p in enumerate(paths):
    print("%-36s M=%d K=%d %s median %.1f us/call (min %.1f, max %.1f)" % (os.path.basename(p), M, K, "unfused" if opts else "", np.median(times[k]), min(times[k]), max(times[k])))
