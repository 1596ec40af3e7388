"""Timing of pmgt_op_linear (bf16, bias-free, optional residual) at one shape for several builds of the library in ONE process.
Usage: python tools/prof/linear_ab.py lib1.so [lib2.so ...] M=393216 N=256 K=1024 [nores]"""
import ctypes as C, sys, os
import numpy as np, torch
paths = [a for a in sys.argv[1:] if a.endswith(".so")]
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
M, N, K = int(kv.get("M", 393216)), int(kv.get("N", 256)), int(kv.get("K", 1024))
libs = [C.CDLL(os.path.abspath(p)) for p in paths]
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(3)]
W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
R = None if "nores" in sys.argv else torch.randn(M, N, device="cuda").bfloat16()
outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in libs]
vp, i, f, u32, i64 = C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_int64
for L in libs:
    L.pmgt_op_linear.argtypes = [i, vp, i64, vp, i64, vp, i64, i, i, i, vp, i, vp, i64, vp, i64, f, u32, vp, vp, vp, vp, vp, f, u32, vp]
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(k, A):
    rc = libs[k].pmgt_op_linear(1, P(A), K, P(W), K, P(outs[k]), N, M, N, K, None, 0, None, 0, P(R), N, 0.0, 0, None, None, None, None, None, 1e-12, 0, st())
    assert rc == 0, rc
for k in range(len(libs)): run(k, As[0])
torch.cuda.synchronize()
for k in range(1, len(libs)):
    print(os.path.basename(paths[k]), "max |diff| vs first:", float((outs[k].float() - outs[0].float()).abs().max()))
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
    print("%-30s M=%d N=%d K=%d median %.1f us/call (min %.1f, max %.1f)" % (os.path.basename(p), M, N, K, np.median(times[k]), min(times[k]), max(times[k])))
