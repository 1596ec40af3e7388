"""A/B of the 256 x 256 NT tile at the dX = dQKVC W shape (M = 393 216, N = 256, K = 1024, residual add), one process, one device.

    python tools/prof/nt_ab.py libA.so [libB.so ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _build

libs = sys.argv[1:] or [_build.hip_lib_path()]
M, N, K = 12 * 1024 * 32, 256, 1024
NS = 3
g = torch.Generator(device="cuda").manual_seed(9)
As = [(torch.randn(M, K, device="cuda", generator=g) * 0.05).bfloat16() for _ in range(NS)]
W = (torch.randn(N, K, device="cuda", generator=g) / 32).bfloat16()
Rs = [torch.randn(M, N, device="cuda", generator=g).bfloat16() for _ in range(NS)]
Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
handles = []
for path in libs:
    L = C.CDLL(os.path.abspath(path))
    vp, i, i64, f, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
    L.pmgt_op_gemm_nt.argtypes = [i, vp, i64, vp, vp, i64, vp, i64, i, i, i, vp, i, vp, i64, vp, i64, f, u32, vp, vp, vp]
    handles.append(L)


def run(L, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for k in range(n):
        i_ = k % NS
        rc = L.pmgt_op_gemm_nt(1, P(As[i_]), K, None, P(W), K, P(Cc), N, M, N, K, None, 0, None, 0, P(Rs[i_]), N, 0.0, 0, None, None, st)
        assert rc == 0, rc
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / n


outs = []
for L in handles:
    run(L, 4)
    outs.append(Cc.clone())
res = [[] for _ in handles]
for rnd in range(6):
    for k, L in enumerate(handles):
        res[k].append(run(L, 12))
ref = (As[3 % NS][:4096].float() @ W.float().T + Rs[3 % NS][:4096].float())
for k, path in enumerate(libs):
    r = sorted(res[k])
    err = ((outs[k][:4096].float() - ref).norm() / ref.norm()).item()
    same = "" if k == 0 else f"  (equal to first: {torch.equal(outs[k], outs[0])})"
    print(f"{path}: median {r[len(r) // 2]:.1f} us/launch, min {r[0]:.1f}, max {r[-1]:.1f}; rel err vs torch {err:.2e}{same}")
