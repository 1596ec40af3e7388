"""A/B of the streaming GEMM with the residual + dropout + LayerNorm epilogue (fwd.gemm_attn_out / fwd.gemm_ffn2 shape) across libraries,
one process, one device.   python tools/prof/ws_ab.py libA.so [libB.so ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _build

libs = sys.argv[1:] or [_build.hip_lib_path()]
M, N, K = 12 * 1024 * 32, 256, 256
NS = 3
g0 = torch.Generator(device="cuda").manual_seed(7)
As = [torch.randn(M, K, device="cuda", generator=g0).bfloat16() for _ in range(NS)]
W = (torch.randn(N, K, device="cuda", generator=g0) / 16).bfloat16()
Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
Rs = [torch.randn(M, N, device="cuda", generator=g0).bfloat16() for _ in range(NS)]
lno = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
stats = torch.empty(M, 2, device="cuda")
g = torch.ones(N, device="cuda")
bias = torch.zeros(N, device="cuda")
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
handles = []
for path in libs:
    L = C.CDLL(os.path.abspath(path))
    vp, i, i64, f, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
    L.pmgt_op_linear.argtypes = [i, vp, i64, vp, i64, vp, i64, i, i, i, vp, i, vp, i64, vp, i64, f, u32, vp, vp, vp, vp, vp, f, vp]
    handles.append(L)


def run(L, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for k in range(n):
        i_ = k % NS
        rc = L.pmgt_op_linear(1, P(As[i_]), K, P(W), K, P(Cc), N, M, N, K, P(bias), 0, None, N, P(Rs[i_]), N, 0.1, 5, P(rng), P(lno), P(stats), P(g), P(g), 1e-12, st)
        assert rc == 0, rc
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / n


for L in handles: run(L, 4)
res = [[] for _ in handles]
for rnd in range(6):
    for k, L in enumerate(handles):
        res[k].append(run(L, 12))
for k, path in enumerate(libs):
    r = sorted(res[k])
    print(f"{path}: median {r[len(r) // 2]:.1f} us/launch, min {r[0]:.1f}, max {r[-1]:.1f}")
