"""A/B of the LayerNorm backward (+ its slab reduce) inside ONE process on ONE device (cf. qa_ab.py).

    python tools/prof/ln_ab.py libA.so [libB.so ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _build

libs = sys.argv[1:] or [_build.hip_lib_path()]
M, d = 12288 * 32, 256
NS = 3
g = torch.Generator(device="cuda").manual_seed(3)
sets = [dict(dy=torch.randn(M, d, device="cuda", generator=g).bfloat16(), x=torch.randn(M, d, device="cuda", generator=g).bfloat16()) for _ in range(NS)]
stats = torch.stack([torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")], dim=1).contiguous()
gamma = torch.ones(d, device="cuda")
dx = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
dxd = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
part = torch.empty((M + 63) // 64 * 3 * d, device="cuda")
out = torch.empty(3 * d, device="cuda")
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
handles = []
for path in libs:
    L = C.CDLL(os.path.abspath(path))
    vp, i, f, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32
    L.pmgt_op_layernorm_bwd.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, i, i, f, u32, f, u32, vp, vp]
    handles.append(L)


def run(L, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for k in range(n):
        s = sets[k % NS]
        rc = L.pmgt_op_layernorm_bwd(1, P(s["dy"]), P(s["x"]), P(stats), P(gamma), P(dx), P(dxd), P(part), P(out), M, d, 0.0, 0, 0.1, 7, P(rng), st)
        assert rc == 0, rc
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / n


outs = []
for L in handles:
    run(L, 4)
    outs.append((dx.clone(), dxd.clone(), out.clone()))
res = [[] for _ in handles]
for rnd in range(6):
    for k, L in enumerate(handles):
        res[k].append(run(L, 15))
for k, path in enumerate(libs):
    r = sorted(res[k])
    same = "" if k == 0 else f"  (dx equal: {torch.equal(outs[k][0], outs[0][0])}, dx_drop equal: {torch.equal(outs[k][1], outs[0][1])}, sums max diff {(outs[k][2] - outs[0][2]).abs().max().item():.3g})"
    print(f"{path}: median {r[len(r) // 2]:.1f} us per LayerNorm backward + reduce, min {r[0]:.1f}, max {r[-1]:.1f}{same}")
