"""A/B of the fused attention backward + weight-gradient kernel inside ONE process on ONE device (cf. qa_ab.py).

    python tools/prof/abw_ab.py libA.so [libB.so ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _build

libs = sys.argv[1:] or [_build.hip_lib_path()]
T, S, H, dh = 12288, 32, 8, 32
d = H * dh
NS = 3
g = torch.Generator(device="cuda").manual_seed(2)
sets = [dict(q=torch.randn(T, S, 4 * d, device="cuda", generator=g).bfloat16(), do=(torch.randn(T, S, d, device="cuda", generator=g) * 0.01).bfloat16(),
             x=torch.randn(T, S, d, device="cuda", generator=g).bfloat16()) for _ in range(NS)]
dx = torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16)
mask = (torch.rand(T, S, device="cuda", generator=g) > 0.3).float()
mask[:, 0] = 1
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
handles = []
for path in libs:
    L = C.CDLL(os.path.abspath(path))
    vp, i, f, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32
    L.pmgt_op_attention_bwd_wgrad.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, f, f, u32, u32, vp, i, vp]
    L.pmgt_op_attention_bwd_wgrad_parts.argtypes = [i]
    handles.append(L)
parts = handles[0].pmgt_op_attention_bwd_wgrad_parts(H)
slab = torch.empty(parts, 4 * d, d, device="cuda")
bslab = torch.empty(parts, 4 * d, device="cuda")


def run(L, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for k in range(n):
        s = sets[k % NS]
        rc = L.pmgt_op_attention_bwd_wgrad(P(s["q"]), P(mask), P(s["do"]), P(s["x"]), P(dx), P(slab), P(bslab), T, H, 0.5, float(os.environ.get("QA_AB_DROP", "0.1")), 11, 12, P(rng), 1, st)
        assert rc == 0, rc
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / n


outs = []
for L in handles:
    run(L, 4)
    outs.append((dx.clone(), slab.clone()))
res = [[] for _ in handles]
for rnd in range(6):
    for k, L in enumerate(handles):
        res[k].append(run(L, 15))
for k, path in enumerate(libs):
    r = sorted(res[k])
    same = "" if k == 0 else f"  (dQKVC equal to first: {torch.equal(outs[k][0], outs[0][0])}, dW equal: {torch.equal(outs[k][1], outs[0][1])})"
    print(f"{path}: median {r[len(r) // 2]:.1f} us/launch, min {r[0]:.1f}, max {r[-1]:.1f}{same}")
