"""A/B of the fused forward kernel inside ONE process on ONE device (devices differ by several per cent in clock
under load, so numbers from different gpurun calls do not compare).

    python tools/prof/qa_ab.py [libA.so libB.so ...]

Every library given (default: the in-tree one) is timed at the headline shape (T = 12288 sequences, S = 32, H = 8,
hidden 256, mask + dropout on) with PMGT_QA_FORM unset; the in-tree library is also timed with the older forms when
the environment variable QA_AB_FORMS=1 is set (each form needs its own process: the form is latched at first use).
Rounds are interleaved (A B A B ...) so that clock drift hits every arm alike."""
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
g = torch.Generator(device="cuda").manual_seed(1)
xs = [torch.randn(T, S, d, device="cuda", generator=g).bfloat16() for _ in range(NS)]
W = (torch.randn(4 * d, d, device="cuda", generator=g) / d ** 0.5).bfloat16()
bias = torch.randn(4 * d, device="cuda", generator=g) * 0.1
mask = (torch.rand(T, S, device="cuda", generator=g) > 0.3).float()
mask[:, 0] = 1
qk = torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16)
ctx = torch.empty(T, S, d, device="cuda", dtype=torch.bfloat16)
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())

handles = []
for path in libs:
    L = C.CDLL(os.path.abspath(path))
    vp, i, f, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32
    L.pmgt_op_qkvc_attention_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f, f, u32, u32, vp, vp]
    handles.append(L)


def run(L, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for k in range(n):
        rc = L.pmgt_op_qkvc_attention_fwd(P(xs[k % NS]), P(W), P(bias), None if os.environ.get("QA_AB_NOMASK") else P(mask), P(qk), P(ctx), T, S, H, dh, 0.5, float(os.environ.get("QA_AB_DROP", "0.1")), 1, 2, P(rng), st)
        assert rc == 0, rc
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / n


outs = []
for L in handles:
    run(L, 5)
    outs.append((qk.clone(), ctx.clone()))
res = [[] for _ in handles]
for rnd in range(6):
    for k, L in enumerate(handles):
        res[k].append(run(L, 20))
for k, path in enumerate(libs):
    r = sorted(res[k])
    same = "" if k == 0 else f"  (Q|K|V|C equal to first: {torch.equal(outs[k][0], outs[0][0])}, ctx max diff {(outs[k][1].float() - outs[0][1].float()).abs().max().item():.3g})"
    print(f"{path}: median {r[len(r) // 2]:.1f} us/launch, min {r[0]:.1f}, max {r[-1]:.1f}{same}")
