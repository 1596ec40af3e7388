"""Cycles per interval inside attn_bwd_wgrad_kernel (library built with PMGT_EXTRA_HIP_FLAGS=-DPMGT_ABW_PROF).
w = work of the role in the interval, b = its wait at the barrier that closes it."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _lib, _build
if len(sys.argv) > 1:          # a library of its own (built with -DPMGT_ABW_PROF), e.g. build/abl/abw3_PROF.so
    _build.hip_lib_path = lambda: os.path.abspath(sys.argv[1])
L = _lib.ops()
P = lambda t: C.c_void_p(t.data_ptr())
T, S, H, dh = 12288, 32, 8, 32
d = H * dh
NSET = 3
sets = []
for i in range(NSET):
    sets.append(dict(q=torch.randn(T, S, 4 * d, device="cuda").bfloat16(), do=torch.randn(T, S, d, device="cuda").bfloat16(),
                     x=torch.randn(T, S, d, device="cuda").bfloat16(), dx=torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16)))
mask = torch.ones(T, S, device="cuda")
parts = L.pmgt_op_attention_bwd_wgrad_parts(H)
slab = torch.empty(parts, 4 * d, d, device="cuda")
bslab = torch.empty(parts, 4 * d, device="cuda")
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(s, p=0.1):
    _lib.check(L.pmgt_op_attention_bwd_wgrad(P(s["q"]), P(mask), P(s["do"]), P(s["x"]), P(s["dx"]), P(slab), P(bslab), T, H, 0.5, p, 11, 12, P(rng), 1, st()))
for s in sets: run(s)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(5):
    for s in sets: run(s)
ev[1].record(); torch.cuda.synchronize()
print("us per launch", ev[0].elapsed_time(ev[1]) / 15 * 1e3)
if hasattr(L, "pmgt_debug_abw_prof_read"):
    out = (C.c_ulonglong * 128)()
    L.pmgt_debug_abw_prof_read(out)
    a = np.array(list(out), dtype=np.float64).reshape(16, 8)
    names = [f"attention wave {w} (pair {w >> 2}, tile {(w >> 1) & 1}, {'cosine' if w % 2 == 0 else 'dot'})" for w in range(8)] + [f"GEMM wave {w}" for w in range(8, 16)]
    for role, name in enumerate(names):
        n = max(a[role, 7], 1)
        print(name, "steps", int(n), "cycles/step:", " ".join(f"{lbl}={a[role, k] / n:.0f}" for k, lbl in enumerate(("w1", "b1", "w2", "b2", "w3", "b0"))),
              "total", f"{a[role, :6].sum() / n:.0f}")
