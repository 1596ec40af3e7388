"""Ablation timing of attn_bwd_wgrad_kernel: several builds of the library (each with parts of the kernel compiled out: results are
garbage, only the time matters) timed in ONE process, interleaved.  Usage: python tools/prof/abw_ablate.py lib1.so lib2.so ... [rounds=N]"""
import ctypes as C, sys, os
import numpy as np, torch
paths = [a for a in sys.argv[1:] if not a.startswith("rounds=")]
rounds = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("rounds=")), 7)
libs = [C.CDLL(os.path.abspath(p)) for p in paths]
P = lambda t: C.c_void_p(t.data_ptr())
T, S, H, dh = 12288, 32, 8, 32
d = H * dh
sets = [dict(q=torch.randn(T, S, 4 * d, device="cuda").bfloat16(), do=torch.randn(T, S, d, device="cuda").bfloat16(),
             x=torch.randn(T, S, d, device="cuda").bfloat16()) for _ in range(3)]
mask = torch.ones(T, S, device="cuda")
for L in libs:
    vp, i, f, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32
    L.pmgt_op_attention_bwd_wgrad.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, f, f, u32, u32, vp, i, vp]
parts = libs[0].pmgt_op_attention_bwd_wgrad_parts(H)
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
dx = torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16)
slab = torch.empty(parts, 4 * d, d, device="cuda")
bslab = torch.empty(parts, 4 * d, device="cuda")
def run(k, s):
    rc = libs[k].pmgt_op_attention_bwd_wgrad(P(s["q"]), P(mask), P(s["do"]), P(s["x"]), P(dx), P(slab), P(bslab), T, H, 0.5, 0.1, 11, 12, P(rng), 1, st())
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
