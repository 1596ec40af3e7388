"""A/B of attn_bwd_wgrad_kernel between two builds of the library INSIDE one process (boxes differ by 3-5 % in clock under this
kernel: numbers from different gpurun calls do not compare).  Usage: python tools/prof/abw_ab.py libA.so libB.so [rounds]
Prints the median us per launch of each library over interleaved rounds, and whether their outputs agree."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
libs = [C.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
P = lambda t: C.c_void_p(t.data_ptr())
T, S, H, dh = 12288, 32, 8, 32
d = H * dh
sets = [dict(q=torch.randn(T, S, 4 * d, device="cuda").bfloat16(), do=torch.randn(T, S, d, device="cuda").bfloat16(),
             x=torch.randn(T, S, d, device="cuda").bfloat16()) for _ in range(3)]
mask = torch.ones(T, S, device="cuda")
mask[::7, 20:] = 0
for L in libs:
    L.pmgt_op_attention_bwd_wgrad_parts.argtypes = [C.c_int]
    vp, i, f, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32
    L.pmgt_op_attention_bwd_wgrad.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, f, f, u32, u32, vp, i, vp]
parts = libs[0].pmgt_op_attention_bwd_wgrad_parts(H)
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
outs = []
for L in libs:
    dx = torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16)
    slab = torch.empty(parts, 4 * d, d, device="cuda")
    bslab = torch.empty(parts, 4 * d, device="cuda")
    outs.append((dx, slab, bslab))
def run(k, s, p=0.1):
    dx, slab, bslab = outs[k]
    rc = libs[k].pmgt_op_attention_bwd_wgrad(P(s["q"]), P(mask), P(s["do"]), P(s["x"]), P(dx), P(slab), P(bslab), T, H, 0.5, p, 11, 12, P(rng), 1, st())
    assert rc == 0, rc
for k in range(len(libs)):
    run(k, sets[0])
torch.cuda.synchronize()
a, b = outs[0], outs[1]
print("dQKVC max |a - b|:", float((a[0].float() - b[0].float()).abs().max()), " max |a|:", float(a[0].float().abs().max()),
      " identical:", bool(torch.equal(a[0], b[0])), " dW rel diff:", float((a[1].sum(0) - b[1].sum(0)).abs().max() / a[1].sum(0).abs().max()))
times = [[], []]
for r in range(rounds):
    for k in (0, 1) if r % 2 == 0 else (1, 0):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for s in sets: run(k, s)
        ev[0].record()
        for _ in range(4):
            for s in sets: run(k, s)
        ev[1].record(); torch.cuda.synchronize()
        times[k].append(ev[0].elapsed_time(ev[1]) / 12 * 1e3)
for k in (0, 1):
    print(sys.argv[1 + k], "median %.1f us/launch (min %.1f, max %.1f)" % (np.median(times[k]), min(times[k]), max(times[k])))
