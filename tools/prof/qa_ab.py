"""A/B of the fused Q|K|V|C projection + attention forward (qkvc_attn_fwd3_kernel) between builds of the library inside ONE process.
Usage: python tools/prof/qa_ab.py libA.so libB.so [rounds]   (C2 shapes: 12 288 sequences of 32, H = 8, d = 256, mask + dropout on)"""
import ctypes as C, sys, os
import numpy as np, torch
libs = [C.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 9
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
T, S, H, dh = 12288, 32, 8, 32
d = H * dh
xs = [torch.randn(T * S, d, device="cuda").bfloat16() for _ in range(3)]
W = (torch.randn(4 * d, d, device="cuda") * 0.06).bfloat16()
bias = torch.randn(4 * d, device="cuda") * 0.1
mask = torch.ones(T, S, device="cuda")
mask[::7, 20:] = 0
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
vp, i, f, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32
for L in libs:
    L.pmgt_op_qkvc_attention_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f, f, u32, u32, vp, vp]
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
outs = [(torch.empty(T * S, 4 * d, device="cuda", dtype=torch.bfloat16), torch.empty(T * S, d, device="cuda", dtype=torch.bfloat16)) for _ in libs]
def run(k, x):
    rc = libs[k].pmgt_op_qkvc_attention_fwd(P(x), P(W), P(bias), P(mask), P(outs[k][0]), P(outs[k][1]), T, S, H, dh, 0.5, 0.1, 11, 12, P(rng), st())
    assert rc == 0, rc
for k in range(2): run(k, xs[0])
torch.cuda.synchronize()
print("Q|K|V|C identical:", bool(torch.equal(outs[0][0], outs[1][0])), " ctx identical:", bool(torch.equal(outs[0][1], outs[1][1])))
times = [[], []]
for r in range(rounds):
    for k in (0, 1) if r % 2 == 0 else (1, 0):
        for x in xs: run(k, x)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(4):
            for x in xs: run(k, x)
        ev[1].record(); torch.cuda.synchronize()
        times[k].append(ev[0].elapsed_time(ev[1]) / 12 * 1e3)
for k in (0, 1):
    print(sys.argv[1 + k], "median %.1f us/launch (min %.1f, max %.1f)" % (np.median(times[k]), min(times[k]), max(times[k])))
