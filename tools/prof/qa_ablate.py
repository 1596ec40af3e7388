"""Ablation timing of qkvc_attn_fwd3_kernel: several builds of the library (parts of the kernel compiled out: results are garbage, only
the time matters) timed in ONE process, interleaved.  Usage: python tools/prof/qa_ablate.py lib1.so lib2.so ... [rounds=N]"""
import ctypes as C, sys, os
import numpy as np, torch
paths = [a for a in sys.argv[1:] if not a.startswith("rounds=") and a != "check"]
rounds = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("rounds=")), 7)
libs = [C.CDLL(os.path.abspath(p)) for p in paths]
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
qk = torch.empty(T * S, 4 * d, device="cuda", dtype=torch.bfloat16)
cx = torch.empty(T * S, d, device="cuda", dtype=torch.bfloat16)
def run(k, x):
    rc = libs[k].pmgt_op_qkvc_attention_fwd(P(x), P(W), P(bias), P(mask), P(qk), P(cx), T, S, H, dh, 0.5, 0.1, 11, 12, P(rng), st())
    assert rc == 0, rc
if "check" in sys.argv:      # outputs of every build against the first one (variants that must be bit-identical)
    ref = None
    for k, p_ in enumerate(paths):
        qk.zero_(); cx.zero_()
        run(k, xs[0]); torch.cuda.synchronize()
        if ref is None: ref = (qk.clone(), cx.clone())
        else: print("%-40s Q|K|V|C identical: %s  ctx identical: %s" % (os.path.basename(p_), bool(torch.equal(qk, ref[0])), bool(torch.equal(cx, ref[1]))))
times = [[] for _ in libs]
for r in range(rounds):
    order = list(range(len(libs)))
    if r % 2: order.reverse()
    for k in order:
        for x in xs: run(k, x)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(4):
            for x in xs: run(k, x)
        ev[1].record(); torch.cuda.synchronize()
        times[k].append(ev[0].elapsed_time(ev[1]) / 12 * 1e3)
for k, p in enumerate(paths):
    print("%-40s median %.1f us/launch (min %.1f, max %.1f)" % (os.path.basename(p), np.median(times[k]), min(times[k]), max(times[k])))
