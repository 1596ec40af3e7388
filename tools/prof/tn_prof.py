"""Cycles per k-step phase of gemm_tn_big_kernel (library built with -DPMGT_TN_PROF)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _lib
L = _lib.ops()
P = lambda t: C.c_void_p(t.data_ptr())
M, N1, N2 = 12 * 1024 * 32, 1024, 256
NS = 3
Ps = [torch.randn(M, N1, device="cuda").bfloat16() for _ in range(NS)]
Qs = [torch.randn(M, N2, device="cuda").bfloat16() for _ in range(NS)]
slab = torch.empty(L.pmgt_op_gemm_tn_slab_elems(1, M, N1, N2), device="cuda")
o = torch.empty(N1, N2, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for k in range(7):
    i = k % NS
    if k == 6: ev[0].record()
    _lib.check(L.pmgt_op_gemm_tn(1, P(Ps[i]), N1, P(Qs[i]), N2, None, M, N1, N2, P(slab), P(o), 0, None, st))
ev[1].record()
torch.cuda.synchronize()
print("last launch (GEMM + reduce): %.1f us" % (ev[0].elapsed_time(ev[1]) * 1e3))
buf = np.zeros((2, 8, 4), dtype=np.uint32)
raw = C.CDLL(_lib._build.hip_lib_path())
raw.pmgt_debug_tn_prof_read.argtypes = [C.c_void_p]
assert raw.pmgt_debug_tn_prof_read(buf.ctypes.data) == 0
nk = M // 64 // 32
lab = ["vmcnt wait", "barrier", "DMA issue", "LDS reads + MFMA"]
for slot in range(2):
    for w in (0, 3, 4, 7):
        v = buf[slot, w].astype(np.float64) / nk
        print(f"block slot {slot} wave {w}: per k-step {v.sum():6.0f} | " + " | ".join(f"{lab[k]} {v[k]:5.0f}" for k in range(4)))
