"""Cycles per phase of gemm_nt_big_kernel (library built with -DPMGT_TN_PROF)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _lib
L = _lib.ops()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
M, N, K = 12 * 1024 * 32, 256, 1024
NS = 3
As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(NS)]
W = torch.randn(N, K, device="cuda").bfloat16()
Cs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
Rs = [torch.randn(M, N, device="cuda").bfloat16() for _ in range(NS)]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for k in range(7):
    i = k % NS
    if k == 6: ev[0].record()
    _lib.check(L.pmgt_op_gemm_nt(1, P(As[i]), K, None, P(W), K, P(Cs[i]), N, M, N, K, None, 0, None, 0, P(Rs[i]), N, 0.0, 0, None, None, st))
ev[1].record()
torch.cuda.synchronize()
print("last launch: %.1f us" % (ev[0].elapsed_time(ev[1]) * 1e3))
buf = np.zeros((2, 8, 6), dtype=np.uint32)
raw = C.CDLL(_lib._build.hip_lib_path())
raw.pmgt_debug_nt_prof_read.argtypes = [C.c_void_p]
assert raw.pmgt_debug_nt_prof_read(buf.ctypes.data) == 0
nk = K // 32
for slot in range(2):
    for w in (0, 3, 4, 7):
        v = buf[slot, w].astype(np.float64)
        print(f"block slot {slot} wave {w}: tile total {v.sum():7.0f} | prologue {v[4]:5.0f} | per k-step: vmcnt {v[0]/nk:5.0f} barrier {v[1]/nk:5.0f} issue {v[2]/nk:5.0f} "
              f"LDS+MFMA {v[3]/nk:5.0f} | epilogue {v[5]:6.0f}")
