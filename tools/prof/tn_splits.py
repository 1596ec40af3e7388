"""One 256 x 256 weight gradient (GEMM + slab reduce) at the headline M; run once per PMGT_TN_WGS value (workgroups the 128 x 128 tile aims at).
Operator level: 256 -> 119 us, 512 -> 102 us; in the step 256 stays best (9.84 vs 9.96 ms: the slabs double)."""
import ctypes as C, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _lib
L = _lib.ops()
P = lambda t: C.c_void_p(t.data_ptr())
M, N1, N2 = 12 * 1024 * 32, 256, 256
NS = 3
g = torch.Generator(device="cuda").manual_seed(5)
Ps = [(torch.randn(M, N1, device="cuda", generator=g) * 0.05).bfloat16() for _ in range(NS)]
Qs = [torch.randn(M, N2, device="cuda", generator=g).bfloat16() for _ in range(NS)]
slab = torch.empty(L.pmgt_op_gemm_tn_slab_elems(1, M, N1, N2), device="cuda")
o = torch.empty(N1, N2, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for k in range(n):
        i = k % NS
        _lib.check(L.pmgt_op_gemm_tn(1, P(Ps[i]), N1, P(Qs[i]), N2, None, M, N1, N2, P(slab), P(o), 0, None, st))
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / n
run(5)
r = sorted(run(20) for _ in range(5))
print(os.environ.get("PMGT_TN_WGS", "256"), "slab MB %.1f" % (slab.numel() * 4 / 1e6), "GEMM + reduce: median %.1f us, min %.1f" % (r[2], r[0]))
