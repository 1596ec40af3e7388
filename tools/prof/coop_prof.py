"""Cycles per interval inside attn_bwd_coop_kernel<64, 4> (library built with PMGT_EXTRA_HIP_FLAGS=-DPMGT_COOP_PROF=<workgroup index>)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _lib
L = _lib.ops()
P = lambda t: None if t is None else t.data_ptr()
n_seq, S, H, dh = 3072, 64, 8, 64
d, M = H * dh, n_seq * S
qkvc = (torch.randn(M, 4 * d, device="cuda") * 0.5).to(torch.bfloat16)
dctx = (torch.randn(M, d, device="cuda") * 0.1).to(torch.bfloat16)
dq = torch.empty_like(qkvc)
mask = torch.ones(n_seq, S, device="cuda")
rng = torch.tensor([1, 0], dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def bwd(): _lib.check(L.pmgt_op_attention_bwd(1, P(qkvc), P(mask), P(dctx), P(dq), n_seq, S, H, dh, 0.5, 0.1, 9, 10, P(rng), st))
for _ in range(3): bwd()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): bwd()
b.record(); torch.cuda.synchronize()
print("us per launch", a.elapsed_time(b) / 10 * 1e3)
if hasattr(_lib.hip(), "pmgt_debug_coop_prof_read"):
    out = (C.c_ulonglong * 32)()
    _lib.hip().pmgt_debug_coop_prof_read(out)
    v = np.array(list(out), dtype=np.float64).reshape(4, 8)
    for w in range(4):
        print("wave", w, " ".join(f"{lbl}={v[w, k]:.0f}" for k, lbl in enumerate(("issue", "landed+tiles", "bar1", "half1", "bar2", "half2+stores"))), "total", f"{v[w, :6].sum():.0f}")
