"""Cycles per phase of attn_bwd_mfma_kernel (library built with -DPMGT_AB_PROF)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _lib
L = _lib.ops()
P = lambda t: C.c_void_p(t.data_ptr())
T, S, H, dh = 12288, 32, 8, 32
d = H * dh
NS = 3
xs = [torch.randn(T, S, 4 * d, device="cuda").bfloat16() for _ in range(NS)]
dos = [torch.randn(T, S, d, device="cuda").bfloat16() for _ in range(NS)]
dxs = [torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for k in range(7):
    i = k % NS
    if k == 6: ev[0].record()
    _lib.check(L.pmgt_op_attention_bwd(1, P(xs[i]), None, P(dos[i]), P(dxs[i]), T, S, H, dh, 0.5, 0.1, 1, 2, P(rng), st))
ev[1].record()
torch.cuda.synchronize()
print("last launch: %.1f us" % (ev[0].elapsed_time(ev[1]) * 1e3))
buf = np.zeros((2, 4, 10), dtype=np.uint32)
raw = C.CDLL(_lib._build.hip_lib_path())
raw.pmgt_debug_ab_prof_read.argtypes = [C.c_void_p]
assert raw.pmgt_debug_ab_prof_read(buf.ctypes.data) == 0
lab = ["issue loads,norms,tiles", "wave_sync", "probs", "dP", "softmax bwd+images", "products 1+stores", "alias images", "products 2+stores"]
for slot in range(2):
    for w in range(4):
        v = buf[slot, w, :8].astype(np.float64)
        if v.sum() == 0: continue
        print(f"block slot {slot} wave {w}: total {v.sum():7.0f} | " + " | ".join(f"{lab[k]} {v[k]:5.0f}" for k in range(8)))
