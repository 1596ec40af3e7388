"""Cycles per tile phase of gemm_ws_kernel (library built with -DPMGT_WS_PROF).  argv[1]: mode 0 plain, 1 gelu, 2 gelu', 3 res+drop, 4 res+drop+LN;
argv[2:5]: M N K (default 393216 256 256)"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _lib
L = _lib.ops()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (12 * 1024 * 32, 256, 256)
NS = 4
As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(NS)]
W = torch.randn(N, K, device="cuda").bfloat16()
Cs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
Rs = [torch.randn(M, N, device="cuda").bfloat16() for _ in range(NS)]
Xs = [torch.randn(M, N, device="cuda").bfloat16() for _ in range(NS)]
lnos = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
stats = torch.empty(M, 2, device="cuda")
g = torch.ones(N, device="cuda")
if mode == 4: assert N == 256
bias = torch.zeros(N, device="cuda")
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def run(i):
    epi = {0: 0, 1: 1, 2: 2, 3: 0, 4: 0}[mode]
    aux = P(Xs[i]) if mode in (1, 2) else None
    res = P(Rs[i]) if mode >= 3 else None
    dp = 0.1 if mode >= 3 else 0.0
    ln = mode == 4
    _lib.check(L.pmgt_op_linear(1, P(As[i]), K, P(W), K, P(Cs[i]), N, M, N, K, P(bias), epi, aux, N, res, N, dp, 5, P(rng),
                                P(lnos[i]) if ln else None, P(stats) if ln else None, P(g) if ln else None, P(g) if ln else None, 1e-12, st))
for k in range(9):
    if k == 8: ev[0].record()
    run(k % NS)
ev[1].record()
torch.cuda.synchronize()
print("mode %d last launch: %.1f us" % (mode, ev[0].elapsed_time(ev[1]) * 1e3))
buf = np.zeros((2, 8, 8), dtype=np.uint32)
raw = C.CDLL(_lib._build.hip_lib_path())
raw.pmgt_debug_ws_prof_read.argtypes = [C.c_void_p]
assert raw.pmgt_debug_ws_prof_read(buf.ctypes.data) == 0
lab = ["epi-end->top", "pf,sstore,gload", "barrier1", "frags+MFMA", "stage writes", "barrier2", "epilogue(prev)"]
for slot in range(2):
    for w in (0, 4):
        v = buf[slot, w].astype(np.float64)
        n = max(v[7], 1)
        # stamp k accumulates the interval ENDING at stamp k: 0 = epilogue of the previous tile .. top; 1 = pf/sstore/gload; 2 = barrier1; 3 = MFMA; 4 = stage; 5 = barrier2; 6 = tail
        print(f"slot {slot} wave {w}: tiles {int(n)} per tile {v[:6].sum() / n:6.0f} | epilogue {v[0]/n:5.0f} | pf,sstore,gload {v[1]/n:5.0f} | barrier1 {v[2]/n:5.0f} | frags+MFMA {v[3]/n:5.0f} | stage {v[4]/n:5.0f} | barrier2 {v[5]/n:5.0f}")
