"""Streaming GEMM (256 x 256 weights, headline M) per epilogue mode, one process: how its time follows bytes vs epilogue work.
us per launch; U = one [M, 256] bf16 activation (201 MB).
One box, random data: plain 77, GELU 112, GELU' 143, residual 111, + dropout 121, residual + LayerNorm 152, + dropout 164
(the GELU' and LayerNorm epilogues are vector-issue-bound: same bytes as their neighbours, 30-40 us more)."""
import ctypes as C, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _lib
L = _lib.ops()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
M, N, K = 12 * 1024 * 32, 256, 256
NS = 3
g0 = torch.Generator(device="cuda").manual_seed(7)
As = [torch.randn(M, K, device="cuda", generator=g0).bfloat16() for _ in range(NS)]
W = (torch.randn(N, K, device="cuda", generator=g0) / 16).bfloat16()
Cs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
Rs = [torch.randn(M, N, device="cuda", generator=g0).bfloat16() for _ in range(NS)]
Xs = [torch.randn(M, N, device="cuda", generator=g0).bfloat16() for _ in range(NS)]
lno = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
stats = torch.empty(M, 2, device="cuda")
g = torch.ones(N, device="cuda")
bias = torch.zeros(N, device="cuda")
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def launch(i, epi, aux, res, dp, ln):
    _lib.check(L.pmgt_op_linear(1, P(As[i]), K, P(W), K, P(Cs[i]), N, M, N, K, P(bias), epi, P(Xs[i]) if aux else None, N, P(Rs[i]) if res else None, N, dp, 5, P(rng),
                                P(lno) if ln else None, P(stats) if ln else None, P(g) if ln else None, P(g) if ln else None, 1e-12, st))
cases = [("plain                  (1U in, 1U out)", 0, False, False, 0.0, False),
         ("GELU + pre-activation  (1U in, 2U out)", 1, True, False, 0.0, False),
         ("GELU' of saved pre-act (2U in, 1U out)", 2, True, False, 0.0, False),
         ("residual               (2U in, 1U out)", 0, False, True, 0.0, False),
         ("residual + dropout     (2U in, 1U out)", 0, False, True, 0.1, False),
         ("residual + LayerNorm   (2U in, 2U out)", 0, False, True, 0.0, True),
         ("res + dropout + LN     (2U in, 2U out)", 0, False, True, 0.1, True)]
def timeit(c, n=12):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for k in range(n): launch(k % NS, *c[1:])
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / n
for c in cases: timeit(c, 3)
for rnd in range(3):
    print("  ".join(f"{timeit(c):6.1f}" for c in cases))
for k, c in enumerate(cases): print(k, c[0])
