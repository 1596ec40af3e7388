import sys, torch, numpy as np
sys.path.insert(0, ".")
from tests.test_ops_gpu import P, _setup, stream
from tests.test_fp8_gpu import _rand_e4m3
_lib, L = _setup()
M, N, K = 77, 40, 48
g = torch.Generator().manual_seed(M + 3 * N + K)
A8, Af = _rand_e4m3((M, K), g, 4.0)
B8, Bf = _rand_e4m3((N, K), g, 0.5)
sb = torch.rand(N, generator=g) + 0.5
bias = torch.randn(N, generator=g)
sa = torch.rand(M, generator=g) + 0.25
Cd = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
A8d, B8d, sad, sbd, bd = A8.cuda(), B8.cuda(), sa.cuda(), sb.cuda(), bias.cuda()
_lib.check(L.pmgt_op_gemm_nt_f8(P(A8d), K, None, P(sad), 1.0, P(B8d), K, P(sbd), P(Cd), N, M, N, K, P(bd), None, stream()))
ref = (Af @ Bf.T) * sa.double().unsqueeze(1) * sb.double() + bias.double()
mag = (Af.abs() @ Bf.abs().T) * sa.double().unsqueeze(1) * sb.double()
err = (Cd.double().cpu() - ref).abs()
bad = err > 1.02 * 2 ** -8 * ref.abs() + 1e-5 * mag
idx = bad.nonzero()
print("bad", int(bad.sum()))
for i, j in idx[:10].tolist():
    print(i, j, float(Cd[i, j]), float(ref[i, j]), float(err[i, j]), float(1.02 * 2 ** -8 * ref.abs()[i, j]), float(mag[i, j]))
