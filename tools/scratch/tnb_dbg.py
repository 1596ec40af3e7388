import sys, torch
sys.path.insert(0, ".")
from tests.test_ops_gpu import P, _setup, stream, to_dev
_lib, L = _setup()
M, N1, N2 = 70016, 1024, 256
g = torch.Generator().manual_seed(1)
Pm = torch.randn(M, N1, generator=g) + 0.25
Q = torch.randn(M, N2, generator=g)
Pd, Qd = to_dev(Pm, torch.bfloat16), to_dev(Q, torch.bfloat16)
slab = torch.empty(L.pmgt_op_gemm_tn_slab_elems(1, M, N1, N2), device="cuda")
bslab = torch.full((512 * N1,), float("nan"), device="cuda")
out = torch.full((N1, N2), float("nan"), device="cuda")
bout = torch.full((N1,), float("nan"), device="cuda")
_lib.check(L.pmgt_op_gemm_tn_bias(1, P(Pd), N1, P(Qd), N2, M, N1, N2, P(slab), P(out), P(bslab), P(bout), 0, 0, stream()))
torch.cuda.synchronize()
bs = bslab.view(512, N1)[:64].cpu()
print("nan rows per split (first 8 splits):", [int(torch.isnan(bs[s]).sum()) for s in range(8)])
nanc = torch.isnan(bs[0]).nonzero().flatten().tolist()
print("nan cols in split 0:", nanc[:40], "...", len(nanc))
ref = Pd.double().cpu()
chunk = (M + 63) // 64
chunk = (chunk + 31) // 32 * 32
r0 = ref[:chunk].sum(0)
good = ~torch.isnan(bs[0])
print("max err on written:", float((bs[0][good].double() - r0[good]).abs().max()), "ref scale", float(r0.abs().max()))
