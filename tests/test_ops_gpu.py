"""Per-kernel parity tests: every HIP kernel is called through the C ABI (pmgt_op_*) and compared
with a plain torch fp32/fp64 CPU computation of the same op (for attention: the oracle's
restatement of pmgt/pmgt/modeling_pmgt.py:420-534 with autograd for the backward)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {"fp32": (0, torch.float32), "bf16": (1, torch.bfloat16)}


def _setup():
    from pmgt_amd import _lib
    return _lib, _lib.ops()       # pmgt_op_* with the path_opts argument taken from L.path (L.use("tile_gemm"), ...)


def P(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_dev(x, tdt):
    return x.to("cuda", tdt).contiguous()


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def tol(dt):
    return 2e-5 if dt == "fp32" else 2e-2


def rounded(x, tdt):
    return x.to(tdt).to(torch.float64)


# ------------------------------------------------------------------------------------------- gemm_nt
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 96), (77, 40, 24), (1000, 256, 1536), (513, 264, 256)])
def test_gemm_nt_plain(dt, M, N, K):
    _lib, L = _setup()
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) * 0.3
    bias = torch.randn(N, generator=g)
    Ad, Bd, bd = to_dev(A, tdt), to_dev(B, tdt), to_dev(bias, torch.float32)
    Cd = torch.full((M, N), float("nan"), device="cuda", dtype=tdt)
    _lib.check(L.pmgt_op_gemm_nt(code, P(Ad), K, None, P(Bd), K, P(Cd), N, M, N, K, P(bd), 0, None, 0, None, 0, 0.0, 0,
                                 None, None, stream()))
    ref = rounded(A, tdt) @ rounded(B, tdt).T + bias.double()
    assert rel_err(Cd, ref) < tol(dt), rel_err(Cd, ref)


@pytest.mark.parametrize("M,N,K", [(24600, 512, 1536), (30000, 256, 768)])
def test_gemm_nt_big_tile_row_gather(M, N, K):
    """The per-token feature projection (token mode: A = rows of the frozen table picked by node id, written into a column block of
    a wider buffer): 256 x 256 tile with the row gather in its LDS-DMA addresses, against the 128 x 128 tile kernel and fp64."""
    _lib, L = _setup()
    g = torch.Generator().manual_seed(M + K)
    R = 5000
    table = torch.randn(R, K, generator=g).cuda().bfloat16()
    rows = torch.randint(0, R, (M,), generator=g).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).cuda().bfloat16()
    bias = torch.randn(N, generator=g).cuda()
    outs = []
    for opts in ((), ("tile_gemm",)):
        L.use(*opts)
        E = torch.full((M, 2 * N), float("nan"), device="cuda", dtype=torch.bfloat16)
        Cv = E[:, N:]                      # second column block, row stride 2 N
        _lib.check(L.pmgt_op_gemm_nt(1, P(table), K, P(rows), P(W), K, C.c_void_p(Cv.data_ptr()), 2 * N, M, N, K, P(bias), 0, None, 0, None, 0,
                                     0.0, 0, None, None, stream()))
        assert torch.isnan(E[:, :N].float()).all()           # the neighbouring block is untouched
        outs.append(E[:, N:].float())
    L.use()
    assert rel_err(outs[0], outs[1]) < 5e-3
    ref = table.double().cpu()[rows.cpu()] @ W.double().cpu().T + bias.double().cpu()
    assert rel_err(outs[0], ref) < 1e-2


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("K", [72, 96, 256])
def test_gemm_nt_epilogues_and_gather(dt, K):
    _lib, L = _setup()
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(5)
    M, N, R = 333, 136, 50
    table = torch.randn(R, K, generator=g)
    rows = torch.randint(0, R, (M,), generator=g)
    B = torch.randn(N, K, generator=g) * 0.3
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    td, Bd, bd, rd, rowd = to_dev(table, tdt), to_dev(B, tdt), to_dev(bias, torch.float32), to_dev(res, tdt), rows.cuda()
    Cd = torch.zeros(M, N, device="cuda", dtype=tdt)
    aux = torch.zeros(M, N, device="cuda", dtype=tdt)
    # gather + bias + gelu (stores pre-activation) + residual
    _lib.check(L.pmgt_op_gemm_nt(code, P(td), K, P(rowd), P(Bd), K, P(Cd), N, M, N, K, P(bd), 1, P(aux), N, P(rd), N, 0.0, 0,
                                 None, None, stream()))
    pre = rounded(table, tdt)[rows] @ rounded(B, tdt).T + bias.double()
    pre_r = rounded(pre.float(), tdt)
    ref = torch.nn.functional.gelu(pre_r) + rounded(res, tdt)
    assert rel_err(aux, pre) < tol(dt)
    assert rel_err(Cd, ref) < tol(dt)
    # gelu-grad epilogue: C = (A B^T) * gelu'(aux)
    A = torch.randn(M, K, generator=g)
    Ad = to_dev(A, tdt)
    _lib.check(L.pmgt_op_gemm_nt(code, P(Ad), K, None, P(Bd), K, P(Cd), N, M, N, K, None, 2, P(aux), N, None, 0, 0.0, 0, None,
                                 None, stream()))
    x = aux.double().cpu().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    ref2 = (rounded(A, tdt) @ rounded(B, tdt).T) * x.grad
    assert rel_err(Cd, ref2) < tol(dt)
    # device-side row count: rows >= m_dev untouched
    Cd.fill_(7.0)
    cnt = torch.tensor([100], dtype=torch.int32, device="cuda")
    _lib.check(L.pmgt_op_gemm_nt(code, P(Ad), K, None, P(Bd), K, P(Cd), N, M, N, K, None, 0, None, 0, None, 0, 0.0, 0, None,
                                 P(cnt), stream()))
    ref3 = rounded(A, tdt) @ rounded(B, tdt).T
    assert rel_err(Cd[:100], ref3[:100]) < tol(dt)
    assert bool((Cd[100:].float() == 7.0).all())


# ------------------------------------------------------------------------------------------- gemm_tn
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N1,N2", [(500, 64, 96), (4096, 256, 256), (130, 24, 1536), (3000, 1024, 256),
                                     (40000, 128, 256), (70001, 128, 128),      # long splits: steady state of the DMA rings
                                     (70016, 1024, 256), (66000, 512, 512),      # 256 x 256 tile (bf16, M >= 65536)
                                     (70001, 256, 256), (65600, 128, 256), (66011, 256, 512)])      # large M on the 128 x 128 tile, ragged tails
def test_gemm_tn(dt, M, N1, N2):
    _lib, L = _setup()
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(M + N1)
    Pm = torch.randn(M, N1, generator=g)
    Q = torch.randn(M, N2, generator=g)
    Pd, Qd = to_dev(Pm, tdt), to_dev(Q, tdt)
    slab = torch.empty(L.pmgt_op_gemm_tn_slab_elems(code, M, N1, N2), device="cuda")
    out = torch.full((N1, N2), float("nan"), device="cuda")
    _lib.check(L.pmgt_op_gemm_tn(code, P(Pd), N1, P(Qd), N2, None, M, N1, N2, P(slab), P(out), 0, None, stream()))
    ref = rounded(Pm, tdt).T @ rounded(Q, tdt)
    assert rel_err(out, ref) < (1e-5 if dt == "fp32" else 2e-3)
    # accumulate + gather on Q + device row count
    R = 40
    table = torch.randn(R, N2, generator=g)
    rows = torch.randint(0, R, (M,), generator=g)
    td, rowd = to_dev(table, tdt), rows.cuda()
    cnt = torch.tensor([M - 37], dtype=torch.int32, device="cuda")
    _lib.check(L.pmgt_op_gemm_tn(code, P(Pd), N1, P(td), N2, P(rowd), M, N1, N2, P(slab), P(out), 1, P(cnt), stream()))
    ref2 = ref + rounded(Pm, tdt)[: M - 37].T @ rounded(table, tdt)[rows[: M - 37]]
    assert rel_err(out, ref2) < (1e-5 if dt == "fp32" else 2e-3)
    # device row count without a gather (the compacted last layer's weight gradients), overwrite
    _lib.check(L.pmgt_op_gemm_tn(code, P(Pd), N1, P(Qd), N2, None, M, N1, N2, P(slab), P(out), 0, P(cnt), stream()))
    ref3 = rounded(Pm, tdt)[: M - 37].T @ rounded(Q, tdt)[: M - 37]
    assert rel_err(out, ref3) < (1e-5 if dt == "fp32" else 2e-3)


@pytest.mark.parametrize("dt,M,N1,N2,gather,kern", [("bf16", 31744, 256, 256, False, b"tn_dma"), ("bf16", 31744, 256, 1536, True, b"tn_dma_gather"),
                                                    ("bf16", 70016, 1024, 256, False, b"tn_big"), ("bf16", 66000, 512, 512, True, b"tn_big_gather"),
                                                    ("fp32", 31744, 256, 256, False, b"tn_tile"), ("fp32", 3000, 256, 64, True, b"tn_tile")])
def test_gemm_tn_row_chunks_follow_the_live_row_count(dt, M, N1, N2, gather, kern):
    """Weight gradients whose row count is a device-side number (masked NFR rows, the compacted last layer): every TN kernel re-derives its
    split's row chunk from *m_dev (tn_live_chunk), so the cases that matter are a live count FAR below the capacity (5 000 of 31 744 rows),
    below splits * 64 (some splits own no rows and must still write a zero slab) and 0 (the output is overwritten with zeros, not left
    alone).  Each kernel by launch trace; overwrite and accumulate forms against fp64."""
    _lib, L = _setup()
    H = _lib.hip()
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(M + N2)
    Pm = torch.randn(M, N1, generator=g)
    R = 4096
    Q = torch.randn(R if gather else M, N2, generator=g)
    rows = torch.randint(0, R, (M,), generator=g) if gather else None
    Pd, Qd = to_dev(Pm, tdt), to_dev(Q, tdt)
    rowd = rows.cuda() if gather else None
    slab = torch.empty(L.pmgt_op_gemm_tn_slab_elems(code, M, N1, N2), device="cuda")
    Pr, Qr = rounded(Pm, tdt), rounded(Q, tdt)
    tolr = 1e-5 if dt == "fp32" else 2e-3
    for live in (0, 1, 63, 5000, M // 7, M - 37, M):
        if live > M:
            continue
        cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
        out = torch.full((N1, N2), float("nan"), device="cuda")
        H.pmgt_launch_trace_reset()
        _lib.check(L.pmgt_op_gemm_tn(code, P(Pd), N1, P(Qd), N2, P(rowd), M, N1, N2, P(slab), P(out), 0, P(cnt), stream()))
        torch.cuda.synchronize()
        assert H.pmgt_launch_trace_count(kern) == 1, (kern, live)
        q = Qr[rows[:live]] if gather else Qr[:live]
        ref = Pr[:live].T @ q
        if live == 0:
            assert bool((out == 0).all()), "live == 0: every split writes a zero slab and the reduction overwrites the output"
        else:
            assert rel_err(out, ref) < tolr, live
        # accumulate on top of a known value
        out.fill_(1.5)
        _lib.check(L.pmgt_op_gemm_tn(code, P(Pd), N1, P(Qd), N2, P(rowd), M, N1, N2, P(slab), P(out), 1, P(cnt), stream()))
        assert float((out.double().cpu() - (ref + 1.5)).abs().max()) <= tolr * max(float(ref.abs().max()), 1.0) + 1e-6, live


@pytest.mark.parametrize("dt,M,N1,N2,hm", [("fp32", 700, 256, 64, False), ("bf16", 3000, 1024, 256, False), ("bf16", 3000, 1024, 256, True),
                                           ("bf16", 70016, 1024, 256, False), ("bf16", 70016, 1024, 256, True),      # 256 x 256 tile
                                           ("bf16", 66000, 512, 512, False), ("bf16", 40000, 256, 1536, False),
                                           ("bf16", 70001, 256, 256, False), ("bf16", 70001, 256, 256, True),      # 8-wave 128 x 128 tile, ragged tail
                                           ("bf16", 65600, 128, 512, False)])
def test_gemm_tn_bias_sums_and_head_major_rows(dt, M, N1, N2, hm):
    """The bias gradient rides along in the weight-gradient kernels as MFMAs against a ones fragment, and the Q|K|V|C
    projection's P operand arrives head-major (rows of the result are permuted back to q | k | v | c): every TN kernel
    (register-staged, LDS-DMA, 256 x 256 tile) against fp64 column sums / products."""
    _lib, L = _setup()
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(M + N1 + N2)
    Pm = torch.randn(M, N1, generator=g) + 0.25          # non-zero column means
    Q = torch.randn(M, N2, generator=g)
    Pd, Qd = to_dev(Pm, tdt), to_dev(Q, tdt)
    slab = torch.empty(L.pmgt_op_gemm_tn_slab_elems(code, M, N1, N2), device="cuda")
    bslab = torch.full((512 * N1,), float("nan"), device="cuda")
    out = torch.full((N1, N2), float("nan"), device="cuda")
    bout = torch.full((N1,), float("nan"), device="cuda")
    d, dh = (N1 // 4, 32) if hm else (0, 0)
    _lib.check(L.pmgt_op_gemm_tn_bias(code, P(Pd), N1, P(Qd), N2, M, N1, N2, P(slab), P(out), P(bslab), P(bout), d, dh, stream()))
    ref = rounded(Pm, tdt).T @ rounded(Q, tdt)
    bref = rounded(Pm, tdt).sum(0)
    if hm:      # column (head h, matrix m, w) of P -> row m * d + h * dh + w of the result
        n1 = torch.arange(N1)
        dst = ((n1 // dh) % 4) * d + (n1 // (4 * dh)) * dh + n1 % dh
        r2, b2 = torch.empty_like(ref), torch.empty_like(bref)
        r2[dst], b2[dst] = ref, bref
        ref, bref = r2, b2
    assert rel_err(out, ref) < (1e-5 if dt == "fp32" else 2e-3)
    assert torch.isfinite(bout).all()
    assert rel_err(bout, bref) < (1e-5 if dt == "fp32" else 2e-3), rel_err(bout, bref)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_colsum(dt):
    _lib, L = _setup()
    code, tdt = DT[dt]
    M, N = 1000, 1032
    Y = torch.randn(M, N)
    Yd = to_dev(Y, tdt)
    slab = torch.empty(((M + 95) // 96) * N, device="cuda")       # ceil(M / 96) partial rows (include/pmgt_capi.h)
    out = torch.empty(N, device="cuda")
    _lib.check(L.pmgt_op_colsum(code, P(Yd), N, M, N, P(slab), P(out), stream()))
    assert rel_err(out, rounded(Y, tdt).sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------- layernorm
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("M,d", [(100, 64), (257, 256), (64, 512), (33, 1024), (10, 128)])
def test_layernorm_fwd_bwd(dt, M, d):
    _lib, L = _setup()
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(d)
    x = torch.randn(M, d, generator=g) * 2 + 0.5
    gam = 1 + 0.1 * torch.randn(d, generator=g)
    bet = 0.1 * torch.randn(d, generator=g)
    dy = torch.randn(M, d, generator=g)
    xd, gd, bd, dyd = to_dev(x, tdt), to_dev(gam, torch.float32), to_dev(bet, torch.float32), to_dev(dy, tdt)
    y = torch.empty(M, d, device="cuda", dtype=tdt)
    stats = torch.empty(M, 2, device="cuda")
    _lib.check(L.pmgt_op_layernorm_fwd(code, P(xd), P(y), P(stats), P(gd), P(bd), M, d, 1e-12, 0.0, 0, None, stream()))
    xr = rounded(x, tdt).requires_grad_(True)
    gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-12)
    assert rel_err(y, ref.detach()) < tol(dt)
    ref.backward(rounded(dy, tdt))
    dx = torch.empty(M, d, device="cuda", dtype=tdt)
    part = torch.empty(((M + 63) // 64) * 3 * d, device="cuda")
    dgb = torch.empty(3 * d, device="cuda")
    _lib.check(L.pmgt_op_layernorm_bwd(code, P(dyd), P(xd), P(stats), P(gd), P(dx), None, P(part), P(dgb), M, d, 0.0, 0, 0.0, 0,
                                       None, stream()))
    assert rel_err(dx, xr.grad) < tol(dt)
    assert rel_err(dgb[:d], gr.grad) < 1e-4
    assert rel_err(dgb[d:2 * d], br.grad) < 1e-4
    assert rel_err(dgb[2 * d:], dx.double().sum(0)) < 1e-4       # bias gradient of the dense layer in front


def test_dropout_masks_consistent_and_calibrated():
    """The GEMM epilogue, LayerNorm-backward output mask and LayerNorm-forward mask of the same
    (seed, step, site) must coincide; the keep rate must be 1-p with scale 1/(1-p)."""
    _lib, L = _setup()
    M, d, K = 640, 256, 32
    p = 0.1
    rng = torch.tensor([1234, 5], dtype=torch.int64, device="cuda")
    A = torch.ones(M, K, device="cuda")
    B = torch.ones(d, K, device="cuda") / K
    out = torch.empty(M, d, device="cuda")
    _lib.check(L.pmgt_op_gemm_nt(0, P(A), K, None, P(B), K, P(out), d, M, d, K, None, 0, None, 0, None, 0, p, 77, P(rng), None,
                                 stream()))
    keep = out != 0
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.01
    assert torch.allclose(out[keep], torch.full_like(out[keep], 1 / (1 - p)), rtol=1e-5)
    # LN backward second output uses the same mask
    x = torch.randn(M, d, device="cuda")
    dy = torch.randn(M, d, device="cuda")
    gam = torch.ones(d, device="cuda")
    bet = torch.zeros(d, device="cuda")
    y = torch.empty_like(x)
    stats = torch.empty(M, 2, device="cuda")
    _lib.check(L.pmgt_op_layernorm_fwd(0, P(x), P(y), P(stats), P(gam), P(bet), M, d, 1e-12, p, 77, P(rng), stream()))
    assert bool(((y != 0) == keep).all())
    dx, dxd = torch.empty_like(x), torch.empty_like(x)
    part = torch.empty(((M + 63) // 64) * 3 * d, device="cuda")
    dgb = torch.empty(3 * d, device="cuda")
    _lib.check(L.pmgt_op_layernorm_bwd(0, P(dy), P(x), P(stats), P(gam), P(dx), P(dxd), P(part), P(dgb), M, d, 0.0, 0, p, 77,
                                       P(rng), stream()))
    assert torch.allclose(dxd, dx * keep / (1 - p), rtol=1e-5, atol=1e-7)
    assert rel_err(dgb[2 * d:], dxd.double().sum(0)) < 1e-4
    # a different step gives a different mask
    rng2 = torch.tensor([1234, 6], dtype=torch.int64, device="cuda")
    out2 = torch.empty_like(out)
    _lib.check(L.pmgt_op_gemm_nt(0, P(A), K, None, P(B), K, P(out2), d, M, d, K, None, 0, None, 0, None, 0, p, 77, P(rng2), None,
                                 stream()))
    assert float(((out2 != 0) != keep).float().mean()) > 0.05


# ------------------------------------------------------------------------------------------- attention
def _attn_ref(qkvc, mask, H, beta):
    """Oracle restatement on a packed [T,S,4d] tensor (fp64)."""
    T, S, d4 = qkvc.shape
    d = d4 // 4
    dh = d // H

    def heads(x):
        return x.view(T, S, H, dh).permute(0, 2, 1, 3)

    q, k, v, c = (heads(qkvc[..., i * d:(i + 1) * d]) for i in range(4))
    add = (1.0 - mask)[:, None, None, :] * -10000.0
    rho = torch.linalg.norm(c, dim=-1, keepdim=True)
    s1 = 1.0 - (c @ c.transpose(-1, -2)) / (rho @ rho.transpose(-1, -2)) + torch.eye(S, dtype=qkvc.dtype) + add
    s2 = (q @ k.transpose(-1, -2)) / math.sqrt(dh) + add
    w = beta * torch.softmax(s1, -1) + (1 - beta) * torch.softmax(s2, -1)
    return (w @ v).permute(0, 2, 1, 3).reshape(T, S, d), w


def _to_head_major(t, H, dh):
    """[..., 4d] in q | k | v | c column order -> head-major (h, matrix, w)."""
    lead = t.shape[:-1]
    return t.reshape(*lead, 4, H, dh).transpose(-3, -2).reshape(*lead, 4 * H * dh).contiguous()


def _from_head_major(t, H, dh):
    lead = t.shape[:-1]
    return t.reshape(*lead, H, 4, dh).transpose(-3, -2).reshape(*lead, 4 * H * dh).contiguous()


@pytest.mark.parametrize("T,H,hm,p", [(7, 8, 1, 0.0), (64, 8, 1, 0.0), (9, 4, 1, 0.0), (6, 8, 0, 0.0), (700, 8, 1, 0.0), (23, 8, 1, 0.2),
                                      (2, 4, 0, 0.2)])
def test_attention_backward_fused_with_qkvc_weight_gradient(T, H, hm, p):
    """attn_bwd_wgrad_kernel (attention waves + GEMM waves in one workgroup): dQ|dK|dV|dC against autograd through the fp64
    restatement of pmgt/pmgt/modeling_pmgt.py:435-526 (p = 0) or against the unfused MFMA backward with the same dropout masks
    (p > 0); dW / db partials against dQKVC^T x and column sums of the bf16 dQKVC the kernel itself stored.  Odd sequence counts
    (half-empty last step), more steps than one per workgroup (T = 700), both column layouts, both hidden sizes."""
    _lib, L = _setup()
    S, dh, beta = 32, 32, 0.5
    d = H * dh
    g = torch.Generator().manual_seed(T * 10 + H)
    qkvc = torch.randn(T, S, 4 * d, generator=g)
    dctx = torch.randn(T, S, d, generator=g)
    xin = torch.randn(T, S, d, generator=g)
    mask = torch.ones(T, S)
    for t in range(T):
        mask[t, 1 + (t * 7) % S:] = 0
    mask[0] = 1
    qd, dod, xd, md = to_dev(qkvc, torch.bfloat16), to_dev(dctx, torch.bfloat16), to_dev(xin, torch.bfloat16), mask.cuda()
    q_in = _to_head_major(qd, H, dh) if hm else qd
    parts = L.pmgt_op_attention_bwd_wgrad_parts(H)
    rng = torch.tensor([99, 3], dtype=torch.int64, device="cuda")
    dx = torch.full((T, S, 4 * d), float("nan"), device="cuda", dtype=torch.bfloat16)
    slab = torch.full((parts, 4 * d, d), float("nan"), device="cuda")
    bslab = torch.full((parts, 4 * d), float("nan"), device="cuda")
    _lib.check(L.pmgt_op_attention_bwd_wgrad(P(q_in), P(md), P(dod), P(xd), P(dx), P(slab), P(bslab), T, H, beta, p, 11, 12, P(rng), hm,
                                             stream()))
    torch.cuda.synchronize()
    got = _from_head_major(dx, H, dh) if hm else dx
    assert torch.isfinite(got.float()).all() and torch.isfinite(slab).all() and torch.isfinite(bslab).all()
    if p == 0.0:
        xr = rounded(qkvc, torch.bfloat16).requires_grad_(True)
        ref, _ = _attn_ref(xr, mask.double(), H, beta)
        ref.backward(rounded(dctx, torch.bfloat16))
        assert rel_err(got, xr.grad) < 2e-2
    # the unfused one-wave MFMA backward on the same inputs (same dropout masks: same (seed, step, site, row, column) hash)
    dx2 = torch.empty_like(dx)
    L.use("wave_attention_bwd")
    _lib.check(L.pmgt_op_attention_bwd(1, P(qd), P(md), P(dod), P(dx2), T, S, H, dh, beta, p, 11, 12, P(rng), stream()))
    L.use()
    assert rel_err(got, dx2) < 2e-2
    assert ((got.float() == 0) == (dx2.float() == 0)).float().mean() > 0.995
    # weight / bias gradient: exactly the sums of what the kernel stored (fp32 accumulation of bf16 products), q | k | v | c rows
    G = got.double().reshape(T * S, 4 * d).cpu()
    X = xd.double().reshape(T * S, d).cpu()
    dW, db = slab.double().sum(0).cpu(), bslab.double().sum(0).cpu()
    assert rel_err(dW, G.T @ X) < 1e-4
    assert float((db - G.sum(0)).abs().max()) < 1e-4 * float(G.abs().sum(0).max())


@pytest.mark.parametrize("dt", ["fp32", "bf16", "bf16-wave"])     # bf16-wave: the one-wave-per-(sequence, head) backward
@pytest.mark.parametrize("T,S,H,dh,beta", [(5, 16, 4, 16, 0.5), (7, 32, 8, 32, 0.5), (3, 6, 2, 64, 0.3), (2, 64, 2, 64, 0.5),
                                           (4, 32, 1, 128, 1.0), (6, 20, 4, 32, 0.0), (9, 33, 2, 32, 0.7), (5, 48, 3, 64, 0.5),
                                           (11, 64, 8, 32, 0.4), (3, 17, 2, 64, 0.6)])
def test_attention_fwd_bwd(dt, T, S, H, dh, beta):
    _lib, L = _setup()
    L.use(*(["wave_attention_bwd"] if dt == "bf16-wave" else []))
    dt = dt.split("-")[0]
    code, tdt = DT[dt]
    d = H * dh
    g = torch.Generator().manual_seed(S * 100 + dh)
    x = torch.randn(T, S, 4 * d, generator=g)
    mask = torch.ones(T, S)
    for t in range(T):            # ragged valid lengths, position 0 always valid
        n = 1 + (t * 5) % S
        mask[t, n:] = 0
    mask[0] = 1
    dctx = torch.randn(T, S, d, generator=g)
    xd, md, dod = to_dev(x, tdt), mask.cuda(), to_dev(dctx, tdt)
    ctx = torch.empty(T, S, d, device="cuda", dtype=tdt)
    probs = torch.empty(T, H, S, S, device="cuda")
    _lib.check(L.pmgt_op_attention_fwd(code, P(xd), P(md), P(ctx), P(probs), T, S, H, dh, beta, 0.0, 0, 0, None, stream()))
    xr = rounded(x, tdt).requires_grad_(True)
    ref, w = _attn_ref(xr, mask.double(), H, beta)
    assert rel_err(probs, w.detach()) < (1e-5 if dt == "fp32" else 2e-2)
    assert rel_err(ctx, ref.detach()) < tol(dt)
    ref.backward(rounded(dctx, tdt))
    dx = torch.empty(T, S, 4 * d, device="cuda", dtype=tdt)
    _lib.check(L.pmgt_op_attention_bwd(code, P(xd), P(md), P(dod), P(dx), T, S, H, dh, beta, 0.0, 0, 0, None, stream()))
    e = rel_err(dx, xr.grad)
    assert e < (2e-4 if dt == "fp32" else 3e-2), e
    # without a mask pointer == all ones
    _lib.check(L.pmgt_op_attention_fwd(code, P(xd), None, P(ctx), None, T, S, H, dh, beta, 0.0, 0, 0, None, stream()))
    ref1, _ = _attn_ref(rounded(x, tdt), torch.ones(T, S, dtype=torch.float64), H, beta)
    assert rel_err(ctx, ref1) < tol(dt)


def test_attention_dropout_forward_backward_consistent():
    """With dropout on, the backward must regenerate the forward masks: check dV against
    P_dropped^T dO using the probabilities the forward reports."""
    _lib, L = _setup()
    T, S, H, dh, beta, p = 4, 32, 4, 32, 0.5, 0.2
    d = H * dh
    x = torch.randn(T, S, 4 * d).cuda()
    dctx = torch.randn(T, S, d).cuda()
    rng = torch.tensor([99, 3], dtype=torch.int64, device="cuda")
    ctx = torch.empty(T, S, d, device="cuda")
    probs = torch.empty(T, H, S, S, device="cuda")
    _lib.check(L.pmgt_op_attention_fwd(0, P(x), None, P(ctx), P(probs), T, S, H, dh, beta, p, 11, 12, P(rng), stream()))
    dx = torch.empty(T, S, 4 * d, device="cuda")
    _lib.check(L.pmgt_op_attention_bwd(0, P(x), None, P(dctx), P(dx), T, S, H, dh, beta, p, 11, 12, P(rng), stream()))
    v = x[..., 2 * d:3 * d].view(T, S, H, dh).permute(0, 2, 1, 3)
    do = dctx.view(T, S, H, dh).permute(0, 2, 1, 3)
    assert rel_err(ctx, (probs @ v).permute(0, 2, 1, 3).reshape(T, S, d)) < 1e-5
    dv_ref = (probs.transpose(-1, -2) @ do).permute(0, 2, 1, 3).reshape(T, S, d)
    assert rel_err(dx[..., 2 * d:3 * d], dv_ref) < 1e-5
    zero_frac = float((probs == 0).float().mean())
    assert 0.01 < zero_frac < 0.1        # both branches dropped together: ~p^2


def test_attention_tile_forms_match_the_cooperative_forms():
    """S = 64 / head size 64 (the C4 / C5 shapes): the tile form (rows staged once per workgroup as LDS tiles, raw C with the
    inverse norms folded into dS1) against the cooperative form (per-wave fragment loads, normalised C tile) on the same inputs,
    dropout and a ragged mask on; and both against the fp32 engine kernel."""
    _lib, L = _setup()
    T, S, H, dh, beta, p = 7, 64, 3, 64, 0.4, 0.15
    d = H * dh
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(T, S, 4 * d, generator=g) * 0.7).cuda()
    dctx = torch.randn(T, S, d, generator=g).cuda()
    mask = torch.ones(T, S)
    for t in range(1, T):
        mask[t, 1 + (t * 11) % S:] = 0
    md = mask.cuda()
    rng = torch.tensor([7, 5], dtype=torch.int64, device="cuda")
    xb, db = x.bfloat16(), dctx.bfloat16()
    outs = {}
    for name, opts in (("tile", []), ("coop", ["no_tile_attention"])):
        L.use(*opts)
        dx = torch.full((T, S, 4 * d), float("nan"), device="cuda", dtype=torch.bfloat16)
        _lib.check(L.pmgt_op_attention_bwd(1, P(xb), P(md), P(db), P(dx), T, S, H, dh, beta, p, 11, 12, P(rng), stream()))
        assert torch.isfinite(dx.float()).all()
        outs[name] = dx.float()
    L.use()
    assert rel_err(outs["tile"], outs["coop"]) < 1e-2
    ref = torch.empty(T, S, 4 * d, device="cuda")
    _lib.check(L.pmgt_op_attention_bwd(0, P(xb.float()), P(md), P(db.float()), P(ref), T, S, H, dh, beta, p, 11, 12, P(rng), stream()))
    for k in outs:
        for m in range(4):          # dQ, dK, dV, dC blocks separately
            assert rel_err(outs[k][..., m * d:(m + 1) * d], ref[..., m * d:(m + 1) * d]) < 3e-2, (k, m)
    # forward: tile form against the cooperative form and the fp32 kernel, probabilities included
    fo = {}
    for name, opts in (("tile", []), ("coop", ["no_tile_attention"])):
        L.use(*opts)
        ctx = torch.full((T, S, d), float("nan"), device="cuda", dtype=torch.bfloat16)
        probs = torch.full((T, H, S, S), float("nan"), device="cuda")
        _lib.check(L.pmgt_op_attention_fwd(1, P(xb), P(md), P(ctx), P(probs), T, S, H, dh, beta, p, 11, 12, P(rng), stream()))
        fo[name] = (ctx.float(), probs)
    L.use()
    assert rel_err(fo["tile"][0], fo["coop"][0]) < 1e-2 and rel_err(fo["tile"][1], fo["coop"][1]) < 1e-2
    ctx32 = torch.empty(T, S, d, device="cuda")
    probs32 = torch.empty(T, H, S, S, device="cuda")
    _lib.check(L.pmgt_op_attention_fwd(0, P(xb.float()), P(md), P(ctx32), P(probs32), T, S, H, dh, beta, p, 11, 12, P(rng), stream()))
    for k in fo:
        assert rel_err(fo[k][0], ctx32) < 2e-2 and rel_err(fo[k][1], probs32) < 2e-2, k


@pytest.mark.parametrize("S,H,dh", [(32, 4, 32), (48, 2, 64), (64, 2, 32), (64, 2, 64)])
def test_attention_mfma_dropout_consistent_and_matches_valu(S, H, dh):
    """bf16 MFMA attention: (a) with dropout on, forward output and dV are consistent with the reported
    (dropped) probabilities, i.e. backward regenerates the forward masks; (b) it agrees with the generic
    VALU kernel on the same bf16 inputs."""
    _lib, L = _setup()
    T, beta, p = 5, 0.5, 0.2
    d = H * dh
    g = torch.Generator().manual_seed(S)
    x = torch.randn(T, S, 4 * d, generator=g).cuda().bfloat16()
    dctx = torch.randn(T, S, d, generator=g).cuda().bfloat16()
    mask = torch.ones(T, S)
    mask[1, S // 2:] = 0
    mask = mask.cuda()
    rng = torch.tensor([99, 3], dtype=torch.int64, device="cuda")
    outs = {}
    for force in (0, 1):
        L.use(*(['valu_attention'] if force else []))
        ctx = torch.empty(T, S, d, device="cuda", dtype=torch.bfloat16)
        probs = torch.empty(T, H, S, S, device="cuda")
        dx = torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16)
        _lib.check(L.pmgt_op_attention_fwd(1, P(x), P(mask), P(ctx), P(probs), T, S, H, dh, beta, p, 11, 12, P(rng), stream()))
        _lib.check(L.pmgt_op_attention_bwd(1, P(x), P(mask), P(dctx), P(dx), T, S, H, dh, beta, p, 11, 12, P(rng), stream()))
        outs[force] = (ctx.float(), probs, dx.float())
    ctx, probs, dx = outs[0]
    v = x.float()[..., 2 * d:3 * d].view(T, S, H, dh).permute(0, 2, 1, 3)
    do = dctx.float().view(T, S, H, dh).permute(0, 2, 1, 3)
    assert rel_err(ctx, (probs @ v).permute(0, 2, 1, 3).reshape(T, S, d)) < 2e-2
    assert rel_err(dx[..., 2 * d:3 * d], (probs.transpose(-1, -2) @ do).permute(0, 2, 1, 3).reshape(T, S, d)) < 2e-2
    assert rel_err(probs, outs[1][1]) < 2e-2          # same masks, same probabilities
    assert ((probs == 0) == (outs[1][1] == 0)).float().mean() > 0.999
    assert rel_err(ctx, outs[1][0]) < 3e-2
    assert rel_err(dx, outs[1][2]) < 4e-2


@pytest.mark.parametrize("M,drop", [(8192, 0.1), (8193, 0.0), (9000, 0.1), (65537, 0.1), (12288, 0.0)])
def test_role_split_layernorm_gemm_is_bit_identical_to_the_streaming_kernel(M, drop):
    """gemm_wsr.hip (sixteen waves per CU: eight keep W and run the MFMAs, eight run the bias / dropout / residual / LayerNorm
    epilogue of the previous 32-row tile) against gemm_ws_kernel<8, WS_RES_LN>: the same arithmetic per element in the same
    order, so the sum, the LayerNorm output and the statistics must agree BIT FOR BIT; ragged M, one step and many steps per
    workgroup, dropout on and off; and against fp64 torch."""
    _lib, L = _setup()
    N = K = 256
    g = torch.Generator().manual_seed(M)
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    W = (torch.randn(N, K, generator=g) * 0.2).cuda().bfloat16()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda().bfloat16()
    gam = (1 + 0.1 * torch.randn(N, generator=g)).cuda()
    bet = (0.1 * torch.randn(N, generator=g)).cuda()
    rng = torch.tensor([5, 9], dtype=torch.int64, device="cuda")
    outs = []
    for opts in ((), ("no_role_split_ln",)):
        L.use(*opts)
        Cd = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        lno = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        stats = torch.full((M, 2), float("nan"), device="cuda")
        _lib.check(L.pmgt_op_linear(1, P(A), K, P(W), K, P(Cd), N, M, N, K, P(bias), 0, None, N, P(R), N, drop, 33, P(rng),
                                    P(lno), P(stats), P(gam), P(bet), 1e-12, stream()))
        torch.cuda.synchronize()
        outs.append((Cd, lno, stats))
    L.use()
    for a, b in zip(outs[0], outs[1]):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, b)
    if drop == 0.0:
        x = (A.double().cpu() @ W.double().cpu().T + bias.double().cpu() + R.double().cpu()).to(torch.bfloat16).double()
        ref = torch.nn.functional.layer_norm(x, (N,), gam.double().cpu(), bet.double().cpu(), 1e-12)
        assert rel_err(outs[0][1], ref) < 2e-2


@pytest.mark.parametrize("M,N,epi,res,drop", [(9001, 512, 0, False, 0.0), (8192, 2048, 0, False, 0.0), (20000, 512, 1, False, 0.0),
                                              (12345, 512, 2, False, 0.0), (9001, 512, 0, True, 0.1), (16384, 1024, 0, True, 0.0)])
def test_role_split_k512_gemm_is_bit_identical_to_the_lockstep_kernel(M, N, epi, res, drop):
    """K = 512 (the d = 512 shapes) from 8 192 rows on: the 12-wave role-split kernel (gemm_wsr512_kernel: 8 GEMM waves + 4 epilogue
    waves, 32-row tiles) against gemm_ws_kernel<16, MODE> (option no_role_split_ln) -- same arithmetic per element in the same
    order, so the outputs must be EQUAL, ragged last tile and dropout masks included; and against torch for the plain mode."""
    _lib, L = _setup()
    K = 512
    g = torch.Generator().manual_seed(M + N + epi)
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    W = (torch.randn(N, K, generator=g) * 0.1).cuda().bfloat16()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda().bfloat16() if res else None
    aux_in = torch.randn(M, N, generator=g).cuda().bfloat16()
    rng = torch.tensor([5, 9], dtype=torch.int64, device="cuda")
    outs = []
    for opts in ((), ("no_role_split_ln",)):
        L.use(*opts)
        Cd = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        aux = aux_in.clone() if epi == 2 else torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        _lib.check(L.pmgt_op_linear(1, P(A), K, P(W), K, P(Cd), N, M, N, K, P(bias), epi, P(aux) if epi else None, N,
                                    P(R), N, drop, 33, P(rng), None, None, None, None, 1e-12, stream()))
        torch.cuda.synchronize()
        outs.append((Cd, aux))
    L.use()
    assert torch.isfinite(outs[0][0].float()).all()
    assert torch.equal(outs[0][0], outs[1][0])
    if epi == 1:
        assert torch.isfinite(outs[0][1].float()).all() and torch.equal(outs[0][1], outs[1][1])
    if epi == 0 and drop == 0.0:
        ref = A.double().cpu() @ W.double().cpu().T + bias.double().cpu() + (R.double().cpu() if res else 0.0)
        assert rel_err(outs[0][0], ref) < 1e-2


@pytest.mark.parametrize("M,N,K,epi,res,drop,ln", [
    (4096, 256, 256, 0, True, 0.1, True),      # attn-out / FFN2 shape: dropout + residual + fused LayerNorm
    (1000, 256, 256, 1, False, 0.0, False),    # FFN1: GELU + pre-activation store, ragged M
    (777, 256, 256, 2, True, 0.0, False),      # dgrad with GELU' and residual
    (3000, 1024, 256, 0, False, 0.0, False),   # QKVC: four column slabs
    (640, 128, 128, 0, True, 0.0, True),       # d = 128: streaming kernel, LayerNorm unfused
    (500, 64, 64, 1, False, 0.0, False),
    (100, 264, 256, 0, False, 0.0, False),     # N not a multiple of the slab
    (5000, 512, 512, 0, True, 0.1, True),      # d = 512 (C4 shapes): full-row tile with the LayerNorm in its epilogue (gemm_rowln.hip), ragged M
    (4097, 512, 512, 0, True, 0.0, True),      # ... one row in the last tile, no dropout: also against fp64
    (4100, 512, 512, 0, False, 0.0, True),     # ... no residual
    (8192, 512, 2048, 0, True, 0.1, True),     # ... FFN2 with I = 4 d (K = 2048)
    (3000, 512, 512, 0, True, 0.1, True),      # below the full-row form's size: K = 512 streaming form, LayerNorm unfused
    (3001, 512, 512, 1, False, 0.0, False),    # ... GELU
    (2000, 512, 512, 2, False, 0.0, False),    # ... GELU'
    (4096, 2048, 512, 0, False, 0.0, False),   # ... eight column slabs
    (3001, 256, 512, 0, True, 0.1, True),      # FFN2 with I = 2 d: K = 512 form WITH the fused LayerNorm (N = 256), ragged M
    (2048, 256, 512, 0, True, 0.0, False),     # ... residual only
    (24577, 256, 1024, 0, True, 0.1, True),    # FFN2 with I = 4 d at d = 256 (K = 1024): the 256 x 256 tile with the residual + LayerNorm epilogue (nt_lnf), one row in the last tile
    (24576, 256, 1024, 0, True, 0.0, True),    # ... no dropout: also against fp64
    (24000, 256, 1024, 0, True, 0.0, True),    # ... below 96 tiles: tile GEMM + LayerNorm launch
])
def test_linear_streaming_kernel_matches_tiled_and_reference(M, N, K, epi, res, drop, ln):
    """bf16 weight-stationary streaming GEMM (gemm_ws.hip) == tiled kernel (same dropout masks) == torch."""
    _lib, L = _setup()
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    W = (torch.randn(N, K, generator=g) * 0.2).cuda().bfloat16()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).cuda().bfloat16() if res else None
    gam = (1 + 0.1 * torch.randn(N, generator=g)).cuda()
    bet = (0.1 * torch.randn(N, generator=g)).cuda()
    aux_in = torch.randn(M, N, generator=g).cuda().bfloat16()
    rng = torch.tensor([5, 9], dtype=torch.int64, device="cuda")
    outs = []
    H = _lib.hip()
    for force in (0, 1):
        L.use(*(['tile_gemm'] if force else []))
        Cd = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        aux = aux_in.clone() if epi == 2 else torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        lno = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if ln else None
        stats = torch.zeros(M, 2, device="cuda") if ln else None
        H.pmgt_launch_trace_reset()
        _lib.check(L.pmgt_op_linear(1, P(A), K, P(W), K, P(Cd), N, M, N, K, P(bias), epi, P(aux) if epi else None, N,
                                    P(R), N, drop, 33, P(rng), P(lno), P(stats), P(gam) if ln else None, P(bet) if ln else None,
                                    1e-12, stream()))
        # the full-row form runs exactly where the docstring of gemm_rowln.hip says (N = 512 with a LayerNorm behind it, M >= 4096)
        assert H.pmgt_launch_trace_count(b"gemm_rowln") == (1 if (not force and ln and N == 512 and M >= 4096) else 0)
        assert H.pmgt_launch_trace_count(b"nt_lnf") == (1 if (not force and ln and N == 256 and K > 512 and M >= 96 * 256 - 255) else 0)
        outs.append((Cd.float(), aux.float(), None if lno is None else lno.float(), stats))
    ws, tile = outs
    if ln and N == 512 and M >= 4096:      # the full-row tile combines its row statistics across waves in a fixed order: run to run identical
        L.use()
        Cd2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        lno2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        stats2 = torch.zeros(M, 2, device="cuda")
        _lib.check(L.pmgt_op_linear(1, P(A), K, P(W), K, P(Cd2), N, M, N, K, P(bias), epi, None, N, P(R), N, drop, 33, P(rng), P(lno2), P(stats2),
                                    P(gam), P(bet), 1e-12, stream()))
        assert torch.equal(Cd2.float(), ws[0]) and torch.equal(lno2.float(), ws[2]) and torch.equal(stats2, ws[3])
    assert rel_err(ws[0], tile[0]) < 1e-2
    if epi == 1:
        assert rel_err(ws[1], tile[1]) < 1e-2
    if ln:
        assert rel_err(ws[2], tile[2]) < 2e-2
        assert rel_err(ws[3], tile[3]) < 1e-3
    if drop > 0:
        assert bool(((ws[0] == 0) == (tile[0] == 0)).float().mean() > 0.999)     # identical masks
    if drop == 0:
        pre = A.double() @ W.double().T + bias.double()
        if epi == 1:
            ref = torch.nn.functional.gelu(pre.float().bfloat16().double())
        elif epi == 2:
            x = aux_in.double().requires_grad_(True)
            torch.nn.functional.gelu(x).sum().backward()
            ref = pre * x.grad
        else:
            ref = pre
        if res:
            ref = ref + R.double()
        assert rel_err(ws[0], ref) < 2e-2
        if ln:
            lref = torch.nn.functional.layer_norm(ws[0].double(), (N,), gam.double(), bet.double(), 1e-12)
            assert rel_err(ws[2], lref) < 2e-2


# ------------------------------------------------------------------------------------------- fused projection + attention
@pytest.mark.parametrize("T,H,beta,drop", [(7, 8, 0.5, 0.0), (64, 8, 0.3, 0.0), (5, 4, 1.0, 0.0), (2, 8, 0.0, 0.0),
                                           (33, 8, 0.5, 0.1), (130, 4, 0.5, 0.1)])
def test_fused_qkvc_attention_matches_the_two_kernel_path(T, H, beta, drop):
    """qkvc_attn.hip (one kernel: x W^T + b -> Q|K|V|C -> dual-softmax attention) against the unfused pair
    (streaming GEMM, then the MFMA attention kernel) on the same inputs, and both against fp64 torch.  The two HIP
    paths round at the same points (bf16 Q|K|V|C, fp32 softmax), so they must agree to bf16 round-off of ctx;
    with dropout on they must also draw the same masks (same counter-based keys)."""
    _lib, L = _setup()
    S, dh = 32, 32
    d = H * dh
    g = torch.Generator().manual_seed(T * 10 + H)
    x = torch.randn(T, S, d, generator=g)
    W = torch.randn(4 * d, d, generator=g) / math.sqrt(d)
    bias = torch.randn(4 * d, generator=g) * 0.1
    mask = torch.ones(T, S)
    for t in range(T):
        mask[t, 1 + (t * 7) % S:] = 0
    mask[0] = 1
    xd, Wd, bd, md = to_dev(x, torch.bfloat16), to_dev(W, torch.bfloat16), bias.cuda(), mask.cuda()
    rng = torch.tensor([1234, 5], dtype=torch.int64, device="cuda")
    q1 = torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16)
    c1 = torch.empty(T, S, d, device="cuda", dtype=torch.bfloat16)
    _lib.check(L.pmgt_op_qkvc_attention_fwd(P(xd), P(Wd), P(bd), P(md), P(q1), P(c1), T, S, H, dh, beta, drop, 17, 18, P(rng), stream()))
    q2 = torch.empty_like(q1)
    c2 = torch.empty_like(c1)
    M = T * S
    _lib.check(L.pmgt_op_linear(1, P(xd), d, P(Wd), d, P(q2), 4 * d, M, 4 * d, d, P(bd), 0, None, 0, None, 0, 0.0, 0, None,
                                None, None, None, None, 1e-12, stream()))
    # (the attention kernel runs on the fused kernel's own Q|K|V|C, so that the two contexts compare on identical inputs)
    _lib.check(L.pmgt_op_attention_fwd(1, P(q1), P(md), P(c2), None, T, S, H, dh, beta, drop, 17, 18, P(rng), stream()))
    # Same products; the fused kernel accumulates 16 k per MFMA (32x32x16) from the bias upwards, the streaming GEMM 32 k per MFMA with
    # the bias added last: fp32 round-off apart before the bf16 rounding, i.e. equal except for single-ulp flips on a few per cent of elements
    dq = (q1.float() - q2.float()).abs()
    assert (dq <= 2 ** -7 * q2.float().abs().clamp_min(2 ** -6)).all() and (dq > 0).float().mean().item() < 0.05
    # The fused kernel evaluates the softmaxes in the log2 domain and folds normalisation, beta and the dropout scale into
    # one factor, so the two paths differ by fp32 round-off BEFORE P is rounded to bf16: single-ulp flips of P and ctx
    # (bf16 eps = 3.9e-3) are expected; a different dropout mask or a wrong score would be orders of magnitude larger.
    assert rel_err(c1, c2) < 3e-3 and (c1.float() - c2.float()).abs().max().item() <= 2 ** -6 * c2.float().abs().max().item()
    if drop == 0.0:
        qr = rounded(x, torch.bfloat16).reshape(M, d) @ rounded(W, torch.bfloat16).T + bias.double()
        assert rel_err(q1.reshape(M, 4 * d), qr) < 4e-3
        ref, _ = _attn_ref(q1.float().cpu().double(), mask.double(), H, beta)
        assert rel_err(c1, ref) < tol("bf16")
    # no mask pointer == all ones
    _lib.check(L.pmgt_op_qkvc_attention_fwd(P(xd), P(Wd), P(bd), None, P(q1), P(c1), T, S, H, dh, beta, 0.0, 0, 0, None, stream()))
    ref1, _ = _attn_ref(q1.float().cpu().double(), torch.ones(T, S, dtype=torch.float64), H, beta)
    assert rel_err(c1, ref1) < tol("bf16")
    # unsupported shapes are refused, not silently mis-computed
    assert L.pmgt_op_qkvc_attention_fwd(P(xd), P(Wd), P(bd), None, P(q1), P(c1), T, 16, H, dh, beta, 0.0, 0, 0, None, stream()) == -3


def test_fused_qkvc_attention_fully_masked_sequence():
    """A sequence whose mask is all zero: every key carries -10000, which cancels in both softmaxes (the reference
    gives softmax(scores)).  The fused kernel shifts the mask term by its maximum over the keys and skips the row
    maximum of the bounded cosine branch -- this is the case that shift exists for."""
    _lib, L = _setup()
    T, S, H, dh = 6, 32, 8, 32
    d = H * dh
    g = torch.Generator().manual_seed(99)
    x = torch.randn(T, S, d, generator=g)
    W = torch.randn(4 * d, d, generator=g) / math.sqrt(d)
    mask = torch.ones(T, S)
    mask[1] = 0                       # fully masked
    mask[2, 5:] = 0
    mask[4, 1:] = 0                   # only the target node is valid
    xd, Wd, md = to_dev(x, torch.bfloat16), to_dev(W, torch.bfloat16), mask.cuda()
    q1 = torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16)
    c1 = torch.empty(T, S, d, device="cuda", dtype=torch.bfloat16)
    _lib.check(L.pmgt_op_qkvc_attention_fwd(P(xd), P(Wd), None, P(md), P(q1), P(c1), T, S, H, dh, 0.5, 0.0, 0, 0, None, stream()))
    ref, _ = _attn_ref(q1.float().cpu().double(), mask.double(), H, 0.5)
    assert torch.isfinite(c1.float()).all()
    for t in range(T):
        assert rel_err(c1[t], ref[t]) < tol("bf16"), t


# ------------------------------------------------------------------------------------------- 256 x 256 NT tile
@pytest.mark.parametrize("M,N,K", [(24577, 256, 1024), (12300, 512, 160), (24600, 256, 128), (12300, 384, 96), (24600, 128, 256),   # N % 256 != 0: the 256 x 128 variant
                                   (16640, 512, 512), (20000, 768, 256),
                                   (4096, 256, 1024), (6000, 128, 256)])       # < 96 tiles: stays on the 128 x 128 kernel (same checks)
def test_gemm_nt_big_tile_epilogues(M, N, K):
    """The 256 x 256 LDS-DMA tile (bf16, >= 96 tiles of 256 rows, N % 128 == 0, K % 32 == 0) against fp64 torch for every
    epilogue it implements, and against the 128 x 128 kernel for the dropout mask (same counter-based keys)."""
    _lib, L = _setup()
    tdt = torch.bfloat16
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) * 0.3
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    Ad, Bd, bd, rd = to_dev(A, tdt), to_dev(B, tdt), bias.cuda(), to_dev(res, tdt)
    Cd = torch.full((M, N), float("nan"), device="cuda", dtype=tdt)
    aux = torch.zeros(M, N, device="cuda", dtype=tdt)
    base = rounded(A, tdt) @ rounded(B, tdt).T
    # bias + residual
    _lib.check(L.pmgt_op_gemm_nt(1, P(Ad), K, None, P(Bd), K, P(Cd), N, M, N, K, P(bd), 0, None, 0, P(rd), N, 0.0, 0, None, None, stream()))
    assert rel_err(Cd, base + bias.double() + rounded(res, tdt)) < tol("bf16")
    # bias + GELU (stores the pre-activation)
    _lib.check(L.pmgt_op_gemm_nt(1, P(Ad), K, None, P(Bd), K, P(Cd), N, M, N, K, P(bd), 1, P(aux), N, None, 0, 0.0, 0, None, None, stream()))
    pre = base + bias.double()
    assert rel_err(aux, pre) < tol("bf16")
    assert rel_err(Cd, torch.nn.functional.gelu(rounded(pre.float(), tdt))) < tol("bf16")
    # GELU' of the stored pre-activation
    _lib.check(L.pmgt_op_gemm_nt(1, P(Ad), K, None, P(Bd), K, P(Cd), N, M, N, K, None, 2, P(aux), N, None, 0, 0.0, 0, None, None, stream()))
    x = aux.double().cpu().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert rel_err(Cd, base * x.grad) < tol("bf16")
    # dropout: identical mask and values as the small-tile kernel
    rng = torch.tensor([99, 3], dtype=torch.int64, device="cuda")
    _lib.check(L.pmgt_op_gemm_nt(1, P(Ad), K, None, P(Bd), K, P(Cd), N, M, N, K, P(bd), 0, None, 0, P(rd), N, 0.25, 21, P(rng), None, stream()))
    C2 = torch.empty_like(Cd)
    L.use("tile_gemm")
    try:
        _lib.check(L.pmgt_op_gemm_nt(1, P(Ad), K, None, P(Bd), K, P(C2), N, M, N, K, P(bd), 0, None, 0, P(rd), N, 0.25, 21, P(rng), None, stream()))
    finally:
        L.use()
    assert rel_err(Cd, C2.double()) < 1e-3
    kept = ((Cd.double() - rounded(res, tdt).cuda()).abs() > 1e-6).double().mean().item()
    assert abs(kept - 0.75) < 0.02


@pytest.mark.parametrize("M,K,drop", [(8192, 256, 0.1), (8193, 256, 0.0), (9000, 256, 0.1), (65537, 256, 0.1), (12288, 256, 0.0), (300000, 256, 0.1),
                                      (24577, 1024, 0.1), (24576, 1024, 0.0), (100001, 1024, 0.1), (30000, 512, 0.1), (393216, 1024, 0.1)])
def test_layernorm_backward_in_the_data_gradient_epilogue(M, K, drop):
    """dy = A W^T + residual followed by the backward of a LayerNorm whose OUTPUT y the forward kept (x^ = (y - beta) / gamma):
    gemm_wsr_kernel<true> (K = 256: epilogue role of the streaming kernel) and gemm_nt_big_kernel<256, 8, true> (larger K: a phase
    behind the tile's main loop) do both in one launch, dy never reaches HBM; checked against the two-launch form
    (streaming GEMM -> dy in bf16 -> ln_bwd_kernel) and against fp64 autograd through torch's layer_norm
    (BertSelfOutput / BertOutput backward, pmgt/pmgt/modeling_pmgt.py:293-294,322-325).  Ragged M, one and many steps per workgroup."""
    _lib, L = _setup()
    N = 256
    g = torch.Generator().manual_seed(M)
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).cuda().bfloat16()
    R = torch.randn(M, N, generator=g).cuda().bfloat16()
    x = torch.randn(M, N, generator=g) * 1.5 + 0.3                      # the LayerNorm's input (never given to the kernels)
    gam = (1 + 0.1 * torch.randn(N, generator=g))
    bet = (0.1 * torch.randn(N, generator=g))
    xd = x.double().requires_grad_(True)
    gd, bd = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    yref = torch.nn.functional.layer_norm(xd, (N,), gd, bd, 1e-12)
    y = yref.detach().float().cuda().bfloat16()
    mean = x.double().mean(1)
    rstd = 1.0 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-12)
    stats = torch.stack([mean, rstd], 1).float().cuda().contiguous()
    gamd, betd = gam.cuda(), bet.cuda()
    rng = torch.tensor([5, 9], dtype=torch.int64, device="cuda")
    outs = []
    for opts in ((), ("unfused_ln_bwd",)):
        L.use(*opts)
        dx = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        dxd = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        tmp = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        part = torch.empty(max(256, (M + 63) // 64) * 3 * N, device="cuda")
        dgb = torch.full((3 * N,), float("nan"), device="cuda")
        _lib.check(L.pmgt_op_linear_ln_bwd(P(A), K, P(W), K, M, N, K, P(R), N, P(y), P(stats), P(gamd), P(betd), P(tmp), P(dx), P(dxd),
                                           drop, 35, P(rng), P(part), P(dgb), stream()))
        torch.cuda.synchronize()
        outs.append((dx, dxd, dgb))
    L.use()
    (dx, dxd, dgb), (dx2, dxd2, dgb2) = outs
    assert torch.isfinite(dx.float()).all() and torch.isfinite(dxd.float()).all() and torch.isfinite(dgb).all()
    # fused vs two-launch: the only differences are dy kept in fp32 instead of rounded to bf16 and the order of the row sums
    assert rel_err(dx, dx2.float()) < 1.5e-2
    # (the two-launch form sums bf16-rounded dy: ~2^-9 sqrt(M) |dy| of noise on column sums of size ~sqrt(M) |dy|)
    assert rel_err(dgb[:N], dgb2[:N]) < 1e-2 and rel_err(dgb[N:2 * N], dgb2[N:2 * N]) < 1e-2 and rel_err(dgb[2 * N:], dgb2[2 * N:]) < 1e-2
    keep = dxd2 != 0
    if drop > 0:
        assert abs(float(keep.float().mean()) - (1 - drop)) < 0.01
        assert float(((dxd != 0) == keep).float().mean()) > 0.9999      # the same dropout mask (up to exact zeros of dx)
    # against fp64 autograd: dy = A W^T + R, x^ from the bf16 y (the carrier error of engine.py's guard is part of the tolerance)
    dy = A.double().cpu() @ W.double().cpu().T + R.double().cpu()
    yref.backward(dy)
    assert rel_err(dx, xd.grad) < 3e-2
    # (K > 256: the tile form passes dy to its LayerNorm phase as bf16 -- what the two-launch form rounds it to in HBM)
    assert rel_err(dgb[:N], gd.grad) < 1.5e-2
    assert rel_err(dgb[N:2 * N], bd.grad) < (1e-3 if K == 256 else 5e-3)
    assert rel_err(dgb[2 * N:], dxd.double().sum(0)) < 1e-4               # the dense bias gradient = column sums of what was stored
    if drop > 0:
        scale = 1.0 / (1.0 - drop)
        assert rel_err(dxd, (dx.double().cpu() * scale).to(torch.bfloat16).double() * keep.double().cpu()) < 1e-2
    else:
        assert torch.equal(dx, dxd)


@pytest.mark.parametrize("opts", [(), ("no_tile_attention",)])
def test_attention_tile_forms_match_the_oracle_restatement(opts):
    """S = 64 / head size 64 / H = 8 (the C4 / C5 attention shape, d = 512): forward output, probabilities and dQ | dK | dV | dC of the
    tile form (and of the cooperative form it replaced) against the fp64 autograd restatement of
    pmgt/pmgt/modeling_pmgt.py:420-534 -- not against another kernel of this library; ragged masks, beta != 0.5."""
    _lib, L = _setup()
    T, S, H, dh, beta = 9, 64, 8, 64, 0.3
    d = H * dh
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(T, S, 4 * d, generator=g) * 0.7)
    dctx = torch.randn(T, S, d, generator=g)
    mask = torch.ones(T, S)
    for t in range(1, T):
        mask[t, 1 + (t * 13) % S:] = 0
    xb, db, md = x.cuda().bfloat16(), dctx.cuda().bfloat16(), mask.cuda()
    xr = xb.double().cpu().requires_grad_(True)
    ctx_ref, w_ref = _attn_ref(xr, mask.double(), H, beta)
    ctx_ref.backward(db.double().cpu())
    L.use(*opts)
    ctx = torch.full((T, S, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    probs = torch.full((T, H, S, S), float("nan"), device="cuda")
    _lib.check(L.pmgt_op_attention_fwd(1, P(xb), P(md), P(ctx), P(probs), T, S, H, dh, beta, 0.0, 11, 12, None, stream()))
    dx = torch.full((T, S, 4 * d), float("nan"), device="cuda", dtype=torch.bfloat16)
    _lib.check(L.pmgt_op_attention_bwd(1, P(xb), P(md), P(db), P(dx), T, S, H, dh, beta, 0.0, 11, 12, None, stream()))
    L.use()
    assert rel_err(ctx, ctx_ref.detach()) < 2e-2
    assert rel_err(probs, w_ref.detach()) < 2e-2
    for m in range(4):          # dQ, dK, dV, dC blocks separately (their scales differ)
        assert rel_err(dx[..., m * d:(m + 1) * d], xr.grad[..., m * d:(m + 1) * d]) < 3e-2, m


# ------------------------------------------------------------------------------------------- token order of the table-mode backward
@pytest.mark.parametrize("M,n_rows,skew", [(393216, 7254, True), (393216, 7254, False), (1000, 40, False), (4096, 2048, False), (4097, 3, True),
                                           (73728, 20002, True), (300001, 150000, False), (1572864, 70000, True), (65, 130, False),
                                           # 256 tiles: the folded bases; 257 and more: the column-scan launch (advisor, round 5: O(tiles^2) walks)
                                           (524288, 7254, False), (524289, 7254, True), (600000, 200, True)])
def test_token_sort_by_node_id_is_the_stable_sort(M, n_rows, skew):
    """seg_sort (segsum.hip): the hand-written LSD radix sort that orders a step's tokens by node id for the per-node gradient sums (the
    gather of pmgt/pmgt/utils.py:43-50 in reverse).  Integer work, so bit-exact: sorted keys, permutation and segment offsets equal numpy's
    STABLE argsort / searchsorted -- one, two and three counting passes (n_rows <= 2^8, 2^16, > 2^16), ragged last tiles, heavy
    duplicates (padded tokens carry id 0: the pad segment alone can be a third of the batch), run to run identical."""
    _lib, L = _setup()
    H = _lib.hip()
    rs = np.random.RandomState(M % 1000 + n_rows)
    ids = rs.randint(0, n_rows, size=M).astype(np.int64)
    if skew:
        ids[rs.rand(M) < 0.3] = 0
        ids[rs.rand(M) < 0.05] = 1
    d_ids = torch.from_numpy(ids).cuda()
    mk = lambda n, dt: torch.full((n,), -1, dtype=dt, device="cuda")
    nb = int(H.pmgt_op_seg_sort_temp_bytes(M))
    outs = []
    for rep in range(2):
        k1, v1, sk, pm = mk(M, torch.int32), mk(M, torch.int32), mk(M, torch.int32), mk(M, torch.int32)
        off = mk(n_rows + 1, torch.int32)
        tmp = torch.empty(nb, dtype=torch.uint8, device="cuda")
        _lib.check(H.pmgt_op_seg_sort(P(d_ids), M, n_rows, P(k1), P(v1), P(sk), P(pm), P(off), P(tmp), nb, stream()))
        torch.cuda.synchronize()
        outs.append((sk.cpu().numpy(), pm.cpu().numpy(), off.cpu().numpy()))
    order = np.argsort(ids, kind="stable")
    assert np.array_equal(outs[0][1].astype(np.int64), order)
    assert np.array_equal(outs[0][0].astype(np.int64), ids[order])
    assert np.array_equal(outs[0][2].astype(np.int64), np.searchsorted(ids[order], np.arange(n_rows + 1), side="left"))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------------------------------- beta == 1: the dead dot-product branch skipped
@pytest.mark.parametrize("T,H,hm,p", [(7, 8, 1, 0.0), (64, 8, 0, 0.0), (9, 4, 1, 0.0), (700, 8, 1, 0.1), (2, 4, 0, 0.2), (33, 8, 1, 0.1)])
def test_fused_kernels_at_beta_one_skip_the_dead_branch(T, H, hm, p):
    """beta = 1 (scripts/run_pmgt.sh:24): attention_probs = 1 * A1 + 0 * A2 (pmgt/pmgt/modeling_pmgt.py:519-521) -- Q, K, the dot-product
    softmax and their backward contribute exactly nothing.  The vc_only forms of the two fused kernels (forward: {V, C} x four heads per
    slab, cosine branch alone; backward: Q / K neither fetched nor differentiated, dW_{v,c} only) against
      * the fp64 autograd restatement of modeling_pmgt.py:435-526 at beta = 1 (p = 0),
      * the GENERAL fused kernels at beta = 1 on the same inputs and dropout keys: same context, same dV | dC, same V / C weight-gradient
        partial sums; dQ = dK = 0 there, untouched (NaN-filled) columns here; query / key weight and bias partials exactly 0 in both."""
    _lib, L = _setup()
    S, dh, beta = 32, 32, 1.0
    d = H * dh
    g = torch.Generator().manual_seed(T * 10 + H + 5)
    x = torch.randn(T, S, d, generator=g)
    W = torch.randn(4 * d, d, generator=g) / math.sqrt(d)
    bias = torch.randn(4 * d, generator=g) * 0.1
    dctx = torch.randn(T, S, d, generator=g)
    mask = torch.ones(T, S)
    for t in range(T):
        mask[t, 1 + (t * 7) % S:] = 0
    mask[0] = 1
    xd, Wd, bd, md, dod = to_dev(x, torch.bfloat16), to_dev(W, torch.bfloat16), bias.cuda(), mask.cuda(), to_dev(dctx, torch.bfloat16)
    rng = torch.tensor([77, 4], dtype=torch.int64, device="cuda")
    nan = float("nan")
    res = {}
    for vc in (0, 1):
        qk = torch.full((T, S, 4 * d), nan, device="cuda", dtype=torch.bfloat16)
        cx = torch.full((T, S, d), nan, device="cuda", dtype=torch.bfloat16)
        _lib.check(L.pmgt_op_qkvc_attention_fwd_ex(P(xd), P(Wd), P(bd), P(md), P(qk), P(cx), T, S, H, dh, beta, p, 17, 18, P(rng), hm | (2 * vc),
                                                   stream()))
        parts = L.pmgt_op_attention_bwd_wgrad_parts(H)
        dx = torch.full((T, S, 4 * d), nan, device="cuda", dtype=torch.bfloat16)
        slab = torch.full((parts, 4 * d, d), nan, device="cuda")
        bslab = torch.full((parts, 4 * d), nan, device="cuda")
        _lib.check(L.pmgt_op_attention_bwd_wgrad(P(qk), P(md), P(dod), P(xd), P(dx), P(slab), P(bslab), T, H, beta, p, 17, 18, P(rng),
                                                 hm | (2 * vc), stream()))
        torch.cuda.synchronize()
        q_std = _from_head_major(qk, H, dh) if hm else qk
        dx_std = _from_head_major(dx, H, dh) if hm else dx
        res[vc] = (q_std.float(), cx.float(), dx_std.float(), slab.sum(0), bslab.sum(0))
    gen, vcr = res[0], res[1]
    assert torch.isfinite(gen[0]).all() and torch.isfinite(gen[2]).all()
    # forward: V | C and the context agree; Q | K of the vc form were never written
    assert torch.isnan(vcr[0][..., :2 * d]).all() and torch.isnan(vcr[2][..., :2 * d]).all()
    assert torch.equal(vcr[0][..., 2 * d:], gen[0][..., 2 * d:])
    assert torch.isfinite(vcr[1]).all() and rel_err(vcr[1], gen[1]) < 3e-3
    # backward: dQ = dK = 0 in the general form; dV | dC agree
    assert float(gen[2][..., :2 * d].abs().max()) == 0.0
    assert torch.isfinite(vcr[2][..., 2 * d:]).all() and rel_err(vcr[2][..., 2 * d:], gen[2][..., 2 * d:]) < 1e-2
    # weight / bias partial sums: query / key rows exactly zero, value / ctx_attention rows = (dV | dC)^T x of what the kernel stored
    assert float(vcr[3][:2 * d].abs().max()) == 0.0 and float(vcr[4][:2 * d].abs().max()) == 0.0
    assert float(gen[3][:2 * d].abs().max()) == 0.0
    G = vcr[2][..., 2 * d:].double().reshape(T * S, 2 * d).cpu()
    X = xd.double().reshape(T * S, d).cpu()
    assert rel_err(vcr[3][2 * d:].double().cpu(), G.T @ X) < 1e-4
    assert float((vcr[4][2 * d:].double().cpu() - G.sum(0)).abs().max()) < 1e-4 * float(G.abs().sum(0).max())
    if p == 0.0:
        qr = rounded(x, torch.bfloat16).reshape(T * S, d) @ rounded(W, torch.bfloat16).T + bias.double()
        qin = torch.zeros(T, S, 4 * d, dtype=torch.float64)
        qin[..., 2 * d:] = vcr[0][..., 2 * d:].double().cpu()          # the bf16 V | C the kernel stored; Q | K do not matter at beta = 1
        assert rel_err(vcr[0][..., 2 * d:].reshape(T * S, 2 * d), qr[:, 2 * d:]) < 4e-3
        qin.requires_grad_(True)
        ref, _ = _attn_ref(qin, mask.double(), H, 1.0)
        assert rel_err(vcr[1], ref.detach()) < tol("bf16")
        ref.backward(rounded(dctx, torch.bfloat16))
        assert rel_err(vcr[2][..., 2 * d:], qin.grad[..., 2 * d:]) < 2e-2
        assert float(qin.grad[..., :2 * d].abs().max()) == 0.0            # autograd: exact zeros for Q and K
    # ---- the two-heads-per-step backward (flag bit 2): the same V | C gradients BIT FOR BIT (same per-element arithmetic, other step structure),
    # partial sums [vc2_parts][2 d, d] of the value | ctx_attention rows only
    H_ = _lib.hip()
    parts2 = H_.pmgt_op_attention_bwd_wgrad_vc2_parts(H)
    qk = torch.full((T, S, 4 * d), nan, device="cuda", dtype=torch.bfloat16)
    cx = torch.full((T, S, d), nan, device="cuda", dtype=torch.bfloat16)
    _lib.check(L.pmgt_op_qkvc_attention_fwd_ex(P(xd), P(Wd), P(bd), P(md), P(qk), P(cx), T, S, H, dh, beta, p, 17, 18, P(rng), hm | 2, stream()))
    dx2 = torch.full((T, S, 4 * d), nan, device="cuda", dtype=torch.bfloat16)
    slab2 = torch.full((parts2, 2 * d, d), nan, device="cuda")
    bslab2 = torch.full((parts2, 2 * d), nan, device="cuda")
    H_.pmgt_launch_trace_reset()
    _lib.check(L.pmgt_op_attention_bwd_wgrad(P(qk), P(md), P(dod), P(xd), P(dx2), P(slab2), P(bslab2), T, H, beta, p, 17, 18, P(rng), hm | 2 | 4, stream()))
    torch.cuda.synchronize()
    assert H_.pmgt_launch_trace_count(b"attn_bwd_wgrad_vc2") == 1
    dx2_std = (_from_head_major(dx2, H, dh) if hm else dx2).float()
    assert torch.isnan(dx2_std[..., :2 * d]).all() and torch.equal(dx2_std[..., 2 * d:], vcr[2][..., 2 * d:])
    assert torch.isfinite(slab2).all() and torch.isfinite(bslab2).all()
    assert rel_err(slab2.double().sum(0).cpu(), G.T @ X) < 1e-4
    assert float((bslab2.double().sum(0).cpu() - G.sum(0)).abs().max()) < 1e-4 * float(G.abs().sum(0).max())
    # the vc form refuses anything but beta == 1
    assert L.pmgt_op_qkvc_attention_fwd_ex(P(xd), P(Wd), P(bd), P(md), P(qk), P(cx), T, S, H, dh, 0.5, p, 17, 18, P(rng), hm | 2, stream()) == -3
