"""fp8 mode (BASELINE.json config 5: "fp8 (e4m3) feature-projection + QKV MFMA path") through the C ABI.

The reference has no fp8 path, so the checker is the oracle's restatement of the quantisation contract of
pmgt_amd/csrc/fp8.h (oracle/pmgt_oracle.py: fake_quant_rows / fake_quant_tensor / _LinearQ):
  * quantisation kernels: BIT-exact against torch's OCP float8_e4m3fn conversion (pins the encoding: gfx950 is e4m3fn,
    not the MI300 fnuz format) and its fp32 scale arithmetic;
  * fp8 MFMA GEMMs: against fp64 products of the dequantised operands, to the rounding of their bf16 output;
  * the engine in fp8 mode: losses against the fp8-emulating oracle to bf16-mode tolerances (rtol 2e-2; the fp8 rounding
    itself is part of BOTH sides), gradient cosine > 0.99, and against the fp32 reference goldens to 5e-2."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from oracle import pmgt_oracle as po
from tests import golden_util as gu
from tests.test_engine_gpu import dev_batch, inject_for, make_engine
from tests.test_ops_gpu import P, _attn_ref, _setup, rel_err, stream, to_dev

pytestmark = pytest.mark.gpu


def _q_rows_ref(x32):
    """fp8.h contract on fp32 rows -> (bytes, scale)."""
    amax = x32.abs().amax(dim=-1, keepdim=True)
    ok = amax > 0
    inv = torch.where(ok, torch.tensor(448.0) / amax, torch.ones_like(amax))
    scale = torch.where(ok, amax / torch.tensor(448.0), torch.ones_like(amax))
    q = (x32 * inv).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale.squeeze(-1), q.to(torch.float32)


@pytest.mark.parametrize("src", ["fp32", "bf16"])
@pytest.mark.parametrize("rows,cols", [(5, 256), (64, 1536), (33, 768), (7, 24), (1000, 256)])
def test_quant_rows_is_bit_exact_ocp_e4m3(src, rows, cols):
    _lib, L = _setup()
    g = torch.Generator().manual_seed(rows * 3 + cols)
    x = torch.randn(rows, cols, generator=g) * torch.logspace(-4, 2, rows).unsqueeze(1)      # rows of very different scale
    x[rows // 2] = 0                                                                         # an all-zero row: scale 1, bytes 0
    x[0, :8] = torch.tensor([1e-9, -1e-9, 3e-4, 448.0, -448.0, 0.0, 1.0, -1.0])
    tdt = torch.float32 if src == "fp32" else torch.bfloat16
    xd = to_dev(x, tdt)
    q = torch.full((rows, cols), 0xAB, dtype=torch.uint8, device="cuda")
    sc = torch.zeros(rows, device="cuda")
    _lib.check(L.pmgt_op_quant_rows_e4m3(0 if src == "fp32" else 1, P(xd), cols, rows, cols, P(q), cols, P(sc), stream()))
    qb, sref, _ = _q_rows_ref(xd.float().cpu())
    assert torch.equal(sc.cpu(), sref)
    assert torch.equal(q.cpu(), qb), (q.cpu() != qb).sum()


def test_quantize_tensor_and_back():
    _lib, L = _setup()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(302, 768, generator=g)
    x[:2] = 0
    deq_ref, scale = po.fake_quant_tensor(x)
    xd = x.cuda()
    amax = np.float32(x.abs().max().item())
    q = torch.empty(x.shape, dtype=torch.uint8, device="cuda")
    _lib.check(L.pmgt_quantize_e4m3(P(xd), P(q), x.numel(), float(np.float32(448.0) / amax), stream()))
    back = torch.empty_like(xd)
    _lib.check(L.pmgt_dequantize_e4m3(P(q), P(back), x.numel(), float(amax / np.float32(448.0)), stream()))
    assert abs(scale - float(amax / np.float32(448.0))) == 0.0
    assert torch.equal(back.cpu(), deq_ref)
    # relative error of e4m3 (3 mantissa bits): at most 2^-4 of the value for normal numbers
    big = x.abs() > 0.05 * x.abs().max()
    assert ((back.cpu() - x).abs()[big] / x.abs()[big]).max() <= 2 ** -4


def _rand_e4m3(shape, g, spread=1.0):
    x = torch.randn(*shape, generator=g) * spread
    q = x.clamp(-448, 448).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), q.to(torch.float64)


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (300, 200, 1536), (77, 40, 48), (1000, 256, 768), (513, 1024, 256),
                                   (4100, 512, 512), (8192, 2048, 512), (5000, 256, 192),       # these three: the 256 x 256 LDS-DMA tile (8192 rows, no gather: role split)
                                   (9001, 512, 512), (40000, 2048, 512)])                       # K = 512, M >= 8192, no gather: the role-split streaming kernel (ragged M, many steps)
@pytest.mark.parametrize("gather", [False, True])
def test_gemm_nt_f8(M, N, K, gather):
    """C = (A8 B8^T) sa sb + bias on v_mfma_f32_16x16x32_fp8_fp8.  Random (asymmetric) e4m3 operands: every product is
    exact in fp32, so the error is the bf16 store (2^-8 relative) plus the fp32 accumulation of the MFMA adder tree, which
    truncates inside a 32-term dot product (measured up to 1.3e-5 of sum |a||b|, at K = 48 as at K = 768; bound 4e-5)."""
    _lib, L = _setup()
    g = torch.Generator().manual_seed(M + 3 * N + K)
    R = 57 if gather else M
    A8, Af = _rand_e4m3((R, K), g, 4.0)
    B8, Bf = _rand_e4m3((N, K), g, 0.5)
    rows = torch.randint(0, R, (M,), generator=g) if gather else None
    sb = torch.rand(N, generator=g) + 0.5
    bias = torch.randn(N, generator=g)
    Cd = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    A8d, B8d, sbd, bd = A8.cuda(), B8.cuda(), sb.cuda(), bias.cuda()
    rowd = rows.cuda() if gather else None
    Ag = Af[rows] if gather else Af
    if gather:      # one scale for the table
        _lib.check(L.pmgt_op_gemm_nt_f8(P(A8d), K, P(rowd), None, 0.37, P(B8d), K, P(sbd), P(Cd), N, M, N, K, P(bd), None, stream()))
        ref = (Ag @ Bf.T) * 0.37 * sb.double() + bias.double()
        mag = (Ag.abs() @ Bf.abs().T) * 0.37 * sb.double()
    else:           # one scale per row
        sa = torch.rand(M, generator=g) + 0.25
        sad = sa.cuda()
        _lib.check(L.pmgt_op_gemm_nt_f8(P(A8d), K, None, P(sad), 1.0, P(B8d), K, P(sbd), P(Cd), N, M, N, K, P(bd), None, stream()))
        ref = (Ag @ Bf.T) * sa.double().unsqueeze(1) * sb.double() + bias.double()
        mag = (Ag.abs() @ Bf.abs().T) * sa.double().unsqueeze(1) * sb.double()
    assert torch.isfinite(Cd.float()).all()
    err = (Cd.double().cpu() - ref).abs()
    assert (err <= 1.02 * 2 ** -8 * ref.abs() + 4e-5 * mag).all(), float((err / (ref.abs() + 1e-9)).max())
    if K == 512 and M >= 8192 and N % 256 == 0 and not gather:      # the shape of the d = 512 Q|K|V|C projection: its own kernel must be the one that ran
        H = _lib.hip()
        H.pmgt_launch_trace_reset()
        _lib.check(L.pmgt_op_gemm_nt_f8(P(A8d), K, None, P(sad), 1.0, P(B8d), K, P(sbd), P(Cd), N, M, N, K, P(bd), None, stream()))
        assert H.pmgt_launch_trace_count(b"f8_wsr512") == 1 and H.pmgt_launch_trace_count(b"f8_big") == 0
    # device-side row count: rows beyond it are not written
    if not gather:
        Cd.fill_(7.0)
        cnt = torch.tensor([M // 2 + 3], dtype=torch.int32, device="cuda")
        _lib.check(L.pmgt_op_gemm_nt_f8(P(A8d), K, None, P(sad), 1.0, P(B8d), K, P(sbd), P(Cd), N, M, N, K, P(bd), P(cnt), stream()))
        assert (Cd[M // 2 + 3:] == 7.0).all() and rel_err(Cd[: M // 2 + 3], ref[: M // 2 + 3]) < 6e-3


@pytest.mark.parametrize("M,N1,N2", [(512, 128, 128), (3000, 256, 1536), (777, 64, 768), (100, 256, 136)])
@pytest.mark.parametrize("gather", [False, True])
def test_gemm_tn_with_e4m3_table_operand(M, N1, N2, gather):
    """Feature-projection weight gradient of the fp8 mode: dW = P^T (Q8 * scale), Q8 rows widened to bf16 in flight."""
    _lib, L = _setup()
    g = torch.Generator().manual_seed(M + N1 + N2)
    R = 91 if gather else M
    Pm = torch.randn(M, N1, generator=g)
    Q8, Qf = _rand_e4m3((R, N2), g, 8.0)
    rows = torch.randint(0, R, (M,), generator=g) if gather else None
    Pd, Q8d = to_dev(Pm, torch.bfloat16), Q8.cuda()
    rowd = rows.cuda() if gather else None
    scale = 0.0123
    slab = torch.empty(int(L.pmgt_op_gemm_tn_slab_elems(1, M, N1, N2)), device="cuda")
    out = torch.full((N1, N2), float("nan"), device="cuda")
    _lib.check(L.pmgt_op_gemm_tn_f8(P(Pd), N1, P(Q8d), N2, scale, P(rowd), M, N1, N2, P(slab), P(out), 0, None, stream()))
    Qg = (Qf[rows] if gather else Qf)
    Qb = (Qg.float() * np.float32(scale)).to(torch.bfloat16).double()        # what the kernel feeds the bf16 MFMA
    ref = Pd.double().cpu().T @ Qb
    assert rel_err(out, ref) < 2e-5, rel_err(out, ref)
    # accumulate flag
    _lib.check(L.pmgt_op_gemm_tn_f8(P(Pd), N1, P(Q8d), N2, scale, P(rowd), M, N1, N2, P(slab), P(out), 1, None, stream()))
    assert rel_err(out, 2 * ref) < 2e-5


@pytest.mark.parametrize("T,beta,drop", [(2, 0.5, 0.0), (37, 0.3, 0.0), (300, 0.5, 0.1), (64, 1.0, 0.0)])
def test_fused_qkvc_attention_with_fp8_projection(T, beta, drop):
    """qkvc_attn_fwd2_kernel<8, true>: per-row e4m3 of x inside the kernel, per-channel e4m3 W, fp8 MFMA -> the bf16
    Q|K|V|C the emulation predicts; the attention half against (a) the unfused attention kernel on the SAME Q|K|V|C
    (same dropout keys) and (b) fp64."""
    _lib, L = _setup()
    S, H, dh = 32, 8, 32
    d = H * dh
    M = T * S
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, S, d, generator=g) * (torch.rand(T, S, 1, generator=g) * 3 + 0.1)      # rows of different magnitude
    x[0, 3] = 0                                                                              # an all-zero row
    W = torch.randn(4 * d, d, generator=g) / math.sqrt(d)
    bias = torch.randn(4 * d, generator=g) * 0.1
    mask = torch.ones(T, S)
    for t in range(T):
        mask[t, 1 + (t * 7) % S:] = 0
    mask[0] = 1
    xd, Wd, bd, md = to_dev(x, torch.bfloat16), W.cuda(), bias.cuda(), mask.cuda()
    W8 = torch.empty(4 * d, d, dtype=torch.uint8, device="cuda")
    ws = torch.empty(4 * d, device="cuda")
    _lib.check(L.pmgt_op_quant_rows_e4m3(0, P(Wd), d, 4 * d, d, P(W8), d, P(ws), stream()))
    rng = torch.tensor([1234, 5], dtype=torch.int64, device="cuda")
    q1 = torch.full((T, S, 4 * d), float("nan"), device="cuda", dtype=torch.bfloat16)
    c1 = torch.full((T, S, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    _lib.check(L.pmgt_op_qkvc_attention_fwd_f8(P(xd), None, None, P(W8), P(ws), P(bd), P(md), P(q1), P(c1), T, S, H, dh, beta, drop, 17, 18,
                                               P(rng), stream()))
    assert torch.isfinite(q1.float()).all() and torch.isfinite(c1.float()).all()
    # projection against the emulation (fp64 product of the dequantised operands)
    xq = po.fake_quant_rows(xd.float().cpu().reshape(M, d)).double()
    Wq = po.fake_quant_rows(W).double()
    qr = xq @ Wq.T + bias.double()
    err = (q1.double().cpu().reshape(M, 4 * d) - qr).abs()
    assert (err <= 1.02 * 2 ** -8 * qr.abs() + 4e-5 * (xq.abs() @ Wq.abs().T) + 1e-6).all(), float(err.max())
    # ... and the quantisation error against the unquantised product is the e4m3 level, not something else
    q0 = xd.double().cpu().reshape(M, d) @ W.double().T + bias.double()
    assert 1e-3 < rel_err(q1.reshape(M, 4 * d), q0) < 6e-2
    # the same x through the unfused fp8 pair (quant_rows + gemm_nt_f8) gives the same bf16 Q|K|V|C
    x8 = torch.empty(M, d, dtype=torch.uint8, device="cuda")
    xs = torch.empty(M, device="cuda")
    q2 = torch.empty_like(q1)
    _lib.check(L.pmgt_op_quant_rows_e4m3(1, P(xd), d, M, d, P(x8), d, P(xs), stream()))
    _lib.check(L.pmgt_op_gemm_nt_f8(P(x8), d, None, P(xs), 1.0, P(W8), d, P(ws), P(q2), 4 * d, M, 4 * d, d, P(bd), None, stream()))
    assert rel_err(q1, q2) < 4e-3 and (q1 != q2).float().mean().item() < 0.02      # same products; fp32 sum order differs
    # x handed over already quantised (what the engine does: the producer of x writes the e4m3 rows): bit-identical outputs
    q3, c3 = torch.full_like(q1, float("nan")), torch.full_like(c1, float("nan"))
    _lib.check(L.pmgt_op_qkvc_attention_fwd_f8(None, P(x8), P(xs), P(W8), P(ws), P(bd), P(md), P(q3), P(c3), T, S, H, dh, beta, drop, 17, 18,
                                               P(rng), stream()))
    assert torch.equal(q3, q1) and torch.equal(c3, c1)
    # attention half
    c2 = torch.empty_like(c1)
    _lib.check(L.pmgt_op_attention_fwd(1, P(q1), P(md), P(c2), None, T, S, H, dh, beta, drop, 17, 18, P(rng), stream()))
    assert rel_err(c1, c2) < 3e-3
    if drop == 0.0:
        ref, _ = _attn_ref(q1.float().cpu().double(), mask.double(), H, beta)
        assert rel_err(c1, ref) < 2e-2


def _fp8_case(name):
    case = gu.model_case(name)
    deq, scales = zip(*[po.fake_quant_tensor(t) for t in case["tables"]])
    return case, list(deq), scales


@pytest.mark.parametrize("name", ["m1", "m1_pad", "m2", "m3", "f3", "f1"])      # f3 / f1: three modalities, one
def test_fp8_engine_matches_the_fp8_emulating_oracle(name):
    case, deq_tables, scales = _fp8_case(name)
    gold = case["gold"]
    eng = make_engine(case, dtype="fp8")
    # the e4m3 tables the engine holds ARE the emulation's tables
    for a, b, s_eng, s_ref in zip(eng.dequantized_tables(), deq_tables, eng.table_scale, scales):
        assert s_eng == s_ref and torch.equal(a.cpu(), b)
    cfg8 = dict(case["cfg"], fp8=True)
    inj, inj_cpu = inject_for(case)
    # eval
    out = eng.pretrain_step(dev_batch(case["batch"]), training=False)
    ref = po.pretrain_forward(case["params"], cfg8, deq_tables, case["batch"], training=False)
    np.testing.assert_allclose(out["loss"].item(), ref["loss"].item(), rtol=2e-2)
    np.testing.assert_allclose(out["logits"].cpu().numpy(), ref["logits"].numpy(), rtol=0, atol=3e-2)
    # train + gradients
    p = {k: v.clone().requires_grad_(True) for k, v in case["params"].items()}
    out = eng.pretrain_step(dev_batch(case["batch"]), training=True, backward=True, nfr_inject=inj)
    ref = po.pretrain_forward(p, cfg8, deq_tables, case["batch"], training=True, nfr_inject=inj_cpu)
    if torch.isnan(ref["loss"]):                      # reference quirk Q2 (no masked token): NaN on both sides
        assert torch.isnan(out["loss"]).item()
        return
    ref["loss"].backward()
    np.testing.assert_allclose(out["loss"].item(), ref["loss"].item(), rtol=2e-2)
    np.testing.assert_allclose(out["nfr"].item(), ref["nfr"].item(), rtol=2e-2)
    # the fp8 mode is a coarser approximation of the fp32 reference than bf16, but still tracks its golden loss
    np.testing.assert_allclose(out["loss"].item(), gold["train_loss"], rtol=5e-2)
    flat_ref = torch.cat([p[e["name"]].grad.reshape(-1) if p[e["name"]].grad is not None else torch.zeros(e["numel"])
                          for e in eng.entries])
    flat_eng = torch.cat([eng.view(e["name"], grad=True).reshape(-1) for e in eng.entries]).cpu()
    cos = torch.nn.functional.cosine_similarity(flat_eng, flat_ref, dim=0).item()
    assert cos > 0.99, cos
    nf = len(case["cfg"]["feat_hidden_sizes"])
    for nm in [f"bert.embeddings.feat_linear.{i}.weight" for i in range(nf)] + [
            "bert.encoder.layer.0.attention.self.query.weight", "bert.encoder.layer.0.attention.self.ctx_attention.weight"]:
        c = torch.nn.functional.cosine_similarity(eng.view(nm, grad=True).reshape(-1).cpu(), p[nm].grad.reshape(-1), dim=0).item()
        assert c > 0.98, (nm, c)


def test_fp8_engine_paths_agree_and_train():
    """Fused (fp8 projection inside qkvc_attn) vs unfused (quant_rows + gemm_nt_f8 + attention) and table vs per-token
    feature projection give the same step; a few optimizer steps on one batch descend."""
    _lib, L = _setup()
    case, _, _ = _fp8_case("m3")
    inj, _ = inject_for(case)
    batch = dev_batch(case["batch"])
    eng = make_engine(case, dtype="fp8")
    base = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj)
    g0 = eng.grads.clone()
    # layer inputs quantised by their producer (default) or by their consumer: the same bytes, hence the same step
    eng.set_option("consumer_quant", 1)
    eng.rng_state[1] = 0
    o = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj)
    eng.set_option("consumer_quant", 0)
    assert o["loss"].item() == base["loss"].item() and torch.equal(eng.grads, g0)
    for key in ("no_fused_qkvc_attention", "no_table_projection", "no_shortcut"):
        eng.set_option(key, 1)
        eng.rng_state[1] = 0
        o = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj)
        eng.set_option(key, 0)
        np.testing.assert_allclose(o["loss"].item(), base["loss"].item(), rtol=3e-3)
        assert torch.nn.functional.cosine_similarity(eng.grads, g0, dim=0).item() > 0.999
    losses = []
    for _ in range(8):
        o = eng.pretrain_step(batch, training=True, backward=True, nfr_inject=inj)
        eng.optimizer_step(lr=1e-3, weight_decay=1e-2, max_grad_norm=5.0)
        losses.append(o["loss"].item())
    assert losses[-1] < losses[0] and all(np.isfinite(losses))
    # pre-gathered feature tensors are refused in fp8 mode (the e4m3 gather is the A operand of the fp8 GEMM)
    tgt = case["batch"][0]
    feats = [f.cuda() for f in po.gather_feats(tgt["node_ids"], case["tables"])]
    with pytest.raises(RuntimeError, match="fp8 mode"):
        eng.encode(feats=feats, attention_mask=tgt["attention_mask"].cuda())
    last, _, _ = eng.encode(ids=tgt["node_ids"].cuda(), attention_mask=tgt["attention_mask"].cuda())
    assert torch.isfinite(last.float()).all()


@pytest.mark.parametrize("name", ["ncf_mlp", "ncf_neumf"])
def test_fp8_second_caller_tracks_the_reference(name):
    """PMGT_NCF (second caller of the encoder boundary: pmgt_encode_train / pmgt_encode_backward under autograd) in fp8 mode
    against the reference's fp32 goldens: logits and loss to fp8-level tolerances, encoder gradients by direction."""
    from pmgt_amd.configuration_pmgt import PMGTConfig
    from pmgt_amd.pmgt_ncf import PMGT_NCF
    c = gu.ncf_case(name)
    gold = c["gold"]
    model = PMGT_NCF(user_num=c["users"], item_num=c["n_nodes"], factor_num=c["factor"], num_layers=c["num_layers"],
                     model=c["model"], config=PMGTConfig(**c["cfg"]), dtype="fp8")
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in list(c["params"].items()) + list(c["head"].items()):
            sd[k].copy_(v)
    model.set_features([t.numpy() for t in c["tables"]])
    model.train()
    logits = model(c["user"], c["item"])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, c["labels"].cuda())
    loss.backward()
    np.testing.assert_allclose(loss.item(), gold["loss"], rtol=3e-2)
    np.testing.assert_allclose(logits.detach().float().cpu().numpy(), gold["logits"], rtol=0, atol=0.1 * np.abs(gold["logits"]).max() + 0.02)
    flat, ref = [], []
    for k, p in model.named_parameters():
        if not p.requires_grad or ("grad/" + k) not in gold.files:
            continue
        flat.append(p.grad.detach().float().cpu().reshape(-1))
        ref.append(torch.from_numpy(gold["grad/" + k].astype(np.float32)).reshape(-1))
    assert len(flat) > 10
    cos = torch.nn.functional.cosine_similarity(torch.cat(flat), torch.cat(ref), dim=0).item()
    assert cos > 0.98, cos
