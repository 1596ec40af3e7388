"""CPU checks of the oracle's fp8 (OCP e4m3) emulation -- the checker of the HIP fp8 mode (tests/test_fp8_gpu.py)."""
import torch

from oracle import pmgt_oracle as po


def test_fake_quant_rows_properties():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(50, 256, generator=g) * torch.logspace(-3, 3, 50).unsqueeze(1)
    x[7] = 0
    q = po.fake_quant_rows(x)
    assert torch.equal(q[7], torch.zeros(256))
    torch.testing.assert_close(q.abs().amax(-1), x.abs().amax(-1), rtol=1e-6, atol=0)   # the row maximum maps to +-448
    torch.testing.assert_close(po.fake_quant_rows(q), q, rtol=1e-6, atol=0)             # re-quantising changes nothing
    big = x.abs() > x.abs().amax(-1, keepdim=True) * 2 ** -6             # normal e4m3 numbers after scaling
    assert ((q - x).abs()[big] <= x.abs()[big] * 2 ** -4).all()          # 3 mantissa bits, round to nearest
    # e4m3fn, not fnuz: 448 is the largest finite value and survives the cast
    assert float(torch.tensor(448.0).to(torch.float8_e4m3fn).float()) == 448.0


def test_quantised_linear_forward_and_straight_through_backward():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 5, 64, generator=g, requires_grad=True)
    w = torch.randn(32, 64, generator=g, requires_grad=True)
    b = torch.randn(32, generator=g, requires_grad=True)
    y = po.linear_maybe_q(dict(fp8=True), x, w, b)
    ref = torch.nn.functional.linear(po.fake_quant_rows(x), po.fake_quant_rows(w), b)
    assert torch.equal(y, ref)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    x2, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    torch.nn.functional.linear(x2, w2, b2).backward(dy)
    torch.testing.assert_close(x.grad, x2.grad)
    torch.testing.assert_close(w.grad, w2.grad)
    torch.testing.assert_close(b.grad, b2.grad)
    # off by default: the restatement of the reference is untouched
    assert torch.equal(po.linear_maybe_q(dict(), x, w, b), torch.nn.functional.linear(x, w, b))


def test_fp8_emulation_stays_close_to_the_reference_math():
    from tests import golden_util as gu
    case = gu.model_case("m1")
    deq = [po.fake_quant_tensor(t)[0] for t in case["tables"]]
    a = po.pretrain_forward(case["params"], case["cfg"], case["tables"], case["batch"], training=False)
    b = po.pretrain_forward(case["params"], dict(case["cfg"], fp8=True), deq, case["batch"], training=False)
    assert abs(a["loss"].item() - b["loss"].item()) < 5e-2 * abs(a["loss"].item())
    assert a["loss"].item() != b["loss"].item()
