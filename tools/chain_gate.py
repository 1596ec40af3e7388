#!/usr/bin/env python3
"""Evidence gate for a fused per-layer dense chain, with NO kernel changes: every launch of the chain
(attn-out + LN -> FFN1 + GELU -> FFN2 + LN and dgrad_ffn2 -> dgrad_ffn1 + LN-backward -> dgrad_attn_out) is timed at the bench's
token count with its HBM streams collapsed one by one -- leading dimension 0 makes every row of an operand the SAME row (reads hit L2,
writes land on one line), which is exactly what a fused chain would save (the tensor stays on chip) while the kernel's instruction stream
is unchanged.  A launch whose time does not move when its bytes disappear is issue-bound: fusing it with its neighbour saves its launch
and nothing else.  Results are garbage by construction (an ablation, not a path).

Usage (GPU box):  python tools/chain_gate.py [B]     -> one table on stdout
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmgt_amd import _lib  # noqa: E402

L = _lib.ops()
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, rounds=5, iters=10):
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters * 1e3)
    return min(ts)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    M, d, I = 12 * B * 32, 256, 256
    U = M * d * 2 / 1e6
    bf = lambda *s: torch.randn(*s, device="cuda").bfloat16()
    A, R, out, pre, act, y = bf(M, d), bf(M, d), bf(M, d), bf(M, I), bf(M, I), bf(M, d)
    dxd, tmp = bf(M, d), bf(M, d)
    W = bf(256, 256) * 0.05
    bias = torch.zeros(256, device="cuda")
    gam, bet = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
    stats = torch.rand(M, 2, device="cuda") + 0.5
    rng = torch.tensor([1234, 5], dtype=torch.int64, device="cuda")
    part = torch.empty(max(256, (M + 63) // 64) * 3 * 256, device="cuda")
    dgb = torch.empty(3 * 256, device="cuda")
    rows = []

    def run(name, algo_u, variants):
        base = None
        for label, u_left, fn in variants:
            t = timeit(fn)
            base = base or t
            rows.append((name, label, algo_u, u_left, t, (u_left * U / t) if u_left else 0.0, t / base))

    # ---- forward: y = LN(dropout(A W^T + b) + res)   (attn-out and FFN2: gemm_wsr_kernel<false>; the engine does not store the LayerNorm input
    #      -- skip_c, not reachable through the op entry: here it goes to ONE row, ldc = 0)
    def f_ln(lda, ldr):
        return lambda: _lib.check(L.pmgt_op_linear(1, P(A), lda, P(W), 256, P(tmp), 0, M, 256, 256, P(bias), 0, None, 0, P(R), ldr, 0.1, 7, P(rng),
                                                   P(out), P(stats), P(gam), P(bet), 1e-12, st()))
    run("fwd attn-out / ffn2 (+res +LN)", 3, [("as is", 3, f_ln(256, 256)), ("A from one row", 2, f_ln(0, 256)),
                                               ("A and residual from one row", 1, f_ln(0, 0))])

    # ---- forward FFN1: g = gelu(pre), pre = A W^T + b, both stored   (gemm_ws2_kernel<8, 1>)
    def f_gelu(lda, ldc, ldaux):
        return lambda: _lib.check(L.pmgt_op_linear(1, P(A), lda, P(W), 256, P(act), ldc, M, I, 256, P(bias), 1, P(pre), ldaux, None, 0, 0.0, 0, None,
                                                   None, None, None, None, 0.0, st()))
    run("fwd ffn1 (+GELU, pre + act out)", 3, [("as is", 3, f_gelu(256, I, I)), ("act to one row", 2, f_gelu(256, 0, I)),
                                                ("u from one row, act to one row", 1, f_gelu(0, 0, I)),
                                                ("nothing through HBM", 0, f_gelu(0, 0, 0))])

    # ---- backward dgrad_ffn2: dAct = (dy W2) * gelu'(pre)   (gemm_ws2_kernel<8, 2>)
    def b_gelu(lda, ldc, ldaux):
        return lambda: _lib.check(L.pmgt_op_linear(1, P(A), lda, P(W), 256, P(act), ldc, M, I, 256, None, 2, P(pre), ldaux, None, 0, 0.0, 0, None,
                                                   None, None, None, None, 0.0, st()))
    run("bwd dgrad_ffn2 (x GELU')", 3, [("as is", 3, b_gelu(256, I, I)), ("dAct to one row", 2, b_gelu(256, 0, I)),
                                         ("dy from one row, dAct to one row", 1, b_gelu(0, 0, I)), ("nothing through HBM", 0, b_gelu(0, 0, 0))])

    # ---- backward dgrad_ffn1 + residual + LN1 backward   (gemm_wsr_kernel<true>): dAct in, dres in, y in; dx, dx_drop out
    def b_lnb(lda, ldr):
        return lambda: _lib.check(L.pmgt_op_linear_ln_bwd(P(act), lda, P(W), 256, M, 256, 256, P(R), ldr, P(y), P(stats), P(gam), P(bet), P(tmp), P(out), P(dxd),
                                                          0.1, 35, P(rng), P(part), P(dgb), st()))
    run("bwd dgrad_ffn1 + LN1 backward", 5, [("as is", 5, b_lnb(256, 256)), ("dAct from one row", 4, b_lnb(0, 256)),
                                              ("dAct and residual from one row", 3, b_lnb(0, 0))])

    # ---- backward dgrad_attn_out: dctx = dU Wo   (gemm_ws2_kernel<8, 0>)
    def b_plain(lda, ldc):
        return lambda: _lib.check(L.pmgt_op_linear(1, P(A), lda, P(W), 256, P(out), ldc, M, 256, 256, None, 0, None, 0, None, 0, 0.0, 0, None,
                                                   None, None, None, None, 0.0, st()))
    run("bwd dgrad_attn_out", 2, [("as is", 2, b_plain(256, 256)), ("dU from one row", 1, b_plain(0, 256)), ("nothing through HBM", 0, b_plain(0, 0))])

    print(f"# dense-chain evidence gate: B = {B} targets, M = {M} tokens, d = I = 256, U = {U:.1f} MB; times = best of 5 x 10 launches (HIP events)")
    print(f"{'launch':34s} {'variant':36s} {'U moved':>8s} {'us':>8s} {'TB/s':>6s} {'vs as-is':>9s}")
    for name, label, algo, left, t, tbs, rel in rows:
        print(f"{name:34s} {label:36s} {left:>4d}/{algo:<3d} {t:8.1f} {tbs:6.2f} {rel:9.3f}")


if __name__ == "__main__":
    main()
