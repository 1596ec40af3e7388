#!/usr/bin/env python3
"""Op-level timing on the GPU box (A/B of kernel variants in ONE process, HIP events, interleaved rounds)."""
import ctypes as C
import sys
import os
import torch

DROP = float(os.environ.get('MB_DROP', '0.1'))

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmgt_amd import _lib  # noqa: E402

L = _lib.ops()
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, rounds=5, iters=10):
    best = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / iters * 1e3)
    best.sort()
    return best[0], best[len(best) // 2]


def nt(M, N, K, res=False, name=""):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(N, K, device="cuda").bfloat16()
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    R = torch.randn(M, N, device="cuda").bfloat16() if res else None
    byts = (M * K + M * N * (2 if res else 1)) * 2
    out = {}
    for mode in (0, 1):
        L.use(*(["tile_gemm"] if mode else []))
        f = lambda: _lib.check(L.pmgt_op_gemm_nt(1, P(A), K, None, P(W), K, P(Cc), N, M, N, K, None, 0, None, 0, P(R), N, 0.0, 0, None, None, st()))
        out[mode] = timeit(f)
    L.use()
    print(f"NT {name} M={M} N={N} K={K} res={res}: product path {out[0][0]:.1f}us ({byts / out[0][0] / 1e6:.2f} TB/s, {2 * M * N * K / out[0][0] / 1e6:.0f} TF/s) | "
          f"128x128 tile {out[1][0]:.1f}us ({byts / out[1][0] / 1e6:.2f} TB/s)")


def linear(M, N, K, res=False, ln=False, name=""):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(N, K, device="cuda").bfloat16()
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    R = torch.randn(M, N, device="cuda").bfloat16() if res else None
    lno = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if ln else None
    stats = torch.empty(M, 2, device="cuda") if ln else None
    g = torch.ones(N, device="cuda")
    byts = (M * K + M * N * (1 + (1 if res else 0) + (1 if ln else 0))) * 2
    out = {}
    for mode in (0, 1):
        L.use(*(["tile_gemm"] if mode else []))
        f = lambda: _lib.check(L.pmgt_op_linear(1, P(A), K, P(W), K, P(Cc), N, M, N, K, None, 0, None, 0, P(R), N, 0.0, 0, None,
                                                P(lno), P(stats), P(g) if ln else None, P(g) if ln else None, 1e-12, st()))
        out[mode] = timeit(f)
    L.use()
    print(f"LINEAR {name} M={M} N={N} K={K} res={res} ln={ln}: stream {out[0][0]:.1f}us ({byts / out[0][0] / 1e6:.2f} TB/s) | "
          f"tile {out[1][0]:.1f}us ({byts / out[1][0] / 1e6:.2f} TB/s)")


def tn(M, N1, N2, name=""):
    Pm = torch.randn(M, N1, device="cuda").bfloat16()
    Q = torch.randn(M, N2, device="cuda").bfloat16()
    slab = torch.empty(L.pmgt_op_gemm_tn_slab_elems(1, M, N1, N2), device="cuda")
    o = torch.empty(N1, N2, device="cuda")
    byts = M * (N1 + N2) * 2
    out = {}
    for mode in (0, 1):
        L.use(*(["tile_gemm"] if mode else []))
        f = lambda: _lib.check(L.pmgt_op_gemm_tn(1, P(Pm), N1, P(Q), N2, None, M, N1, N2, P(slab), P(o), 0, None, st()))
        out[mode] = timeit(f)
    L.use()
    print(f"TN {name} M={M} N1={N1} N2={N2}: dma {out[0][0]:.1f}us ({byts / out[0][0] / 1e6:.2f} TB/s, {2 * M * N1 * N2 / out[0][0] / 1e6:.0f} TF/s) | "
          f"regstage {out[1][0]:.1f}us ({byts / out[1][0] / 1e6:.2f} TB/s)")


def tn_gather(M, N1, N2, R, name=""):
    Pm = torch.randn(M, 2 * N1, device="cuda").bfloat16()
    table = torch.randn(R, N2, device="cuda").bfloat16()
    rows = torch.randint(0, R, (M,), device="cuda")
    slab = torch.empty(L.pmgt_op_gemm_tn_slab_elems(1, M, N1, N2), device="cuda")
    o = torch.empty(N1, N2, device="cuda")
    out = {}
    for mode in (0, 1):
        L.use(*(["tile_gemm"] if mode else []))
        f = lambda: _lib.check(L.pmgt_op_gemm_tn(1, P(Pm), 2 * N1, P(table), N2, P(rows), M, N1, N2, P(slab), P(o), 0, None, st()))
        out[mode] = timeit(f)
    L.use()
    print(f"TN-gather {name} M={M} N1={N1} N2={N2}: dma {out[0][0]:.1f}us ({2 * M * N1 * N2 / out[0][0] / 1e6:.0f} TF/s) | "
          f"regstage {out[1][0]:.1f}us ({2 * M * N1 * N2 / out[1][0] / 1e6:.0f} TF/s)")


def fused(T, H, name=""):
    S, dh = 32, 32
    d = H * dh
    x = torch.randn(T, S, d, device="cuda").bfloat16()
    W = (torch.randn(4 * d, d, device="cuda") / d ** 0.5).bfloat16()
    bias = torch.zeros(4 * d, device="cuda")
    qk = torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16)
    ctx = torch.empty(T, S, d, device="cuda", dtype=torch.bfloat16)
    rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
    M = T * S
    f = lambda: _lib.check(L.pmgt_op_qkvc_attention_fwd(P(x), P(W), P(bias), None, P(qk), P(ctx), T, S, H, dh, 0.5, DROP, 1, 2, P(rng), st()))

    def two():
        _lib.check(L.pmgt_op_linear(1, P(x), d, P(W), d, P(qk), 4 * d, M, 4 * d, d, P(bias), 0, None, 0, None, 0, 0.0, 0, None,
                                    None, None, None, None, 1e-12, st()))
        _lib.check(L.pmgt_op_attention_fwd(1, P(qk), None, P(ctx), None, T, S, H, dh, 0.5, 0.1, 1, 2, P(rng), st()))
    tf, tt = timeit(f), timeit(two)
    print(f"FUSED qkvc+attn {name} T={T} H={H}: fused {tf[0]:.1f}us ({M * 6 * d * 2 / tf[0] / 1e6:.2f} TB/s algorithmic) | gemm+attention {tt[0]:.1f}us")


def attn(T, S, H, dh, name=""):
    d = H * dh
    x = torch.randn(T, S, 4 * d, device="cuda").bfloat16()
    do = torch.randn(T, S, d, device="cuda").bfloat16()
    ctx = torch.empty(T, S, d, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16)
    rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
    f = lambda: _lib.check(L.pmgt_op_attention_fwd(1, P(x), None, P(ctx), None, T, S, H, dh, 0.5, DROP, 1, 2, P(rng), st()))
    b = lambda: _lib.check(L.pmgt_op_attention_bwd(1, P(x), None, P(do), P(dx), T, S, H, dh, 0.5, DROP, 1, 2, P(rng), st()))
    tf, tb = timeit(f), timeit(b)
    M = T * S
    print(f"ATTN {name} T={T} S={S} H={H} dh={dh} : fwd {tf[0]:.1f}us ({M * 5 * d * 2 / tf[0] / 1e6:.2f} TB/s) "
          f"bwd {tb[0]:.1f}us ({M * 9 * d * 2 / tb[0] / 1e6:.2f} TB/s)")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "attn":
        attn(12 * int(sys.argv[1]), 32, 8, 32)
        attn(12 * int(sys.argv[1]), 32, 4, 64)
        attn(3 * int(sys.argv[1]), 64, 8, 64)
        attn(3 * int(sys.argv[1]), 64, 8, 32)
        attn(6 * int(sys.argv[1]), 48, 8, 32)
        attn(24 * int(sys.argv[1]), 16, 8, 32)
        sys.exit(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    M = 12 * B * 32
    if len(sys.argv) > 2 and sys.argv[2] == "fused":
        fused(12 * B, 8)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "nt":
        nt(M, 256, 1024, res=True, name="dgrad_qkvc")
        M4 = 12 * 256 * 64
        nt(M4, 2048, 512, name="c4 qkvc fwd")
        nt(M4, 512, 2048, res=True, name="c4 dgrad_qkvc")
        nt(M4, 512, 512, res=True, name="c4 d x d")
        sys.exit(0)
    nt(M, 256, 1024, res=True, name="dgrad_qkvc")
    nt(M, 256, 256, res=False, name="dxd")
    linear(M, 1024, 256, name="qkvc_fwd")
    linear(M, 256, 256, res=True, ln=True, name="attn_out")
    linear(M, 256, 256, name="dgrad_dxd")
    tn(M, 1024, 256, name="wgrad_qkvc")
    tn(M, 256, 256, name="wgrad_dxd")
    tn_gather(M, 256, 1536, 7254, name="wgrad_featproj_v")
    tn_gather(M, 256, 768, 7254, name="wgrad_featproj_t")
