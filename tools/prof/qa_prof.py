"""Cycles per phase inside the fused forward kernels (library built with -DPMGT_QA_PROF)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _lib
L = _lib.ops()
P = lambda t: C.c_void_p(t.data_ptr())
T, S, H, dh = 12288, 32, 8, 32
d = H * dh
NS = 4
xs = [torch.randn(T, S, d, device="cuda").bfloat16() for _ in range(NS)]
W = (torch.randn(4 * d, d, device="cuda") / d ** 0.5).bfloat16()
bias = torch.zeros(4 * d, device="cuda")
qks = [torch.empty(T, S, 4 * d, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
ctxs = [torch.empty(T, S, d, device="cuda", dtype=torch.bfloat16) for _ in range(NS)]
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for k in range(9):
    i = k % NS
    if k == 8: ev[0].record()
    _lib.check(L.pmgt_op_qkvc_attention_fwd(P(xs[i]), P(W), P(bias), None, P(qks[i]), P(ctxs[i]), T, S, H, dh, 0.5, 0.1, 1, 2, P(rng), st))
ev[1].record()
torch.cuda.synchronize()
print("last launch: %.1f us" % (ev[0].elapsed_time(ev[1]) * 1e3))
if hasattr(C.CDLL(_lib._build.hip_lib_path()), "pmgt_debug_qa3_prof_read"):      # bf16: the role-split form is the only one
    # role-split form: workgroup 0, [wave][interval]: 0 = work that ends at a barrier (incl. the vmcnt / lgkmcnt waits), 1 = barrier wait,
    # GEMM waves: 2 = loop top, 3 = copy-out, 4 = DMA issue + fragment reads + MFMAs, 5 = bias / projection-tile writes;
    # attention waves: 2 = second half of a problem (the first half ends at the mid barrier: interval 0)
    raw3 = C.CDLL(_lib._build.hip_lib_path())
    b3 = np.zeros((16, 8), dtype=np.uint32)
    raw3.pmgt_debug_qa3_prof_read.argtypes = [C.c_void_p]
    assert raw3.pmgt_debug_qa3_prof_read(b3.ctypes.data) == 0
    for w in range(16):
        n = max(1, int(b3[w, 7]))
        per = b3[w, :7].astype(np.float64) / n
        print(f"  wave {w:2d} ({'GEMM' if w < 8 else 'attn'}): iterations {n} total/iter {per.sum():7.0f} | " + " | ".join(f"[{k}] {per[k]:6.0f}" for k in range(7)))
    sys.exit(0)
buf = np.zeros((2, 8, 8), dtype=np.uint32)
raw = C.CDLL(_lib._build.hip_lib_path())
raw.pmgt_debug_qa_prof_read.argtypes = [C.c_void_p]
assert raw.pmgt_debug_qa_prof_read(buf.ctypes.data) == 0
b = buf.astype(np.float64)
lab = ["attn-end->0", "0->1 sstore,gload,bar1", "1->2 frags+MFMA", "2->3 bias,qt writes", "3->4 bar2", "4->5 copy-out", "5->6 attention"]
for slot in range(2):
    print("block slot", slot)
    for w in range(8):
        n = b[slot, w, 7]
        if n == 0: continue
        per = b[slot, w, :7] / n
        print(f"  wave {w}: steps {int(n)} total/step {per.sum():7.0f} | " + " | ".join(f"{lab[k].split()[0]} {per[k]:6.0f}" for k in range(7)))
blk = np.zeros((1024, 4), dtype=np.uint64)
raw.pmgt_debug_qa_blk_read.argtypes = [C.c_void_p]
assert raw.pmgt_debug_qa_blk_read(blk.ctypes.data) == 0
nb = int((blk[:, 1] > 0).sum())
b = blk[:nb].astype(np.int64)
t0 = b[:, 0].min()
st_, en_ = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0     # us
hw = b[:, 2]; xcc = b[:, 3] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f"blocks {nb}: start min/median/max {st_.min():.1f}/{np.median(st_):.1f}/{st_.max():.1f} us, end min/median/max {en_.min():.1f}/{np.median(en_):.1f}/{en_.max():.1f} us, "
      f"lifetime median {np.median(en_ - st_):.1f} us")
uniq, cnt = np.unique(cuid, return_counts=True)
print(f"distinct CUs {len(uniq)}, blocks per CU: " + ", ".join(f"{k}x{int((cnt == k).sum())}" for k in sorted(set(cnt))))
late = st_ > 20
print(f"blocks starting later than 20 us: {int(late.sum())}")
life = en_ - st_
for x_ in range(8):
    m_ = xcc == x_
    if m_.any(): print(f"  xcc {x_}: blocks {int(m_.sum())} lifetime min/median/max {life[m_].min():.0f}/{np.median(life[m_]):.0f}/{life[m_].max():.0f} us; blockIdx%8 -> {sorted(set((np.arange(nb)[m_] % 8).tolist()))}")
ny_ = 4
yy = (np.arange(nb) >> 3) % ny_
for y_ in range(ny_):
    print(f"  slab {y_}: lifetime median {np.median(life[yy == y_]):.0f} max {life[yy == y_].max():.0f}")
m_ = (xcc == 0)
idx = np.arange(nb)[m_]
print("xcc0 lifetimes by block order:", " ".join(f"{int(v)}" for v in life[m_]))
print("xcc0 (se,sh,cu):", " ".join(f"{int(a_)}{int(b_)}{int(c_):x}" for a_, b_, c_ in zip(se[m_], sh[m_], cu[m_])))
