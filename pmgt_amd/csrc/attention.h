// PMGT dual-softmax attention (declarations); see attention.hip.
#pragma once
#include "common.h"

namespace pmgt {

struct AttnArgs {
    const void* qkvc = nullptr;   // [Tseq*S, 4d]: q | k | v | c (ctx_attention), head h at columns h*dh
    const float* mask = nullptr;  // [Tseq, S] 1 = valid key, 0 = padded key (nullable = all valid)
    void* ctx = nullptr;          // [Tseq*S, d] forward output
    float* probs = nullptr;       // optional [Tseq, H, S, S] mixed probabilities (output_attentions)
    int Tseq = 0, S = 0, H = 0, dh = 0;
    float beta = 0.5f;
    DropCfg drop1 = {nullptr, 0.f, 0}, drop2 = {nullptr, 0.f, 0};
    // Last layer of the training fast path: sequences t < cls_only_seqs are read by the loss at row 0 only, so only their
    // first 16-query tile needs attention output (forward) / carries a gradient (backward); 0 = every query matters.
    int cls_only_seqs = 0;
    // Head-major Q|K|V|C (and dQ|dK|dV|dC): column of (matrix m, head h, w) is (h * 4 + m) * dh + w instead of
    // m * d + h * dh + w, i.e. 4 * dh contiguous elements per (row, head).  Written by the fused forward, understood by
    // the one-wave MFMA backward (the only pair used together).
    bool hm = false;
    // beta == 1 with the dead branch skipped (pmgt/pmgt/modeling_pmgt.py:519-521: the dot-product softmax then contributes exactly nothing):
    // Q and K are NEITHER READ NOR WRITTEN by the fused kernels -- the forward that set it left their columns of qkvc unwritten, the fused
    // backward leaves the dQ / dK columns of dqkvc unwritten and reports zero weight / bias gradients for query / key (what autograd gives)
    bool vc_only = false;
    const void* dctx = nullptr;   // backward: [Tseq*S, d]
    void* dqkvc = nullptr;        // backward: [Tseq*S, 4d]
    uint32_t opts = 0;            // PathOpt bits: OPT_VALU_ATTENTION, OPT_WAVE_ATTENTION_BWD
};
// bf16 MFMA path (attention_mfma.hip): S <= 64, head size 32 or 64
bool attn_mfma_supported(const AttnArgs& a);
int attn_mfma(const AttnArgs& a, bool bwd, hipStream_t st);
template <typename T> int attn_fwd(const AttnArgs& a, hipStream_t st);
template <typename T> int attn_bwd(const AttnArgs& a, hipStream_t st);


// Attention backward fused with the Q|K|V|C weight gradient (attention_mfma.hip): bf16, S = 32, head size 32, hidden 256 / 128.
// One launch replaces attn_bwd + the Q|K|V|C gemm_tn: dQ|dK|dV|dC still go to HBM once (the data gradient reads them), the
// weight / bias gradients leave as `parts` = attn_bwd_wgrad_parts(H) partial slabs [parts][4d, d] / [parts][4d] for slab_reduce.
struct AttnBwdWg {
    AttnArgs a;                                    // qkvc, dctx, dqkvc, mask, Tseq, S, H, dh, beta, dropout, hm, cls_only_seqs (their query rows 16 .. 31 are skipped: d ctx must be zero there)
    const void* x = nullptr; int64_t ldx = 0;      // [Tseq*32, d] the layer input the projection was applied to
    float* slab = nullptr;                         // [parts][4d * d]
    float* bias_slab = nullptr;                    // [parts][4d] or NULL
};
int attn_bwd_wgrad_parts(int H);
bool attn_bwd_wgrad_supported(const AttnBwdWg& w);
bool attn_bwd_wgrad_shape_ok(int Tseq, int S, int dh, int H);
int attn_bwd_wgrad(const AttnBwdWg& w, hipStream_t st);
// beta == 1 (vc_only), two heads per step: `slab` is [attn_bwd_wgrad_vc2_parts(H)][2 d * d] and `bias_slab` [..][2 d] -- the value | ctx_attention rows
// only (row = matrix * d + head * 32 + w); the query / key gradients are zeros the CALLER writes
int attn_bwd_wgrad_vc2_parts(int H);
bool attn_bwd_wgrad_vc2_supported(const AttnBwdWg& w);
int attn_bwd_wgrad_vc2(const AttnBwdWg& w, hipStream_t st);

// Fused Q|K|V|C projection + attention forward (qkvc_attn.hip): bf16, S = 32, head size 32, hidden 256 or 128.
struct QkvcAttn {
    const void* X = nullptr; int64_t ldx = 0;     // [Tseq*32, d] layer input
    const void* W = nullptr; int64_t ldw = 0;     // [4d, d]: rows q | k | v | c (nn.Linear layout)
    const float* bias = nullptr;                  // [4d]
    void* qkvc = nullptr; int64_t ldq = 0;        // [Tseq*32, 4d] out (kept for the backward pass)
    void* ctx = nullptr; int64_t ldc = 0;         // [Tseq*32, d] out
    const float* mask = nullptr;                  // [Tseq, 32] or NULL
    int Tseq = 0, S = 0, H = 0, dh = 0;
    float beta = 0.5f;
    DropCfg drop1 = {nullptr, 0.f, 0}, drop2 = {nullptr, 0.f, 0};
    int cls_only_seqs = 0;                        // see AttnArgs
    bool hm = false;                              // write Q|K|V|C head-major (see AttnArgs)
    // beta == 1 only (the author's own setting, scripts/run_pmgt.sh:24): project V and C only -- a column slab is {V, C} x FOUR heads instead of
    // {Q, K, V, C} x two -- and run the cosine branch alone; the Q / K columns of qkvc are not written (see AttnArgs::vc_only)
    bool vc_only = false;
    // fp8 mode (hidden 256 only): the projection runs on the block-scaled fp8 MFMA (unit scales) -- W8 = e4m3 copy of W quantised per
    // output channel (value = byte * wscale[n]), x quantised per row inside the kernel (fp8.h contract); the outputs and the
    // attention stay bf16
    const void* W8 = nullptr;                     // [4d, d] e4m3, row stride ldw bytes
    const float* wscale = nullptr;                // [4d]
    // ... and, when the producer of x already quantised it (gemm_ws fused-LayerNorm epilogue, embed_mix): e4m3 rows [Tseq*32, d]
    // with row stride ldx BYTES and one scale per row; X is then not read
    const void* X8 = nullptr;
    const float* xscale = nullptr;
    uint32_t opts = 0;                            // PathOpt bits of the calling engine (none is read here today)
};
bool qkvc_attn_supported(const QkvcAttn& a);
int qkvc_attn_fwd(const QkvcAttn& a, hipStream_t st);

}  // namespace pmgt
