/*
 * pmgt_ops.h -- TEST AND A/B SURFACE of libpmgt_hip.so: single-kernel entry points (unit / parity tests of each kernel
 * against the oracle) and the catalogue of path options.  Not needed to use the engine (include/pmgt_capi.h is the
 * product ABI); same conventions as there.
 */
#ifndef PMGT_OPS_H
#define PMGT_OPS_H

#include "pmgt_capi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- path options ---------------------------------------------------------------------------------
 * Keys of the per-engine option setter declared in pmgt_capi.h, and the bit each one has in the `path_opts` argument of the pmgt_op_* entry
 * points below (which have no engine).  0 everywhere = the product path. */
#define PMGT_OPT_TILE_GEMM (1u << 0)                 /* "tile_gemm": register-staged 128 x 128 GEMM tiles only (no streaming kernel, LDS-DMA or 256 x 256 tiles) */
#define PMGT_OPT_VALU_ATTENTION (1u << 1)            /* "valu_attention": bf16 attention on the generic fp32-VALU kernel instead of the MFMA one */
#define PMGT_OPT_WAVE_ATTENTION_BWD (1u << 2)        /* "wave_attention_bwd": MFMA attention backward with one wave per (sequence, head) instead of cooperating waves */
#define PMGT_OPT_NO_SHORTCUT (1u << 3)               /* "no_shortcut": last layer on every token even when last_hidden is not requested */
#define PMGT_OPT_NO_FUSED_QKVC_ATTENTION (1u << 4)   /* "no_fused_qkvc_attention": projection GEMM and attention as two kernels */
#define PMGT_OPT_NO_HEAD_MAJOR (1u << 5)             /* "no_head_major": Q|K|V|C stays q | k | v | c between the fused forward and the backward */
#define PMGT_OPT_NO_TABLE_PROJECTION (1u << 6)       /* "no_table_projection": features projected per token even when the table is smaller than half the batch's tokens */
#define PMGT_OPT_NO_SEGMENT_SUM (1u << 7)            /* "no_segment_sum": table mode keeps the per-token weight-gradient GEMM of the feature projection */
#define PMGT_OPT_CONSUMER_QUANT (1u << 8)            /* "consumer_quant": fp8 mode, layer inputs quantised inside the fused forward instead of by their producer (bit-identical) */
#define PMGT_OPT_NO_FUSED_ATTENTION_BWD (1u << 9)    /* "no_fused_attention_bwd": attention backward and Q|K|V|C weight gradient as two kernels */
#define PMGT_OPT_STORE_LN_INPUT (1u << 10)           /* "store_ln_input": every LayerNorm site stores its input; default (bf16, hidden 256): x^ = (y - beta) / gamma from the OUTPUT */
#define PMGT_OPT_EAGER_REDUCE (1u << 11)             /* "eager_reduce": partial sums reduced by a launch per producer instead of one per gradient bucket */
#define PMGT_OPT_SIDE_STREAM_REDUCE (1u << 12)       /* "side_stream_reduce": ... on the engine's second stream (fork / join by events); measured neutral */
#define PMGT_OPT_UNFUSED_LN (1u << 13)               /* "unfused_ln": LayerNorm as its own launch after the streaming GEMM */
#define PMGT_OPT_ONE_BUCKET (1u << 14)               /* "one_bucket": the gradient-ready callback fires once per backward pass (whole buffer) */
#define PMGT_OPT_SMALL_ARENA (1u << 15)              /* "small_arena": (test) partial-sum arena sized for one producer: a batched reduction per producer */
#define PMGT_OPT_NO_ROLE_SPLIT_LN (1u << 16)         /* "no_role_split_ln": the 8-wave lockstep streaming kernels instead of the role-split ones of gemm_wsr.hip (K = N = 256 residual + LayerNorm; K = 512 plain / GELU / GELU' / residual) -- bit-identical results */
#define PMGT_OPT_UNFUSED_LN_BWD (1u << 18)           /* "unfused_ln_bwd": LayerNorm backward as its own launch behind the data-gradient GEMM that produces its dy (default, bf16 / hidden 256: epilogue of that GEMM) */
#define PMGT_OPT_LOCKSTEP_ATTENTION_BWD (1u << 19)   /* "lockstep_attention_bwd": fused attention backward with both (sequence, head) pairs of a step in the same phase instead of one barrier interval apart -- bit-identical results */
#define PMGT_OPT_SIDE_STREAM_WGRAD (1u << 20)        /* "side_stream_wgrad": the dense weight-gradient GEMMs of a layer on the engine's side stream, next to the data-gradient chain (same kernels, same reduction order: identical results; opt-in -- the cross-stream hand-offs cost more than the overlap gives at every batch size measured) */
#define PMGT_OPT_NO_CLS_ONLY_ATTENTION_BWD (1u << 21) /* "no_cls_only_attention_bwd": fused attention backward of the last (shortcut) layer without the skip of query tiles whose d ctx rows are zero -- identical results */
#define PMGT_OPT_NO_BETA_SKIP (1u << 22)            /* "no_beta_skip": beta == 1 (scripts/run_pmgt.sh:24) on the general fused kernels: Q / K projected, dot-product branch run and differentiated although it contributes exactly nothing (pmgt/pmgt/modeling_pmgt.py:519-521); default: skipped */
#define PMGT_OPT_NO_VC2_ATTENTION_BWD (1u << 23)    /* "no_vc2_attention_bwd": beta == 1 on the one-head-per-step vc_only backward instead of the two-heads-per-step form */
#define PMGT_OPT_NO_TILE_ATTENTION (1u << 17)        /* "no_tile_attention": S = 64 / head size 64 attention on the cooperative kernels (per-wave fragment loads) instead of the tile forms */

/* ---- which kernel families the calling thread has launched since the last reset (test instrumentation: a parity test at a given
 * size only covers a kernel if the dispatcher actually picked it).  Families: gemm_wsr, gemm_wsr_lnb, gemm_wsr512, gemm_ws, nt_big,
 * nt_big_gather, nt_big_128, nt_lnb, nt_tile, tn_big, tn_big_gather, tn_dma, tn_dma_gather, tn_tile, attn_tiles_fwd, attn_tiles_bwd,
 * qkvc_attn_fwd, attn_bwd_wgrad, f8_big, f8_tile, f8_wsr512, gemm_rowln, nt_lnf, embed_tok8, qkvc_attn_fwd_vc, attn_bwd_wgrad_vc, nt_vc, attn_bwd_wgrad_vc2 (the vc families: the beta == 1 forms, counted in addition to their general family).  Unknown name: -1. */
void pmgt_launch_trace_reset(void);
int64_t pmgt_launch_trace_count(const char* family);

/* ---- single-kernel entry points (unit/parity tests of each kernel against the oracle) ------------ */
/* Token order of the table-mode backward (the gather of pmgt/pmgt/utils.py:43-50 run in reverse: per-node sums of per-token gradients):
 * the stable sort of the M tokens by node id -- skeys [M] sorted ids, perm [M] token index at each sorted position (ties in index order),
 * seg_off [n_rows + 1] first sorted position of every id.  ids [M] int64 with values < n_rows; scratch_keys / scratch_vals [M] uint32;
 * temp: pmgt_op_seg_sort_temp_bytes(M) bytes.  Integer work: bit-exact against any stable sort. */
/* Measurement plumbing: `blocks` one-wave workgroups each write {shader-cycle counter, 100 MHz wall counter, XCC id, 1} to out [blocks][4]
 * (uint64).  Two probes on a stream around a timed region -> the shader clock the region sustained (bench.py: sustained_sclk_mhz). */
int pmgt_op_clock_probe(uint64_t* out, int blocks, void* stream);
int64_t pmgt_op_seg_sort_temp_bytes(int M);
int pmgt_op_seg_sort(const int64_t* ids, int M, int n_rows, uint32_t* scratch_keys, uint32_t* scratch_vals, uint32_t* skeys, uint32_t* perm,
                     int* seg_off, void* temp, int64_t temp_bytes, void* stream);
/* per-row absmax e4m3 quantisation (weights per output channel, activations per token): scale[r] = max|row| / 448 */
int pmgt_op_quant_rows_e4m3(int src_dtype, const void* src, int64_t lds, int rows, int cols, void* dst, int64_t ldd,
                            float* scale, void* stream);
/* C (bf16) = (A8 B8^T) * a_scale(row) * b_row_scale[n] + bias on the fp8 MFMA; a_rows = optional row gather on A */
int pmgt_op_gemm_nt_f8(const void* A, int64_t lda, const int64_t* a_rows, const float* a_row_scale, float a_scale,
                       const void* B, int64_t ldb, const float* b_row_scale, void* C, int64_t ldc, int M, int N, int K,
                       const float* bias, const int* m_dev, void* stream);
/* weight gradient with an e4m3 Q operand (feature-table rows): out = P^T (Q8 * q_scale), P bf16 */
int pmgt_op_gemm_tn_f8(const void* P, int64_t ldp, const void* Q8, int64_t ldq, float q_scale, const int64_t* q_rows, int M,
                       int N1, int N2, float* slab, float* out, int accumulate, const int* m_dev, void* stream);
/* fused projection + attention forward with the projection on the fp8 MFMA (d = 256): w8 [4d, d] e4m3, wscale [4d]; the layer
 * input either as bf16 x (quantised per row inside the kernel) or, x8 != NULL, as e4m3 rows + one scale per row */
int pmgt_op_qkvc_attention_fwd_f8(const void* x, const void* x8, const float* xscale, const void* w8, const float* wscale, const float* bias, const float* mask,
                                  void* qkvc, void* ctx, int n_seq, int S, int H, int dh, float beta, float drop_p,
                                  uint32_t site1, uint32_t site2, const uint64_t* rng, void* stream);
int pmgt_op_gemm_nt(int dtype, const void* A, int64_t lda, const int64_t* a_rows, const void* B, int64_t ldb, void* C,
                    int64_t ldc, int M, int N, int K, const float* bias, int epilogue, void* aux, int64_t ldaux,
                    const void* residual, int64_t ldr, float drop_p, uint32_t drop_site, const uint64_t* rng,
                    const int* m_dev, uint32_t path_opts, void* stream);
int64_t pmgt_op_gemm_tn_slab_elems(int dtype, int M, int N1, int N2, uint32_t path_opts);
int pmgt_op_gemm_tn(int dtype, const void* P, int64_t ldp, const void* Q, int64_t ldq, const int64_t* q_rows, int M,
                    int N1, int N2, float* slab, float* out, int accumulate, const int* m_dev, uint32_t path_opts, void* stream);
/* the same with the bias gradient (column sums of P; bias_slab: [512][N1] scratch) and the head-major row permutation of
 * the Q|K|V|C projection (perm_d = hidden size, perm_dh = head size; 0 = none) */
int pmgt_op_gemm_tn_bias(int dtype, const void* P, int64_t ldp, const void* Q, int64_t ldq, int M, int N1, int N2, float* slab,
                         float* out, float* bias_slab, float* bias_out, int perm_d, int perm_dh, uint32_t path_opts, void* stream);
/* column sums of Y [M, N]; slab: ceil(M / 96) * N floats of scratch */
int pmgt_op_colsum(int dtype, const void* Y, int64_t ldy, int M, int N, float* slab, float* out, void* stream);
int pmgt_op_layernorm_fwd(int dtype, const void* x, void* y, float* stats, const float* gamma, const float* beta,
                          int M, int d, float eps, float drop_p, uint32_t drop_site, const uint64_t* rng, void* stream);
/* part: [ceil(M/64)][3][d] scratch; dgamma_dbeta: [3*d] out = dgamma | dbeta | column sum of dx_drop (or dx) */
int pmgt_op_layernorm_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                          void* dx_drop, float* part, float* dgamma_dbeta, int M, int d, float in_drop_p,
                          uint32_t in_site, float out_drop_p, uint32_t out_site, const uint64_t* rng, void* stream);
/* One linear layer through the engine's dispatcher: bf16 with K <= 256 runs the weight-stationary streaming
 * kernel (gemm_ws.hip), everything else the tiled one; ln_out != NULL adds LayerNorm(C) (fused when N == 256). */
int pmgt_op_linear(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                   const float* bias, int epilogue, void* aux, int64_t ldaux, const void* residual, int64_t ldr, float drop_p,
                   uint32_t drop_site, const uint64_t* rng, void* ln_out, float* ln_stats, const float* ln_gamma,
                   const float* ln_beta, float ln_eps, uint32_t path_opts, void* stream);
/* dy = A B^T + residual, then the backward of the LayerNorm whose OUTPUT is y (x^ = (y - beta) / gamma; stats = its {mean, rstd}
 * rows): dx, dx_drop = dx * dropout mask (optional), dgamma | dbeta | column sums of the bf16 dx_drop (or dx) [3 N].  bf16.  One launch
 * where the role-split streaming kernel (K = N = 256, M >= 8192) or the 256 x 256 tile with the LayerNorm-backward phase (N = 256,
 * K % 64 == 0, >= 96 tiles) applies, else GEMM -> dy_tmp [M, N] -> LayerNorm backward (autograd through BertSelfOutput /
 * BertOutput, pmgt/pmgt/modeling_pmgt.py:293-294,322-325,332,371).  part: scratch of max(256, ceil(M / 64)) * 3 N floats. */
int pmgt_op_linear_ln_bwd(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, const void* residual, int64_t ldr,
                          const void* y, const float* stats, const float* gamma, const float* beta, void* dy_tmp, void* dx, void* dx_drop,
                          float drop_p, uint32_t drop_site, const uint64_t* rng, float* part, float* dgamma_dbeta_dbias,
                          uint32_t path_opts, void* stream);
/* Fused Q|K|V|C projection + attention forward (bf16; S = 32, dh = 32, hidden 128 or 256; returns -3 otherwise):
 * x [n_seq*S, d], w [4d, d] (rows q | k | v | c), bias [4d] fp32 -> qkvc [n_seq*S, 4d], ctx [n_seq*S, d]. */
int pmgt_op_qkvc_attention_fwd(const void* x, const void* w, const float* bias, const float* mask, void* qkvc, void* ctx,
                               int n_seq, int S, int H, int dh, float beta, float drop_p, uint32_t site1, uint32_t site2,
                               const uint64_t* rng, void* stream);
/* the same with layout / mode flags: bit 0 = qkvc written head-major ((head, matrix, w) columns), bit 1 = vc_only (beta == 1, H % 4 == 0: V and C
 * projected and stored only, cosine branch alone; the Q / K columns of qkvc are left untouched) */
int pmgt_op_qkvc_attention_fwd_ex(const void* x, const void* w, const float* bias, const float* mask, void* qkvc, void* ctx,
                                  int n_seq, int S, int H, int dh, float beta, float drop_p, uint32_t site1, uint32_t site2,
                                  const uint64_t* rng, int flags, void* stream);
int pmgt_op_attention_fwd(int dtype, const void* qkvc, const float* mask, void* ctx, float* probs, int n_seq, int S,
                          int H, int dh, float beta, float drop_p, uint32_t site1, uint32_t site2,
                          const uint64_t* rng, uint32_t path_opts, void* stream);
int pmgt_op_attention_bwd(int dtype, const void* qkvc, const float* mask, const void* dctx, void* dqkvc, int n_seq,
                          int S, int H, int dh, float beta, float drop_p, uint32_t site1, uint32_t site2,
                          const uint64_t* rng, uint32_t path_opts, void* stream);
/* Attention backward fused with the weight / bias gradient of the Q|K|V|C projection (bf16, S = 32, head size 32, hidden 128 or
 * 256; replaces pmgt_op_attention_bwd + the [M, 4d]^T [M, d] weight-gradient GEMM, i.e. autograd through
 * pmgt/pmgt/modeling_pmgt.py:429-433 and :435-526).  x = the layer input [n_seq * 32, d]; dqkvc as pmgt_op_attention_bwd;
 * slab [parts][4d * d] / bias_slab [parts][4d] (parts = pmgt_op_attention_bwd_wgrad_parts(H)) receive per-workgroup partial
 * sums in q | k | v | c row order, to be added up by the caller.  head_major: bit 0 = the column layout of qkvc / dqkvc, bit 1 = vc_only
 * (beta == 1: Q / K columns of qkvc are not read, dQ / dK columns of dqkvc not written, query / key rows of the partial sums are zeros). */
int pmgt_op_attention_bwd_wgrad_parts(int H);
int pmgt_op_attention_bwd_wgrad_vc2_parts(int H);      /* head_major bit 3: the two-barrier kernel; bit 2 (with bit 1): two heads per step; slab [vc2_parts][2d * d], bias_slab [vc2_parts][2d]: value | ctx_attention rows only */
int pmgt_op_attention_bwd_wgrad(const void* qkvc, const float* mask, const void* dctx, const void* x, void* dqkvc, float* slab,
                                float* bias_slab, int n_seq, int H, float beta, float drop_p, uint32_t site1, uint32_t site2,
                                const uint64_t* rng, int head_major, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PMGT_OPS_H */
