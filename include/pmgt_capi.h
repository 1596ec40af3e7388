/*
 * pmgt_capi.h — C ABI of the MI355X-native PMGT pre-training engine (libpmgt_hip.so) and of the
 * host MCNSampling library (libpmgt_sampler.so).
 *
 * The reference (uoo723/PMGT) is pure Python and has no FFI of its own; this ABI is what a
 * maintainer binds (ctypes) behind the reference's Python surface.  Each entry
 * point names the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions: plain pointers and sizes only (no torch types); all tensor memory is owned by the
 * caller (device pointers unless stated); row-major contiguous; int64 ids, fp32 masks, nn.Linear
 * weights as [out, in]; every call is asynchronous on the given hipStream_t (passed as void*), never
 * synchronises, never throws; returns 0 or a negative code, message via pmgt_last_error().
 */
#ifndef PMGT_CAPI_H
#define PMGT_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMGT_DTYPE_F32 0  /* parity mode: fp32 storage, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) */
#define PMGT_DTYPE_BF16 1 /* perf mode: bf16 activations/weight copies, fp32 accumulate + master weights */
/* fp8 mode (BASELINE.json config 5): the bf16 engine with OCP e4m3 where the north star names it -- frozen feature tables
 * stored as e4m3 (one scale per table), feature-projection and Q|K|V|C projections on the fp8 MFMA
 * (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales) with
 * weights quantised per output channel and activations per row; every other tensor, the attention, the backward GEMMs and
 * the optimizer as in PMGT_DTYPE_BF16. */
#define PMGT_DTYPE_FP8 2

/* Mirrors PMGTConfig (pmgt/pmgt/configuration_pmgt.py:11-41). */
typedef struct pmgt_config {
    int hidden_size;
    int num_hidden_layers;
    int num_attention_heads;
    int intermediate_size;
    int feat_size_v; /* feat_hidden_sizes[0] (visual, 1536) */
    int feat_size_t; /* feat_hidden_sizes[1] (textual, 768) */
    int max_position_embeddings;
    float layer_norm_eps;
    float beta;
    float hidden_dropout_prob;
    float attention_probs_dropout_prob;
    int dtype; /* PMGT_DTYPE_* */
} pmgt_config;

typedef struct pmgt_engine pmgt_engine;

const char* pmgt_last_error(void);
int pmgt_abi_version(void);

/* ---- engine lifetime & parameter layout ---------------------------------------------------------
 * Replaces PMGT.__init__ / PMGTModel.__init__ module construction (pmgt/pmgt/models.py:22-54,
 * pmgt/pmgt/modeling_pmgt.py:65-74,155-187,213-220,287-294,378-410,549-558): parameters live in ONE
 * flat fp32 buffer; pmgt_param_entry() reports where each reference-named tensor sits in it. */
pmgt_engine* pmgt_engine_create(const pmgt_config* cfg);
void pmgt_engine_destroy(pmgt_engine* e);
int64_t pmgt_param_count(const pmgt_engine* e);
int pmgt_param_num_entries(const pmgt_engine* e);
/* name: reference state_dict key (e.g. "bert.encoder.layer.0.attention.self.query.weight");
 * offset/numel in floats; rows/cols (cols = 0 for vectors); decay = 1 if AdamW weight decay applies
 * (pmgt/base_trainer.py:38-59). */
int pmgt_param_entry(const pmgt_engine* e, int index, char* name, int name_cap, int64_t* offset, int64_t* numel,
                     int* rows, int* cols, int* decay);
/* bytes of scratch the calls below need for n_seq sequences of seq_len tokens (n_targets of them targets). */
int64_t pmgt_workspace_bytes(const pmgt_engine* e, int n_seq, int seq_len, int n_targets, int training);

/* persistent device tensors owned by the caller */
typedef struct pmgt_tensors {
    float* params;       /* [pmgt_param_count] fp32 master weights */
    float* grads;        /* same shape; written (or accumulated) by pmgt_pretrain_step */
    const void* table_v; /* [n_nodes + 2, feat_size_v] frozen features in the engine dtype (models.py:40-54); e4m3 in fp8 mode */
    const void* table_t; /* [n_nodes + 2, feat_size_t] */
    int64_t n_nodes;
    uint64_t* rng_state; /* device [2]: {seed, step}; drives dropout + NFR masking */
    /* PMGT_DTYPE_FP8 only: the tables are e4m3 bytes, feature value = byte value * table_scale_{v,t} (pmgt_quantize_e4m3) */
    float table_scale_v;
    float table_scale_t;
} pmgt_tensors;

/* One collated batch, exactly what pmgt_collate_fn returns (pmgt/pmgt/datasets.py:186-208). */
typedef struct pmgt_batch {
    int n_targets;            /* B */
    int n_pairs;              /* sum(num_pairs) */
    int seq_len;              /* S = max_ctx_neigh + 1 */
    const int64_t* tgt_ids;   /* [B, S] */
    const float* tgt_mask;    /* [B, S] */
    const int64_t* pair_ids;  /* [P, S] (NULL with n_pairs = 0: inference) */
    const float* pair_mask;   /* [P, S] */
    const int64_t* num_pairs; /* [B] */
    const float* labels;      /* [P] */
    /* optional injected NFR masking (parity tests): masked ids and, per position, the id to
     * reconstruct (-1 = not masked).  NULL = draw on device (pmgt/pmgt/models.py:132-151). */
    const int64_t* nfr_masked_ids; /* [B, S] */
    const int64_t* nfr_targets;    /* [B, S] */
    float random_node_ratio;
    float mask_node_ratio;
} pmgt_batch;

typedef struct pmgt_outputs {
    float* loss;       /* device [3]: loss, gsr, nfr */
    float* logits;     /* device [P] prediction_logits */
    void* last_hidden; /* device [B, S, d] in the engine dtype (target sequences), nullable */
    int* nfr_count;    /* device [1], nullable: number of masked positions */
} pmgt_outputs;

#define PMGT_FLAG_TRAINING 1   /* dropout + NFR branch (module.training) */
#define PMGT_FLAG_BACKWARD 2   /* also compute parameter gradients */
#define PMGT_FLAG_ACCUMULATE 4 /* grads += instead of grads = */

/* PMGT.forward + loss.backward() of one batch (pmgt/pmgt/models.py:56-176; trainer step
 * pmgt/pmgt/trainer.py:156-160). */
int pmgt_pretrain_step(pmgt_engine* e, const pmgt_tensors* t, const pmgt_batch* b, const pmgt_outputs* o,
                       void* workspace, int64_t workspace_bytes, int flags, void* stream);

/* PMGTModel.forward on node ids (gather fused): inference/export path
 * (pmgt/pmgt/modeling_pmgt.py:80-152, pmgt/pmgt/trainer.py:153-154).  hidden_states: optional
 * [L+1, n_seq, S, d]; attn_probs: optional [L, n_seq, H, S, S] fp32 (output_attentions). */
int pmgt_encode_ids(pmgt_engine* e, const pmgt_tensors* t, const int64_t* ids, const float* mask, int n_seq,
                    int seq_len, void* last_hidden, void* hidden_states, float* attn_probs, void* workspace,
                    int64_t workspace_bytes, void* stream);
/* PMGTModel.forward(*input_feat_embeds) on already-gathered features [n_seq, S, F_m] in the engine dtype. */
int pmgt_encode_feats(pmgt_engine* e, const pmgt_tensors* t, const void* feat_v, const void* feat_t,
                      const float* mask, int n_seq, int seq_len, void* last_hidden, void* hidden_states,
                      float* attn_probs, void* workspace, int64_t workspace_bytes, void* stream);

/* PMGTModel.forward + backward for a caller that owns the head (second caller of the boundary: PMGT_NCF,
 * pmgt/pmgt_ncf/models.py:77-105 -- `self.bert(*input_feat_embeds, attention_mask=...)[0][:, 0]` followed by
 * autograd).  pmgt_encode_train runs the encoder on node ids (ids != NULL, gather fused) or on gathered features
 * (feat_v/feat_t [n_seq, S, F_m], engine dtype), keeps every activation in `workspace`
 * (pmgt_workspace_bytes(e, n_seq, S, 1, 1) bytes, untouched until the backward call) and snapshots the dropout
 * counter; PMGT_FLAG_TRAINING turns dropout on and advances the counter.  pmgt_encode_backward takes
 * d loss / d last_hidden_state [n_seq, S, d] (engine dtype) and leaves the gradients of every `bert.*` entry in
 * t->grads (= or += with PMGT_FLAG_ACCUMULATE; pass the same TRAINING flag); the frozen tables get none
 * (pmgt/pmgt_ncf/models.py:45-47).  feat_v/feat_t are needed again only when the forward used them. */
int pmgt_encode_train(pmgt_engine* e, const pmgt_tensors* t, const int64_t* ids, const void* feat_v, const void* feat_t,
                      const float* mask, int n_seq, int seq_len, void* last_hidden, void* workspace,
                      int64_t workspace_bytes, int flags, void* stream);
int pmgt_encode_backward(pmgt_engine* e, const pmgt_tensors* t, const void* feat_v, const void* feat_t,
                         const void* d_last_hidden, int n_seq, int seq_len, void* workspace, int64_t workspace_bytes,
                         int flags, void* stream);

/* Global-norm clip + DenseSparseAdamW dense step over the flat buffers
 * (pmgt/base_trainer.py:312-315, pmgt/optimizers.py:256-270). */
typedef struct pmgt_adam {
    float* exp_avg;       /* [count] */
    float* exp_avg_sq;    /* [count] */
    const uint8_t* decay; /* [count] 1 where weight decay applies */
    float lr, weight_decay, beta1, beta2, eps;
    float max_grad_norm; /* <= 0: no clipping */
    int64_t* step;       /* device [1] */
    float* scalars;      /* device [4] out: clip coef, lr/bc1, 1/sqrt(bc2), grad norm */
    float* scratch;      /* device [1024] */
} pmgt_adam;
int pmgt_optimizer_step(pmgt_engine* e, const pmgt_tensors* t, const pmgt_adam* a, void* stream);

/* Gradient-ready notification for the data-parallel exchange (replaces DDP's autograd hooks + buckets,
 * pmgt/base_trainer.py:309-322 -> pl.Trainer(gpus=N)): during a backward pass the engine calls cb(user, offset, numel) on
 * the CALLING host thread right after it has enqueued the last launch that writes grads[offset, offset + numel) -- i.e.
 * work the callee enqueues on `stream` (an event record, an all-reduce that waits for that event) is ordered after those
 * writes and overlaps the rest of the backward pass.  Buckets arrive in backward order and tile the flat buffer exactly
 * once per backward call: NFR head (pmgt_pretrain_step only; pmgt_encode_backward produces no gradient for it and its
 * buckets tile the `bert.*` range), encoder layers L-1 .. 0, embeddings.  cb = NULL switches it off.  With the
 * side-stream reductions on (pmgt_engine_set_overlap) one bucket covering the whole buffer is reported at the end. */
typedef void (*pmgt_grad_ready_fn)(void* user, int64_t offset, int64_t numel);
void pmgt_engine_set_grad_ready_callback(pmgt_engine* e, pmgt_grad_ready_fn cb, void* user);

/* Per-phase timers (what the reference lacks entirely): between begin and end every group of
 * kernel launches is bracketed by HIP events on its stream; end() waits for them and writes one
 * "name count total_ms" line per phase into buf.  Off by default. */
int pmgt_profile_begin(pmgt_engine* e);
int pmgt_profile_end(pmgt_engine* e, char* buf, int cap);

/* dtype plumbing */
int pmgt_cast_from_f32(int dtype, const float* src, void* dst, int64_t n, void* stream);
int pmgt_cast_to_f32(int dtype, const void* src, float* dst, int64_t n, void* stream);

/* fp8 mode plumbing: dst[i] = e4m3_rne(clamp(src[i] * inv_scale, +-448)) and back (n % 8 == 0).  The frozen tables are
 * quantised once by the caller with inv_scale = 448 / max|table|; table_scale = max|table| / 448. */
int pmgt_quantize_e4m3(const float* src, void* dst, int64_t n, float inv_scale, void* stream);
int pmgt_dequantize_e4m3(const void* src, float* dst, int64_t n, float scale, void* stream);

/* ---- single-kernel entry points (unit/parity tests of each kernel against the oracle) ------------ */
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
                    const int* m_dev, void* stream);
int64_t pmgt_op_gemm_tn_slab_elems(int dtype, int M, int N1, int N2);
int pmgt_op_gemm_tn(int dtype, const void* P, int64_t ldp, const void* Q, int64_t ldq, const int64_t* q_rows, int M,
                    int N1, int N2, float* slab, float* out, int accumulate, const int* m_dev, void* stream);
/* the same with the bias gradient (column sums of P; bias_slab: [512][N1] scratch) and the head-major row permutation of
 * the Q|K|V|C projection (perm_d = hidden size, perm_dh = head size; 0 = none) */
int pmgt_op_gemm_tn_bias(int dtype, const void* P, int64_t ldp, const void* Q, int64_t ldq, int M, int N1, int N2, float* slab,
                         float* out, float* bias_slab, float* bias_out, int perm_d, int perm_dh, void* stream);
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
                   const float* ln_beta, float ln_eps, void* stream);
/* A/B switch: 1 runs the last layer on every token even when last_hidden is not requested */
void pmgt_debug_disable_last_layer_shortcut(int on);
/* The engine owns a second HIP stream for work that is off the dependent chain: the partial-sum reductions of the
 * backward pass (weight-gradient slabs, bias and LayerNorm partials; double-buffered, fork/join by events) and the token
 * sort of the table mode.  Everything is joined before the call returns control of `stream`.  The sort always
 * overlaps the forward pass; the reductions move only with on = 1 (or PMGT_OVERLAP=1 at engine creation): measured
 * neutral on MI355X (the launch queue already hides them). */
void pmgt_engine_set_overlap(pmgt_engine* e, int on);
/* A/B switch: 1 selects the LDS-DMA variant of the tiled NT kernel (default: register-staged; same speed) */
void pmgt_debug_enable_nt_dma(int on);
/* A/B switch: 1 forces the register-staged tiled GEMM kernels everywhere (no streaming, no LDS-DMA) */
void pmgt_debug_force_tile_gemm(int on);
/* A/B switch: 1 routes bf16 attention through the generic fp32-VALU kernel instead of the MFMA one */
void pmgt_debug_force_valu_attention(int on);
/* A/B switch: 1 = one wave per (sequence, head) in the MFMA attention backward instead of NT cooperating waves */
void pmgt_debug_disable_coop_attention_bwd(int on);
/* Fused Q|K|V|C projection + attention forward (bf16; S = 32, dh = 32, hidden 128 or 256; returns -3 otherwise):
 * x [n_seq*S, d], w [4d, d] (rows q | k | v | c), bias [4d] fp32 -> qkvc [n_seq*S, 4d], ctx [n_seq*S, d]. */
int pmgt_op_qkvc_attention_fwd(const void* x, const void* w, const float* bias, const float* mask, void* qkvc, void* ctx,
                               int n_seq, int S, int H, int dh, float beta, float drop_p, uint32_t site1, uint32_t site2,
                               const uint64_t* rng, void* stream);
/* A/B switch: 1 projects features per token even when the whole table is smaller than half the batch's tokens */
void pmgt_debug_disable_table_projection(int on);
/* A/B switch: 1 keeps the per-token weight-gradient GEMM of the feature projection in table mode (no segment sums) */
void pmgt_debug_disable_segment_sum(int on);
/* A/B switch: 1 keeps Q|K|V|C in q | k | v | c column order between the fused forward and the attention backward
 * (default in training: head-major, 4 * dh contiguous elements per (row, head)) */
void pmgt_debug_disable_head_major(int on);
/* A/B switch (fp8 mode): 1 = layer inputs are quantised by their consumer (inside the fused projection + attention kernel)
 * instead of by the kernel that produces them (fused-LayerNorm epilogue of the FFN2 GEMM, embed_mix); bit-identical results */
void pmgt_debug_disable_producer_quant(int on);
/* A/B switch: 1 sums every set of partial sums of the backward pass (weight-gradient slabs, bias and LayerNorm partials) with a
 * launch of its own right after its producer, instead of one batched launch per gradient bucket */
void pmgt_debug_disable_deferred_reductions(int on);
/* A/B switch: 1 makes every LayerNorm site store its input for the backward pass; by default the sites whose LayerNorm runs in the
 * epilogue of the streaming GEMM (bf16, hidden size 256) do not, and their backward takes the normalised row from the LayerNorm
 * OUTPUT: x^ = (y - beta) / gamma */
void pmgt_debug_disable_layernorm_from_output(int on);
/* A/B switch: 1 keeps the attention backward and the Q|K|V|C weight gradient as two kernels */
void pmgt_debug_disable_fused_attention_backward(int on);
/* A/B switch: 1 keeps the projection GEMM and the attention as two kernels */
void pmgt_debug_disable_fused_qkvc_attention(int on);
int pmgt_op_attention_fwd(int dtype, const void* qkvc, const float* mask, void* ctx, float* probs, int n_seq, int S,
                          int H, int dh, float beta, float drop_p, uint32_t site1, uint32_t site2,
                          const uint64_t* rng, void* stream);
int pmgt_op_attention_bwd(int dtype, const void* qkvc, const float* mask, const void* dctx, void* dqkvc, int n_seq,
                          int S, int H, int dh, float beta, float drop_p, uint32_t site1, uint32_t site2,
                          const uint64_t* rng, void* stream);
/* Attention backward fused with the weight / bias gradient of the Q|K|V|C projection (bf16, S = 32, head size 32, hidden 128 or
 * 256; replaces pmgt_op_attention_bwd + the [M, 4d]^T [M, d] weight-gradient GEMM, i.e. autograd through
 * pmgt/pmgt/modeling_pmgt.py:429-433 and :435-526).  x = the layer input [n_seq * 32, d]; dqkvc as pmgt_op_attention_bwd;
 * slab [parts][4d * d] / bias_slab [parts][4d] (parts = pmgt_op_attention_bwd_wgrad_parts(H)) receive per-workgroup partial
 * sums in q | k | v | c row order, to be added up by the caller.  head_major = the column layout of qkvc / dqkvc. */
int pmgt_op_attention_bwd_wgrad_parts(int H);
int pmgt_op_attention_bwd_wgrad(const void* qkvc, const float* mask, const void* dctx, const void* x, void* dqkvc, float* slab,
                                float* bias_slab, int n_seq, int H, float beta, float drop_p, uint32_t site1, uint32_t site2,
                                const uint64_t* rng, int head_major, void* stream);

/* ---- host MCNSampling (libpmgt_sampler.so; pure host code, no HIP) --------------------------------
 * Replaces _sample_context_neigh / get_input_tensor / PMGTDataset.__getitem__ / pmgt_collate_fn
 * (pmgt/pmgt/datasets.py:14-208).  The graph is an ordered adjacency in CSR form: node ids 2..N+1
 * (0 = <pad>, 1 = <mask>), indptr has N+3 entries (rows 0 and 1 empty), neighbours in networkx
 * insertion order, float64 edge weights. */
typedef struct pmgt_sampler pmgt_sampler;
pmgt_sampler* pmgt_sampler_create(int64_t n_nodes, const int64_t* indptr, const int64_t* indices,
                                  const double* weights, const int* hop_sizes, int n_hops, int max_ctx_neigh,
                                  int max_total_samples, int min_neg_samples);
void pmgt_sampler_destroy(pmgt_sampler* s);
const char* pmgt_sampler_last_error(void);
/* np.random.seed(seed) of the reference's process-global legacy MT19937 stream (pmgt/utils/base.py:37). */
void pmgt_sampler_seed(pmgt_sampler* s, uint32_t seed);
/* One context: ids[S] (target first), mask[S]; returns num_ctx or <0 (datasets.py:64-79). */
int pmgt_sampler_context(pmgt_sampler* s, int64_t target, int64_t* ids, float* mask);
/* Collated batch in dataset order from ONE sequential stream (bit-exact with the reference run with
 * num_workers=0).  mode: 0 train, 1 eval, 2 inference.  pair buffers sized n*max_pairs(mode) rows.
 * Returns total pairs or <0. */
int pmgt_sampler_batch(pmgt_sampler* s, const int64_t* targets, int n, int mode, int64_t* tgt_ids, float* tgt_mask,
                       int64_t* pair_ids, float* pair_mask, int64_t* num_pairs, float* labels);
/* Same batch layout, sampled by n_threads host threads; every target gets its own stream seeded from
 * (base_seed, counter), so results do not depend on the thread count (statistical, not bit, parity
 * with the reference — the reference's own worker streams depend on the torch version). */
int pmgt_sampler_batch_mt(pmgt_sampler* s, const int64_t* targets, int n, int mode, uint64_t base_seed,
                          uint64_t counter, int n_threads, int64_t* tgt_ids, float* tgt_mask, int64_t* pair_ids,
                          float* pair_mask, int64_t* num_pairs, float* labels);
int pmgt_sampler_max_pairs(const pmgt_sampler* s, int mode);
/* legacy-stream primitives exposed for tests */
double pmgt_sampler_random_sample(pmgt_sampler* s);
int64_t pmgt_sampler_randint(pmgt_sampler* s, int64_t n);
/* sklearn train_test_split(arange(2, N+2), test_size, random_state=seed) (pmgt/pmgt/trainer.py:45-52) */
int pmgt_train_valid_split(int64_t n_nodes, double valid_size, uint32_t seed, int64_t* train_out, int64_t* valid_out);

#ifdef __cplusplus
}
#endif
#endif /* PMGT_CAPI_H */
