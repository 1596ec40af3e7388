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

/* Modalities: PMGTEmbeddings / PMGTNodeConstructLoss / PMGT.feat_embeddings are generic over len(feat_hidden_sizes)
 * (pmgt/pmgt/modeling_pmgt.py:163-173,195-201,549-569; pmgt/pmgt/models.py:38-54); the reference's trainer builds two
 * (visual 1536, textual 768; pmgt/pmgt/trainer.py:114-125).  The engine takes 1 .. PMGT_MAX_FEATS of them. */
#define PMGT_MAX_FEATS 4

/* Mirrors PMGTConfig (pmgt/pmgt/configuration_pmgt.py:11-41). */
typedef struct pmgt_config {
    int hidden_size;
    int num_hidden_layers;
    int num_attention_heads;
    int intermediate_size;
    int n_feats;                     /* len(feat_hidden_sizes) */
    int feat_sizes[PMGT_MAX_FEATS];  /* feat_hidden_sizes (each a multiple of 8); entries past n_feats are ignored */
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
    const void* tables[PMGT_MAX_FEATS]; /* tables[m]: [n_nodes + 2, feat_sizes[m]] frozen features in the engine dtype
                                          * (PMGT.feat_embeddings, models.py:40-54); e4m3 bytes in fp8 mode */
    int64_t n_nodes;
    uint64_t* rng_state; /* device [2]: {seed, step}; drives dropout + NFR masking */
    /* PMGT_DTYPE_FP8 only: the tables are e4m3 bytes, feature value = byte value * table_scales[m] (pmgt_quantize_e4m3) */
    float table_scales[PMGT_MAX_FEATS];
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
/* PMGTModel.forward(*input_feat_embeds) on already-gathered features: feats[m] = [n_seq, S, feat_sizes[m]] in the engine
 * dtype, m < n_feats (a host array of device pointers). */
int pmgt_encode_feats(pmgt_engine* e, const pmgt_tensors* t, const void* const* feats,
                      const float* mask, int n_seq, int seq_len, void* last_hidden, void* hidden_states,
                      float* attn_probs, void* workspace, int64_t workspace_bytes, void* stream);

/* PMGTModel.forward + backward for a caller that owns the head (second caller of the boundary: PMGT_NCF,
 * pmgt/pmgt_ncf/models.py:77-105 -- `self.bert(*input_feat_embeds, attention_mask=...)[0][:, 0]` followed by
 * autograd).  pmgt_encode_train runs the encoder on node ids (ids != NULL, gather fused) or on gathered features
 * (ids == NULL: feats[m] = [n_seq, S, feat_sizes[m]], engine dtype), keeps every activation in `workspace`
 * (pmgt_workspace_bytes(e, n_seq, S, 1, 1) bytes, untouched until the backward call) and snapshots the dropout
 * counter; PMGT_FLAG_TRAINING turns dropout on and advances the counter.  pmgt_encode_backward takes
 * d loss / d last_hidden_state [n_seq, S, d] (engine dtype) and leaves the gradients of every `bert.*` entry in
 * t->grads (= or += with PMGT_FLAG_ACCUMULATE; pass the same TRAINING flag); the frozen tables get none
 * (pmgt/pmgt_ncf/models.py:45-47).  `feats` is needed again only when the forward used it (NULL otherwise). */
int pmgt_encode_train(pmgt_engine* e, const pmgt_tensors* t, const int64_t* ids, const void* const* feats,
                      const float* mask, int n_seq, int seq_len, void* last_hidden, void* workspace,
                      int64_t workspace_bytes, int flags, void* stream);
int pmgt_encode_backward(pmgt_engine* e, const pmgt_tensors* t, const void* const* feats,
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
 * options "side_stream_reduce" or "one_bucket" set, one bucket covering the whole buffer is reported at the end. */
typedef void (*pmgt_grad_ready_fn)(void* user, int64_t offset, int64_t numel);
void pmgt_engine_set_grad_ready_callback(pmgt_engine* e, pmgt_grad_ready_fn cb, void* user);

/* Per-phase timers (what the reference lacks entirely): between begin and end every group of
 * kernel launches is bracketed by HIP events on its stream; end() waits for them and writes one
 * "name count total_ms" line per phase into buf.  Off by default. */
int pmgt_profile_begin(pmgt_engine* e);
int pmgt_profile_end(pmgt_engine* e, char* buf, int cap);
/* the phases recorded so far, in launch order, one name per line (between begin and end; no wait) */
int pmgt_profile_sequence(pmgt_engine* e, char* buf, int cap);
/* the same records with their times, "name ms" per line in launch order (waits for the events; the records stay for pmgt_profile_end) */
int pmgt_profile_records(pmgt_engine* e, char* buf, int cap);

/* dtype plumbing */
int pmgt_cast_from_f32(int dtype, const float* src, void* dst, int64_t n, void* stream);
int pmgt_cast_to_f32(int dtype, const void* src, float* dst, int64_t n, void* stream);

/* fp8 mode plumbing: dst[i] = e4m3_rne(clamp(src[i] * inv_scale, +-448)) and back (n % 8 == 0).  The frozen tables are
 * quantised once by the caller with inv_scale = 448 / max|table|; table_scale = max|table| / 448. */
int pmgt_quantize_e4m3(const float* src, void* dst, int64_t n, float inv_scale, void* stream);
int pmgt_dequantize_e4m3(const void* src, float* dst, int64_t n, float scale, void* stream);

/* Path options: PER-ENGINE switches between the fused / streaming kernels of the product path and their plain
 * counterparts (parity A/B, bisecting).  key = one of the names listed in include/pmgt_ops.h ("store_ln_input",
 * "no_fused_attention_bwd", ...), value 0 = product path (default), 1 = alternative.  Returns 0, or -2 for an unknown key.
 * Options are state of THIS engine only: two engines in one process do not see each other's choices, and nothing in the
 * library reads the environment.  Set them before the first pmgt_workspace_bytes / step call of a shape (workspace
 * carving depends on some of them). */
int pmgt_engine_set_option(pmgt_engine* e, const char* key, int value);
/* current value (0 / 1), or -2 for an unknown key */
int pmgt_engine_get_option(const pmgt_engine* e, const char* key);

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
/* Same batch layout, sampled by n_threads host threads; target i gets its own stream seeded from
 * (base_seed, counter + counter_stride * i), so results do not depend on the thread count (statistical, not bit, parity
 * with the reference — the reference's own worker streams depend on the torch version).  counter_stride = 1
 * for consecutive items; a rank that evaluates items r, r + W, r + 2W, ... of a list passes counter = r (+ W * offset)
 * and counter_stride = W and so draws exactly what a single process draws for the same items. */
int pmgt_sampler_batch_mt(pmgt_sampler* s, const int64_t* targets, int n, int mode, uint64_t base_seed,
                          uint64_t counter, uint64_t counter_stride, int n_threads, int64_t* tgt_ids, float* tgt_mask, int64_t* pair_ids,
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
