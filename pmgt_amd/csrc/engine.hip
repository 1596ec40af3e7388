// Orchestration of one PMGT pre-training step on one MI355X and the C ABI (include/pmgt_capi.h).
//
// One call = one stream-ordered chain of kernel launches (no host sync, no allocation): the whole
// step (mask -> mirror -> embeddings -> L layers -> GSR/NFR -> backward) can therefore be captured
// in a hipGraph by the caller.  All B(1 + pairs + 1) sequences of a step go through ONE batched
// encoder pass (the reference makes B+2 separate calls, pmgt/pmgt/models.py:93,113,153).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <map>
#include <vector>

#include "../../include/pmgt_ops.h"
#include "attention.h"
#include "fp8.h"
#include "gemm.h"
#include "loss.h"
#include "optim.h"
#include "rowops.h"
#include "segsum.h"

namespace pmgt {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local int64_t g_launch_count[LT_COUNT] = {};
void note_launch(int tag) { if (tag >= 0 && tag < LT_COUNT) ++g_launch_count[tag]; }
static const char* const g_launch_names[LT_COUNT] = {
    "gemm_wsr", "gemm_wsr_lnb", "gemm_wsr512", "gemm_ws", "nt_big", "nt_big_gather", "nt_big_128", "nt_lnb", "nt_tile",
    "tn_big", "tn_big_gather", "tn_dma", "tn_dma_gather", "tn_tile", "attn_tiles_fwd", "attn_tiles_bwd", "qkvc_attn_fwd",
    "attn_bwd_wgrad", "f8_big", "f8_tile", "f8_wsr512", "gemm_rowln", "nt_lnf", "embed_tok8", "qkvc_attn_fwd_vc", "attn_bwd_wgrad_vc", "nt_vc", "attn_bwd_wgrad_vc2"};

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Optional per-phase timers: HIP events recorded on the launch stream around each group of kernels.
// Off by default (zero overhead); bench.py turns them on for a separate, untimed pass.
struct Profiler {
    bool on = false;
    struct Rec { const char* name; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
    void clear() {
        for (auto& r : recs) { pool.push_back(r.a); pool.push_back(r.b); }
        recs.clear();
    }
};

struct ParamEntry {
    std::string name;
    int64_t offset, numel;
    int rows, cols, decay;
};

struct LayerOff {
    int64_t Wqkvc, bqkvc, Wo, bo, ln1g, ln1b, W1, b1, W2, b2, ln2g, ln2b;
    // mirror offsets (elements of T); -1 = not mirrored (fp32 reads the master copy)
    int64_t mWqkvc, mWqkvcT, mWo, mWoT, mW1, mW1T, mW2, mW2T;
    int64_t mWqkvcT_hm = -1;     // transposed copy with the 4d index in head-major order (dgrad of a head-major Q|K|V|C)
    int64_t m8Wqkvc = -1, s8Wqkvc = -1;   // fp8 mode: byte offset of the e4m3 copy / offset of its per-channel scales
};

}  // namespace pmgt

using namespace pmgt;

struct pmgt_engine {
    pmgt_config cfg;
    int d, L, H, I, dh;
    // modalities (PMGTConfig.feat_hidden_sizes): sizes, their sum, and each one's first column in the concatenated NFR projection
    int NF = 0, F[PMGT_MAX_FEATS] = {0, 0, 0, 0}, Fsum = 0, Foff[PMGT_MAX_FEATS] = {0, 0, 0, 0};
    // flat parameter layout (floats); Wf[m] = feat_linear.m.weight, bf = the NF feat_linear biases side by side
    int64_t pos, role, Wf[PMGT_MAX_FEATS], bf, ln_g, ln_b, Wa, ba, Wn, bn, total;
    std::vector<LayerOff> layers;
    std::vector<ParamEntry> entries;
    // mirror
    int64_t mWf[PMGT_MAX_FEATS], mWn, mWnT, mirror_elems;
    // fp8 mode (PMGT_DTYPE_FP8): bf16 engine + e4m3 copies (per-output-channel scales) of the feature-projection and
    // Q|K|V|C weights, rebuilt every forward by one launch; the frozen tables arrive as e4m3 from the caller
    bool fp8 = false;
    int64_t m8Wf[PMGT_MAX_FEATS] = {-1, -1, -1, -1}, s8Wf[PMGT_MAX_FEATS] = {-1, -1, -1, -1}, mirror8_bytes = 0, mscale_elems = 0;
    std::vector<QuantDesc> desc8;
    QuantDesc* desc8_dev = nullptr;
    int desc8_rows = 0;
    std::vector<MirrorDesc> desc;
    MirrorDesc* desc_dev = nullptr;
    int mirror_tiles = 0;
    const void* zeros = nullptr;      // device zero page (padding source of the LDS-DMA kernels)
    pmgt::Profiler prof;
    // Side stream: the partial-sum reductions of the backward pass and the token sort of the table mode run here
    // (fork/join with events); the reductions only with PMGT_OVERLAP=1 / pmgt_engine_set_overlap(e, 1).
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> sync_ev;
    size_t sync_next = 0;
    uint32_t opts = 0;        // PathOpt bits (pmgt_engine_set_option): per engine, read by every dispatch decision below
    // pmgt_encode_train -> pmgt_encode_backward are two calls over one workspace: the backward re-derives which buffers the forward
    // filled (head-major Q|K|V|C, unstored LayerNorm inputs, table mode ...) from the options, so it must see the SAME options.
    // The forward records its option bits per workspace; a backward over that workspace under different bits is refused
    // (it would read buffers the forward never wrote).  Bits that only move reductions / callbacks do not count.
    static constexpr uint32_t SCHED_OPTS = OPT_EAGER_REDUCE | OPT_SIDE_STREAM_REDUCE | OPT_ONE_BUCKET | OPT_SMALL_ARENA;
    std::map<const void*, uint32_t> fwd_opts;
    bool overlap() const { return (opts & OPT_SIDE_STREAM_REDUCE) != 0; }     // partial-sum reductions on the side stream (see SideReduce): measured neutral, opt-in
    // data-parallel exchange: called when a contiguous range of the flat gradient buffer is final in stream order
    pmgt_grad_ready_fn grad_cb = nullptr;
    void* grad_cb_user = nullptr;
    bool grad_cb_fine() const { return grad_cb != nullptr && !(overlap() && side != nullptr) && !(opts & OPT_ONE_BUCKET); }
    void grad_ready(int64_t off, int64_t numel) const { if (grad_cb && numel > 0) grad_cb(grad_cb_user, off, numel); }
    // Ring of fork / join events.  An event handed out may still be pending when the ring comes round to it again (side_last of a
    // side-stream weight gradient is held across several later next_sync() calls), so the ring must be longer than the number of events
    // one pass consumes: reserve_sync() grows it to the caller's bound before a pass starts.
    void reserve_sync(size_t n) {
        while (sync_ev.size() < n) {
            hipEvent_t ev = nullptr;
            (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            sync_ev.push_back(ev);
        }
    }
    hipEvent_t next_sync() {
        reserve_sync(256);
        return sync_ev[sync_next++ % sync_ev.size()];
    }
};

namespace pmgt {

static int64_t take(int64_t& cur, int64_t n) {
    int64_t o = cur;
    cur += align_up(n, 4);     // keep every tensor 16-byte aligned
    return o;
}

static void add_entry(pmgt_engine* e, const std::string& name, int64_t off, int rows, int cols) {
    ParamEntry p;
    p.name = name; p.offset = off; p.rows = rows; p.cols = cols;
    p.numel = (int64_t)rows * (cols > 0 ? cols : 1);
    p.decay = !(name.find("bias") != std::string::npos || name.find("LayerNorm.weight") != std::string::npos);
    e->entries.push_back(p);
}

static void build_layout(pmgt_engine* e) {
    const int d = e->d, I = e->I, NF = e->NF, P = e->cfg.max_position_embeddings;
    int64_t cur = 0;
    const std::string em = "bert.embeddings.";
    e->pos = take(cur, (int64_t)P * d);   add_entry(e, em + "position_embeddings.weight", e->pos, P, d);
    e->role = take(cur, 2 * d);           add_entry(e, em + "role_embeddings.weight", e->role, 2, d);
    for (int m = 0; m < NF; ++m) {
        e->Wf[m] = take(cur, (int64_t)d * e->F[m]);
        add_entry(e, em + "feat_linear." + std::to_string(m) + ".weight", e->Wf[m], d, e->F[m]);
    }
    e->bf = take(cur, NF * d);
    for (int m = 0; m < NF; ++m) add_entry(e, em + "feat_linear." + std::to_string(m) + ".bias", e->bf + (int64_t)m * d, d, 0);
    // ln_g | ln_b | Wa | ba are contiguous in exactly the order of embed_mix_bwd's partials (embed_part_elems(d, NF))
    e->ln_g = take(cur, d);               add_entry(e, em + "LayerNorm.weight", e->ln_g, d, 0);
    e->ln_b = take(cur, d);               add_entry(e, em + "LayerNorm.bias", e->ln_b, d, 0);
    e->Wa = take(cur, (int64_t)NF * NF * d);   add_entry(e, em + "attention.1.weight", e->Wa, NF, NF * d);
    e->ba = take(cur, 4);                 add_entry(e, em + "attention.1.bias", e->ba, NF, 0);
    e->layers.resize(e->L);
    for (int l = 0; l < e->L; ++l) {
        LayerOff& o = e->layers[l];
        const std::string p = "bert.encoder.layer." + std::to_string(l) + ".";
        o.Wqkvc = take(cur, (int64_t)4 * d * d);
        const char* nm[4] = {"query", "key", "value", "ctx_attention"};
        for (int k = 0; k < 4; ++k) add_entry(e, p + "attention.self." + nm[k] + ".weight", o.Wqkvc + (int64_t)k * d * d, d, d);
        o.bqkvc = take(cur, 4 * d);
        for (int k = 0; k < 4; ++k) add_entry(e, p + "attention.self." + nm[k] + ".bias", o.bqkvc + (int64_t)k * d, d, 0);
        // ln_g | ln_b | dense bias contiguous = the order of ln_bwd's three partials
        o.Wo = take(cur, (int64_t)d * d);  add_entry(e, p + "attention.output.dense.weight", o.Wo, d, d);
        o.ln1g = take(cur, d);             add_entry(e, p + "attention.output.LayerNorm.weight", o.ln1g, d, 0);
        o.ln1b = take(cur, d);             add_entry(e, p + "attention.output.LayerNorm.bias", o.ln1b, d, 0);
        o.bo = take(cur, d);               add_entry(e, p + "attention.output.dense.bias", o.bo, d, 0);
        o.W1 = take(cur, (int64_t)I * d);  add_entry(e, p + "intermediate.dense.weight", o.W1, I, d);
        o.b1 = take(cur, I);               add_entry(e, p + "intermediate.dense.bias", o.b1, I, 0);
        o.W2 = take(cur, (int64_t)d * I);  add_entry(e, p + "output.dense.weight", o.W2, d, I);
        o.ln2g = take(cur, d);             add_entry(e, p + "output.LayerNorm.weight", o.ln2g, d, 0);
        o.ln2b = take(cur, d);             add_entry(e, p + "output.LayerNorm.bias", o.ln2b, d, 0);
        o.b2 = take(cur, d);               add_entry(e, p + "output.dense.bias", o.b2, d, 0);
    }
    // the NF projections of the NFR head stacked: one [sum F_m, d] weight = one GEMM over the masked rows
    e->Wn = take(cur, (int64_t)e->Fsum * d);
    for (int m = 0; m < NF; ++m) add_entry(e, "nfr_loss.projections." + std::to_string(m) + ".weight", e->Wn + (int64_t)e->Foff[m] * d, e->F[m], d);
    e->bn = take(cur, e->Fsum);
    for (int m = 0; m < NF; ++m) add_entry(e, "nfr_loss.projections." + std::to_string(m) + ".bias", e->bn + e->Foff[m], e->F[m], 0);
    e->total = cur;

    // ---- mirror layout
    const bool half = e->cfg.dtype != PMGT_DTYPE_F32;
    int64_t mc = 0;
    int tiles = 0;
    auto add_m = [&](int64_t src, int rows, int cols, bool copy, bool transpose, int64_t* dst, int64_t* dst_t, int64_t* dst_t_hm = nullptr) {
        MirrorDesc m;
        m.src = src; m.rows = rows; m.cols = cols;
        m.dst = copy ? mc : -1;
        if (copy) mc += align_up((int64_t)rows * cols, 8);
        m.dst_t = transpose ? mc : -1;
        if (transpose) mc += align_up((int64_t)rows * cols, 8);
        m.dst_t_hm = -1; m.hm_d = 1; m.hm_dh = 1;
        if (dst_t_hm && transpose) {
            m.dst_t_hm = mc;
            mc += align_up((int64_t)rows * cols, 8);
            m.hm_d = e->d; m.hm_dh = e->dh;
            *dst_t_hm = m.dst_t_hm;
        }
        m.tile_start = tiles;
        tiles += cdiv(rows, 32) * cdiv(cols, 32);
        if (dst) *dst = m.dst;
        if (dst_t) *dst_t = m.dst_t;
        if (copy || transpose) e->desc.push_back(m);
        else tiles = m.tile_start;
    };
    for (int m = 0; m < NF; ++m) add_m(e->Wf[m], d, e->F[m], half, false, &e->mWf[m], nullptr);
    for (int l = 0; l < e->L; ++l) {
        LayerOff& o = e->layers[l];
        add_m(o.Wqkvc, 4 * d, d, half, true, &o.mWqkvc, &o.mWqkvcT, half ? &o.mWqkvcT_hm : nullptr);
        add_m(o.Wo, d, d, half, true, &o.mWo, &o.mWoT);
        add_m(o.W1, I, d, half, true, &o.mW1, &o.mW1T);
        add_m(o.W2, d, I, half, true, &o.mW2, &o.mW2T);
    }
    add_m(e->Wn, e->Fsum, d, half, true, &e->mWn, &e->mWnT);
    e->mirror_elems = mc;
    e->mirror_tiles = tiles;
    if (e->fp8) {
        int64_t bc = 0, sc = 0;
        int rows = 0;
        auto add_q = [&](int64_t src, int r, int c, int64_t* dst, int64_t* scale) {
            QuantDesc q;
            q.src = src; q.dst = bc; q.scale = sc; q.rows = r; q.cols = c; q.row_start = rows;
            *dst = bc; *scale = sc;
            bc += align_up((int64_t)r * c, 16);
            sc += align_up(r, 4);
            rows += r;
            e->desc8.push_back(q);
        };
        for (int m = 0; m < NF; ++m) add_q(e->Wf[m], d, e->F[m], &e->m8Wf[m], &e->s8Wf[m]);
        for (int l = 0; l < e->L; ++l) add_q(e->layers[l].Wqkvc, 4 * d, d, &e->layers[l].m8Wqkvc, &e->layers[l].s8Wqkvc);
        e->mirror8_bytes = bc; e->mscale_elems = sc; e->desc8_rows = rows;
    }
}

// ---- workspace carving ---------------------------------------------------------------------------
struct Carver {
    char* base;
    int64_t cur = 0;
    explicit Carver(void* b) : base((char*)b) {}
    void* raw(int64_t bytes) {
        void* p = base ? base + cur : nullptr;
        cur += align_up(std::max<int64_t>(bytes, 16), 256);
        return p;
    }
    template <typename U> U* get(int64_t n) { return (U*)raw(n * (int64_t)sizeof(U)); }
};

template <typename T> struct LayerBufs {
    T *qkvc, *ctx, *ao_pre, *u, *ff_pre, *g, *fo_pre, *hout;
    float *stats1, *stats2;
};

template <typename T> struct Bufs {
    int64_t* rng_snap;  // [2] (seed, step) of the forward, replayed by pmgt_encode_backward
    int64_t* ids;      // [Tseq, S] concatenated node ids
    float* mask;       // [Tseq, S]
    T* mirror;
    char* mirror8 = nullptr;   // fp8 mode: e4m3 weight copies
    float* mscale = nullptr;   //           their per-channel scales
    char* x8 = nullptr;        //           per-row quantised layer input of the unfused Q|K|V|C projection [M, d]
    float* xscale = nullptr;   //           [M]
    T *E, *emb_pre, *h0;
    hipEvent_t sort_done = nullptr;   // set when the token sort of this step already runs on the side stream
    // dense weight gradients of a layer on the side stream (option side_stream_wgrad; nothing in the data-gradient chain depends on them): side_last
    // = "this wgrad has finished" (the caller waits for it before a launch that overwrites its operands), side_pending = something on the
    // side stream has written arena regions since the last join
    bool side_wgrad = false, side_pending = false;
    hipEvent_t side_last = nullptr;
    bool qkvc_hm = false;   // Q|K|V|C (and its gradient) are stored head-major (fused forward + one-wave MFMA backward)
    bool qkvc_vc = false;   // beta == 1: only the V | C columns of Q|K|V|C (and of its gradient) exist (vc_only_applies)
    bool e_by_id = false;   // E holds the projection of the whole table (rows = node ids) instead of one row per token
    float *a, *emb_stats;
    std::vector<LayerBufs<T>> layer;
    // backward temporaries
    T *bA, *bB, *bC, *bD, *big;
    float *slab, *part;     // gemm_tn slabs; LN / embed / colsum partials
    // Deferred reductions: every producer of partial sums takes its own region of `arena` and queues a job; the jobs of a gradient
    // bucket (NFR head, a layer, the embeddings) are summed by ONE multi_reduce launch (two with > 128 slabs) before the bucket is
    // reported ready, instead of a launch per producer (40 per step, 0.43 ms at every batch size).
    float* arena = nullptr;
    int64_t arena_elems = 0, arena_cur = 0;
    bool defer = false;
    std::vector<ReduceJob> pend;
    // segment-sum scratch (table mode backward)
    uint32_t *sg_keys = nullptr, *sg_vals = nullptr, *sg_skeys = nullptr, *sg_perm = nullptr;
    int* sg_off = nullptr;
    void* sg_tmp = nullptr;
    int64_t sg_tmp_bytes = 0;
    float* sg_part = nullptr;
    float* part_side;       // bias partials of the wgrad kernels
    // double-buffered partials + completion events of their side-stream reductions
    int64_t slab_elems = 0, part_side_elems = 0, ln_part_elems = 0;
    float* ln_part = nullptr;
    int wg_idx = 0, ln_idx = 0;
    hipEvent_t wg_done[2] = {nullptr, nullptr}, ln_done[2] = {nullptr, nullptr};
    float* possum;
    // losses
    int* off;
    float *gsr_part, *sse_part;
    int64_t *nfr_masked, *nfr_tgt, *nfr_rows, *nfr_tids;
    int* nfr_count;
    T *pred, *dq;
    // last-layer shortcut (training fast path): only the rows the loss reads go through the last layer's
    // attn-out / FFN blocks.  Compact row order: target CLS (B), pair CLS (P), masked rows (count).
    int rcap = 0;
    int64_t* need_rows = nullptr;
    int* need_cnt = nullptr;
    int* need_inv = nullptr;      // [M] token row -> compact row or -1 (the residual of the shortcut layer, read by the fused dX + LayerNorm-backward tile)
    LayerBufs<T> ctail;
    T *c_dh = nullptr, *c_bB = nullptr, *c_bC = nullptr, *c_bD = nullptr, *c_big = nullptr;
};

static int64_t sort_temp_bytes(int M) {      // rocPRIM's size query, cached per token count
    static std::map<int, int64_t> cache;
    auto itr = cache.find(M);
    if (itr != cache.end()) return itr->second;
    const int64_t v = seg_sort_temp_bytes(M);
    cache[M] = v;
    return v;
}

static int64_t tn_slab_elems(int dtype, int M, int N1, int N2, uint32_t opts) {
    const int bkm = dtype != PMGT_DTYPE_F32 ? 64 : 32;
    return (int64_t)gemm_tn_pick_splits(M, N1, N2, bkm, opts) * N1 * N2;
}

template <typename T>
static void carve(const pmgt_engine* e, Carver& c, Bufs<T>& b, int Tseq, int S, int B, bool training) {
    const int d = e->d, I = e->I, L = e->L, F = e->Fsum, NF = e->NF;
    const int64_t M = (int64_t)Tseq * S;
    b.rng_snap = c.get<int64_t>(2);
    b.ids = c.get<int64_t>(M);
    b.mask = c.get<float>(M);
    b.mirror = c.get<T>(e->mirror_elems);
    if (e->fp8) {
        b.mirror8 = c.get<char>(e->mirror8_bytes);
        b.mscale = c.get<float>(e->mscale_elems);
        b.x8 = c.get<char>(M * d);
        b.xscale = c.get<float>(M);
    }
    b.E = c.get<T>(M * std::max(NF, 2) * d);
    b.a = c.get<float>(M * NF);
    b.emb_pre = c.get<T>(M * d);
    b.emb_stats = c.get<float>(M * 2);
    b.h0 = c.get<T>(M * d);
    const int nl = training ? L : std::min(L, 2);
    b.layer.resize(L);
    for (int l = 0; l < nl; ++l) {
        LayerBufs<T>& lb = b.layer[l];
        lb.qkvc = c.get<T>(M * 4 * d);
        lb.ctx = c.get<T>(M * d);
        lb.ao_pre = c.get<T>(M * d);
        lb.stats1 = c.get<float>(M * 2);
        lb.u = c.get<T>(M * d);
        lb.ff_pre = c.get<T>(M * I);
        lb.g = c.get<T>(M * I);
        lb.fo_pre = c.get<T>(M * d);
        lb.stats2 = c.get<float>(M * 2);
        lb.hout = c.get<T>(M * d);
    }
    for (int l = nl; l < L; ++l) b.layer[l] = b.layer[l & 1];
    // losses (eval needs GSR only)
    b.off = c.get<int>(B + 1);
    b.gsr_part = c.get<float>(std::max(B, 1));
    b.nfr_count = c.get<int>(4);
    if (!training) return;
    const int cap = B * std::max(S - 1, 1);
    b.bA = c.get<T>(M * d);
    b.bB = c.get<T>(M * d);
    b.bC = c.get<T>(M * d);
    b.bD = c.get<T>(M * d);
    b.big = c.get<T>(M * std::max(I, 4 * d));
    int64_t slab = 0;
    const int dt = e->cfg.dtype;
    slab = std::max(slab, tn_slab_elems(dt, (int)M, 4 * d, d, e->opts));
    slab = std::max(slab, (int64_t)attn_bwd_wgrad_parts(e->H) * 4 * d * d);          // fused attention backward + weight gradient
    slab = std::max(slab, tn_slab_elems(dt, (int)M, d, d, e->opts));
    slab = std::max(slab, tn_slab_elems(dt, (int)M, I, d, e->opts));
    slab = std::max(slab, tn_slab_elems(dt, (int)M, d, I, e->opts));
    for (int m = 0; m < NF; ++m) slab = std::max(slab, tn_slab_elems(dt, (int)M, d, e->F[m], e->opts));
    slab = std::max(slab, tn_slab_elems(dt, std::max(256, cap / 5), F, d, e->opts));
    b.slab_elems = align_up(slab, 64);
    b.slab = c.get<float>(2 * b.slab_elems);
    int64_t part = 0;
    part = std::max(part, (int64_t)ln_bwd_parts((int)M) * 3 * d);
    part = std::max(part, (int64_t)embed_bwd_parts((int)M) * embed_part_elems(d, NF));
    part = std::max(part, colsum_slab_elems((int)M, std::max(I, 4 * d)));
    part = std::max(part, colsum_slab_elems(Tseq, S * d));
    part = std::max(part, colsum_slab_elems(cap, F));
    part = std::max(part, (int64_t)512 * std::max(std::max(I, 4 * d), F));     // wgrad bias slabs [splits <= 512][N1]
    b.part = c.get<float>(part);
    b.part_side_elems = (int64_t)512 * std::max(std::max(I, 4 * d), F);
    b.part_side = c.get<float>(2 * b.part_side_elems);
    b.ln_part_elems = align_up((int64_t)ln_bwd_parts((int)M) * 3 * d, 64);
    b.ln_part = c.get<float>(2 * b.ln_part_elems);
    b.arena_elems = 2 * b.slab_elems + 4 * b.part_side_elems + 2 * b.ln_part_elems + align_up(part, 64);
    if (e->opts & OPT_SMALL_ARENA)      // room for the largest single producer (slab + bias slab, or one set of row partials) and nothing more
        b.arena_elems = std::max(b.slab_elems + b.part_side_elems, std::max(b.ln_part_elems, align_up(part, 64))) + 128;
    b.arena = c.get<float>(b.arena_elems);
    b.defer = !(e->opts & OPT_EAGER_REDUCE) && !(e->overlap() && e->side != nullptr);      // side-stream reductions keep the per-producer launches
    b.sg_keys = c.get<uint32_t>(M); b.sg_vals = c.get<uint32_t>(M); b.sg_skeys = c.get<uint32_t>(M); b.sg_perm = c.get<uint32_t>(M);
    b.sg_off = c.get<int>(M / 2 + 4);                     // table mode implies N + 2 <= M / 2
    b.sg_tmp_bytes = sort_temp_bytes((int)M);
    b.sg_tmp = c.raw(b.sg_tmp_bytes);
    b.sg_part = c.get<float>(seg_part_elems((int)M, 2 * d));
    b.possum = c.get<float>((int64_t)S * d);
    b.sse_part = c.get<float>((int64_t)nfr_diff_parts(cap) * MAX_FEATS);
    b.nfr_masked = c.get<int64_t>((int64_t)B * S);
    b.nfr_tgt = c.get<int64_t>((int64_t)B * S);
    b.nfr_rows = c.get<int64_t>(cap);
    b.nfr_tids = c.get<int64_t>(cap);
    b.pred = c.get<T>((int64_t)cap * F);
    b.dq = c.get<T>((int64_t)cap * d);
    const int64_t R = (int64_t)B + (Tseq - 2 * B) + cap;      // B targets + P pairs + masked rows (upper bound)
    b.rcap = (int)R;
    b.need_rows = c.get<int64_t>(R);
    b.need_cnt = c.get<int>(4);
    b.need_inv = c.get<int>(M);
    b.ctail.qkvc = nullptr; b.ctail.ctx = nullptr;
    b.ctail.ao_pre = c.get<T>(R * d);
    b.ctail.stats1 = c.get<float>(R * 2);
    b.ctail.u = c.get<T>(R * d);
    b.ctail.ff_pre = c.get<T>(R * I);
    b.ctail.g = c.get<T>(R * I);
    b.ctail.fo_pre = c.get<T>(R * d);
    b.ctail.stats2 = c.get<float>(R * 2);
    b.ctail.hout = c.get<T>(R * d);
    b.c_dh = c.get<T>(R * d);
    b.c_bB = c.get<T>(R * d);
    b.c_bC = c.get<T>(R * d);
    b.c_bD = c.get<T>(R * d);
    b.c_big = c.get<T>(R * std::max(I, d));
}

template <typename T> static inline const T* wsel(const pmgt_engine* e, const pmgt_tensors* t, const Bufs<T>& b, int64_t master_off, int64_t mirror_off) {
    if (mirror_off >= 0) return b.mirror + mirror_off;
    return (const T*)(t->params + master_off);     // only reachable when T == float
}

static inline DropCfg dropcfg(const pmgt_tensors* t, bool on, float p, int layer, int kind) {
    DropCfg c;
    c.rng = t->rng_state;
    c.p = on ? p : 0.f;
    c.site = site_id(layer, kind);
    return c;
}

#define RUN(x)                                                                                    \
    do {                                                                                          \
        int rc__ = (x);                                                                           \
        if (rc__ != 0) return rc__;                                                               \
    } while (0)
// RUN with an optional event bracket (phase timers)
#define RUNP(name, x)                                                                             \
    do {                                                                                          \
        Profiler& pf__ = const_cast<pmgt_engine*>(e)->prof;                                       \
        hipEvent_t ea__ = nullptr, eb__ = nullptr;                                                \
        if (pf__.on) { ea__ = pf__.get(); eb__ = pf__.get(); (void)hipEventRecord(ea__, st); }     \
        int rc__ = (x);                                                                           \
        if (pf__.on) { (void)hipEventRecord(eb__, st); pf__.recs.push_back({name, ea__, eb__}); }  \
        if (rc__ != 0) return rc__;                                                               \
    } while (0)

// ---- linear layer dispatcher: weight-stationary streaming kernel when it applies (bf16, K <= 256), else the
// tiled kernel (+ a separate LayerNorm launch when the caller asked for the fused one) -------------------
static void* g_zero_page = nullptr;
static const void* zero_page() {
    if (!g_zero_page) {
        if (hipMalloc(&g_zero_page, 4096) != hipSuccess || hipMemset(g_zero_page, 0, 4096) != hipSuccess) g_zero_page = nullptr;
    }
    return g_zero_page;
}
// Whether the LayerNorm site behind a [Mt, K] x [d, K]^T dense block keeps NO copy of its input: the fused-LayerNorm streaming GEMM then
// skips the store of the pre-LayerNorm sum (ao_pre / fo_pre hold nothing) and the backward takes x^ from the LayerNorm OUTPUT (u / hout).
// A pure function of configuration, shape and options -- the forward and a SEPARATE backward call (pmgt_encode_backward carves its own
// Bufs) must take the same decision; linear() refuses (loudly) a skip_c it cannot honour.
template <typename T>
static inline bool ln_from_y_applies(const pmgt_engine* e, int Mt, int K, bool compacted) {
    if (sizeof(T) != 2 || compacted || (e->opts & (OPT_STORE_LN_INPUT | OPT_TILE_GEMM | OPT_UNFUSED_LN))) return false;
    if (e->d != 256) return false;
    if (K > 512) return !e->fp8 && gemm_nt_lnf_shape(Mt, 256, K);      // whole-row 256 x 256 tile with the LayerNorm epilogue (gemm_nt_lnf)
    return Mt >= 64 && K >= 64 && (K & (K - 1)) == 0;
}

template <typename T>
static int linear(const pmgt_engine* e, const char* name, const GemmWS& g, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        if (g.ln_out && gemm_rowln_ok(g)) {      // N = 512: whole rows in one workgroup, LayerNorm in the epilogue (gemm_rowln.hip)
            RUNP(name, gemm_rowln(g, st));
            return 0;
        }
        if (g.ln_out && gemm_nt_lnf_ok(g)) {     // N = 256, K > 512: whole rows in the 256 x 256 tile, LayerNorm behind its main loop (gemm.hip)
            RUNP(name, gemm_nt_lnf(g, st));
            return 0;
        }
        if (!(g.opts & OPT_TILE_GEMM) && gemm_ws_supported(g)) {
            PMGT_CHECK(!g.skip_c || gemm_ws_fuses_ln(g), -2, "linear: skip_c needs the fused-LayerNorm form");
            RUNP(name, gemm_ws(g, st));
            if (g.ln_out && !gemm_ws_fuses_ln(g))
                RUNP("fwd.layernorm", ln_fwd<T>((const T*)g.C, (T*)g.ln_out, g.ln_stats, g.ln_gamma, g.ln_beta, g.M, g.N, g.ln_eps,
                                                DropCfg{nullptr, 0.f, 0}, st, g.m_dev, g.q8, g.q8_scale));
            return 0;
        }
    }
    PMGT_CHECK(!g.skip_c, -2, "linear: skip_c needs the fused-LayerNorm form");
    RUNP(name, gemm_nt<T>(g, st));
    if (g.ln_out)
        RUNP("fwd.layernorm", ln_fwd<T>((const T*)g.C, (T*)g.ln_out, g.ln_stats, g.ln_gamma, g.ln_beta, g.M, g.N, g.ln_eps,
                                        DropCfg{nullptr, 0.f, 0}, st, g.m_dev, g.q8, g.q8_scale));
    return 0;
}

// weight copies of a forward pass: bf16 W / W^T mirror (+ the e4m3 copies of the fp8 mode)
template <typename T>
static int build_mirrors(const pmgt_engine* e, const pmgt_tensors* t, Bufs<T>& b, hipStream_t st) {
    RUNP("mirror", build_mirror<T>(t->params, b.mirror, e->desc_dev, (int)e->desc.size(), e->mirror_tiles, st));
    if (e->fp8) RUNP("mirror", quant_params_e4m3(t->params, e->desc8_dev, (int)e->desc8.size(), e->desc8_rows, b.mirror8, b.mscale, st));
    return 0;
}

// Table mode with per-node segment sums at hidden size 256 in bf16: the token phase of the embedding runs embed_tok8_*_kernel, which do not
// store the pre-LayerNorm sum (forward and backward take the same decision from the same engine state: encode_backward refuses changed options)
template <typename T>
static inline bool embed_recomputes(const pmgt_engine* e, bool table_mode) {
    return table_mode && !(e->opts & OPT_NO_SEGMENT_SUM) && e->d == 256 && sizeof(T) == 2;
}

// whether encoder_forward takes the fused projection + attention kernel (and, in training, stores Q|K|V|C head-major):
// a pure function of the configuration, the shape and the debug switches, so a separate backward call can re-derive it
template <typename T>
static inline bool fused_qa_applies(const pmgt_engine* e, int Tseq, int S, bool want_probs) {
    if (sizeof(T) != 2 || (e->opts & (OPT_NO_FUSED_QKVC_ATTENTION | OPT_TILE_GEMM | OPT_VALU_ATTENTION)) || want_probs) return false;
    return S == 32 && e->dh == 32 && (e->d == 256 || e->d == 128) && e->H % 2 == 0 && Tseq >= 2;
}

// beta == 1 (the author's own setting, scripts/run_pmgt.sh:24): the dot-product softmax is multiplied by exactly 0 (pmgt/pmgt/modeling_pmgt.py:519-521),
// so Q and K, their softmax and its backward are dead.  The fused forward / backward pair then projects, stores and differentiates V | C only
// (AttnArgs::vc_only); the query / key weight and bias gradients are exact zeros, as autograd reports them.  A pure function of configuration,
// shape and options: the forward and a separate backward call take the same decision.  (The generic kernels still READ Q and K, so the pair
// is all-or-nothing: no fused backward, no skip.)
template <typename T>
static inline bool vc_only_applies(const pmgt_engine* e, int Tseq, int S, bool want_probs) {
    if (e->cfg.beta != 1.f || e->fp8 || e->H % 4 != 0 || (e->opts & (OPT_NO_BETA_SKIP | OPT_NO_FUSED_ATTENTION_BWD))) return false;
    // (the backward's own conditions: its shape test -- the 32-bit offset bound on Tseq included -- here; its slab capacity is carve()'s, its operands
    //  are workspace buffers with ldx = d)
    return fused_qa_applies<T>(e, Tseq, S, want_probs) && attn_bwd_wgrad_shape_ok(Tseq, S, e->dh, e->H);
}

static inline bool use_table_projection(const pmgt_engine* e, const pmgt_tensors* t, int64_t n_tokens, bool by_ids) {
    // (the per-node buffers of the table mode live inside per-token buffers: [N+2, (NF+1) d] in E [M, max(NF, 2) d],
    //  dE [N+2, NF d] in a [M, d] temporary)
    return by_ids && !(e->opts & OPT_NO_TABLE_PROJECTION) && t->n_nodes > 0 && (t->n_nodes + 2) * std::max(2, e->NF) <= n_tokens;
}

// ---- encoder forward ---------------------------------------------------------------------------
template <typename T>
static int encoder_forward(const pmgt_engine* e, const pmgt_tensors* t, Bufs<T>& b, int Tseq, int S, const int64_t* ids,
                           const void* const* feats, const float* mask, bool train, T* hidden_states,
                           float* attn_probs, hipStream_t st, bool shortcut = false, int n_cls_only = 0) {
    const int d = e->d, I = e->I, L = e->L, H = e->H, NF = e->NF;
    const int M = Tseq * S;
    const float* P = t->params;
    PMGT_CHECK(S <= e->cfg.max_position_embeddings, -2, "sequence length %d exceeds max_position_embeddings %d", S,
               e->cfg.max_position_embeddings);
    const float pd = e->cfg.hidden_dropout_prob, pa = e->cfg.attention_probs_dropout_prob;
    // Feature projections.  Token mode: gather fused into the A-operand load, one projected row per token.
    // Table mode (small graphs: every node appears many times in a batch): project the WHOLE table once,
    // [N+2, F_m] -> [N+2, NF d], and let the mix kernel pick rows by node id -- M / (N+2) times fewer flops and
    // table bytes, same value per token (a row's projection does not depend on which rows share its tile).
    const int64_t n_rows = t->n_nodes + 2;
    const bool table_mode = use_table_projection(e, t, M, ids != nullptr);
    b.e_by_id = table_mode;
    PMGT_CHECK(!e->fp8 || ids != nullptr, -3, "fp8 mode gathers e4m3 feature rows by node id: pre-gathered feature tensors are not supported");
    for (int mod = 0; mod < NF && e->fp8; ++mod) {      // e4m3 table rows x e4m3 weights on the fp8 MFMA
        GemmF8 g; g.opts = e->opts;
        const int F = e->F[mod];
        g.A = t->tables[mod]; g.lda = F; g.a_rows = table_mode ? nullptr : ids;
        g.a_scale = t->table_scales[mod];
        g.B = b.mirror8 + e->m8Wf[mod]; g.ldb = F; g.b_row_scale = b.mscale + e->s8Wf[mod];
        g.C = b.E + mod * d; g.ldc = NF * d;
        g.M = table_mode ? (int)n_rows : M; g.N = d; g.K = F;
        g.bias = P + e->bf + mod * d;
        RUNP("fwd.gemm_featproj", gemm_nt_f8(g, st));
    }
    for (int mod = 0; mod < NF && !e->fp8; ++mod) {
        GemmNT g; g.opts = e->opts;
        const int F = e->F[mod];
        if (ids) { g.A = t->tables[mod]; g.a_rows = table_mode ? nullptr : ids; }
        else g.A = feats[mod];
        g.lda = F;
        g.B = wsel<T>(e, t, b, e->Wf[mod], e->mWf[mod]);
        g.ldb = F;
        g.C = b.E + mod * d; g.ldc = NF * d;
        g.M = table_mode ? (int)n_rows : M; g.N = d; g.K = F;
        g.bias = P + e->bf + mod * d;
        RUNP("fwd.gemm_featproj", gemm_nt<T>(g, st));
    }
    {
        EmbedMix m;
        m.S = S; m.d = d; m.nf = NF; m.Wa = P + e->Wa; m.ba = P + e->ba; m.pos = P + e->pos; m.role = P + e->role;
        m.gamma = P + e->ln_g; m.beta = P + e->ln_b; m.eps = e->cfg.layer_norm_eps;
        m.a = b.a; m.stats = b.emb_stats; m.h0 = b.h0;
        m.drop = dropcfg(t, train, pd, -1, SITE_EMB);
        // fp8 mode: whoever PRODUCES a layer input also writes it as per-row e4m3 (x8 / xscale), so the projection that
        // consumes it neither re-reads the bf16 row nor quantises it (x8_ok tracks whether the current `hin` has that copy)
        const bool pq = e->fp8 && !(e->opts & OPT_CONSUMER_QUANT);
        if (table_mode && !(e->opts & OPT_NO_SEGMENT_SUM)) {
            // the modality mix sum_k a_k e_k depends on the node only: once per node, then one [d] row per token
            T* F_all = b.E + n_rows * NF * d;                // fits: (N + 2) * (NF + 1) d <= M * max(NF, 2) d
            m.phase = 1; m.M = (int)n_rows; m.E = b.E; m.pre = F_all;
            RUNP("fwd.embed_mix", embed_mix_fwd<T>(m, st));
            m.phase = 2; m.M = M; m.E = F_all; m.e_rows = ids; m.pre = embed_recomputes<T>(e, true) ? nullptr : b.emb_pre;
            if (pq) { m.q8 = b.x8; m.q8_scale = b.xscale; }
            RUNP("fwd.embed_mix", embed_mix_fwd<T>(m, st));
        } else {
            m.M = M; m.E = b.E; m.e_rows = table_mode ? ids : nullptr; m.pre = b.emb_pre;
            if (pq) { m.q8 = b.x8; m.q8_scale = b.xscale; }
            RUNP("fwd.embed_mix", embed_mix_fwd<T>(m, st));
        }
    }
    bool x8_ok = e->fp8 && !(e->opts & OPT_CONSUMER_QUANT);
    if (hidden_states) PMGT_HIP(hipMemcpyAsync(hidden_states, b.h0, (size_t)M * d * sizeof(T), hipMemcpyDeviceToDevice, st));
    const T* hin = b.h0;
    for (int l = 0; l < L; ++l) {
        const LayerOff& o = e->layers[l];
        LayerBufs<T>& lb = b.layer[l];
        bool fused = false;
        if constexpr (sizeof(T) == 2) {   // headline shape: projection + attention in one kernel (Q|K|V|C never re-read from HBM)
            QkvcAttn f; f.opts = e->opts;
            f.X = hin; f.ldx = d; f.W = b.mirror + o.mWqkvc; f.ldw = d; f.bias = P + o.bqkvc;
            f.qkvc = lb.qkvc; f.ldq = 4 * d; f.ctx = lb.ctx; f.ldc = d; f.mask = mask;
            f.Tseq = Tseq; f.S = S; f.H = H; f.dh = e->dh; f.beta = e->cfg.beta;
            f.drop1 = dropcfg(t, train, pa, l, SITE_A1);
            f.drop2 = dropcfg(t, train, pa, l, SITE_A2);
            f.cls_only_seqs = (shortcut && l == L - 1) ? n_cls_only : 0;
            f.hm = train && !(e->opts & OPT_NO_HEAD_MAJOR);        // the backward that reads it understands the layout; inference keeps q | k | v | c
            f.vc_only = vc_only_applies<T>(e, Tseq, S, attn_probs != nullptr);
            if (e->fp8) { f.W8 = b.mirror8 + o.m8Wqkvc; f.wscale = b.mscale + o.s8Wqkvc; }
            if (e->fp8 && x8_ok) { f.X8 = b.x8; f.xscale = b.xscale; f.ldx = d; }
            if (fused_qa_applies<T>(e, Tseq, S, attn_probs != nullptr) && qkvc_attn_supported(f)) {
                RUNP("fwd.qkvc_attention", qkvc_attn_fwd(f, st));
                fused = true;
                b.qkvc_hm = f.hm;
                b.qkvc_vc = f.vc_only;
            } else {
                PMGT_CHECK(!f.vc_only, -2, "beta == 1 skip: the fused projection + attention kernel refused a shape vc_only_applies() accepted");
            }
        }
        if (!fused && e->fp8) {   // per-row e4m3 of the layer input, then the fp8 GEMM
            if (!x8_ok) RUNP("fwd.quant_x", quant_rows_e4m3<T>(hin, d, M, d, b.x8, d, b.xscale, st));
            GemmF8 g; g.opts = e->opts;
            g.A = b.x8; g.lda = d; g.a_row_scale = b.xscale; g.B = b.mirror8 + o.m8Wqkvc; g.ldb = d; g.b_row_scale = b.mscale + o.s8Wqkvc;
            g.C = lb.qkvc; g.ldc = 4 * d; g.M = M; g.N = 4 * d; g.K = d; g.bias = P + o.bqkvc;
            RUNP("fwd.gemm_qkvc", gemm_nt_f8(g, st));
        } else if (!fused) {   // Q,K,V,C projections as one [M,d] x [4d,d]^T GEMM
            GemmWS g; g.opts = e->opts;
            g.A = hin; g.lda = d; g.B = wsel<T>(e, t, b, o.Wqkvc, o.mWqkvc); g.ldb = d;
            g.C = lb.qkvc; g.ldc = 4 * d; g.M = M; g.N = 4 * d; g.K = d; g.bias = P + o.bqkvc;
            RUN(linear<T>(e, "fwd.gemm_qkvc", g, st));
        }
        if (!fused) {
            AttnArgs a; a.opts = e->opts;
            a.qkvc = lb.qkvc; a.mask = mask; a.ctx = lb.ctx;
            a.probs = attn_probs ? attn_probs + (int64_t)l * Tseq * H * S * S : nullptr;
            a.Tseq = Tseq; a.S = S; a.H = H; a.dh = e->dh; a.beta = e->cfg.beta;
            a.drop1 = dropcfg(t, train, pa, l, SITE_A1);
            a.drop2 = dropcfg(t, train, pa, l, SITE_A2);
            RUNP("fwd.attention", attn_fwd<T>(a, st));
        }
        // Tail of the layer (attn-out, FFN): on every token, or — last layer of the training fast path — only on
        // the rows the loss reads (padded and unread positions never influence them: SURVEY Q6).
        const bool sc = shortcut && l == L - 1;
        LayerBufs<T>& tb = sc ? b.ctail : lb;
        const int Mt = sc ? b.rcap : M;
        const int64_t* rows = sc ? b.need_rows : nullptr;
        const int* mdev = sc ? b.need_cnt : nullptr;
        {   // BertSelfOutput: LN(dropout(dense(ctx)) + hin)
            GemmWS g; g.opts = e->opts;
            g.A = lb.ctx; g.lda = d; g.a_rows = rows; g.B = wsel<T>(e, t, b, o.Wo, o.mWo); g.ldb = d;
            g.C = tb.ao_pre; g.ldc = d; g.M = Mt; g.N = d; g.K = d; g.bias = P + o.bo; g.m_dev = mdev;
            g.drop = dropcfg(t, train, pd, l, SITE_AO);
            g.res = hin; g.ldr = d; g.res_gather = sc;
            g.ln_out = tb.u; g.ln_stats = tb.stats1; g.ln_gamma = P + o.ln1g; g.ln_beta = P + o.ln1b; g.ln_eps = e->cfg.layer_norm_eps;
            g.skip_c = ln_from_y_applies<T>(e, Mt, d, sc);
            RUN(linear<T>(e, "fwd.gemm_attn_out", g, st));
        }
        {   // BertIntermediate: gelu(dense(u))
            GemmWS g; g.opts = e->opts;
            g.A = tb.u; g.lda = d; g.B = wsel<T>(e, t, b, o.W1, o.mW1); g.ldb = d;
            g.C = tb.g; g.ldc = I; g.M = Mt; g.N = I; g.K = d; g.bias = P + o.b1; g.m_dev = mdev;
            g.epi = EPI_GELU; g.aux = tb.ff_pre; g.ldaux = I;
            RUN(linear<T>(e, "fwd.gemm_ffn1", g, st));
        }
        {   // BertOutput: LN(dropout(dense(g)) + u)
            GemmWS g; g.opts = e->opts;
            g.A = tb.g; g.lda = I; g.B = wsel<T>(e, t, b, o.W2, o.mW2); g.ldb = I;
            g.C = tb.fo_pre; g.ldc = d; g.M = Mt; g.N = d; g.K = I; g.bias = P + o.b2; g.m_dev = mdev;
            g.drop = dropcfg(t, train, pd, l, SITE_FO);
            g.res = tb.u; g.ldr = d;
            g.ln_out = tb.hout; g.ln_stats = tb.stats2; g.ln_gamma = P + o.ln2g; g.ln_beta = P + o.ln2b; g.ln_eps = e->cfg.layer_norm_eps;
            x8_ok = false;
            if constexpr (sizeof(T) == 2) {
                if (e->fp8 && !(e->opts & OPT_CONSUMER_QUANT) && !sc && l + 1 < L) {
                    // the next layer's input, quantised where it is produced: the fused-LayerNorm epilogue of the streaming GEMM,
                    // or the LayerNorm launch that follows the tiled GEMM (d = 512 shapes)
                    g.q8 = b.x8; g.q8_scale = b.xscale;
                    x8_ok = true;
                }
            }
            g.skip_c = ln_from_y_applies<T>(e, Mt, I, sc);
            RUN(linear<T>(e, "fwd.gemm_ffn2", g, st));
        }
        if (hidden_states)
            PMGT_HIP(hipMemcpyAsync(hidden_states + (int64_t)(l + 1) * M * d, lb.hout, (size_t)M * d * sizeof(T),
                                    hipMemcpyDeviceToDevice, st));
        hin = lb.hout;
    }
    return 0;
}

// Partial-sum reductions (weight-gradient slabs, bias partials, LayerNorm partials) are tiny launches that sit between
// two big kernels of the dependent chain; with `overlap` on (opt-in: A/B on one box 82.5k / 81.4k vs 81.2k / 81.8k nodes/s,
// i.e. neutral -- the launch queue already hides them) they run on the engine's side stream: the
// producer kernel records an event on `main`, the side stream waits for it and reduces, and the partial buffers are
// double-buffered so the next producer never waits (the one after next waits on an event that is long signalled).
struct SideReduce {
    pmgt_engine* e;
    hipStream_t main;
    bool on;
    SideReduce(const pmgt_engine* eng, hipStream_t m) : e(const_cast<pmgt_engine*>(eng)), main(m), on(eng->overlap() && eng->side != nullptr) {}
    // before a producer overwrites a partial buffer: the reduction that last read it is done
    int acquire(hipEvent_t& done) {
        if (done) PMGT_HIP(hipStreamWaitEvent(main, done, 0));
        done = nullptr;
        return 0;
    }
    // after the producer launch: returns the stream the reductions go to
    int begin(hipStream_t* st) {
        *st = main;
        if (!on) return 0;
        hipEvent_t ev = e->next_sync();
        PMGT_HIP(hipEventRecord(ev, main));
        PMGT_HIP(hipStreamWaitEvent(e->side, ev, 0));
        *st = e->side;
        return 0;
    }
    int end(hipEvent_t& done) {
        done = nullptr;
        if (!on) return 0;
        done = e->next_sync();
        PMGT_HIP(hipEventRecord(done, e->side));
        return 0;
    }
};

template <typename T>
static int join_side_wgrads(const pmgt_engine* e, Bufs<T>& b, hipStream_t main) {
    if (!b.side_pending) return 0;
    pmgt_engine* em = const_cast<pmgt_engine*>(e);
    hipEvent_t ev = em->next_sync();
    PMGT_HIP(hipEventRecord(ev, em->side));
    PMGT_HIP(hipStreamWaitEvent(main, ev, 0));
    b.side_pending = false;
    return 0;
}
static inline int wait_event(hipStream_t st, hipEvent_t ev) {
    if (ev) PMGT_HIP(hipStreamWaitEvent(st, ev, 0));
    return 0;
}
template <typename T>
static int flush_reduces(const pmgt_engine* e, Bufs<T>& b, hipStream_t st) {
    RUN(join_side_wgrads<T>(e, b, st));      // the slabs of side-stream weight gradients are complete before anything sums them
    if (!b.pend.empty()) RUNP("bwd.slab_reduce", multi_reduce(b.pend.data(), (int)b.pend.size(), st));
    b.pend.clear();
    b.arena_cur = 0;          // later producers are stream-ordered behind the launch that read the arena
    return 0;
}
// One or two regions of the arena for ONE producer launch (weight-gradient slab + its bias slab).  The flush that makes room comes
// BEFORE the first region is handed out, never between the two: a flush rewinds the arena, and the first region -- handed out but
// not queued yet -- would be overwritten by later producers before multi_reduce reads it.
template <typename T>
static int take_partials2(const pmgt_engine* e, Bufs<T>& b, int64_t n1, float** out1, int64_t n2, float** out2, hipStream_t st) {
    n1 = align_up(n1, 64);
    n2 = out2 ? align_up(n2, 64) : 0;
    PMGT_CHECK(n1 + n2 <= b.arena_elems, -4, "partial-sum arena too small: %lld > %lld floats", (long long)(n1 + n2), (long long)b.arena_elems);
    // (small_arena, a test option: every producer starts from an empty arena, i.e. the jobs queued so far are reduced first)
    if (b.arena_cur + n1 + n2 > b.arena_elems || ((e->opts & OPT_SMALL_ARENA) && b.arena_cur > 0)) RUN(flush_reduces<T>(e, b, st));
    *out1 = b.arena + b.arena_cur;
    b.arena_cur += n1;
    if (out2) { *out2 = b.arena + b.arena_cur; b.arena_cur += n2; }
    return 0;
}
template <typename T>
static int take_partials(const pmgt_engine* e, Bufs<T>& b, int64_t n, float** out, hipStream_t st) {
    return take_partials2<T>(e, b, n, out, 0, nullptr, st);
}
template <typename T>
static int queue_reduce(const pmgt_engine* e, Bufs<T>& b, const float* src, int rows, int64_t n, float* dst, bool acc, hipStream_t st) {
    for (const ReduceJob& j : b.pend)
        if (j.dst == dst) {       // two sums into one destination must not share a launch: the second one reads the first one's result
            // (the arena keeps its contents: only the job list is run)
            RUN(join_side_wgrads<T>(e, b, st));
            RUNP("bwd.slab_reduce", multi_reduce(b.pend.data(), (int)b.pend.size(), st));
            b.pend.clear();
            break;
        }
    b.pend.push_back(ReduceJob{src, dst, n, rows, acc});
    return 0;
}

// wgrad helper: dst[N1,N2] (+)= P^T Q through the split slabs
template <typename T>
static int wgrad(const char* name, const pmgt_engine* e, Bufs<T>& b, const T* Pm, int64_t ldp, const T* Qm, int64_t ldq, const int64_t* q_rows,
                 int M, int m_for_splits, int N1, int N2, float* dst, bool acc, const int* m_dev, hipStream_t main,
                 float* bias_dst = nullptr, int perm_d = 0, int perm_dh = 0, float q_f8_scale = 0.f, bool side_ok = false) {
    b.side_last = nullptr;
    if (b.defer) {
        hipStream_t st = main;
        // side stream: forked behind everything `main` has launched so far (the operands are final there); phase timers bracket launches on
        // `main`, so a profiled step keeps the weight gradients in line
        const bool side = side_ok && b.side_wgrad && !e->prof.on;
        pmgt_engine* em = const_cast<pmgt_engine*>(e);
        GemmTN g; g.opts = e->opts;
        g.P = Pm; g.ldp = ldp; g.Q = Qm; g.ldq = ldq; g.q_rows = q_rows; g.M = M; g.N1 = N1; g.N2 = N2;
        g.m_dev = m_dev; g.zeros = e->zeros; g.perm_d = perm_d; g.perm_dh = perm_dh;
        g.q_f8 = q_f8_scale > 0.f; g.q_scale = q_f8_scale;
        g.splits = gemm_tn_pick_splits(m_for_splits, N1, N2, gemm_tn_bkm<T>(), e->opts);
        g.bias_slab = nullptr;
        RUN(take_partials2<T>(e, b, (int64_t)g.splits * N1 * N2, &g.slab, (int64_t)g.splits * N1, bias_dst ? &g.bias_slab : nullptr, main));
        hipStream_t ws = main;
        if (side) {
            hipEvent_t ev = em->next_sync();
            PMGT_HIP(hipEventRecord(ev, main));
            PMGT_HIP(hipStreamWaitEvent(em->side, ev, 0));
            ws = em->side;
        }
        RUNP(name, gemm_tn<T>(g, ws));
        if (side) {
            b.side_last = em->next_sync();
            PMGT_HIP(hipEventRecord(b.side_last, em->side));
            b.side_pending = true;
        }
        RUN(queue_reduce<T>(e, b, g.slab, g.splits, (int64_t)N1 * N2, dst, acc, main));
        if (bias_dst) RUN(queue_reduce<T>(e, b, g.bias_slab, g.splits, N1, bias_dst, acc, main));
        return 0;
    }
    SideReduce sr(e, main);
    const int slot = b.wg_idx++ & 1;
    float* slab = b.slab + (int64_t)slot * b.slab_elems;
    float* bpart = b.part_side + (int64_t)slot * b.part_side_elems;
    RUN(sr.acquire(b.wg_done[slot]));
    hipStream_t st = main;
    GemmTN g; g.opts = e->opts;
    g.P = Pm; g.ldp = ldp; g.Q = Qm; g.ldq = ldq; g.q_rows = q_rows; g.M = M; g.N1 = N1; g.N2 = N2;
    g.slab = slab; g.m_dev = m_dev; g.zeros = e->zeros; g.perm_d = perm_d; g.perm_dh = perm_dh;
    g.q_f8 = q_f8_scale > 0.f; g.q_scale = q_f8_scale;      // fp8 mode: Q = e4m3 feature table
    g.splits = gemm_tn_pick_splits(m_for_splits, N1, N2, gemm_tn_bkm<T>(), e->opts);
    g.bias_slab = bias_dst ? bpart : nullptr;          // [splits <= 512][N1]
    RUNP(name, gemm_tn<T>(g, st));
    RUN(sr.begin(&st));
    RUNP("bwd.slab_reduce", slab_reduce(slab, g.splits, (int64_t)N1 * N2, dst, acc, st));
    if (bias_dst) RUNP("bwd.slab_reduce", slab_reduce(bpart, g.splits, N1, bias_dst, acc, st));
    RUN(sr.end(b.wg_done[slot]));
    return 0;
}
// LayerNorm backward + the reduction of its dgamma | dbeta | dbias partials
template <typename T>
static int ln_bwd_reduce(const pmgt_engine* e, Bufs<T>& b, const T* dy, const T* x, const float* stats, const float* gamma, T* dx, T* dx_drop,
                         int Mt, int d, DropCfg out_drop, const int* mdev, float* dst, bool acc, hipStream_t main, const float* beta_y = nullptr) {
    if (b.defer) {
        hipStream_t st = main;
        float* part = nullptr;
        RUN(take_partials<T>(e, b, (int64_t)ln_bwd_parts(Mt) * 3 * d, &part, main));
        RUNP("bwd.layernorm", ln_bwd<T>(dy, x, stats, gamma, dx, dx_drop, part, Mt, d, DropCfg{nullptr, 0.f, 0}, out_drop, main, mdev, beta_y));
        RUN(queue_reduce<T>(e, b, part, ln_bwd_parts(Mt), 3 * d, dst, acc, main));
        return 0;
    }
    SideReduce sr(e, main);
    const int slot = b.ln_idx++ & 1;
    float* part = b.ln_part + (int64_t)slot * b.ln_part_elems;
    RUN(sr.acquire(b.ln_done[slot]));
    hipStream_t st = main;
    RUNP("bwd.layernorm", ln_bwd<T>(dy, x, stats, gamma, dx, dx_drop, part, Mt, d, DropCfg{nullptr, 0.f, 0}, out_drop, st, mdev, beta_y));
    RUN(sr.begin(&st));
    RUNP("bwd.slab_reduce", slab_reduce(part, ln_bwd_parts(Mt), 3 * d, dst, acc, st));
    RUN(sr.end(b.ln_done[slot]));
    return 0;
}
// everything queued on the side stream so far is ordered before what `main` launches next
template <typename T>
static inline int join_side_all(const pmgt_engine* e, Bufs<T>& b, hipStream_t main) {
    pmgt_engine* em = const_cast<pmgt_engine*>(e);
    for (int i = 0; i < 2; ++i) { b.wg_done[i] = nullptr; b.ln_done[i] = nullptr; }
    if (!em->side) return 0;
    hipEvent_t ev = em->next_sync();
    PMGT_HIP(hipEventRecord(ev, em->side));
    PMGT_HIP(hipStreamWaitEvent(main, ev, 0));
    return 0;
}

// ---- encoder backward: dcur (in b.bA) = d loss / d h_L; leaves parameter grads in t->grads ----------
template <typename T>
static int encoder_backward(const pmgt_engine* e, const pmgt_tensors* t, Bufs<T>& b, int Tseq, int S, bool acc, hipStream_t st,
                            bool shortcut = false, bool train = true, const void* const* feats = nullptr,
                            int n_cls_only = 0, bool whole_buffer = false) {
    const int d = e->d, I = e->I, L = e->L, H = e->H, NF = e->NF;
    const int M = Tseq * S;
    // dense weight gradients next to the data-gradient chain (side stream): opt-in.  Measured at the bench model (profiles/r04/NOTES.md section 9):
    // B = 32 1.25 -> 1.32 ms, B = 256 3.12 -> 3.14, B = 1 024 9.24 -> 9.45 eager, worse still under graph replay -- a fork / join through events
    // costs more than a 10 - 25 us weight gradient overlaps.
    b.side_wgrad = b.defer && e->side != nullptr && (e->opts & OPT_SIDE_STREAM_WGRAD);
    b.side_pending = false;
    // events one backward pass can take from the ring: per layer <= 3 side-stream weight gradients x (fork + done) + a join per flush
    // (<= 4) + the fused attention backward's reduce pair; plus the token sort's pair, the embedding pass and the final join
    const_cast<pmgt_engine*>(e)->reserve_sync((size_t)2 * (12 * (size_t)L + 32));
    const float* P = t->params;
    float* G = t->grads;
    const float pd = e->cfg.hidden_dropout_prob, pa = e->cfg.attention_probs_dropout_prob;
    const bool dd = train && pd > 0.f;
    bool ln2_done = false;      // this layer's LN2 backward ran inside the previous iteration's last GEMM
    for (int l = L - 1; l >= 0; --l) {
        const LayerOff& o = e->layers[l];
        LayerBufs<T>& lb = b.layer[l];
        const T* hin = l == 0 ? b.h0 : b.layer[l - 1].hout;
        // tail of the layer on all tokens, or (last layer, fast path) on the compacted rows the loss read
        const bool sc = shortcut && l == L - 1;
        LayerBufs<T>& tb = sc ? b.ctail : lb;
        const int Mt = sc ? b.rcap : M;
        const int msp = sc ? std::max(256, Mt / 3) : M;               // row count used to size the wgrad splits
        const int64_t* rows = sc ? b.need_rows : nullptr;
        const int* mdev = sc ? b.need_cnt : nullptr;
        T* gA = sc ? b.c_dh : b.bA;
        T* gB = sc ? b.c_bB : b.bB;
        T* gC = sc ? b.c_bC : b.bC;
        T* gD = sc ? b.c_bD : b.bD;
        T* gbig = sc ? b.c_big : b.big;
        const bool ln1_from_y = ln_from_y_applies<T>(e, Mt, d, sc), ln2_from_y = ln_from_y_applies<T>(e, Mt, I, sc);      // as the forward decided
        // LN2 backward: gA -> gB (residual branch), gC (masked: gradient of the FFN2 dense output) -- unless it already ran as the
        // epilogue of the layer above's dX = dQKVC W (below): gB / gC are then in place and gA was never written
        if (!ln2_done)
            RUN(ln_bwd_reduce<T>(e, b, gA, ln2_from_y ? tb.hout : tb.fo_pre, tb.stats2, P + o.ln2g, gB, dd ? gC : nullptr, Mt, d,
                                 dropcfg(t, train, pd, l, SITE_FO), mdev, G + o.ln2g, acc, st, ln2_from_y ? P + o.ln2b : nullptr));                                          // dgamma | dbeta | db2
        ln2_done = false;
        const T* dY2 = dd ? gC : gB;
        RUN(wgrad<T>("bwd.wgrad_ffn2", e, b, dY2, d, tb.g, I, nullptr, Mt, msp, d, I, G + o.W2, acc, mdev, st, nullptr, 0, 0, 0.f, true));
        const hipEvent_t w2_done = b.side_last;      // (side stream) waited for before dY2 is overwritten
        {   // d ff_pre = (dY2 W2) * gelu'(ff_pre)
            GemmWS g; g.opts = e->opts;
            g.A = dY2; g.lda = d; g.B = b.mirror + o.mW2T; g.ldb = d; g.C = gbig; g.ldc = I; g.m_dev = mdev;
            g.M = Mt; g.N = I; g.K = d; g.epi = EPI_GELU_GRAD; g.aux = tb.ff_pre; g.ldaux = I;
            RUN(linear<T>(e, "bwd.dgrad_ffn2", g, st));
        }
        RUN(wgrad<T>("bwd.wgrad_ffn1", e, b, gbig, I, tb.u, d, nullptr, Mt, msp, I, d, G + o.W1, acc, mdev, st, G + o.b1, 0, 0, 0.f, true));
        const hipEvent_t w1_done = b.side_last;      // ... before d ff_pre (gbig / b.big) is overwritten
        RUN(wait_event(st, w2_done));                // the launches below write the residual-branch / masked-gradient buffers dY2 lives in
        bool ln1_fused = false;
        if constexpr (sizeof(T) == 2) {
            // du = dff W1 + residual branch, and the LN1 backward of du in the same launch (the role-split streaming kernel's epilogue
            // role: du never reaches HBM; dx overwrites the residual branch in place, row tile by row tile)
            GemmWS g; g.opts = e->opts;
            g.A = gbig; g.lda = I; g.B = b.mirror + o.mW1T; g.ldb = I; g.C = gB; g.ldc = d; g.m_dev = mdev;
            g.M = Mt; g.N = d; g.K = I; g.res = gB; g.ldr = d;
            g.lnb_y = tb.u; g.lnb_ldy = d; g.lnb_stats = tb.stats1; g.lnb_gamma = P + o.ln1g; g.lnb_beta = P + o.ln1b;
            g.lnb_dx_drop = dd ? gC : nullptr; g.lnb_lddx = d; g.lnb_drop = dropcfg(t, train, pd, l, SITE_AO);
            if (ln1_from_y && b.defer && gemm_wsr_lnb_ok(g)) {
                const int parts = gemm_wsr_lnb_parts(Mt);
                RUN(take_partials<T>(e, b, (int64_t)parts * 3 * d, &g.lnb_part, st));
                RUNP("bwd.dgrad_ffn1_lnb", gemm_wsr_lnb(g, st));
                RUN(queue_reduce<T>(e, b, g.lnb_part, parts, 3 * d, G + o.ln1g, acc, st));                                                                                 // dgamma | dbeta | dbo
                ln1_fused = true;
            } else if (ln1_from_y && b.defer && I > 512) {
                // intermediate sizes above 512: K = I is outside the streaming family; the 256 x 256 tile owns whole rows of du as well
                // (the form that runs the LN2 backward behind dX = dQKVC W below)
                GemmNT f = g;
                if (gemm_nt_lnb_ok(f)) {
                    const int parts = gemm_nt_lnb_parts(Mt);
                    RUN(take_partials<T>(e, b, (int64_t)parts * 3 * d, &f.lnb_part, st));
                    RUNP("bwd.dgrad_ffn1_lnb", gemm_nt_lnb(f, st));
                    RUN(queue_reduce<T>(e, b, f.lnb_part, parts, 3 * d, G + o.ln1g, acc, st));                                                                             // dgamma | dbeta | dbo
                    ln1_fused = true;
                }
            }
        }
        if (!ln1_fused) {
            {   // du = dff W1 + residual branch
                GemmWS g; g.opts = e->opts;
                g.A = gbig; g.lda = I; g.B = b.mirror + o.mW1T; g.ldb = I; g.C = gD; g.ldc = d; g.m_dev = mdev;
                g.M = Mt; g.N = d; g.K = I; g.res = gB; g.ldr = d;
                RUN(linear<T>(e, "bwd.dgrad_ffn1", g, st));
            }
            // LN1 backward
            RUN(ln_bwd_reduce<T>(e, b, gD, ln1_from_y ? tb.u : tb.ao_pre, tb.stats1, P + o.ln1g, gB, dd ? gC : nullptr, Mt, d,
                                 dropcfg(t, train, pd, l, SITE_AO), mdev, G + o.ln1g, acc, st, ln1_from_y ? P + o.ln1b : nullptr));                                          // dgamma | dbeta | dbo
        }
        const T* dYo = dd ? gC : gB;
        RUN(wgrad<T>("bwd.wgrad_attn_out", e, b, dYo, d, lb.ctx, d, rows, Mt, msp, d, d, G + o.Wo, acc, mdev, st, nullptr, 0, 0, 0.f, true));
        // (its operands -- dYo, ctx -- are next overwritten behind the flush below, which joins the side stream)
        {   // dctx = dYo Wo
            GemmWS g; g.opts = e->opts;
            g.A = dYo; g.lda = d; g.B = b.mirror + o.mWoT; g.ldb = d; g.C = gD; g.ldc = d; g.M = Mt; g.N = d; g.K = d; g.m_dev = mdev;
            RUN(linear<T>(e, "bwd.dgrad_attn_out", g, st));
        }
        if (sc) {   // back to the full token layout: dctx is zero everywhere except the compacted rows; the residual branch
                    // (non-zero on those rows only) is ADDED to them after the dense dgrad below instead of travelling as
                    // a dense, mostly-zero residual operand
            PMGT_HIP(hipMemsetAsync(b.bD, 0, (size_t)M * d * sizeof(T), st));
            RUN(scatter_rows<T>(gD, b.need_rows, b.need_cnt, Mt, d, b.bD, st));
        }
        RUN(wait_event(st, w1_done));                // the attention backward writes dQ|dK|dV|dC into the buffer d ff_pre lived in
        bool fused_bw = false;
        // (beta == 1 skip) whether dX = dQKVC W can walk the V | C blocks only: decided here, in front of the launch that writes dQKVC
        bool vc_dgrad_kmap = false;
        if constexpr (sizeof(T) == 2) {
            if (b.qkvc_vc && b.qkvc_hm) {
                GemmNT gp; gp.opts = e->opts;
                gp.A = b.big; gp.lda = 4 * d; gp.B = b.mirror + o.mWqkvcT_hm; gp.ldb = 4 * d; gp.C = b.bA; gp.ldc = d; gp.M = M; gp.N = d; gp.K = 2 * d;
                gp.res = sc ? nullptr : b.bB; gp.ldr = d; gp.kmap_vc = 1;
                vc_dgrad_kmap = gemm_nt_big_applies(gp);
            }
        }
        {
            AttnArgs a; a.opts = e->opts;
            a.qkvc = lb.qkvc; a.mask = b.mask; a.Tseq = Tseq; a.S = S; a.H = H; a.dh = e->dh; a.beta = e->cfg.beta;
            a.drop1 = dropcfg(t, train, pa, l, SITE_A1);
            a.drop2 = dropcfg(t, train, pa, l, SITE_A2);
            a.dctx = b.bD; a.dqkvc = b.big;
            a.cls_only_seqs = sc ? n_cls_only : 0;       // their dctx is non-zero at row 0 only (scatter_rows above)
            a.hm = b.qkvc_hm;
            a.vc_only = b.qkvc_vc;
            if constexpr (sizeof(T) == 2) {
                // headline shape: attention backward and the Q|K|V|C weight gradient in ONE launch (attention waves + GEMM waves per
                // CU): dQ|dK|dV|dC are not re-read for the weight gradient, and its partial sums shrink from one slab per M-split to
                // one [128, d] block per workgroup
                AttnBwdWg w;
                w.a = a; w.x = hin; w.ldx = d;
                // last layer of the fast path: dctx is zero outside the rows the loss read (memset + scatter_rows above), so the fused
                // kernel computes the same sums as the CLS-only pair of kernels (one launch against 380 + 250 us); the attention waves of
                // query rows 16 .. 31 skip their softmax phases for the sequences that are read at row 0 only (their dO rows are zero:
                // exact zeros either way, option no_cls_only_attention_bwd runs them)
                if (e->opts & OPT_NO_CLS_ONLY_ATTENTION_BWD) w.a.cls_only_seqs = 0;
                const int parts = attn_bwd_wgrad_parts(H);
                SideReduce sr(e, st);
                const int slot = b.wg_idx & 1;
                w.slab = b.slab + (int64_t)slot * b.slab_elems;
                w.bias_slab = b.part_side + (int64_t)slot * b.part_side_elems;
                const bool abw_ok = !(e->opts & (OPT_NO_FUSED_ATTENTION_BWD | OPT_TILE_GEMM | OPT_VALU_ATTENTION)) && (int64_t)parts * 4 * d * d <= b.slab_elems &&
                                    (int64_t)parts * 4 * d <= b.part_side_elems && attn_bwd_wgrad_supported(w);
                PMGT_CHECK(abw_ok || !b.qkvc_vc, -2, "beta == 1 skip: the forward left Q / K unwritten but the fused attention backward does not apply here");
                // dX = dV W_v + dC W_c below skips the dQ / dK blocks of a head-major gradient by k-step on the 256 x 256 tile; where that tile
                // does not run (small M) the full product runs over dQ = dK = 0
                if (b.qkvc_vc && b.qkvc_hm && !vc_dgrad_kmap) PMGT_HIP(hipMemsetAsync(b.big, 0, (size_t)M * 4 * d * sizeof(T), st));
                if (abw_ok && b.defer && attn_bwd_wgrad_vc2_supported(w)) {
                    // beta == 1, two heads per step: partial sums of the value | ctx_attention rows only ([parts2][2 d, d]); query / key are exact zeros
                    const int parts2 = attn_bwd_wgrad_vc2_parts(H);
                    PMGT_CHECK((int64_t)parts2 * 2 * d * d <= b.arena_elems, -4, "partial-sum arena too small for the two-heads-per-step attention backward");
                    RUN(take_partials2<T>(e, b, (int64_t)parts2 * 2 * d * d, &w.slab, (int64_t)parts2 * 2 * d, &w.bias_slab, st));
                    RUNP("bwd.attention_wgrad", attn_bwd_wgrad_vc2(w, st));
                    RUN(queue_reduce<T>(e, b, w.slab, parts2, (int64_t)2 * d * d, G + o.Wqkvc + (int64_t)2 * d * d, acc, st));
                    RUN(queue_reduce<T>(e, b, w.bias_slab, parts2, 2 * d, G + o.bqkvc + 2 * d, acc, st));
                    if (!acc) {
                        PMGT_HIP(hipMemsetAsync(G + o.Wqkvc, 0, (size_t)2 * d * d * sizeof(float), st));
                        PMGT_HIP(hipMemsetAsync(G + o.bqkvc, 0, (size_t)2 * d * sizeof(float), st));
                    }
                    fused_bw = true;
                } else if (abw_ok && b.defer) {
                    RUN(take_partials2<T>(e, b, (int64_t)parts * 4 * d * d, &w.slab, (int64_t)parts * 4 * d, &w.bias_slab, st));
                    RUNP("bwd.attention_wgrad", attn_bwd_wgrad(w, st));
                    RUN(queue_reduce<T>(e, b, w.slab, parts, (int64_t)4 * d * d, G + o.Wqkvc, acc, st));
                    RUN(queue_reduce<T>(e, b, w.bias_slab, parts, 4 * d, G + o.bqkvc, acc, st));
                    fused_bw = true;
                } else if (abw_ok) {
                    ++b.wg_idx;
                    RUN(sr.acquire(b.wg_done[slot]));
                    RUNP("bwd.attention_wgrad", attn_bwd_wgrad(w, st));
                    hipStream_t rs = st;
                    RUN(sr.begin(&rs));
                    RUNP("bwd.slab_reduce", slab_reduce(w.slab, parts, (int64_t)4 * d * d, G + o.Wqkvc, acc, rs));
                    RUNP("bwd.slab_reduce", slab_reduce(w.bias_slab, parts, 4 * d, G + o.bqkvc, acc, rs));
                    RUN(sr.end(b.wg_done[slot]));
                    fused_bw = true;
                }
            }
            if (!fused_bw) RUNP("bwd.attention", attn_bwd<T>(a, st));
        }
        if (!fused_bw)
            RUN(wgrad<T>("bwd.wgrad_qkvc", e, b, b.big, 4 * d, hin, d, nullptr, M, M, 4 * d, d, G + o.Wqkvc, acc, nullptr, st, G + o.bqkvc,
                         b.qkvc_hm ? d : 0, b.qkvc_hm ? e->dh : 0));
        RUN(flush_reduces<T>(e, b, st));
        // every gradient of layer l is final in stream order: let the data-parallel exchange of this bucket start now
        if (e->grad_cb_fine()) e->grad_ready(o.Wqkvc, (l + 1 < L ? e->layers[l + 1].Wqkvc : e->Wn) - o.Wqkvc);
        {   // d hin = dqkvc Wqkvc + residual branch
            GemmNT g; g.opts = e->opts;
            g.A = b.big; g.lda = 4 * d; g.B = b.mirror + (b.qkvc_hm ? o.mWqkvcT_hm : o.mWqkvcT); g.ldb = 4 * d; g.C = b.bA; g.ldc = d;
            g.M = M; g.N = d; g.K = 4 * d; g.res = sc ? nullptr : b.bB; g.ldr = d;
            if (b.qkvc_vc && vc_dgrad_kmap) { g.kmap_vc = 1; g.K = 2 * d; }                       // head-major: the V | C block of every head
            else if (b.qkvc_vc && !b.qkvc_hm) { g.A = b.big + 2 * d; g.B = b.mirror + o.mWqkvcT + 2 * d; g.K = 2 * d; }      // q | k | v | c: the upper half
            bool with_ln = false;
            if constexpr (sizeof(T) == 2) {
                // d hin is the gradient of layer l - 1's output LayerNorm (BertOutput): its backward runs on the tile's rows behind the
                // main loop of this GEMM (N = 256 = whole rows per workgroup), d hin never reaches HBM; dx overwrites the residual branch
                // in place (b.bB), the masked copy goes to b.bC -- exactly what the next iteration's LN2 step would have left there
                // (shortcut layer: its residual branch exists on the compacted rows only -- read through the inverse row map instead of
                //  being scattered into d hin by a launch of its own behind the GEMM)
                if (l >= 1 && b.defer && ln_from_y_applies<T>(e, M, I, false) && (!sc || b.need_inv != nullptr)) {
                    const LayerOff& on = e->layers[l - 1];
                    GemmNT f = g;
                    f.C = b.bB;
                    if (sc) { f.res = gB; f.ldr = d; f.lnb_res_inv = b.need_inv; }
                    f.lnb_y = b.layer[l - 1].hout; f.lnb_ldy = d; f.lnb_stats = b.layer[l - 1].stats2; f.lnb_gamma = P + on.ln2g; f.lnb_beta = P + on.ln2b;
                    f.lnb_dx_drop = dd ? b.bC : nullptr; f.lnb_lddx = d; f.lnb_drop = dropcfg(t, train, pd, l - 1, SITE_FO);
                    if (gemm_nt_lnb_ok(f)) {
                        const int parts = gemm_nt_lnb_parts(M);
                        RUN(take_partials<T>(e, b, (int64_t)parts * 3 * d, &f.lnb_part, st));
                        RUNP("bwd.dgrad_qkvc_lnb", gemm_nt_lnb(f, st));
                        RUN(queue_reduce<T>(e, b, f.lnb_part, parts, 3 * d, G + on.ln2g, acc, st));                                                                        // dgamma | dbeta | db2 of layer l - 1
                        with_ln = ln2_done = true;
                    }
                }
            }
            if (!with_ln) {
                RUNP("bwd.dgrad_qkvc", gemm_nt<T>(g, st));
                if (sc) RUN(scatter_rows<T>(gB, b.need_rows, b.need_cnt, Mt, d, b.bA, st, true));
            }
        }
    }
    // embeddings
    {
        EmbedMix m;
        m.S = S; m.d = d; m.nf = NF; m.Wa = P + e->Wa; m.gamma = P + e->ln_g; m.a = b.a;
        m.stats = b.emb_stats; m.drop = dropcfg(t, train, pd, -1, SITE_EMB);
        m.part = b.part;
        const int pe = embed_part_elems(d, NF);
        const bool by_node = b.e_by_id && !(e->opts & OPT_NO_SEGMENT_SUM);
        PMGT_CHECK(!e->fp8 || feats == nullptr, -3, "fp8 mode: pre-gathered feature tensors are not supported");
        if (by_node) {
            // Table mode: per token only the LayerNorm backward (dF); the segment sums of dF per node id feed the
            // per-node backward of the modality mix, whose dE [N+2, NF d] is the P operand of the weight-gradient GEMMs.
            const int n_rows = (int)t->n_nodes + 2;
            m.phase = 2; m.M = M; m.dh0 = b.bA; m.pre = b.emb_pre; m.dF = b.bB;
            if (embed_recomputes<T>(e, true)) {      // the forward kept F_all [n_rows, d] behind the projected table instead of the per-token sum
                m.pre = nullptr; m.E = b.E + (int64_t)n_rows * NF * d; m.e_rows = b.ids; m.pos = P + e->pos; m.role = P + e->role;
            }
            if (b.defer) RUN(take_partials<T>(e, b, (int64_t)embed_bwd_parts(M) * pe, &m.part, st));
            RUNP("bwd.embed_mix", embed_mix_bwd<T>(m, st));
            if (b.defer) RUN(queue_reduce<T>(e, b, m.part, embed_bwd_parts(M), pe, G + e->ln_g, acc, st));
            else RUNP("bwd.slab_reduce", slab_reduce(b.part, embed_bwd_parts(M), pe, G + e->ln_g, acc, st));
            RUNP("bwd.colsum", colsum<T>(b.bB, (int64_t)S * d, Tseq, S * d, b.part, b.possum, false, nullptr, st));
            RUN(pos_role_finish(b.possum, S, d, e->cfg.max_position_embeddings, G + e->pos, G + e->role, acc, st));
            if (b.sort_done) PMGT_HIP(hipStreamWaitEvent(st, b.sort_done, 0));
            else RUNP("bwd.segsum_featproj", seg_sort(b.ids, M, n_rows, b.sg_keys, b.sg_vals, b.sg_skeys, b.sg_perm, b.sg_off, b.sg_tmp, b.sg_tmp_bytes, st));
            RUNP("bwd.segsum_featproj", (seg_sum<T, float>(b.bB, d, b.sg_skeys, b.sg_perm, b.sg_off, M, n_rows, d, (float*)b.bC, b.sg_part, st)));     // (N+2) d fp32 <= M d bf16
            m.phase = 1; m.M = n_rows; m.E = b.E; m.e_rows = nullptr; m.dF = b.bC; m.dF_f32 = true; m.dE = b.bD;      // (N+2) NF d <= M d
            m.part = b.part;
            if (b.defer) RUN(take_partials<T>(e, b, (int64_t)embed_bwd_parts(n_rows) * pe, &m.part, st));
            RUNP("bwd.embed_mix", embed_mix_bwd<T>(m, st));
            if (b.defer) RUN(queue_reduce<T>(e, b, m.part, embed_bwd_parts(n_rows), pe, G + e->ln_g, true, st));
            else RUNP("bwd.slab_reduce", slab_reduce(b.part, embed_bwd_parts(n_rows), pe, G + e->ln_g, true, st));
            for (int mod = 0; mod < NF; ++mod)
                RUN(wgrad<T>("bwd.wgrad_featproj", e, b, b.bD + mod * d, NF * d, (const T*)t->tables[mod], e->F[mod], nullptr, n_rows, n_rows, d, e->F[mod],
                             G + e->Wf[mod], acc, nullptr, st, G + e->bf + mod * d, 0, 0, e->fp8 ? t->table_scales[mod] : 0.f));      // scale > 0: the table is e4m3
        } else {
            m.M = M; m.E = b.E; m.e_rows = b.e_by_id ? b.ids : nullptr; m.pre = b.emb_pre;
            m.dh0 = b.bA; m.dE = b.big; m.dF = b.bB;
            if (b.defer) RUN(take_partials<T>(e, b, (int64_t)embed_bwd_parts(M) * pe, &m.part, st));
            RUNP("bwd.embed_mix", embed_mix_bwd<T>(m, st));
            if (b.defer) RUN(queue_reduce<T>(e, b, m.part, embed_bwd_parts(M), pe, G + e->ln_g, acc, st));
            else RUNP("bwd.slab_reduce", slab_reduce(b.part, embed_bwd_parts(M), pe, G + e->ln_g, acc, st));
            RUNP("bwd.colsum", colsum<T>(b.bB, (int64_t)S * d, Tseq, S * d, b.part, b.possum, false, nullptr, st));
            RUN(pos_role_finish(b.possum, S, d, e->cfg.max_position_embeddings, G + e->pos, G + e->role, acc, st));
            for (int mod = 0; mod < NF; ++mod)
                RUN(wgrad<T>("bwd.wgrad_featproj", e, b, b.big + mod * d, NF * d, feats ? (const T*)feats[mod] : (const T*)t->tables[mod], e->F[mod],
                             feats ? nullptr : b.ids, M, M, d, e->F[mod], G + e->Wf[mod], acc, nullptr, st, G + e->bf + mod * d, 0, 0,
                             e->fp8 ? t->table_scales[mod] : 0.f));
        }
    }
    RUN(flush_reduces<T>(e, b, st));
    RUN(join_side_all<T>(e, b, st));       // the caller's stream sees every gradient
    if (e->grad_cb_fine()) e->grad_ready(0, L > 0 ? e->layers[0].Wqkvc : e->Wn);         // embeddings bucket
    else if (e->grad_cb) e->grad_ready(0, whole_buffer ? e->total : e->Wn);            // side-stream reductions: one bucket
    return 0;
}

template <typename T>
static int pretrain_step(pmgt_engine* e, const pmgt_tensors* t, const pmgt_batch* bt, const pmgt_outputs* o, void* ws,
                         int64_t ws_bytes, int flags, hipStream_t st) {
    const bool train = flags & PMGT_FLAG_TRAINING, bwd = flags & PMGT_FLAG_BACKWARD, acc = flags & PMGT_FLAG_ACCUMULATE;
    const int B = bt->n_targets, Pn = bt->n_pairs, S = bt->seq_len, d = e->d, F = e->Fsum;
    PMGT_CHECK(B > 0 && S > 0, -2, "pretrain_step: empty batch");
    PMGT_CHECK(!bwd || train, -2, "pretrain_step: BACKWARD requires TRAINING (the workspace keeps activations only then)");
    PMGT_CHECK(Pn > 0 && bt->pair_ids && bt->labels && bt->num_pairs, -2,
               "labels must be passed, when set pair_node_inputs (pmgt/pmgt/models.py:66-72)");
    const int Tseq = B + Pn + (train ? B : 0);
    Carver c(ws);
    Bufs<T> b;
    carve<T>(e, c, b, Tseq, S, B, train);
    PMGT_CHECK(c.cur <= ws_bytes, -4, "workspace too small: need %lld bytes, got %lld", (long long)c.cur, (long long)ws_bytes);
    const int64_t bs = (int64_t)B * S, ps = (int64_t)Pn * S;
    {   // the collated batch -> the step's [Tseq, S] id / mask arrays, one launch
        CopyJob cj[8];
        int nc = 0;
        cj[nc++] = CopyJob{bt->tgt_ids, b.ids, bs * 8};
        cj[nc++] = CopyJob{bt->tgt_mask, b.mask, bs * 4};
        cj[nc++] = CopyJob{bt->pair_ids, b.ids + bs, ps * 8};
        cj[nc++] = CopyJob{bt->pair_mask, b.mask + bs, ps * 4};
        if (train) {
            cj[nc++] = CopyJob{bt->tgt_mask, b.mask + bs + ps, bs * 4};                      // models.py:153-156: the masked copy keeps the mask
            if (bt->nfr_masked_ids) {
                PMGT_CHECK(bt->nfr_targets, -2, "nfr_targets must accompany nfr_masked_ids");
                cj[nc++] = CopyJob{bt->nfr_masked_ids, b.ids + bs + ps, bs * 8};
                cj[nc++] = CopyJob{bt->nfr_targets, b.nfr_tgt, bs * 8};
            }
        }
        RUN(multi_copy(cj, nc, st));
    }
    if (train) {
        if (!bt->nfr_masked_ids)
            RUN(nfr_generate(bt->tgt_ids, B, S, (int)t->n_nodes, bt->random_node_ratio, bt->mask_node_ratio, t->rng_state,
                             b.ids + bs + ps, b.nfr_tgt, st));
        RUN(nfr_compact(b.nfr_tgt, B, S, B + Pn, b.nfr_rows, b.nfr_tids, b.nfr_count, st));
    }
    // Table mode backward needs the tokens ordered by node id; the ids are final here, so the (latency-bound, many small
    // launches) stable sort runs on the engine's side stream next to the whole forward pass.
    b.sort_done = nullptr;
    if (bwd && use_table_projection(e, t, (int64_t)Tseq * S, true) && !(e->opts & OPT_NO_SEGMENT_SUM) && e->side) {
        hipEvent_t ev = e->next_sync();
        PMGT_HIP(hipEventRecord(ev, st));
        PMGT_HIP(hipStreamWaitEvent(e->side, ev, 0));
        RUN(seg_sort(b.ids, Tseq * S, (int)t->n_nodes + 2, b.sg_keys, b.sg_vals, b.sg_skeys, b.sg_perm, b.sg_off, b.sg_tmp, b.sg_tmp_bytes, e->side));
        b.sort_done = e->next_sync();
        PMGT_HIP(hipEventRecord(b.sort_done, e->side));
    }
    RUN(build_mirrors<T>(e, t, b, st));
    // Training fast path: the caller does not ask for last_hidden_state, so the last layer's attn-out/FFN blocks
    // only run on the rows the loss reads (compact order: B target CLS, P pair CLS, masked rows).
    const bool sc = train && !(e->opts & OPT_NO_SHORTCUT) && o->last_hidden == nullptr;
    if (sc) RUN(build_need_rows(B, Pn, S, b.nfr_rows, b.nfr_count, b.need_rows, b.need_cnt, st, b.need_inv, (int64_t)Tseq * S));
    RUN(encoder_forward<T>(e, t, b, Tseq, S, b.ids, nullptr, b.mask, train, (T*)nullptr, (float*)nullptr, st, sc, B + Pn));
    T* hL = sc ? b.ctail.hout : b.layer[e->L - 1].hout;
    T* dhL = sc ? b.c_dh : b.bA;
    const int M = Tseq * S;
    if (bwd) PMGT_HIP(hipMemsetAsync(dhL, 0, (size_t)(sc ? b.rcap : M) * d * sizeof(T), st));
    RUN(pair_offsets(bt->num_pairs, B, b.off, st));
    {
        GsrArgs g;
        g.h = hL; g.dh = bwd ? dhL : nullptr; g.B = B; g.S = S; g.d = d; g.off = b.off; g.labels = bt->labels;
        g.cls_stride = sc ? d : (int64_t)S * d;
        g.logits = o->logits; g.loss_part = b.gsr_part;
        RUNP("loss.gsr", gsr_fwd_bwd<T>(g, st));
    }
    const int cap = B * std::max(S - 1, 1);
    // rows of the masked positions inside hL: gathered by token index, or contiguous after the CLS rows when compacted
    const T* hN = sc ? hL + (int64_t)(B + Pn) * d : hL;
    const int64_t* nrows = sc ? nullptr : b.nfr_rows;
    if (train) {
        GemmNT g; g.opts = e->opts;   // projections of the masked rows
        g.A = hN; g.lda = d; g.a_rows = nrows; g.B = wsel<T>(e, t, b, e->Wn, e->mWn); g.ldb = d;
        g.C = b.pred; g.ldc = F; g.M = cap; g.N = F; g.K = d; g.bias = t->params + e->bn; g.m_dev = b.nfr_count;
        RUNP("loss.gemm_nfr", gemm_nt<T>(g, st));
        NfrDiffArgs a;
        a.pred = b.pred; a.tids = b.nfr_tids; a.count = b.nfr_count; a.cap = cap; a.nf = e->NF; a.sse_part = b.sse_part;
        a.tables_f8 = e->fp8;
        for (int m = 0; m < e->NF; ++m) { a.F[m] = e->F[m]; a.table[m] = t->tables[m]; a.scale[m] = t->table_scales[m]; }
        RUNP("loss.nfr_diff", nfr_diff<T>(a, st));
    }
    FeatSizes fs;
    fs.nf = e->NF;
    for (int m = 0; m < MAX_FEATS; ++m) fs.F[m] = e->F[m];
    RUN(loss_finish(b.gsr_part, B, train ? b.sse_part : nullptr, nfr_diff_parts(cap), b.nfr_count, fs, train,
                    o->loss, st, train ? o->nfr_count : nullptr));
    if (o->last_hidden) PMGT_HIP(hipMemcpyAsync(o->last_hidden, hL, (size_t)bs * d * sizeof(T), hipMemcpyDeviceToDevice, st));
    if (bwd) {
        const int msp = std::max(256, cap / 5);
        RUN(wgrad<T>("bwd.wgrad_nfr", e, b, b.pred, F, hN, d, nrows, cap, msp, F, d, t->grads + e->Wn, acc, b.nfr_count, st, t->grads + e->bn));
        RUN(flush_reduces<T>(e, b, st));
        if (e->grad_cb_fine()) e->grad_ready(e->Wn, e->total - e->Wn);          // NFR head bucket
        GemmNT g; g.opts = e->opts;
        g.A = b.pred; g.lda = F; g.B = b.mirror + e->mWnT; g.ldb = F; g.M = cap; g.N = d; g.K = F; g.m_dev = b.nfr_count;
        g.C = sc ? dhL + (int64_t)(B + Pn) * d : b.dq; g.ldc = d;      // compacted: the masked rows ARE rows B+P.. of dhL
        RUNP("bwd.dgrad_nfr", gemm_nt<T>(g, st));
        if (!sc) RUN(scatter_rows<T>(b.dq, b.nfr_rows, b.nfr_count, cap, d, b.bA, st));
        RUN(encoder_backward<T>(e, t, b, Tseq, S, acc, st, sc, true, nullptr, B + Pn, true));
    }
    if (train) RUN(advance_rng(t->rng_state, st));
    return 0;
}

template <typename T>
static int encode(pmgt_engine* e, const pmgt_tensors* t, const int64_t* ids, const void* const* feats,
                  const float* mask, int Tseq, int S, void* last_hidden, void* hidden_states, float* attn_probs, void* ws,
                  int64_t ws_bytes, hipStream_t st) {
    PMGT_CHECK(Tseq > 0 && S > 0, -2, "encode: empty input");
    Carver c(ws);
    Bufs<T> b;
    carve<T>(e, c, b, Tseq, S, 1, false);
    PMGT_CHECK(c.cur <= ws_bytes, -4, "workspace too small: need %lld bytes, got %lld", (long long)c.cur, (long long)ws_bytes);
    const float* m = mask;
    if (!m) {   // attention_mask=None -> ones (pmgt/pmgt/modeling_pmgt.py:113-114)
        m = nullptr;
    }
    RUN(build_mirrors<T>(e, t, b, st));
    RUN(encoder_forward<T>(e, t, b, Tseq, S, ids, feats, m, false, (T*)hidden_states, attn_probs, st));
    if (last_hidden)
        PMGT_HIP(hipMemcpyAsync(last_hidden, b.layer[e->L - 1].hout, (size_t)Tseq * S * e->d * sizeof(T),
                                hipMemcpyDeviceToDevice, st));
    return 0;
}


// Training-mode encoder pass for a caller with its own head (PMGT_NCF, pmgt/pmgt_ncf/models.py:77-105): keeps the
// activations in the workspace and snapshots the dropout counter so encode_backward can replay the masks.
template <typename T>
static int encode_train(pmgt_engine* e, const pmgt_tensors* t, const int64_t* ids, const void* const* feats,
                        const float* mask, int Tseq, int S, void* last_hidden, void* ws, int64_t ws_bytes, int flags,
                        hipStream_t st) {
    const bool train = flags & PMGT_FLAG_TRAINING;
    PMGT_CHECK(Tseq > 0 && S > 0, -2, "encode_train: empty input");
    PMGT_CHECK(mask != nullptr, -2, "encode_train: attention_mask is required (pass ones for None)");
    PMGT_CHECK((ids != nullptr) != (feats != nullptr), -2, "encode_train: pass node ids or the feature tensors");
    Carver c(ws);
    Bufs<T> b;
    carve<T>(e, c, b, Tseq, S, 1, true);
    PMGT_CHECK(c.cur <= ws_bytes, -4, "workspace too small: need %lld bytes, got %lld", (long long)c.cur, (long long)ws_bytes);
    const int64_t M = (int64_t)Tseq * S;
    if (e->fwd_opts.size() > 64) e->fwd_opts.clear();      // (callers that never run the backward: keep the table small)
    e->fwd_opts[ws] = e->opts & ~pmgt_engine::SCHED_OPTS;
    PMGT_HIP(hipMemcpyAsync(b.rng_snap, t->rng_state, 16, hipMemcpyDeviceToDevice, st));
    if (ids) PMGT_HIP(hipMemcpyAsync(b.ids, ids, M * 8, hipMemcpyDeviceToDevice, st));
    PMGT_HIP(hipMemcpyAsync(b.mask, mask, M * 4, hipMemcpyDeviceToDevice, st));
    RUN(build_mirrors<T>(e, t, b, st));
    RUN(encoder_forward<T>(e, t, b, Tseq, S, ids ? b.ids : nullptr, feats, b.mask, train, (T*)nullptr, (float*)nullptr, st));
    if (last_hidden)
        PMGT_HIP(hipMemcpyAsync(last_hidden, b.layer[e->L - 1].hout, (size_t)M * e->d * sizeof(T), hipMemcpyDeviceToDevice, st));
    if (train) RUN(advance_rng(t->rng_state, st));
    return 0;
}

template <typename T>
static int encode_backward(pmgt_engine* e, const pmgt_tensors* t, const void* const* feats, const void* d_last,
                           int Tseq, int S, void* ws, int64_t ws_bytes, int flags, hipStream_t st) {
    const bool train = flags & PMGT_FLAG_TRAINING, acc = flags & PMGT_FLAG_ACCUMULATE;
    PMGT_CHECK(Tseq > 0 && S > 0 && d_last != nullptr, -2, "encode_backward: empty input");
    {
        auto itr = e->fwd_opts.find(ws);
        PMGT_CHECK(itr == e->fwd_opts.end() || itr->second == (e->opts & ~pmgt_engine::SCHED_OPTS), -5,
                   "encode_backward: engine options changed since the forward pass over this workspace (0x%x -> 0x%x): the backward would "
                   "read buffers the forward did not write; run the forward again", itr->second, e->opts & ~pmgt_engine::SCHED_OPTS);
    }
    Carver c(ws);
    Bufs<T> b;
    carve<T>(e, c, b, Tseq, S, 1, true);
    PMGT_CHECK(c.cur <= ws_bytes, -4, "workspace too small: need %lld bytes, got %lld", (long long)c.cur, (long long)ws_bytes);
    pmgt_tensors tt = *t;
    tt.rng_state = (uint64_t*)b.rng_snap;          // the forward's (seed, step)
    b.e_by_id = use_table_projection(e, t, (int64_t)Tseq * S, feats == nullptr);      // same decisions as the forward took
    b.qkvc_hm = train && !(e->opts & OPT_NO_HEAD_MAJOR) && fused_qa_applies<T>(e, Tseq, S, false);
    b.qkvc_vc = vc_only_applies<T>(e, Tseq, S, false);
    PMGT_HIP(hipMemcpyAsync(b.bA, d_last, (size_t)Tseq * S * e->d * sizeof(T), hipMemcpyDeviceToDevice, st));
    RUN(encoder_backward<T>(e, &tt, b, Tseq, S, acc, st, false, train, feats));
    return 0;
}

}  // namespace pmgt

static bool tables_set(const pmgt_engine* e, const pmgt_tensors* t) {
    for (int m = 0; m < e->NF; ++m) if (!t->tables[m]) return false;
    return true;
}
static bool feats_set(const pmgt_engine* e, const void* const* feats) {
    if (!feats) return false;
    for (int m = 0; m < e->NF; ++m) if (!feats[m]) return false;
    return true;
}

// ======================================================================================================
// C ABI
// ======================================================================================================
extern "C" {

const char* pmgt_last_error(void) { return g_err; }
int pmgt_abi_version(void) { return 4; }

pmgt_engine* pmgt_engine_create(const pmgt_config* cfg) {
    if (!cfg) { set_error("config is NULL"); return nullptr; }
    if (cfg->hidden_size <= 0 || cfg->num_attention_heads <= 0 || cfg->hidden_size % cfg->num_attention_heads != 0) {
        set_error("The hidden size (%d) is not a multiple of the number of attention heads (%d)", cfg->hidden_size,
                  cfg->num_attention_heads);   // pmgt/pmgt/modeling_pmgt.py:381-387
        return nullptr;
    }
    const int ali = cfg->dtype == PMGT_DTYPE_FP8 ? 16 : (cfg->dtype == PMGT_DTYPE_BF16 ? 8 : 4);
    if (cfg->n_feats < 1 || cfg->n_feats > PMGT_MAX_FEATS) {
        set_error("feat_hidden_sizes has %d entries: the HIP path takes 1 .. %d modalities", cfg->n_feats, PMGT_MAX_FEATS);
        return nullptr;
    }
    bool feats_ok = true;
    for (int m = 0; m < cfg->n_feats; ++m) feats_ok = feats_ok && cfg->feat_sizes[m] > 0 && cfg->feat_sizes[m] % ali == 0;
    if (cfg->hidden_size % ali || cfg->intermediate_size % ali || !feats_ok || cfg->hidden_size > 1024) {
        set_error("HIP path needs hidden/intermediate/feature sizes that are multiples of %d and hidden_size <= 1024", ali);
        return nullptr;
    }
    if ((int64_t)embed_part_elems(cfg->hidden_size, cfg->n_feats) * 4 > 65536) {
        set_error("hidden_size %d with %d modalities: the embedding backward's workgroup sums exceed 64 KB of LDS", cfg->hidden_size, cfg->n_feats);
        return nullptr;
    }
    if (cfg->dtype != PMGT_DTYPE_F32 && cfg->dtype != PMGT_DTYPE_BF16 && cfg->dtype != PMGT_DTYPE_FP8) { set_error("unknown dtype %d", cfg->dtype); return nullptr; }
    pmgt_engine* e = new pmgt_engine();
    e->cfg = *cfg;
    e->fp8 = cfg->dtype == PMGT_DTYPE_FP8;
    e->d = cfg->hidden_size; e->L = cfg->num_hidden_layers; e->H = cfg->num_attention_heads; e->I = cfg->intermediate_size;
    e->dh = e->d / e->H;
    e->NF = cfg->n_feats;
    for (int m = 0; m < e->NF; ++m) { e->F[m] = cfg->feat_sizes[m]; e->Foff[m] = e->Fsum; e->Fsum += e->F[m]; }
    build_layout(e);
    e->zeros = zero_page();
    if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess) e->side = nullptr;
    if (!e->desc.empty()) {
        if (hipMalloc((void**)&e->desc_dev, e->desc.size() * sizeof(MirrorDesc)) != hipSuccess ||
            hipMemcpy(e->desc_dev, e->desc.data(), e->desc.size() * sizeof(MirrorDesc), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("pmgt_engine_create: no usable HIP device (hipMalloc/hipMemcpy failed)");
            delete e;
            return nullptr;
        }
    }
    if (!e->desc8.empty()) {
        if (hipMalloc((void**)&e->desc8_dev, e->desc8.size() * sizeof(QuantDesc)) != hipSuccess ||
            hipMemcpy(e->desc8_dev, e->desc8.data(), e->desc8.size() * sizeof(QuantDesc), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("pmgt_engine_create: no usable HIP device (hipMalloc/hipMemcpy failed)");
            delete e;
            return nullptr;
        }
    }
    return e;
}

void pmgt_engine_destroy(pmgt_engine* e) {
    if (!e) return;
    if (e->desc_dev) (void)hipFree(e->desc_dev);
    if (e->desc8_dev) (void)hipFree(e->desc8_dev);
    if (e->side) { (void)hipStreamSynchronize(e->side); (void)hipStreamDestroy(e->side); }
    for (auto ev : e->sync_ev) (void)hipEventDestroy(ev);
    delete e;
}

int64_t pmgt_param_count(const pmgt_engine* e) { return e->total; }
int pmgt_param_num_entries(const pmgt_engine* e) { return (int)e->entries.size(); }
int pmgt_param_entry(const pmgt_engine* e, int index, char* name, int name_cap, int64_t* offset, int64_t* numel, int* rows,
                     int* cols, int* decay) {
    PMGT_CHECK(index >= 0 && index < (int)e->entries.size(), -2, "param index %d out of range", index);
    const ParamEntry& p = e->entries[index];
    if (name && name_cap > 0) { strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = p.offset;
    if (numel) *numel = p.numel;
    if (rows) *rows = p.rows;
    if (cols) *cols = p.cols;
    if (decay) *decay = p.decay;
    return 0;
}

int64_t pmgt_workspace_bytes(const pmgt_engine* e, int n_seq, int seq_len, int n_targets, int training) {
    Carver c(nullptr);
    if (e->cfg.dtype != PMGT_DTYPE_F32) { Bufs<bf16> b; carve<bf16>(e, c, b, n_seq, seq_len, std::max(n_targets, 1), training != 0); }
    else { Bufs<float> b; carve<float>(e, c, b, n_seq, seq_len, std::max(n_targets, 1), training != 0); }
    return c.cur;
}

int pmgt_pretrain_step(pmgt_engine* e, const pmgt_tensors* t, const pmgt_batch* b, const pmgt_outputs* o, void* workspace,
                       int64_t workspace_bytes, int flags, void* stream) {
    PMGT_CHECK(e && t && b && o && workspace, -2, "pmgt_pretrain_step: NULL argument");
    PMGT_CHECK(t->params && tables_set(e, t) && t->rng_state && o->loss && o->logits, -2, "pmgt_pretrain_step: NULL tensor");
    for (int m = 0; m < e->NF; ++m) PMGT_CHECK(!e->fp8 || t->table_scales[m] > 0.f, -2, "pmgt_pretrain_step: fp8 mode needs the table scales");
    PMGT_CHECK(!(flags & PMGT_FLAG_BACKWARD) || t->grads, -2, "pmgt_pretrain_step: grads buffer is NULL");
    if (e->cfg.dtype != PMGT_DTYPE_F32) return pretrain_step<bf16>(e, t, b, o, workspace, workspace_bytes, flags, (hipStream_t)stream);
    return pretrain_step<float>(e, t, b, o, workspace, workspace_bytes, flags, (hipStream_t)stream);
}

int pmgt_encode_ids(pmgt_engine* e, const pmgt_tensors* t, const int64_t* ids, const float* mask, int n_seq, int seq_len,
                    void* last_hidden, void* hidden_states, float* attn_probs, void* workspace, int64_t workspace_bytes,
                    void* stream) {
    PMGT_CHECK(e && t && ids && workspace && t->params && tables_set(e, t), -2, "pmgt_encode_ids: NULL argument");
    if (e->cfg.dtype != PMGT_DTYPE_F32)
        return encode<bf16>(e, t, ids, nullptr, mask, n_seq, seq_len, last_hidden, hidden_states, attn_probs, workspace, workspace_bytes, (hipStream_t)stream);
    return encode<float>(e, t, ids, nullptr, mask, n_seq, seq_len, last_hidden, hidden_states, attn_probs, workspace, workspace_bytes, (hipStream_t)stream);
}

int pmgt_encode_feats(pmgt_engine* e, const pmgt_tensors* t, const void* const* feats, const float* mask,
                      int n_seq, int seq_len, void* last_hidden, void* hidden_states, float* attn_probs, void* workspace,
                      int64_t workspace_bytes, void* stream) {
    PMGT_CHECK(e && t && feats_set(e, feats) && workspace && t->params, -2, "pmgt_encode_feats: NULL argument");
    if (e->cfg.dtype != PMGT_DTYPE_F32)
        return encode<bf16>(e, t, nullptr, feats, mask, n_seq, seq_len, last_hidden, hidden_states, attn_probs, workspace, workspace_bytes, (hipStream_t)stream);
    return encode<float>(e, t, nullptr, feats, mask, n_seq, seq_len, last_hidden, hidden_states, attn_probs, workspace, workspace_bytes, (hipStream_t)stream);
}

int pmgt_encode_train(pmgt_engine* e, const pmgt_tensors* t, const int64_t* ids, const void* const* feats,
                      const float* mask, int n_seq, int seq_len, void* last_hidden, void* workspace, int64_t workspace_bytes,
                      int flags, void* stream) {
    PMGT_CHECK(e && t && workspace && t->params && t->rng_state, -2, "pmgt_encode_train: NULL argument");
    PMGT_CHECK(!ids || tables_set(e, t), -2, "pmgt_encode_train: feature tables are not set");
    PMGT_CHECK(ids || feats_set(e, feats), -2, "pmgt_encode_train: pass node ids or one feature tensor per modality");
    if (ids) feats = nullptr;
    if (e->cfg.dtype != PMGT_DTYPE_F32)
        return encode_train<bf16>(e, t, ids, feats, mask, n_seq, seq_len, last_hidden, workspace, workspace_bytes, flags, (hipStream_t)stream);
    return encode_train<float>(e, t, ids, feats, mask, n_seq, seq_len, last_hidden, workspace, workspace_bytes, flags, (hipStream_t)stream);
}

int pmgt_encode_backward(pmgt_engine* e, const pmgt_tensors* t, const void* const* feats, const void* d_last_hidden,
                         int n_seq, int seq_len, void* workspace, int64_t workspace_bytes, int flags, void* stream) {
    PMGT_CHECK(e && t && workspace && t->params && t->grads, -2, "pmgt_encode_backward: NULL argument");
    PMGT_CHECK(feats ? feats_set(e, feats) : tables_set(e, t), -2, "pmgt_encode_backward: feature tables are not set");
    if (e->cfg.dtype != PMGT_DTYPE_F32)
        return encode_backward<bf16>(e, t, feats, d_last_hidden, n_seq, seq_len, workspace, workspace_bytes, flags, (hipStream_t)stream);
    return encode_backward<float>(e, t, feats, d_last_hidden, n_seq, seq_len, workspace, workspace_bytes, flags, (hipStream_t)stream);
}

int pmgt_optimizer_step(pmgt_engine* e, const pmgt_tensors* t, const pmgt_adam* a, void* stream) {
    PMGT_CHECK(e && t && a && t->params && t->grads && a->exp_avg && a->exp_avg_sq && a->decay && a->step && a->scalars && a->scratch,
               -2, "pmgt_optimizer_step: NULL argument");
    AdamArgs x;
    x.p = t->params; x.g = t->grads; x.m = a->exp_avg; x.v = a->exp_avg_sq; x.decay = a->decay; x.n = e->total;
    x.lr = a->lr; x.wd = a->weight_decay; x.b1 = a->beta1; x.b2 = a->beta2; x.eps = a->eps; x.max_norm = a->max_grad_norm;
    x.step = a->step; x.scal = a->scalars; x.part = a->scratch;
    hipStream_t st = (hipStream_t)stream;
    RUNP("optimizer.clip_adamw", adamw_step(x, st));
    return 0;
}

int pmgt_profile_begin(pmgt_engine* e) {
    e->prof.clear();
    e->prof.on = true;
    return 0;
}

// Stops collection, waits for the recorded events and writes one "name count total_ms" line per phase.
int pmgt_profile_end(pmgt_engine* e, char* buf, int cap) {
    e->prof.on = false;
    std::vector<std::string> names;
    std::vector<double> ms;
    std::vector<int> cnt;
    for (auto& r : e->prof.recs) {
        PMGT_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        PMGT_HIP(hipEventElapsedTime(&t, r.a, r.b));
        size_t k = 0;
        for (; k < names.size(); ++k) if (names[k] == r.name) break;
        if (k == names.size()) { names.push_back(r.name); ms.push_back(0.); cnt.push_back(0); }
        ms[k] += t; cnt[k] += 1;
    }
    e->prof.clear();
    std::string out;
    for (size_t k = 0; k < names.size(); ++k) {
        char line[256];
        snprintf(line, sizeof(line), "%s %d %.6f\n", names[k].c_str(), cnt[k], ms[k]);
        out += line;
    }
    if (buf && cap > 0) { strncpy(buf, out.c_str(), cap - 1); buf[cap - 1] = 0; }
    return 0;
}

// The recorded phases in LAUNCH ORDER, one name per line (call between pmgt_profile_begin and pmgt_profile_end; does not wait for anything):
// tools/make_traffic.py walks the dispatch order of a rocprofv3 counter run next to it to attribute a kernel's launches to phases.
int pmgt_profile_sequence(pmgt_engine* e, char* buf, int cap) {
    std::string out;
    for (auto& r : e->prof.recs) { out += r.name; out += '\n'; }
    PMGT_CHECK(buf && cap > (int)out.size(), -4, "pmgt_profile_sequence: %d bytes needed", (int)out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}
// The same records with their event times, "name ms" per line, in launch order (waits for the events; the records stay for pmgt_profile_end):
// a phase whose launches differ in size -- the last layer's dense blocks run on the compacted rows only -- is then priced on its median
// launch instead of on a mean that the small launch pulls down.
int pmgt_profile_records(pmgt_engine* e, char* buf, int cap) {
    std::string out;
    for (auto& r : e->prof.recs) {
        PMGT_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        PMGT_HIP(hipEventElapsedTime(&t, r.a, r.b));
        char line[160];
        snprintf(line, sizeof(line), "%s %.6f\n", r.name, (double)t);
        out += line;
    }
    PMGT_CHECK(buf && cap > (int)out.size(), -4, "pmgt_profile_records: %d bytes needed", (int)out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}

int pmgt_cast_from_f32(int dtype, const float* src, void* dst, int64_t n, void* stream) {
    if (dtype != PMGT_DTYPE_F32) return cast_f32<bf16>(src, (bf16*)dst, n, (hipStream_t)stream);
    return cast_f32<float>(src, (float*)dst, n, (hipStream_t)stream);
}
int pmgt_cast_to_f32(int dtype, const void* src, float* dst, int64_t n, void* stream) {
    if (dtype != PMGT_DTYPE_F32) return cast_to_f32<bf16>((const bf16*)src, dst, n, (hipStream_t)stream);
    return cast_to_f32<float>((const float*)src, dst, n, (hipStream_t)stream);
}

// ---- single-kernel entry points --------------------------------------------------------------------
int pmgt_op_gemm_nt(int dtype, const void* A, int64_t lda, const int64_t* a_rows, const void* B, int64_t ldb, void* C,
                    int64_t ldc, int M, int N, int K, const float* bias, int epilogue, void* aux, int64_t ldaux,
                    const void* residual, int64_t ldr, float drop_p, uint32_t drop_site, const uint64_t* rng, const int* m_dev,
                    uint32_t path_opts, void* stream) {
    GemmNT g; g.opts = path_opts;
    g.A = A; g.lda = lda; g.a_rows = a_rows; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.epi = epilogue; g.aux = aux; g.ldaux = ldaux; g.res = residual; g.ldr = ldr;
    g.drop = DropCfg{rng, rng ? drop_p : 0.f, drop_site}; g.m_dev = m_dev;
    if (dtype == PMGT_DTYPE_BF16) return gemm_nt<bf16>(g, (hipStream_t)stream);
    return gemm_nt<float>(g, (hipStream_t)stream);
}

int64_t pmgt_op_gemm_tn_slab_elems(int dtype, int M, int N1, int N2, uint32_t path_opts) { return tn_slab_elems(dtype, M, N1, N2, path_opts); }

int pmgt_op_gemm_tn(int dtype, const void* P, int64_t ldp, const void* Q, int64_t ldq, const int64_t* q_rows, int M, int N1,
                    int N2, float* slab, float* out, int accumulate, const int* m_dev, uint32_t path_opts, void* stream) {
    GemmTN g; g.opts = path_opts;
    g.P = P; g.ldp = ldp; g.Q = Q; g.ldq = ldq; g.q_rows = q_rows; g.M = M; g.N1 = N1; g.N2 = N2; g.slab = slab; g.m_dev = m_dev;
    g.zeros = zero_page();
    g.splits = gemm_tn_pick_splits(M, N1, N2, dtype == PMGT_DTYPE_BF16 ? 64 : 32, path_opts);
    int rc = dtype == PMGT_DTYPE_BF16 ? gemm_tn<bf16>(g, (hipStream_t)stream) : gemm_tn<float>(g, (hipStream_t)stream);
    if (rc) return rc;
    return slab_reduce(slab, g.splits, (int64_t)N1 * N2, out, accumulate != 0, (hipStream_t)stream);
}

// weight gradient + bias gradient (column sums of P riding along as ones-MFMAs) + optional head-major row permutation:
// exactly what the engine's wgrad helper launches for the Q|K|V|C projection
int pmgt_op_gemm_tn_bias(int dtype, const void* P, int64_t ldp, const void* Q, int64_t ldq, int M, int N1, int N2, float* slab,
                         float* out, float* bias_slab, float* bias_out, int perm_d, int perm_dh, uint32_t path_opts, void* stream) {
    GemmTN g; g.opts = path_opts;
    g.P = P; g.ldp = ldp; g.Q = Q; g.ldq = ldq; g.M = M; g.N1 = N1; g.N2 = N2; g.slab = slab; g.bias_slab = bias_slab;
    g.zeros = zero_page(); g.perm_d = perm_d; g.perm_dh = perm_dh;
    g.splits = gemm_tn_pick_splits(M, N1, N2, dtype != PMGT_DTYPE_F32 ? 64 : 32, path_opts);
    PMGT_CHECK(g.splits <= 512, -2, "pmgt_op_gemm_tn_bias: bias_slab holds at most 512 splits");
    int rc = dtype != PMGT_DTYPE_F32 ? gemm_tn<bf16>(g, (hipStream_t)stream) : gemm_tn<float>(g, (hipStream_t)stream);
    if (rc) return rc;
    rc = slab_reduce(slab, g.splits, (int64_t)N1 * N2, out, false, (hipStream_t)stream);
    if (rc || !bias_slab) return rc;
    return slab_reduce(bias_slab, g.splits, N1, bias_out, false, (hipStream_t)stream);
}

int pmgt_op_clock_probe(uint64_t* out, int blocks, void* stream) { return clock_probe(out, blocks, (hipStream_t)stream); }
int64_t pmgt_op_seg_sort_temp_bytes(int M) { return seg_sort_temp_bytes(M); }
int pmgt_op_seg_sort(const int64_t* ids, int M, int n_rows, uint32_t* scratch_keys, uint32_t* scratch_vals, uint32_t* skeys, uint32_t* perm,
                     int* seg_off, void* temp, int64_t temp_bytes, void* stream) {
    return seg_sort(ids, M, n_rows, scratch_keys, scratch_vals, skeys, perm, seg_off, temp, temp_bytes, (hipStream_t)stream);
}

int pmgt_op_colsum(int dtype, const void* Y, int64_t ldy, int M, int N, float* slab, float* out, void* stream) {
    if (dtype == PMGT_DTYPE_BF16) return colsum<bf16>((const bf16*)Y, ldy, M, N, slab, out, false, nullptr, (hipStream_t)stream);
    return colsum<float>((const float*)Y, ldy, M, N, slab, out, false, nullptr, (hipStream_t)stream);
}

int pmgt_op_layernorm_fwd(int dtype, const void* x, void* y, float* stats, const float* gamma, const float* beta, int M,
                          int d, float eps, float drop_p, uint32_t drop_site, const uint64_t* rng, void* stream) {
    DropCfg dc{rng, rng ? drop_p : 0.f, drop_site};
    if (dtype == PMGT_DTYPE_BF16) return ln_fwd<bf16>((const bf16*)x, (bf16*)y, stats, gamma, beta, M, d, eps, dc, (hipStream_t)stream);
    return ln_fwd<float>((const float*)x, (float*)y, stats, gamma, beta, M, d, eps, dc, (hipStream_t)stream);
}

int pmgt_op_layernorm_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                          void* dx_drop, float* part, float* dgamma_dbeta, int M, int d, float in_drop_p, uint32_t in_site,
                          float out_drop_p, uint32_t out_site, const uint64_t* rng, void* stream) {
    DropCfg di{rng, rng ? in_drop_p : 0.f, in_site}, dout{rng, rng ? out_drop_p : 0.f, out_site};
    int rc;
    if (dtype == PMGT_DTYPE_BF16)
        rc = ln_bwd<bf16>((const bf16*)dy, (const bf16*)x, stats, gamma, (bf16*)dx, (bf16*)dx_drop, part, M, d, di, dout, (hipStream_t)stream);
    else
        rc = ln_bwd<float>((const float*)dy, (const float*)x, stats, gamma, (float*)dx, (float*)dx_drop, part, M, d, di, dout, (hipStream_t)stream);
    if (rc) return rc;
    return slab_reduce(part, ln_bwd_parts(M), 3 * d, dgamma_dbeta, false, (hipStream_t)stream);
}

static AttnArgs mk_attn(const void* qkvc, const float* mask, int n_seq, int S, int H, int dh, float beta, float drop_p,
                        uint32_t s1, uint32_t s2, const uint64_t* rng, uint32_t path_opts = 0) {
    AttnArgs a; a.opts = path_opts;
    a.qkvc = qkvc; a.mask = mask; a.Tseq = n_seq; a.S = S; a.H = H; a.dh = dh; a.beta = beta;
    a.drop1 = DropCfg{rng, rng ? drop_p : 0.f, s1};
    a.drop2 = DropCfg{rng, rng ? drop_p : 0.f, s2};
    return a;
}

// ---- path options: per-engine state, no process globals, no environment reads ------------------------
static_assert(PMGT_OPT_TILE_GEMM == OPT_TILE_GEMM && PMGT_OPT_VALU_ATTENTION == OPT_VALU_ATTENTION && PMGT_OPT_WAVE_ATTENTION_BWD == OPT_WAVE_ATTENTION_BWD &&
              PMGT_OPT_NO_SHORTCUT == OPT_NO_SHORTCUT && PMGT_OPT_NO_FUSED_QKVC_ATTENTION == OPT_NO_FUSED_QKVC_ATTENTION && PMGT_OPT_NO_HEAD_MAJOR == OPT_NO_HEAD_MAJOR &&
              PMGT_OPT_NO_TABLE_PROJECTION == OPT_NO_TABLE_PROJECTION && PMGT_OPT_NO_SEGMENT_SUM == OPT_NO_SEGMENT_SUM && PMGT_OPT_CONSUMER_QUANT == OPT_CONSUMER_QUANT &&
              PMGT_OPT_NO_FUSED_ATTENTION_BWD == OPT_NO_FUSED_ATTENTION_BWD && PMGT_OPT_STORE_LN_INPUT == OPT_STORE_LN_INPUT && PMGT_OPT_EAGER_REDUCE == OPT_EAGER_REDUCE &&
              PMGT_OPT_SIDE_STREAM_REDUCE == OPT_SIDE_STREAM_REDUCE && PMGT_OPT_UNFUSED_LN == OPT_UNFUSED_LN && PMGT_OPT_ONE_BUCKET == OPT_ONE_BUCKET && PMGT_OPT_SMALL_ARENA == OPT_SMALL_ARENA && PMGT_OPT_NO_ROLE_SPLIT_LN == OPT_NO_ROLE_SPLIT_LN && PMGT_OPT_NO_TILE_ATTENTION == OPT_NO_TILE_ATTENTION,
              "include/pmgt_ops.h and csrc/common.h disagree on the option bits");
static uint32_t option_bit(const char* key) {
    static const struct { const char* name; uint32_t bit; } tab[] = {
        {"tile_gemm", OPT_TILE_GEMM}, {"valu_attention", OPT_VALU_ATTENTION}, {"wave_attention_bwd", OPT_WAVE_ATTENTION_BWD},
        {"no_shortcut", OPT_NO_SHORTCUT}, {"no_fused_qkvc_attention", OPT_NO_FUSED_QKVC_ATTENTION}, {"no_head_major", OPT_NO_HEAD_MAJOR},
        {"no_table_projection", OPT_NO_TABLE_PROJECTION}, {"no_segment_sum", OPT_NO_SEGMENT_SUM}, {"consumer_quant", OPT_CONSUMER_QUANT},
        {"no_fused_attention_bwd", OPT_NO_FUSED_ATTENTION_BWD}, {"store_ln_input", OPT_STORE_LN_INPUT}, {"eager_reduce", OPT_EAGER_REDUCE},
        {"side_stream_reduce", OPT_SIDE_STREAM_REDUCE}, {"unfused_ln", OPT_UNFUSED_LN}, {"one_bucket", OPT_ONE_BUCKET}, {"small_arena", OPT_SMALL_ARENA}, {"no_role_split_ln", OPT_NO_ROLE_SPLIT_LN}, {"no_tile_attention", OPT_NO_TILE_ATTENTION}, {"unfused_ln_bwd", OPT_UNFUSED_LN_BWD}, {"lockstep_attention_bwd", OPT_LOCKSTEP_ATTENTION_BWD}, {"side_stream_wgrad", OPT_SIDE_STREAM_WGRAD}, {"no_cls_only_attention_bwd", OPT_NO_CLS_ONLY_ATTENTION_BWD}, {"no_beta_skip", OPT_NO_BETA_SKIP}, {"no_vc2_attention_bwd", OPT_NO_VC2_ATTENTION_BWD}};
    for (const auto& t : tab)
        if (key && strcmp(key, t.name) == 0) return t.bit;
    return 0;
}
int pmgt_engine_set_option(pmgt_engine* e, const char* key, int value) {
    PMGT_CHECK(e != nullptr, -2, "pmgt_engine_set_option: NULL engine");
    const uint32_t bit = option_bit(key);
    PMGT_CHECK(bit != 0, -2, "pmgt_engine_set_option: unknown option '%s'", key ? key : "(null)");
    e->opts = value ? (e->opts | bit) : (e->opts & ~bit);
    return 0;
}
void pmgt_launch_trace_reset(void) { for (auto& c : g_launch_count) c = 0; }
int64_t pmgt_launch_trace_count(const char* family) {
    for (int i = 0; i < LT_COUNT; ++i)
        if (family && strcmp(family, g_launch_names[i]) == 0) return g_launch_count[i];
    set_error("pmgt_launch_trace_count: unknown kernel family '%s'", family ? family : "(null)");
    return -1;
}
int pmgt_engine_get_option(const pmgt_engine* e, const char* key) {
    PMGT_CHECK(e != nullptr, -2, "pmgt_engine_get_option: NULL engine");
    const uint32_t bit = option_bit(key);
    PMGT_CHECK(bit != 0, -2, "pmgt_engine_get_option: unknown option '%s'", key ? key : "(null)");
    return (e->opts & bit) ? 1 : 0;
}
void pmgt_engine_set_grad_ready_callback(pmgt_engine* e, pmgt_grad_ready_fn cb, void* user) {
    if (e) { e->grad_cb = cb; e->grad_cb_user = user; }
}

int pmgt_op_linear(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                   const float* bias, int epilogue, void* aux, int64_t ldaux, const void* residual, int64_t ldr, float drop_p,
                   uint32_t drop_site, const uint64_t* rng, void* ln_out, float* ln_stats, const float* ln_gamma,
                   const float* ln_beta, float ln_eps, uint32_t path_opts, void* stream) {
    // (a NULL output is a GPU fault, not an error code: skip_c -- "do not store the LayerNorm input" -- is an engine-internal form)
    PMGT_CHECK(A && B && C, -2, "pmgt_op_linear: NULL operand (A, B and C are required)");
    PMGT_CHECK(epilogue == 0 || aux, -2, "pmgt_op_linear: epilogue %d needs aux", epilogue);
    GemmWS g; g.opts = path_opts;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.epi = epilogue; g.aux = aux; g.ldaux = ldaux; g.res = residual; g.ldr = ldr;
    g.drop = DropCfg{rng, rng ? drop_p : 0.f, drop_site};
    g.ln_out = ln_out; g.ln_stats = ln_stats; g.ln_gamma = ln_gamma; g.ln_beta = ln_beta; g.ln_eps = ln_eps;
    pmgt_engine dummy;
    const pmgt_engine* e = &dummy;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PMGT_DTYPE_BF16) return linear<bf16>(e, "op.linear", g, st);
    return linear<float>(e, "op.linear", g, st);
}

int pmgt_op_linear_ln_bwd(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, const void* residual, int64_t ldr,
                          const void* y, const float* stats, const float* gamma, const float* beta, void* dy_tmp, void* dx, void* dx_drop,
                          float drop_p, uint32_t drop_site, const uint64_t* rng, float* part, float* dgamma_dbeta_dbias,
                          uint32_t path_opts, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    GemmWS g; g.opts = path_opts;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = dx; g.ldc = N; g.M = M; g.N = N; g.K = K; g.res = residual; g.ldr = ldr;
    g.lnb_y = y; g.lnb_ldy = N; g.lnb_stats = stats; g.lnb_gamma = gamma; g.lnb_beta = beta;
    g.lnb_dx_drop = dx_drop; g.lnb_lddx = N; g.lnb_drop = DropCfg{rng, rng ? drop_p : 0.f, drop_site};
    g.lnb_part = part;
    pmgt_engine dummy;
    const pmgt_engine* e = &dummy;
    if (gemm_wsr_lnb_ok(g)) {
        RUN(gemm_wsr_lnb(g, st));
        return slab_reduce(part, gemm_wsr_lnb_parts(M), 3 * N, dgamma_dbeta_dbias, false, st);
    }
    if (gemm_nt_lnb_ok(g)) {
        RUN(gemm_nt_lnb(g, st));
        return slab_reduce(part, gemm_nt_lnb_parts(M), 3 * N, dgamma_dbeta_dbias, false, st);
    }
    // the two-launch form: dy through HBM, then the standalone LayerNorm backward with x^ from the output
    g.C = dy_tmp;
    RUN(linear<bf16>(e, "op.linear", g, st));
    RUN(ln_bwd<bf16>((const bf16*)dy_tmp, (const bf16*)y, stats, gamma, (bf16*)dx, (bf16*)dx_drop, part, M, N, DropCfg{nullptr, 0.f, 0}, g.lnb_drop, st, nullptr, beta));
    return slab_reduce(part, ln_bwd_parts(M), 3 * N, dgamma_dbeta_dbias, false, st);
}

int pmgt_op_attention_fwd(int dtype, const void* qkvc, const float* mask, void* ctx, float* probs, int n_seq, int S, int H,
                          int dh, float beta, float drop_p, uint32_t site1, uint32_t site2, const uint64_t* rng, uint32_t path_opts, void* stream) {
    AttnArgs a = mk_attn(qkvc, mask, n_seq, S, H, dh, beta, drop_p, site1, site2, rng, path_opts);
    a.ctx = ctx; a.probs = probs;
    if (dtype == PMGT_DTYPE_BF16) return attn_fwd<bf16>(a, (hipStream_t)stream);
    return attn_fwd<float>(a, (hipStream_t)stream);
}

int pmgt_op_qkvc_attention_fwd(const void* x, const void* w, const float* bias, const float* mask, void* qkvc, void* ctx, int n_seq, int S,
                               int H, int dh, float beta, float drop_p, uint32_t site1, uint32_t site2, const uint64_t* rng, void* stream) {
    QkvcAttn f;
    const int d = H * dh;
    f.X = x; f.ldx = d; f.W = w; f.ldw = d; f.bias = bias; f.qkvc = qkvc; f.ldq = 4 * d; f.ctx = ctx; f.ldc = d; f.mask = mask;
    f.Tseq = n_seq; f.S = S; f.H = H; f.dh = dh; f.beta = beta;
    f.drop1 = DropCfg{rng, drop_p, site1};
    f.drop2 = DropCfg{rng, drop_p, site2};
    PMGT_CHECK(x && w && qkvc && ctx, -2, "pmgt_op_qkvc_attention_fwd: NULL argument");
    PMGT_CHECK(qkvc_attn_supported(f), -3, "pmgt_op_qkvc_attention_fwd: unsupported shape (needs bf16, S = 32, dh = 32, d in {128, 256})");
    return qkvc_attn_fwd(f, (hipStream_t)stream);
}
int pmgt_op_qkvc_attention_fwd_ex(const void* x, const void* w, const float* bias, const float* mask, void* qkvc, void* ctx, int n_seq, int S,
                                  int H, int dh, float beta, float drop_p, uint32_t site1, uint32_t site2, const uint64_t* rng, int flags, void* stream) {
    QkvcAttn f;
    const int d = H * dh;
    f.X = x; f.ldx = d; f.W = w; f.ldw = d; f.bias = bias; f.qkvc = qkvc; f.ldq = 4 * d; f.ctx = ctx; f.ldc = d; f.mask = mask;
    f.Tseq = n_seq; f.S = S; f.H = H; f.dh = dh; f.beta = beta;
    f.drop1 = DropCfg{rng, drop_p, site1};
    f.drop2 = DropCfg{rng, drop_p, site2};
    f.hm = (flags & 1) != 0; f.vc_only = (flags & 2) != 0;
    PMGT_CHECK(x && w && qkvc && ctx, -2, "pmgt_op_qkvc_attention_fwd_ex: NULL argument");
    PMGT_CHECK(qkvc_attn_supported(f), -3, "pmgt_op_qkvc_attention_fwd_ex: unsupported shape / flags (bf16, S = 32, dh = 32, d in {128, 256}; vc_only: beta == 1, H %% 4 == 0)");
    return qkvc_attn_fwd(f, (hipStream_t)stream);
}

int pmgt_op_attention_bwd(int dtype, const void* qkvc, const float* mask, const void* dctx, void* dqkvc, int n_seq, int S,
                          int H, int dh, float beta, float drop_p, uint32_t site1, uint32_t site2, const uint64_t* rng,
                          uint32_t path_opts, void* stream) {
    AttnArgs a = mk_attn(qkvc, mask, n_seq, S, H, dh, beta, drop_p, site1, site2, rng, path_opts);
    a.dctx = dctx; a.dqkvc = dqkvc;
    if (dtype == PMGT_DTYPE_BF16) return attn_bwd<bf16>(a, (hipStream_t)stream);
    return attn_bwd<float>(a, (hipStream_t)stream);
}

int pmgt_op_attention_bwd_wgrad_parts(int H) { return attn_bwd_wgrad_parts(H); }
int pmgt_op_attention_bwd_wgrad_vc2_parts(int H) { return attn_bwd_wgrad_vc2_parts(H); }
int pmgt_op_attention_bwd_wgrad(const void* qkvc, const float* mask, const void* dctx, const void* x, void* dqkvc, float* slab,
                                float* bias_slab, int n_seq, int H, float beta, float drop_p, uint32_t site1, uint32_t site2,
                                const uint64_t* rng, int head_major, void* stream) {
    AttnBwdWg w;
    w.a = mk_attn(qkvc, mask, n_seq, 32, H, 32, beta, drop_p, site1, site2, rng);
    w.a.dctx = dctx; w.a.dqkvc = dqkvc; w.a.hm = (head_major & 1) != 0; w.a.vc_only = (head_major & 2) != 0;
    w.x = x; w.ldx = (int64_t)H * 32; w.slab = slab; w.bias_slab = bias_slab;
    PMGT_CHECK(attn_bwd_wgrad_supported(w), -3, "pmgt_op_attention_bwd_wgrad: unsupported shape (needs bf16, S = 32, dh = 32, d in {128, 256}, n_seq >= 2)");
    if (head_major & 4) {      // two-heads-per-step form of the beta == 1 mode: partial sums [vc2_parts][2 d, d] (value | ctx_attention rows)
        PMGT_CHECK(attn_bwd_wgrad_vc2_supported(w), -3, "pmgt_op_attention_bwd_wgrad: the two-heads-per-step form needs vc_only (bit 1) and beta == 1");
        return attn_bwd_wgrad_vc2(w, (hipStream_t)stream);
    }
    return attn_bwd_wgrad(w, (hipStream_t)stream);
}

// ---- fp8 mode ---------------------------------------------------------------------------------------
int pmgt_quantize_e4m3(const float* src, void* dst, int64_t n, float inv_scale, void* stream) {
    return quant_tensor_e4m3(src, dst, n, inv_scale, (hipStream_t)stream);
}
int pmgt_dequantize_e4m3(const void* src, float* dst, int64_t n, float scale, void* stream) {
    return dequant_tensor_e4m3(src, dst, n, scale, (hipStream_t)stream);
}
int pmgt_op_quant_rows_e4m3(int src_dtype, const void* src, int64_t lds, int rows, int cols, void* dst, int64_t ldd, float* scale,
                            void* stream) {
    if (src_dtype == PMGT_DTYPE_F32) return quant_rows_e4m3<float>((const float*)src, lds, rows, cols, dst, ldd, scale, (hipStream_t)stream);
    return quant_rows_e4m3<bf16>((const bf16*)src, lds, rows, cols, dst, ldd, scale, (hipStream_t)stream);
}
int pmgt_op_gemm_nt_f8(const void* A, int64_t lda, const int64_t* a_rows, const float* a_row_scale, float a_scale, const void* B,
                       int64_t ldb, const float* b_row_scale, void* C, int64_t ldc, int M, int N, int K, const float* bias,
                       const int* m_dev, void* stream) {
    GemmF8 g;
    g.A = A; g.lda = lda; g.a_rows = a_rows; g.a_row_scale = a_row_scale; g.a_scale = a_scale; g.B = B; g.ldb = ldb;
    g.b_row_scale = b_row_scale; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.m_dev = m_dev;
    return gemm_nt_f8(g, (hipStream_t)stream);
}
int pmgt_op_gemm_tn_f8(const void* P, int64_t ldp, const void* Q8, int64_t ldq, float q_scale, const int64_t* q_rows, int M, int N1,
                       int N2, float* slab, float* out, int accumulate, const int* m_dev, void* stream) {
    GemmTN g;
    g.P = P; g.ldp = ldp; g.Q = Q8; g.ldq = ldq; g.q_rows = q_rows; g.M = M; g.N1 = N1; g.N2 = N2; g.slab = slab; g.m_dev = m_dev;
    g.q_f8 = true; g.q_scale = q_scale;
    g.splits = gemm_tn_pick_splits(M, N1, N2, 64);
    int rc = gemm_tn<bf16>(g, (hipStream_t)stream);
    if (rc) return rc;
    return slab_reduce(slab, g.splits, (int64_t)N1 * N2, out, accumulate != 0, (hipStream_t)stream);
}
int pmgt_op_qkvc_attention_fwd_f8(const void* x, const void* x8, const float* xscale, const void* w8, const float* wscale, const float* bias, const float* mask, void* qkvc,
                                  void* ctx, int n_seq, int S, int H, int dh, float beta, float drop_p, uint32_t site1, uint32_t site2,
                                  const uint64_t* rng, void* stream) {
    QkvcAttn f;
    const int d = H * dh;
    f.X = x; f.X8 = x8; f.xscale = xscale; f.ldx = d; f.W8 = w8; f.wscale = wscale; f.ldw = d; f.bias = bias; f.qkvc = qkvc; f.ldq = 4 * d; f.ctx = ctx; f.ldc = d;
    f.mask = mask; f.Tseq = n_seq; f.S = S; f.H = H; f.dh = dh; f.beta = beta;
    f.drop1 = DropCfg{rng, drop_p, site1};
    f.drop2 = DropCfg{rng, drop_p, site2};
    PMGT_CHECK((x || (x8 && xscale)) && w8 && wscale && qkvc && ctx, -2, "pmgt_op_qkvc_attention_fwd_f8: NULL argument");
    PMGT_CHECK(qkvc_attn_supported(f), -3, "pmgt_op_qkvc_attention_fwd_f8: unsupported shape (needs S = 32, dh = 32, d = 256)");
    return qkvc_attn_fwd(f, (hipStream_t)stream);
}

}  // extern "C"
