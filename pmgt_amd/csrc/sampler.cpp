// Host MCNSampling for PMGT pre-training (libpmgt_sampler.so; pure C++17, no HIP).
//
// Re-implements pmgt/pmgt/datasets.py:14-208 of the reference on a CSR graph and reproduces the
// reference's random streams bit for bit WITHOUT numpy: the reference draws from numpy's legacy
// process-global MT19937 (np.random.seed / choice / randint), so this file carries
//   * MT19937 with numpy's legacy integer seeding (init_genrand),
//   * random_sample():  ((a >> 5) * 2^26 + (b >> 6)) / 2^53 from two 32-bit outputs,
//   * choice(a, n, replace=True, p): cdf = cumsum(p) / cdf[-1]; searchsorted(cdf, u, 'right'),
//   * randint(n) / permutation(): masked rejection of 32-bit draws (Fisher-Yates from the top),
//   * scipy softmax in float64 with numpy's pairwise summation order,
// as listed in SURVEY.md Appendix C.  Tested bit-exact against numpy, the oracle and the golden
// vectors (tests/test_sampler.py).  Per-node CDFs are computed once at creation (the reference
// recomputes softmax + cumsum on every visit), which does not change any draw.
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/pmgt_capi.h"

namespace {

thread_local char g_err[512] = "";
void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- numpy legacy MT19937 --------------------------------------------------------------------------
struct MT19937 {
    uint32_t key[624];
    uint32_t out[624];                 // tempered outputs of the current state block (filled by gen(), vectorisable)
    int pos;
    void seed(uint32_t s) {            // numpy _legacy_seeding(int) -> mt19937_seed (init_genrand)
        for (int i = 0; i < 624; ++i) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
    void gen() {
        const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, A = 0x9908b0dfu;
        int i;
        uint32_t y;
        for (i = 0; i < 624 - 397; ++i) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
        }
        for (; i < 623; ++i) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
        }
        y = (key[623] & UPPER) | (key[0] & LOWER);
        key[623] = key[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
        for (i = 0; i < 624; ++i) {    // tempering of the whole block at once
            uint32_t t = key[i];
            t ^= (t >> 11);
            t ^= (t << 7) & 0x9d2c5680u;
            t ^= (t << 15) & 0xefc60000u;
            t ^= (t >> 18);
            out[i] = t;
        }
        pos = 0;
    }
    inline uint32_t next32() {
        if (pos == 624) gen();
        return out[pos++];
    }
    inline double next_double() {
        const uint32_t a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    // legacy random_interval / masked bounded draw: uniform integer in [0, max]
    inline uint64_t interval(uint64_t max) {
        if (max == 0) return 0;
        uint64_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        uint64_t v;
        if (max <= 0xffffffffull) {
            while ((v = (next32() & mask)) > max) {}
        } else {
            do {
                const uint64_t hi = next32(), lo = next32();     // legacy next_uint64 = (hi << 32) | lo
                v = ((hi << 32) | lo) & mask;
            } while (v > max);
        }
        return v;
    }
};

// ---- generator of the THREADED entry (per-target streams; no reference stream to reproduce there: the reference samples
// sequentially from the global np.random, which is what MT19937 above restates).  xoshiro256++ seeded by splitmix64 from
// (base seed, item counter): one 64-bit output per uniform instead of two tempered MT words -- the Mersenne Twister was a third
// of the CPU time of a context (1 312 words for its 656 weighted draws), and the live input pipeline is bound by exactly that
// CPU time on a 16-CPU share.  Same algorithm, same distributions (53-bit uniforms, masked-rejection bounded integers).
struct Xoshiro256pp {
    uint64_t s[4];
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    void seed(uint64_t base, uint64_t ctr) {
        uint64_t z = base * 0xD1342543DE82EF95ull + 0x9E3779B97F4A7C15ull * (ctr + 1);
        for (int i = 0; i < 4; ++i) {      // splitmix64
            z += 0x9E3779B97F4A7C15ull;
            uint64_t t = z;
            t = (t ^ (t >> 30)) * 0xBF58476D1CE4E5B9ull;
            t = (t ^ (t >> 27)) * 0x94D049BB133111EBull;
            s[i] = t ^ (t >> 31);
        }
    }
    inline uint64_t next64() {
        const uint64_t r = rotl(s[0] + s[3], 23) + s[0], t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    inline double next_double() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
    inline uint64_t interval(uint64_t max) {      // uniform integer in [0, max]
        if (max == 0) return 0;
        uint64_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        uint64_t v;
        while ((v = (next64() >> 11) & mask) > max) {}
        return v;
    }
};

// numpy pairwise summation (DOUBLE_pairwise_sum) — the order np.sum uses on a contiguous 1-D array
double pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

struct Scratch {     // per-thread sampling state
    MT19937 rng;
    struct Stamp { int32_t cnt, cnt_ver, sc_idx, sc_ver; };      // one 16-byte record per node: a visit touches one cache line
    std::vector<Stamp> st;                                          // dense per-node stamps
    int32_t ver = 0;
    std::vector<int64_t> cur, nxt, hop_order;
    std::vector<std::pair<int64_t, int64_t>> scores;      // (node, score) in first-scored order
    std::vector<int64_t> perm;
    std::vector<int32_t> order, hist;
    void ensure(int64_t n) {
        if ((int64_t)st.size() < n) {
            st.assign(n, Stamp{0, 0, 0, 0});
            ver = 0;
        }
    }
};

// Persistent workers of the threaded entry: spawning n threads per batch cost ~1 ms of a 6 ms batch, and workers that
// compete at equal priority with the process's launch thread and the HIP runtime threads stretch the GPU step
// (round-1 measurement: 64 sampler threads -> step 10.9 -> 13.1 ms).  Workers live as long as the handle, sleep on a
// condition variable between batches and run at nice +10, so launch-side threads win whenever the CPU share is short.
struct WorkerPool {
    std::mutex mu, run_mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> th;
    const std::function<void(int)>* job = nullptr;
    uint64_t gen = 0;
    int active = 0, pending = 0;
    bool stop = false;
    void worker(int idx) {
        (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), 10);      // this thread only (Linux: per-task nice)
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || (gen != seen && idx < active); });
                if (stop) return;
                seen = gen;
                j = job;
            }
            (*j)(idx + 1);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    // fn(0) on the caller, fn(1 .. n_threads - 1) on pool workers; returns when all are done
    void run(int n_threads, const std::function<void(int)>& fn) {
        std::lock_guard<std::mutex> serial(run_mu);              // one batch at a time per handle
        const int extra = n_threads - 1;
        {
            std::lock_guard<std::mutex> lk(mu);
            while ((int)th.size() < extra) { const int idx = (int)th.size(); th.emplace_back([this, idx] { worker(idx); }); }
            job = &fn; active = extra; pending = extra; ++gen;
        }
        if (extra > 0) cv_work.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
        active = 0; job = nullptr;
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_work.notify_all();
        for (auto& t : th) t.join();
    }
};

}  // namespace

struct pmgt_sampler {
    int64_t n_nodes = 0;
    std::vector<int64_t> indptr, indices;
    std::vector<double> cdf;          // per-edge: normalised cumulative softmax of the row (build-time only: moved into `edge`)
    // One 16-byte record per edge -- {cdf, guide, neighbour} -- so that a draw touches ONE run of the row instead of three arrays
    // (the hop-3 sources are 128 random rows per context: the sampler is bound by their cache misses, not by arithmetic)
    struct Edge { double cdf; int32_t guide; int32_t nbr; };
    std::vector<Edge> edge;
    // guide[b + k] = first index i of the row with cdf[i] > k / deg: the exact searchsorted(cdf, u, 'right') of a draw
    // u in [k / deg, (k + 1) / deg) is found by scanning forward from there (1-2 compares instead of a binary search)
    std::vector<int32_t> guide;
    std::vector<int> hops;
    int max_ctx = 0, max_total = 10, min_neg = 5;
    Scratch main;                      // the sequential (reference-order) stream
    // scratch objects of the threaded entry, kept across calls (their dense per-node stamp arrays are 16 B per node:
    // re-creating them per call costs more than the sampling itself on million-node graphs)
    std::mutex pool_mu;
    std::vector<std::unique_ptr<Scratch>> pool;
    // first pair row of every target of a threaded call (kept across calls; one threaded call at a time per handle)
    std::mutex stage_mu;
    std::vector<int64_t> st_off;
    WorkerPool workers;               // declared last: destroyed (joined) first
    // sorted adjacency for the negative-sampling membership test
    std::vector<int64_t> sorted_idx;

    inline int64_t deg(int64_t v) const { return indptr[v + 1] - indptr[v]; }
    bool is_neighbor(int64_t u, int64_t v) const {
        const int64_t* b = sorted_idx.data() + indptr[u];
        const int64_t* e = sorted_idx.data() + indptr[u + 1];
        return std::binary_search(b, e, v);
    }
};

namespace {

// pmgt/pmgt/datasets.py:14-53.  R = the uniform source: the sequential entries pass the numpy legacy stream (bit-exact with the
// reference), the threaded entry a per-target xoshiro stream.
template <class R>
int sample_context(const pmgt_sampler* s, Scratch& sc, R& rng, int64_t target, int64_t* ids, float* mask) {
    const int depth = (int)s->hops.size();
    const int S = s->max_ctx + 1;
    if (target < 2 || target >= s->n_nodes + 2) { set_err("target id %lld out of range", (long long)target); return -2; }
    sc.ensure(s->n_nodes + 2);
    sc.scores.clear();
    sc.cur.assign(1, target);
    if (++sc.ver == INT32_MAX) { for (auto& e : sc.st) { e.sc_ver = 0; e.cnt_ver = 0; } sc.ver = 1; }
    const int32_t score_ver = sc.ver;
    for (int k = 1; k <= depth; ++k) {
        const int size = s->hops[k - 1];
        const size_t ncur = sc.cur.size();
        sc.nxt.resize(ncur * (size_t)size);
        int64_t* wr = sc.nxt.data();
        for (size_t ci = 0; ci < ncur; ++ci) {
            const int64_t node = sc.cur[ci];
            if (ci + 2 < ncur) {      // the rows of the next sources are known: start their cache misses now
                const int64_t nb = s->indptr[sc.cur[ci + 2]];
                __builtin_prefetch(&s->edge[nb]);
                __builtin_prefetch(&s->edge[nb] + 4);
            }
            const int64_t b = s->indptr[node], dg = s->indptr[node + 1] - b;
            if (dg <= 0) { set_err("node %lld has no neighbours (the reference raises here)", (long long)node); return -3; }
            const pmgt_sampler::Edge* row = s->edge.data() + b;
            const double dgd = (double)dg;
            // the draws of one source row: uniforms first (the generator's block refill stays out of the search loop), then the
            // searches write straight into the output array (sized once per hop)
            double us[64];
            const int nd = size < 64 ? size : 64;
            for (int r0 = 0; r0 < size; r0 += nd) {
                const int nb_ = size - r0 < nd ? size - r0 : nd;
                for (int r = 0; r < nb_; ++r) us[r] = rng.next_double();
                for (int r = 0; r < nb_; ++r) {
                    const double u = us[r];
                    // searchsorted(cdf, u, side='right') = first index with cdf[i] > u.  cdf[dg - 1] is exactly 1 > u, so the scan
                    // needs no bound; its first steps are arithmetic, not branches (their count is random: every one mispredicted)
                    int64_t k = (int64_t)(u * dgd);
                    k = k < dg ? k : dg - 1;
                    int64_t idx = row[k].guide;
                    idx += row[idx].cdf <= u;
                    idx += row[idx].cdf <= u;
                    idx += row[idx].cdf <= u;
                    while (row[idx].cdf <= u) ++idx;
                    *wr++ = row[idx].nbr;
                }
            }
        }
        // Counter(sampled[k]) in first-appearance order
        if (++sc.ver == INT32_MAX) { set_err("stamp overflow"); return -4; }
        const int32_t cv = sc.ver;
        sc.hop_order.clear();
        for (int64_t v : sc.nxt) {
            Scratch::Stamp& e = sc.st[v];
            if (e.cnt_ver != cv) { e.cnt_ver = cv; e.cnt = 0; sc.hop_order.push_back(v); }
            ++e.cnt;
        }
        const int64_t w = depth - k + 1;
        for (int64_t v : sc.hop_order) {
            if (v == target) continue;
            Scratch::Stamp& e = sc.st[v];
            if (e.sc_ver != score_ver) {
                e.sc_ver = score_ver;
                e.sc_idx = (int32_t)sc.scores.size();
                sc.scores.emplace_back(v, 0);
            }
            sc.scores[e.sc_idx].second += (int64_t)e.cnt * w;
        }
        sc.cur.swap(sc.nxt);
    }
    if (sc.scores.empty()) { set_err("target %lld has no scored neighbour (reference raises at datasets.py:42)", (long long)target); return -3; }
    // sorted(scores.items(), key=score, reverse=True)[:max_ctx] with Python's stable sort: ties keep first-scored order.
    // Only the first max_ctx entries are read, so select them with the strict order (score desc, position asc) instead of
    // sorting everything (same result, n log k instead of n log n).
    const int num = (int)std::min<int64_t>((int64_t)sc.scores.size(), s->max_ctx);
    // Scores are small integers (frequency x hop weight, at most sum_k hops-product x weight), so the strict order
    // (score descending, first-scored position ascending) is a counting sort: histogram, threshold score, then one pass that
    // drops every candidate into its bucket in position order -- linear in the candidates, no comparisons.
    int64_t smax = 0;
    for (const auto& pr : sc.scores) smax = pr.second > smax ? pr.second : smax;
    sc.hist.assign((size_t)smax + 2, 0);
    for (const auto& pr : sc.scores) ++sc.hist[(size_t)pr.second];
    // first output slot of each score value, highest score first; `thr` = the lowest score that still gets slots
    int64_t thr = smax;
    {
        int32_t run = 0;
        for (int64_t v = smax; v >= 0; --v) {
            const int32_t c = sc.hist[(size_t)v];
            sc.hist[(size_t)v] = run;
            if (run < num) thr = v;
            run += c;
        }
    }
    sc.order.resize((size_t)num);
    for (size_t i = 0; i < sc.scores.size(); ++i) {
        const int64_t v = sc.scores[i].second;
        if (v < thr) continue;
        const int32_t slot = sc.hist[(size_t)v]++;
        if (slot < num) sc.order[(size_t)slot] = (int32_t)i;
    }
    ids[0] = target;
    for (int i = 0; i < s->max_ctx; ++i) ids[1 + i] = i < num ? sc.scores[sc.order[i]].first : 0;
    for (int i = 0; i < S; ++i) mask[i] = i <= num ? 1.f : 0.f;
    return num;
}

int max_pairs(const pmgt_sampler* s, int mode) { return mode == 0 ? s->max_total : (mode == 1 ? 2 : 0); }

// PMGTDataset.__getitem__ (pmgt/pmgt/datasets.py:113-165); returns number of pairs or <0
template <class R>
int sample_item(const pmgt_sampler* s, Scratch& sc, R& rng, int64_t target, int mode, int64_t* tgt_ids, float* tgt_mask,
                int64_t* pair_ids, float* pair_mask, float* labels) {
    const int S = s->max_ctx + 1;
    int rc = sample_context(s, sc, rng, target, tgt_ids, tgt_mask);
    if (rc < 0) return rc;
    if (mode == 2) return 0;
    const int k = mode == 0 ? (s->max_total - s->min_neg) : 1;
    const int64_t b = s->indptr[target], dg = s->indptr[target + 1] - b;
    const int npos = (int)std::min<int64_t>(k, dg);
    // np.random.choice(neigh, npos, replace=False) = permutation(len)[:npos]: Fisher-Yates from the top
    sc.perm.resize(dg);
    for (int64_t i = 0; i < dg; ++i) sc.perm[i] = i;
    for (int64_t i = dg - 1; i >= 1; --i) {
        const int64_t j = (int64_t)rng.interval((uint64_t)i);
        std::swap(sc.perm[i], sc.perm[j]);
    }
    int64_t pos_nodes[64];
    if (npos > 64) { set_err("too many positives"); return -5; }
    for (int i = 0; i < npos; ++i) pos_nodes[i] = s->indices[b + sc.perm[i]];
    int np_ = 0;
    for (int i = 0; i < npos; ++i, ++np_) {
        rc = sample_context(s, sc, rng, pos_nodes[i], pair_ids + (int64_t)np_ * S, pair_mask + (int64_t)np_ * S);
        if (rc < 0) return rc;
        labels[np_] = 1.f;
    }
    const int nneg = mode == 0 ? std::max(s->min_neg, s->max_total - npos) : 1;
    int64_t neg_nodes[64];
    if (nneg > 64) { set_err("too many negatives"); return -5; }
    for (int i = 0; i < nneg; ++i) {       // datasets.py:173-180 (all negatives are drawn before their contexts)
        int64_t cand = (int64_t)rng.interval((uint64_t)(s->n_nodes - 1)) + 2;
        while (s->is_neighbor(target, cand)) cand = (int64_t)rng.interval((uint64_t)(s->n_nodes - 1)) + 2;
        neg_nodes[i] = cand;
    }
    for (int i = 0; i < nneg; ++i, ++np_) {
        rc = sample_context(s, sc, rng, neg_nodes[i], pair_ids + (int64_t)np_ * S, pair_mask + (int64_t)np_ * S);
        if (rc < 0) return rc;
        labels[np_] = 0.f;
    }
    return np_;
}


}  // namespace

extern "C" {

const char* pmgt_sampler_last_error(void) { return g_err; }

pmgt_sampler* pmgt_sampler_create(int64_t n_nodes, const int64_t* indptr, const int64_t* indices, const double* weights,
                                  const int* hop_sizes, int n_hops, int max_ctx_neigh, int max_total_samples,
                                  int min_neg_samples) {
    if (n_nodes <= 0 || !indptr || !indices || !weights || !hop_sizes || n_hops <= 0 || max_ctx_neigh <= 0) {
        set_err("pmgt_sampler_create: bad arguments");
        return nullptr;
    }
    pmgt_sampler* s = new pmgt_sampler();
    s->n_nodes = n_nodes;
    s->indptr.assign(indptr, indptr + n_nodes + 3);
    const int64_t nnz = indptr[n_nodes + 2];
    s->indices.assign(indices, indices + nnz);
    s->hops.assign(hop_sizes, hop_sizes + n_hops);
    s->max_ctx = max_ctx_neigh;
    s->max_total = max_total_samples;
    s->min_neg = min_neg_samples;
    s->cdf.resize(nnz);
    s->guide.resize(nnz);
    s->sorted_idx = s->indices;
    std::vector<double> ex;
    for (int64_t v = 0; v < n_nodes + 2; ++v) {
        const int64_t b = indptr[v], dg = indptr[v + 1] - b;
        if (dg <= 0) continue;
        for (int64_t i = 0; i < dg; ++i) {
            const int64_t u = indices[b + i];
            if (u < 2 || u >= n_nodes + 2) { set_err("neighbour id %lld out of range", (long long)u); delete s; return nullptr; }
        }
        // scipy.special.softmax (float64): exp(x - max) / sum, np.sum = pairwise
        double mx = weights[b];
        for (int64_t i = 1; i < dg; ++i) mx = weights[b + i] > mx ? weights[b + i] : mx;
        ex.resize(dg);
        for (int64_t i = 0; i < dg; ++i) ex[i] = exp(weights[b + i] - mx);
        const double tot = pairwise_sum(ex.data(), dg);
        // choice(): cdf = p.cumsum(); cdf /= cdf[-1]
        double run = 0.;
        for (int64_t i = 0; i < dg; ++i) { run += ex[i] / tot; s->cdf[b + i] = run; }
        const double last = s->cdf[b + dg - 1];
        for (int64_t i = 0; i < dg; ++i) s->cdf[b + i] /= last;
        for (int64_t k = 0; k < dg; ++k)
            s->guide[b + k] = (int32_t)(std::upper_bound(s->cdf.data() + b, s->cdf.data() + b + dg, (double)k / (double)dg) - (s->cdf.data() + b));
        std::sort(s->sorted_idx.begin() + b, s->sorted_idx.begin() + b + dg);
    }
    s->edge.resize(nnz);
    // guide SHIFTED by one bucket: a draw u computes k = floor(u * deg) in floating point, which may come out one too large when
    // u * deg rounds up across a bucket edge; starting the forward scan at the guide of bucket k - 1 is right for both cases
    // (the answer of bucket k is never before it) and saves the exact k / deg > u check -- a division per draw
    for (int64_t v = 0; v < n_nodes + 2; ++v) {
        const int64_t b = indptr[v], dg = indptr[v + 1] - b;
        for (int64_t k = 0; k < dg; ++k)
            s->edge[b + k] = pmgt_sampler::Edge{s->cdf[b + k], s->guide[b + (k > 0 ? k - 1 : 0)], (int32_t)s->indices[b + k]};
    }
    std::vector<double>().swap(s->cdf);
    std::vector<int32_t>().swap(s->guide);
    s->main.rng.seed(0);
    return s;
}

void pmgt_sampler_destroy(pmgt_sampler* s) { delete s; }
void pmgt_sampler_seed(pmgt_sampler* s, uint32_t seed) { s->main.rng.seed(seed); }
int pmgt_sampler_max_pairs(const pmgt_sampler* s, int mode) { return max_pairs(s, mode); }
double pmgt_sampler_random_sample(pmgt_sampler* s) { return s->main.rng.next_double(); }
int64_t pmgt_sampler_randint(pmgt_sampler* s, int64_t n) { return (int64_t)s->main.rng.interval((uint64_t)(n - 1)); }

int pmgt_sampler_context(pmgt_sampler* s, int64_t target, int64_t* ids, float* mask) {
    return sample_context(s, s->main, s->main.rng, target, ids, mask);
}

int pmgt_sampler_batch(pmgt_sampler* s, const int64_t* targets, int n, int mode, int64_t* tgt_ids, float* tgt_mask,
                       int64_t* pair_ids, float* pair_mask, int64_t* num_pairs, float* labels) {
    const int S = s->max_ctx + 1;
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        const int rc = sample_item(s, s->main, s->main.rng, targets[i], mode, tgt_ids + (int64_t)i * S, tgt_mask + (int64_t)i * S,
                                   pair_ids ? pair_ids + tot * S : nullptr, pair_mask ? pair_mask + tot * S : nullptr,
                                   labels ? labels + tot : nullptr);
        if (rc < 0) return rc;
        if (num_pairs) num_pairs[i] = rc;
        tot += rc;
    }
    return (int)tot;
}

int pmgt_sampler_batch_mt(pmgt_sampler* s, const int64_t* targets, int n, int mode, uint64_t base_seed, uint64_t counter,
                          uint64_t counter_stride, int n_threads, int64_t* tgt_ids, float* tgt_mask, int64_t* pair_ids, float* pair_mask,
                          int64_t* num_pairs, float* labels) {
    const int S = s->max_ctx + 1, mp = max_pairs(s, mode);
    if (n <= 0) return 0;
    n_threads = std::max(1, std::min(n_threads, n));
    // The number of pair rows of a target depends on its degree only (datasets.py:128-147: min(k, deg) positives, then
    // max(min_neg, max_total - positives) negatives), so every target's first output row is known BEFORE sampling: the workers
    // write straight into the caller's arrays.  (The staged form -- private slots, then a sequential compaction of ~4 MB per
    // 1 024-target batch on the calling thread -- sat on the critical path of the live input pipeline.)
    std::lock_guard<std::mutex> stage_lock(s->stage_mu);
    std::vector<int64_t>& off = s->st_off;
    off.resize((size_t)n + 1);
    off[0] = 0;
    for (int i = 0; i < n; ++i) {
        int c = 0;
        const int64_t t = targets[i];
        if (mode != 2 && t >= 2 && t < s->n_nodes + 2) {
            const int64_t dg = s->deg(t);
            const int k = mode == 0 ? (s->max_total - s->min_neg) : 1;
            const int npos = (int)std::min<int64_t>(k, dg);
            c = npos + (mode == 0 ? std::max(s->min_neg, s->max_total - npos) : 1);
        }
        off[i + 1] = off[i] + c;      // (an invalid target contributes no rows; its worker reports the error)
    }
    (void)mp;
    std::atomic<int> next(0), fail(0);
    std::vector<std::string> errs(n_threads);
    const std::function<void(int)> work = [&](int tid) {
        std::unique_ptr<Scratch> own;
        {
            std::lock_guard<std::mutex> lk(s->pool_mu);
            if (!s->pool.empty()) { own = std::move(s->pool.back()); s->pool.pop_back(); }
        }
        if (!own) own.reset(new Scratch());
        Scratch& sc = *own;
        struct Return {
            pmgt_sampler* s; std::unique_ptr<Scratch>& o;
            ~Return() { std::lock_guard<std::mutex> lk(s->pool_mu); s->pool.push_back(std::move(o)); }
        } ret{s, own};
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || fail.load()) break;
            Xoshiro256pp rng;
            rng.seed(base_seed, counter + counter_stride * (uint64_t)i);
            const int64_t o = off[i];
            const int rc = sample_item(s, sc, rng, targets[i], mode, tgt_ids + (int64_t)i * S, tgt_mask + (int64_t)i * S,
                                       pair_ids ? pair_ids + o * S : nullptr, pair_mask ? pair_mask + o * S : nullptr, labels ? labels + o : nullptr);
            if (rc < 0) { errs[tid] = g_err; fail.store(rc); break; }
            if (rc != (int)(off[i + 1] - o)) { errs[tid] = "internal: pair count differs from its degree formula"; fail.store(-6); break; }
            if (num_pairs) num_pairs[i] = rc;
        }
    };
    s->workers.run(n_threads, work);
    if (fail.load()) {
        for (auto& e : errs) if (!e.empty()) { set_err("%s", e.c_str()); break; }
        return fail.load();
    }
    return (int)off[n];
}

int pmgt_train_valid_split(int64_t n_nodes, double valid_size, uint32_t seed, int64_t* train_out, int64_t* valid_out) {
    if (n_nodes <= 0) return -1;
    MT19937 rng;
    rng.seed(seed);
    std::vector<int64_t> perm(n_nodes);
    for (int64_t i = 0; i < n_nodes; ++i) perm[i] = i;
    for (int64_t i = n_nodes - 1; i >= 1; --i) std::swap(perm[i], perm[(int64_t)rng.interval((uint64_t)i)]);
    const int64_t n_test = (int64_t)ceil(valid_size * (double)n_nodes);
    for (int64_t i = 0; i < n_test; ++i) valid_out[i] = perm[i] + 2;
    for (int64_t i = n_test; i < n_nodes; ++i) train_out[i - n_test] = perm[i] + 2;
    return (int)n_test;
}

}  // extern "C"
