// Sanitizer harness for the host sampler (ASAN / TSAN builds of the C++ sampler and its threads).
// Built by tests/test_sanitizers_cpu.py together with pmgt_amd/csrc/sampler.cpp, once with -fsanitize=thread and once
// with -fsanitize=address,undefined; exits 0 when every check holds (the sanitizer runtime turns a finding into a
// non-zero exit).  What it drives concurrently:
//   * handle A: pmgt_sampler_batch_mt with 8 workers, from TWO caller threads at once (a handle is thread-safe per
//     handle: calls serialise, worker pool and staging buffers are shared);
//   * handle B: the sequential reference-order entry (pmgt_sampler_batch / pmgt_sampler_context) on its own thread;
//   * handle C: created, used with a different worker count each call, destroyed -- while A and B run;
// and what it checks: the threaded entry is deterministic in (base_seed, counter) whatever the thread count and
// interleaving, the sequential entry replays after a re-seed, and bad targets come back as error codes.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/pmgt_capi.h"

namespace {
const int N = 400, S = 16, MP = 10;

struct Batch {
    std::vector<int64_t> tid, pid, np;
    std::vector<float> tmk, pmk, lab;
    explicit Batch(int n) : tid((size_t)n * S), pid((size_t)n * MP * S), np(n), tmk((size_t)n * S), pmk((size_t)n * MP * S), lab((size_t)n * MP) {}
    bool operator==(const Batch& o) const { return tid == o.tid && pid == o.pid && np == o.np && tmk == o.tmk && lab == o.lab; }
};

pmgt_sampler* make() {
    // ring + chords: every node has degree >= 4, ids 2 .. N + 1
    std::vector<std::vector<int64_t>> adj(N + 2);
    auto add = [&](int u, int v) { adj[u + 2].push_back(v + 2); adj[v + 2].push_back(u + 2); };
    for (int i = 0; i < N; ++i) { add(i, (i + 1) % N); add(i, (i + 7) % N); }
    std::vector<int64_t> indptr(N + 3, 0), idx;
    std::vector<double> w;
    for (int v = 0; v < N + 2; ++v) {
        for (size_t k = 0; k < adj[v].size(); ++k) { idx.push_back(adj[v][k]); w.push_back(0.3 + 0.1 * (double)((v * 31 + k * 17) % 11)); }
        indptr[v + 1] = (int64_t)idx.size();
    }
    const int hops[3] = {16, 8, 4};
    return pmgt_sampler_create(N, indptr.data(), idx.data(), w.data(), hops, 3, S - 1, MP, 5);
}

int run_mt(pmgt_sampler* s, const std::vector<int64_t>& tg, int threads, uint64_t seed, uint64_t ctr, Batch& b) {
    return pmgt_sampler_batch_mt(s, tg.data(), (int)tg.size(), 0, seed, ctr, 1, threads, b.tid.data(), b.tmk.data(), b.pid.data(),
                                 b.pmk.data(), b.np.data(), b.lab.data());
}
}  // namespace

int main() {
    std::atomic<int> failures(0);
    auto fail = [&](const char* what) { fprintf(stderr, "FAIL: %s\n", what); failures.fetch_add(1); };
    pmgt_sampler *A = make(), *B = make();
    if (!A || !B) { fprintf(stderr, "create failed: %s\n", pmgt_sampler_last_error()); return 2; }
    std::vector<int64_t> tg(96);
    for (size_t i = 0; i < tg.size(); ++i) tg[i] = 2 + (int64_t)((i * 37) % N);

    Batch ref(96);
    if (run_mt(A, tg, 1, 99, 1000, ref) <= 0) fail("reference batch");

    auto hammer_a = [&](int threads) {
        for (int it = 0; it < 12; ++it) {
            Batch b(96);
            if (run_mt(A, tg, threads, 99, 1000, b) <= 0) fail("batch_mt");
            if (!(b == ref)) fail("threaded batch depends on thread count / interleaving");
        }
    };
    auto seq_b = [&] {
        Batch first(24), again(24);
        std::vector<int64_t> t24(tg.begin(), tg.begin() + 24);
        for (int it = 0; it < 6; ++it) {
            pmgt_sampler_seed(B, 5);
            Batch& dst = it == 0 ? first : again;
            if (pmgt_sampler_batch(B, t24.data(), 24, 0, dst.tid.data(), dst.tmk.data(), dst.pid.data(), dst.pmk.data(), dst.np.data(), dst.lab.data()) <= 0) fail("sequential batch");
            if (it > 0 && !(again == first)) fail("sequential stream does not replay after re-seed");
            int64_t ids[S];
            float mk[S];
            if (pmgt_sampler_context(B, 2 + it, ids, mk) < 0 || ids[0] != 2 + it) fail("context");
        }
        int64_t bad = (int64_t)N + 100, ids[S];
        float mk[S];
        if (pmgt_sampler_context(B, bad, ids, mk) >= 0) fail("out-of-range target accepted");
    };
    auto churn_c = [&] {
        for (int it = 0; it < 4; ++it) {
            pmgt_sampler* C = make();
            Batch b(96);
            if (run_mt(C, tg, 2 + 3 * it, 99, 1000, b) <= 0) fail("batch on fresh handle");
            if (!(b == ref)) fail("fresh handle disagrees");
            std::vector<int64_t> badt(tg);
            badt[40] = 1;                                   // <mask> id is not a node: the whole call fails with a code
            if (run_mt(C, badt, 4, 99, 1000, b) >= 0) fail("bad target accepted by batch_mt");
            pmgt_sampler_destroy(C);
        }
    };
    std::thread t1(hammer_a, 8), t2(hammer_a, 5), t3(seq_b), t4(churn_c);
    t1.join(); t2.join(); t3.join(); t4.join();
    pmgt_sampler_destroy(A);
    pmgt_sampler_destroy(B);
    if (failures.load()) return 1;
    printf("sanitize harness ok\n");
    return 0;
}
