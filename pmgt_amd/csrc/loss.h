// Loss heads (declarations); see loss.hip.
#pragma once
#include "common.h"

namespace pmgt {

struct CopyJob { const void* src; void* dst; int64_t bytes; };
int multi_copy(const CopyJob* jobs, int njobs, hipStream_t st);      // <= 8 device-to-device copies (4-byte granularity) in one launch
int pair_offsets(const int64_t* num_pairs, int B, int* off, hipStream_t st);
int nfr_generate(const int64_t* ids, int B, int S, int n_nodes, float random_ratio, float mask_ratio,
                 const uint64_t* rng, int64_t* masked_ids, int64_t* tgt_full, hipStream_t st);
int nfr_compact(const int64_t* tgt_full, int B, int S, int seq_off, int64_t* rows, int64_t* tids, int* count,
                hipStream_t st);

struct GsrArgs {
    const void* h = nullptr;        // [Tseq*S, d] encoder output; sequences 0..B-1 = targets, B.. = pairs
    void* dh = nullptr;             // same shape, pre-zeroed; CLS rows receive the gradient (nullable: eval)
    int B = 0, S = 0, d = 0;
    int64_t cls_stride = 0;         // elements between consecutive CLS rows (S*d in the full layout, d when compacted)
    const int* off = nullptr;       // [B+1] pair offsets
    const float* labels = nullptr;  // [P]
    float* logits = nullptr;        // [P] out
    float* loss_part = nullptr;     // [B] out: loss_i / B
};
template <typename T> int gsr_fwd_bwd(const GsrArgs& g, hipStream_t st);

constexpr int MAX_FEATS = 4;        // == PMGT_MAX_FEATS (include/pmgt_capi.h)
struct NfrDiffArgs {
    void* pred = nullptr;           // [cap, sum F_m] in: projections of every modality side by side; out: d loss / d pred
    const int64_t* tids = nullptr;  // [cap] ids to reconstruct
    const int* count = nullptr;     // device row count
    int cap = 0, nf = 0;
    int F[MAX_FEATS] = {0, 0, 0, 0};
    const void* table[MAX_FEATS] = {nullptr, nullptr, nullptr, nullptr};     // table[m]: [N+2, F[m]]
    bool tables_f8 = false;         // fp8 mode: the tables are e4m3 bytes, value = byte * scale[m]
    float scale[MAX_FEATS] = {1.f, 1.f, 1.f, 1.f};
    float* sse_part = nullptr;      // [nfr_diff_parts(cap)][MAX_FEATS]
};
inline int nfr_diff_parts(int cap) { return cdiv(cap, 8); }
template <typename T> int nfr_diff(const NfrDiffArgs& a, hipStream_t st);
template <typename T>
int scatter_rows(const T* src, const int64_t* rows, const int* count, int cap, int d, T* dst, hipStream_t st, bool add = false);
// rows the loss reads from the last layer: CLS of the B targets, CLS of the P pairs, then the masked rows
// `inv` (optional, [n_tokens] ints): filled with the inverse map token row -> compact row, -1 for rows that are not needed
int build_need_rows(int B, int P, int S, const int64_t* nfr_rows, const int* nfr_count, int64_t* rows, int* count,
                    hipStream_t st, int* inv = nullptr, int64_t n_tokens = 0);
struct FeatSizes { int nf; int F[MAX_FEATS]; };
int loss_finish(const float* gsr_part, int B, const float* sse_part, int nparts, const int* count, FeatSizes fs,
                bool with_nfr, float* out, hipStream_t st, int* count_out = nullptr);

}  // namespace pmgt
