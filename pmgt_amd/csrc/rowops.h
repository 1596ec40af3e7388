// Row-wise kernels of the PMGT engine: LayerNorm fwd/bwd, the modality-attention embedding mix
// (PMGTEmbeddings minus its two GEMMs), loss heads, optimizer.  Declarations.
#pragma once
#include "common.h"

namespace pmgt {

// ---- LayerNorm ------------------------------------------------------------------------------
// y = dropout(LN(x)); stats[m] = {mean, rstd}.  d % 4 == 0, d <= 1024.  One wave per row.
template <typename T>
int ln_fwd(const T* x, T* y, float* stats, const float* gamma, const float* beta, int M, int d, float eps,
           DropCfg out_drop, hipStream_t st, const int* m_dev = nullptr, void* q8 = nullptr, float* q8_scale = nullptr);
// (q8 / q8_scale: fp8 mode -- y additionally as per-row e4m3 [M, d] bytes + scales [M], fp8.h contract)
// dx = LN'(dy * in_mask); optional second output dx_drop = dx * out_mask (gradient of the dropout
// that fed the residual sum).  Partials go to `part` ([ln_bwd_parts(M)][3][d] floats): dgamma, dbeta and
// dbias = column sum of dx_drop (or dx): the bias gradient of the dense layer in front of the LayerNorm.
// `beta_y` != NULL: `x` is the LayerNorm OUTPUT y (what the next sublayer reads anyway) and x^ = (y - beta) / gamma -- the forward
// pass then never stores the pre-LayerNorm sum (1U of HBM writes per LayerNorm).  Only rstd of `stats` is read.  A channel with
// gamma == 0 carries no information about x^ in y: the kernels use x^ = 0 there only to stay finite -- the host guard
// (Engine.check_layernorm_carrier) never lets that form run on such parameters: it switches the engine to stored LayerNorm inputs when a
// gamma is (nearly) zero or |beta / gamma| > 8.
template <typename T>
int ln_bwd(const T* dy, const T* x, const float* stats, const float* gamma, T* dx, T* dx_drop, float* part,
           int M, int d, DropCfg in_drop, DropCfg out_drop, hipStream_t st, const int* m_dev = nullptr, const float* beta_y = nullptr);
// rows per workgroup (measured at M = 393k: 64 -> 189 us, 128 -> 192 us, 256 -> 198 us incl. the partial reduction)
__host__ __device__ inline int ln_bwd_rows(int M) { (void)M; return 64; }
inline int ln_bwd_parts(int M) { return cdiv(M, ln_bwd_rows(M)); }

// ---- embedding mix (pmgt/pmgt/modeling_pmgt.py:199-208) ----------------------------------------
struct EmbedMix {
    int M = 0, S = 0, d = 0;
    int nf = 2;                    // modalities (1 .. 4): len(feat_hidden_sizes)
    const void* E = nullptr;       // [M, nf d]: e_0 | e_1 | ... (bias already added by the GEMM epilogue)
    const int64_t* e_rows = nullptr;   // optional: token m reads row e_rows[m] of E (E = projection of the whole table)
    // Table mode splits the work (the modality mix depends on the NODE only):
    //   phase 1 (rows = nodes):  fwd: a[n], F[n] = sum_k a_k e_k -> `pre` used as F_all [rows, d];
    //                            bwd: df read from `dF` (= segment sums per node), writes dE [rows, nf d] + dWa / dba partials
    //   phase 2 (rows = tokens): fwd: x = F_all[e_rows[m]] (read through `E`, row stride d) + pos + role -> LN -> dropout;
    //                            bwd: LayerNorm backward only: writes dF, dgamma / dbeta partials
    int phase = 0;
    bool dF_f32 = false;           // phase 1 backward: `dF` holds fp32 segment sums
    const float* Wa = nullptr;     // [nf, nf d]
    const float* ba = nullptr;     // [nf]
    const float* pos = nullptr;    // [max_pos, d]
    const float* role = nullptr;   // [2, d]
    const float* gamma = nullptr;
    const float* beta = nullptr;
    float eps = 1e-12f;
    float* a = nullptr;            // [M, nf] modality weights (saved for backward)
    void* pre = nullptr;           // [M, d] LayerNorm input (saved)
    float* stats = nullptr;        // [M, 2]
    void* h0 = nullptr;            // [M, d] output
    void* q8 = nullptr;            // fp8 mode: h0 additionally as per-row e4m3 (fp8.h contract) [M, d] bytes ...
    float* q8_scale = nullptr;     // ... and its scales [M], for the first layer's fp8 Q|K|V|C projection
    DropCfg drop = {nullptr, 0.f, 0};
    // backward only
    const void* dh0 = nullptr;     // [M, d]
    void* dE = nullptr;            // [M, nf d]
    void* dF = nullptr;            // [M, d] gradient wrt (mix + pos + role), feeds the pos/role sums
    float* part = nullptr;         // [embed_bwd_parts(M)][embed_part_elems(d, nf)]: dgamma | dbeta | dWa (nf x nf d) | dba (padded to 4)
};
// floats per workgroup partial of the embedding backward = the parameter range LayerNorm.weight .. attention.1.bias of the flat layout
__host__ __device__ inline int embed_part_elems(int d, int nf) { return (2 + nf * nf) * d + 4; }
template <typename T> int embed_mix_fwd(const EmbedMix& e, hipStream_t st);
template <typename T> int embed_mix_bwd(const EmbedMix& e, hipStream_t st);
inline int embed_bwd_parts(int M) { return cdiv(M, ln_bwd_rows(M)); }
// dpos[s,:] = colsum over sequences (given as possum [S*d]); drole[0] = possum[0], drole[1] = sum_{s>=1}
int pos_role_finish(const float* possum, int S, int d, int max_pos, float* dpos, float* drole, bool accumulate,
                    hipStream_t st);

}  // namespace pmgt
