// MFMA GEMM kernels of the PMGT engine (declarations).  See gemm.hip for the tiling.
#pragma once
#include "common.h"

namespace pmgt {

enum GemmEpi { EPI_NONE = 0, EPI_GELU = 1, EPI_GELU_GRAD = 2 };

// C[M,N] = epilogue( A[M,K] * B[N,K]^T ).  Both operands have the reduction index contiguous
// ("NT"): forward linears use B = W[out,in]; data-gradients use B = W^T[in,out] copies.
struct GemmNT {
    const void* A = nullptr; int64_t lda = 0;
    const int64_t* a_rows = nullptr;   // optional row gather: logical row m reads A + a_rows[m]*lda
    const void* B = nullptr; int64_t ldb = 0;
    void* C = nullptr; int64_t ldc = 0;
    int M = 0, N = 0, K = 0;
    const float* bias = nullptr;       // [N] fp32, added first
    int epi = EPI_NONE;
    void* aux = nullptr; int64_t ldaux = 0;   // EPI_GELU: pre-activation out; EPI_GELU_GRAD: pre-activation in
    DropCfg drop = {nullptr, 0.f, 0};  // dropout on (acc+bias [+act]), element index m*N+n
    const void* res = nullptr; int64_t ldr = 0;   // residual added last (same dtype as C)
    bool res_gather = false;           // residual row = a_rows[m] (compacted-row GEMMs)
    const int* m_dev = nullptr;        // optional device-side row count (<= M)
    // LayerNorm-BACKWARD epilogue (gemm_wsr_lnb: K = N = 256 streaming form; gemm_nt_lnb: the 256 x 256 tile at N = 256): the GEMM
    // result (+ residual) is the gradient dy of a LayerNorm whose OUTPUT y the forward kept (lnb_gamma / lnb_beta: its parameters;
    // lnb_stats: its {mean, rstd} rows; x^ = (y - beta) / gamma).  C receives dx = LN'(dy), lnb_dx_drop (optional) dx times the
    // dropout mask lnb_drop of the dense layer in front of that LayerNorm, lnb_part one [3][N] partial per workgroup:
    // dgamma | dbeta | column sums of the bf16-rounded dx_drop (or dx).  dy never reaches HBM.
    const void* lnb_y = nullptr; int64_t lnb_ldy = 0;
    const float* lnb_stats = nullptr;
    const float* lnb_gamma = nullptr;
    const float* lnb_beta = nullptr;
    void* lnb_dx_drop = nullptr; int64_t lnb_lddx = 0;
    DropCfg lnb_drop = {nullptr, 0.f, 0};
    float* lnb_part = nullptr;
    // gemm_nt_lnb only: the residual lives in COMPACT row order -- row m of the output adds row lnb_res_inv[m] of `res`, or nothing
    // where lnb_res_inv[m] < 0 (last-layer shortcut: the residual branch exists on the rows the loss read only)
    const int* lnb_res_inv = nullptr;
    // LayerNorm-FORWARD epilogue (gemm_nt_lnf: the 256 x 256 tile at N = 256, K > 512 -- BertOutput at intermediate sizes above 512):
    // C = bf16(dropout(A B^T + bias) + res) as everywhere, lnf_out = LayerNorm(C) with lnf_gamma / lnf_beta / lnf_eps, lnf_stats =
    // {mean, rstd} per row; lnf_skip_c: C itself is not stored (the backward takes x^ from lnf_out).  Filled by gemm_nt_lnf from GemmWS::ln_*.
    void* lnf_out = nullptr;
    float* lnf_stats = nullptr;
    const float* lnf_gamma = nullptr;
    const float* lnf_beta = nullptr;
    float lnf_eps = 1e-12f;
    bool lnf_skip_c = false;
    // 256 x 256 LDS-DMA tile only (dX = dQKVC W at beta == 1, where dQ = dK = 0 and their columns of A were never written): A and B are
    // head-major Q|K|V|C-shaped along k ([head][q, k, v, c][32]); K counts the V | C elements (2 d) and logical k-step kt reads block
    // 2 + (kt & 1) of head kt >> 1.  Every other kernel refuses it.
    int kmap_vc = 0;
    uint32_t opts = 0;                 // PathOpt bits of the calling engine (OPT_TILE_GEMM: register-staged 128 x 128 tile only)
};
template <typename T> int gemm_nt(const GemmNT& g, hipStream_t st);
bool gemm_nt_big_applies(const GemmNT& g);      // bf16: whether gemm_nt takes the 256-row LDS-DMA tiles for this problem (kmap_vc: the 256-wide one)
// the 256 x 256 LDS-DMA tile with the LayerNorm-backward phase behind its main loop (lnb_* fields; bf16, N = 256, K % 64 == 0);
// partials: [gemm_nt_lnb_parts(M)][3][256]
bool gemm_nt_lnb_ok(const GemmNT& g);
int gemm_nt_lnb_parts(int M);
int gemm_nt_lnb(const GemmNT& g, hipStream_t st);

// Weight-stationary streaming variant for K <= 256 in bf16 (gemm_ws.hip); optional fused LayerNorm of the
// output row when N == 256: ln_out = LN(C) with C the (bf16-rounded) epilogue result, stats = {mean, rstd}.
struct GemmWS : GemmNT {
    void* ln_out = nullptr;
    float* ln_stats = nullptr;
    const float* ln_gamma = nullptr;
    const float* ln_beta = nullptr;
    float ln_eps = 1e-12f;
    // fp8 mode, fused-LayerNorm form only: the LayerNorm output additionally leaves as per-row e4m3 (fp8.h contract) for the
    // next layer's fp8 Q|K|V|C projection: q8 [M, N] bytes, q8_scale [M]
    void* q8 = nullptr;
    float* q8_scale = nullptr;
    // fused-LayerNorm form only: do not store C, the pre-LayerNorm sum (the LayerNorm backward then takes x^ from the OUTPUT: rowops.h)
    bool skip_c = false;
};
bool gemm_ws_supported(const GemmWS& g);
bool gemm_ws_fuses_ln(const GemmWS& g);     // false: the caller runs LayerNorm as its own launch
int gemm_ws(const GemmWS& g, hipStream_t st);
// K > 512, N = 256 (FFN2 forward at intermediate sizes above 512): the 256 x 256 LDS-DMA tile owns whole rows, so the residual + LayerNorm
// epilogue runs on the tile behind the main loop (gemm.hip) instead of tile GEMM -> C -> standalone LayerNorm launch; linear() dispatches
// to it.  gemm_nt_lnf_shape: the part of the predicate that depends on the problem shape only (the engine derives "the LayerNorm
// input is not stored" from it, forward and backward alike).
bool gemm_nt_lnf_shape(int M, int N, int K);
bool gemm_nt_lnf_ok(const GemmWS& g);
int gemm_nt_lnf(const GemmWS& g, hipStream_t st);
// 16-wave role-split form of the residual + LayerNorm mode at K = N = 256 (gemm_wsr.hip); gemm_ws dispatches to it
bool gemm_wsr_ok(const GemmWS& g);
int gemm_wsr(const GemmWS& g, hipStream_t st);
// the same GEMM role with the LayerNorm-backward epilogue (lnb_* fields); partials: [gemm_wsr_lnb_parts(M)][3][256]
bool gemm_wsr_lnb_ok(const GemmWS& g);
int gemm_wsr_lnb_parts(int M);
int gemm_wsr_lnb(const GemmWS& g, hipStream_t st);
// 12-wave role-split form of the plain / GELU / GELU' / residual modes at K = 512 (gemm_wsr.hip); gemm_ws dispatches to it
bool gemm_wsr512_ok(const GemmWS& g);
int gemm_wsr512(const GemmWS& g, hipStream_t st);

// Full-row tile at N = 512 (gemm_rowln.hip): C = dropout(A W^T + bias) + res and ln_out = LayerNorm(C) in one launch (tile 128 x 512, LDS-DMA
// ring; bf16; C is always stored); linear() dispatches to it before gemm_ws.
bool gemm_rowln_ok(const GemmWS& g);
int gemm_rowln(const GemmWS& g, hipStream_t st);

// Partial weight gradients: slab[s][N1,N2] = sum over the s-th chunk of rows m of P[m,n1]*Q[m,n2]
// ("TN": the reduction index is the row).  `slab_reduce` then sums the slabs in a fixed order.
struct GemmTN {
    const void* P = nullptr; int64_t ldp = 0;   // [M,N1]  (dY)
    const void* Q = nullptr; int64_t ldq = 0;   // [M,N2]  (X)
    const int64_t* q_rows = nullptr;            // optional row gather on Q
    int M = 0, N1 = 0, N2 = 0;
    float* slab = nullptr;                      // [splits][N1][N2] fp32 workspace
    float* bias_slab = nullptr;                 // optional [splits][N1]: column sums of P (bias gradient)
    int splits = 1;
    const int* m_dev = nullptr;
    const void* zeros = nullptr;                // >= 16 B of device zeros: enables the LDS-DMA kernel (bf16, no gather)
    // P's columns are in head-major order (head, matrix, w) of a [4 x perm_d] block structure with head size perm_dh:
    // output row n1 is written to row (matrix * perm_d + head * perm_dh + w) of the slab.  0 = identity.
    int perm_d = 0, perm_dh = 0;
    bool q_f8 = false;                          // fp8 mode: Q is an e4m3 feature table (ldq in bytes), value = byte * q_scale
    float q_scale = 1.f;
    uint32_t opts = 0;                          // PathOpt bits (OPT_TILE_GEMM: register-staged kernel only)
};
template <typename T> int gemm_tn(const GemmTN& g, hipStream_t st);
int gemm_tn_pick_splits(int M, int N1, int N2, int bkm, uint32_t opts = 0);
template <typename T> int gemm_tn_bkm();

// dst[i] (+)= sum_s slab[s*n + i]   (n % 4 == 0; the slab is used as scratch and clobbered)
int slab_reduce(const float* slab, int splits, int64_t n, float* dst, bool accumulate, hipStream_t st);
// the same for several (slab, destination) pairs in one launch (two when a job has more than 128 slabs); see gemm.hip
struct ReduceJob { const float* src; float* dst; int64_t n; int rows; bool acc; };
int multi_reduce(const ReduceJob* jobs, int njobs, hipStream_t st);

// Column sums of Y[M,N] (bias gradients): dst[n] (+)= sum_m Y[m,n]; needs slab of colsum_slab_elems(M, N) floats.
template <typename T>
int colsum(const T* Y, int64_t ldy, int M, int N, float* slab, float* dst, bool accumulate,
           const int* m_dev, hipStream_t st);
// rows per block: 96 keeps the position-gradient sums of the bench shape (12 288 rows x 8 192 columns) at 128 row blocks (one
// slab_reduce level) x 8 column blocks = 1 024 workgroups; with 256 rows per block that launch had 384 workgroups and one
// load in flight per lane (2.4 TB/s)
constexpr int COLSUM_ROWS = 96;
inline int64_t colsum_slab_elems(int M, int N) { return (int64_t)cdiv(M, COLSUM_ROWS) * N; }

}  // namespace pmgt
