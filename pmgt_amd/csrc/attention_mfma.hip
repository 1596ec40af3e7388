// bf16 MFMA version of the PMGT dual-softmax attention (forward and backward), S <= 64, head size 32/64.
//
// One wave owns one (sequence, head); a workgroup is 4 independent waves.  Everything S x S stays in
// registers / wave-private LDS.  All score-shaped matrices are held TRANSPOSED in the MFMA C/D layout
// ("key j on (lane>>4, reg), query i on lane&15"): X^T[j][i] = sum_c K[j][c] Q[i][c] is an NT product
// whose fragments load straight from global rows, the softmax over j is a register + 2-shuffle
// reduction, and an accumulator tile is directly the B operand of every product that sums over its
// ROW index j (O^T = V^T P^T, dQ^T = K^T dS2^T, first half of dC) — with the k-slot permutation
// j(q, e) = 32 ks + 16 (e >> 2) + 4 q + (e & 3) applied to the A operand, which is fetched with
// ds_read_b64_tr_b16 from a row-major LDS tile.  Products that sum over the COLUMN index i (dV, dK,
// second half of dC) read the bf16 image of the matrix back from LDS row-wise (one transpose through
// LDS, as cdna_hip_programming.md section 3 prescribes).
#include <stdlib.h>

#include "attention.h"

namespace pmgt {

#ifdef PMGT_AB_PROF
// cycles (s_memtime) per phase of the wave backward kernel, summed over the waves of workgroups 0 and 777
__device__ unsigned int g_ab_prof[2][4][10];
#define AB_STAMP(k)                                                   \
    do {                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        pacc[k] = (unsigned int)(now_ - plast);                       \
        plast = now_;                                                 \
    } while (0)
#else
#define AB_STAMP(k) do { } while (0)
#endif

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// Row `row` of a [S][ld] matrix, 8 elements at column `col`; rows >= S read as zero.  The load itself is
// UNCONDITIONAL on a clamped row (a branch around a load makes the compiler wait for it at the join, which
// turns N independent loads into N serial memory round trips); the select happens on the loaded value.
__device__ __forceinline__ bf16x8 ld_rows(const bf16* base, int64_t ld, int row, int S, int col) {
    const int rc = max(min(row, S - 1), 0);
    const bf16x8 v = *(const bf16x8*)(base + (int64_t)rc * ld + col);
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return row < S ? v : z;
}
// FULL = every row exists (S is a compile-time multiple of 16): no clamp, no select.  A select on a loaded value is not free even
// when it folds to "keep": under register pressure the compiler places it -- and the s_waitcnt vmcnt it needs -- BEFORE the loads
// that follow in program order, which put a whole extra memory round trip at the start of the S = 64 backward kernel.
template <bool FULL> __device__ __forceinline__ bf16x8 ld_rows_f(const bf16* base, int64_t ld, int row, int S, int col) {
    if constexpr (FULL) return *(const bf16x8*)(base + (int64_t)row * ld + col);
    else return ld_rows(base, ld, row, S, col);
}
// A operand (16 "c" rows x 32 k) of a product whose k index is a ROW of the row-major LDS tile
// `tile` ([rows][DH] bf16): element e of lane (r, q) is tile[krow(q, e)][c0 + r].
// PERM = true: krow = k0 + 16 (e >> 2) + 4 q + (e & 3)  (matches an accumulator tile used as B);
// PERM = false: krow = k0 + 8 q + e                      (natural order, matches row-read B fragments).
template <int DH, bool PERM>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int k0, int c0, int r, int q) {
    const int row_lo = PERM ? (k0 + 4 * q + (r >> 2)) : (k0 + 8 * q + (r >> 2));
    const int row_hi = PERM ? (row_lo + 16) : (row_lo + 4);
    const int colb = (c0 + 4 * (r & 3)) * 2;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + row_lo * (DH * 2) + colb));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + row_hi * (DH * 2) + colb));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Bank-conflict-free forms for 128-byte rows (head size 64 tiles, S = 64 images).  A 256-byte LDS bank row holds TWO such rows, so
// rows of equal parity meet on the same banks: the plain layout makes the transposing reads below 4-way conflicted (a 32-lane half
// touches 8 rows x 32 bytes) and the row reads of an image 4-way (measured at S = 64 / head size 64: 77 % of the backward kernel's LDS
// cycles were conflict cycles, the LDS 59 % busy).  Tiles: the 16-byte chunk index is XORed with 2 * tkey(row) -- tkey takes four
// different values on every set of equal-parity rows one transposing read touches ({0,2,8,10}+k, {0,2,4,6}+k, {4,6,12,14}+k), and
// bit 0 stays clear because a lane group reads the chunk PAIR (c, c + 1) of a row.  Images: chunk ^ ((row >> 1) & 7), the eight
// equal-parity rows a 16-lane group of ds_read_b128 touches get eight different chunks.
__device__ __forceinline__ int tkey(int row) { return ((row >> 1) & 1) | ((((row >> 2) ^ (row >> 3)) & 1) << 1); }
template <int PITCH, bool SWZ> __device__ __forceinline__ int tile_off(int row, int colb) {
    if constexpr (SWZ && PITCH == 128) return row * 128 + ((((colb >> 4) ^ (tkey(row) << 1)) & 7) << 4) + (colb & 15);
    else return row * PITCH + colb;
}
template <int PITCH, bool SWZ> __device__ __forceinline__ int image_off(int row, int colb) {
    if constexpr (SWZ && PITCH == 128) return row * 128 + ((((colb >> 4) ^ (row >> 1)) & 7) << 4) + (colb & 15);
    else return row * PITCH + colb;
}
// tr_frag on a tile written through tile_off<DH * 2, true>
template <int DH, bool PERM>
__device__ __forceinline__ bf16x8 tr_frag_swz(const char* tile, int k0, int c0, int r, int q) {
    const int row_lo = PERM ? (k0 + 4 * q + (r >> 2)) : (k0 + 8 * q + (r >> 2));
    const int row_hi = PERM ? (row_lo + 16) : (row_lo + 4);
    const int colb = (c0 + 4 * (r & 3)) * 2;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + tile_off<DH * 2, true>(row_lo, colb)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + tile_off<DH * 2, true>(row_hi, colb)));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Combine the 4 lanes (q = 0..3) that share a query column: lanes l, l^16, l^32, l^48.  gfx950's
// v_permlane16_swap / v_permlane32_swap do the two exchanges in the VALU (vdst = src = v: afterwards the two results
// hold the even-row / odd-row, resp. lower-half / upper-half, copies) instead of two ds_bpermute round trips
// through the LDS crossbar -- these reductions sit on the dependent chain max -> exp -> sum -> reciprocal.
// (inline asm on two copies of v: the builtins fold when both operands are the same value; "s_nop 1" covers the
// VALU-write -> permlane-read hazard, cdna_hip_programming.md T21)
// (raw_max: common.h)
__device__ __forceinline__ float xchg16(float v, bool is_max) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return is_max ? raw_max(a, b) : a + b;
}
__device__ __forceinline__ float xchg32(float v, bool is_max) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return is_max ? raw_max(a, b) : a + b;
}
template <int NT> __device__ __forceinline__ float red_q(float v, bool is_max) { return xchg32(xchg16(v, is_max), is_max); }

// Recompute both normalised probability matrices (transposed, fp32, in registers).
// fq/fk/fc: NT-form fragments of Q, K, C rows; rho[] holds the INVERSE norms 1/|c_row|, isq = 1/sqrt(dh).  On return a1/a2[jt][it][e] = A[i = 16 it + r][j = 16 jt + 4 q + e]
// (0 where i or j is padding).
template <int NT, int KD>
__device__ __forceinline__ void probs_T(const bf16x8 (&fq)[NT][KD], const bf16x8 (&fk)[NT][KD], const bf16x8 (&fc)[NT][KD],
                                        const float* rho, const float* madd, int S, int r, int q, float isq,
                                        f32x4 (&a1)[NT][NT], f32x4 (&a2)[NT][NT], int ntq = NT) {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            f32x4 x1 = {0.f, 0.f, 0.f, 0.f}, x2 = x1;
            if (it < ntq) {          // (wave-uniform) query tiles >= ntq carry no gradient: their probabilities stay 0
#pragma unroll
                for (int ks = 0; ks < KD; ++ks) {
                    x1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[jt][ks], fc[it][ks], x1, 0, 0, 0);
                    x2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[jt][ks], fq[it][ks], x2, 0, 0, 0);
                }
            }
            a1[jt][it] = x1;
            a2[jt][it] = x2;
        }
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        if (it >= ntq) continue;
        const int i = 16 * it + r;
        const bool iv = i < S;
        const float rho_i = rho[i];
        float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * jt + 4 * q + e;
                const bool ok = iv && j < S;
                const float v1 = ok ? (1.f - a1[jt][it][e] * (rho_i * rho[j]) + (i == j ? 1.f : 0.f) + madd[j]) : -INFINITY;
                const float v2 = ok ? (a2[jt][it][e] * isq + madd[j]) : -INFINITY;
                a1[jt][it][e] = v1;
                a2[jt][it][e] = v2;
                m1 = fmaxf(m1, v1);
                m2 = fmaxf(m2, v2);
            }
        m1 = red_q<NT>(m1, true);
        m2 = red_q<NT>(m2, true);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float e1 = iv ? __expf(a1[jt][it][e] - m1) : 0.f;
                const float e2 = iv ? __expf(a2[jt][it][e] - m2) : 0.f;
                a1[jt][it][e] = e1;
                a2[jt][it][e] = e2;
                s1 += e1;
                s2 += e2;
            }
        s1 = red_q<NT>(s1, false);
        s2 = red_q<NT>(s2, false);
        const float i1 = iv ? __frcp_rn(s1) : 0.f, i2 = iv ? __frcp_rn(s2) : 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            a1[jt][it] *= i1;
            a2[jt][it] *= i2;
        }
    }
}

// pack an accumulator-layout matrix column block `it` into B fragments (k-step ks covers key tiles 2ks, 2ks+1)
// The same two probability matrices with a shorter VALU chain (the backward kernel is VALU-issue-bound: 1350 VALU
// instructions per (sequence, head), 3.5 waves per SIMD):
//   * scores in the log2 domain (log2 e folded into 1/sqrt(dh), rho_i and the mask term `madd`, which the caller
//     has ALREADY scaled by log2 e and shifted by its maximum over the keys), so an exponential is one v_exp_f32;
//   * no row maximum for the cosine branch: -cos + I <= 2 (the constant 1 of "1 - cos + I" drops out of the softmax);
//   * "+ I" enters through the initial accumulator of the C^ C^T MFMA: -|c_i|^2 on the diagonal (|c_i|^2 rho_i^2 = 1);
//   * v_rcp_f32 instead of an IEEE division.
// ssq[tt] = |c_row|^2 and rho_own[tt] = 1/|c_row| of row 16 tt + r (every q lane holds them).
template <int NT, int KD, bool HAS_MASK>
__device__ __forceinline__ void probs_T2(const bf16x8 (&fq)[NT][KD], const bf16x8 (&fk)[NT][KD], const bf16x8 (&fc)[NT][KD],
                                         const float (&ssq)[NT], const float (&rho_own)[NT], const float* rho, const float* madd,
                                         int S, int r, int q, float isq, f32x4 (&a1)[NT][NT], f32x4 (&a2)[NT][NT], int ntq) {
    constexpr float L2E = 1.4426950408889634f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            f32x4 x1 = {0.f, 0.f, 0.f, 0.f}, x2 = x1;
            if (it < ntq) {          // (wave-uniform) query tiles >= ntq carry no gradient: their probabilities stay 0
                if (jt == it) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x1[e] = (4 * q + e == r) ? -ssq[it] : 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < KD; ++ks) {
                    x1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[jt][ks], fc[it][ks], x1, 0, 0, 0);
                    x2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[jt][ks], fq[it][ks], x2, 0, 0, 0);
                }
            }
            a1[jt][it] = x1;
            a2[jt][it] = x2;
        }
    const float isql = isq * L2E;
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        if (it >= ntq) continue;
        const int i = 16 * it + r;
        const bool iv = i < S;
        const float rl = rho_own[it] * L2E;
        float m2 = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const f32x4 rj = *(const f32x4*)(rho + 16 * jt + 4 * q);
            f32x4 mj = {0.f, 0.f, 0.f, 0.f};
            if (HAS_MASK) mj = *(const f32x4*)(madd + 16 * jt + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = iv && 16 * jt + 4 * q + e < S;
                const float v1 = fmaf(-a1[jt][it][e], rl * rj[e], mj[e]);
                const float v2 = fmaf(a2[jt][it][e], isql, mj[e]);
                a1[jt][it][e] = ok ? v1 : -INFINITY;
                a2[jt][it][e] = ok ? v2 : -INFINITY;
                m2 = raw_max(m2, a2[jt][it][e]);
            }
        }
        m2 = red_q<NT>(m2, true);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float e1 = __builtin_amdgcn_exp2f(a1[jt][it][e]);
                const float e2 = iv ? __builtin_amdgcn_exp2f(a2[jt][it][e] - m2) : 0.f;
                a1[jt][it][e] = e1;
                a2[jt][it][e] = e2;
                s1 += e1;
                s2 += e2;
            }
        s1 = red_q<NT>(s1, false);
        s2 = red_q<NT>(s2, false);
        const float i1 = iv ? __builtin_amdgcn_rcpf(s1) : 0.f, i2 = iv ? __builtin_amdgcn_rcpf(s2) : 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            a1[jt][it] *= i1;
            a2[jt][it] *= i2;
        }
    }
}

// A operand / B operand fragment of an [i][j] bf16 image (row stride LDB bytes) whose k index runs down the ROWS:
// element e of lane (r, q) = img[k0 + 8 q + e][c0 + r]  (ds_read_b64_tr_b16, as tr_frag<., false> does for the tiles)
// Byte offset of the 8-byte chunk at byte column `colb` of image row `row`.  64-byte rows (S = 32): sixteen lanes of a ds_write_b64 group hold
// sixteen consecutive rows of ONE chunk column -- 64-byte strides, i.e. two bank pairs for all of them (8-way: SQ_LDS_BANK_CONFLICT / IDX_ACTIVE
// = 0.75 for this kernel); the chunk index is XORed with (row >> 1) & 7, which spreads rows of equal parity over the eight chunk positions.
template <int LDB> __device__ __forceinline__ int img_off(int row, int colb) {
    if constexpr (LDB == 64) return row * 64 + ((((colb >> 3) ^ (row >> 1)) & 7) << 3) + (colb & 7);
    else return row * LDB + colb;
}
template <int LDB>
__device__ __forceinline__ bf16x8 img_frag(const char* img, int k0, int c0, int r, int q) {
    const int row_lo = k0 + 8 * q + (r >> 2), row_hi = row_lo + 4;
    const int colb = (c0 + 4 * (r & 3)) * 2;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + img_off<LDB>(row_lo, colb)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + img_off<LDB>(row_hi, colb)));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x4 pack4(const f32x4& v) { return (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}; }

template <int NT>
__device__ __forceinline__ bf16x8 pack_b(const f32x4 (&x)[NT][NT], int it, int ks) {
    bf16x8 b;
    const f32x4 lo = x[2 * ks][it];
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = (bf16)lo[e];
    if (2 * ks + 1 < NT) {
        const f32x4 hi = x[(2 * ks + 1 < NT) ? 2 * ks + 1 : 0][it];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[4 + e] = (bf16)hi[e];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) b[4 + e] = (bf16)0.f;
    }
    return b;
}

// Cooperative (one wave) load of a [rows][DH] bf16 tile from global rows (stride ld elements) into LDS,
// rows >= S zero-filled; optional per-row scale (used to normalise C rows).
template <int DH>
__device__ __forceinline__ void load_tile(char* tile, const bf16* src, int64_t ld, int rows, int S, int lane,
                                          const float* row_scale) {
    constexpr int CPR = DH / 8;
    for (int idx = lane; idx < rows * CPR; idx += 64) {
        const int s = idx / CPR, c = idx % CPR;
        bf16x8 v = ld_rows(src, ld, s, S, c * 8);
        if (row_scale) {
            const float sc = row_scale[s];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] * sc);
        }
        *(bf16x8*)(tile + s * (DH * 2) + c * 16) = v;
    }
}

// Split form of load_tile: issue all global loads of a tile into registers first (so every load of the kernel
// is in flight at once: ONE memory round trip per wave instead of one per tile), commit to LDS later.
template <int DH, int ROWS> struct TileRegs {
    static constexpr int CPR = DH / 8, N = ROWS * CPR / 64;
    static_assert(N * 64 == ROWS * CPR, "tile must be a whole number of chunks per lane");
    bf16x8 v[N];
    __device__ __forceinline__ void issue(const bf16* src, int64_t ld, int S, int lane) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = lane + 64 * i, s = idx / CPR, c = idx % CPR;
            v[i] = ld_rows(src, ld, s, S, c * 8);
        }
    }
    __device__ __forceinline__ void commit(char* tile, int lane, const float* row_scale) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = lane + 64 * i, s = idx / CPR, c = idx % CPR;
            bf16x8 x = v[i];
            if (row_scale) {
                const float sc = row_scale[s];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (bf16)((float)x[e] * sc);
            }
            *(bf16x8*)(tile + s * (DH * 2) + c * 16) = x;
        }
    }
};

template <int DH, int NT> struct FwdSmem {
    static constexpr int SP = NT * 16, SP2 = (SP + 31) / 32 * 32;
    static constexpr int TILE = SP2 * DH * 2;
    static constexpr int BYTES = TILE + 2 * 64 * 4;
};

template <int DH, int NT>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(AttnArgs a) {
    using SM = FwdSmem<DH, NT>;
    constexpr int KD = DH / 32, CT = DH / 16, KS = SM::SP2 / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int gidx = blockIdx.x * 4 + wave;
    const int S = a.S, H = a.H, d = H * DH;
    const bool act = gidx < a.Tseq * H;
    const int t = act ? gidx / H : 0, h = act ? gidx % H : 0;
    char* base = smem + wave * SM::BYTES;
    char* tV = base;
    float* rho = (float*)(base + SM::TILE);
    float* madd = rho + 64;
    const bf16* X = (const bf16*)a.qkvc + (int64_t)t * S * 4 * d + h * DH;
    const int64_t ld = 4 * d;
    const int Sv = act ? S : 0;

    TileRegs<DH, SM::SP2> rV;
    rV.issue(X + 2 * d, ld, Sv, lane);
    bf16x8 fq[NT][KD], fk[NT][KD], fc[NT][KD];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const int row = 16 * tt + r;
        const bool ok = row < Sv;
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            fq[tt][ks] = ld_rows(X, ld, row, Sv, 32 * ks + 8 * q);
            fk[tt][ks] = ld_rows(X + d, ld, row, Sv, 32 * ks + 8 * q);
            fc[tt][ks] = ld_rows(X + 3 * d, ld, row, Sv, 32 * ks + 8 * q);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float c = (float)fc[tt][ks][e]; ss = fmaf(c, c, ss); }
        }
        ss = red_q<NT>(ss, false);
        if (q == 0) rho[row] = ok ? rsqrtf(ss) : 0.f;        // 1 / |c_row|
    }
    madd[lane] = (lane < Sv && a.mask) ? (1.f - a.mask[(int64_t)t * S + lane]) * -10000.f : 0.f;
    rV.commit(tV, lane, nullptr);
    __syncthreads();

    f32x4 a1[NT][NT], a2[NT][NT];
    probs_T<NT, KD>(fq, fk, fc, rho, madd, Sv, r, q, rsqrtf((float)DH), a1, a2);

    // mix + dropout -> P^T in a1
    const DropKey k1 = make_drop_key(a.drop1), k2 = make_drop_key(a.drop2);
    const float beta = a.beta, omb = 1.f - a.beta;
    const uint64_t hbase = ((uint64_t)t * H + h) * S;
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        const int i = 16 * it + r;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            float d1[4] = {1.f, 1.f, 1.f, 1.f}, d2[4] = {1.f, 1.f, 1.f, 1.f};
            if (k1.on) {
                drop_mul4(k1, (uint32_t)(hbase + i), (uint32_t)(4 * jt + q), d1);
                drop_mul4(k2, (uint32_t)(hbase + i), (uint32_t)(4 * jt + q), d2);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * jt + 4 * q + e;
                const float p = beta * d1[e] * a1[jt][it][e] + omb * d2[e] * a2[jt][it][e];
                a1[jt][it][e] = p;
                if (a.probs && i < Sv && j < Sv) a.probs[(hbase + i) * S + j] = p;
            }
        }
    }
    // O^T[c][i] = sum_j V[j][c] P[i][j]
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        bf16x8 pb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) pb[ks] = pack_b<NT>(a1, it, ks);
        const int i = 16 * it + r;
        f32x4 o[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                o[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH, true>(tV, 32 * ks, 16 * ct, r, q), pb[ks], o[ct], 0, 0, 0);
        }
#pragma unroll
        for (int cp = 0; cp < CT / 2; ++cp)
            store_row32((bf16*)a.ctx + ((int64_t)t * S + min(i, S - 1)) * d + h * DH + 32 * cp, o[2 * cp], o[2 * cp + 1], q, i < Sv);
    }
}

template <int DH, int NT> struct BwdSmem {
    static constexpr int SP = NT * 16, SP2 = (SP + 31) / 32 * 32;
    static constexpr int TILE = SP2 * DH * 2;         // Q, K, dO, C-hat
    static constexpr int IMG = SP * SP2 * 2;          // P^T, dS1^T, dS2^T as [j][i] bf16
    static constexpr int BYTES = 4 * TILE + 3 * IMG + 2 * 64 * 4;
};

// FULL: S == 16 NT (no padded rows or keys): the sequence length is a compile-time constant, every bounds select folds
// away (the kernel is VALU-issue-bound: ~1600 VALU instructions per (sequence, head)); idle waves of the last
// workgroup recompute pair 0 and only their stores are predicated off.
// One-wave form: the K and C-hat tiles are dead once dQ / dC are done, so (when an image is no larger than a tile) the
// P^T and dS2^T images, first needed by the dV / dK products that follow, are written INTO them then: 10.5 instead of
// 14.5 KiB per wave at S = 32, dh = 32 -> 14 instead of 10 waves per CU (8 -> 10 waves measured -6 %).
template <int DH, int NT> struct BwdSmemW : BwdSmem<DH, NT> {
    using B = BwdSmem<DH, NT>;
    static constexpr bool ALIAS = B::IMG <= B::TILE;
    static constexpr int BYTES = 4 * B::TILE + (ALIAS ? 1 : 3) * B::IMG + 2 * 64 * 4;
};

template <int DH, int NT, bool FULL>
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(AttnArgs a) {
    using SM = BwdSmemW<DH, NT>;
    constexpr int KD = DH / 32, CT = DH / 16, KS = SM::SP2 / 32, SP = SM::SP, SP2 = SM::SP2;
    constexpr int LDI = SP * 2;      // the three images are [i (query, SP2 rows, zero beyond SP)][j (key, SP)] bf16: a lane's four
                                     // consecutive keys leave as ONE 8-byte write; the products read them with transpose reads
    // every wave works in its own LDS region: the LDS pipeline executes one wave's operations in order, so a compiler
    // barrier (no s_barrier: the waves of a workgroup are independent and must not march in lockstep) orders them
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nw = blockDim.x >> 6;
    // wave index as a KNOWN wave-uniform value: every `if` on t / ntq below becomes a scalar branch (a condition derived from
    // threadIdx.x is divergent to the compiler, which then masks EXEC around the block -- and MFMA ignores EXEC)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int gidx = blockIdx.x * nw + wave;
    const int S = FULL ? NT * 16 : a.S, H = a.H, d = H * DH;
    const bool act = gidx < a.Tseq * H;
    const int t = act ? gidx / H : 0, h = act ? gidx % H : 0;
    char* base = smem + wave * SM::BYTES;
    char* tQ = base;
    char* tK = tQ + SM::TILE;
    char* tO = tK + SM::TILE;
    char* tC = tO + SM::TILE;
    char* iS1 = tC + SM::TILE;
    char* iP = SM::ALIAS ? tK : iS1 + SM::IMG;
    char* iS2 = SM::ALIAS ? tC : iS1 + 2 * SM::IMG;
    float* rho = (float*)(iS1 + (SM::ALIAS ? 1 : 3) * SM::IMG);
    float* madd = rho + 64;
    // matrix m of head h starts at column h * DH + m * d (q | k | v | c blocks) or, head-major, at (4 h + m) * DH
    const int hoff = a.hm ? 4 * h * DH : h * DH, ms = a.hm ? DH : d;
    const bf16* X = (const bf16*)a.qkvc + (int64_t)t * S * 4 * d + hoff;
    const bf16* DO = (const bf16*)a.dctx + (int64_t)t * S * d + h * DH;
    bf16* DX = (bf16*)a.dqkvc + (int64_t)t * S * 4 * d + hoff;
    const int64_t ld = 4 * d;
    const int Sv = FULL ? NT * 16 : (act ? S : 0);
    const int ntq = t < a.cls_only_seqs ? 1 : NT;      // query tiles that carry a gradient (wave-uniform)
    const float isq = rsqrtf((float)DH);

#ifdef PMGT_AB_PROF
    unsigned int pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long plast = __builtin_readcyclecounter();
#endif
    // NT <= 2: every global load of the kernel is issued here, before anything is consumed
    constexpr bool PRE = NT <= 2;
    // (the row-major LDS tiles of Q, K, dO, C-hat are written from the FRAGMENT registers: lane (r, q) of fragment
    // (tt, ks) holds exactly the 16 bytes of tile row 16 tt + r at byte 64 ks + 16 q -- no second set of global loads)
    bf16x8 fv[NT][KD], fo[NT][KD];
    if constexpr (PRE) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int row = 16 * tt + r;
#pragma unroll
            for (int ks = 0; ks < KD; ++ks) {
                fv[tt][ks] = ld_rows(X + 2 * ms, ld, row, Sv, 32 * ks + 8 * q);
                fo[tt][ks] = ld_rows(DO, d, row, Sv, 32 * ks + 8 * q);
            }
        }
    } else {
        load_tile<DH>(tQ, X, ld, SP2, Sv, lane, nullptr);
        load_tile<DH>(tK, X + ms, ld, SP2, Sv, lane, nullptr);
        load_tile<DH>(tO, DO, d, SP2, Sv, lane, nullptr);
    }
    f32x4 a1[NT][NT], a2[NT][NT];
    bf16x8 fq[NT][KD], fk[NT][KD], fc[NT][KD];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            fq[tt][ks] = ld_rows(X, ld, 16 * tt + r, Sv, 32 * ks + 8 * q);
            fk[tt][ks] = ld_rows(X + ms, ld, 16 * tt + r, Sv, 32 * ks + 8 * q);
            fc[tt][ks] = ld_rows(X + 3 * ms, ld, 16 * tt + r, Sv, 32 * ks + 8 * q);
        }
    // dropout keys: AFTER the fragment loads are in flight (their scalar loads of {seed, step} would otherwise hold the
    // vector loads back), long before their first use
    const DropKey k1 = make_drop_key(a.drop1), k2 = make_drop_key(a.drop2);
    {
        float rho_own[NT], ssq[NT];
        rho[lane] = 0.f;      // rows [16 NT, 64) are never written below but scale the (zero) padding rows of the C-hat tile
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int row = 16 * tt + r;
            const bool ok = row < Sv;
            float ss = 0.f;
#pragma unroll
            for (int ks = 0; ks < KD; ++ks) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16x2_t c2 = {fc[tt][ks][2 * e], fc[tt][ks][2 * e + 1]};
                    ss = __builtin_amdgcn_fdot2_f32_bf16(c2, c2, ss, false);
                }
            }
            ss = red_q<NT>(ss, false);
            ssq[tt] = ss;
            rho_own[tt] = ok ? __builtin_amdgcn_rsqf(ss) : 0.f;   // 1 / |c_row| (all four q lanes of the row hold it)
            if (q == 0) rho[row] = rho_own[tt];
        }
        const bool has_mask = a.mask != nullptr;     // (uniform)
        if (has_mask) {     // mask term in the log2 domain, shifted by its maximum over the keys (see probs_T2)
            const float mv = lane < Sv ? (1.f - a.mask[(int64_t)t * S + lane]) * -10000.f : -INFINITY;
            float mm = mv;
            mm = raw_max(mm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mm), 0x121, 0xf, 0xf, false)));
            mm = raw_max(mm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mm), 0x122, 0xf, 0xf, false)));
            mm = raw_max(mm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mm), 0x124, 0xf, 0xf, false)));
            mm = raw_max(mm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mm), 0x128, 0xf, 0xf, false)));
            mm = red_q<NT>(mm, true);
            madd[lane] = lane < Sv ? (mv - mm) * 1.4426950408889634f : 0.f;
        } else {
            madd[lane] = 0.f;
        }
        if constexpr (PRE) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int row = 16 * tt + r;
                const float rr = rho_own[tt];
#pragma unroll
                for (int ks = 0; ks < KD; ++ks) {
                    const int off = row * (DH * 2) + (32 * ks + 8 * q) * 2;
                    *(bf16x8*)(tQ + off) = fq[tt][ks];
                    *(bf16x8*)(tK + off) = fk[tt][ks];
                    *(bf16x8*)(tO + off) = fo[tt][ks];
                    bf16x8 ch;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ch[e] = (bf16)((float)fc[tt][ks][e] * rr);
                    *(bf16x8*)(tC + off) = ch;
                }
            }
            if (SP2 > SP) {        // zero rows [SP, SP2) of the four tiles (k padding of the transposed reads)
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int idx = lane; idx < (SP2 - SP) * (DH / 8); idx += 64) {
                    const int off = (SP + idx / (DH / 8)) * (DH * 2) + (idx % (DH / 8)) * 16;
                    *(bf16x8*)(tQ + off) = z;
                    *(bf16x8*)(tK + off) = z;
                    *(bf16x8*)(tO + off) = z;
                    *(bf16x8*)(tC + off) = z;
                }
            }
        }
        AB_STAMP(0);
        wave_sync();
        AB_STAMP(1);
        probs_T2<NT, KD, true>(fq, fk, fc, ssq, rho_own, rho, madd, Sv, r, q, isq, a1, a2, ntq);
    }
    AB_STAMP(2);
    // C-hat tile (rows scaled by the inverse norms): written above from the fragments when PRE
    if constexpr (!PRE) load_tile<DH>(tC, X + 3 * ms, ld, SP2, Sv, lane, rho);

    // dP^T[j][i] = sum_c V[j][c] dO[i][c]
    f32x4 dp[NT][NT];
    {
        if constexpr (!PRE) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int row = 16 * tt + r;
                const bool ok = row < Sv;
#pragma unroll
                for (int ks = 0; ks < KD; ++ks) {
                    fv[tt][ks] = ld_rows(X + 2 * ms, ld, row, Sv, 32 * ks + 8 * q);
                    fo[tt][ks] = ld_rows(DO, d, row, Sv, 32 * ks + 8 * q);
                }
            }
        }
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int it = 0; it < NT; ++it) {
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (it < ntq) {
#pragma unroll
                    for (int ks = 0; ks < KD; ++ks) x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[jt][ks], fo[it][ks], x, 0, 0, 0);
                }
                dp[jt][it] = x;
            }
    }
    AB_STAMP(3);
    // softmax backward (both branches) in registers; images of P^T, dS1^T, dS2^T to LDS
    f32x4 pmr[SM::ALIAS ? NT : 1][SM::ALIAS ? NT : 1];      // P^T kept in registers until its (aliased) image can be written
    const float cb = a.beta * k1.scale, co = (1.f - a.beta) * k2.scale;     // branch weight x dropout scale
    const uint64_t hbase = ((uint64_t)t * H + h) * S;
    const bf16x4 z4 = {0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        const int i = 16 * it + r;
        if (it >= ntq) {        // no gradient through these queries: zero rows in the three images (a1 / a2 are set to 0)
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                *(bf16x4*)(iS1 + img_off<LDI>(i, (16 * jt + 4 * q) * 2)) = z4;
                a1[jt][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                a2[jt][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (SM::ALIAS) {
                    pmr[jt][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
                } else {
                    *(bf16x4*)(iP + img_off<LDI>(i, (16 * jt + 4 * q) * 2)) = z4;
                    *(bf16x4*)(iS2 + img_off<LDI>(i, (16 * jt + 4 * q) * 2)) = z4;
                }
            }
            continue;
        }
        float rd1 = 0.f, rd2 = 0.f;
        f32x4 g1[NT], g2[NT], pm[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            bool kp1[4] = {true, true, true, true}, kp2[4] = {true, true, true, true};
            if (k1.on) {
                drop_keep4(k1, (uint32_t)(hbase + i), (uint32_t)(4 * jt + q), kp1);
                drop_keep4(k2, (uint32_t)(hbase + i), (uint32_t)(4 * jt + q), kp2);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x1 = kp1[e] ? cb * dp[jt][it][e] : 0.f, x2 = kp2[e] ? co * dp[jt][it][e] : 0.f;
                g1[jt][e] = x1;
                g2[jt][e] = x2;
                pm[jt][e] = fmaf(cb, kp1[e] ? a1[jt][it][e] : 0.f, kp2[e] ? co * a2[jt][it][e] : 0.f);
                rd1 = fmaf(a1[jt][it][e], x1, rd1);
                rd2 = fmaf(a2[jt][it][e], x2, rd2);
            }
        }
        rd1 = red_q<NT>(rd1, false);
        rd2 = red_q<NT>(rd2, false);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a1[jt][it][e] = a1[jt][it][e] * (g1[jt][e] - rd1);     // dS1; a == 0 on padding -> ds == 0
                a2[jt][it][e] = a2[jt][it][e] * (g2[jt][e] - rd2);     // dS2
            }
            *(bf16x4*)(iS1 + img_off<LDI>(i, (16 * jt + 4 * q) * 2)) = pack4(a1[jt][it]);
            if constexpr (SM::ALIAS) {
                pmr[jt][it] = pm[jt];
            } else {
                *(bf16x4*)(iP + img_off<LDI>(i, (16 * jt + 4 * q) * 2)) = pack4(pm[jt]);
                *(bf16x4*)(iS2 + img_off<LDI>(i, (16 * jt + 4 * q) * 2)) = pack4(a2[jt][it]);
            }
        }
    }
    if (SP2 > SP) {     // zero the padding rows i in [SP, SP2) of the images (k padding of the products that sum over i)
        for (int idx = lane; idx < (SP2 - SP) * (SP / 4); idx += 64) {
            const int off = (SP + idx / (SP / 4)) * LDI + (idx % (SP / 4)) * 8;
            *(bf16x4*)(iS1 + off) = z4;
            if constexpr (!SM::ALIAS) {
                *(bf16x4*)(iP + off) = z4;
                *(bf16x4*)(iS2 + off) = z4;
            }
        }
    }
    wave_sync();

    AB_STAMP(4);
    // ---- products with the query / "x" index on the lane
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        bf16x8 b2[KS], b1[KS], bt[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            b2[ks] = pack_b<NT>(a2, it, ks);
            b1[ks] = pack_b<NT>(a1, it, ks);
            // dS1 with the roles swapped: B[k = y][n = x] = dS1[y][x]  (image rows y, column x)
            bt[ks] = img_frag<LDI>(iS1, 32 * ks, 16 * it, r, q);
        }
        const int x = 16 * it + r;
        f32x4 dch[CT], dqv[CT];
        float dt = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f32x4 dq = {0.f, 0.f, 0.f, 0.f}, dc = dq;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (it < ntq) {
                    dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH, true>(tK, 32 * ks, 16 * ct, r, q), b2[ks], dq, 0, 0, 0);
                    dc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH, true>(tC, 32 * ks, 16 * ct, r, q), b1[ks], dc, 0, 0, 0);
                }
                dc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH, false>(tC, 32 * ks, 16 * ct, r, q), bt[ks], dc, 0, 0, 0);
            }
            dqv[ct] = dq * isq;
            dch[ct] = -dc;       // dN = -dS1
            const f32x4 ch = load4<bf16>((const bf16*)(tC + (x * DH + 16 * ct + 4 * q) * 2));
            dt += (dch[ct][0] * ch[0] + dch[ct][1] * ch[1]) + (dch[ct][2] * ch[2] + dch[ct][3] * ch[3]);
        }
        dt = red_q<NT>(dt, false);
        const float inv = rho[x < 64 ? x : 0];      // 1 / |c_x|
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const f32x4 ch = load4<bf16>((const bf16*)(tC + (x * DH + 16 * ct + 4 * q) * 2));
            dch[ct] = (dch[ct] - ch * dt) * inv;
        }
        bf16* rowx = DX + (int64_t)min(x, S - 1) * ld;
#pragma unroll
        for (int cp = 0; cp < CT / 2; ++cp) {
            store_row32(rowx + 32 * cp, dqv[2 * cp], dqv[2 * cp + 1], q, act && x < Sv);
            store_row32(rowx + 3 * ms + 32 * cp, dch[2 * cp], dch[2 * cp + 1], q, act && x < Sv);
        }
    }
    AB_STAMP(5);
    if constexpr (SM::ALIAS) {      // the K and C-hat tiles are dead: they become the P and dS2 images
        wave_sync();
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const int off = img_off<LDI>(16 * it + r, (16 * jt + 4 * q) * 2);
                *(bf16x4*)(iP + off) = pack4(pmr[jt][it]);
                *(bf16x4*)(iS2 + off) = pack4(a2[jt][it]);
            }
        if (SP2 > SP) {
            for (int idx = lane; idx < (SP2 - SP) * (SP / 4); idx += 64) {
                const int off = (SP + idx / (SP / 4)) * LDI + (idx % (SP / 4)) * 8;
                *(bf16x4*)(iP + off) = z4;
                *(bf16x4*)(iS2 + off) = z4;
            }
        }
        wave_sync();
    }
    AB_STAMP(6);
    // ---- products with the key index on the lane: dV^T[c][j] = sum_i dO[i][c] P[i][j], dK^T[c][j] = sum_i Q[i][c] dS2[i][j]
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        bf16x8 bp[KS], bs[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bp[ks] = img_frag<LDI>(iP, 32 * ks, 16 * jt, r, q);        // B[k = i][n = j] = P[i][j]
            bs[ks] = img_frag<LDI>(iS2, 32 * ks, 16 * jt, r, q);
        }
        const int j = 16 * jt + r;
        f32x4 dvv[CT], dkv[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f32x4 dv = {0.f, 0.f, 0.f, 0.f}, dk = dv;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH, false>(tO, 32 * ks, 16 * ct, r, q), bp[ks], dv, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH, false>(tQ, 32 * ks, 16 * ct, r, q), bs[ks], dk, 0, 0, 0);
            }
            dvv[ct] = dv;
            dkv[ct] = dk * isq;
        }
        bf16* rowj = DX + (int64_t)min(j, S - 1) * ld;
#pragma unroll
        for (int cp = 0; cp < CT / 2; ++cp) {
            store_row32(rowj + 2 * ms + 32 * cp, dvv[2 * cp], dvv[2 * cp + 1], q, act && j < Sv);
            store_row32(rowj + ms + 32 * cp, dkv[2 * cp], dkv[2 * cp + 1], q, act && j < Sv);
        }
    }
    AB_STAMP(7);
#ifdef PMGT_AB_PROF
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 777) && wave < 4)
        for (int k_ = 0; k_ < 8; ++k_) g_ab_prof[blockIdx.x == 0 ? 0 : 1][wave][k_] = pacc[k_];
#endif
}

// ------------------------------------------------------------------------------------------------
// Backward, cooperative form: NT waves per (sequence, head); wave `it` owns query tile it in the first half
// (scores, both softmax backward passes, dQ, first half of dC) and key tile it in the second half (dV, dK, second
// half of dC).  The LDS tiles/images of the unit are shared, so LDS and VGPRs per wave drop by ~NT (S = 64,
// dh = 64: 57 KiB + 360 VGPRs per WAVE before, per 4 waves now; S = 32, dh = 32: 20 waves per CU instead of 10).
// Every wave loads the K / C / V fragments of all key tiles itself (L1/L2 hits) and writes only its own 16 rows
// of the Q, K, dO, C-hat tiles, so the first barrier is the only dependency between the waves' loads.
// ------------------------------------------------------------------------------------------------
template <int NT> __device__ __forceinline__ bf16x8 pack_col(const f32x4 (&x)[NT], int ks) {
    bf16x8 b;
    const f32x4 lo = x[2 * ks];
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = (bf16)lo[e];
    if (2 * ks + 1 < NT) {
        const f32x4 hi = x[(2 * ks + 1 < NT) ? 2 * ks + 1 : 0];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[4 + e] = (bf16)hi[e];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) b[4 + e] = (bf16)0.f;
    }
    return b;
}

template <int NT> struct CoopCfg { static constexpr int G = NT == 1 ? 4 : (NT == 2 ? 2 : 1), THREADS = 64 * NT * G; };
// LDS of the cooperative backward.  Four query tiles (S = 64) with an image no larger than a tile (head size 64): the dS2^T image is
// written INTO the K tile once every wave is done with it (one more barrier), 48.5 instead of 56.5 KB per workgroup -> three
// workgroups per CU instead of two (the kernel is latency-bound: its waves wait for global loads and LDS round trips).
template <int DH, int NT> struct BwdSmemC : BwdSmem<DH, NT> {
    using B = BwdSmem<DH, NT>;
    static constexpr bool ALIAS = NT == 4 && B::IMG <= B::TILE;
    static constexpr int BYTES = 4 * B::TILE + (ALIAS ? 2 : 3) * B::IMG + 2 * 64 * 4;
};

#ifdef PMGT_COOP_PROF
// cycles per interval of workgroup PMGT_COOP_PROF (a mid-grid index), per wave: loads issued | landed + tiles written | barrier |
// first half | barrier(s) | second half + stores issued
__device__ unsigned long long g_coop_prof[4][8];
#define COOP_STAMP(k_) do { const unsigned long long n_ = __builtin_readcyclecounter(); if (blockIdx.x == PMGT_COOP_PROF && lane == 0) g_coop_prof[wave][k_] = n_ - plast; plast = n_; } while (0)
#else
#define COOP_STAMP(k_) do { } while (0)
#endif
// FULL: S == 16 NT -- every bounds test is a compile-time constant (only the stores of an idle group are predicated off)
template <int DH, int NT, bool FULL>
__global__ __launch_bounds__(CoopCfg<NT>::THREADS) __attribute__((amdgpu_waves_per_eu(3))) void attn_bwd_coop_kernel(AttnArgs a) {
    using SM = BwdSmemC<DH, NT>;
    constexpr int KD = DH / 32, CT = DH / 16, KS = SM::SP2 / 32, SP = SM::SP, SP2 = SM::SP2, G = CoopCfg<NT>::G;
    constexpr bool TSW = DH == 64, ISW = SP2 == 64;      // 128-byte rows: swizzled layouts (tile_off / image_off)
    constexpr int TP = DH * 2, IP = SP2 * 2;             // row pitches in bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
#ifdef PMGT_COOP_PROF
    unsigned long long plast = __builtin_readcyclecounter();
#endif
    const int ul = wave / NT, it = wave % NT;
    const int gidx = blockIdx.x * G + ul;
    const int S = a.S, H = a.H, d = H * DH;
    const bool act = gidx < a.Tseq * H;
    const int t = act ? gidx / H : 0, h = act ? gidx % H : 0;
    char* base = smem + ul * SM::BYTES;
    char* tQ = base;
    char* tK = tQ + SM::TILE;
    char* tO = tK + SM::TILE;
    char* tC = tO + SM::TILE;
    char* iP = tC + SM::TILE;
    char* iS1 = iP + SM::IMG;
    char* iS2 = SM::ALIAS ? tK : iS1 + SM::IMG;
    float* rho = (float*)(iS1 + (SM::ALIAS ? 1 : 2) * SM::IMG);
    float* madd = rho + 64;
    const bf16* X = (const bf16*)a.qkvc + (int64_t)t * S * 4 * d + h * DH;
    const bf16* DO = (const bf16*)a.dctx + (int64_t)t * S * d + h * DH;
    bf16* DX = (bf16*)a.dqkvc + (int64_t)t * S * 4 * d + h * DH;
    const int64_t ld = 4 * d;
    const int Sv = FULL ? 16 * NT : (act ? S : 0);
    const bool live = FULL ? act : true;                    // FULL: an idle group recomputes pair (0, 0), its stores are predicated off
    const float isq = rsqrtf((float)DH);
    // the mask value of key `lane` first: the memory counter retires in order, so a load issued AFTER the tile loads would make its
    // consumer wait for all of them
    const float mval = a.mask ? a.mask[(int64_t)t * S + min(lane, S - 1)] : 1.f;
    const int x = 16 * it + r;                  // this lane's row (query i in the first half, key j in the second)
    auto trf = [&]<bool PERM>(const char* tile, int k0, int c0) {
        if constexpr (TSW) return tr_frag_swz<DH, PERM>(tile, k0, c0, r, q);
        else return tr_frag<DH, PERM>(tile, k0, c0, r, q);
    };

    // ---- all global loads of the wave, unconditional and clamped
    bf16x8 fq[KD], fo[KD], fko[KD], fco[KD], fk[NT][KD], fc[NT][KD], fv[NT][KD];
#pragma unroll
    for (int ks = 0; ks < KD; ++ks) {
        fq[ks] = ld_rows_f<FULL>(X, ld, x, Sv, 32 * ks + 8 * q);
        fko[ks] = ld_rows_f<FULL>(X + d, ld, x, Sv, 32 * ks + 8 * q);
        fco[ks] = ld_rows_f<FULL>(X + 3 * d, ld, x, Sv, 32 * ks + 8 * q);
        fo[ks] = ld_rows_f<FULL>(DO, d, x, Sv, 32 * ks + 8 * q);
    }
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            fk[jt][ks] = ld_rows_f<FULL>(X + d, ld, 16 * jt + r, Sv, 32 * ks + 8 * q);
            fv[jt][ks] = ld_rows_f<FULL>(X + 2 * d, ld, 16 * jt + r, Sv, 32 * ks + 8 * q);
            fc[jt][ks] = ld_rows_f<FULL>(X + 3 * d, ld, 16 * jt + r, Sv, 32 * ks + 8 * q);
        }
    COOP_STAMP(0);
    // own rows: inverse norm, mask term, and the four tiles
    float rho_x;
    {
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < KD; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float c = (float)fco[ks][e]; ss = fmaf(c, c, ss); }
        ss = red_q<NT>(ss, false);
        rho_x = x < Sv ? rsqrtf(ss) : 0.f;
        if (q == 0) rho[x] = rho_x;
        if ((lane >> 4) == it) madd[lane] = lane < Sv ? (1.f - mval) * -10000.f : 0.f;
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            const int off = tile_off<TP, TSW>(x, (32 * ks + 8 * q) * 2);
            *(bf16x8*)(tQ + off) = fq[ks];
            *(bf16x8*)(tK + off) = fko[ks];
            *(bf16x8*)(tO + off) = fo[ks];
            bf16x8 ch;
#pragma unroll
            for (int e = 0; e < 8; ++e) ch[e] = (bf16)((float)fco[ks][e] * rho_x);
            *(bf16x8*)(tC + off) = ch;
        }
        if (SP2 > SP && it == NT - 1) {        // zero rows [SP, SP2) of the four tiles (k padding of the transposed reads)
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int idx = lane; idx < (SP2 - SP) * (DH / 8); idx += 64) {
                const int off = tile_off<TP, TSW>(SP + idx / (DH / 8), (idx % (DH / 8)) * 16);
                *(bf16x8*)(tQ + off) = z;
                *(bf16x8*)(tK + off) = z;
                *(bf16x8*)(tO + off) = z;
                *(bf16x8*)(tC + off) = z;
            }
        }
    }
    COOP_STAMP(1);
    __syncthreads();
    COOP_STAMP(2);

    // ---- first half: query tile `it`
    f32x4 a1[NT], a2[NT], dp[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        f32x4 x1 = {0.f, 0.f, 0.f, 0.f}, x2 = x1, x3 = x1;
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            x1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[jt][ks], fco[ks], x1, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[jt][ks], fq[ks], x2, 0, 0, 0);
            x3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[jt][ks], fo[ks], x3, 0, 0, 0);
        }
        a1[jt] = x1; a2[jt] = x2; dp[jt] = x3;
    }
    {
        const bool iv = x < Sv;
        float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * jt + 4 * q + e;
                const bool ok = iv && j < Sv;
                const float v1 = ok ? (1.f - a1[jt][e] * (rho_x * rho[j]) + (x == j ? 1.f : 0.f) + madd[j]) : -INFINITY;
                const float v2 = ok ? (a2[jt][e] * isq + madd[j]) : -INFINITY;
                a1[jt][e] = v1;
                a2[jt][e] = v2;
                m1 = fmaxf(m1, v1);
                m2 = fmaxf(m2, v2);
            }
        m1 = red_q<NT>(m1, true);
        m2 = red_q<NT>(m2, true);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float e1 = iv ? __expf(a1[jt][e] - m1) : 0.f;
                const float e2 = iv ? __expf(a2[jt][e] - m2) : 0.f;
                a1[jt][e] = e1;
                a2[jt][e] = e2;
                s1 += e1;
                s2 += e2;
            }
        s1 = red_q<NT>(s1, false);
        s2 = red_q<NT>(s2, false);
        const float i1 = iv ? __frcp_rn(s1) : 0.f, i2 = iv ? __frcp_rn(s2) : 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) { a1[jt] *= i1; a2[jt] *= i2; }
    }
    {   // softmax backward of both branches; images of P^T, dS1^T, dS2^T (columns of this query tile)
        const DropKey k1 = make_drop_key(a.drop1), k2 = make_drop_key(a.drop2);
        const float beta = a.beta, omb = 1.f - a.beta;
        const uint64_t hbase = ((uint64_t)t * H + h) * S;
        float rd1 = 0.f, rd2 = 0.f;
        f32x4 g1[NT], g2[NT], pm[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            float d1[4] = {1.f, 1.f, 1.f, 1.f}, d2[4] = {1.f, 1.f, 1.f, 1.f};
            if (k1.on) {
                drop_mul4(k1, (uint32_t)(hbase + x), (uint32_t)(4 * jt + q), d1);
                drop_mul4(k2, (uint32_t)(hbase + x), (uint32_t)(4 * jt + q), d2);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float m1 = beta * d1[e], m2 = omb * d2[e];
                const float x1 = m1 * dp[jt][e], x2 = m2 * dp[jt][e];
                g1[jt][e] = x1;
                g2[jt][e] = x2;
                pm[jt][e] = m1 * a1[jt][e] + m2 * a2[jt][e];
                rd1 = fmaf(a1[jt][e], x1, rd1);
                rd2 = fmaf(a2[jt][e], x2, rd2);
            }
        }
        rd1 = red_q<NT>(rd1, false);
        rd2 = red_q<NT>(rd2, false);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * jt + 4 * q + e;
                const float ds1 = a1[jt][e] * (g1[jt][e] - rd1);     // a == 0 on padding -> ds == 0
                const float ds2 = a2[jt][e] * (g2[jt][e] - rd2);
                a1[jt][e] = ds1;
                a2[jt][e] = ds2;
                const int io = image_off<IP, ISW>(j, x * 2);
                *(bf16*)(iP + io) = (bf16)pm[jt][e];
                *(bf16*)(iS1 + io) = (bf16)ds1;
                if (!SM::ALIAS) *(bf16*)(iS2 + io) = (bf16)ds2;      // (aliased: after the dQ products below)
            }
        if (SP2 > SP && it == NT - 1) {     // zero the padding columns i in [SP, SP2) of the images
            for (int idx = lane; idx < SP * (SP2 - SP); idx += 64) {
                const int j = idx / (SP2 - SP), i = SP + idx % (SP2 - SP);
                const int io = image_off<IP, ISW>(j, i * 2);
                *(bf16*)(iP + io) = (bf16)0.f;
                *(bf16*)(iS1 + io) = (bf16)0.f;
                if (!SM::ALIAS) *(bf16*)(iS2 + io) = (bf16)0.f;
            }
        }
    }
    // dQ^T and the accumulator-operand half of dC^T for this query tile (two 16-column blocks at a time: one 64-byte row store)
    f32x4 dch[CT];
    {
        bf16x8 b2[KS], b1[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { b2[ks] = pack_col<NT>(a2, ks); b1[ks] = pack_col<NT>(a1, ks); }
        bf16* rowq = DX + (int64_t)min(x, S - 1) * ld;
#pragma unroll
        for (int cp = 0; cp < CT / 2; ++cp) {
            f32x4 dqv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ct = 2 * cp + u;
                f32x4 dq = {0.f, 0.f, 0.f, 0.f}, dc = dq;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trf.template operator()<true>(tK, 32 * ks, 16 * ct), b2[ks], dq, 0, 0, 0);
                    dc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trf.template operator()<true>(tC, 32 * ks, 16 * ct), b1[ks], dc, 0, 0, 0);
                }
                dqv[u] = dq * isq;
                dch[ct] = dc;
            }
            store_row32(rowq + 32 * cp, dqv[0], dqv[1], q, live && x < Sv);
        }
    }
    COOP_STAMP(3);
    if constexpr (SM::ALIAS) {      // every wave is done with the K tile: the dS2^T image takes its place (SP2 == SP here: no padding columns)
        __syncthreads();
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) *(bf16*)(iS2 + image_off<IP, ISW>(16 * jt + 4 * q + e, x * 2)) = (bf16)a2[jt][e];
    }
    __syncthreads();
    COOP_STAMP(4);

    // ---- second half: rows x of the images (x as key index)
    {
        bf16x8 bt[KS], bp[KS], bs[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int off = image_off<IP, ISW>(x, (32 * ks + 8 * q) * 2);
            bt[ks] = *(const bf16x8*)(iS1 + off);
            bp[ks] = *(const bf16x8*)(iP + off);
            bs[ks] = *(const bf16x8*)(iS2 + off);
        }
        float dt = 0.f;
        bf16* rowx = DX + (int64_t)min(x, S - 1) * ld;
#pragma unroll
        for (int cp = 0; cp < CT / 2; ++cp) {
            f32x4 dvv[2], dkv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ct = 2 * cp + u;
                f32x4 dc = dch[ct], dv = {0.f, 0.f, 0.f, 0.f}, dk = dv;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    dc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trf.template operator()<false>(tC, 32 * ks, 16 * ct), bt[ks], dc, 0, 0, 0);
                    dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trf.template operator()<false>(tO, 32 * ks, 16 * ct), bp[ks], dv, 0, 0, 0);
                    dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trf.template operator()<false>(tQ, 32 * ks, 16 * ct), bs[ks], dk, 0, 0, 0);
                }
                dvv[u] = dv;
                dkv[u] = dk * isq;
                dch[ct] = -dc;       // dN = -dS1
                const f32x4 ch = load4<bf16>((const bf16*)(tC + tile_off<TP, TSW>(x, (16 * ct + 4 * q) * 2)));
                dt += (dch[ct][0] * ch[0] + dch[ct][1] * ch[1]) + (dch[ct][2] * ch[2] + dch[ct][3] * ch[3]);
            }
            store_row32(rowx + 2 * d + 32 * cp, dvv[0], dvv[1], q, live && x < Sv);
            store_row32(rowx + d + 32 * cp, dkv[0], dkv[1], q, live && x < Sv);
        }
        dt = red_q<NT>(dt, false);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const f32x4 ch = load4<bf16>((const bf16*)(tC + tile_off<TP, TSW>(x, (16 * ct + 4 * q) * 2)));
            dch[ct] = (dch[ct] - ch * dt) * rho_x;
        }
#pragma unroll
        for (int cp = 0; cp < CT / 2; ++cp) store_row32(rowx + 3 * d + 32 * cp, dch[2 * cp], dch[2 * cp + 1], q, live && x < Sv);
    }
    COOP_STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// Backward, tile form (S = 16 NT exactly, an S x S image no larger than an S x DH tile; built for S = 64 / head size 64, the
// C4 / C5 shapes).  Same arithmetic as attn_bwd_coop_kernel; what differs is how the operands reach the waves.  The cooperative
// form has every wave load its own MFMA fragments from global memory: 37 loads per lane, each touching sixteen 64-byte halves
// of 128-byte lines, the K / V / C rows four times per workgroup -- in-kernel stamps showed 5 600 - 9 300 cycles of a 26 000-cycle
// workgroup life spent ISSUING them (the CU's address unit serves twelve such waves).  Here the workgroup reads each row of
// Q, K, V, C and dO ONCE, as whole 128-byte lines (8 lanes per row, 10 loads per lane), stages them as five LDS tiles, and
// every fragment -- row reads for the score products, transposing reads for the gradient products -- comes from LDS.
// C stays RAW in its tile; the inverse norms are folded where C^ = C / |c| is meant: dS1 carries 1 / |c_j| into the first-half
// product, its image 1 / |c_i| into the second-half one.  LDS: five tiles + the dS1 image (48.5 KB at S = 64: three workgroups per
// CU); the P and dS2 images are written into the V and K tiles once every wave is done with those.
// ------------------------------------------------------------------------------------------------
template <int DH, int NT> struct BwdSmemT {
    static constexpr int S = 16 * NT;
    static constexpr int TILE = S * DH * 2, IMG = S * S * 2;
    static constexpr int BYTES = 5 * TILE + IMG + 2 * 64 * 4;
    static_assert(IMG <= TILE && NT % 2 == 0, "the images must fit the tiles they replace; S a multiple of 32");
};

template <int DH, int NT>
__global__ __launch_bounds__(64 * NT) __attribute__((amdgpu_waves_per_eu(3))) void attn_bwd_tiles_kernel(AttnArgs a) {
    using SM = BwdSmemT<DH, NT>;
    constexpr int S = SM::S, KD = DH / 32, CT = DH / 16, KS = S / 32;
    constexpr bool TSW = DH == 64, ISW = S == 64;        // 128-byte rows: swizzled layouts (tile_off / image_off)
    constexpr int TP = DH * 2, IP = S * 2;               // row pitches in bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int it = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int H = a.H, d = H * DH;
    const int t = blockIdx.x / H, h = blockIdx.x % H;
    char* tQ = smem;
    char* tK = tQ + SM::TILE;
    char* tV = tK + SM::TILE;
    char* tO = tV + SM::TILE;
    char* tC = tO + SM::TILE;
    char* iS1 = tC + SM::TILE;
    char* iP = tV;                  // after the first-half products
    char* iS2 = tK;
    float* rho = (float*)(iS1 + SM::IMG);
    float* madd = rho + 64;
    const bf16* X = (const bf16*)a.qkvc + (int64_t)t * S * 4 * d + h * DH;
    const bf16* DO = (const bf16*)a.dctx + (int64_t)t * S * d + h * DH;
    bf16* DX = (bf16*)a.dqkvc + (int64_t)t * S * 4 * d + h * DH;
    const int64_t ld = 4 * d;
    const float isq = rsqrtf((float)DH);
    const float mval = a.mask ? a.mask[(int64_t)t * S + lane] : 1.f;      // first (the memory counter retires in order); S <= 64 lanes
    const int x = 16 * it + r;                  // this lane's row (query i in the first half, key j in the second)
    auto trf = [&]<bool PERM>(const char* tile, int k0, int c0) {
        if constexpr (TSW) return tr_frag_swz<DH, PERM>(tile, k0, c0, r, q);
        else return tr_frag<DH, PERM>(tile, k0, c0, r, q);
    };

    // ---- this wave's 16 rows of the five tiles: whole rows, CPR lanes each
    {
        constexpr int CPR = DH / 8, RPI = 64 / CPR, NI = 16 / RPI;
        const int lr = lane / CPR, lc = lane % CPR;
        bf16x8 gq[NI], gk[NI], gv[NI], gc[NI], go[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = 16 * it + RPI * i + lr;
            const bf16* px = X + (int64_t)row * ld + lc * 8;
            gq[i] = *(const bf16x8*)px;
            gk[i] = *(const bf16x8*)(px + d);
            gv[i] = *(const bf16x8*)(px + 2 * d);
            gc[i] = *(const bf16x8*)(px + 3 * d);
            go[i] = *(const bf16x8*)(DO + (int64_t)row * d + lc * 8);
        }
        if ((lane >> 4) == it) madd[lane] = lane < S ? (1.f - mval) * -10000.f : 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = 16 * it + RPI * i + lr;
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float c = (float)gc[i][e]; ss = fmaf(c, c, ss); }
#pragma unroll
            for (int m = 1; m < CPR; m <<= 1) ss += __shfl_xor(ss, m);
            if (lc == 0) rho[row] = rsqrtf(ss);
            const int off = tile_off<TP, TSW>(row, lc * 16);
            *(bf16x8*)(tQ + off) = gq[i];
            *(bf16x8*)(tK + off) = gk[i];
            *(bf16x8*)(tV + off) = gv[i];
            *(bf16x8*)(tO + off) = go[i];
            *(bf16x8*)(tC + off) = gc[i];
        }
    }
    __syncthreads();

    // ---- first half: query tile `it`
    const float rho_x = rho[x];
    f32x4 a1[NT], a2[NT], dp[NT];
    {
        bf16x8 fq[KD], fco[KD], fo[KD];
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            const int off = tile_off<TP, TSW>(x, (32 * ks + 8 * q) * 2);
            fq[ks] = *(const bf16x8*)(tQ + off);
            fco[ks] = *(const bf16x8*)(tC + off);
            fo[ks] = *(const bf16x8*)(tO + off);
        }
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4 x1 = {0.f, 0.f, 0.f, 0.f}, x2 = x1, x3 = x1;
#pragma unroll
            for (int ks = 0; ks < KD; ++ks) {
                const int off = tile_off<TP, TSW>(16 * jt + r, (32 * ks + 8 * q) * 2);
                x1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(tC + off), fco[ks], x1, 0, 0, 0);
                x2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(tK + off), fq[ks], x2, 0, 0, 0);
                x3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(tV + off), fo[ks], x3, 0, 0, 0);
            }
            a1[jt] = x1; a2[jt] = x2; dp[jt] = x3;
        }
    }
    {
        float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const f32x4 mj = *(const f32x4*)(madd + 16 * jt + 4 * q), rj = *(const f32x4*)(rho + 16 * jt + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * jt + 4 * q + e;
                const float v1 = 1.f - a1[jt][e] * (rho_x * rj[e]) + (x == j ? 1.f : 0.f) + mj[e];
                const float v2 = a2[jt][e] * isq + mj[e];
                a1[jt][e] = v1;
                a2[jt][e] = v2;
                m1 = fmaxf(m1, v1);
                m2 = fmaxf(m2, v2);
            }
        }
        m1 = red_q<NT>(m1, true);
        m2 = red_q<NT>(m2, true);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float e1 = __expf(a1[jt][e] - m1);
                const float e2 = __expf(a2[jt][e] - m2);
                a1[jt][e] = e1;
                a2[jt][e] = e2;
                s1 += e1;
                s2 += e2;
            }
        s1 = red_q<NT>(s1, false);
        s2 = red_q<NT>(s2, false);
        const float i1 = __frcp_rn(s1), i2 = __frcp_rn(s2);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) { a1[jt] *= i1; a2[jt] *= i2; }
    }
    f32x4 pm[NT];
    {   // softmax backward of both branches; the dS1^T image (rows scaled by 1 / |c_i|); P^T and dS2^T stay in registers for now
        const DropKey k1 = make_drop_key(a.drop1), k2 = make_drop_key(a.drop2);
        const float beta = a.beta, omb = 1.f - a.beta;
        const uint64_t hbase = ((uint64_t)t * H + h) * S;
        float rd1 = 0.f, rd2 = 0.f;
        f32x4 g1[NT], g2[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            float d1[4] = {1.f, 1.f, 1.f, 1.f}, d2[4] = {1.f, 1.f, 1.f, 1.f};
            if (k1.on) {
                drop_mul4(k1, (uint32_t)(hbase + x), (uint32_t)(4 * jt + q), d1);
                drop_mul4(k2, (uint32_t)(hbase + x), (uint32_t)(4 * jt + q), d2);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float m1 = beta * d1[e], m2 = omb * d2[e];
                const float x1 = m1 * dp[jt][e], x2 = m2 * dp[jt][e];
                g1[jt][e] = x1;
                g2[jt][e] = x2;
                pm[jt][e] = m1 * a1[jt][e] + m2 * a2[jt][e];
                rd1 = fmaf(a1[jt][e], x1, rd1);
                rd2 = fmaf(a2[jt][e], x2, rd2);
            }
        }
        rd1 = red_q<NT>(rd1, false);
        rd2 = red_q<NT>(rd2, false);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const f32x4 rj = *(const f32x4*)(rho + 16 * jt + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * jt + 4 * q + e;
                const float ds1 = a1[jt][e] * (g1[jt][e] - rd1);
                a2[jt][e] = a2[jt][e] * (g2[jt][e] - rd2);
                *(bf16*)(iS1 + image_off<IP, ISW>(j, x * 2)) = (bf16)(ds1 * rho_x);      // meets C_i in the second half: carries 1 / |c_i|
                a1[jt][e] = ds1 * rj[e];                                                 // meets C_j below: carries 1 / |c_j|
            }
        }
    }
    // dQ^T and the accumulator-operand half of dC^T for this query tile (two 16-column blocks at a time: one 64-byte row store)
    f32x4 dch[CT];
    {
        bf16x8 b2[KS], b1[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { b2[ks] = pack_col<NT>(a2, ks); b1[ks] = pack_col<NT>(a1, ks); }
        bf16* rowq = DX + (int64_t)x * ld;
#pragma unroll
        for (int cp = 0; cp < CT / 2; ++cp) {
            f32x4 dqv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ct = 2 * cp + u;
                f32x4 dq = {0.f, 0.f, 0.f, 0.f}, dc = dq;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trf.template operator()<true>(tK, 32 * ks, 16 * ct), b2[ks], dq, 0, 0, 0);
                    dc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trf.template operator()<true>(tC, 32 * ks, 16 * ct), b1[ks], dc, 0, 0, 0);
                }
                dqv[u] = dq * isq;
                dch[ct] = dc;
            }
            store_row32(rowq + 32 * cp, dqv[0], dqv[1], q, true);
        }
    }
    __syncthreads();        // every wave is done with the K and V tiles: the dS2^T and P^T images take their places
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int io = image_off<IP, ISW>(16 * jt + 4 * q + e, x * 2);
            *(bf16*)(iS2 + io) = (bf16)a2[jt][e];
            *(bf16*)(iP + io) = (bf16)pm[jt][e];
        }
    __syncthreads();

    // ---- second half: rows x of the images (x as key index)
    {
        bf16x8 bt[KS], bp[KS], bs[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int off = image_off<IP, ISW>(x, (32 * ks + 8 * q) * 2);
            bt[ks] = *(const bf16x8*)(iS1 + off);
            bp[ks] = *(const bf16x8*)(iP + off);
            bs[ks] = *(const bf16x8*)(iS2 + off);
        }
        float dt = 0.f;
        f32x4 chv[CT];
        bf16* rowx = DX + (int64_t)x * ld;
#pragma unroll
        for (int cp = 0; cp < CT / 2; ++cp) {
            f32x4 dvv[2], dkv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ct = 2 * cp + u;
                f32x4 dc = dch[ct], dv = {0.f, 0.f, 0.f, 0.f}, dk = dv;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    dc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trf.template operator()<false>(tC, 32 * ks, 16 * ct), bt[ks], dc, 0, 0, 0);
                    dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trf.template operator()<false>(tO, 32 * ks, 16 * ct), bp[ks], dv, 0, 0, 0);
                    dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trf.template operator()<false>(tQ, 32 * ks, 16 * ct), bs[ks], dk, 0, 0, 0);
                }
                dvv[u] = dv;
                dkv[u] = dk * isq;
                dch[ct] = -dc;       // dN = -dS1
                chv[ct] = load4<bf16>((const bf16*)(tC + tile_off<TP, TSW>(x, (16 * ct + 4 * q) * 2))) * rho_x;        // C^_x
                dt += (dch[ct][0] * chv[ct][0] + dch[ct][1] * chv[ct][1]) + (dch[ct][2] * chv[ct][2] + dch[ct][3] * chv[ct][3]);
            }
            store_row32(rowx + 2 * d + 32 * cp, dvv[0], dvv[1], q, true);
            store_row32(rowx + d + 32 * cp, dkv[0], dkv[1], q, true);
        }
        dt = red_q<NT>(dt, false);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dch[ct] = (dch[ct] - chv[ct] * dt) * rho_x;
#pragma unroll
        for (int cp = 0; cp < CT / 2; ++cp) store_row32(rowx + 3 * d + 32 * cp, dch[2 * cp], dch[2 * cp + 1], q, true);
    }
}

// Forward, cooperative form: wave `it` of the NT waves of a (sequence, head) owns query tile it; the V tile is
// shared through LDS (each wave stores its own 16 rows), everything else stays in registers.
template <int DH, int NT, bool FULL>
__global__ __launch_bounds__(CoopCfg<NT>::THREADS) void attn_fwd_coop_kernel(AttnArgs a) {
    using SM = FwdSmem<DH, NT>;
    constexpr int KD = DH / 32, CT = DH / 16, KS = SM::SP2 / 32, SP = SM::SP, SP2 = SM::SP2, G = CoopCfg<NT>::G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int ul = wave / NT, it = wave % NT;
    const int gidx = blockIdx.x * G + ul;
    const int S = a.S, H = a.H, d = H * DH;
    const bool act = gidx < a.Tseq * H;
    const int t = act ? gidx / H : 0, h = act ? gidx % H : 0;
    char* base = smem + ul * SM::BYTES;
    char* tV = base;
    float* rho = (float*)(base + SM::TILE);
    float* madd = rho + 64;
    const bf16* X = (const bf16*)a.qkvc + (int64_t)t * S * 4 * d + h * DH;
    const int64_t ld = 4 * d;
    const int Sv = FULL ? 16 * NT : (act ? S : 0);
    const bool live = FULL ? act : true;
    const int x = 16 * it + r;
    const float mval = a.mask ? a.mask[(int64_t)t * S + min(lane, S - 1)] : 1.f;      // first: see attn_bwd_coop_kernel

    bf16x8 fq[KD], fco[KD], fvo[KD], fk[NT][KD], fc[NT][KD];
#pragma unroll
    for (int ks = 0; ks < KD; ++ks) {
        fq[ks] = ld_rows_f<FULL>(X, ld, x, Sv, 32 * ks + 8 * q);
        fvo[ks] = ld_rows_f<FULL>(X + 2 * d, ld, x, Sv, 32 * ks + 8 * q);
        fco[ks] = ld_rows_f<FULL>(X + 3 * d, ld, x, Sv, 32 * ks + 8 * q);
    }
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            fk[jt][ks] = ld_rows_f<FULL>(X + d, ld, 16 * jt + r, Sv, 32 * ks + 8 * q);
            fc[jt][ks] = ld_rows_f<FULL>(X + 3 * d, ld, 16 * jt + r, Sv, 32 * ks + 8 * q);
        }
    float rho_x;
    {
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < KD; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float c = (float)fco[ks][e]; ss = fmaf(c, c, ss); }
        ss = red_q<NT>(ss, false);
        rho_x = x < Sv ? rsqrtf(ss) : 0.f;
        if (q == 0) rho[x] = rho_x;
        if ((lane >> 4) == it) madd[lane] = lane < Sv ? (1.f - mval) * -10000.f : 0.f;
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) *(bf16x8*)(tV + x * (DH * 2) + (32 * ks + 8 * q) * 2) = fvo[ks];
        if (SP2 > SP && it == NT - 1) {
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int idx = lane; idx < (SP2 - SP) * (DH / 8); idx += 64)
                *(bf16x8*)(tV + (SP + idx / (DH / 8)) * (DH * 2) + (idx % (DH / 8)) * 16) = z;
        }
    }
    __syncthreads();

    f32x4 a1[NT], a2[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        f32x4 x1 = {0.f, 0.f, 0.f, 0.f}, x2 = x1;
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            x1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[jt][ks], fco[ks], x1, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[jt][ks], fq[ks], x2, 0, 0, 0);
        }
        a1[jt] = x1; a2[jt] = x2;
    }
    {
        const bool iv = x < Sv;
        const float isq = rsqrtf((float)DH);
        float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * jt + 4 * q + e;
                const bool ok = iv && j < Sv;
                const float v1 = ok ? (1.f - a1[jt][e] * (rho_x * rho[j]) + (x == j ? 1.f : 0.f) + madd[j]) : -INFINITY;
                const float v2 = ok ? (a2[jt][e] * isq + madd[j]) : -INFINITY;
                a1[jt][e] = v1;
                a2[jt][e] = v2;
                m1 = fmaxf(m1, v1);
                m2 = fmaxf(m2, v2);
            }
        m1 = red_q<NT>(m1, true);
        m2 = red_q<NT>(m2, true);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float e1 = iv ? __expf(a1[jt][e] - m1) : 0.f;
                const float e2 = iv ? __expf(a2[jt][e] - m2) : 0.f;
                a1[jt][e] = e1;
                a2[jt][e] = e2;
                s1 += e1;
                s2 += e2;
            }
        s1 = red_q<NT>(s1, false);
        s2 = red_q<NT>(s2, false);
        const float i1 = iv ? __frcp_rn(s1) : 0.f, i2 = iv ? __frcp_rn(s2) : 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) { a1[jt] *= i1; a2[jt] *= i2; }
    }
    {   // mix + dropout -> P^T (in a1)
        const DropKey k1 = make_drop_key(a.drop1), k2 = make_drop_key(a.drop2);
        const float beta = a.beta, omb = 1.f - a.beta;
        const uint64_t hbase = ((uint64_t)t * H + h) * S;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            float d1[4] = {1.f, 1.f, 1.f, 1.f}, d2[4] = {1.f, 1.f, 1.f, 1.f};
            if (k1.on) {
                drop_mul4(k1, (uint32_t)(hbase + x), (uint32_t)(4 * jt + q), d1);
                drop_mul4(k2, (uint32_t)(hbase + x), (uint32_t)(4 * jt + q), d2);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * jt + 4 * q + e;
                const float p = beta * d1[e] * a1[jt][e] + omb * d2[e] * a2[jt][e];
                a1[jt][e] = p;
                if (a.probs && live && x < Sv && j < Sv) a.probs[(hbase + x) * S + j] = p;
            }
        }
    }
    bf16x8 pb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) pb[ks] = pack_col<NT>(a1, ks);
    f32x4 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            o[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH, true>(tV, 32 * ks, 16 * ct, r, q), pb[ks], o[ct], 0, 0, 0);
    }
#pragma unroll
    for (int cp = 0; cp < CT / 2; ++cp)
        store_row32((bf16*)a.ctx + ((int64_t)t * S + min(x, S - 1)) * d + h * DH + 32 * cp, o[2 * cp], o[2 * cp + 1], q, live && x < Sv);
}

// ------------------------------------------------------------------------------------------------
// Forward, tile form (S = 16 NT exactly; built for S = 64 / head size 64): the counterpart of attn_bwd_tiles_kernel.  The
// workgroup reads each row of Q, K, V, C once as whole 128-byte lines (8 loads per lane instead of 22 that touch sixteen half
// lines each), stages four LDS tiles, and the score / context products take their fragments from there.  Same arithmetic as
// attn_fwd_coop_kernel.  LDS 32.5 KB at S = 64: four workgroups per CU.
// ------------------------------------------------------------------------------------------------
template <int DH, int NT> struct FwdSmemT {
    static constexpr int S = 16 * NT;
    static constexpr int TILE = S * DH * 2;
    static constexpr int BYTES = 4 * TILE + 2 * 64 * 4;
    static_assert(NT % 2 == 0, "S a multiple of 32");
};

template <int DH, int NT>
__global__ __launch_bounds__(64 * NT) void attn_fwd_tiles_kernel(AttnArgs a) {
    using SM = FwdSmemT<DH, NT>;
    constexpr int S = SM::S, KD = DH / 32, CT = DH / 16, KS = S / 32;
    constexpr bool TSW = DH == 64;
    constexpr int TP = DH * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int it = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int H = a.H, d = H * DH;
    const int t = blockIdx.x / H, h = blockIdx.x % H;
    char* tQ = smem;
    char* tK = tQ + SM::TILE;
    char* tV = tK + SM::TILE;
    char* tC = tV + SM::TILE;
    float* rho = (float*)(tC + SM::TILE);
    float* madd = rho + 64;
    const bf16* X = (const bf16*)a.qkvc + (int64_t)t * S * 4 * d + h * DH;
    const int64_t ld = 4 * d;
    const float mval = a.mask ? a.mask[(int64_t)t * S + lane] : 1.f;      // first: the memory counter retires in order
    const int x = 16 * it + r;
    {
        constexpr int CPR = DH / 8, RPI = 64 / CPR, NI = 16 / RPI;
        const int lr = lane / CPR, lc = lane % CPR;
        bf16x8 gq[NI], gk[NI], gv[NI], gc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bf16* px = X + (int64_t)(16 * it + RPI * i + lr) * ld + lc * 8;
            gq[i] = *(const bf16x8*)px;
            gk[i] = *(const bf16x8*)(px + d);
            gv[i] = *(const bf16x8*)(px + 2 * d);
            gc[i] = *(const bf16x8*)(px + 3 * d);
        }
        if ((lane >> 4) == it) madd[lane] = lane < S ? (1.f - mval) * -10000.f : 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = 16 * it + RPI * i + lr;
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float c = (float)gc[i][e]; ss = fmaf(c, c, ss); }
#pragma unroll
            for (int m = 1; m < CPR; m <<= 1) ss += __shfl_xor(ss, m);
            if (lc == 0) rho[row] = rsqrtf(ss);
            const int off = tile_off<TP, TSW>(row, lc * 16);
            *(bf16x8*)(tQ + off) = gq[i];
            *(bf16x8*)(tK + off) = gk[i];
            *(bf16x8*)(tV + off) = gv[i];
            *(bf16x8*)(tC + off) = gc[i];
        }
    }
    __syncthreads();

    const float rho_x = rho[x];
    f32x4 a1[NT], a2[NT];
    {
        bf16x8 fq[KD], fco[KD];
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            const int off = tile_off<TP, TSW>(x, (32 * ks + 8 * q) * 2);
            fq[ks] = *(const bf16x8*)(tQ + off);
            fco[ks] = *(const bf16x8*)(tC + off);
        }
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4 x1 = {0.f, 0.f, 0.f, 0.f}, x2 = x1;
#pragma unroll
            for (int ks = 0; ks < KD; ++ks) {
                const int off = tile_off<TP, TSW>(16 * jt + r, (32 * ks + 8 * q) * 2);
                x1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(tC + off), fco[ks], x1, 0, 0, 0);
                x2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(tK + off), fq[ks], x2, 0, 0, 0);
            }
            a1[jt] = x1; a2[jt] = x2;
        }
    }
    {
        const float isq = rsqrtf((float)DH);
        float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const f32x4 mj = *(const f32x4*)(madd + 16 * jt + 4 * q), rj = *(const f32x4*)(rho + 16 * jt + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 16 * jt + 4 * q + e;
                const float v1 = 1.f - a1[jt][e] * (rho_x * rj[e]) + (x == j ? 1.f : 0.f) + mj[e];
                const float v2 = a2[jt][e] * isq + mj[e];
                a1[jt][e] = v1;
                a2[jt][e] = v2;
                m1 = fmaxf(m1, v1);
                m2 = fmaxf(m2, v2);
            }
        }
        m1 = red_q<NT>(m1, true);
        m2 = red_q<NT>(m2, true);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float e1 = __expf(a1[jt][e] - m1);
                const float e2 = __expf(a2[jt][e] - m2);
                a1[jt][e] = e1;
                a2[jt][e] = e2;
                s1 += e1;
                s2 += e2;
            }
        s1 = red_q<NT>(s1, false);
        s2 = red_q<NT>(s2, false);
        const float i1 = __frcp_rn(s1), i2 = __frcp_rn(s2);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) { a1[jt] *= i1; a2[jt] *= i2; }
    }
    {   // mix + dropout -> P^T (in a1)
        const DropKey k1 = make_drop_key(a.drop1), k2 = make_drop_key(a.drop2);
        const float beta = a.beta, omb = 1.f - a.beta;
        const uint64_t hbase = ((uint64_t)t * H + h) * S;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            float d1[4] = {1.f, 1.f, 1.f, 1.f}, d2[4] = {1.f, 1.f, 1.f, 1.f};
            if (k1.on) {
                drop_mul4(k1, (uint32_t)(hbase + x), (uint32_t)(4 * jt + q), d1);
                drop_mul4(k2, (uint32_t)(hbase + x), (uint32_t)(4 * jt + q), d2);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) a1[jt][e] = beta * d1[e] * a1[jt][e] + omb * d2[e] * a2[jt][e];
            if (a.probs) *(f32x4*)(a.probs + (hbase + x) * S + 16 * jt + 4 * q) = a1[jt];
        }
    }
    bf16x8 pb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) pb[ks] = pack_col<NT>(a1, ks);
    f32x4 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 fa;
            if constexpr (TSW) fa = tr_frag_swz<DH, true>(tV, 32 * ks, 16 * ct, r, q);
            else fa = tr_frag<DH, true>(tV, 32 * ks, 16 * ct, r, q);
            o[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, pb[ks], o[ct], 0, 0, 0);
        }
    }
#pragma unroll
    for (int cp = 0; cp < CT / 2; ++cp)
        store_row32((bf16*)a.ctx + ((int64_t)t * S + x) * d + h * DH + 32 * cp, o[2 * cp], o[2 * cp + 1], q, true);
}

template <int DH, int NT> static int launch_mfma(const AttnArgs& a, bool bwd, hipStream_t st) {
    const int groups = a.Tseq * a.H;
    // NT cooperating waves per (sequence, head) everywhere except S in 17..32 with head size 32, where the
    // one-wave form measures faster (464 vs 555 us backward at 98k pairs: the shared loads outweigh the occupancy)
    const bool use_coop = !(a.opts & OPT_WAVE_ATTENTION_BWD) && !(NT == 2 && DH == 32);
    if constexpr (NT == 4 && DH == 64) {
        if (!bwd && use_coop && a.S == 16 * NT && !(a.opts & OPT_NO_TILE_ATTENTION)) {
            constexpr int lds = FwdSmemT<DH, NT>::BYTES;
            note_launch(LT_ATTN_TILES_FWD);
            hipLaunchKernelGGL((attn_fwd_tiles_kernel<DH, NT>), dim3(groups), dim3(64 * NT), lds, st, a);
            PMGT_LAUNCH_OK();
            return 0;
        }
    }
    if (!bwd && use_coop) {
        constexpr int G = CoopCfg<NT>::G;
        const size_t shmem = (size_t)FwdSmem<DH, NT>::BYTES * G;
        if (a.S == 16 * NT) hipLaunchKernelGGL((attn_fwd_coop_kernel<DH, NT, true>), dim3(cdiv(groups, G)), dim3(CoopCfg<NT>::THREADS), shmem, st, a);
        else hipLaunchKernelGGL((attn_fwd_coop_kernel<DH, NT, false>), dim3(cdiv(groups, G)), dim3(CoopCfg<NT>::THREADS), shmem, st, a);
        PMGT_LAUNCH_OK();
        return 0;
    }
    if (!bwd) {
        const size_t shmem = (size_t)FwdSmem<DH, NT>::BYTES * 4;
        auto kern = attn_fwd_mfma_kernel<DH, NT>;
        if (shmem > 64 * 1024) PMGT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(kern, dim3(cdiv(groups, 4)), dim3(256), shmem, st, a);
    } else {
        if constexpr (NT == 4 && DH == 64) {
            if (use_coop && a.S == 16 * NT && !(a.opts & OPT_NO_TILE_ATTENTION)) {
                auto kern = attn_bwd_tiles_kernel<DH, NT>;
                constexpr int lds = BwdSmemT<DH, NT>::BYTES;
                PMGT_SMEM_ATTR((const void*)kern, lds);
                note_launch(LT_ATTN_TILES_BWD);
                hipLaunchKernelGGL(kern, dim3(groups), dim3(64 * NT), lds, st, a);
                PMGT_LAUNCH_OK();
                return 0;
            }
        }
        if (use_coop) {
            constexpr int G = CoopCfg<NT>::G;
            const size_t shmem = (size_t)BwdSmemC<DH, NT>::BYTES * G;
            auto kern = a.S == 16 * NT ? attn_bwd_coop_kernel<DH, NT, true> : attn_bwd_coop_kernel<DH, NT, false>;
            if (shmem > 64 * 1024) PMGT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            hipLaunchKernelGGL(kern, dim3(cdiv(groups, G)), dim3(CoopCfg<NT>::THREADS), shmem, st, a);
            PMGT_LAUNCH_OK();
            return 0;
        }
        const size_t per = BwdSmemW<DH, NT>::BYTES;
        // waves per workgroup: 2 when that raises the LDS-limited wave count per CU (S=32/dh=32: 14.5 KiB per wave ->
        // 5 x 2 waves instead of 2 x 4; measured 476 vs 500 us at 98k (sequence, head) pairs), never 1 (slower)
        constexpr size_t LDS = 160 * 1024;
        int nw = per * 2 <= LDS ? 2 : 1;
        if (per * 4 <= LDS && (LDS / (per * 4)) * 4 > (LDS / (per * 2)) * 2) nw = 4;
        const size_t shmem = per * nw;
        const bool full = a.S == NT * 16 && NT % 2 == 0;
        auto kern = full ? attn_bwd_mfma_kernel<DH, NT, true> : attn_bwd_mfma_kernel<DH, NT, false>;
        if (shmem > 64 * 1024) PMGT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(kern, dim3(cdiv(groups, nw)), dim3(64 * nw), shmem, st, a);
    }
    PMGT_LAUNCH_OK();
    return 0;
}

template <int DH> static int launch_nt(const AttnArgs& a, bool bwd, hipStream_t st) {
    const int nt = cdiv(a.S, 16);
    switch (nt) {
        case 1: return launch_mfma<DH, 1>(a, bwd, st);
        case 2: return launch_mfma<DH, 2>(a, bwd, st);
        case 3: return launch_mfma<DH, 3>(a, bwd, st);
        default: return launch_mfma<DH, 4>(a, bwd, st);
    }
}


// =================================================================================================
// Attention backward FUSED with the weight gradient of the Q|K|V|C projection (bf16, S = 32, head size 32).
//
// The unfused pair is VALU-bound (attention backward: ~1 100 VALU instructions per (sequence, head), matrix pipe 6 % busy)
// followed by memory/MFMA-bound (weight gradient: re-reads the 4U of dQ|dK|dV|dC the attention just wrote, plus x, and
// pushes 256 MB of split-K partial slabs through HBM per layer).  Here one 1024-thread workgroup per CU owns ONE head
// (128 columns of Q|K|V|C) and walks over pairs of sequences with its sixteen waves in two ROLES (wave-uniform, scalar branches):
//   * waves 0-7 (two per SIMD): attention backward of the step's two (sequence, head) pairs, FOUR waves per pair:
//     (query / key tile it) x (softmax branch br: 1 = the cosine "diversity" branch, 2 = the scaled dot-product branch).
//     A branch wave computes its own scores, softmax, dropout mask and softmax backward; the only things the two branches
//     share are dP = dO V^T (each computes it: two MFMAs) and the mixed probabilities, which meet as two images that the
//     dV product accumulates one after the other.  They touch HBM only for the 32 mask values of their sequence: Q|K|V|C
//     and dO rows arrive in LDS tiles (below), dQ|dK|dV|dC leave through an LDS tile.  Written for a short VALU chain
//     (log2-domain scores, predicated dropout, 8-byte swizzled image writes read back with ds_read_b64_tr_b16; C^ = C / |C|
//     is never materialised: the inverse norms are folded into dS1 on its way into the two products that consume it);
//   * waves 8-15 (the other two waves of each SIMD): the matrix pipe's customers and the workgroup's DMA engine.  Per step
//     they request the NEXT step's rows of Q|K|V|C (256 contiguous bytes per row and head in the head-major layout), dO and x
//     by LDS-DMA (no staging registers), accumulate dW_head[128, d] += dQKVC_tile^T x_tile of the PREVIOUS step in 64
//     accumulator registers each for the whole life of the workgroup (+ the bias gradient as one more MFMA against an
//     all-ones operand), and copy that tile to HBM with 16-byte row-contiguous stores.
// Why sixteen waves: a wave alone on its SIMD issues one VALU instruction per 4 cycles, two waves together one per 2 -- the
// first form of this kernel (4 + 4 waves) spent 5 800 cycles per step in its attention role with the vector ALU half idle.
// Why the attention waves load nothing themselves: 16 waves leave 128 VGPRs per wave, a register prefetch of the next step's
// fragments does not fit next to the working set, and a spilled register's reload waits with vmcnt(0) -- for the prefetch.
// The weight gradient leaves the kernel as ONE [128, d] partial per workgroup (32 partials per head instead of 256 MB of
// slabs).  Three raw s_barriers per step order the hand-offs; every LDS access of the DMA-issuing waves is inline asm with its
// own waits, because the compiler puts s_waitcnt vmcnt(0) in front of any LDS read it can see while an LDS-DMA is in flight.
// LDS (d = 256): x tiles 2 x 32 KB | Q|K|V|C-in / dQKVC-out ring 4 x 16 KB | dO tiles 2 x 4 KB | images + norms 2 x 8.25 KB.
// =================================================================================================
#ifdef PMGT_ABW_PROF
// cycles per interval of a step (work / barrier wait alternating: w1 b1 w2 b2 w3 b0), summed over the steps of workgroup 0:
// [wave][interval]; slot 7 = steps
__device__ unsigned long long g_abw_prof[16][8];
#define ABW_STAMP(k_) do { const unsigned long long n_ = __builtin_readcyclecounter(); pacc[k_] += n_ - plast; plast = n_; } while (0)
#else
#define ABW_STAMP(k_) do { } while (0)
#endif

// PMGT_ABW_MARK (tools/isa_mix.py builds with it): comment lines in the ISA that name the role / phase the following instructions
// belong to -- the per-role, per-phase instruction-mix table under profiles/.  No instruction is emitted for a marker.
#ifdef PMGT_ABW_MARK
#define ABW_MARK(txt) asm volatile("; ABW_MARK " txt ::: "memory")
#define ABW_MARK2(txt, a_, b_) asm volatile("; ABW_MARK " txt " it=%0 br=%1" :: "n"(a_), "n"(b_) : "memory")
#else
#define ABW_MARK(txt) do { } while (0)
#define ABW_MARK2(txt, a_, b_) do { } while (0)
#endif

template <int KT> struct AbwCfg {
    static constexpr int D = 16 * KT;                 // hidden size (256 or 128)
    static constexpr int XROW = D * 2;                // bytes per row of x
    static constexpr int XB = 64 * XROW;              // x tile: the two sequences of a step
    static constexpr int GB = 64 * 256;               // Q|K|V|C (in) or dQ|dK|dV|dC (out) tile of one head for the two sequences
    static constexpr int OB = 64 * 64;                // dO tile
    static constexpr int SCR = 4 * 2048 + 256;        // per pair: dS1, dS2, P1, P2 images | rho, madd
    // ring of four 16 KB tiles: step i reads in-tile (2 i) & 3 ... out-tile (2 i + 1) & 3
    static constexpr int G0 = 2 * XB, O0 = G0 + 4 * GB, S0 = O0 + 2 * OB, SMEM = S0 + 2 * SCR;
    static constexpr int KQ = KT / 4;                 // 16-column k tiles per GEMM wave
};

// swizzles (all at the granularity the reads and writes share, so they are transparent to the transposing reads):
//   x / Q|K|V|C / dQKVC tiles: 16-byte chunk c of row -> c ^ abw_f(row), abw_f = 2 * (bits 0, 1, 3 of the row): the 8 rows x 2
//   chunks one ds_read_b64_tr_b16 half-wave touches land on 16 different chunk slots, and "+4 rows" stays a plain address offset;
//   dO tile (64-byte rows): 16-byte chunk c -> c ^ abw_kt(row); images: 8-byte slot s -> s ^ ((row >> 1) & 7)
__device__ __forceinline__ int abw_f(int row) { return 2 * ((row & 3) | ((row >> 1) & 4)); }
__device__ __forceinline__ int abw_kt(int row) { return ((row >> 1) ^ (row >> 3)) & 3; }
__device__ __forceinline__ int abw_o_addr(int row, int byte_col) {         // byte_col multiple of 8, 64-byte rows
    return row * 64 + ((((byte_col >> 4) ^ abw_kt(row))) << 4) + (byte_col & 15);
}
__device__ __forceinline__ int abw_img_addr(int row, int slot) { return row * 64 + ((slot ^ ((row >> 1) & 7)) << 3); }
__device__ __forceinline__ int abw_g_addr(int row, int byte_col) {         // byte_col multiple of 8, 256-byte rows
    return row * 256 + (((byte_col >> 4) ^ abw_f(row)) << 4) + (byte_col & 15);
}

// transposing fragment reads: element e of lane (r, q) = tile[krow(q, e)][column c0 + r of the 32-column block at byte column mb]
//   PERM: krow = 16 (e >> 2) + 4 q + (e & 3) (matches an accumulator tile used as the other operand); else krow = 8 q + e
template <bool PERM> __device__ __forceinline__ bf16x8 abw_tr_g(const char* tile, int mb, int c0, int r, int q) {
    const int row_lo = PERM ? (4 * q + (r >> 2)) : (8 * q + (r >> 2)), row_hi = row_lo + (PERM ? 16 : 4);
    const int colb = mb + (c0 + 4 * (r & 3)) * 2;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + abw_g_addr(row_lo, colb)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + abw_g_addr(row_hi, colb)));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 abw_tr_o(const char* tile, int c0, int r, int q) {        // dO tile, natural k order
    const int row_lo = 8 * q + (r >> 2), row_hi = row_lo + 4;
    const int colb = (c0 + 4 * (r & 3)) * 2;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + abw_o_addr(row_lo, colb)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + abw_o_addr(row_hi, colb)));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 abw_tr_img(const char* img, int c0, int r, int q) {      // element e = img[8 q + e][c0 + r]
    const int row_lo = 8 * q + (r >> 2), row_hi = row_lo + 4, slot = (c0 >> 2) + (r & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + abw_img_addr(row_lo, slot)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + abw_img_addr(row_hi, slot)));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// One step of an attention wave, as three PHASES separated by the workgroup's barriers: pair = rows [0, 32) of `gin` (Q|K|V|C, 256-byte rows) /
// `oin` (dO, 64-byte rows), query / key tile IT, softmax branch BR; results into `gout`; `scr` = the pair's images and norms.
//   phase 1: fragments from the tiles, inverse norms of C + mask term (cosine wave), score and dP = dO V^T MFMAs
//   phase 2: this branch's softmax, dropout, softmax backward, dS / P images; dQ (dot-product wave) or the accumulator half of dC (cosine wave)
//   phase 3: key tile IT: dV, dK (dot-product wave) or dC (cosine wave)
// Phase k + 1 reads what the OTHER waves of the pair wrote in phase k, so consecutive phases of a pair sit on either side of a barrier.  What a
// wave carries between phases is explicit (AbwCarry): the kernel runs the two pairs of a step ONE INTERVAL APART (pair 1 is in phase k - 1
// while pair 0 is in phase k), so the waves that share a SIMD -- w and w + 4: same tile and branch, the two pairs -- are never both in the
// instruction-heavy second phase.
struct AbwStep {      // (wave-uniform) what a phase needs to know about its step
    const char* gin;
    const char* oin;
    char* gout;
    int t;            // sequence (clamped into the tensor), for the dropout row index
    bool act;         // the sequence exists
    bool zq1;         // query tile 1 (rows 16 .. 31) of this sequence has no gradient: d ctx is non-zero at row 0 only (t < cls_only_seqs)
    int vcol = 128, ccol = 192;      // byte columns of this wave's V and C blocks inside a tile row ([Q K V C] of one head; VC 2: [V0 C0 V1 C1])
};
struct AbwCarry {
    f32x4 sc[2], dp[2];      // phase 1 -> 2: raw scores and dP of this wave's query tile against both key tiles
    f32x4 dch[2];            // phase 2 -> 3 (cosine wave): accumulator half of dC^T
    float rho_x, ss;         // phase 1 -> 2 -> 3 (cosine wave): 1 / |c_x| (0 for a sequence that does not exist), |c_x|^2
    f32x4 mval[2];           // phase 1 -> 2 (dot-product wave): dropout multipliers of its eight (query, key) elements
};

// dropout multipliers of a wave's eight (query, key) elements: branch weight x dropout scale, or 0 where dropped -- a function of (sequence,
// head, row, key) only
template <int BR>
__device__ __forceinline__ void abw_draw_mval(const AttnArgs& a, const DropKey& kd, int t, int h, int x, int q, f32x4 (&mval)[2]) {
    const float cw = (BR == 1 ? a.beta : 1.f - a.beta) * kd.scale;
    const uint32_t hrow = (uint32_t)((((uint64_t)t * a.H + h) * 32) + x);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        bool kp[4] = {true, true, true, true};
        if (kd.on) drop_keep4(kd, hrow, (uint32_t)(4 * jt + q), kp);
#pragma unroll
        for (int e = 0; e < 4; ++e) mval[jt][e] = kp[e] ? cw : 0.f;
    }
}

// VC 1 (beta == 1, AttnArgs::vc_only): the dot-product branch is dead -- its wave (BR 2) keeps only its 16 columns of dV (third phase, from the
// cosine branch's image alone); no Q / K fragments, scores, softmax, dS2 / P2 images, dQ or dK.  VC 2 (the two-heads-per-step form of the same
// mode): every attention wave is a cosine wave (BR 1) of one (pair, head, tile); it also has all 32 columns of its dV, and the pair's scratch
// holds dS1 | P1 | norms only.
template <int VC> struct AbwScr { static constexpr int P1 = VC == 2 ? 2048 : 4096, P2 = 6144, RHO = VC == 2 ? 4096 : 8192; };
template <int IT, int BR, int VC = 0>
__device__ __forceinline__ void abw_phase1(const AttnArgs& a, const DropKey& kd, const AbwStep& st, AbwCarry& cy, float mraw, int h, char* scr, int r, int q, int lane) {
    constexpr float L2E = 1.4426950408889634f;
    if (VC == 1 && BR == 2) return;
    float* rho = (float*)(scr + AbwScr<VC>::RHO);
    float* madd = rho + 32;
    const int x = 16 * IT + r;
    const char* gin = st.gin;
    // A query tile without a gradient (last layer of the shortcut path: dO rows 16 .. 31 of a CLS-only sequence are zero) gives dP = 0, hence
    // dS = 0, dQ = 0 and a zero accumulator half of dC for its rows, exactly: its waves keep what the OTHER tile's waves read from them (the
    // norms 1 / |c_x| and the mask terms of keys 16 .. 31: cosine wave) and write zero images in phase 2; phase 3 (their KEY tile) is unchanged.
    const bool zq = IT == 1 && st.zq1;      // (wave-uniform)
    if (zq && BR == 2) return;
    ABW_MARK2("attn I1.fragments+norms", IT, BR);
    bf16x8 fown, kc[2], fv[2], fo;      // own Q rows (branch 2) | K or C rows of all keys | V rows of all keys | own dO rows
    if (BR == 2) fown = *(const bf16x8*)(gin + abw_g_addr(x, 16 * q));
    fo = *(const bf16x8*)(st.oin + abw_o_addr(x, 16 * q));
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        kc[jt] = *(const bf16x8*)(gin + abw_g_addr(16 * jt + r, (BR == 1 ? st.ccol : 64) + 16 * q));
        fv[jt] = *(const bf16x8*)(gin + abw_g_addr(16 * jt + r, st.vcol + 16 * q));
    }
    float ss = 0.f, rho_x = 0.f;
    if (BR == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16x2_t c2 = {kc[IT][2 * e], kc[IT][2 * e + 1]};
            ss = __builtin_amdgcn_fdot2_f32_bf16(c2, c2, ss, false);
        }
        ss = red_q<2>(ss, false);
        rho_x = st.act ? __builtin_amdgcn_rsqf(ss) : 0.f;
        if (q == 0) rho[x] = rho_x;
    }
    if (BR == 1) {
        // the mask term sits with the cosine wave: the dot-product wave reads one fragment more, and the in-kernel stamps showed its
        // first interval at 1 340 cycles against 770 here with the mask term over there
        const float mv = (1.f - mraw) * -10000.f;
        float mm = mv;
        mm = raw_max(mm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mm), 0x121, 0xf, 0xf, false)));
        mm = raw_max(mm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mm), 0x122, 0xf, 0xf, false)));
        mm = raw_max(mm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mm), 0x124, 0xf, 0xf, false)));
        mm = raw_max(mm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mm), 0x128, 0xf, 0xf, false)));
        mm = red_q<2>(mm, true);
        if ((lane >> 4) == IT) madd[lane & 31] = (mv - mm) * L2E;      // (each tile's cosine wave writes 16 of the 32 entries)
    }
    cy.rho_x = rho_x;
    cy.ss = ss;
    if (zq) return;
    // scores (transposed: key on (q, e), query on r) do not depend on the other waves: the matrix pipe starts before the barrier
    ABW_MARK2("attn I1.scores_mfma", IT, BR);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        f32x4 z1 = {0.f, 0.f, 0.f, 0.f};
        if (BR == 1 && jt == IT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) z1[e] = (4 * q + e == r) ? -ss : 0.f;      // "+ I": |c_i|^2 rho_i^2 = 1 on the diagonal
        }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        cy.sc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc[jt], BR == 1 ? kc[IT] : fown, z1, 0, 0, 0);
        cy.dp[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[jt], fo, z, 0, 0, 0);
    }
    // The dot-product wave draws its dropout multipliers HERE, behind its MFMAs: it reaches the barrier long before the cosine wave (stamps:
    // 840 against 1 310 cycles), while its second phase is the longer of the two.  The cosine wave, the pole of this phase, draws them in phase 2.
    if (BR == 2) abw_draw_mval<BR>(a, kd, st.t, h, x, q, cy.mval);
}

template <int IT, int BR, int VC = 0>
__device__ __forceinline__ void abw_phase2(const AttnArgs& a, const DropKey& kd, const AbwStep& st, AbwCarry& cy, int h, char* scr, int r, int q) {
    constexpr float L2E = 1.4426950408889634f;
    constexpr float ISQ = 0.17677669529663687f;          // 1 / sqrt(32)
    if (VC == 1 && BR == 2) return;
    char* iS = scr + (BR == 1 ? 0 : 2048);               // dS1 (rows scaled by 1 / |c_i|) | dS2
    char* iP1 = scr + AbwScr<VC>::P1;
    char* iP2 = scr + AbwScr<VC>::P2;
    const float* rho = (const float*)(scr + AbwScr<VC>::RHO);
    const float* madd = rho + 32;
    const int x = 16 * IT + r;
    const float rho_x = cy.rho_x;
    f32x4 (&sc)[2] = cy.sc;
    f32x4 (&dp)[2] = cy.dp;
    if (IT == 1 && st.zq1) {      // (wave-uniform) no gradient reaches this query tile: dS rows, P rows (they meet dO = 0), dQ rows and the dC half are zero
        const bf16x4 zero4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int ia = abw_img_addr(x, 4 * jt + q);
            *(bf16x4*)(iS + ia) = zero4;
            *(bf16x4*)((BR == 1 ? iP1 : iP2) + ia) = zero4;
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            if (BR == 2) *(bf16x4*)(st.gout + abw_g_addr(x, (16 * ct + 4 * q) * 2)) = zero4;
            else cy.dch[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    ABW_MARK2("attn I2.softmax", IT, BR);
    // first half, query tile IT: this branch's softmax and its backward
    f32x4 rj[2];
    {
        const float fac = BR == 1 ? -rho_x * L2E : ISQ * L2E;
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const f32x4 mj = *(const f32x4*)(madd + 16 * jt + 4 * q);
            rj[jt] = (f32x4){1.f, 1.f, 1.f, 1.f};
            if (BR == 1) rj[jt] = *(const f32x4*)(rho + 16 * jt + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sc[jt][e] = fmaf(sc[jt][e], BR == 1 ? fac * rj[jt][e] : fac, mj[e]);
                if (BR == 2) mx = raw_max(mx, sc[jt][e]);      // the cosine branch is bounded (-cos + I <= 2): no row maximum
            }
        }
        if (BR == 2) mx = red_q<2>(mx, true);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sc[jt][e] = __builtin_amdgcn_exp2f(BR == 2 ? sc[jt][e] - mx : sc[jt][e]);
                sum += sc[jt][e];
            }
        sum = red_q<2>(sum, false);
        const float inv = st.act ? __builtin_amdgcn_rcpf(sum) : 0.f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) sc[jt] *= inv;
    }
    ABW_MARK2("attn I2.dropout+softmax_bwd+images", IT, BR);
    {
        if (BR == 1) abw_draw_mval<BR>(a, kd, st.t, h, x, q, cy.mval);
        float rd = 0.f;
        f32x4 gr[2], pm[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gr[jt][e] = cy.mval[jt][e] * dp[jt][e];
                pm[jt][e] = cy.mval[jt][e] * sc[jt][e];
                rd = fmaf(sc[jt][e], gr[jt][e], rd);
            }
        }
        rd = red_q<2>(rd, false);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sc[jt][e] = sc[jt][e] * (gr[jt][e] - rd);      // dS of this branch
            const int ia = abw_img_addr(x, 4 * jt + q);          // images are [query i][key j]: 4 consecutive keys = one 8-byte write
            // dS1 meets C^_i = C_i / |c_i| in the second half (sum over queries i): the image rows carry the 1 / |c_i|
            *(bf16x4*)(iS + ia) = pack4(BR == 1 ? sc[jt] * rho_x : sc[jt]);
            *(bf16x4*)((BR == 1 ? iP1 : iP2) + ia) = pack4(pm[jt]);
            if (BR == 1) sc[jt] *= rj[jt];                       // ... and C^_j in the first half (sum over keys j): 1 / |c_j| per key
        }
    }
    ABW_MARK2("attn I2.dq_or_dc_half_mfma", IT, BR);
    // branch 2: dQ^T; branch 1: the accumulator-operand half of dC^T -- both for this query tile
    {
        const bf16x8 bs = pack_col<2>(sc, 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (BR == 2) {
                const f32x4 dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(abw_tr_g<true>(st.gin, 64, 16 * ct, r, q), bs, z, 0, 0, 0);     // K block
                *(bf16x4*)(st.gout + abw_g_addr(x, (16 * ct + 4 * q) * 2)) = pack4(dq * ISQ);      // dQ block: columns 0..31
            } else {
                cy.dch[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(abw_tr_g<true>(st.gin, st.ccol, 16 * ct, r, q), bs, z, 0, 0, 0);        // C block
            }
        }
    }
}

template <int IT, int BR, int VC = 0>
__device__ __forceinline__ void abw_phase3(const AbwStep& st, AbwCarry& cy, char* scr, int r, int q) {
    constexpr float ISQ = 0.17677669529663687f;          // 1 / sqrt(32)
    const char* iS = scr + (BR == 1 ? 0 : 2048);
    const char* iP1 = scr + AbwScr<VC>::P1;
    const char* iP2 = scr + AbwScr<VC>::P2;
    const int x = 16 * IT + r;           // the KEY index now
    const char* gin = st.gin;
    char* gout = st.gout;
    const float rho_x = cy.rho_x;
    ABW_MARK2("attn I3.dv", IT, BR);
    // second half, key tile IT.  dV^T = dO^T (P1 + P2) is split between the two branch waves (16 columns each): the cosine wave also has
    // dC, the dot-product wave dK.
    {
        const bf16x8 bp1 = abw_tr_img(iP1, 16 * IT, r, q);
#pragma unroll
        for (int CV = (VC == 2 ? 0 : (BR == 1 ? 0 : 1)); CV <= (VC == 2 ? 1 : (BR == 1 ? 0 : 1)); ++CV) {
            const bf16x8 ao = abw_tr_o(st.oin, 16 * CV, r, q);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao, bp1, z, 0, 0, 0);
            if (VC == 0) dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao, abw_tr_img(iP2, 16 * IT, r, q), dv, 0, 0, 0);      // P = P1 + P2 meets in the accumulator
            *(bf16x4*)(gout + abw_g_addr(x, st.vcol + (16 * CV + 4 * q) * 2)) = pack4(dv);              // dV block
        }
    }
    ABW_MARK2("attn I3.dk_or_dc", IT, BR);
    if (BR == 2 && VC != 0) {
    } else if (BR == 2) {
        const bf16x8 bs = abw_tr_img(iS, 16 * IT, r, q);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(abw_tr_g<false>(gin, 0, 16 * ct, r, q), bs, z, 0, 0, 0);         // Q block
            *(bf16x4*)(gout + abw_g_addr(x, 64 + (16 * ct + 4 * q) * 2)) = pack4(dk * ISQ);       // dK block
        }
    } else {
        const bf16x8 bt = abw_tr_img(iS, 16 * IT, r, q);
        float dt = 0.f;
        f32x4 chv[2], dch[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const f32x4 dc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(abw_tr_g<false>(gin, st.ccol, 16 * ct, r, q), bt, cy.dch[ct], 0, 0, 0);
            const int gc = (16 * ct + 4 * q) * 2;
            dch[ct] = -dc;           // dN = -dS1
            chv[ct] = load4<bf16>((const bf16*)(gin + abw_g_addr(x, st.ccol + gc))) * rho_x;        // C^_x
            dt += (dch[ct][0] * chv[ct][0] + dch[ct][1] * chv[ct][1]) + (dch[ct][2] * chv[ct][2] + dch[ct][3] * chv[ct][3]);
        }
        dt = red_q<2>(dt, false);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
            *(bf16x4*)(gout + abw_g_addr(x, st.ccol + (16 * ct + 4 * q) * 2)) = pack4((dch[ct] - chv[ct] * dt) * rho_x);    // dC block
    }
    ABW_MARK2("attn end", IT, BR);
}

// ---- GEMM-wave LDS access (inline asm: see the header) ----------------------------------------------------------------
// four fragments by transposing 8-byte reads (lo rows, hi rows = + HI bytes), one wait at the end
template <int HI>
__device__ __forceinline__ void abw_read_frags(const uint32_t (&ad)[4], bf16x8 (&fr)[4]) {
    u32x2 t[8];
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %2, %9\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %4, %10\n\t"
        "ds_read_b64_tr_b16 %5, %10 offset:%12\n\t"
        "ds_read_b64_tr_b16 %6, %11\n\t"
        "ds_read_b64_tr_b16 %7, %11 offset:%12\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
        : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "n"(HI)
        : "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) fr[i] = __builtin_bit_cast(bf16x8, (u32x4){t[2 * i][0], t[2 * i][1], t[2 * i + 1][0], t[2 * i + 1][1]});
}
__device__ __forceinline__ void abw_read128x2(uint32_t a0, uint32_t a1, u32x4 (&out)[2]) {
    asm volatile(
        "ds_read_b128 %0, %2\n\t"
        "ds_read_b128 %1, %3\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(out[0]), "=&v"(out[1])
        : "v"(a0), "v"(a1)
        : "memory");
}

// VC (beta == 1, AttnArgs::vc_only): the Q / K halves of every tile are dead.  Attention role: see the phases.  GEMM role: the Q / K rows
// are not fetched (the DMA lanes of their chunks are masked off), dQ / dK are not copied out, and the weight gradient is dW_{v,c}[64, d] only --
// all eight waves share ITS n tiles (4 .. 7) and split the k tiles eight ways (half the MFMAs and accumulators per wave); the workgroup's
// partial carries zeros in the query / key rows, which is what autograd reports for them (pmgt/pmgt/modeling_pmgt.py:519-521).
template <int KT, bool VC = false>
__global__ __launch_bounds__(1024) void attn_bwd_wgrad_kernel(AttnBwdWg w) {
    using C = AbwCfg<KT>;
    constexpr int D = C::D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AttnArgs& a = w.a;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int H = a.H;
    const int b = blockIdx.x;
    const int h = (b >> 3) % H, xs = (b & 7) + 8 * (b / (8 * H));      // head; row slot (workgroups of one slot share an XCD)
    const int gx = gridDim.x / H;
    const int npair = (a.Tseq + 1) / 2;
    const int nsteps = xs < npair ? (npair - xs + gx - 1) / gx : 0;
    const int M = a.Tseq * 32;
    // Iterations i = 0 .. nsteps + 1, three barriers each.  Iteration i: DMA of the inputs of step i (GEMM waves), attention of
    // step i - 1 (attention waves), weight-gradient products + copy-out of step i - 2 (GEMM waves).
    // LDS visibility = every wave waits for its own LDS operations before the barrier.
#ifdef PMGT_ABW_PROF
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long plast = __builtin_readcyclecounter();
    // k = index of the work interval that ends here; the barrier wait is accounted to k + 1
    auto bar = [&](int k) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ABW_STAMP(k); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); ABW_STAMP(k + 1); };
#else
    auto bar = [](int) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
#endif
    // A SIMD holds waves {k, k + 4, k + 8, k + 12}; issue arbitration is oldest-first, and the per-wave stamps showed the younger wave
    // of each role (4-7, 12-15) taking 40 % longer per interval than its older twin -- at every barrier the workgroup waited for it.
    // Static priority for the younger half evens the two out (MI355X_MICROARCH.md, "Two waves per SIMD", item 4).
    if (VC ? (wave < 4 || wave >= 12) : (wave & 4) != 0) __builtin_amdgcn_s_setprio(1);      // (VC: the four cosine waves instead of the idle half of the attention role)
    // tile ring: step s reads its inputs from ring slot (2 s) & 3 and writes its results to slot (2 s + 1) & 3
    auto in_tile = [](int s) { return C::G0 + ((2 * s) & 3) * C::GB; };
    auto out_tile = [](int s) { return C::G0 + ((2 * s + 1) & 3) * C::GB; };

    if (wave < 8) {
        // ================================ attention role ================================
        // waves w and w + 4 share a SIMD: same (tile, branch), the two pairs of the step
        // br = 0: cosine branch (BR 1), 1: dot-product branch (BR 2).  VC: the dot-product waves only keep their dV columns, so the four
        // cosine waves -- the whole attention role then -- are spread one per SIMD (waves 0 .. 3) instead of two on SIMDs 0 and 2
        const int ul = VC ? (wave & 3) >> 1 : wave >> 2, it = VC ? wave & 1 : (wave >> 1) & 1, br = VC ? wave >> 2 : wave & 1;
        char* scr = smem + C::S0 + ul * C::SCR;
        const DropKey kd = make_drop_key(br == 0 ? a.drop1 : a.drop2);
        auto seq_of = [&](int s) { return 2 * (xs + s * gx) + ul; };
        const int role = 2 * it + br;
        auto mask_of = [&](int s) {      // raw mask value of key (lane & 31) of this wave's sequence in step s (clamped: unused past the end)
            const int t = min(seq_of(s), a.Tseq - 1);
            return a.mask ? a.mask[(int64_t)t * 32 + (lane & 31)] : 1.f;
        };
        auto step_of = [&](int s) {      // (uniform) tiles and sequence of this wave's pair in step s
            AbwStep st;
            const int t = seq_of(s);
            st.act = t < a.Tseq;
            st.t = min(t, a.Tseq - 1);
            st.zq1 = st.act && t < a.cls_only_seqs;
            st.gin = smem + in_tile(s) + ul * (32 * 256);
            st.gout = smem + out_tile(s) + ul * (32 * 256);
            st.oin = smem + C::O0 + (s & 1) * C::OB + ul * (32 * 64);
            return st;
        };
        // Pair 0 runs the three phases of step i - 1 in the three intervals of iteration i.  Pair 1 runs ONE INTERVAL LATER: phase 3 of step
        // i - 2 | phase 1 of step i - 1 | phase 2 of step i - 1.  Every tile it reads or writes allows that without a change to the ring or to
        // the GEMM role: the in-tile and dO slot of step s are overwritten by the DMA of step s + 2, issued after the FIRST barrier of
        // iteration s + 2 (pair 1's last read of them is before it); the out-tile rows 32..63 of step s are first consumed by the copy-out
        // (second interval of iteration s + 2) and by k-step 1 (third interval), both after pair 1's phase 3.  What it buys: the two attention
        // waves of a SIMD (w and w + 4: same tile and branch, the two pairs) are never both in the instruction-heavy second phase -- the
        // interval that was 2 360 of a step's 5 810 cycles with both of them in it (profiles/r04/abw_interval_stamps.txt).
        auto run = [&](auto ITc, auto BRc) __attribute__((always_inline)) {
            constexpr int IT = decltype(ITc)::value, BR = decltype(BRc)::value;
            AbwCarry cy;
            cy.rho_x = 0.f; cy.ss = 0.f;
#pragma unroll
            for (int k_ = 0; k_ < 2; ++k_) cy.sc[k_] = cy.dp[k_] = cy.dch[k_] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float mnext = mask_of(0);
#ifdef PMGT_ABW_LOCKSTEP
            const bool late = false;      // (A/B build of tools/prof: the option below as a compile-time constant)
#else
            const bool late = ul == 1 && !(a.opts & OPT_LOCKSTEP_ATTENTION_BWD);      // (uniform)
#endif
            if (!late) {
                for (int i = 0; i <= nsteps + 1; ++i) {
                    const int s = i - 1;
                    const bool on = s >= 0 && s < nsteps;
                    AbwStep st = step_of(on ? s : 0);
#ifdef PMGT_ABW_NO_ATTN
                    const bool on_ = false;      // (ablation build: the attention role only keeps the barriers; results are garbage)
#else
                    const bool on_ = on;
#endif
                    if (on_) {
                        const float mraw = mnext;
                        mnext = mask_of(s + 1);
                        abw_phase1<IT, BR, (VC ? 1 : 0)>(a, kd, st, cy, mraw, h, scr, r, q, lane);
                    }
                    bar(0);
                    if (on_) abw_phase2<IT, BR, (VC ? 1 : 0)>(a, kd, st, cy, h, scr, r, q);
                    bar(2);
                    if (on_) abw_phase3<IT, BR, (VC ? 1 : 0)>(st, cy, scr, r, q);
                    bar(4);
                }
            } else {
                for (int i = 0; i <= nsteps + 1; ++i) {
                    const int s3 = i - 2, s1 = i - 1;
#ifdef PMGT_ABW_NO_ATTN
                    if (false) {
#else
                    if (s3 >= 0 && s3 < nsteps) {
#endif
                        const AbwStep st3 = step_of(s3);
                        abw_phase3<IT, BR, (VC ? 1 : 0)>(st3, cy, scr, r, q);
                    }
                    bar(0);
#ifdef PMGT_ABW_NO_ATTN
                    const bool on = false;
#else
                    const bool on = s1 >= 0 && s1 < nsteps;
#endif
                    AbwStep st = step_of(on ? s1 : 0);
                    if (on) {
                        const float mraw = mnext;
                        mnext = mask_of(s1 + 1);
                        abw_phase1<IT, BR, (VC ? 1 : 0)>(a, kd, st, cy, mraw, h, scr, r, q, lane);
                    }
                    bar(2);
                    if (on) abw_phase2<IT, BR, (VC ? 1 : 0)>(a, kd, st, cy, h, scr, r, q);
                    bar(4);
                }
            }
        };
        if (role == 0) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        else if (role == 1) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
        else if (role == 2) run(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
#ifdef PMGT_ABW_PROF
        if (blockIdx.x == 0 && lane == 0) { for (int k_ = 0; k_ < 6; ++k_) g_abw_prof[wave][k_] = pacc[k_]; g_abw_prof[wave][7] = (unsigned long long)nsteps; }
#endif
        return;
    }
    // ==================================== GEMM role ====================================
    const int g = wave - 8, gn = g & 1, gk = g >> 1;                 // n tiles 4 gn .. 4 gn + 3, k tiles KQ gk .. KQ gk + KQ - 1
    constexpr int KQM = VC ? C::KQ / 2 : C::KQ;                      // (VC: n tiles 4 .. 7 for every wave, k tiles KQM g .. KQM g + KQM - 1)
    static_assert(KQM >= 1, "k tiles per GEMM wave");
    const int nbase = VC ? 4 : 4 * gn, kbase = VC ? KQM * g : C::KQ * gk, bsel = VC ? (g & 3) : (gk & 3);
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef __attribute__((address_space(1))) const void gbl_void_t;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
    f32x4 acc[4][KQM], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < KQM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // per-lane pieces of the fragment addresses (row part and swizzle key of the "lo" rows; "hi" rows = + 4 rows)
    const int frow = 8 * q + (r >> 2), fkey = abw_f(frow), fsub = (r & 3) >> 1, fhalf = 8 * (r & 1);
    // (fsub is bit 0 of the chunk index, fkey has bit 0 clear; tile bases are multiples of 16 KB from the start of the dynamic LDS)
    const uint32_t fla = (uint32_t)(frow * 256 + ((fkey ^ fsub) << 4) + fhalf), flb = (uint32_t)(frow * C::XROW + ((fkey ^ fsub) << 4) + fhalf);
    static_assert(C::G0 % 512 == 0 && C::GB % 512 == 0 && C::XB % 512 == 0, "tile bases must keep the row / chunk / half bit fields disjoint");
    const int hoff = a.hm ? 4 * h * 32 : h * 32, ms = a.hm ? 32 : D;
    // lane parts of the DMA source addresses (see the DMA block): x rows (lane / LPR = row inside the instruction, lane % LPR = LDS chunk
    // slot) and Q|K|V|C rows (lane >> 4, lane & 15)
    constexpr int LPRX = 64 / (1024 / C::XROW);
    const uint32_t dx_chunk = (uint32_t)((lane % LPRX) ^ abw_f(lane / LPRX)), dx_row = (uint32_t)(lane / LPRX) * (uint32_t)w.ldx * 2u;
    const uint32_t dq_chunk = (uint32_t)((lane & 15) ^ abw_f(lane >> 4)), dq_row = (uint32_t)(lane >> 4) * 4u * (uint32_t)D * 2u;
    auto kstep = [&](int s, int ks) {
        const uint32_t gb = lds0 + out_tile(s), xb = lds0 + (s & 1) * C::XB;
        // A fragments: dQKVC^T, n tiles 4 gn .. 4 gn + 3 (A[n][k = row]); B fragments: x, this wave's k tiles (B[k = row][n = x column])
        bf16x8 fa[4], fb[4];
        uint32_t aa[4], ab[4];
        // Eight fragment addresses from TWO loop-invariant lane registers: row part | swizzle key | half are disjoint bit fields of an address
        // inside a tile that starts on a multiple of its row pitch times 64, so "chunk c ^ key" is the lane register XOR (c << 4) -- one add of
        // the (uniform) tile base per operand and one XOR per fragment instead of a multiply-add, an XOR and a shift-add each (this role's
        // integer address arithmetic was two thirds of its vector instructions: profiles/r04/abw_instruction_mix.md).
        uint32_t ba = fla + gb + (uint32_t)(32 * ks * 256), bb = flb + xb + (uint32_t)(32 * ks * C::XROW);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            aa[u] = ba ^ (uint32_t)((2 * (nbase + u)) << 4);
            ab[u] = bb ^ (uint32_t)((2 * (kbase + (u % KQM))) << 4);
        }
        // All sixteen transposing reads of the k-step go out back to back (x fragments first), and the MFMAs of dQKVC fragment nt start as
        // soon as ITS two reads have landed (LDS returns in order: lgkmcnt counts down): ONE LDS round trip per k-step instead of two
        // serialised ones.  (Ablation, profiles/r04: the GEMM role alone -- no attention work at all -- takes 458 of the launch's 552 us,
        // its two k-steps 194 of them: this role, not the attention arithmetic, is the pole of the kernel.)
        u32x2 tb[8], ta[8];
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\t"
            "ds_read_b64_tr_b16 %1, %16 offset:%24\n\t"
            "ds_read_b64_tr_b16 %2, %17\n\t"
            "ds_read_b64_tr_b16 %3, %17 offset:%24\n\t"
            "ds_read_b64_tr_b16 %4, %18\n\t"
            "ds_read_b64_tr_b16 %5, %18 offset:%24\n\t"
            "ds_read_b64_tr_b16 %6, %19\n\t"
            "ds_read_b64_tr_b16 %7, %19 offset:%24\n\t"
            "ds_read_b64_tr_b16 %8, %20\n\t"
            "ds_read_b64_tr_b16 %9, %20 offset:%25\n\t"
            "ds_read_b64_tr_b16 %10, %21\n\t"
            "ds_read_b64_tr_b16 %11, %21 offset:%25\n\t"
            "ds_read_b64_tr_b16 %12, %22\n\t"
            "ds_read_b64_tr_b16 %13, %22 offset:%25\n\t"
            "ds_read_b64_tr_b16 %14, %23\n\t"
            "ds_read_b64_tr_b16 %15, %23 offset:%25"
            : "=&v"(tb[0]), "=&v"(tb[1]), "=&v"(tb[2]), "=&v"(tb[3]), "=&v"(tb[4]), "=&v"(tb[5]), "=&v"(tb[6]), "=&v"(tb[7]),
              "=&v"(ta[0]), "=&v"(ta[1]), "=&v"(ta[2]), "=&v"(ta[3]), "=&v"(ta[4]), "=&v"(ta[5]), "=&v"(ta[6]), "=&v"(ta[7])
            : "v"(ab[0]), "v"(ab[1]), "v"(ab[2]), "v"(ab[3]), "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "n"(4 * C::XROW), "n"(4 * 256)
            : "memory");
        // the x fragments and dQKVC fragment 0 have landed when at most 6 reads are outstanding
        asm volatile("s_waitcnt lgkmcnt(6)"
                     : "+v"(tb[0]), "+v"(tb[1]), "+v"(tb[2]), "+v"(tb[3]), "+v"(tb[4]), "+v"(tb[5]), "+v"(tb[6]), "+v"(tb[7]), "+v"(ta[0]), "+v"(ta[1]));
#pragma unroll
        for (int u = 0; u < 4; ++u) fb[u] = __builtin_bit_cast(bf16x8, (u32x4){tb[2 * u][0], tb[2 * u][1], tb[2 * u + 1][0], tb[2 * u + 1][1]});
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ta[2]), "+v"(ta[3]));
            if (nt == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ta[4]), "+v"(ta[5]));
            if (nt == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[6]), "+v"(ta[7]));
            fa[nt] = __builtin_bit_cast(bf16x8, (u32x4){ta[2 * nt][0], ta[2 * nt][1], ta[2 * nt + 1][0], ta[2 * nt + 1][1]});
#pragma unroll
            for (int u = 0; u < KQM; ++u) acc[nt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[nt], fb[u], acc[nt][u], 0, 0, 0);
        }
        // bias gradient = column sums of the tile: one more product against all-ones; this wave sums n tile nbase + bsel
        const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
        const bf16x8 fsel = bsel == 0 ? fa[0] : (bsel == 1 ? fa[1] : (bsel == 2 ? fa[2] : fa[3]));
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fsel, ones, accb, 0, 0, 0);
    };
    // LDS-DMA: 52 (d = 256) one-KB instructions per iteration, fixed shares per GEMM wave, straight-line code.  Per-lane address = two
    // loop-invariant registers XOR / plus wave-uniform terms (the swizzle key of a row splits into a lane part and a row-block part): ~4 VALU
    // instructions per DMA instead of ~12.
    //   x rows of step i - 1 (RPI rows per instruction): wave g takes rows 8 g ..  They are only needed when that step's results are consumed,
    //   one iteration from now, and their tile was last read in the third interval of the previous iteration: they go out FIRST in the
    //   iteration, in front of k-step 0 (round 3 issued all pieces behind the first barrier: a burst of 7 per wave in the interval that also
    //   carries the copy-out; ablation: without its DMAs the launch takes 454 us instead of 552).
    auto dma_x = [&](int i) __attribute__((always_inline)) {
        const int m0x = 64 * (xs + (i - 1) * gx);
        constexpr int RPI = 1024 / C::XROW;
        if (i >= 1 && i - 1 < nsteps) {
            char* xb = smem + ((i - 1) & 1) * C::XB;
#pragma unroll
            for (int j = 0; j < 8 / RPI; ++j) {
                const int row0 = 8 * g + RPI * j;                            // (uniform) rows row0 .. row0 + RPI - 1, lane -> row0 + lane / LPR
                const int mrow = min(m0x + row0, M - RPI);                    // (uniform) whole instruction clamped into the tensor
                // abw_f(row0 + lane / LPR) = abw_f(row0) ^ abw_f(lane / LPR): row0 is a multiple of RPI, lane / LPR < RPI <= 4 (bits 0, 1)
                const uint32_t off = (uint32_t)mrow * (uint32_t)w.ldx * 2u + dx_row + ((dx_chunk ^ (uint32_t)abw_f(row0)) << 4);
                __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)w.x + (size_t)off), (lds_void_t*)(xb + row0 * C::XROW), 16, 0, 0);
            }
        }
    };
    //   Q|K|V|C rows of the head for step i, 4 rows per instruction: wave g takes rows 8 g .. 8 g + 7; dO rows of step i, 16 per instruction:
    //   waves 0 - 3.  Rows 0..31 of these tiles belong to pair 0, whose last read of the slot's previous occupant is in the THIRD interval of
    //   the previous iteration: their pieces are EARLY (first interval, with the x rows) and have to land by the end of this iteration.  Rows
    //   32..63 belong to pair 1, which runs an interval late: their pieces are LATE (second interval) and have to land by the first barrier
    //   of the NEXT iteration.  Either way a piece has a whole step to arrive (round 3: issue behind the first barrier, wait before the
    //   third -- two thirds of a step, about the loaded HBM latency).
    auto dma_attn = [&](int i, bool early) __attribute__((always_inline)) {
        const int m0 = 64 * (xs + i * gx);
        if (i < nsteps) {
#ifdef PMGT_ABW_NO_QKVC_IN
            // (ablation build, profiles/r05: the Q|K|V|C rows are not fetched -- what a backward that RECOMPUTES them from the x tile would
            //  save on the DMA side, before paying for the recompute; results are garbage)
            if (false) {
#else
            if ((g < 4) == early) {
#endif
                char* gt = smem + in_tile(i);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row0 = 8 * g + 4 * j;
                    const int mrow = min(m0 + row0, M - 4);
                    const uint32_t c = dq_chunk ^ (uint32_t)abw_f(row0);          // source chunk of the head's 256 bytes
                    const uint32_t col = a.hm ? c * 8u : (c >> 2) * (uint32_t)D + (c & 3u) * 8u;
                    const uint32_t off = ((uint32_t)mrow * 4u * (uint32_t)D + (uint32_t)hoff + col) * 2u + dq_row;
                    // (VC: chunks 0 .. 7 of a row are its Q | K halves -- those lanes sit the instruction out)
                    if (!VC || c >= 8u)
                        __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)a.qkvc + (size_t)off), (lds_void_t*)(gt + row0 * 256), 16, 0, 0);
                }
            }
            if (g < 4 && (g < 2) == early) {
                const int row0 = 16 * g, row = row0 + (lane >> 2), c = (lane & 3) ^ abw_kt(row);
                const uint32_t m = (uint32_t)min(m0 + row, M - 1);
                const char* src = (const char*)a.dctx + (size_t)((m * (uint32_t)D + (uint32_t)(h * 32 + c * 8)) * 2u);
                __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(smem + C::O0 + (i & 1) * C::OB + row0 * 64), 16, 0, 0);
            }
        }
    };
    // vector-memory operations of this wave per iteration, in issue order: [E: early pieces] [L: x rows, late pieces] [2 copy-out stores]
#ifdef PMGT_ABW_NO_QKVC_IN
    const int attnE = (g < 2 ? 1 : 0), attnL = ((g == 2 || g == 3) ? 1 : 0);
#else
    const int attnE = (g < 4 ? 2 : 0) + (g < 2 ? 1 : 0), attnL = (g >= 4 ? 2 : 0) + ((g == 2 || g == 3) ? 1 : 0);      // (uniform)
#endif
    auto wait_vm = [](int n) __attribute__((always_inline)) {      // s_waitcnt vmcnt(n) as an immediate; n <= 9 here
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        }
    };
#ifdef PMGT_ABW_LOCKSTEP
    const bool lock = true;
#else
    const bool lock = (a.opts & OPT_LOCKSTEP_ATTENTION_BWD) != 0;      // (uniform) both pairs in the same phase: every piece is due at the END of its iteration
#endif
    int prev_tail = 0;       // operations issued BEHIND the late pieces in the previous iteration (its full tile's 2 stores), -1 = do not count on any
    for (int i = 0; i <= nsteps + 1; ++i) {
        const int sg = i - 2;                     // the step whose results are consumed in this iteration
        ABW_MARK("gemm A.dma_early+kstep0");
        int nE = 0;
#ifndef PMGT_ABW_NO_DMA
        dma_attn(i, true);
        nE = i < nsteps ? attnE : 0;
#endif
#ifndef PMGT_ABW_NO_GEMM
        if (sg >= 0) kstep(sg, 0);
#endif
        // the LATE pieces of the previous iteration (pair 1's rows of this iteration's attention inputs) have landed for this wave: vmcnt retires in
        // issue order, and only that iteration's copy-out stores and this iteration's early pieces were issued behind them (a smaller count
        // than the true one only waits longer)
        if (!lock) wait_vm(prev_tail >= 0 ? min(prev_tail + nE, 9) : 0);
        bar(0);
        ABW_MARK("gemm B.copy_out+dma_late");
        // copy-out of the finished dQ|dK|dV|dC tile (64 rows x 16 chunks of 16 bytes, 2 per lane): its two LDS reads go out first and
        // travel under the issue of the late DMA pieces; the stores follow
        u32x4 v[2];
        bool copy = false;
#if !defined(PMGT_ABW_NO_COPY) && !defined(PMGT_ABW_NO_GEMM)
        copy = sg >= 0;
#endif
        int lc = lane;
        asm volatile("" : "+v"(lc));
        if (copy) {
            const uint32_t gb = lds0 + out_tile(sg);
            uint32_t ad[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int idx = 64 * (8 * p + g) + lc, row = idx >> 4, c = idx & 15;
                ad[p] = gb + row * 256 + ((c ^ abw_f(row)) << 4);
            }
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(v[0]), "=&v"(v[1]) : "v"(ad[0]), "v"(ad[1]) : "memory");
        }
        // (the x rows go out HERE, not with the early pieces: stamps of the first form showed the GEMM waves at 1 970 cycles in the first
        // interval -- seven DMA issues in front of k-step 0 -- against 630 - 860 in this one)
        int nL = 0;
#ifndef PMGT_ABW_NO_DMA
        dma_x(i);                       // issued BEFORE the late pieces: the wait that closes the iteration lets only those (and the stores) stay in flight
        dma_attn(i, false);
        nL = i < nsteps ? attnL : 0;
#endif
        const bool full = copy && 64 * (xs + sg * gx) + 64 <= M;      // (uniform) no store instruction skipped by an all-false row predicate
        if (copy) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]) :: "memory");
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int idx = 64 * (8 * p + g) + lc, row = idx >> 4, c = idx & 15;
                const int m = 64 * (xs + sg * gx) + row;
                if (m < M && (!VC || c >= 8)) *(u32x4*)((char*)a.dqkvc + (size_t)(((uint32_t)m * 4u * D + (uint32_t)(hoff + (c >> 2) * ms + (c & 3) * 8)) * 2u)) = v[p];
            }
        }
        bar(2);
        ABW_MARK("gemm C.kstep1");
#ifndef PMGT_ABW_NO_GEMM
        if (sg >= 0) kstep(sg, 1);
#endif
        // This iteration's EARLY pieces have landed (x rows: needed an iteration from now; pair 0's rows: at the next barrier).  Behind them
        // in issue order: the late pieces and, for a full tile, the two copy-out stores -- they stay in flight (waiting for the stores'
        // acknowledgement as well put a store round trip on every iteration's critical path).
#ifndef PMGT_ABW_WAIT_STORES
        if (full) wait_vm(lock ? 2 : nL + 2);
        else if (!copy) wait_vm(lock ? 0 : nL);
        else
#endif
        wait_vm(0);
        prev_tail = full ? 2 : (!copy ? 0 : -1);
#ifdef PMGT_ABW_WAIT_STORES
        prev_tail = -1;
#endif
        bar(4);
        ABW_MARK("gemm loop_end");
    }
#ifdef PMGT_ABW_PROF
    if (blockIdx.x == 0 && lane == 0) { for (int k_ = 0; k_ < 6; ++k_) g_abw_prof[wave][k_] = pacc[k_]; g_abw_prof[wave][7] = (unsigned long long)nsteps; }
#endif
    // ---- the workgroup's partial of dW_head (and db_head): rows of W_qkvc = matrix * d + head * 32 + w
    float* slab = w.slab + (int64_t)xs * 4 * D * D;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = nbase + nt;                               // 16-row block of the head's 128 rows: matrix n >> 1, half n & 1
        const int wrow = (n >> 1) * D + h * 32 + 16 * (n & 1) + 4 * q;
#pragma unroll
        for (int j = 0; j < KQM; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) slab[(int64_t)(wrow + e) * D + 16 * (kbase + j) + r] = acc[nt][j][e];
    }
    if (r == 0 && w.bias_slab && (!VC || g < 4)) {
        const int n = nbase + bsel;
        const int wrow = (n >> 1) * D + h * 32 + 16 * (n & 1) + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) w.bias_slab[(int64_t)xs * 4 * D + wrow + e] = accb[e];
    }
    if constexpr (VC) {      // the query / key rows of the head: exact zeros (wave g: rows 8 g .. 8 g + 7 of the 64)
        for (int i = 0; i < 8; ++i) {
            const int rr = 8 * g + i, wrow = (rr >> 5) * D + h * 32 + (rr & 31);
            for (int cc = lane * 4; cc < D; cc += 256) *(f32x4*)(slab + (int64_t)wrow * D + cc) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (g == 0 && w.bias_slab) w.bias_slab[(int64_t)xs * 4 * D + (lane >> 5) * D + h * 32 + (lane & 31)] = 0.f;
    }
}

// =================================================================================================
// beta == 1, TWO HEADS PER STEP (round 5).  The one-head vc_only form above does half the work per step and still takes 2.2 us per step:
// three barrier intervals of latency are the floor of a step, whatever is in it.  Here a workgroup owns a head PAIR: the tile row of a
// step is [V0 C0 V1 C1] -- the general kernel's 256 bytes per row, so its GEMM role runs unchanged (n tiles 0 .. 7, KQ k tiles per wave:
// dW_{v,c} of two heads = the general accumulator footprint) -- and the eight attention waves are eight cosine waves, one per (pair,
// head, tile), each with all 32 columns of its dV.  Half the steps per launch.  LDS: the x tile is SINGLE-buffered -- its two row halves are
// consumed by k-step 0 (first interval) and k-step 1 (third interval) of the following iteration, so rows 0 .. 31 of the next tile are requested
// behind the first barrier (as before) and rows 32 .. 63 at the top of the next iteration, behind the last reader of their predecessors:
// x 32 KB | tile ring 4 x 16 KB | dO 4 x 4 KB (two heads, two steps) | scratch 4 x 4.25 KB = 129 KB.
// The weight-gradient partial of a workgroup is [2 d, d] (value | ctx_attention rows of its two heads); the caller zero-fills query / key.
// =================================================================================================
template <int KT> struct AbwCfg2 {
    static constexpr int D = 16 * KT, XROW = D * 2, XB = 64 * XROW, GB = 64 * 256, OB = 64 * 64;
    static constexpr int SCR = 2 * 2048 + 256;        // per (pair, head): dS1, P1 images | rho, madd
    static constexpr int G0 = XB, O0 = G0 + 4 * GB, S0 = O0 + 4 * OB, SMEM = S0 + 4 * SCR;
    static constexpr int KQ = KT / 4;
};

template <int KT>
__global__ __launch_bounds__(1024) void attn_bwd_wgrad_vc2_kernel(AttnBwdWg w) {
    using C = AbwCfg2<KT>;
    constexpr int D = C::D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AttnArgs& a = w.a;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int H = a.H, HP = H / 2;
    const int b = blockIdx.x;
    const int hp = (b >> 3) % HP, xs = (b & 7) + 8 * (b / (8 * HP));      // head pair; row slot (workgroups of one slot share an XCD)
    const int gx = gridDim.x / HP;
    const int npair = (a.Tseq + 1) / 2;
    const int nsteps = xs < npair ? (npair - xs + gx - 1) / gx : 0;
    const int M = a.Tseq * 32;
    auto bar = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
    if (wave & 4) __builtin_amdgcn_s_setprio(1);
    auto in_tile = [](int s) { return C::G0 + ((2 * s) & 3) * C::GB; };
    auto out_tile = [](int s) { return C::G0 + ((2 * s + 1) & 3) * C::GB; };

    if (wave < 8) {
        // ================================ attention role: eight cosine waves ================================
        const int ul = wave >> 2, uh = (wave >> 1) & 1, it = wave & 1;         // waves w and w + 4 share a SIMD: same (head, tile), the two pairs
        const int h = 2 * hp + uh;
        char* scr = smem + C::S0 + (2 * ul + uh) * C::SCR;
        const DropKey kd = make_drop_key(a.drop1);
        auto seq_of = [&](int s) { return 2 * (xs + s * gx) + ul; };
        auto mask_of = [&](int s) {
            const int t = min(seq_of(s), a.Tseq - 1);
            return a.mask ? a.mask[(int64_t)t * 32 + (lane & 31)] : 1.f;
        };
        auto step_of = [&](int s) {
            AbwStep st;
            const int t = seq_of(s);
            st.act = t < a.Tseq;
            st.t = min(t, a.Tseq - 1);
            st.zq1 = st.act && t < a.cls_only_seqs;
            st.gin = smem + in_tile(s) + ul * (32 * 256);
            st.gout = smem + out_tile(s) + ul * (32 * 256);
            st.oin = smem + C::O0 + (2 * (s & 1) + uh) * C::OB + ul * (32 * 64);
            st.vcol = uh * 128; st.ccol = uh * 128 + 64;
            return st;
        };
        auto run = [&](auto ITc) __attribute__((always_inline)) {
            constexpr int IT = decltype(ITc)::value;
            AbwCarry cy;
            cy.rho_x = 0.f; cy.ss = 0.f;
#pragma unroll
            for (int k_ = 0; k_ < 2; ++k_) cy.sc[k_] = cy.dp[k_] = cy.dch[k_] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float mnext = mask_of(0);
            if (ul == 0) {
                for (int i = 0; i <= nsteps + 1; ++i) {
                    const int s = i - 1;
                    const bool on = s >= 0 && s < nsteps;
                    AbwStep st = step_of(on ? s : 0);
                    if (on) {
                        const float mraw = mnext;
                        mnext = mask_of(s + 1);
                        abw_phase1<IT, 1, 2>(a, kd, st, cy, mraw, h, scr, r, q, lane);
                    }
                    bar();
                    if (on) abw_phase2<IT, 1, 2>(a, kd, st, cy, h, scr, r, q);
                    bar();
                    if (on) abw_phase3<IT, 1, 2>(st, cy, scr, r, q);
                    bar();
                }
            } else {          // pair 1: one interval late (see the one-head kernel)
                for (int i = 0; i <= nsteps + 1; ++i) {
                    const int s3 = i - 2, s1 = i - 1;
                    if (s3 >= 0 && s3 < nsteps) {
                        const AbwStep st3 = step_of(s3);
                        abw_phase3<IT, 1, 2>(st3, cy, scr, r, q);
                    }
                    bar();
                    const bool on = s1 >= 0 && s1 < nsteps;
                    AbwStep st = step_of(on ? s1 : 0);
                    if (on) {
                        const float mraw = mnext;
                        mnext = mask_of(s1 + 1);
                        abw_phase1<IT, 1, 2>(a, kd, st, cy, mraw, h, scr, r, q, lane);
                    }
                    bar();
                    if (on) abw_phase2<IT, 1, 2>(a, kd, st, cy, h, scr, r, q);
                    bar();
                }
            }
        };
        if (it == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 1>{});
        return;
    }
    // ==================================== GEMM role (the general kernel's, on the [V0 C0 V1 C1] tile) ====================================
    const int g = wave - 8, gn = g & 1, gk = g >> 1;                 // n tiles 4 gn .. 4 gn + 3 (= head gn), k tiles KQ gk .. KQ gk + KQ - 1
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef __attribute__((address_space(1))) const void gbl_void_t;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
    f32x4 acc[4][C::KQ], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < C::KQ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = 8 * q + (r >> 2), fkey = abw_f(frow), fsub = (r & 3) >> 1, fhalf = 8 * (r & 1);
    const uint32_t fla = (uint32_t)(frow * 256 + ((fkey ^ fsub) << 4) + fhalf), flb = (uint32_t)(frow * C::XROW + ((fkey ^ fsub) << 4) + fhalf);
    static_assert(C::G0 % 512 == 0 && C::GB % 512 == 0 && C::XB % 512 == 0, "tile bases must keep the row / chunk / half bit fields disjoint");
    constexpr int LPRX = 64 / (1024 / C::XROW);
    const uint32_t dx_chunk = (uint32_t)((lane % LPRX) ^ abw_f(lane / LPRX)), dx_row = (uint32_t)(lane / LPRX) * (uint32_t)w.ldx * 2u;
    const uint32_t dq_chunk = (uint32_t)((lane & 15) ^ abw_f(lane >> 4)), dq_row = (uint32_t)(lane >> 4) * 4u * (uint32_t)D * 2u;
    // tile chunk c (16 bytes, 0 .. 15) of a row = (head 2 hp + (c >> 3), matrix v | c = 2 + ((c >> 2) & 1), elements 8 (c & 3) ..): its element column
    // inside a Q|K|V|C row
    auto gcol = [&](uint32_t c) {
        const uint32_t hd = 2u * (uint32_t)hp + (c >> 3), mt = 2u + ((c >> 2) & 1u);
        return a.hm ? (hd * 4u + mt) * 32u + (c & 3u) * 8u : mt * (uint32_t)D + hd * 32u + (c & 3u) * 8u;
    };
    auto kstep = [&](int s, int ks) {
        const uint32_t gb = lds0 + out_tile(s), xb = lds0;
        bf16x8 fa[4], fb[4];
        uint32_t aa[4], ab[4];
        uint32_t ba = fla + gb + (uint32_t)(32 * ks * 256), bb = flb + xb + (uint32_t)(32 * ks * C::XROW);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            aa[u] = ba ^ (uint32_t)((2 * (4 * gn + u)) << 4);
            ab[u] = bb ^ (uint32_t)((2 * (C::KQ * gk + (u % C::KQ))) << 4);
        }
        u32x2 tb[8], ta[8];
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\t"
            "ds_read_b64_tr_b16 %1, %16 offset:%24\n\t"
            "ds_read_b64_tr_b16 %2, %17\n\t"
            "ds_read_b64_tr_b16 %3, %17 offset:%24\n\t"
            "ds_read_b64_tr_b16 %4, %18\n\t"
            "ds_read_b64_tr_b16 %5, %18 offset:%24\n\t"
            "ds_read_b64_tr_b16 %6, %19\n\t"
            "ds_read_b64_tr_b16 %7, %19 offset:%24\n\t"
            "ds_read_b64_tr_b16 %8, %20\n\t"
            "ds_read_b64_tr_b16 %9, %20 offset:%25\n\t"
            "ds_read_b64_tr_b16 %10, %21\n\t"
            "ds_read_b64_tr_b16 %11, %21 offset:%25\n\t"
            "ds_read_b64_tr_b16 %12, %22\n\t"
            "ds_read_b64_tr_b16 %13, %22 offset:%25\n\t"
            "ds_read_b64_tr_b16 %14, %23\n\t"
            "ds_read_b64_tr_b16 %15, %23 offset:%25"
            : "=&v"(tb[0]), "=&v"(tb[1]), "=&v"(tb[2]), "=&v"(tb[3]), "=&v"(tb[4]), "=&v"(tb[5]), "=&v"(tb[6]), "=&v"(tb[7]),
              "=&v"(ta[0]), "=&v"(ta[1]), "=&v"(ta[2]), "=&v"(ta[3]), "=&v"(ta[4]), "=&v"(ta[5]), "=&v"(ta[6]), "=&v"(ta[7])
            : "v"(ab[0]), "v"(ab[1]), "v"(ab[2]), "v"(ab[3]), "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "n"(4 * C::XROW), "n"(4 * 256)
            : "memory");
        asm volatile("s_waitcnt lgkmcnt(6)"
                     : "+v"(tb[0]), "+v"(tb[1]), "+v"(tb[2]), "+v"(tb[3]), "+v"(tb[4]), "+v"(tb[5]), "+v"(tb[6]), "+v"(tb[7]), "+v"(ta[0]), "+v"(ta[1]));
#pragma unroll
        for (int u = 0; u < 4; ++u) fb[u] = __builtin_bit_cast(bf16x8, (u32x4){tb[2 * u][0], tb[2 * u][1], tb[2 * u + 1][0], tb[2 * u + 1][1]});
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ta[2]), "+v"(ta[3]));
            if (nt == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ta[4]), "+v"(ta[5]));
            if (nt == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[6]), "+v"(ta[7]));
            fa[nt] = __builtin_bit_cast(bf16x8, (u32x4){ta[2 * nt][0], ta[2 * nt][1], ta[2 * nt + 1][0], ta[2 * nt + 1][1]});
#pragma unroll
            for (int u = 0; u < C::KQ; ++u) acc[nt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[nt], fb[u], acc[nt][u], 0, 0, 0);
        }
        const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
        const bf16x8 fsel = (gk & 3) == 0 ? fa[0] : ((gk & 3) == 1 ? fa[1] : ((gk & 3) == 2 ? fa[2] : fa[3]));
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fsel, ones, accb, 0, 0, 0);
    };
    // x rows of step sx into the (single) x tile: wave g owns rows 8 g .. 8 g + 7, i.e. waves 0 - 3 the half k-step 0 reads, waves 4 - 7 the half
    // k-step 1 reads
    constexpr int RPI = 1024 / C::XROW, NX = 8 / RPI;                 // rows per instruction, instructions per wave
    auto dma_x = [&](int sx) __attribute__((always_inline)) {
        const int m0x = 64 * (xs + sx * gx);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int row0 = 8 * g + RPI * j;
            const int mrow = min(m0x + row0, M - RPI);
            const uint32_t off = (uint32_t)mrow * (uint32_t)w.ldx * 2u + dx_row + ((dx_chunk ^ (uint32_t)abw_f(row0)) << 4);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)w.x + (size_t)off), (lds_void_t*)(smem + row0 * C::XROW), 16, 0, 0);
        }
    };
    // V | C rows of the two heads for step i (4 rows per instruction: wave g takes rows 8 g .. 8 g + 7) and dO rows (16 per instruction: wave g
    // takes head g >> 2, rows 16 (g & 3) ..).  Rows 0 .. 31 (pair 0) are EARLY pieces, rows 32 .. 63 (pair 1, an interval late) LATE ones.
    auto dma_attn = [&](int i, bool early) __attribute__((always_inline)) {
        const int m0 = 64 * (xs + i * gx);
        if ((g < 4) == early) {
            char* gt = smem + in_tile(i);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row0 = 8 * g + 4 * j;
                const int mrow = min(m0 + row0, M - 4);
                const uint32_t c = dq_chunk ^ (uint32_t)abw_f(row0);
                const uint32_t off = ((uint32_t)mrow * 4u * (uint32_t)D + gcol(c)) * 2u + dq_row;
                __builtin_amdgcn_global_load_lds((gbl_void_t*)((const char*)a.qkvc + (size_t)off), (lds_void_t*)(gt + row0 * 256), 16, 0, 0);
            }
        }
        if (((g & 3) < 2) == early) {
            const int uh = g >> 2, row0 = 16 * (g & 3), row = row0 + (lane >> 2), c = (lane & 3) ^ abw_kt(row);
            const uint32_t m = (uint32_t)min(m0 + row, M - 1);
            const char* src = (const char*)a.dctx + (size_t)((m * (uint32_t)D + (uint32_t)((2 * hp + uh) * 32 + c * 8)) * 2u);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(smem + C::O0 + (2 * (i & 1) + uh) * C::OB + row0 * 64), 16, 0, 0);
        }
    };
    const int attnE = (g < 4 ? 2 : 0) + ((g & 3) < 2 ? 1 : 0), attnL = (g >= 4 ? 2 : 0) + ((g & 3) >= 2 ? 1 : 0);      // (uniform)
    auto wait_vm = [](int n) __attribute__((always_inline)) {
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        }
    };
    // Vector-memory operations of this wave per iteration, in issue order:
    //   [top]  Xhi (waves 4 - 7: NX instructions, rows 32 .. 63 of x(i - 2))   E (early pieces of step i)
    //   [behind barrier 0]  Xlo (waves 0 - 3: NX instructions, rows 0 .. 31 of x(i - 1))   L (late pieces of step i)   2 copy-out stores
    // vmcnt retires in issue order, so "everything up to X has landed" = at most (operations issued behind X) outstanding; a smaller count
    // only waits longer.  Deadlines: L of iteration i - 1 by barrier 0 of iteration i; Xhi by barrier 1 (k-step 1 reads it in the third interval);
    // E and Xlo by barrier 2 (pair 0's first phase / k-step 0 of the next iteration).
    int prev_tail = 0;       // operations issued BEHIND the late pieces in the previous iteration (its full tile's 2 stores), -1 = do not count on any
    for (int i = 0; i <= nsteps + 1; ++i) {
        const int sg = i - 2;                     // the step whose results are consumed in this iteration
        const bool xhi = g >= 4 && sg >= 0 && sg < nsteps, xlo = g < 4 && i >= 1 && i - 1 < nsteps, at = i < nsteps;      // (uniform)
        if (xhi) dma_x(sg);
        if (at) dma_attn(i, true);
        const int nXhi = xhi ? NX : 0, nE = at ? attnE : 0, nXlo = xlo ? NX : 0, nL = at ? attnL : 0;
        if (sg >= 0) kstep(sg, 0);
        wait_vm(prev_tail >= 0 ? min(prev_tail + nXhi + nE, 9) : 0);      // the previous iteration's late pieces have landed (its stores may stay in flight)
        bar();
        u32x4 v[2];
        const bool copy = sg >= 0;
        int lc = lane;
        asm volatile("" : "+v"(lc));
        if (copy) {
            const uint32_t gb = lds0 + out_tile(sg);
            uint32_t ad[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int idx = 64 * (8 * p + g) + lc, row = idx >> 4, c = idx & 15;
                ad[p] = gb + row * 256 + ((c ^ abw_f(row)) << 4);
            }
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(v[0]), "=&v"(v[1]) : "v"(ad[0]), "v"(ad[1]) : "memory");
        }
        if (xlo) dma_x(i - 1);
        if (at) dma_attn(i, false);
        const bool full = copy && 64 * (xs + sg * gx) + 64 <= M;      // (uniform) no store instruction skipped by an all-false row predicate
        if (copy) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]) :: "memory");
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int idx = 64 * (8 * p + g) + lc, row = idx >> 4, c = idx & 15;
                const int m = 64 * (xs + sg * gx) + row;
                if (m < M) *(u32x4*)((char*)a.dqkvc + (size_t)(((uint32_t)m * 4u * D + gcol((uint32_t)c)) * 2u)) = v[p];
            }
        }
        // Xhi of this iteration has landed (k-step 1 below reads it): behind it in issue order E, Xlo, L and the stores
        if (full) wait_vm(min(nE + nXlo + nL + 2, 9));
        else if (!copy) wait_vm(min(nE + nXlo + nL, 9));
        else wait_vm(0);
        bar();
        if (sg >= 0) kstep(sg, 1);
        // E and Xlo have landed: behind them L and the stores
        if (full) wait_vm(nL + 2);
        else if (!copy) wait_vm(nL);
        else wait_vm(0);
        prev_tail = full ? 2 : (!copy ? 0 : -1);
        bar();
    }
    // ---- the workgroup's partial of dW_{v,c} (and the bias sums) of its two heads: rows = matrix (0 = value, 1 = ctx_attention) * d + head * 32 + w
    float* slab = w.slab + (int64_t)xs * 2 * D * D;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = 4 * gn + nt;                              // 16-row block of the tile's 128 columns: head n >> 2, matrix (n >> 1) & 1, half n & 1
        const int wrow = ((n >> 1) & 1) * D + (2 * hp + (n >> 2)) * 32 + 16 * (n & 1) + 4 * q;
#pragma unroll
        for (int j = 0; j < C::KQ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) slab[(int64_t)(wrow + e) * D + 16 * (C::KQ * gk + j) + r] = acc[nt][j][e];
    }
    if (r == 0 && w.bias_slab) {
        const int n = 4 * gn + (gk & 3);
        const int wrow = ((n >> 1) & 1) * D + (2 * hp + (n >> 2)) * 32 + 16 * (n & 1) + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) w.bias_slab[(int64_t)xs * 2 * D + wrow + e] = accb[e];
    }
}


int attn_bwd_wgrad_parts(int H) { return std::max(8, (256 / std::max(H, 1)) / 8 * 8); }
// two-heads-per-step form of the beta == 1 mode: row slots = partial slabs of [2 d, d] (value | ctx_attention rows; twice as many, half as large)
int attn_bwd_wgrad_vc2_parts(int H) { return std::max(8, (256 / std::max(H / 2, 1)) / 8 * 8); }
bool attn_bwd_wgrad_vc2_supported(const AttnBwdWg& w) {
    return w.a.vc_only && w.a.beta == 1.f && w.a.H % 2 == 0 && !(w.a.opts & OPT_NO_VC2_ATTENTION_BWD) && attn_bwd_wgrad_supported(w);
}
template <int KT> static int launch_abw_vc2(const AttnBwdWg& w, hipStream_t st) {
    using C = AbwCfg2<KT>;
    auto kern = attn_bwd_wgrad_vc2_kernel<KT>;
    PMGT_SMEM_ATTR((const void*)kern, C::SMEM);
    const int gx = attn_bwd_wgrad_vc2_parts(w.a.H);
    note_launch(LT_ATTN_BWD_WGRAD);
    note_launch(LT_ATTN_BWD_WGRAD_VC);
    note_launch(LT_ATTN_BWD_WGRAD_VC2);
    hipLaunchKernelGGL(kern, dim3(gx * (w.a.H / 2)), dim3(1024), C::SMEM, st, w);
    PMGT_LAUNCH_OK();
    return 0;
}
int attn_bwd_wgrad_vc2(const AttnBwdWg& w, hipStream_t st) {
    PMGT_CHECK(attn_bwd_wgrad_vc2_supported(w), -2, "attn_bwd_wgrad_vc2: unsupported shape / mode S=%d dh=%d H=%d beta=%g", w.a.S, w.a.dh, w.a.H, (double)w.a.beta);
    return w.a.H * 32 == 256 ? launch_abw_vc2<16>(w, st) : launch_abw_vc2<8>(w, st);
}

// the part of attn_bwd_wgrad_supported() that is a function of the shape alone: the forward's beta == 1 decision (engine.hip vc_only_applies) asks it
// BEFORE leaving Q / K unwritten, so that forward and backward cannot disagree (a shape past the offset bound takes the general kernels in both)
bool attn_bwd_wgrad_shape_ok(int Tseq, int S, int dh, int H) {
    const int d = H * 32;
    if ((int64_t)Tseq * 32 * 4 * d * 2 >= (int64_t)1 << 32) return false;      // 32-bit byte offsets inside Q|K|V|C (the saddr addressing form)
    return S == 32 && dh == 32 && (d == 256 || d == 128) && Tseq >= 2;
}

bool attn_bwd_wgrad_supported(const AttnBwdWg& w) {
    const AttnArgs& a = w.a;
    return attn_bwd_wgrad_shape_ok(a.Tseq, a.S, a.dh, a.H) && a.cls_only_seqs >= 0 && w.x != nullptr && w.ldx % 8 == 0 &&
           w.slab != nullptr && a.qkvc != nullptr && a.dctx != nullptr && a.dqkvc != nullptr && ((uintptr_t)w.x % 16) == 0 &&
           ((uintptr_t)a.qkvc % 16) == 0 && ((uintptr_t)a.dctx % 16) == 0 && ((uintptr_t)a.dqkvc % 16) == 0;
}

template <int KT, bool VC = false> static int launch_abw(const AttnBwdWg& w, hipStream_t st) {
    using C = AbwCfg<KT>;
    auto kern = attn_bwd_wgrad_kernel<KT, VC>;
    PMGT_SMEM_ATTR((const void*)kern, C::SMEM);
    const int gx = attn_bwd_wgrad_parts(w.a.H);
    note_launch(LT_ATTN_BWD_WGRAD);
    if (VC) note_launch(LT_ATTN_BWD_WGRAD_VC);
    hipLaunchKernelGGL(kern, dim3(gx * w.a.H), dim3(1024), C::SMEM, st, w);
    PMGT_LAUNCH_OK();
    return 0;
}

int attn_bwd_wgrad(const AttnBwdWg& w, hipStream_t st) {
    PMGT_CHECK(attn_bwd_wgrad_supported(w), -2, "attn_bwd_wgrad: unsupported shape S=%d dh=%d H=%d", w.a.S, w.a.dh, w.a.H);
    if (w.a.vc_only) {
        PMGT_CHECK(w.a.beta == 1.f, -2, "attn_bwd_wgrad: vc_only is the beta == 1 form (beta = %g)", (double)w.a.beta);
        return w.a.H * 32 == 256 ? launch_abw<16, true>(w, st) : launch_abw<8, true>(w, st);
    }
    return w.a.H * 32 == 256 ? launch_abw<16>(w, st) : launch_abw<8>(w, st);
}

bool attn_mfma_supported(const AttnArgs& a) { return a.S >= 1 && a.S <= 64 && (a.dh == 32 || a.dh == 64); }

int attn_mfma(const AttnArgs& a, bool bwd, hipStream_t st) {
    if (a.Tseq <= 0) return 0;
    if (a.dh == 32) return launch_nt<32>(a, bwd, st);
    return launch_nt<64>(a, bwd, st);
}

}  // namespace pmgt

#ifdef PMGT_ABW_PROF
extern "C" int pmgt_debug_abw_prof_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmgt::g_abw_prof), sizeof(pmgt::g_abw_prof));
}
#endif
#ifdef PMGT_COOP_PROF
extern "C" int pmgt_debug_coop_prof_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmgt::g_coop_prof), sizeof(pmgt::g_coop_prof));
}
#endif
#ifdef PMGT_AB_PROF
extern "C" int pmgt_debug_ab_prof_read(unsigned int* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmgt::g_ab_prof), sizeof(pmgt::g_ab_prof));
}
#endif
