// Fused forward of one PMGT layer's front half (pmgt/pmgt/modeling_pmgt.py:420-534):
//
//     Q | K | V | C = x W_{q,k,v,c}^T + b      (written to HBM once, for the backward pass)
//     ctx           = (beta softmax(1 - C^ C^T + I + mask) + (1 - beta) softmax(Q K^T / sqrt(dh) + mask)) V
//
// for the headline shape S = 32, head size 32, hidden size 256 (or 128), bf16.  The unfused pair (streaming GEMM +
// attention kernel) writes Q|K|V|C (4 x [M, d]) and immediately reads it back; here the attention of a tile runs
// from the LDS copy of the projection output, so that read never reaches HBM.
//
// One 512-thread workgroup per CU, weight-stationary like gemm_ws.hip: a column slab is the 8 x 32 columns
// {Q, K, V, C} x {head 2y, head 2y+1}; wave w = (matrix w >> 1, head 2y + (w & 1)) keeps its 32 x K block of W in
// registers and streams 64-row tiles of x (= two sequences).  Per tile:
//   global -> registers (two tiles in flight) -> XOR-swizzled LDS A tile -> 64 MFMAs per wave with the operands
//   swapped (D = W_frag x x_frag^T, so a lane ends up with 4 CONSECUTIVE output columns of one row) -> + bias,
//   bf16, 8-byte LDS writes into the swizzled [64][256] projection tile -> barrier ->
//   (a) all threads: 16-byte row-contiguous copies of the tile to HBM (columns un-permuted to q | k | v | c);
//   (b) wave w: attention of (sequence w >> 2, head w >> 1 & 1, query half w & 1) straight from the LDS tile,
//       same arithmetic as attention_mfma.hip (transposed scores in the MFMA C/D layout, register softmax,
//       accumulator tile reused as the B operand of P V), context written to HBM.
#include <type_traits>

#include "attention.h"
#include "fp8.h"
#include "gemm.h"

namespace pmgt {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

#ifdef PMGT_QA_PROF
// cycles (s_memtime) per phase of the tile steps of two workgroups, accumulated in SGPRs: [block slot][wave][phase];
// phase 7 = number of steps
__device__ unsigned int g_qa_prof[2][8][8];
// per workgroup: start / end time (s_memrealtime, 100 MHz), HW_ID, XCC_ID
__device__ unsigned long long g_qa_blk[1024][4];
#define QA_STAMP(k)                                                   \
    do {                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        pacc[k] += (unsigned int)(now_ - plast);                      \
        plast = now_;                                                 \
    } while (0)
#define QA_PROF_DECL                                                                                        \
    const unsigned long long pstart = __builtin_amdgcn_s_memrealtime();                                     \
    const int pslot = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 + 3 ? 1 : -1);                     \
    unsigned int pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                                        \
    unsigned long long plast = __builtin_readcyclecounter();
#define QA_PROF_FLUSH                                                                                       \
    do {                                                                                                    \
        if (lane == 0 && pslot >= 0) {                                                                      \
            for (int k_ = 0; k_ < 8; ++k_) g_qa_prof[pslot][wave][k_] = pacc[k_];                           \
        }                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                        \
            g_qa_blk[blockIdx.x][0] = pstart;                                                               \
            g_qa_blk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();                                     \
            g_qa_blk[blockIdx.x][2] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);                        \
            g_qa_blk[blockIdx.x][3] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 20);                       \
        }                                                                                                   \
    } while (0)
#else
#define QA_STAMP(k) do { } while (0)
#define QA_PROF_DECL
#define QA_PROF_FLUSH do { } while (0)
#endif

#ifdef PMGT_QA_PROF
// role-split kernel: cycles per interval of workgroup 0, [wave][interval]; [wave][7] = iterations
__device__ unsigned int g_qa3_prof[16][8];
#define QA3_PROF_DECL unsigned int pacc3[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long plast3 = __builtin_readcyclecounter();
#define QA3_STAMP(k)                                                   \
    do {                                                               \
        const unsigned long long now_ = __builtin_readcyclecounter();  \
        pacc3[k] += (unsigned int)(now_ - plast3);                     \
        plast3 = now_;                                                 \
    } while (0)
#define QA3_PROF_FLUSH                                                                              \
    do {                                                                                            \
        if (blockIdx.x == 0 && lane == 0) {                                                         \
            for (int k_ = 0; k_ < 7; ++k_) g_qa3_prof[wave][k_] = pacc3[k_];                        \
            g_qa3_prof[wave][7] = (unsigned int)NI;                                                 \
        }                                                                                           \
    } while (0)
#else
#define QA3_PROF_DECL
#define QA3_STAMP(k) do { } while (0)
#define QA3_PROF_FLUSH do { } while (0)
#endif


// reduction over the 4 lanes l, l^16, l^32, l^48 on the LDS crossbar (ds_swizzle xor 16 inside the 32-lane halves, ds_bpermute for
// l ^ 32): two LDS-port instructions and two VALU instructions.  The v_permlane16/32_swap form cost 8 VALU instructions (each swap needs
// two copies of the value) plus two s_nop; the kernels here are VALU-issue-bound and run this six times per attention problem.
__device__ __forceinline__ float qred(float v, bool is_max) {
    float o = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));      // lane ^ 16
    v = is_max ? raw_max(v, o) : v + o;
    o = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)((threadIdx.x & 63) ^ 32) << 2, __builtin_bit_cast(int, v)));
    return is_max ? raw_max(v, o) : v + o;
}

// byte address of 16-byte chunk `ch` (0..31) of row `row` inside the swizzled projection tile
__device__ __forceinline__ int qt_addr(int row, int ch) { return row * 512 + ((ch ^ (row & 15)) << 4); }

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));


// max over the whole wave (every lane gets it): DPP row rotations + the two permlane swaps
__device__ __forceinline__ float qa_wave_max(float v) {
    v = raw_max(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false)));
    v = raw_max(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false)));
    v = raw_max(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false)));
    v = raw_max(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));
    return qred(v, true);
}

// Attention of ONE (sequence, head, 16-query tile) by one wave, straight from the swizzled projection tile in LDS
// (rows R0 .. R0 + 31 = the sequence; uh = head inside the slab).  The phase is VALU-issue-bound (in-kernel
// timestamps: ~390 VALU instructions for 2 x 8 score elements per lane), so the arithmetic is arranged to be short:
//   * scores live in the log2 domain from the start (log2 e folded into 1/sqrt(dh), rho_i and the mask term), so an
//     exponential is one v_exp_f32;
//   * the cosine branch needs no row maximum: -cos + I <= 2 and the mask term is shifted by its maximum over the
//     keys (softmax is shift-invariant; the constant 1 of "1 - cos + I" is dropped for the same reason);
//   * the "+ I" enters as the initial accumulator of the C^ C^T MFMA (-|c_i|^2 on the diagonal: |c_i|^2 rho_i^2 = 1);
//   * |c|^2 with v_dot2c_f32_bf16; softmax normalisation, beta and the dropout scale are ONE factor per branch, the
//     dropout decisions are predicates (v_cndmask), not multipliers.
struct QaAttnConst {
    DropKey k1, k2;
    float c_beta, c_omb;        // beta / (1 - p), (1 - beta) / (1 - p)
    bool dg[4];                 // lane-constant: key 4 q + e is the query r (diagonal of a 16 x 16 block)
};

// `mv` = additive mask term of key (lane & 31), (1 - mask) * -10000 (0 for an inactive tile), loaded by the caller -- the role-split kernel
// fetches it one problem ahead; `mid()` runs between the softmax and the P V half (that kernel's workgroup barrier; empty otherwise).
// IT >= 0: the query half as a compile-time constant (the diagonal term and the operand selects below fold away)
// MODE 1 (beta == 1): the tile holds V | C of FOUR heads (chunks 4 uh + q | 16 + 4 uh + q) and only the cosine branch runs -- no Q / K fragments,
// no dot-product scores, row maximum, exponentials or dropout draws of that branch (about half of this VALU-bound phase)
template <int IT = -1, int MODE = 0, typename Mid>
__device__ __forceinline__ void qa_attention(const QaAttnConst& kc, const char* qt, int R0, int uh, int it_rt, int t, int h, int H,
                                            bool has_mask, float mv, float* rho, float* madd, bf16* ctx_row, bool act, int r, int q,
                                            int lane, Mid&& mid) {
    const int it = IT >= 0 ? IT : it_rt;
    constexpr float L2E = 1.4426950408889634f;
    auto blk = [&](int mtx) { return MODE == 1 ? 4 * ((mtx == 3 ? 4 : 0) + uh) + q : 4 * (2 * mtx + uh) + q; };
    bf16x8 fq, fk[2], fc[2];
    float ss[2], rho_own[2];
    if (MODE == 0) fq = *(const bf16x8*)(qt + qt_addr(R0 + 16 * it + r, blk(0)));
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        if (MODE == 0) fk[jt] = *(const bf16x8*)(qt + qt_addr(R0 + 16 * jt + r, blk(1)));
        fc[jt] = *(const bf16x8*)(qt + qt_addr(R0 + 16 * jt + r, blk(3)));
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16x2_t c2 = {fc[jt][2 * e], fc[jt][2 * e + 1]};
            s = __builtin_amdgcn_fdot2_f32_bf16(c2, c2, s, false);
        }
        ss[jt] = qred(s, false);
        rho_own[jt] = __builtin_amdgcn_rsqf(ss[jt]);     // 1 / |c_row|
        if (q == 0) rho[16 * jt + r] = rho_own[jt];
    }
    if (has_mask) {                                      // (uniform)
        const float mm = qa_wave_max(mv);
        if (lane < 32) madd[lane] = (mv - mm) * L2E;
    }
    // rho / madd are private to this wave: LDS operations of one wave execute in order, no barrier needed
    __builtin_amdgcn_wave_barrier();
    f32x4 a1[2], a2[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        f32x4 z1;
#pragma unroll
        for (int e = 0; e < 4; ++e) z1[e] = (kc.dg[e] && it == jt) ? -ss[jt] : 0.f;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        a1[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[jt], it == 0 ? fc[0] : fc[1], z1, 0, 0, 0);
        if (MODE == 0) a2[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[jt], fq, z, 0, 0, 0);
    }
    const float rl = (it == 0 ? rho_own[0] : rho_own[1]) * L2E;
    constexpr float isql = 0.17677669529663687f * L2E;      // log2(e) / sqrt(32)
    float m2 = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const f32x4 rj = *(const f32x4*)(rho + 16 * jt + 4 * q);
        f32x4 mj = {0.f, 0.f, 0.f, 0.f};
        if (has_mask) mj = *(const f32x4*)(madd + 16 * jt + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a1[jt][e] = fmaf(-a1[jt][e], rl * rj[e], mj[e]);
            if (MODE == 0) {
                a2[jt][e] = fmaf(a2[jt][e], isql, mj[e]);
                m2 = raw_max(m2, a2[jt][e]);
            }
        }
    }
    if (MODE == 0) m2 = qred(m2, true);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a1[jt][e] = __builtin_amdgcn_exp2f(a1[jt][e]);
            s1 += a1[jt][e];
            if (MODE == 0) {
                a2[jt][e] = __builtin_amdgcn_exp2f(a2[jt][e] - m2);
                s2 += a2[jt][e];
            }
        }
    s1 = qred(s1, false);
    if (MODE == 0) s2 = qred(s2, false);
    const float c1 = kc.c_beta * __builtin_amdgcn_rcpf(s1), c2 = MODE == 0 ? kc.c_omb * __builtin_amdgcn_rcpf(s2) : 0.f;
    mid();
    // mix + dropout -> P^T, packed as the B operand (k slot e of lane (r, q): key 16 (e >> 2) + 4 q + (e & 3))
    bf16x8 pb;
    const int i = 16 * it + r;
    {
        const uint32_t hrow = (uint32_t)((((uint64_t)t * H + h) * 32) + i);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            bool kp1[4] = {true, true, true, true}, kp2[4] = {true, true, true, true};
            if (kc.k1.on) {
                drop_keep4(kc.k1, hrow, (uint32_t)(4 * jt + q), kp1);
                if (MODE == 0) drop_keep4(kc.k2, hrow, (uint32_t)(4 * jt + q), kp2);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                pb[4 * jt + e] = MODE == 0 ? (bf16)fmaf(c1, kp1[e] ? a1[jt][e] : 0.f, kp2[e] ? c2 * a2[jt][e] : 0.f) : (bf16)(kp1[e] ? c1 * a1[jt][e] : 0.f);
        }
    }
    // O^T[c][i] = sum_j V[j][c] P[i][j]: A operand = transposed read of the V block (rows = keys)
    f32x4 o[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int row_lo = R0 + 4 * q + (r >> 2), row_hi = row_lo + 16;
        const int cb = (MODE == 1 ? uh : 2 * 2 + uh) * 64 + (16 * ct + 4 * (r & 3)) * 2;      // byte column inside the row: V block of this head
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(qt + qt_addr(row_lo, cb >> 4) + (cb & 15)));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(qt + qt_addr(row_hi, cb >> 4) + (cb & 15)));
        const bf16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        o[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb, z, 0, 0, 0);
    }
    // one 16-byte store per lane: the 64 bytes of this (row, head) leave as one run (common.h: store_row32)
#ifdef PMGT_QA3_NO_CTX
    store_row32(ctx_row, o[0], o[1], q, act && t < 0);
#else
    store_row32(ctx_row, o[0], o[1], q, act);
#endif
}

__device__ __forceinline__ QaAttnConst qa_attn_const(const QkvcAttn& a, int r, int q) {
    QaAttnConst kc;
    kc.k1 = make_drop_key(a.drop1);
    kc.k2 = make_drop_key(a.drop2);
    kc.c_beta = a.beta * kc.k1.scale;
    kc.c_omb = (1.f - a.beta) * kc.k2.scale;
#pragma unroll
    for (int e = 0; e < 4; ++e) kc.dg[e] = (4 * q + e) == r;
    return kc;
}

// ------------------------------------------------------------------------------------------------
// Two-workgroups-per-CU form (fp8 projection modes).  In-kernel timestamps of a one-8-wave-workgroup form (round 1, removed) showed
// its tile step as a SEQUENCE of phases bound by different units -- projection (MFMA pipe + LDS fragment reads, ~2850 cycles for the two
// waves of a SIMD), attention (VALU issue: ~390 instructions x 4 cycles x 2 waves, ~3700 cycles) -- that one 8-wave
// workgroup marches through in lockstep, so the MFMA pipe idles during attention and the VALU during projection.
// Here a workgroup has FOUR waves, each holding a 64-column block of the slab's W (128 VGPRs), and works on one
// sequence (32 rows) per step; two such workgroups share a CU (256 VGPRs per wave, 49 KB LDS each) and drift out of
// phase, so one's projection overlaps the other's attention.  Every A-tile fragment read now feeds 8 MFMAs instead of
// 4: half the LDS traffic per row.
//   wave w = (head uh = w >> 1, matrix pair mh = w & 1: {Q, K} or {V, C});  attention role (head uh, query half w & 1).
// F8: 1 = fp8 projection with x quantised inside the kernel, 2 = fp8 projection on x that arrives as e4m3 rows + scales
// (0 was the bf16 instance of this form: superseded by the role-split kernel below and no longer instantiated)
template <int KS, int F8> struct QaCfg2 {
    static constexpr int K = 32 * KS;
    static constexpr int ROWB = F8 ? K : K * 2;               // LDS bytes per row of the x tile (e4m3 in the fp8 mode)
    static constexpr int CPR = K / 8;                         // 8-element chunks per row (16 B of bf16 in HBM)
    static constexpr int TILEB = 32 * ROWB;
    static constexpr int LPT = 32 * CPR / 256;
    static_assert(LPT * 256 == 32 * CPR, "tile must be a whole number of chunks per thread");
    static_assert(!F8 || CPR == 32, "fp8 form: one row per 32 lanes (hidden size 256)");
    static_assert(F8 >= 1 && F8 <= 2, "F8 mode");
    static constexpr int SWZ = CPR >= 16 ? 15 : CPR - 1;
    static constexpr int QTB = 32 * 512;                      // projection tile: 32 rows x 256 bf16
    // A ring + projection tile + per-wave {rho, madd} + bias (+ fp8: channel scales of W, row scales of the two x tiles)
    static constexpr int SMEM = 2 * TILEB + QTB + 4 * 64 * 4 + 256 * 4 + (F8 ? 256 * 4 + 2 * 32 * 4 : 0);
};

// F8: same structure, the projection on the fp8 MFMA.  W fragments are 8 bytes (64 VGPRs per wave instead of 128); the x tile
// is quantised per row (absmax over the 32 lanes that hold the row -> e4m3, fp8.h contract) on its way from the staging
// registers into LDS, so the fragment reads move half the bytes (16-byte chunks XOR-swizzled by row & 15, as the bf16 tile).  acc * rowscale * wscale + bias
// gives the same bf16 projection tile; everything after it is unchanged.  F8 == 2: the producer of x (the fused-LayerNorm
// epilogue of gemm_ws / embed_mix, both HBM-bound with the row in 32 lanes) already wrote the e4m3 rows and their scales
// (bit-identical to what the in-kernel quantisation computes from the bf16 x): the kernel reads half the bytes and its
// VALU-bound attention phase loses the ~160 instructions of the quantisation.
template <int KS, int F8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void qkvc_attn_fwd2_kernel(QkvcAttn a) {
    using C = QaCfg2<KS, F8>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* qt = smem + 2 * C::TILEB;
    float* wl = (float*)(qt + C::QTB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int d = a.H * 32;
    const int ny = a.H / 2;
    const int b = blockIdx.x;
    const int y = (b >> 3) % ny, x = (b & 7) + 8 * (b / (8 * ny));
    const int gx = gridDim.x / ny;
    const int num_mt = a.Tseq;                  // one sequence per step
    const int uh = wave >> 1, mh = wave & 1, it = wave & 1;
    const int h = 2 * y + uh;
    auto gcol = [&](int c) { return (c >> 6) * d + (2 * y + ((c >> 5) & 1)) * 32 + (c & 31); };
    auto ocol = [&](int c) { return a.hm ? ((2 * y + ((c >> 5) & 1)) * 4 + (c >> 6)) * 32 + (c & 31) : gcol(c); };
    // local column of this wave's 16-column block cb: matrix 2 mh + (cb >> 1), head uh, half cb & 1
    auto cbase = [&](int cb) { return 64 * (2 * mh + (cb >> 1)) + 32 * uh + 16 * (cb & 1); };

    // fp8 forms: the block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 with every scale = 1 (E8M0 0x7f) -- 128 k per
    // instruction at twice the per-k rate of the non-scaled fp8 / bf16 forms (the only fp8 MFMA of gfx950 that reaches the fp8
    // peak): 16 instead of 64 MFMAs per wave and step.  A lane holds 32 consecutive k (32 bytes) of its row for both operands
    // (k = 128 s + 32 q + 0..31), W from HBM once, x from LDS with two 16-byte reads.
    typedef int i32x8_t __attribute__((ext_vector_type(8)));
    constexpr int KQ = F8 ? KS / 4 : KS;                 // MFMA k-steps
    using wfrag_t = std::conditional_t<F8 != 0, i32x8_t, bf16x8>;
    wfrag_t wf[4][KQ];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int n = gcol(cbase(cb) + r);
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            if constexpr (F8) {
                const u32x4 lo = *(const u32x4*)((const char*)a.W8 + (int64_t)n * a.ldw + 128 * ks + 32 * q);
                const u32x4 hi = *(const u32x4*)((const char*)a.W8 + (int64_t)n * a.ldw + 128 * ks + 32 * q + 16);
                wf[cb][ks] = (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            } else {
                wf[cb][ks] = *(const bf16x8*)((const bf16*)a.W + (int64_t)n * a.ldw + 32 * ks + 8 * q);
            }
        }
    }
    float* bias_l = wl + 4 * 64;          // bias of the slab's 256 local columns (LDS: 16 VGPRs fewer)
    bias_l[tid] = a.bias ? a.bias[gcol(tid)] : 0.f;
    float* wscale_l = bias_l + 256;       // fp8: dequantisation scale of the slab's 256 output channels
    float* rowscale_l = wscale_l + 256;   // fp8: [2 tiles][32 rows]
    if constexpr (F8) wscale_l[tid] = a.wscale[gcol(tid)];

    u32x4 ra[1][C::LPT];
    float rsc = 1.f;                   // F8 == 2: scale of row (tid & 31) of the tile in flight
    auto gload = [&](int mt, int set) {
#pragma unroll
        for (int i = 0; i < C::LPT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / C::CPR, ch = idx % C::CPR;
            if constexpr (F8 == 2) {
                const u32x2 w = *(const u32x2*)((const char*)a.X8 + (int64_t)(mt * 32 + row) * a.ldx + ch * 8);
                ra[set][i] = (u32x4){w[0], w[1], 0u, 0u};
            } else {
                ra[set][i] = *(const u32x4*)((const char*)a.X + ((int64_t)(mt * 32 + row) * a.ldx) * 2 + ch * 16);
            }
        }
        if constexpr (F8 == 2) rsc = a.xscale[(int64_t)mt * 32 + (tid & 31)];
    };
    auto sstore = [&](int buf, int set) {
#pragma unroll
        for (int i = 0; i < C::LPT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / C::CPR, ch = idx % C::CPR;
            if constexpr (F8 == 2) {
                *(u32x2*)(sA + buf * C::TILEB + row * C::ROWB + (((ch >> 1) ^ (row & 15)) << 4) + 8 * (ch & 1)) = (u32x2){ra[set][i][0], ra[set][i][1]};
            } else if constexpr (F8 == 1) {
                const bf16x8 xv = __builtin_bit_cast(bf16x8, ra[set][i]);
                float f[8], m = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { f[e] = (float)xv[e]; m = raw_max(m, fabsf(f[e])); }
                m = max_lanes32(m);                       // the 32 lanes of this half-wave hold the row
                const float inv = m > 0.f ? E4M3_MAX / m : 1.f;
                if (ch == 0) rowscale_l[buf * 32 + row] = m > 0.f ? m / E4M3_MAX : 1.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(f[e] * inv, -E4M3_MAX, E4M3_MAX);
                *(u32x2*)(sA + buf * C::TILEB + row * C::ROWB + (((ch >> 1) ^ (row & 15)) << 4) + 8 * (ch & 1)) = pack8_e4m3(f);
            } else {
                *(u32x4*)(sA + buf * C::TILEB + row * C::ROWB + ((ch ^ (row & C::SWZ)) << 4)) = ra[set][i];
            }
        }
        if constexpr (F8 == 2) { if (tid < 32) rowscale_l[buf * 32 + tid] = rsc; }
    };

    const QaAttnConst kc = qa_attn_const(a, r, q);
    float* rho = wl + wave * 64;
    float* madd = rho + 32;
    bf16* QKVC = (bf16*)a.qkvc;
    bf16* CTX = (bf16*)a.ctx;
    const int erow = tid >> 5, ech = tid & 31;

    QA_PROF_DECL
    auto tile_step = [&](int P, int mt) {
#ifdef PMGT_QA_PROF
        ++pacc[7];
#endif
        QA_STAMP(0);
        __syncthreads();                 // A tile visible; every wave is past the previous step's attention phase
        QA_STAMP(1);
        // the next tile travels through registers only during the projection phase (the attention phase, where register
        // pressure peaks, never holds it): loaded here, parked in the other LDS buffer after the second barrier
        const bool more = mt + gx < num_mt;
        if (more) gload(mt + gx, 0);
        // the projection phase issues one MFMA per 16 cycles: give it priority over the other workgroup's VALU-bound
        // attention phase, so the MFMA pipe never waits for an issue slot (and the younger workgroup of the CU is
        // not starved by oldest-first arbitration)
        __builtin_amdgcn_s_setprio(3);
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[i][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* a_base = sA + P * C::TILEB;
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            wfrag_t fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 16 * i + r;
                if constexpr (F8) {      // 32 bytes of the row: 16-byte chunks 2 (4 ks + q) and + 1, XOR-swizzled by row & 15
                    const int c0 = 2 * (4 * ks + q);
                    const u32x4 lo = *(const u32x4*)(a_base + row * C::ROWB + ((c0 ^ (row & 15)) << 4));
                    const u32x4 hi = *(const u32x4*)(a_base + row * C::ROWB + (((c0 + 1) ^ (row & 15)) << 4));
                    fa[i] = (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                } else {
                    fa[i] = *(const bf16x8*)(a_base + row * C::ROWB + (((4 * ks + q) ^ (row & C::SWZ)) << 4));
                }
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int i = 0; i < 2; ++i) {    // D[n = 4 q + e][m = r]: acc[i][cb][e] = out[16 i + r][cbase(cb) + 4 q + e]
                    if constexpr (F8)      // formats 0 / 0 = e4m3 x e4m3; block scales 0x7f = 2^0
                        acc[i][cb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[cb][ks], fa[i], acc[i][cb], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    else acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb][ks], fa[i], acc[i][cb], 0, 0, 0);
                }
        }
        QA_STAMP(2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int row = 16 * i + r;
                f32x4 v = acc[i][cb];
                if constexpr (F8) v = v * (*(const f32x4*)(wscale_l + cbase(cb) + 4 * q) * rowscale_l[P * 32 + row]);
                v += *(const f32x4*)(bias_l + cbase(cb) + 4 * q);
                bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                *(bf16x4*)(qt + qt_addr(row, (cbase(cb) >> 3) + (q >> 1)) + 8 * (q & 1)) = o;
            }
        __builtin_amdgcn_s_setprio(0);
        QA_STAMP(3);
        __syncthreads();
        QA_STAMP(4);
        if (more) sstore(P ^ 1, 0);      // that buffer was last read in the previous step's projection phase
        // ---- (a) projection tile -> HBM
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = erow + 8 * ps;
            *(u32x4*)(QKVC + (int64_t)(mt * 32 + row) * a.ldq + ocol(8 * ech)) = *(const u32x4*)(qt + qt_addr(row, ech));
        }
        QA_STAMP(5);
        // ---- (b) attention of (sequence mt, head h, queries 16 it .. 16 it + 15)
        const int t = mt;
        if (it == 1 && t < a.cls_only_seqs) return;      // (wave-uniform) only query row 0 of this sequence is ever read
        const float mv = a.mask ? (1.f - a.mask[(int64_t)t * 32 + (lane & 31)]) * -10000.f : 0.f;
        qa_attention(kc, qt, 0, uh, it, t, h, a.H, a.mask != nullptr, mv, rho, madd, CTX + ((int64_t)t * 32 + 16 * it + r) * a.ldc + h * 32, true, r, q,
                     lane, [] {});
        QA_STAMP(6);
    };

    int mt = x;
    if (mt < num_mt) { gload(mt, 0); sstore(0, 0); }
    for (int P = 0; mt < num_mt; mt += gx, P ^= 1) tile_step(P, mt);
    QA_PROF_FLUSH;
}


// ------------------------------------------------------------------------------------------------
// Role-split form (bf16).  Stamps of the two-workgroup form above (tools/prof/qa_prof.py): a wave marches through
// projection (1580 cycles for its 64 MFMAs) -> bias / projection-tile writes (1050) -> copy-out (660) -> attention (2470 - 2760, ~390
// dependent VALU instructions) in sequence, 6200 - 6700 cycles per (sequence, slab) with TWO waves per SIMD: the matrix pipe is busy a
// quarter of the time and the VALU issues one instruction per ~6 cycles -- there are not enough waves to fill either, and there cannot
// be more while every wave holds 128 VGPRs of W.  Here ONE 16-wave workgroup per CU splits the roles (as attn_bwd_wgrad_kernel does):
//   waves 0-7  GEMM role: wave g keeps the W rows of 32 output columns (matrix g >> 1, head g & 1 of the slab; 64 VGPRs), streams the
//              x tiles in by LDS-DMA (a 4-slot ring, three tiles ahead), per step 32 MFMAs (the 32 x 32 block of the sequence's
//              projection) -> + bias -> bf16 -> projection-tile ring in LDS; it also copies the previous step's tile to HBM;
//   waves 8-15 attention role: group (w >> 2) & 1 takes the even / odd steps, wave = (head uh, query half it); a problem spans TWO
//              iterations (softmax half | barrier | P V half), so each SIMD always holds two GEMM waves feeding the matrix pipe and two
//              attention waves in different halves feeding the VALU.
// One s_barrier per iteration; iteration i: GEMM projects step i, copies out step i - 1; attention runs the first half of step i - 1
// (group (i - 1) & 1) and the second half of step i - 2 (the other group).  Projection tile s lives in ring slot s & 3 from iteration s
// (written) to s + 2 (last read); x tile s is DMA'd during iteration s - 3 into the slot x tile s - 4 left at the end of iteration s - 4.
// ------------------------------------------------------------------------------------------------
template <int KS> struct QaCfg3 {
    static constexpr int K = 32 * KS, ROWB = K * 2, CPR = K / 8;
    static constexpr int XT = 32 * ROWB, QTB = 32 * 512;              // x tile, projection tile (32 rows x 256 bf16)
    static constexpr int RPI = 1024 / ROWB, LPR = 64 / RPI;           // rows per 1-KB DMA instruction, lanes per row
    static constexpr int NDMA = XT / 1024 / 8;                        // DMA instructions per GEMM wave and tile
    static_assert(NDMA >= 1 && NDMA * 8 * 1024 == XT && CPR >= 16, "x tile = whole 1-KB instructions per GEMM wave, >= 16 chunks per row");
    static constexpr int X0 = 0, Q0 = 4 * XT, WL0 = Q0 + 4 * QTB, MK0 = WL0 + 8 * 64 * 4, SMEM = MK0 + 8 * 128;      // + mask rows of 8 steps
};

typedef __attribute__((address_space(3))) void qa_lds_void_t;
typedef __attribute__((address_space(1))) const void qa_gbl_void_t;

// VC (beta == 1, QkvcAttn::vc_only): a column slab is {V, C} x {heads 4 y .. 4 y + 3} -- GEMM wave g = (matrix 2 + (g >> 2), head 4 y + (g & 3)) --
// so a step covers FOUR heads of its sequence with the same projection work per wave, and the eight attention waves (head uh = aw >> 1,
// query half aw & 1) each run the cosine branch of one problem per step instead of both branches of one problem per two steps: half the
// iterations per launch, and Q / K are neither projected nor written.
template <int KS, bool VC = false>
__global__ __launch_bounds__(1024) void qkvc_attn_fwd3_kernel(QkvcAttn a) {
    using C = QaCfg3<KS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int d = a.H * 32, ny = VC ? a.H / 4 : a.H / 2;
    const int b = blockIdx.x;
    const int y = (b >> 3) % ny, x = (b & 7) + 8 * (b / (8 * ny));
    const int gx = gridDim.x / ny;
    const int nsteps = x < a.Tseq ? (a.Tseq - x + gx - 1) / gx : 0;       // sequences x, x + gx, ... (uniform over the workgroup)
    const int NI = nsteps + 2;
    // tile column c -> (matrix, head): {q, k, v, c} x two heads, or (VC) {v, c} x four heads
    auto cmat = [&](int c) { return VC ? 2 + (c >> 7) : (c >> 6); };
    auto chead = [&](int c) { return VC ? 4 * y + ((c >> 5) & 3) : 2 * y + ((c >> 5) & 1); };
    auto gcol = [&](int c) { return cmat(c) * d + chead(c) * 32 + (c & 31); };
    auto ocol = [&](int c) { return a.hm ? (chead(c) * 4 + cmat(c)) * 32 + (c & 31) : gcol(c); };
    QA3_PROF_DECL
    // LDS operations of this wave done, then the workgroup barrier (no vmcnt wait: the DMA ring stays in flight across it)
    auto bar = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        QA3_STAMP(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        QA3_STAMP(1);
    };

    if (wave >= 8) {
        // ================================ attention role ================================
        const int aw = wave - 8, grp = aw >> 2, uh = VC ? aw >> 1 : (aw >> 1) & 1, it = aw & 1;
        const int h = (VC ? 4 : 2) * y + uh;
        const QaAttnConst kc = qa_attn_const(a, r, q);
        float* rho = (float*)(smem + C::WL0) + aw * 64;
        float* madd = rho + 32;
        bf16* CTX = (bf16*)a.ctx;
        const bool has_mask = a.mask != nullptr;
        // The mask row of a step arrives in LDS with its x tile (GEMM wave 0's DMA): these waves issue no vector-memory LOAD at all, so nothing
        // ever makes them wait on vmcnt -- a global load of the mask value, even one problem ahead, put an s_waitcnt vmcnt(0) at the top of
        // every problem, which waited for the previous problem's context STORE to be acknowledged (65 us of a 390 us launch).
        __builtin_amdgcn_s_barrier();                     // the GEMM role's "x(0) has landed" barrier
        if constexpr (VC) {
            // every wave, every step: the problem of step s (tile written in iteration s) runs inside iteration s + 1
            bar();
            auto nomid = []() {};
            for (int s = 0; s < nsteps; ++s) {
                const int t = x + s * gx;
                const float mv = has_mask ? (1.f - *(const float*)(smem + C::MK0 + (s & 7) * 128 + (lane & 31) * 4)) * -10000.f : 0.f;
                const char* qt = smem + C::Q0 + (s & 3) * C::QTB;
                if (!(it == 1 && t < a.cls_only_seqs)) {
                    bf16* crow = CTX + ((int64_t)t * 32 + 16 * it + r) * a.ldc + h * 32;
                    if (it == 0) qa_attention<0, 1>(kc, qt, 0, uh, 0, t, h, a.H, has_mask, mv, rho, madd, crow, true, r, q, lane, nomid);
                    else qa_attention<1, 1>(kc, qt, 0, uh, 1, t, h, a.H, has_mask, mv, rho, madd, crow, true, r, q, lane, nomid);
                }
                QA3_STAMP(2);
                bar();
            }
            bar();
            QA3_PROF_FLUSH;
            return;
        }
        int nb = 0;
        for (; nb < 1 + grp; ++nb) bar();                 // the iterations before this group's first tile exists
        for (int s = grp; s < nsteps; s += 2, nb += 2) {
            const int t = x + s * gx;
            const float mv = has_mask ? (1.f - *(const float*)(smem + C::MK0 + (s & 7) * 128 + (lane & 31) * 4)) * -10000.f : 0.f;
            const char* qt = smem + C::Q0 + (s & 3) * C::QTB;
#ifdef PMGT_QA3_NO_ATTN
            if (t >= 0) {                                 // (ablation build: the attention waves only keep the barriers)
#else
            if (it == 1 && t < a.cls_only_seqs) {         // (wave-uniform) only query row 0 of this sequence is ever read
#endif
                bar();
            } else {
                bf16* crow = CTX + ((int64_t)t * 32 + 16 * it + r) * a.ldc + h * 32;
                if (it == 0) qa_attention<0>(kc, qt, 0, uh, 0, t, h, a.H, has_mask, mv, rho, madd, crow, true, r, q, lane, bar);
                else qa_attention<1>(kc, qt, 0, uh, 1, t, h, a.H, has_mask, mv, rho, madd, crow, true, r, q, lane, bar);
            }
            QA3_STAMP(2);
            bar();
        }
        for (; nb < NI; ++nb) bar();
        QA3_PROF_FLUSH;
        return;
    }
    // ==================================== GEMM role ====================================
    // The kernel is VALU-issue-bound (one wave-instruction per 4 cycles and SIMD, all 16 waves counted: ~2500 issue cycles per iteration
    // against 1024 of the matrix pipe), so this role spends as few vector instructions as it can: every per-lane address is a loop-invariant
    // register plus an IMMEDIATE ring-slot offset (the loop is unrolled over the four slots) or a scalar base (global side), and the bias is
    // the C operand of the first MFMA of each accumulator.
    __builtin_amdgcn_s_setprio(2);         // the matrix pipe's waves first (measured neutral against 0 / attention-first: the kernel is bound by issue THROUGHPUT)
    const int g = wave;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(qa_lds_void_t*)smem;
    // v_mfma_f32_32x32x16_bf16, operands swapped (A = W rows = output columns, B = x rows): the 32 x 32 block of a wave is ONE
    // accumulator tile, 16 instructions per step instead of 32 of the 16x16x32 form -- an MFMA holds the SIMD's vector issue for 8 cycles
    // whatever its shape, and vector issue is what this kernel runs out of.  Lane l: A row / B column l & 31, k = 16 kk + 8 (l >> 5) .. + 7;
    // D register v = 4 g4 + e: output column 32 g + 8 g4 + 4 (l >> 5) + e of x row l & 31.
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int KK = 2 * KS;
    const int l31 = lane & 31, lh = lane >> 5;
    bf16x8 wf[KK];
    {
        const int n = gcol(32 * g + l31);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wf[kk] = *(const bf16x8*)((const bf16*)a.W + (int64_t)n * a.ldw + 16 * kk + 8 * lh);
    }
    // The bias enters as one more MFMA (k slots 0, 1 of an extra k-step: bias split into bf16 hi + lo against ones; every other slot of the A
    // fragment is zero): 8 VGPRs instead of the 16 a C-operand copy of it would hold, and no VALU adds in the epilogue.  |bias - (hi + lo)| <=
    // 2^-17 |bias|.
    bf16x8 wb = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    if (a.bias && lh == 0) {
        const float bf = a.bias[gcol(32 * g + l31)];
        const bf16 hi = (bf16)bf;
        wb[0] = hi;
        wb[1] = (bf16)(bf - (float)hi);
    }
    const bf16 one_ = (bf16)1.f;
    const bf16x8 ones = {one_, one_, one_, one_, one_, one_, one_, one_};

    // x tile by LDS-DMA: instruction j of wave g covers rows (NDMA g + j) RPI ..; lane -> row + lane / LPR, LDS chunk slot lane % LPR, which
    // holds global chunk slot ^ (row & 15) (the XOR swizzle of the fragment reads below).  (row0 + lane / LPR) & 15 = (row0 & 15) ^ (lane / LPR):
    // row0 is a multiple of RPI.
    uint32_t dma_off[C::NDMA];
#pragma unroll
    for (int j = 0; j < C::NDMA; ++j) {
        const int row0 = (C::NDMA * g + j) * C::RPI, row = row0 + lane / C::LPR;
        dma_off[j] = (uint32_t)row * (uint32_t)a.ldx * 2u + (uint32_t)(((lane % C::LPR) ^ (row & 15)) << 4);
    }
    const bool mask_dma = g == 0 && a.mask != nullptr;                  // (wave-uniform) wave 0 also brings the step's 128-byte mask row
    const int ND = C::NDMA + (mask_dma ? 1 : 0);                        // vector-memory loads of this wave per step
    auto dma_x = [&](int s, int slot) __attribute__((always_inline)) {
        const char* src = (const char*)a.X + (size_t)((uint32_t)(x + s * gx) * 32u * (uint32_t)a.ldx * 2u);      // (scalar)
        char* dst = smem + C::X0 + slot * C::XT;
#pragma unroll
        for (int j = 0; j < C::NDMA; ++j)
            __builtin_amdgcn_global_load_lds((qa_gbl_void_t*)(src + (size_t)dma_off[j]), (qa_lds_void_t*)(dst + (C::NDMA * g + j) * C::RPI * C::ROWB), 16, 0, 0);
        if (mask_dma) {
            const char* msrc = (const char*)a.mask + (size_t)(x + s * gx) * 128;
            if (lane < 8)
                __builtin_amdgcn_global_load_lds((qa_gbl_void_t*)(msrc + lane * 16), (qa_lds_void_t*)(smem + C::MK0 + (s & 7) * 128), 16, 0, 0);
        }
    };
    // s_waitcnt vmcnt(n), n <= 10, as an immediate
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
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        }
    };
    // fragment reads of an x tile: row l & 31, chunk (2 kk + (l >> 5)) ^ (row & 15): bits 5.. of the chunk's byte offset are the k-step,
    // so the 2 KS addresses of a tile are ONE register XOR a constant (+ the slot as the instruction's offset)
    const uint32_t fr0 = (uint32_t)(l31 * C::ROWB + (((lh ^ (l31 & 1)) << 4) | ((l31 & 15) >> 1 << 5)));      // bits 5.. 5 + log2(KK) - 1 hold (row & 15) >> 1 only
    // projection-tile addresses.  Writes: row l & 31, columns 32 g + 8 g4 + 4 (l >> 5) .. + 3 = chunk 4 g + g4, half l >> 5.
    // Copy-out reads: row 16 p + 2 g + (lane >> 5), chunk lane & 31 -> 16 bytes to Q|K|V|C row t * 32 + row, column ocol(8 chunk).
    uint32_t qw[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) qw[g4] = lds0 + C::Q0 + (uint32_t)(l31 * 512 + (((4 * g + g4) ^ (l31 & 15)) << 4) + 8 * lh);
    const int co_row = 2 * g + (lane >> 5);
    const uint32_t co_lds = lds0 + C::Q0 + (uint32_t)(co_row * 512 + (((lane & 31) ^ (co_row & 15)) << 4));
    const uint32_t co_glb = ((uint32_t)co_row * (uint32_t)a.ldq + (uint32_t)ocol(8 * (lane & 31))) * 2u;

    if (0 < nsteps) dma_x(0, 0);
    if (1 < nsteps) dma_x(1, 1);
    if (2 < nsteps) dma_x(2, 2);
    // x(0) has landed for this wave (W / bias loads are older than the DMAs and drain first), then for everyone
    wait_vm(max(0, min(2, nsteps - 1)) * ND);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // one iteration; SL = i & 3 as a compile-time constant
    auto iteration = [&](auto SLc, int i) __attribute__((always_inline)) {
        constexpr int SL = decltype(SLc)::value, SLP = (SL + 3) & 3;        // ring slots of step i and of step i - 1 (= of x(i + 3))
        QA3_STAMP(2);
        // ---- (1) projection tile of step i - 1 -> HBM (first: its stores have drained by the vmcnt wait at the end of the iteration)
        if (i >= 1 && i - 1 < nsteps) {
            char* dst = (char*)a.qkvc + (size_t)((uint32_t)(x + (i - 1) * gx) * 32u * (uint32_t)a.ldq * 2u);      // (scalar)
            u32x4 v[2];
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]) : "v"(co_lds), "n"(SLP * C::QTB), "n"(SLP * C::QTB + 16 * 512) : "memory");
#ifndef PMGT_QA3_NO_COPY
            *(u32x4*)(dst + (size_t)co_glb) = v[0];
            *(u32x4*)(dst + (size_t)(16u * (uint32_t)a.ldq * 2u) + (size_t)co_glb) = v[1];
#else
            if (a.Tseq < 0) { *(u32x4*)(dst + (size_t)co_glb) = v[0]; *(u32x4*)(dst + (size_t)(16u * (uint32_t)a.ldq * 2u) + (size_t)co_glb) = v[1]; }
#endif
        }
        QA3_STAMP(3);
        // ---- (2) the ring slot x(i - 1) left at the last barrier takes x(i + 3)
        if (i + 3 < nsteps) dma_x(i + 3, SLP);
        // ---- (3) projection of step i
        if (i < nsteps) {
            f32x16 acc;
            u32x4 fa[2];
            auto rd = [&](int kk) __attribute__((always_inline)) {
                const uint32_t ad = (fr0 ^ (uint32_t)(kk << 5)) + (lds0 + C::X0);
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(fa[kk & 1]) : "v"(ad), "n"(SL * C::XT) : "memory");
            };
            rd(0);
            {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, ones, z, 0, 0, 0);
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if (kk + 1 < KK) {
                    rd(kk + 1);
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa[kk & 1]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[kk & 1]));
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk], __builtin_bit_cast(bf16x8, fa[kk & 1]), acc, 0, 0, 0);
            }
            QA3_STAMP(4);
            // bf16, into projection-tile slot SL (8 bytes per lane and column group)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const bf16x4 o = {(bf16)acc[4 * g4], (bf16)acc[4 * g4 + 1], (bf16)acc[4 * g4 + 2], (bf16)acc[4 * g4 + 3]};
                asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(qw[g4]), "v"(__builtin_bit_cast(u32x2, o)), "n"(SL * C::QTB) : "memory");
            }
        }
        QA3_STAMP(5);
        // ---- (4) x(i + 1) has landed for this wave.  vmcnt counts loads and stores alike and retires them in issue order; issued after
        // x(i + 1)'s DMA: [2 stores, x(i + 2)] in iteration i - 1 and [2 stores, x(i + 3)] in this one.
        if (i + 1 < nsteps) {
            const int younger = min(i + 3, nsteps - 1) - (i + 1);
            wait_vm(younger * ND + ((younger >= 2 && i >= 2) ? 4 : 0));
        }
        bar();
    };
    for (int i = 0; i < NI; i += 4) {
        iteration(std::integral_constant<int, 0>{}, i);
        if (i + 1 < NI) iteration(std::integral_constant<int, 1>{}, i + 1);
        if (i + 2 < NI) iteration(std::integral_constant<int, 2>{}, i + 2);
        if (i + 3 < NI) iteration(std::integral_constant<int, 3>{}, i + 3);
    }
    QA3_PROF_FLUSH;
}

// the role-split form addresses x and Q|K|V|C with 32-bit byte offsets
static bool qa3_ok(const QkvcAttn& a) {
    const int d = a.H * 32;
    return (int64_t)a.Tseq * 32 * a.ldx * 2 < ((int64_t)1 << 32) && (int64_t)a.Tseq * 32 * a.ldq * 2 < ((int64_t)1 << 32) && (d == 256 || d == 128);
}

bool qkvc_attn_supported(const QkvcAttn& a) {
    const int d = a.H * 32;
    if (!a.W8 && !qa3_ok(a)) return false;      // bf16: the role-split form only
    if (a.W8 && !(d == 256 && a.wscale && ((uintptr_t)a.W8 % 8) == 0)) return false;
    if (a.X8 && !(a.W8 && a.xscale && ((uintptr_t)a.X8 % 8) == 0)) return false;
    if (a.vc_only && !(a.beta == 1.f && a.H % 4 == 0 && !a.W8)) return false;      // V | C of four heads per slab, cosine branch alone
    return a.S == 32 && a.dh == 32 && (d == 256 || d == 128) && a.H % 2 == 0 && a.Tseq >= 2 && a.ldx % 8 == 0 && a.ldw % 8 == 0 &&
           a.ldq % 8 == 0 && a.ldc % 4 == 0 && (a.X8 != nullptr || (a.X != nullptr && ((uintptr_t)a.X % 16) == 0)) && ((uintptr_t)a.W % 16) == 0 &&
           ((uintptr_t)a.qkvc % 16) == 0 && ((uintptr_t)a.ctx % 8) == 0 && (a.bias == nullptr || ((uintptr_t)a.bias % 16) == 0) &&
           (a.W != nullptr || a.W8 != nullptr);
}


template <int KS, int F8> static int launch_qa2(const QkvcAttn& a, hipStream_t st) {
    using C = QaCfg2<KS, F8>;
    auto kern = qkvc_attn_fwd2_kernel<KS, F8>;
    PMGT_SMEM_ATTR((const void*)kern, C::SMEM);
    const int ny = a.H / 2;
    const int gx = std::max(8, std::min(512 / ny, a.Tseq) / 8 * 8);       // two workgroups per CU
    note_launch(LT_QKVC_ATTN_FWD);
    hipLaunchKernelGGL(kern, dim3(gx * ny), dim3(256), C::SMEM, st, a);
    PMGT_LAUNCH_OK();
    return 0;
}

template <int KS, bool VC = false> static int launch_qa3(const QkvcAttn& a, hipStream_t st) {
    using C = QaCfg3<KS>;
    auto kern = qkvc_attn_fwd3_kernel<KS, VC>;
    PMGT_SMEM_ATTR((const void*)kern, C::SMEM);
    const int ny = VC ? a.H / 4 : a.H / 2;
    const int gx = std::max(8, std::min(256 / ny, a.Tseq) / 8 * 8);       // one 16-wave workgroup per CU
    note_launch(LT_QKVC_ATTN_FWD);
    if (VC) note_launch(LT_QKVC_ATTN_FWD_VC);
    hipLaunchKernelGGL(kern, dim3(gx * ny), dim3(1024), C::SMEM, st, a);
    PMGT_LAUNCH_OK();
    return 0;
}



int qkvc_attn_fwd(const QkvcAttn& a, hipStream_t st) {
    if (a.Tseq <= 0) return 0;
    PMGT_CHECK(qkvc_attn_supported(a), -2, "qkvc_attn_fwd: unsupported shape S=%d dh=%d H=%d", a.S, a.dh, a.H);
    if (a.W8 && a.X8) return launch_qa2<8, 2>(a, st);
    if (a.W8) return launch_qa2<8, 1>(a, st);
    if (a.vc_only) return a.H * 32 == 256 ? launch_qa3<8, true>(a, st) : launch_qa3<4, true>(a, st);
    return a.H * 32 == 256 ? launch_qa3<8>(a, st) : launch_qa3<4>(a, st);
}

}  // namespace pmgt

#ifdef PMGT_QA_PROF
extern "C" int pmgt_debug_qa_prof_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmgt::g_qa_prof), sizeof(pmgt::g_qa_prof));
}
extern "C" int pmgt_debug_qa3_prof_read(unsigned int* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmgt::g_qa3_prof), sizeof(pmgt::g_qa3_prof));
}
extern "C" int pmgt_debug_qa_blk_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmgt::g_qa_blk), sizeof(pmgt::g_qa_blk));
}
#endif
