// MFMA GEMMs for the PMGT engine on gfx950.
//
// gemm_nt : C[M,N] = epi(A[M,K] * B[N,K]^T)      forward linears and data-gradients
// gemm_tn : slab[s] = P[rows_s,N1]^T * Q[rows_s,N2]   weight-gradients, split over row chunks
//
// Tiling (both): 256 threads = 4 waves in a 2x2 grid, 128x128 output tile, each wave 64x64 as
// 4x4 MFMA 16x16 tiles (64 accumulator VGPRs).  Operands are staged global -> registers -> LDS in
// 16-byte chunks (double-buffered LDS, one barrier per K-step, next tile's global loads issued
// before the MFMAs of the current one).
//   * bf16 mode: v_mfma_f32_16x16x32_bf16, K-step 64; fp32 parity mode: v_mfma_f32_16x16x4_f32
//     (exact fp32 FMA chain), K-step 32.  Same kernel body, `T` selects the fragment path.
//   * NT LDS image: [rows][128 B] with the 16-B chunk index XOR-swizzled by (row & 7) so the
//     ds_read_b128 fragment reads of 16 rows x same k-chunk are bank-conflict free.
//   * TN LDS image: the tiles as they sit in memory ([m][128 cols]); the bf16 path builds the
//     k-strided MFMA fragments with ds_read_b64_tr_b16 (hardware transpose read) and an XOR
//     swizzle chosen so the 32 lanes of a half-wave hit 32 distinct 8-byte slots.
//   * blockIdx -> tile mapping keeps the N-tiles of one M-tile on one XCD (ids b, b+8 share an
//     XCD's L2), so the A tile is fetched from HBM once.
#include <stdlib.h>

#include "gemm.h"

namespace pmgt {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// ------------------------------------------------------------------------------------------------
// NT
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int nt_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// Chunk swizzle of the 64-byte-row LDS-DMA stages (16 rows x 4 chunks of 16 B share the 256 bytes of the 64 banks).  ds_read_b128 is
// served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS table), i.e. a group holds rows
// r = 0-3, 12-15 with k-chunk q and rows 4-11 with k-chunk q ^ 1: the four rows that share a bank window (equal r & 3) must land in four
// different chunks THERE.  chunk = q ^ ((0 - (r >> 2)) & 3) does (keys 0, 3, 2, 1 for r >> 2 = 0..3: {0, 1, 3 ^ 1, 2 ^ 1} are distinct); the
// natural key (r >> 2) & 3 is conflict-free only for 16 CONSECUTIVE lanes and measured 2-way on every fragment read
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 for the 256 x 256 tile).
__device__ __forceinline__ int nt_swz(int row) { return (0 - (row >> 2)) & 3; }

// Shared epilogue of the NT kernels: accumulators -> LDS -> row-contiguous bias / GELU / dropout / residual.
template <typename T, int BM, int BN>
__device__ __forceinline__ void nt_epilogue(const GemmNT& g, f32x4 (&acc)[BM / 32][BN / 32], char* smem, int m0, int n0, int Mlim,
                                            int tid, int wm, int wn, int r, int q) {
    constexpr int TM = BM / 32, TN = BN / 32;
    // ---- epilogue.  C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg.
    // The accumulators go through LDS (two passes of 64 rows, fp32, row stride 132 words: the
    // ds_write_b32 pattern is 2-way = free) so that bias / GELU / dropout / residual run on
    // row-contiguous 8-element chunks and every global access is a full 16-byte vector.
    const DropKey dk = make_drop_key(g.drop);
    T* C = (T*)g.C;
    const T* R = (const T*)g.res;
    T* AUX = (T*)g.aux;
    constexpr int ES = BN + 4;                      // staged row stride in floats
    float* stage = (float*)smem;                    // 64 x ES floats = 33 KiB (BN = 128)
    const int er = tid >> 4, ec = (tid & 15) * (BN / 16);
#pragma unroll
    for (int pass = 0; pass < BM / 64; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        stage[(i * 16 + 4 * q + e) * ES + wn * (BN / 2) + j * 16 + r] = acc[i][j][e];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = er + 16 * it;
            const int m = m0 + pass * 64 + row;
            if (m < Mlim) {
#pragma unroll
                for (int h = 0; h < BN / 64; ++h) {            // 4-element pieces of this thread's chunk
                    const int n = n0 + ec + 4 * h;
                    if (n < g.N) {                              // N % 4 == 0 is checked on the host
                        f32x4 v = *(const f32x4*)(stage + row * ES + ec + 4 * h);
                        if (g.bias) v += *(const f32x4*)(g.bias + n);
                        if (g.epi == EPI_GELU) {
                            f32x4 pre;                              // rounded to T: what backward re-reads
#pragma unroll
                            for (int e = 0; e < 4; ++e) pre[e] = to_f<T>(from_f<T>(v[e]));
                            store4<T>(AUX + (int64_t)m * g.ldaux + n, pre);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_fwd<T>(pre[e]);
                        } else if (g.epi == EPI_GELU_GRAD) {
                            f32x4 pre = load4<T>(AUX + (int64_t)m * g.ldaux + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= gelu_bwd<T>(pre[e]);
                        }
                        if (dk.on) {
                            float dm[4];
                            drop_mul4(dk, (uint32_t)m, (uint32_t)n >> 2, dm);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= dm[e];
                        }
                        if (R) v += load4<T>(R + (g.res_gather ? g.a_rows[m] : (int64_t)m) * g.ldr + n);
                        store4<T>(C + (int64_t)m * g.ldc + n, v);
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT g) {
    constexpr int EPC = 16 / sizeof(T);   // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;           // 128-byte LDS rows
    constexpr int TM = BM / 32, TN = BN / 32;
    constexpr int LA = BM / 32, LB = BN / 32;   // 16-B chunks per thread per tile
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * 128];
    char* sA = smem;
    char* sB = smem + 2 * BM * 128;

    const int num_n = (g.N + BN - 1) / BN;
    const int num_m = (g.M + BM - 1) / BM;
    const int b = blockIdx.x;
    const int grp = b / (8 * num_n), within = b % (8 * num_n);
    const int m_tile = grp * 8 + (within & 7), n_tile = within >> 3;
    if (m_tile >= num_m) return;
    const int Mlim = g.m_dev ? min(g.M, *g.m_dev) : g.M;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    if (m0 >= Mlim) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;

    // per-thread staging assignment: chunk c of rows r0 + 32 i
    const int c = tid & 7, r0 = tid >> 3;
    const char* arow[LA];
    const char* brow[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int m = min(m0 + r0 + 32 * i, Mlim - 1);
        int64_t row = g.a_rows ? g.a_rows[m] : (int64_t)m;
        arow[i] = (const char*)g.A + row * g.lda * (int64_t)sizeof(T);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        int n = min(n0 + r0 + 32 * i, g.N - 1);
        brow[i] = (const char*)g.B + (int64_t)n * g.ldb * (int64_t)sizeof(T);
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x4 ra[LA], rb[LB];
    auto gload = [&](int k0) {
        const int k = k0 + c * EPC;
        const bool ok = k < g.K;
#pragma unroll
        for (int i = 0; i < LA; ++i)
            ra[i] = ok ? *(const u32x4*)(arow[i] + (int64_t)k * sizeof(T)) : (u32x4){0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < LB; ++i)
            rb[i] = ok ? *(const u32x4*)(brow[i] + (int64_t)k * sizeof(T)) : (u32x4){0, 0, 0, 0};
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) *(u32x4*)(sA + buf * BM * 128 + nt_off(r0 + 32 * i, c)) = ra[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *(u32x4*)(sB + buf * BN * 128 + nt_off(r0 + 32 * i, c)) = rb[i];
    };

    const int nk = (g.K + BK - 1) / BK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const char* a_base = sA + buf * BM * 128;
        const char* b_base = sB + buf * BN * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if constexpr (sizeof(T) == 2) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = *(const bf16x8*)(a_base + nt_off(wm * (BM / 2) + i * 16 + r, 4 * kk + q));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = *(const bf16x8*)(b_base + nt_off(wn * (BN / 2) + j * 16 + r, 4 * kk + q));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            } else {
                // lane (r,q) holds k = 16kk + 4q + e, e = 0..3; MFMA e pairs the e-th elements, so the
                // hardware k-slot q of MFMA e is the same real k for A and B.
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = *(const f32x4*)(a_base + nt_off(wm * (BM / 2) + i * 16 + r, 4 * kk + q));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = *(const f32x4*)(b_base + nt_off(wn * (BN / 2) + j * 16 + r, 4 * kk + q));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    nt_epilogue<T, BM, BN>(g, acc, smem, m0, n0, Mlim, tid, wm, wn, r, q);
}

// ------------------------------------------------------------------------------------------------
// NT, bf16, 256 x 256 tile, 512 threads (8 waves as 2 x 4, 128 x 64 outputs each), same 4-stage LDS-DMA ring
// (K-step 32).  The 128 x 128 kernels re-read the W tile once per 128 rows and A once per 128 columns, all
// through L2 (for dX = dQKVC W at M = 393k, K = 1024, N = 256 that is 3.2 GB of L2 -> CU traffic for 1.2 GB of
// HBM bytes); this tile halves both.  No row gather, no device row count (those stay on the small tile).
// ------------------------------------------------------------------------------------------------
// BN = 256 / 8 waves: one workgroup per CU (128 KiB ring).  BN = 128 / 4 waves: 72 KiB ring (3 stages), TWO workgroups
// per CU, so one's prologue / epilogue overlaps the other's main loop.
#ifdef PMGT_TN_PROF
// cycles of gemm_nt_big_kernel: [block slot][wave][0 vmcnt wait | 1 barrier | 2 DMA issue | 3 LDS reads + MFMA | 4 prologue | 5 epilogue]
__device__ unsigned int g_nt_prof[2][8][6];
#endif
// LNB (BN = 256 only): the tile holds whole rows of an N = 256 output that is the gradient dy of a LayerNorm (GemmNT::lnb_*): behind
// the main loop the workgroup runs that LayerNorm's backward on its 256 rows (see the phase below) instead of storing dy.
// EPL = 2 (BN = 256 only; GemmNT::lnf_*): the rows are the input of a LayerNorm -- x = bf16(dropout(acc + bias) + residual) goes to the same
// LDS image and the LayerNorm FORWARD runs on it (BertOutput at intermediate sizes above 512: K = I, N = 256).
template <int BN, int NW, int EPL = 0>
__global__ __launch_bounds__(64 * NW) void gemm_nt_big_kernel(GemmNT g) {
    typedef bf16 T;
    constexpr bool LNB = EPL == 1, LNF = EPL == 2;
    static_assert(EPL == 0 || BN == 256, "LayerNorm phases: whole rows in the tile");
    constexpr int BM = 256, ROWB = 64, STAGE = (BM + BN) * ROWB, NST = BN == 256 ? 4 : 3;
    constexpr int WN = BN / 64, AI = 16 / NW, BI = BN / 16 / NW, PER = AI + BI;     // DMA instructions per wave per stage
    static_assert(NW / WN == 2, "two wave rows of 128 output rows each");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int num_n = g.N / BN;
    const int num_m = (g.M + BM - 1) / BM;
    const int b = blockIdx.x;
    const int grp = b / (8 * num_n), within = b % (8 * num_n);
    const int m_tile = grp * 8 + (within & 7), n_tile = within >> 3;
    if (m_tile >= num_m) return;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 15, q = lane >> 4;

    const char* asrc[AI];
    const char* bsrc[BI];
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int row = 16 * (AI * wave + j) + (lane >> 2);
        const int ch = (lane & 3) ^ nt_swz(row);
        const int m = min(m0 + row, g.M - 1);
        // (row gather: token m reads row a_rows[m] of A -- the per-token feature projection straight from the frozen table; the
        // row pointer is per lane anyway and lives in registers for the whole K loop)
        const int64_t arow = g.a_rows ? g.a_rows[m] : (int64_t)m;
        asrc[j] = (const char*)g.A + arow * g.lda * 2 + ch * 16;
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = 16 * (BI * wave + j) + (lane >> 2);
        const int ch = (lane & 3) ^ nt_swz(row);
        bsrc[j] = (const char*)g.B + (int64_t)(n0 + row) * g.ldb * 2 + ch * 16;
    }
    // (uniform) byte offset of logical k-step kt inside a row of A / B.  kmap_vc: the operands are head-major Q|K|V|C-shaped ([.., head, {q, k, v, c},
    // 32]) and only the V | C blocks take part (beta == 1: dQ = dK = 0 exactly) -- logical step kt is block 2 + (kt & 1) of head kt >> 1
    const bool kvc = g.kmap_vc != 0;
    auto koff = [&](int kt) { return (int64_t)(kvc ? ((kt >> 1) << 2) + 2 + (kt & 1) : kt) * ROWB; };
    auto issue = [&](int kt) {
        char* st = smem + (kt % NST) * STAGE;
        const int64_t ko = koff(kt);
#pragma unroll
        for (int j = 0; j < AI; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc[j] + ko),
                                             (lds_void_t*)(st + 16 * (AI * wave + j) * ROWB), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < BI; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(bsrc[j] + ko),
                                             (lds_void_t*)(st + BM * ROWB + 16 * (BI * wave + j) * ROWB), 16, 0, 0);
    };
    // one DMA instruction of a stage (piece 0 .. AI - 1: A rows, AI .. PER - 1: B rows)
    auto issue_piece = [&](int kt, int piece) __attribute__((always_inline)) {
        char* st = smem + (kt % NST) * STAGE;
#ifdef PMGT_NT_NO_A
        if (piece < AI && g.M > 0) return;      // (ablation builds: the A / the W stream is not fetched; results are garbage)
#endif
#ifdef PMGT_NT_NO_W
        if (piece >= AI && g.M > 0) return;
#endif
        if (piece < AI)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc[piece] + koff(kt)),
                                             (lds_void_t*)(st + 16 * (AI * wave + piece) * ROWB), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(bsrc[piece - AI] + koff(kt)),
                                             (lds_void_t*)(st + BM * ROWB + 16 * (BI * wave + piece - AI) * ROWB), 16, 0, 0);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void_t*)smem;
    uint32_t offa[8], offb[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ra = wm * 128 + i * 16 + r;
        offa[i] = (uint32_t)(ra * ROWB + ((q ^ nt_swz(ra)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rb = wn * 64 + j * 16 + r;
        offb[j] = (uint32_t)(BM * ROWB + rb * ROWB + ((q ^ nt_swz(rb)) << 4));
    }
    const int nk = g.K / 32;
#ifdef PMGT_TN_PROF
    unsigned int pacc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long plast = __builtin_readcyclecounter();
#define NT_STAMP(k_) do { const unsigned long long n_ = __builtin_readcyclecounter(); pacc[k_] += (unsigned int)(n_ - plast); plast = n_; } while (0)
#else
#define NT_STAMP(k_) do { } while (0)
#endif
    if constexpr (BN == 256) {
        // ---- main loop, rolling-prefetch form (no extra registers).  Per k-step the eight waves read 96 KB of fragments from
        // LDS (768 cycles of the LDS pipe) and run 1024 cycles of MFMA per SIMD; with all twelve reads of a step issued at its
        // top, both waves of a SIMD first wait for LDS and then queue for the matrix pipe (stamps: ~2000 cycles per step).
        // The MFMA operands are SWAPPED (D = W_frag x A_frag^T): acc[i][j][e] = out[16 i + r][16 j + 4 q + e], a lane owns four
        // CONSECUTIVE output columns of one row -- the epilogue below pairs two such blocks with v_permlane16_swap and stores 16
        // bytes per lane straight from the registers (or stages them with ds_write_b128 instead of four ds_write_b32).
        // Here the step is four QUADRANTS of 8 MFMAs (A halves A0 = fragments 0..3, A1 = 4..7; B halves B0 = 8, 9, B1 = 10, 11)
        // walked in snake order, and each half is re-loaded for the NEXT step as soon as its last quadrant has been issued:
        //     even step:  Q(A0,B0)  Q(A0,B1) | mid |  ld A0'  Q(A1,B1)  ld B1'  Q(A1,B0)  ld B0', A1'
        //     odd step:   Q(A0,B1)  Q(A0,B0) | mid |  ld A0'  Q(A1,B0)  ld B0'  Q(A1,B1)  ld B1', A1'
        // so every read is in flight for at least one quadrant (128 matrix-pipe cycles) before its first use and the LDS
        // traffic is spread over the whole step.  `mid` = lgkmcnt(0) [every read of stage kt has landed] -> vmcnt [own DMAs of
        // stage kt + 1] -> s_barrier -> DMA of stage kt + 4 into ring slot kt % 4 (free: all waves are past their last read
        // of stage kt).  One barrier per step as before; prefetch distance 4 on the 4-slot ring.
        u32x4 t[12];
        const uint32_t aA0 = lds_base + offa[0], aB0 = lds_base + offb[0];      // offa[i] = offa[0] + 1024 i, offb[j] = offb[0] + 1024 j
#define NT_LD_A(H, ADDR)                                                                                                  \
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\t"     \
                     "ds_read_b128 %3, %4 offset:%8"                                                                       \
                     : "=&v"(t[4 * (H)]), "=&v"(t[4 * (H) + 1]), "=&v"(t[4 * (H) + 2]), "=&v"(t[4 * (H) + 3])                  \
                     : "v"(ADDR), "n"(4096 * (H)), "n"(4096 * (H) + 1024), "n"(4096 * (H) + 2048), "n"(4096 * (H) + 3072)     \
                     : "memory")
#define NT_LD_B(H, ADDR)                                                                                                  \
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                                       \
                     : "=&v"(t[8 + 2 * (H)]), "=&v"(t[8 + 2 * (H) + 1])                                                     \
                     : "v"(ADDR), "n"(2048 * (H)), "n"(2048 * (H) + 1024)                                                   \
                     : "memory")
#define NT_QUAD(AH, BH)                                                                                                    \
        do {                                                                                                              \
            _Pragma("unroll") for (int i_ = 4 * (AH); i_ < 4 * (AH) + 4; ++i_)                                             \
                _Pragma("unroll") for (int j_ = 2 * (BH); j_ < 2 * (BH) + 2; ++j_)                                         \
                    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, t[8 + j_]),           \
                                                                          __builtin_bit_cast(bf16x8, t[i_]), acc[i_][j_], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
        } while (0)
        static_assert(PER == 4 && NST == 4, "vmcnt constants below: 4 DMA instructions per wave per stage, 4 ring slots");
        // (a static s_setprio(1) for the younger half of the workgroup only moves the 300-cycle lag to the other half: the two
        // waves of a SIMD share one matrix pipe)
#pragma unroll
        for (int kt = 0; kt < NST; ++kt)
            if (kt < nk) issue(kt);
        NT_STAMP(4);
        {   // stage 0 landed for everyone, then its twelve reads in the order an odd step leaves them: A0, B0, B1, A1
            const int younger = min(NST - 1, nk - 1);
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            NT_LD_A(0, aA0); NT_LD_B(0, aB0); NT_LD_B(1, aB0); NT_LD_A(1, aA0);
        }
        auto mid = [&](int kt) __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
            if (kt + 1 < nk) {
                const int younger = min(kt + 3, nk - 1) - (kt + 1);      // stages issued so far: 0 .. min(kt + 3, nk - 1)
                if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            NT_STAMP(0);
            __builtin_amdgcn_s_barrier();
            NT_STAMP(1);
            __builtin_amdgcn_sched_barrier(0);
        };
        // The four DMA instructions of stage kt + 4 go out ONE PER QUADRANT after the barrier of step kt (ring slot kt % 4 is
        // free from there on), the last one in the first half of step kt + 1: issued together right after the barrier they cost
        // every wave 380-630 cycles (stamps; the address path takes 64 B per clock per CU, i.e. 512 cycles for the 32 KB of a
        // stage) during which it issues no MFMA.  Every piece of a stage is still issued before the NEXT step's barrier, so the
        // vmcnt ladder above is unchanged.
#define NT_DMA(KT, PIECE)                                                 \
        do {                                                              \
            if ((KT) + NST < nk) issue_piece((KT) + NST, (PIECE));        \
            __builtin_amdgcn_sched_barrier(0);                            \
        } while (0)
        // two steps per loop iteration, straight-line (nk is even: K % 64 == 0 is a launch condition of this tile) -- with the
        // even / odd forms under an `if` the accumulators meet in phi nodes and the compiler stops updating them in place
        for (int kt = 0; kt < nk; kt += 2) {
            {
                const uint32_t so = (uint32_t)(((kt + 1) % NST) * STAGE);
                const uint32_t aA = aA0 + so, aB = aB0 + so;
                // entry: outstanding reads in issue order A0(4) B0(2) B1(2) A1(4)
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[8]), "+v"(t[9]));
                NT_QUAD(0, 0);
                if (kt > 0) NT_DMA(kt - 1, 3);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(t[10]), "+v"(t[11]));
                NT_QUAD(0, 1);
                mid(kt);
                NT_DMA(kt, 0);
                NT_LD_A(0, aA);                          // (kt + 1 < nk always: nk is even)
                NT_QUAD(1, 1);
                NT_DMA(kt, 1);
                NT_LD_B(1, aB);
                NT_QUAD(1, 0);
                NT_DMA(kt, 2);
                NT_LD_B(0, aB); NT_LD_A(1, aA);
            }
            {
                const bool more = kt + 2 < nk;          // (uniform)
                const uint32_t so = (uint32_t)(((kt + 2) % NST) * STAGE);
                const uint32_t aA = aA0 + so, aB = aB0 + so;
                // entry: outstanding reads in issue order A0(4) B1(2) B0(2) A1(4)
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[10]), "+v"(t[11]));
                NT_QUAD(0, 1);
                NT_DMA(kt, 3);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(t[8]), "+v"(t[9]));
                NT_QUAD(0, 0);
                mid(kt + 1);
                NT_DMA(kt + 1, 0);
                if (more) NT_LD_A(0, aA);
                NT_QUAD(1, 0);
                NT_DMA(kt + 1, 1);
                if (more) NT_LD_B(0, aB);
                NT_QUAD(1, 1);
                NT_DMA(kt + 1, 2);
                if (more) { NT_LD_B(1, aB); NT_LD_A(1, aA); }
            }
        }
#undef NT_LD_A
#undef NT_DMA
#undef NT_LD_B
#undef NT_QUAD
    } else {
#pragma unroll
    for (int kt = 0; kt < NST - 1; ++kt)
        if (kt < nk) issue(kt);
    NT_STAMP(4);
    for (int kt = 0; kt < nk; ++kt) {
        const int younger = min(NST - 2, nk - 1 - kt);
        if (younger == 2) { if constexpr (PER == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        else if (younger == 1) { if constexpr (PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NT_STAMP(0);
        __builtin_amdgcn_s_barrier();
        NT_STAMP(1);
        NT_STAMP(2);
        const uint32_t sbase = lds_base + (uint32_t)((kt % NST) * STAGE);
        // the four B fragments go out first, then the eight A fragments; the MFMAs of A-fragment i start as soon as it
        // has landed (LDS returns in order), so most of the LDS latency hides behind the matrix pipe
        u32x4 t[12];
        asm volatile(
            "ds_read_b128 %8, %20\n\t"
            "ds_read_b128 %9, %21\n\t"
            "ds_read_b128 %10, %22\n\t"
            "ds_read_b128 %11, %23\n\t"
            "ds_read_b128 %0, %12\n\t"
            "ds_read_b128 %1, %13\n\t"
            "ds_read_b128 %2, %14\n\t"
            "ds_read_b128 %3, %15\n\t"
            "ds_read_b128 %4, %16\n\t"
            "ds_read_b128 %5, %17\n\t"
            "ds_read_b128 %6, %18\n\t"
            "ds_read_b128 %7, %19"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
              "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11])
            : "v"(sbase + offa[0]), "v"(sbase + offa[1]), "v"(sbase + offa[2]), "v"(sbase + offa[3]),
              "v"(sbase + offa[4]), "v"(sbase + offa[5]), "v"(sbase + offa[6]), "v"(sbase + offa[7]),
              "v"(sbase + offb[0]), "v"(sbase + offb[1]), "v"(sbase + offb[2]), "v"(sbase + offb[3])
            : "memory");
        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]), "+v"(t[0]));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i == 1) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(t[1]));
            if (i == 2) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(t[2]));
            if (i == 3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(t[3]));
            if (i == 4) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(t[4]));
            if (i == 5) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(t[5]));
            if (i == 6) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(t[6]));
            if (i == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[7]));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, t[i]), __builtin_bit_cast(bf16x8, t[8 + j]),
                                                                    acc[i][j], 0, 0, 0);
            // the next stage's DMA goes out behind the MFMAs queued above (its ring slot was last read in step kt - 1, and
            // every wave has passed this step's barrier)
            if (i == 3 && kt + NST - 1 < nk) issue(kt + NST - 1);
        }
    }
    }
    NT_STAMP(3);
    __builtin_amdgcn_s_barrier();       // the ring becomes the staging buffer

    // ---- epilogue: four passes of 64 rows through an fp32 staging tile, then 32 lanes per row, 8 columns per lane
    const DropKey dk = make_drop_key(g.drop);
    T* Cp = (T*)g.C;
    const T* R = (const T*)g.res;
    T* AUX = (T*)g.aux;
    constexpr int ES = BN + 4;
    float* stage = (float*)smem;
    constexpr int LPR = BN / 8;                     // lanes per output row
    const int er = tid / LPR, ec = (tid % LPR) * 8;
    // residual rows travel one pass ahead (the workgroup is alone on its CU: a load issued where it is consumed costs a
    // full memory round trip per pass -- in-kernel timestamps put the epilogue at 30 % of the tile time)
    bf16x8 rv[2][4];
    auto load_res = [&](int pass, bf16x8 (&dst)[4]) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = min(m0 + pass * 64 + er + 16 * it, g.M - 1);
            dst[it] = *(const bf16x8*)(R + (int64_t)m * g.ldr + n0 + ec);
        }
    };
    constexpr bool SWAPPED = BN == 256;             // accumulator layout of the main loop above (see there)
    if constexpr (LNB) {
        // ---- dy = acc + residual (the direct epilogue's register layout: lane (r, q) owns 8 consecutive columns of row 16 i + r per
        // block pair) goes to LDS as a bf16 [256][256] image -- 128 KB, exactly the ring; 16-byte chunk c of row R sits in slot
        // c ^ (R & 31): sixteen consecutive rows of one chunk column (a ds_write_b128 of 16 lanes) hit sixteen different slots, and
        // a row read back by 32 lanes is one permuted 512-byte line.  (bf16: what the two-launch form rounds dy to on its way through HBM.)
        const int cb = ((q & 1) << 4) | ((q & 2) << 2);          // q = 0, 1, 2, 3 -> columns 0, 16, 8, 24 of the 32-column pair
        const int ncol = wn * 64 + cb;
        {
            u32x4 rr[8][2];
            if (R) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int m = min(m0 + wm * 128 + 16 * i + r, g.M - 1);
                    if (g.lnb_res_inv) m = g.lnb_res_inv[m];      // (uniform branch) compact residual: its row, or < 0 = none
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr)
                        rr[i][pr] = m >= 0 ? *(const u32x4*)(R + (int64_t)m * g.ldr + ncol + 32 * pr) : (u32x4){0u, 0u, 0u, 0u};
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = wm * 128 + 16 * i + r;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    f32x4 a = acc[i][2 * pr], b = acc[i][2 * pr + 1];
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                                 "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7"
                                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
                    if (R) {
                        const bf16x8 rv8 = __builtin_bit_cast(bf16x8, rr[i][pr]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a[e] += (float)rv8[e]; b[e] += (float)rv8[4 + e]; }
                    }
                    const bf16x8 o = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
                    const int ch = (ncol + 32 * pr) >> 3;
                    *(bf16x8*)(smem + row * 512 + ((ch ^ (row & 31)) << 4)) = o;
                }
            }
        }
        // ---- LayerNorm backward of the tile's rows: 32 lanes per row, 8 columns per lane, sixteen passes of 16 rows.  The rows of the
        // LayerNorm OUTPUT y and rstd travel eight passes ahead (requested before the barrier: the first ones arrive under the LDS round
        // trip), each slot refilled as soon as its pass has converted it.
        __builtin_amdgcn_sched_barrier(0);      // (the requests below must not be scheduled into the block above: its accumulators and residual rows fill the register file)
        const int erow = tid >> 5, ecl = (tid & 31) * 8;
        const char* Yp = (const char*)g.lnb_y;
        char* DX = (char*)g.C;
        char* DXD = (char*)g.lnb_dx_drop;
        const uint32_t ld2 = (uint32_t)g.ldc * 2u, ec2 = (uint32_t)ecl * 2u;      // equal leading dimensions, < 4 GB each (host)
        constexpr int PFD = 8;      // passes in flight (40 VGPRs)
        bf16x8 yv[PFD];
        float rs[PFD];
        auto load_y = [&](int p) __attribute__((always_inline)) {
            const uint32_t m = (uint32_t)min(m0 + 16 * p + erow, g.M - 1);
            yv[p % PFD] = *(const bf16x8*)(Yp + (m * ld2 + ec2));
            rs[p % PFD] = *(const float*)((const char*)g.lnb_stats + (m * 8u + 4u));
        };
#pragma unroll
        for (int p = 0; p < PFD; ++p) load_y(p);
        const DropKey lk = make_drop_key(g.lnb_drop);
        float gam[8], nbet[8], igam[8], dgam[8], dbet[8], dbia[8];      // x^ = y / gamma - beta / gamma
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gam[e] = g.lnb_gamma[ecl + e];
            igam[e] = gam[e] != 0.f ? __builtin_amdgcn_rcpf(gam[e]) : 0.f;      // (a dead channel: the host guard stores LayerNorm inputs instead, engine.py)
            nbet[e] = -g.lnb_beta[ecl + e] * igam[e];
            dgam[e] = dbet[e] = dbia[e] = 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int row = 16 * p + erow;
            const int m = m0 + row;
            const bool ok = m < g.M;
            const bf16x8 dyv = *(const bf16x8*)(smem + row * 512 + (((tid & 31) ^ (row & 31)) << 4));
            float v[8], xh[8], gg[8];
            float sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) xh[e] = fmaf((float)yv[p % PFD][e], igam[e], nbet[e]);
            const float rsp = rs[p % PFD];
            __builtin_amdgcn_sched_barrier(0);      // (the refill targets the registers converted above)
            if (p + PFD < 16) load_y(p + PFD);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = ok ? (float)dyv[e] : 0.f;      // rows past M (clamped operands): no term in any sum, nothing stored
                gg[e] = v[e] * gam[e];
                dgam[e] = fmaf(v[e], xh[e], dgam[e]);
                dbet[e] += v[e];
                sg += gg[e];
                sgx = fmaf(gg[e], xh[e], sgx);
            }
            sg = sum_lanes32(sg) * (1.f / 256.f);
            sgx = sum_lanes32(sgx) * (1.f / 256.f);
            float o[8];
            bf16x8 ob;
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = (gg[e] - sg - xh[e] * sgx) * rsp; ob[e] = (bf16)o[e]; }
            const uint32_t off = (uint32_t)m * ld2 + ec2;
            if (ok) *(bf16x8*)(DX + off) = ob;
            if (DXD) {      // (uniform)
                if (lk.on) {
                    float d0[4], d1[4];
                    drop_mul4(lk, (uint32_t)m, (uint32_t)ecl >> 2, d0);
                    drop_mul4(lk, (uint32_t)m, ((uint32_t)ecl >> 2) + 1, d1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] *= d0[e]; o[4 + e] *= d1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = (bf16)o[e];
                if (ok) *(bf16x8*)(DXD + off) = ob;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) dbia[e] += (float)ob[e];      // column sum of what the GEMMs behind it will read
            // the running sums are pinned here, pass by pass: the predicated stores above split the unrolled passes into basic blocks,
            // and the compiler otherwise SINKS all sixteen passes' terms into the block behind the last one (their operands alive, or
            // spilled, until then: 96 VGPRs of scratch)
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(dgam[e]), "+v"(dbet[e]), "+v"(dbia[e]));
            __builtin_amdgcn_sched_barrier(0);      // one pass at a time
        }
        __syncthreads();      // every wave is done with the dy image
        float* red = (float*)smem;      // [16][768]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[erow * 768 + ecl + e] = dgam[e];
            red[erow * 768 + 256 + ecl + e] = dbet[e];
            red[erow * 768 + 512 + ecl + e] = dbia[e];
        }
        __syncthreads();
        for (int idx = tid; idx < 768; idx += 512) {
            float a = 0.f;
#pragma unroll
            for (int rr2 = 0; rr2 < 16; ++rr2) a += red[rr2 * 768 + idx];
            g.lnb_part[(int64_t)m_tile * 768 + idx] = a;
        }
        return;
    }
    if constexpr (LNF) {
        // ---- x = bf16(dropout(acc + bias) + residual): the register layout of the direct epilogue (lane (r, q) owns 8 consecutive columns of
        // row 16 i + r per block pair), stored as C unless the backward takes x^ from the LayerNorm output, and written to LDS as the bf16
        // [256][256] image of the LayerNorm-backward form above (chunk c of row R in slot c ^ (R & 31))
        const int cb = ((q & 1) << 4) | ((q & 2) << 2);
        const int ncol = wn * 64 + cb;
        {
            u32x4 rr[8][2];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = min(m0 + wm * 128 + 16 * i + r, g.M - 1);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
                    rr[i][pr] = R ? *(const u32x4*)(R + (int64_t)m * g.ldr + ncol + 32 * pr) : (u32x4){0u, 0u, 0u, 0u};
            }
            f32x4 bv[2][2];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                bv[pr][0] = g.bias ? *(const f32x4*)(g.bias + ncol + 32 * pr) : (f32x4){0.f, 0.f, 0.f, 0.f};
                bv[pr][1] = g.bias ? *(const f32x4*)(g.bias + ncol + 32 * pr + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = wm * 128 + 16 * i + r;
                const int m = m0 + row;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    f32x4 a = acc[i][2 * pr], b = acc[i][2 * pr + 1];
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                                 "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7"
                                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
                    a += bv[pr][0];
                    b += bv[pr][1];
                    if (dk.on) {
                        float d0[4], d1[4];
                        drop_mul4(dk, (uint32_t)m, (uint32_t)(ncol + 32 * pr) >> 2, d0);
                        drop_mul4(dk, (uint32_t)m, ((uint32_t)(ncol + 32 * pr) >> 2) + 1, d1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a[e] *= d0[e]; b[e] *= d1[e]; }
                    }
                    const bf16x8 rv8 = __builtin_bit_cast(bf16x8, rr[i][pr]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] += (float)rv8[e]; b[e] += (float)rv8[4 + e]; }
                    const bf16x8 o = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
                    const int ch = (ncol + 32 * pr) >> 3;
                    *(bf16x8*)(smem + row * 512 + ((ch ^ (row & 31)) << 4)) = o;
                    if (!g.lnf_skip_c && m < g.M) *(bf16x8*)(Cp + (int64_t)m * g.ldc + ncol + 32 * pr) = o;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- LayerNorm of the tile's rows: 32 lanes per row, 8 columns per lane, sixteen passes of 16 rows; statistics exactly as the streaming
        // kernels compute them (gemm_ws.hip: mean, centred sum of squares, v_rsq_f32) on the bf16-rounded row the backward re-derives
        const int erow = tid >> 5, ecl = (tid & 31) * 8;
        float gam[8], bet[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { gam[e] = g.lnf_gamma[ecl + e]; bet[e] = g.lnf_beta[ecl + e]; }
        T* LNO = (T*)g.lnf_out;
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < 16; ++p) {
            const int row = 16 * p + erow;
            const int m = m0 + row;
            const bf16x8 xv = *(const bf16x8*)(smem + row * 512 + (((tid & 31) ^ (row & 31)) << 4));
            float v[8], sacc = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = (float)xv[e]; sacc += v[e]; }
            const float mean = sum_lanes32(sacc) * (1.f / 256.f);
            float ss = 0.f, tc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { tc[e] = v[e] - mean; ss = fmaf(tc[e], tc[e], ss); }
            ss = sum_lanes32(ss);
            const float rstd = __builtin_amdgcn_rsqf(ss * (1.f / 256.f) + g.lnf_eps);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)(tc[e] * rstd * gam[e] + bet[e]);
            if (m < g.M) {
                *(bf16x8*)(LNO + (int64_t)m * g.ldc + ecl) = o;
                if ((tid & 31) == 0) *(float2*)(g.lnf_stats + 2 * (int64_t)m) = make_float2(mean, rstd);
            }
        }
        return;
    }
    if constexpr (SWAPPED) {
        if (g.epi == EPI_NONE && !dk.on) {
            // ---- direct epilogue (bias / residual only: the data-gradient GEMMs): no LDS staging, no barriers.  Blocks (j, j + 1)
            // of a row are paired with v_permlane16_swap (fp32), after which lane (r, q) owns 8 consecutive columns of row
            // 16 i + r: one 16-byte residual load and one 16-byte store per lane, 64 contiguous bytes per row and instruction.
            // The staged form cost 15.5k of a tile's 74k cycles (stamps), with one workgroup per CU nothing overlaps it.
            const int cb = ((q & 1) << 4) | ((q & 2) << 2);          // q = 0, 1, 2, 3 -> columns 0, 16, 8, 24 of the 32-column pair
            const int ncol = n0 + wn * 64 + cb;
            u32x4 rr[8][2];
            if (R) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = min(m0 + wm * 128 + 16 * i + r, g.M - 1);
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) rr[i][pr] = *(const u32x4*)(R + (int64_t)m * g.ldr + ncol + 32 * pr);
                }
            }
            f32x4 bv[2][2];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                bv[pr][0] = g.bias ? *(const f32x4*)(g.bias + ncol + 32 * pr) : (f32x4){0.f, 0.f, 0.f, 0.f};
                bv[pr][1] = g.bias ? *(const f32x4*)(g.bias + ncol + 32 * pr + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + wm * 128 + 16 * i + r;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    f32x4 a = acc[i][2 * pr], b = acc[i][2 * pr + 1];
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                                 "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7"
                                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
                    a += bv[pr][0];
                    b += bv[pr][1];
                    if (R) {
                        const bf16x8 rv8 = __builtin_bit_cast(bf16x8, rr[i][pr]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a[e] += (float)rv8[e]; b[e] += (float)rv8[4 + e]; }
                    }
                    const bf16x8 o = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
                    if (m < g.M) *(bf16x8*)(Cp + (int64_t)m * g.ldc + ncol + 32 * pr) = o;
                }
            }
#ifdef PMGT_TN_PROF
            NT_STAMP(5);
            if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133))
                for (int k_ = 0; k_ < 6; ++k_) g_nt_prof[blockIdx.x == 0 ? 0 : 1][wave][k_] = pacc[k_];
#endif
            return;
        }
    }
    if (R) load_res(0, rv[0]);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        if (R && pass < 3) load_res(pass + 1, rv[(pass + 1) & 1]);
        if (wm == (pass >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (SWAPPED) {
                        *(f32x4*)(stage + (ii * 16 + r) * ES + wn * 64 + j * 16 + 4 * q) = acc[(pass & 1) * 4 + ii][j];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            stage[(ii * 16 + 4 * q + e) * ES + wn * 64 + j * 16 + r] = acc[(pass & 1) * 4 + ii][j][e];
                    }
                }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = er + 16 * it;
            const int m = m0 + pass * 64 + row;
            const int n = n0 + ec;
            if (m < g.M) {
                const f32x4 s0 = *(const f32x4*)(stage + row * ES + ec), s1 = *(const f32x4*)(stage + row * ES + ec + 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = s0[e]; v[4 + e] = s1[e]; }
                if (g.bias) {
                    const f32x4 b0 = *(const f32x4*)(g.bias + n), b1 = *(const f32x4*)(g.bias + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                }
                if (g.epi == EPI_GELU) {
                    bf16x8 pre;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { pre[e] = (bf16)v[e]; v[e] = gelu_fast((float)pre[e]); }
                    *(bf16x8*)(AUX + (int64_t)m * g.ldaux + n) = pre;
                } else if (g.epi == EPI_GELU_GRAD) {
                    const bf16x8 pre = *(const bf16x8*)(AUX + (int64_t)m * g.ldaux + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= gelu_fast_grad((float)pre[e]);
                }
                if (dk.on) {
                    float d0[4], d1[4];
                    drop_mul4(dk, (uint32_t)m, (uint32_t)n >> 2, d0);
                    drop_mul4(dk, (uint32_t)m, ((uint32_t)n >> 2) + 1, d1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] *= d0[e]; v[4 + e] *= d1[e]; }
                }
                if (R) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[pass & 1][it][e];
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                *(bf16x8*)(Cp + (int64_t)m * g.ldc + n) = o;
            }
        }
        __syncthreads();
    }
#ifdef PMGT_TN_PROF
    NT_STAMP(5);
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133))
        for (int k_ = 0; k_ < 6; ++k_) g_nt_prof[blockIdx.x == 0 ? 0 : 1][wave][k_] = pacc[k_];
#endif
}

static bool nt_big_ok(const GemmNT& g) {
    if (g.kmap_vc && !(g.N == 256 && g.K % 64 == 0 && g.lda >= 2 * g.K && g.ldb >= 2 * g.K)) return false;      // V | C-only k-steps: the 256-wide tile only
    // fewer than ~96 tiles of 256 rows leave most of the 256 CUs idle: the 128 x 128 tile then wins (dX = dQKVC W at B = 32 targets,
    // M = 12 288, K = 1 024, N = 256: 48 big tiles 32.5 us, 192 small ones 22.6 us; equal at M = 24 576)
    if ((int64_t)cdiv(g.M, 256) * cdiv(g.N, 256) < 96) return false;
    return !(g.opts & OPT_TILE_GEMM) && g.m_dev == nullptr && !g.res_gather && g.M >= 4096 && g.N % 128 == 0 && g.K % 32 == 0 &&
           g.K >= 128 && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.ldc % 8 == 0 && (g.res == nullptr || g.ldr % 8 == 0) &&
           (g.aux == nullptr || g.ldaux % 8 == 0) && ((uintptr_t)g.C % 16) == 0 && (g.res == nullptr || ((uintptr_t)g.res % 16) == 0) &&
           (g.aux == nullptr || ((uintptr_t)g.aux % 16) == 0) && (g.bias == nullptr || ((uintptr_t)g.bias % 16) == 0);
}

bool gemm_nt_big_applies(const GemmNT& g) { return nt_big_ok(g) && (!g.kmap_vc || (g.N % 256 == 0 && g.K % 64 == 0)); }

// ---- dy = A W^T + res (K > 512: dX = dQKVC W) followed by the LayerNorm backward of dy in the same launch
int gemm_nt_lnb_parts(int M) { return cdiv(M, 256); }
bool gemm_nt_lnb_ok(const GemmNT& g) {
    return nt_big_ok(g) && !(g.opts & OPT_UNFUSED_LN_BWD) && g.N == 256 && g.K % 64 == 0 && g.a_rows == nullptr && g.epi == EPI_NONE && g.bias == nullptr &&
           g.drop.p == 0.f && g.lnb_y != nullptr && g.lnb_stats != nullptr && g.lnb_gamma != nullptr && g.lnb_beta != nullptr &&
           (g.res == nullptr || g.ldr == g.ldc) && (g.lnb_res_inv == nullptr || g.res != nullptr) && g.lnb_ldy == g.ldc && (g.lnb_dx_drop == nullptr || g.lnb_lddx == g.ldc) &&
           (int64_t)g.M * g.ldc * 2 < (int64_t)1 << 32 && ((uintptr_t)g.lnb_y % 16) == 0 && ((uintptr_t)g.lnb_dx_drop % 16) == 0 &&
           ((uintptr_t)g.lnb_stats % 8) == 0 && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0;
}
int gemm_nt_lnb(const GemmNT& g, hipStream_t st) {
    PMGT_CHECK(gemm_nt_lnb_ok(g) && g.lnb_part != nullptr, -2, "gemm_nt_lnb: unsupported shape / epilogue M=%d N=%d K=%d", g.M, g.N, g.K);
    constexpr int smem = 4 * (256 + 256) * 64;
    PMGT_SMEM_ATTR(((const void*)gemm_nt_big_kernel<256, 8, 1>), smem);
    note_launch(LT_NT_LNB);
    if (g.kmap_vc) note_launch(LT_NT_VC);
    hipLaunchKernelGGL((gemm_nt_big_kernel<256, 8, 1>), dim3(cdiv(cdiv(g.M, 256), 8) * 8), dim3(512), smem, st, g);
    PMGT_LAUNCH_OK();
    return 0;
}

// ---- C = dropout(A W^T + bias) + res, ln_out = LayerNorm(C) in the same launch (K > 512, N = 256: FFN2 forward at I = 4d)
bool gemm_nt_lnf_shape(int M, int N, int K) {
    // (K <= 512 belongs to the weight-stationary streaming kernels; fewer than 96 tiles: see nt_big_ok)
    return N == 256 && K > 512 && K % 64 == 0 && M >= 4096 && cdiv(M, 256) >= 96 && (int64_t)M * N * 2 < (int64_t)1 << 32;
}
bool gemm_nt_lnf_ok(const GemmWS& g) {
    return gemm_nt_lnf_shape(g.M, g.N, g.K) && nt_big_ok(g) && !(g.opts & OPT_UNFUSED_LN) && g.ln_out != nullptr && g.ln_stats != nullptr &&
           g.ln_gamma != nullptr && g.ln_beta != nullptr && g.q8 == nullptr && g.a_rows == nullptr && g.epi == EPI_NONE && g.res != nullptr &&
           g.ldr % 8 == 0 && ((uintptr_t)g.ln_out % 16) == 0 && ((uintptr_t)g.ln_stats % 8) == 0 && ((uintptr_t)g.A % 16) == 0 &&
           ((uintptr_t)g.B % 16) == 0 && ((uintptr_t)g.ln_gamma % 16) == 0 && ((uintptr_t)g.ln_beta % 16) == 0;
}
int gemm_nt_lnf(const GemmWS& gw, hipStream_t st) {
    PMGT_CHECK(gemm_nt_lnf_ok(gw), -2, "gemm_nt_lnf: unsupported shape / epilogue M=%d N=%d K=%d", gw.M, gw.N, gw.K);
    GemmNT g = gw;
    g.lnf_out = gw.ln_out; g.lnf_stats = gw.ln_stats; g.lnf_gamma = gw.ln_gamma; g.lnf_beta = gw.ln_beta; g.lnf_eps = gw.ln_eps;
    g.lnf_skip_c = gw.skip_c;
    constexpr int smem = 4 * (256 + 256) * 64;
    PMGT_SMEM_ATTR(((const void*)gemm_nt_big_kernel<256, 8, 2>), smem);
    note_launch(LT_NT_LNF);
    hipLaunchKernelGGL((gemm_nt_big_kernel<256, 8, 2>), dim3(cdiv(cdiv(g.M, 256), 8) * 8), dim3(512), smem, st, g);
    PMGT_LAUNCH_OK();
    return 0;
}

template <typename T> int gemm_nt(const GemmNT& g, hipStream_t st) {
    constexpr int EPC = 16 / sizeof(T);
    if (g.M <= 0 || g.N <= 0) return 0;
    PMGT_CHECK(g.K > 0 && g.K % EPC == 0, -2, "gemm_nt: K=%d must be a positive multiple of %d", g.K, EPC);
    PMGT_CHECK(g.lda % EPC == 0 && g.ldb % EPC == 0, -2, "gemm_nt: lda=%lld ldb=%lld must be multiples of %d",
               (long long)g.lda, (long long)g.ldb, EPC);
    PMGT_CHECK(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0, -2, "gemm_nt: operands must be 16-byte aligned");
    PMGT_CHECK(g.lda >= g.K && g.ldb >= g.K && g.ldc >= g.N, -2, "gemm_nt: leading dimensions too small");
    PMGT_CHECK(g.epi == EPI_NONE || g.aux != nullptr, -2, "gemm_nt: epilogue %d needs aux", g.epi);
    PMGT_CHECK(g.N % 4 == 0 && g.ldc % 4 == 0 && (g.res == nullptr || g.ldr % 4 == 0) && (g.aux == nullptr || g.ldaux % 4 == 0),
               -2, "gemm_nt: N and the output leading dimensions must be multiples of 4 (N=%d ldc=%lld)", g.N, (long long)g.ldc);
    PMGT_CHECK(!g.kmap_vc || (sizeof(T) == 2 && gemm_nt_big_applies(g)), -2, "gemm_nt: the V | C-only k-step map needs the 256 x 256 tile (M=%d N=%d K=%d)", g.M, g.N, g.K);
    constexpr int BM = 128, BN = 128;
    const int num_m = cdiv(g.M, BM), num_n = cdiv(g.N, BN);
    const int grid = cdiv(num_m, 8) * 8 * num_n;
    if constexpr (sizeof(T) == 2) {
        if (nt_big_ok(g)) {
            const int nm = cdiv(g.M, 256);
            if (g.N % 256 == 0 && g.K % 64 == 0) {      // (K % 64: the 256-wide tile walks k-steps in pairs)
                constexpr int smem = 4 * (256 + 256) * 64;
                PMGT_SMEM_ATTR(((const void*)gemm_nt_big_kernel<256, 8>), smem);
                note_launch(g.a_rows ? LT_NT_BIG_GATHER : LT_NT_BIG);
                if (g.kmap_vc) note_launch(LT_NT_VC);
                hipLaunchKernelGGL((gemm_nt_big_kernel<256, 8>), dim3(cdiv(nm, 8) * 8 * (g.N / 256)), dim3(512), smem, st, g);
            } else {
                constexpr int smem = 3 * (256 + 128) * 64;
                PMGT_SMEM_ATTR(((const void*)gemm_nt_big_kernel<128, 4>), smem);
                note_launch(LT_NT_BIG_128);
                hipLaunchKernelGGL((gemm_nt_big_kernel<128, 4>), dim3(cdiv(nm, 8) * 8 * (g.N / 128)), dim3(256), smem, st, g);
            }
            PMGT_LAUNCH_OK();
            return 0;
        }
    }
    note_launch(LT_NT_TILE);
    hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN>), dim3(grid), dim3(256), 0, st, g);
    PMGT_LAUNCH_OK();
    return 0;
}
template int gemm_nt<float>(const GemmNT&, hipStream_t);
template int gemm_nt<bf16>(const GemmNT&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// TN (weight gradients)
// ------------------------------------------------------------------------------------------------
// A device-side row count (the compacted last layer, the masked NFR rows): the host sizes the row chunks of the splits for the CAPACITY of the
// buffer, which leaves all live rows to the first few splits (wgrad_nfr at c2: 5 000 live of 31 744 rows -> two of seven splits busy, 75 us).
// Re-derived here from the live count, every split works; the partial slabs are summed in split order either way.
__device__ __forceinline__ int tn_live_chunk(const GemmTN& g, int Mlim, int chunk_rows, int gran) {
    return g.m_dev ? ((Mlim + g.splits - 1) / g.splits + gran - 1) / gran * gran : chunk_rows;
}

__device__ __forceinline__ int tn_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename T> int gemm_tn_bkm() { return sizeof(T) == 2 ? 64 : 32; }
template int gemm_tn_bkm<float>();
template int gemm_tn_bkm<bf16>();

// Q8 (fp8 mode, feature-projection weight gradient): the rows of Q are e4m3 bytes of a frozen feature table (value = byte *
// g.q_scale); a staging chunk is 8 bytes instead of 16 and is widened to bf16 on its way into LDS, so the gather moves
// half the bytes and the MFMA loop is the bf16 one.
template <typename T, bool Q8 = false>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN g, int chunk_rows) {
    static_assert(!Q8 || sizeof(T) == 2, "e4m3 Q operand: bf16 path only");
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BKM = sizeof(T) == 2 ? 64 : 32;    // reduction rows per step
    constexpr int ROWB = 128 * sizeof(T);            // LDS bytes per tile row (128 columns)
    constexpr int CPR = ROWB / 16;                   // 16-B chunks per row
    constexpr int TILEB = BKM * ROWB;                // 16 KiB either way
    constexpr int LPT = BKM * CPR / 256;             // chunks per thread per tile (= 4)
    __shared__ __attribute__((aligned(16))) char smem[4 * TILEB];
    char* sP = smem;
    char* sQ = smem + 2 * TILEB;

    // 1-D grid: ids b and b + 8 share an XCD (and its L2).  All tiles of one row-chunk ("split") are
    // placed on one XCD so the chunk's P and Q rows are fetched from HBM once and re-read from L2.
    const int tn1 = (g.N1 + 127) / 128, tn2 = (g.N2 + 127) / 128, tiles = tn1 * tn2;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tile = idx % tiles, split = xcd + 8 * (idx / tiles);
    if (split >= g.splits) return;
    const int n1_0 = (tile / tn2) * 128, n2_0 = (tile % tn2) * 128;
    const int Mlim = g.m_dev ? min(g.M, *g.m_dev) : g.M;
    chunk_rows = tn_live_chunk(g, Mlim, chunk_rows, 64);
    const int mbeg = split * chunk_rows, mend = min(Mlim, mbeg + chunk_rows);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // column sums of P (= bias gradient of the layer) ride along as one extra MFMA per tile against an
    // all-ones B operand, in the blocks of the first N2 tile
    const bool do_bias = g.bias_slab != nullptr && n2_0 == 0 && wn == 0;
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Staging loads are UNCONDITIONAL on clamped addresses (zero-selected afterwards): a branch around a load
    // makes the compiler wait at the join and serialises the memory round trips.  With a row gather on Q the
    // indices of the NEXT K-step are fetched one step ahead, so the Q loads never wait on an index load.
    u32x4 rp[LPT], rq[LPT];
    int64_t qidx[LPT];
    auto iload = [&](int mb) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int row = (tid + 256 * i) / CPR;
            const int mc = max(min(mb + row, mend - 1), 0);
            qidx[i] = g.q_rows ? g.q_rows[mc] : (int64_t)mc;
        }
    };
    auto gload = [&](int mb) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / CPR, ch = idx % CPR;
            const int m = mb + row;
            const int cp = n1_0 + ch * EPC, cq = n2_0 + ch * EPC;
            const bool okm = m < mend;
            const int mc = max(min(m, mend - 1), 0);
            const int cpc = min(cp, g.N1 - EPC), cqc = min(cq, g.N2 - EPC);
            (void)okm;      // the zero-select for padding rows/columns happens at LDS-store time (sstore), so the
                            // loads stay in flight across the MFMA phase instead of being waited for here
            rp[i] = *(const u32x4*)((const char*)g.P + ((int64_t)mc * g.ldp + cpc) * (int64_t)sizeof(T));
            if constexpr (Q8) {
                const u32x2 w = *(const u32x2*)((const char*)g.Q + qidx[i] * g.ldq + cqc);
                rq[i] = (u32x4){w[0], w[1], 0u, 0u};
            } else {
                rq[i] = *(const u32x4*)((const char*)g.Q + (qidx[i] * g.ldq + cqc) * (int64_t)sizeof(T));
            }
        }
    };
    // 8 e4m3 bytes -> 8 bf16 (times the table scale)
    auto widen = [&](u32x4 w) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[0], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[0], true);
        const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[1], false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[1], true);
        const float sc = g.q_scale;
        const bf16x8 o = {(bf16)(a[0] * sc), (bf16)(a[1] * sc), (bf16)(b[0] * sc), (bf16)(b[1] * sc),
                          (bf16)(c[0] * sc), (bf16)(c[1] * sc), (bf16)(d[0] * sc), (bf16)(d[1] * sc)};
        return __builtin_bit_cast(u32x4, o);
    };
    auto sstore = [&](int buf, int mb) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / CPR, ch = idx % CPR;
            int off;
            if constexpr (sizeof(T) == 2) off = row * ROWB + ((ch ^ (tn_f(row) << 1)) << 4);
            else off = row * ROWB + (ch << 4);
            const bool okm = mb + row < mend;
            const u32x4 z = {0, 0, 0, 0};
            *(u32x4*)(sP + buf * TILEB + off) = (okm && n1_0 + ch * EPC < g.N1) ? rp[i] : z;
            u32x4 qv = rq[i];
            if constexpr (Q8) qv = widen(qv);
            *(u32x4*)(sQ + buf * TILEB + off) = (okm && n2_0 + ch * EPC < g.N2) ? qv : z;
        }
    };

    const int nk = mend > mbeg ? (mend - mbeg + BKM - 1) / BKM : 0;
    if (nk > 0) {
        iload(mbeg);
        gload(mbeg);
        if (nk > 1) iload(mbeg + BKM);
        sstore(0, mbeg);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            gload(mbeg + (kt + 1) * BKM);                       // uses the indices fetched one step ago
            if (kt + 2 < nk) iload(mbeg + (kt + 2) * BKM);
        }
        const char* p_base = sP + buf * TILEB;
        const char* q_base = sQ + buf * TILEB;
        if constexpr (sizeof(T) == 2) {
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // ds_read_b64_tr_b16: the 16 lanes of group q read a 4(row) x 16(col) block and each lane
                // receives one column (4 rows).  Lane i of the group supplies row (i >> 2), cols 4*(i & 3).
                const int row = 32 * kk + 8 * q + (r >> 2);
                const int sw = tn_f(row) << 1;                 // same for row and row + 4
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = wm * 64 + i * 16 + 4 * (r & 3);
                    const int off = row * ROWB + (((col >> 3) ^ sw) << 4) + ((col & 7) << 1);
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p_base + off));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p_base + off + 4 * ROWB));
                    fa[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = wn * 64 + j * 16 + 4 * (r & 3);
                    const int off = row * ROWB + (((col >> 3) ^ sw) << 4) + ((col & 7) << 1);
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q_base + off));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q_base + off + 4 * ROWB));
                    fb[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if (do_bias) {
                    const bf16 one = (bf16)1.f;
                    const bf16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
                    for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < BKM / 4; ++s) {
                const int row = 4 * s + q;
                float fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *(const float*)(p_base + row * ROWB + (wm * 64 + i * 16 + r) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *(const float*)(q_base + row * ROWB + (wn * 64 + j * 16 + r) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], 1.f, accb[i], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < nk) sstore(buf ^ 1, mbeg + (kt + 1) * BKM);
        __syncthreads();
    }

    if (do_bias && r == 0) {        // every column of accb holds the same sums; lane column 0 writes
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n1 = n1_0 + wm * 64 + i * 16 + 4 * q + e;
                if (n1 < g.N1) g.bias_slab[(int64_t)split * g.N1 + n1] = accb[i][e];
            }
    }
    float* out = g.slab + (int64_t)split * g.N1 * g.N2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n2 = n2_0 + wn * 64 + j * 16 + r;
            if (n2 >= g.N2) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n1 = n1_0 + wm * 64 + i * 16 + 4 * q + e;
                if (n1 < g.N1) out[(int64_t)n1 * g.N2 + n2] = acc[i][j][e];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// TN, bf16, LDS-DMA pipeline.  Same tiling and fragment code as gemm_tn_kernel, but the operand tiles go
// global -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave-instruction, no staging registers, no
// ds_write), four 32-row stages in a ring, THREE stages in flight.  The register-staged kernel has at most
// one K-step of loads in flight per workgroup and spends most of its time waiting for them; this one keeps
// 48 KiB per workgroup in flight behind counted s_waitcnt vmcnt(N) and a raw s_barrier per stage:
//     wait own DMAs of stage kt  ->  s_barrier (everyone's landed; everyone finished reading stage kt-1)
//     ->  issue stage kt+3 into the ring slot stage kt-1 used  ->  tr-reads + MFMAs of stage kt.
// An LDS-DMA destination is lane-linear (wave base + 16 * lane), so the bank-conflict swizzle of the
// ds_read_b64_tr_b16 image is applied to the per-lane SOURCE address (chunk ^ swz(row)); rows / columns
// outside the matrix read from a zero page.  No row gather (the gather variant stays register-staged).
// ------------------------------------------------------------------------------------------------

// GATHER = true: rows of Q come through g.q_rows (the feature tables of the feature-projection wgrad).  The 32 row
// indices of a stage are themselves fetched by LDS-DMA (one 4-byte-per-lane instruction, issued six stages ahead
// by every wave into the same 256-byte slot of an 8-slot ring) and read back next to the fragments, three stages
// before the data DMA that needs them: no register-destination load ever enters the counted vmcnt sequence.
// NW = 8 (non-gather form; round 3): the same 128 x 128 tile on EIGHT waves (2 x 4, 64 x 32 outputs each).  With four waves -- one per
// SIMD -- a wave walks through "wait for its DMA, barrier, issue 4 DMA pieces (~70 cycles each), 16 transposed reads + lgkmcnt(0), 16
// MFMAs" strictly in sequence and nobody else uses the SIMD meanwhile: ~670 cycles per 32-row stage against 256 of MFMA, 3.5 TB/s on
// HBM bytes the Infinity Cache has already cut to 0.7 of the algorithmic ones.  Two waves per SIMD (2 DMA pieces, 12 reads, 8 MFMAs per
// stage each) fill each other's waits.
template <bool GATHER, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_tn_dma_kernel(GemmTN g, int chunk_rows) {
    static_assert(NW == 4 || (NW == 8 && !GATHER), "8 waves: non-gather form only");
    constexpr int WN = NW / 2, JN = 8 / WN, NJ = 8 / NW, RPW = 32 / NW, PER = 2 * NJ;     // wave grid 2 x WN, 64 x 16 JN outputs per wave
    constexpr int BKM = 32, ROWB = 256, STAGE = 2 * BKM * ROWB, NST = 4;     // 16 KiB per stage (P + Q)
    constexpr int IDX_SLOTS = 8, IDX_BYTES = BKM * 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];              // NST * STAGE (+ IDX_SLOTS * IDX_BYTES)
    const int tn1 = (g.N1 + 127) / 128, tn2 = (g.N2 + 127) / 128, tiles = tn1 * tn2;
    const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
    const int tile = bidx % tiles, split = xcd + 8 * (bidx / tiles);
    if (split >= g.splits) return;
    const int n1_0 = (tile / tn2) * 128, n2_0 = (tile % tn2) * 128;
    const int Mlim = g.m_dev ? min(g.M, *g.m_dev) : g.M;
    chunk_rows = tn_live_chunk(g, Mlim, chunk_rows, 64);
    const int mbeg = split * chunk_rows, mend = min(Mlim, mbeg + chunk_rows);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 15, q = lane >> 4;

    f32x4 acc[4][JN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = g.bias_slab != nullptr && n2_0 == 0 && wn == 0;
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane's part of a stage: rows RPW wave + 4 j + (lane >> 4), j < NJ; LDS chunk slot lane & 15 holds
    // global chunk (lane & 15) ^ swz(row)
    const char* zero = (const char*)g.zeros;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void_t*)smem;
    char* idx_ring = smem + NST * STAGE;
    auto issue = [&](int kt, uint32_t q0, uint32_t q1) {
        const int mb = mbeg + kt * BKM;
        char* st = smem + (kt & (NST - 1)) * STAGE;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = RPW * wave + 4 * j + (lane >> 4);
            const int ch = (lane & 15) ^ (tn_f(row) << 1);
            const int m = mb + row;
            const int cp = n1_0 + ch * 8, cq = n2_0 + ch * 8;
            const int64_t qrow = GATHER ? (int64_t)(j == 0 ? q0 : q1) : (int64_t)m;
            const char* sp = (m < mend && cp < g.N1) ? (const char*)g.P + ((int64_t)m * g.ldp + cp) * 2 : zero;
            const char* sq = (m < mend && cq < g.N2) ? (const char*)g.Q + (qrow * g.ldq + cq) * 2 : zero;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)sp, (lds_void_t*)(st + (RPW * wave + 4 * j) * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)sq, (lds_void_t*)(st + BKM * ROWB + (RPW * wave + 4 * j) * ROWB), 16, 0, 0);
        }
    };
    // index DMA of stage kt: lane l fetches dword l of q_rows[mb .. mb + 32) (clamped inside the array)
    auto issue_idx = [&](int kt) {
        const int64_t m = min((int64_t)mbeg + (int64_t)kt * BKM + (lane >> 1), (int64_t)g.M - 1);
        const char* src = (const char*)(g.q_rows + m) + 4 * (lane & 1);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(idx_ring + (kt & (IDX_SLOTS - 1)) * IDX_BYTES), 4, 0, 0);
    };
    const uint32_t idx_off = (uint32_t)((8 * wave + (lane >> 4)) * 8);      // row of j = 0; j = 1 is 4 rows (32 B) further

    const int nk = mend > mbeg ? (mend - mbeg + BKM - 1) / BKM : 0;
    if constexpr (GATHER) {
        if (nk > 0) {
#pragma unroll
            for (int kt = 0; kt < 6; ++kt) issue_idx(kt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                uint32_t q0, q1;
                const uint32_t ia = lds_base + (uint32_t)(NST * STAGE + kt * IDX_BYTES) + idx_off;
                asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:32\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(q0), "=&v"(q1) : "v"(ia) : "memory");
                if (kt < nk) issue(kt, q0, q1);
            }
        }
    } else {
        if (nk > 0) issue(0, 0, 0);
        if (nk > 1) issue(1, 0, 0);
        if (nk > 2) issue(2, 0, 0);
    }
    // per-lane byte offsets of the 4 + 4 fragment reads inside a stage (row = 8 q + (r >> 2); +4 rows = +1024 B)
    uint32_t offa[4], offb[4];
    {
        const int row = 8 * q + (r >> 2);
        const int sw = tn_f(row) << 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ca = wm * 64 + i * 16 + 4 * (r & 3), cb = wn * (16 * JN) + (i % JN) * 16 + 4 * (r & 3);
            offa[i] = (uint32_t)(row * ROWB + (((ca >> 3) ^ sw) << 4) + ((ca & 7) << 1));
            offb[i] = (uint32_t)(row * ROWB + (((cb >> 3) ^ sw) << 4) + ((cb & 7) << 1));
        }
    }
    for (int kt = 0; kt < nk; ++kt) {
        // 4 DMA instructions per wave per stage; leave the younger stages in flight.  GATHER adds one index DMA per
        // iteration (issued before the data DMAs of that iteration, for every kt, also past the end: clamped reads):
        // younger-than-D(kt) operations = 4 * younger stages + the index DMAs of the last min(kt, 2) iterations.
        const int younger = min(2, nk - 1 - kt);
        if constexpr (GATHER) {
            switch (4 * younger + min(kt, 2)) {
                case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            if constexpr (PER == 4) {
                if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (younger == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (GATHER) issue_idx(kt + 6);
        else { if (kt + 3 < nk) issue(kt + 3, 0, 0); }
        // The fragment reads are inline asm on purpose: the compiler treats an in-flight LDS-DMA as a pending
        // store to LDS and would put s_waitcnt vmcnt(0) in front of any ds_read it can see, draining the ring.
        // Ordering is ours: the counted vmcnt + s_barrier above retire stage kt; reads and their lgkmcnt(0) wait
        // sit in ONE statement with early-clobber outputs (cdna_hip_programming.md section 5.7, form (i)).
        const uint32_t sbase = lds_base + (uint32_t)((kt & (NST - 1)) * STAGE);
        u32x2 t[16];
        if constexpr (NW == 8) {      // four P fragments, two Q fragments
            asm volatile(
                "ds_read_b64_tr_b16 %0, %12\n\t"
                "ds_read_b64_tr_b16 %1, %12 offset:1024\n\t"
                "ds_read_b64_tr_b16 %2, %13\n\t"
                "ds_read_b64_tr_b16 %3, %13 offset:1024\n\t"
                "ds_read_b64_tr_b16 %4, %14\n\t"
                "ds_read_b64_tr_b16 %5, %14 offset:1024\n\t"
                "ds_read_b64_tr_b16 %6, %15\n\t"
                "ds_read_b64_tr_b16 %7, %15 offset:1024\n\t"
                "ds_read_b64_tr_b16 %8, %16 offset:8192\n\t"
                "ds_read_b64_tr_b16 %9, %16 offset:9216\n\t"
                "ds_read_b64_tr_b16 %10, %17 offset:8192\n\t"
                "ds_read_b64_tr_b16 %11, %17 offset:9216\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
                  "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11])
                : "v"(sbase + offa[0]), "v"(sbase + offa[1]), "v"(sbase + offa[2]), "v"(sbase + offa[3]),
                  "v"(sbase + offb[0]), "v"(sbase + offb[1])
                : "memory");
        } else
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\t"
            "ds_read_b64_tr_b16 %1, %16 offset:1024\n\t"
            "ds_read_b64_tr_b16 %2, %17\n\t"
            "ds_read_b64_tr_b16 %3, %17 offset:1024\n\t"
            "ds_read_b64_tr_b16 %4, %18\n\t"
            "ds_read_b64_tr_b16 %5, %18 offset:1024\n\t"
            "ds_read_b64_tr_b16 %6, %19\n\t"
            "ds_read_b64_tr_b16 %7, %19 offset:1024\n\t"
            "ds_read_b64_tr_b16 %8, %20 offset:8192\n\t"
            "ds_read_b64_tr_b16 %9, %20 offset:9216\n\t"
            "ds_read_b64_tr_b16 %10, %21 offset:8192\n\t"
            "ds_read_b64_tr_b16 %11, %21 offset:9216\n\t"
            "ds_read_b64_tr_b16 %12, %22 offset:8192\n\t"
            "ds_read_b64_tr_b16 %13, %22 offset:9216\n\t"
            "ds_read_b64_tr_b16 %14, %23 offset:8192\n\t"
            "ds_read_b64_tr_b16 %15, %23 offset:9216\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
              "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12]), "=&v"(t[13]), "=&v"(t[14]), "=&v"(t[15])
            : "v"(sbase + offa[0]), "v"(sbase + offa[1]), "v"(sbase + offa[2]), "v"(sbase + offa[3]),
              "v"(sbase + offb[0]), "v"(sbase + offb[1]), "v"(sbase + offb[2]), "v"(sbase + offb[3])
            : "memory");
        if constexpr (GATHER) {     // row indices of stage kt + 3 (their DMA retired with the wait above), then its data DMAs
            uint32_t q0, q1;
            const uint32_t ia = lds_base + (uint32_t)(NST * STAGE + ((kt + 3) & (IDX_SLOTS - 1)) * IDX_BYTES) + idx_off;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:32\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1) : "v"(ia) : "memory");
            if (kt + 3 < nk) issue(kt + 3, q0, q1);
        }
        bf16x8 fa[4], fb[JN];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            fa[i] = __builtin_bit_cast(bf16x8, (u32x4){t[2 * i][0], t[2 * i][1], t[2 * i + 1][0], t[2 * i + 1][1]});
#pragma unroll
        for (int i = 0; i < JN; ++i)
            fb[i] = __builtin_bit_cast(bf16x8, (u32x4){t[8 + 2 * i][0], t[8 + 2 * i][1], t[9 + 2 * i][0], t[9 + 2 * i][1]});
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (do_bias) {
            const bf16 one = (bf16)1.f;
            const bf16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
            for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
        }
    }

    if constexpr (GATHER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // trailing index DMAs must land before the LDS is released
    auto rowmap = [&](int n1) {       // head-major column of P -> row of dW in q | k | v | c order
        if (g.perm_dh <= 0) return n1;
        const int w = n1 % g.perm_dh, hm = n1 / g.perm_dh;
        return (hm & 3) * g.perm_d + (hm >> 2) * g.perm_dh + w;
    };
    if (do_bias && r == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n1 = n1_0 + wm * 64 + i * 16 + 4 * q + e;
                if (n1 < g.N1) g.bias_slab[(int64_t)split * g.N1 + rowmap(n1)] = accb[i][e];
            }
    }
    float* out = g.slab + (int64_t)split * g.N1 * g.N2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int n2 = n2_0 + wn * (16 * JN) + j * 16 + r;
            if (n2 >= g.N2) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n1 = n1_0 + wm * 64 + i * 16 + 4 * q + e;
                if (n1 < g.N1) out[(int64_t)rowmap(n1) * g.N2 + n2] = acc[i][j][e];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// TN, bf16, 256 x 256 output tile, 512 threads (8 waves as 2 x 4, 128 x 64 outputs each), same 4-stage LDS-DMA ring
// (32 rows of P and of Q per stage, 512-byte rows).  The 128 x 128 kernel re-reads P once per 128 columns of Q and Q
// once per 128 columns of P through L2 (dW_qkvc at M = 393k: 3.2 GB of L2 -> CU traffic for 1 GB of HBM bytes, i.e.
// 10.8 TB/s in 295 us: L2-bound); this tile halves both.  No row gather.
// ------------------------------------------------------------------------------------------------
// swizzle of the 32-row stages of gemm_tn_big_kernel: a transposed fragment read touches rows 8 q + (r >> 2) (+ 4), i.e.
// row bits {0, 1} and {3, 4} -- all four must enter the XOR, or rows 16 apart land on the same banks (2-way conflict on
// every ds_read_b64_tr_b16: the kernel was LDS-bound at 1536 of its 2150 cycles per step)
__device__ __forceinline__ int tn_f5(int row) { return (row & 3) | (((row >> 3) & 3) << 2); }

#ifdef PMGT_TN_PROF
// cycles per k-step phase of gemm_tn_big_kernel: [block slot][wave][0 vmcnt wait | 1 barrier | 2 DMA issue | 3 LDS reads + MFMA]
__device__ unsigned int g_tn_prof[2][8][4];
#endif
// GATHER: Q rows are picked by q_rows[m] (the token-mode feature-projection weight gradient: Q = the frozen table).  The row indices
// of a stage travel ahead of it by LDS-DMA into a small ring (each wave fetches the 32 indices itself: no cross-wave dependency), are
// read at the END of the step before the one that issues the stage's data DMA, and the data pieces carry per-lane source rows.
template <bool GATHER>
__global__ __launch_bounds__(512) void gemm_tn_big_kernel(GemmTN g, int chunk_rows) {
    constexpr int BKM = 32, ROWB = 512, STAGE = 2 * BKM * ROWB, NST = 4;     // 32 KiB per stage (P + Q)
    constexpr int IDX_SLOTS = 8, IDX_BYTES = BKM * 8, IDX_AHEAD = 7;         // index ring (GATHER): after the stages
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tn1 = (g.N1 + 255) / 256, tn2 = (g.N2 + 255) / 256, tiles = tn1 * tn2;
    const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
    const int tile = bidx % tiles, split = xcd + 8 * (bidx / tiles);
    if (split >= g.splits) return;
    const int n1_0 = (tile / tn2) * 256, n2_0 = (tile % tn2) * 256;
    const int Mlim = g.m_dev ? min(g.M, *g.m_dev) : g.M;
    chunk_rows = tn_live_chunk(g, Mlim, chunk_rows, 64);
    const int mbeg = split * chunk_rows, mend = min(Mlim, mbeg + chunk_rows);
    // The wave index must be KNOWN to be wave-uniform (readfirstlane): the bias MFMAs below sit under `(i >> 1) == wn`, and
    // a condition derived from threadIdx.x is divergent to the compiler, which then guards the block with an EXEC mask
    // instead of a branch -- MFMA ignores EXEC (every wave summed every fragment into accb) while the v_mov that builds
    // the ones operand obeys it (uninitialised operand): bias gradients of 1e27 at B = 1024, the clip coefficient 0.
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, q = lane >> 4;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // column sums of P (the bias gradient) ride along as MFMAs against a ones fragment: wave (wm, wn) of the n2 = 0 tile
    // sums the A fragments i = 2 wn, 2 wn + 1 of its row half (two extra MFMAs per step on every wave, not eight on one)
    const bool do_bias = g.bias_slab != nullptr && n2_0 == 0;
    f32x4 accb[2];
    accb[0] = accb[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane's part of a stage: rows 4 wave + 2 j + (lane >> 5), j = 0, 1; LDS chunk slot lane & 31 holds global
    // chunk (lane & 31) ^ swz(row).  The source pointers advance by one stage per step; only the LAST stage of a split
    // (rows beyond `mend`) and out-of-range column chunks read the zero page instead.
    const char* zero = (const char*)g.zeros;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void_t*)smem;
    const char* sp0[2];
    const char* sq0[2];
    bool okp[2], okq[2];
    int rowj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 4 * wave + 2 * j + (lane >> 5);
        const int ch = (lane & 31) ^ (tn_f5(row) << 1);
        const int cp = n1_0 + ch * 8, cq = n2_0 + ch * 8;
        rowj[j] = row;
        okp[j] = cp < g.N1;
        okq[j] = cq < g.N2;
        sp0[j] = (const char*)g.P + ((int64_t)(mbeg + row) * g.ldp + cp) * 2;
        sq0[j] = (const char*)g.Q + ((int64_t)(mbeg + row) * g.ldq + cq) * 2;
    }
    const int64_t stp = (int64_t)BKM * g.ldp * 2, stq = (int64_t)BKM * g.ldq * 2;
    uint32_t qcol[2];                 // GATHER: byte column of this lane's chunk inside a Q row
#pragma unroll
    for (int j = 0; j < 2; ++j) qcol[j] = (uint32_t)((n2_0 + (((lane & 31) ^ (tn_f5(rowj[j]) << 1)) * 8)) * 2);
    auto issue = [&](int kt, uint32_t q0, uint32_t q1) {
        const int mb = mbeg + kt * BKM;
        char* st = smem + (kt & (NST - 1)) * STAGE;
        const bool full = mb + BKM <= mend;          // (uniform) every row of the stage exists
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool in = full || mb + rowj[j] < mend;
            const char* sp = (in && okp[j]) ? sp0[j] + kt * stp : zero;
            const char* sq;
            if constexpr (GATHER) sq = (in && okq[j]) ? (const char*)g.Q + (int64_t)(j == 0 ? q0 : q1) * g.ldq * 2 + qcol[j] : zero;
            else sq = (in && okq[j]) ? sq0[j] + kt * stq : zero;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)sp, (lds_void_t*)(st + (4 * wave + 2 * j) * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)sq, (lds_void_t*)(st + BKM * ROWB + (4 * wave + 2 * j) * ROWB), 16, 0, 0);
        }
    };
    // index DMA of stage kt: lane l fetches dword l of q_rows[mb .. mb + 32) (clamped inside the array); its own rows' indices sit at
    // byte 8 (4 wave + 2 j + (lane >> 5)) of the slot (low dword: node ids are far below 2^31)
    char* idx_ring = smem + NST * STAGE;
    auto issue_idx = [&](int kt) {
        const int64_t m = min((int64_t)mbeg + (int64_t)kt * BKM + (lane >> 1), (int64_t)g.M - 1);
        const char* src = (const char*)(g.q_rows + m) + 4 * (lane & 1);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(idx_ring + (kt & (IDX_SLOTS - 1)) * IDX_BYTES), 4, 0, 0);
    };
    const uint32_t idx_off = (uint32_t)((4 * wave + (lane >> 5)) * 8);      // row of j = 0; j = 1 is 2 rows (16 B) further
    auto read_idx = [&](int kt, uint32_t& q0, uint32_t& q1) __attribute__((always_inline)) {
        const uint32_t ia = lds_base + (uint32_t)(NST * STAGE + (kt & (IDX_SLOTS - 1)) * IDX_BYTES) + idx_off;
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q0), "=&v"(q1) : "v"(ia) : "memory");
    };
    const int nk = mend > mbeg ? (mend - mbeg + BKM - 1) / BKM : 0;
    uint32_t qn0 = 0, qn1 = 0;        // GATHER: row indices of the stage whose data DMA the NEXT step issues
    if constexpr (GATHER) {
        if (nk > 0) {
#pragma unroll
            for (int kt = 0; kt < IDX_AHEAD; ++kt) issue_idx(kt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                uint32_t q0, q1;
                read_idx(kt, q0, q1);
                if (kt < nk) issue(kt, q0, q1);
            }
            read_idx(3, qn0, qn1);
        }
    } else {
        if (nk > 0) issue(0, 0, 0);
        if (nk > 1) issue(1, 0, 0);
        if (nk > 2) issue(2, 0, 0);
    }
    // per-lane byte offsets of the fragment reads inside a stage (row = 8 q + (r >> 2); +4 rows = +2048 B)
    uint32_t offa[8], offb[4];
    {
        const int row = 8 * q + (r >> 2);
        const int sw = tn_f5(row) << 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ca = wm * 128 + i * 16 + 4 * (r & 3);
            offa[i] = (uint32_t)(row * ROWB + (((ca >> 3) ^ sw) << 4) + ((ca & 7) << 1));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cb = wn * 64 + j * 16 + 4 * (r & 3);
            offb[j] = (uint32_t)(BKM * ROWB + row * ROWB + (((cb >> 3) ^ sw) << 4) + ((cb & 7) << 1));
        }
    }
#ifdef PMGT_TN_PROF
    unsigned int pacc[4] = {0, 0, 0, 0};
    unsigned long long plast = __builtin_readcyclecounter();
#define TN_STAMP(k_) do { const unsigned long long n_ = __builtin_readcyclecounter(); pacc[k_] += (unsigned int)(n_ - plast); plast = n_; } while (0)
#else
#define TN_STAMP(k_) do { } while (0)
#endif
    for (int kt = 0; kt < nk; ++kt) {
        TN_STAMP(3);
        const int younger = min(2, nk - 1 - kt);
        if constexpr (GATHER) {
            // 4 data pieces per younger stage + the index DMAs of the last min(kt, 2) steps (one per step, issued after the barrier
            // below for every kt, also past the end: clamped reads)
            switch (4 * younger + min(kt, 2)) {
                case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        TN_STAMP(0);
        __builtin_amdgcn_s_barrier();
        if constexpr (GATHER) issue_idx(kt + IDX_AHEAD);
        TN_STAMP(1);
        TN_STAMP(2);
        const uint32_t sbase = lds_base + (uint32_t)((kt & (NST - 1)) * STAGE);
        // all 24 transposed fragment reads of the step go out back to back (B first), and the MFMAs of A-fragment i start
        // as soon as ITS two reads have landed (LDS returns in order: lgkmcnt counts down), so the tail of the LDS
        // traffic overlaps the matrix pipe instead of preceding it
        u32x2 ta[16], tb[8];
        asm volatile(
            "ds_read_b64_tr_b16 %0, %8\n\t"
            "ds_read_b64_tr_b16 %1, %8 offset:2048\n\t"
            "ds_read_b64_tr_b16 %2, %9\n\t"
            "ds_read_b64_tr_b16 %3, %9 offset:2048\n\t"
            "ds_read_b64_tr_b16 %4, %10\n\t"
            "ds_read_b64_tr_b16 %5, %10 offset:2048\n\t"
            "ds_read_b64_tr_b16 %6, %11\n\t"
            "ds_read_b64_tr_b16 %7, %11 offset:2048"
            : "=&v"(tb[0]), "=&v"(tb[1]), "=&v"(tb[2]), "=&v"(tb[3]), "=&v"(tb[4]), "=&v"(tb[5]), "=&v"(tb[6]), "=&v"(tb[7])
            : "v"(sbase + offb[0]), "v"(sbase + offb[1]), "v"(sbase + offb[2]), "v"(sbase + offb[3])
            : "memory");
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\t"
            "ds_read_b64_tr_b16 %1, %16 offset:2048\n\t"
            "ds_read_b64_tr_b16 %2, %17\n\t"
            "ds_read_b64_tr_b16 %3, %17 offset:2048\n\t"
            "ds_read_b64_tr_b16 %4, %18\n\t"
            "ds_read_b64_tr_b16 %5, %18 offset:2048\n\t"
            "ds_read_b64_tr_b16 %6, %19\n\t"
            "ds_read_b64_tr_b16 %7, %19 offset:2048\n\t"
            "ds_read_b64_tr_b16 %8, %20\n\t"
            "ds_read_b64_tr_b16 %9, %20 offset:2048\n\t"
            "ds_read_b64_tr_b16 %10, %21\n\t"
            "ds_read_b64_tr_b16 %11, %21 offset:2048\n\t"
            "ds_read_b64_tr_b16 %12, %22\n\t"
            "ds_read_b64_tr_b16 %13, %22 offset:2048\n\t"
            "ds_read_b64_tr_b16 %14, %23\n\t"
            "ds_read_b64_tr_b16 %15, %23 offset:2048"
            : "=&v"(ta[0]), "=&v"(ta[1]), "=&v"(ta[2]), "=&v"(ta[3]), "=&v"(ta[4]), "=&v"(ta[5]), "=&v"(ta[6]), "=&v"(ta[7]),
              "=&v"(ta[8]), "=&v"(ta[9]), "=&v"(ta[10]), "=&v"(ta[11]), "=&v"(ta[12]), "=&v"(ta[13]), "=&v"(ta[14]), "=&v"(ta[15])
            : "v"(sbase + offa[0]), "v"(sbase + offa[1]), "v"(sbase + offa[2]), "v"(sbase + offa[3]),
              "v"(sbase + offa[4]), "v"(sbase + offa[5]), "v"(sbase + offa[6]), "v"(sbase + offa[7])
            : "memory");
        // B fragments + A fragment 0 have landed when at most 14 reads are outstanding
        asm volatile("s_waitcnt lgkmcnt(14)"
                     : "+v"(tb[0]), "+v"(tb[1]), "+v"(tb[2]), "+v"(tb[3]), "+v"(tb[4]), "+v"(tb[5]), "+v"(tb[6]), "+v"(tb[7]),
                       "+v"(ta[0]), "+v"(ta[1]));
        bf16x8 fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            fb[j] = __builtin_bit_cast(bf16x8, (u32x4){tb[2 * j][0], tb[2 * j][1], tb[2 * j + 1][0], tb[2 * j + 1][1]});
        const bf16 one = (bf16)1.f;
        const bf16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i == 1) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(ta[2]), "+v"(ta[3]));
            if (i == 2) asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(ta[4]), "+v"(ta[5]));
            if (i == 3) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ta[6]), "+v"(ta[7]));
            if (i == 4) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ta[8]), "+v"(ta[9]));
            if (i == 5) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ta[10]), "+v"(ta[11]));
            if (i == 6) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ta[12]), "+v"(ta[13]));
            if (i == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[14]), "+v"(ta[15]));
            const bf16x8 fa = __builtin_bit_cast(bf16x8, (u32x4){ta[2 * i][0], ta[2 * i][1], ta[2 * i + 1][0], ta[2 * i + 1][1]});
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[j], acc[i][j], 0, 0, 0);
            if (do_bias && (i >> 1) == wn) accb[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, ones, accb[i & 1], 0, 0, 0);
            // the next stage's DMA goes out while the matrix pipe works through the MFMAs queued above: its address
            // arithmetic costs VALU issue slots only (ring slot (kt + 3) % 4 was last read in step kt - 1, and every wave
            // has passed this step's barrier)
            if (i == 3 && kt + 3 < nk) issue(kt + 3, qn0, qn1);
        }
        // GATHER: the indices of stage kt + 4 (their DMA went out at step kt - 3: the wait at the top of this step covered it)
        if constexpr (GATHER) read_idx(kt + 4, qn0, qn1);
    }
    if constexpr (GATHER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // trailing index DMAs must land before the LDS is released
#ifdef PMGT_TN_PROF
    TN_STAMP(3);
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133))
        for (int k_ = 0; k_ < 4; ++k_) g_tn_prof[blockIdx.x == 0 ? 0 : 1][wave][k_] = pacc[k_];
#endif
    auto rowmap = [&](int n1) {
        if (g.perm_dh <= 0) return n1;
        const int w = n1 % g.perm_dh, hm = n1 / g.perm_dh;
        return (hm & 3) * g.perm_d + (hm >> 2) * g.perm_dh + w;
    };
    if (do_bias && r == 0) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n1 = n1_0 + wm * 128 + (2 * wn + ib) * 16 + 4 * q + e;
                if (n1 < g.N1) g.bias_slab[(int64_t)split * g.N1 + rowmap(n1)] = accb[ib][e];
            }
    }
    float* out = g.slab + (int64_t)split * g.N1 * g.N2;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n2 = n2_0 + wn * 64 + j * 16 + r;
            if (n2 >= g.N2) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n1 = n1_0 + wm * 128 + i * 16 + 4 * q + e;
                if (n1 < g.N1) out[(int64_t)rowmap(n1) * g.N2 + n2] = acc[i][j][e];
            }
        }
}

// the 256 x 256 tile pays off when the 128 x 128 kernel would re-read its operands through L2 four times or more
static bool tn_big_shape(int M, int N1, int N2, int bkm, uint32_t opts) {
    return bkm == 64 && !(opts & OPT_TILE_GEMM) && M >= 65536 && N1 % 256 == 0 && N2 % 256 == 0 && N1 * N2 >= 4 * 256 * 256;
}

int gemm_tn_pick_splits(int M, int N1, int N2, int bkm, uint32_t opts) {
    if (tn_big_shape(M, N1, N2, bkm, opts)) {
        const int tiles = (N1 / 256) * (N2 / 256);
        int splits = cdiv(256, tiles);                   // one 8-wave workgroup per CU
        splits = std::max(8, splits / 8 * 8);
        return splits;
    }
    const int tiles = cdiv(N1, 128) * cdiv(N2, 128);
    // one workgroup per CU: the LDS-DMA ring hides the latency by itself, and the slab traffic (splits * N1 * N2 * 4 B
    // written here, read again by slab_reduce) halves against two per CU -- measured 11.49 vs 11.57 ms/step (c2, B = 1024)
    int splits = cdiv(256, tiles);
    const int max_by_rows = std::max(1, M / (4 * bkm));  // at least 4 K-steps per split
    splits = std::max(1, std::min(splits, max_by_rows));
    if (splits > 8) splits = splits / 8 * 8;             // whole XCD groups (see the block mapping in the kernel)
    return splits;
}

template <typename T> int gemm_tn(const GemmTN& g, hipStream_t st) {
    constexpr int EPC = 16 / sizeof(T);
    if (g.N1 <= 0 || g.N2 <= 0) return 0;
    PMGT_CHECK(g.N1 % EPC == 0 && g.N2 % EPC == 0, -2, "gemm_tn: N1=%d N2=%d must be multiples of %d", g.N1, g.N2, EPC);
    PMGT_CHECK(g.ldp % EPC == 0 && g.ldq % EPC == 0, -2, "gemm_tn: leading dims must be multiples of %d", EPC);
    PMGT_CHECK(((uintptr_t)g.P % 16) == 0 && ((uintptr_t)g.Q % (g.q_f8 ? 8 : 16)) == 0, -2, "gemm_tn: operands must be 16-byte aligned");
    PMGT_CHECK(g.splits >= 1 && g.slab, -2, "gemm_tn: bad splits/slab");
    PMGT_CHECK(g.perm_dh == 0 || (sizeof(T) == 2 && !(g.opts & OPT_TILE_GEMM) && g.zeros != nullptr), -2, "gemm_tn: the row permutation needs the LDS-DMA kernel");
    const int bkm = gemm_tn_bkm<T>();
    int chunk = cdiv(cdiv(std::max(g.M, 1), g.splits), bkm) * bkm;
    const int tiles = cdiv(g.N1, 128) * cdiv(g.N2, 128);
    dim3 grid(8 * tiles * cdiv(g.splits, 8));
    if (g.q_f8) {
        if constexpr (sizeof(T) == 2) {
            PMGT_CHECK(g.perm_dh == 0 && ((uintptr_t)g.Q % 8) == 0, -2, "gemm_tn: e4m3 Q operand needs 8-byte rows and no row permutation");
            hipLaunchKernelGGL((gemm_tn_kernel<T, true>), grid, dim3(256), 0, st, g, chunk);
            PMGT_LAUNCH_OK();
            return 0;
        } else {
            PMGT_CHECK(false, -2, "gemm_tn: e4m3 Q operand belongs to the fp8 mode (bf16 activations)");
        }
    }
    if constexpr (sizeof(T) == 2) {
        if (g.zeros != nullptr && tn_big_shape(g.M, g.N1, g.N2, bkm, g.opts) && g.splits == gemm_tn_pick_splits(g.M, g.N1, g.N2, bkm, g.opts) &&
            (g.q_rows == nullptr || g.M >= 1)) {
            constexpr int smem = 4 * 2 * 32 * 512 + 8 * 32 * 8;      // stage ring + index ring
            PMGT_SMEM_ATTR((const void*)gemm_tn_big_kernel<false>, smem); PMGT_SMEM_ATTR((const void*)gemm_tn_big_kernel<true>, smem);
            const int tiles256 = (g.N1 / 256) * (g.N2 / 256);
            int chunk256 = cdiv(cdiv(std::max(g.M, 1), g.splits), 32) * 32;
            note_launch(g.q_rows ? LT_TN_BIG_GATHER : LT_TN_BIG);
            if (g.q_rows) hipLaunchKernelGGL(gemm_tn_big_kernel<true>, dim3(8 * tiles256 * cdiv(g.splits, 8)), dim3(512), smem, st, g, chunk256);
            else hipLaunchKernelGGL(gemm_tn_big_kernel<false>, dim3(8 * tiles256 * cdiv(g.splits, 8)), dim3(512), smem, st, g, chunk256);
            PMGT_LAUNCH_OK();
            return 0;
        }
        if (!(g.opts & OPT_TILE_GEMM) && g.zeros != nullptr) {
            constexpr size_t ring = 4 * 2 * 32 * 256;
            if (g.q_rows == nullptr) {
                note_launch(LT_TN_DMA);
                hipLaunchKernelGGL((gemm_tn_dma_kernel<false, 8>), grid, dim3(512), ring, st, g, chunk);
            } else {
                PMGT_CHECK(g.M >= 1, -2, "gemm_tn: gather needs at least one row");
                constexpr size_t smem = ring + 8 * 32 * 8;
                PMGT_SMEM_ATTR((const void*)gemm_tn_dma_kernel<true>, (int)smem);
                note_launch(LT_TN_DMA_GATHER);
                hipLaunchKernelGGL(gemm_tn_dma_kernel<true>, grid, dim3(256), smem, st, g, chunk);
            }
            PMGT_LAUNCH_OK();
            return 0;
        }
    }
    note_launch(LT_TN_TILE);
    hipLaunchKernelGGL((gemm_tn_kernel<T>), grid, dim3(256), 0, st, g, chunk);
    PMGT_LAUNCH_OK();
    return 0;
}
template int gemm_tn<float>(const GemmTN&, hipStream_t);
template int gemm_tn<bf16>(const GemmTN&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// slab reduce / column sums
// ------------------------------------------------------------------------------------------------
// Deterministic tree: rows of `src` (stride `rs` floats) are summed in a fixed order.  A block is
// 64 float4-columns x 4 row-lanes; blockIdx.y selects a group of `group` consecutive rows.  Level 1
// (final == 0) writes each group's sum over the group's first row (only that block touches those
// columns of those rows); level 2 sums the group heads into dst.
// Sum of rows r, r + 4, r + 8, ... < r1 of one float4 column: two interleaved accumulators (s0: r, r + 8, ...; s1: r + 4, r + 12, ...)
// added at the end.  The loads of FOUR such pairs are issued before the first add (eight 16-byte loads in flight per thread
// instead of two: the loop is a chain of dependent HBM/L2 latencies otherwise, ~25 us for 32 rows per thread); the order of the
// additions is unchanged, so the result is bit-identical to the two-load loop.
__device__ __forceinline__ f32x4 rows_sum_strided(const float* __restrict__ base, int64_t stride, int r, int r1) {
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    for (; r + 28 < r1; r += 32) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(base + (int64_t)(r + 4 * u) * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s0 += v[2 * u]; s1 += v[2 * u + 1]; }
    }
    for (; r + 4 < r1; r += 8) {
        const f32x4 v0 = *(const f32x4*)(base + (int64_t)r * stride), v1 = *(const f32x4*)(base + (int64_t)(r + 4) * stride);
        s0 += v0;
        s1 += v1;
    }
    if (r < r1) s0 += *(const f32x4*)(base + (int64_t)r * stride);
    return s0 + s1;
}

__global__ __launch_bounds__(256) void rows_reduce_kernel(float* __restrict__ src, int64_t rs, int nrows, int group,
                                                          int64_t row_step, int64_t n, float* __restrict__ dst,
                                                          int accumulate, int final) {
    __shared__ f32x4 red[256];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t i4 = ((int64_t)blockIdx.x * 64 + c) * 4;
    const int r0 = blockIdx.y * group, r1 = min(nrows, r0 + group);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i4 < n) s = rows_sum_strided(src + i4, row_step * rs, r0 + rl, r1);
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && i4 < n) {
        f32x4 t = (red[c] + red[64 + c]) + (red[128 + c] + red[192 + c]);
        if (final) {
            if (accumulate) t += *(const f32x4*)(dst + i4);
            *(f32x4*)(dst + i4) = t;
        } else {
            *(f32x4*)(src + (int64_t)r0 * row_step * rs + i4) = t;
        }
    }
}

// dst[i] (+)= sum_s slab[s*n + i]; n must be a multiple of 4 (callers pad); the slab is clobbered.
int slab_reduce(const float* slab_c, int splits, int64_t n, float* dst, bool accumulate, hipStream_t st) {
    if (n <= 0) return 0;
    float* slab = const_cast<float*>(slab_c);
    PMGT_CHECK(n % 4 == 0, -2, "slab_reduce: n=%lld must be a multiple of 4", (long long)n);
    PMGT_CHECK(((uintptr_t)slab % 16) == 0 && ((uintptr_t)dst % 16) == 0, -2, "slab_reduce: unaligned buffers");
    const unsigned cb = (unsigned)cdiv64(n / 4, 64);
    const int G = 128;          // up to 128 slabs in one launch (4 row lanes x 32 loads each); beyond that two levels
    if (splits <= G) {
        hipLaunchKernelGGL(rows_reduce_kernel, dim3(cb, 1), dim3(256), 0, st, slab, n, splits, splits, (int64_t)1, n, dst,
                           accumulate ? 1 : 0, 1);
    } else {
        const int groups = cdiv(splits, G);
        hipLaunchKernelGGL(rows_reduce_kernel, dim3(cb, groups), dim3(256), 0, st, slab, n, splits, G, (int64_t)1, n, dst, 0, 0);
        PMGT_LAUNCH_OK();
        hipLaunchKernelGGL(rows_reduce_kernel, dim3(cb, 1), dim3(256), 0, st, slab, n, groups, groups, (int64_t)G, n, dst,
                           accumulate ? 1 : 0, 1);
    }
    PMGT_LAUNCH_OK();
    return 0;
}

// Several slab reductions in one launch (two when a job has more than 128 slabs): the backward pass produces ~10 sets of partial
// sums per layer (weight-gradient slabs, bias slabs, LayerNorm partials), and a launch per set costs more in launch latency and
// serialisation than in bytes -- 40 launches per step took 0.43 ms at every batch size.  Same arithmetic per element as
// slab_reduce (fixed order: bit-reproducible), jobs packed into the grid through cumulative block counts.
struct MultiReduceArgs {
    struct J { float* src; float* dst; int64_t n, row_step; int nrows, group, final, acc, cb; } j[8];
    int blk0[9];
    int njobs;
};
__global__ __launch_bounds__(256) void multi_reduce_kernel(MultiReduceArgs a) {
    __shared__ f32x4 red[256];
    int k = 0;
    while (k + 1 < a.njobs && (int)blockIdx.x >= a.blk0[k + 1]) ++k;       // (uniform)
    const MultiReduceArgs::J& jb = a.j[k];
    const int lb = blockIdx.x - a.blk0[k];
    const int cbk = lb % jb.cb, grp = lb / jb.cb;
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t i4 = ((int64_t)cbk * 64 + c) * 4;
    const int r0 = grp * jb.group, r1 = min(jb.nrows, r0 + jb.group);
    const int64_t rs = jb.n;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i4 < jb.n) s = rows_sum_strided(jb.src + i4, jb.row_step * rs, r0 + rl, r1);
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && i4 < jb.n) {
        f32x4 t = (red[c] + red[64 + c]) + (red[128 + c] + red[192 + c]);
        if (jb.final) {
            if (jb.acc) t += *(const f32x4*)(jb.dst + i4);
            *(f32x4*)(jb.dst + i4) = t;
        } else {
            *(f32x4*)(jb.src + (int64_t)r0 * jb.row_step * rs + i4) = t;
        }
    }
}

int multi_reduce(const ReduceJob* jobs, int njobs, hipStream_t st) {
    constexpr int G = 128;
    for (int base = 0; base < njobs; base += 8) {
        const int nj = std::min(8, njobs - base);
        for (int level = 0; level < 2; ++level) {
            MultiReduceArgs a;
            int cnt = 0, blocks = 0;
            for (int q = 0; q < nj; ++q) {
                const ReduceJob& rj = jobs[base + q];
                if (rj.n <= 0 || rj.rows <= 0) continue;
                PMGT_CHECK(rj.n % 4 == 0 && ((uintptr_t)rj.src % 16) == 0 && ((uintptr_t)rj.dst % 16) == 0, -2, "multi_reduce: unaligned job");
                const bool two = rj.rows > G;
                if (level == 0 && !two) continue;
                MultiReduceArgs::J& d = a.j[cnt];
                d.src = const_cast<float*>(rj.src); d.dst = rj.dst; d.n = rj.n;
                d.cb = (int)cdiv64(rj.n / 4, 64);
                if (level == 0) { d.nrows = rj.rows; d.group = G; d.row_step = 1; d.final = 0; d.acc = 0; }
                else {
                    d.nrows = two ? cdiv(rj.rows, G) : rj.rows; d.group = d.nrows; d.row_step = two ? G : 1; d.final = 1; d.acc = rj.acc ? 1 : 0;
                }
                a.blk0[cnt] = blocks;
                blocks += d.cb * (level == 0 ? cdiv(rj.rows, G) : 1);
                ++cnt;
            }
            if (cnt == 0) continue;
            a.blk0[cnt] = blocks;
            a.njobs = cnt;
            hipLaunchKernelGGL(multi_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
            PMGT_LAUNCH_OK();
        }
    }
    return 0;
}

// Column sums: a block owns 256 rows x (CC chunk-columns of 16 bytes); thread = (chunk, row lane).
template <typename T, int CC>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ Y, int64_t ldy, int M, int N,
                                                     float* __restrict__ slab, const int* m_dev) {
    constexpr int EPC = 16 / sizeof(T), RL = 256 / CC;
    __shared__ float red[256 * EPC];
    const int Mlim = m_dev ? min(M, *m_dev) : M;
    const int cc = threadIdx.x % CC, rl = threadIdx.x / CC;
    const int col = (blockIdx.y * CC + cc) * EPC;
    const int mb = blockIdx.x * COLSUM_ROWS, me = min(Mlim, mb + COLSUM_ROWS);
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    if (col < N) {
        int m = mb + rl;
        if constexpr (sizeof(T) == 2) {      // four rows per trip: four independent 16-byte loads in flight per lane
            for (; m + 3 * RL < me; m += 4 * RL) {
                const bf16x8 v0 = *(const bf16x8*)(Y + (int64_t)m * ldy + col), v1 = *(const bf16x8*)(Y + (int64_t)(m + RL) * ldy + col);
                const bf16x8 v2 = *(const bf16x8*)(Y + (int64_t)(m + 2 * RL) * ldy + col), v3 = *(const bf16x8*)(Y + (int64_t)(m + 3 * RL) * ldy + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += ((float)v0[e] + (float)v1[e]) + ((float)v2[e] + (float)v3[e]);
            }
        }
        for (; m < me; m += RL) {
            if constexpr (sizeof(T) == 2) {
                bf16x8 v = *(const bf16x8*)(Y + (int64_t)m * ldy + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            } else {
                f32x4 v = *(const f32x4*)(Y + (int64_t)m * ldy + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[threadIdx.x * EPC + e] = acc[e];
    __syncthreads();
    if (rl == 0 && col < N) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float t = 0.f;
            for (int k = 0; k < RL; ++k) t += red[(k * CC + cc) * EPC + e];
            slab[(int64_t)blockIdx.x * N + col + e] = t;
        }
    }
}

template <typename T>
int colsum(const T* Y, int64_t ldy, int M, int N, float* slab, float* dst, bool accumulate, const int* m_dev,
           hipStream_t st) {
    if (N <= 0) return 0;
    constexpr int EPC = 16 / sizeof(T);
    PMGT_CHECK(N % EPC == 0 && ldy % EPC == 0 && ((uintptr_t)Y % 16) == 0, -2,
               "colsum: N=%d / ld must be multiples of %d and Y 16-byte aligned", N, EPC);
    const int rb = std::max(1, cdiv(M, COLSUM_ROWS));
    const int chunks = N / EPC;
    if (chunks <= 32) hipLaunchKernelGGL((colsum_kernel<T, 32>), dim3(rb, cdiv(chunks, 32)), dim3(256), 0, st, Y, ldy, M, N, slab, m_dev);
    else if (chunks <= 64) hipLaunchKernelGGL((colsum_kernel<T, 64>), dim3(rb, cdiv(chunks, 64)), dim3(256), 0, st, Y, ldy, M, N, slab, m_dev);
    else hipLaunchKernelGGL((colsum_kernel<T, 128>), dim3(rb, cdiv(chunks, 128)), dim3(256), 0, st, Y, ldy, M, N, slab, m_dev);
    PMGT_LAUNCH_OK();
    return slab_reduce(slab, rb, N, dst, accumulate, st);
}
template int colsum<float>(const float*, int64_t, int, int, float*, float*, bool, const int*, hipStream_t);
template int colsum<bf16>(const bf16*, int64_t, int, int, float*, float*, bool, const int*, hipStream_t);

}  // namespace pmgt

#ifdef PMGT_TN_PROF
extern "C" int pmgt_debug_nt_prof_read(unsigned int* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmgt::g_nt_prof), sizeof(pmgt::g_nt_prof));
}
extern "C" int pmgt_debug_tn_prof_read(unsigned int* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmgt::g_tn_prof), sizeof(pmgt::g_tn_prof));
}
#endif
