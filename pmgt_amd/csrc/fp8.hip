// fp8 (OCP e4m3) mode: quantisation kernels and the fp8 MFMA "NT" GEMM (see fp8.h).
//
// gemm_nt_f8_kernel: 128 x 128 output tile, 4 waves (2 x 2, 64 x 64 each as 4 x 4 tiles of the block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales: the fp8 MFMA of gfx950 that runs at the fp8 rate),
// K-step 128 (one 128-byte LDS row per operand row, as in the bf16 tile kernel: half the bytes per k), operands staged
// global -> registers -> LDS in 16-byte chunks, double-buffered, next tile's loads issued before the MFMAs.  A fragment
// is 32 bytes (k = 32 q .. 32 q + 31, two 16-byte reads): the 16-byte chunk index is XOR-swizzled by row & 7, so the 16
// rows of a quarter-wave read land on 8 distinct chunks x 2 row parities = all 64 banks.  The row gather (feature rows by node
// id) is the A-operand row index, as in the bf16 path.  Epilogue: acc * sa[m] * sb[n] + bias -> bf16 through an fp32 LDS
// stage so that every global store is a full 16-byte vector.
#include <stdlib.h>

#include "fp8.h"

namespace pmgt {

__device__ __forceinline__ int f8_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

__global__ __launch_bounds__(256) void gemm_nt_f8_kernel(GemmF8 g) {
    constexpr int BM = 128, BN = 128, BK = 128;
    constexpr int TM = 4, TN = 4, LA = 4, LB = 4;
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * 128];
    char* sA = smem;
    char* sB = smem + 2 * BM * 128;

    const int num_n = (g.N + BN - 1) / BN;
    const int num_m = (g.M + BM - 1) / BM;
    const int b = blockIdx.x;
    const int grp = b / (8 * num_n), within = b % (8 * num_n);
    const int m_tile = grp * 8 + (within & 7), n_tile = within >> 3;       // N-tiles of one M-tile share an XCD
    if (m_tile >= num_m) return;
    const int Mlim = g.m_dev ? min(g.M, *g.m_dev) : g.M;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    if (m0 >= Mlim) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const int c = tid & 7, r0 = tid >> 3;
    const char* arow[LA];
    const char* brow[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int m = min(m0 + r0 + 32 * i, Mlim - 1);
        const int64_t row = g.a_rows ? g.a_rows[m] : (int64_t)m;
        arow[i] = (const char*)g.A + row * g.lda;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int n = min(n0 + r0 + 32 * i, g.N - 1);
        brow[i] = (const char*)g.B + (int64_t)n * g.ldb;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x4 ra[LA], rb[LB];
    auto gload = [&](int k0) {
        const int k = k0 + c * 16;
        const bool ok = k < g.K;                       // K % 16 == 0 (host check)
#pragma unroll
        for (int i = 0; i < LA; ++i) ra[i] = ok ? *(const u32x4*)(arow[i] + k) : (u32x4){0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < LB; ++i) rb[i] = ok ? *(const u32x4*)(brow[i] + k) : (u32x4){0, 0, 0, 0};
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) *(u32x4*)(sA + buf * BM * 128 + f8_off(r0 + 32 * i, c)) = ra[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *(u32x4*)(sB + buf * BN * 128 + f8_off(r0 + 32 * i, c)) = rb[i];
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
        {   // one block-scaled MFMA per 16 x 16 tile and 128-k step (unit scales): lane (r, q) holds k = 32 q .. 32 q + 31 of its row
            typedef int i32x8_t __attribute__((ext_vector_type(8)));
            i32x8_t fa[TM], fb[TN];
            auto frag = [&](const char* base, int row) {
                const u32x4 lo = *(const u32x4*)(base + f8_off(row, 2 * q)), hi = *(const u32x4*)(base + f8_off(row, 2 * q + 1));
                return (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            };
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = frag(a_base, wm * 64 + i * 16 + r);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = frag(b_base, wn * 64 + j * 16 + r);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[i], fb[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  C/D layout: col = lane & 15, row = 4 (lane >> 4) + reg
    constexpr int ES = BN + 4;
    float* stage = (float*)smem;                        // 64 x ES floats
    bf16* C = (bf16*)g.C;
    const int er = tid >> 4, ec = (tid & 15) * 8;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) stage[(i * 16 + 4 * q + e) * ES + wn * 64 + j * 16 + r] = acc[i][j][e];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = er + 16 * it;
            const int m = m0 + pass * 64 + row;
            if (m < Mlim) {
                const float sa = g.a_row_scale ? g.a_row_scale[m] : g.a_scale;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int n = n0 + ec + 4 * h;
                    if (n < g.N) {                          // N % 4 == 0 (host check)
                        f32x4 v = *(const f32x4*)(stage + row * ES + ec + 4 * h);
                        f32x4 sb = {1.f, 1.f, 1.f, 1.f};
                        if (g.b_row_scale) sb = *(const f32x4*)(g.b_row_scale + n);
                        v = v * (sb * sa);
                        if (g.bias) v += *(const f32x4*)(g.bias + n);
                        store4<bf16>(C + (int64_t)m * g.ldc + n, v);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tile for the large fp8 GEMMs (the Q|K|V|C projection of the d = 512 shapes: M = 196 608, N = 2 048, K = 512): the
// structure of gemm_nt_big_kernel (gemm.hip) with e4m3 operands -- 8 waves (2 x 4, 128 x 64 outputs each), 4-slot LDS-DMA ring
// of 64-byte rows = 64 k per stage (half the DMA and LDS bytes per k of the bf16 tile), counted vmcnt, one barrier per stage,
// the four DMA instructions of stage kt + 3 spread between the MFMAs of stage kt.  v_mfma_f32_16x16x32_fp8_fp8 with swapped
// operands (a lane owns four consecutive output columns), 8-byte fragments (k = 32 h + 8 q .. + 7: source chunk 2 h + (q >> 1),
// XOR-swizzled by (row >> 2) & 3 on the DMA source side: the 32 lanes of a half-wave hit 64 distinct banks); direct epilogue
// acc * sa[m] * sb[n] + bias -> bf16 from the registers (v_permlane16_swap pairs two 16-column blocks: 16-byte stores).
// (The block-scaled 16x16x128 MFMA would need 128-byte rows = a 2-slot ring; this tile is bound by its DMA stream first.)
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_f8_t;
typedef __attribute__((address_space(1))) const void gbl_void_f8_t;

__global__ __launch_bounds__(512) void gemm_nt_f8_big_kernel(GemmF8 g) {
    constexpr int BM = 256, BN = 256, ROWB = 64, STAGE = (BM + BN) * ROWB, NST = 4, AI = 2, BI = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int num_n = g.N / BN;
    const int num_m = (g.M + BM - 1) / BM;
    const int b = blockIdx.x;
    const int grp = b / (8 * num_n), within = b % (8 * num_n);
    const int m_tile = grp * 8 + (within & 7), n_tile = within >> 3;
    if (m_tile >= num_m) return;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, q = lane >> 4;

    const char* asrc[AI];
    const char* bsrc[BI];
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int row = 16 * (AI * wave + j) + (lane >> 2);
        const int ch = (lane & 3) ^ ((row >> 2) & 3);
        const int m = min(m0 + row, g.M - 1);
        asrc[j] = (const char*)g.A + (int64_t)m * g.lda + ch * 16;
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = 16 * (BI * wave + j) + (lane >> 2);
        const int ch = (lane & 3) ^ ((row >> 2) & 3);
        bsrc[j] = (const char*)g.B + (int64_t)(n0 + row) * g.ldb + ch * 16;
    }
    auto issue_piece = [&](int kt, int piece) __attribute__((always_inline)) {
        char* st = smem + (kt % NST) * STAGE;
        if (piece < AI)
            __builtin_amdgcn_global_load_lds((gbl_void_f8_t*)(asrc[piece] + (int64_t)kt * ROWB),
                                             (lds_void_f8_t*)(st + 16 * (AI * wave + piece) * ROWB), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((gbl_void_f8_t*)(bsrc[piece - AI] + (int64_t)kt * ROWB),
                                             (lds_void_f8_t*)(st + BM * ROWB + 16 * (BI * wave + piece - AI) * ROWB), 16, 0, 0);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void_f8_t*)smem;
    const int sw = (r >> 2) & 3;
    uint32_t aA[2], aB[2];      // per k-half h: byte address of this lane's 8-byte fragment in row (wm 128 + r) / (wn 64 + r) of stage 0
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int pos = ((2 * h + (q >> 1)) ^ sw) << 4;
        aA[h] = lds_base + (uint32_t)((wm * 128 + r) * ROWB + pos + 8 * (q & 1));
        aB[h] = lds_base + (uint32_t)(BM * ROWB + (wn * 64 + r) * ROWB + pos + 8 * (q & 1));
    }
    const int nk = g.K / 64;
#pragma unroll
    for (int kt = 0; kt < NST - 1; ++kt)
        if (kt < nk) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) issue_piece(kt, pc);
        }
    for (int kt = 0; kt < nk; ++kt) {
        const int younger = min(NST - 2, nk - 1 - kt);
        if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint32_t so = (uint32_t)((kt % NST) * STAGE);
        u32x2 fa[2][8], fb[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            asm volatile(
                "ds_read_b64 %0, %12\n\t"
                "ds_read_b64 %1, %12 offset:1024\n\t"
                "ds_read_b64 %2, %12 offset:2048\n\t"
                "ds_read_b64 %3, %12 offset:3072\n\t"
                "ds_read_b64 %4, %13\n\t"
                "ds_read_b64 %5, %13 offset:1024\n\t"
                "ds_read_b64 %6, %13 offset:2048\n\t"
                "ds_read_b64 %7, %13 offset:3072\n\t"
                "ds_read_b64 %8, %13 offset:4096\n\t"
                "ds_read_b64 %9, %13 offset:5120\n\t"
                "ds_read_b64 %10, %13 offset:6144\n\t"
                "ds_read_b64 %11, %13 offset:7168"
                : "=&v"(fb[h][0]), "=&v"(fb[h][1]), "=&v"(fb[h][2]), "=&v"(fb[h][3]), "=&v"(fa[h][0]), "=&v"(fa[h][1]), "=&v"(fa[h][2]),
                  "=&v"(fa[h][3]), "=&v"(fa[h][4]), "=&v"(fa[h][5]), "=&v"(fa[h][6]), "=&v"(fa[h][7])
                : "v"(aB[h] + so), "v"(aA[h] + so)
                : "memory");
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // the twelve reads of half 0 have landed when at most the twelve of half 1 are outstanding
            if (h == 0)
                asm volatile("s_waitcnt lgkmcnt(12)"
                             : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]), "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]),
                               "+v"(fa[0][3]), "+v"(fa[0][4]), "+v"(fa[0][5]), "+v"(fa[0][6]), "+v"(fa[0][7]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]), "+v"(fb[1][3]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]),
                               "+v"(fa[1][3]), "+v"(fa[1][4]), "+v"(fa[1][5]), "+v"(fa[1][6]), "+v"(fa[1][7]));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(__builtin_bit_cast(long, fb[h][j]), __builtin_bit_cast(long, fa[h][i]),
                                                                           acc[i][j], 0, 0, 0);
                // ring slot (kt + 3) % 4 was last read in stage kt - 1, and every wave is past this stage's barrier
                if ((i & 3) == 3 && kt + NST - 1 < nk) issue_piece(kt + NST - 1, 2 * h + (i >> 2));
            }
        }
    }

    // ---- direct epilogue: acc[i][j][e] = out[wm 128 + 16 i + r][wn 64 + 16 j + 4 q + e]
    bf16* Cp = (bf16*)g.C;
    const int cb = ((q & 1) << 4) | ((q & 2) << 2);          // q = 0, 1, 2, 3 -> columns 0, 16, 8, 24 of the 32-column pair
    const int ncol = n0 + wn * 64 + cb;
    f32x4 bv[2][2], sv[2][2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            bv[pr][k] = g.bias ? *(const f32x4*)(g.bias + ncol + 32 * pr + 4 * k) : (f32x4){0.f, 0.f, 0.f, 0.f};
            sv[pr][k] = g.b_row_scale ? *(const f32x4*)(g.b_row_scale + ncol + 32 * pr + 4 * k) : (f32x4){1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wm * 128 + 16 * i + r;
        const float sa = g.a_row_scale ? g.a_row_scale[min(m, g.M - 1)] : g.a_scale;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            f32x4 a = acc[i][2 * pr], c = acc[i][2 * pr + 1];
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                         "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
            a = a * (sv[pr][0] * sa) + bv[pr][0];
            c = c * (sv[pr][1] * sa) + bv[pr][1];
            const bf16x8 o = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)c[0], (bf16)c[1], (bf16)c[2], (bf16)c[3]};
            if (m < g.M) *(bf16x8*)(Cp + (int64_t)m * g.ldc + ncol + 32 * pr) = o;
        }
    }
}


// ================================================================================================
// Role-split weight-stationary fp8 GEMM for K = 512 (the d = 512 Q|K|V|C projection of the fp8 mode, N = 2048): the e4m3 twin of
// gemm_wsr512_kernel<W5_PLAIN> (gemm_wsr.hip).  Twelve waves per CU:
//   waves 0-7  (GEMM role):     e4m3 W columns 32 g .. 32 g + 31 of a 256-column slab in 64 VGPRs (half of what bf16 takes); per 32-row
//                               A tile 16 block-scaled MFMAs (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales: 4 k-steps of 128) into
//                               the fp32 staging buffer of the step;
//   waves 8-11 (epilogue role): LDS-DMA of the NEXT tile (32 rows x 512 e4m3 bytes: four 1-KB pieces per wave), and the epilogue of the
//                               previous step: acc * (wscale[n] * xscale[m]) + bias -> bf16, 16-byte row-contiguous stores.
// The 256 x 256 fp8 tile this replaces for that shape re-stages W through LDS for every 256 rows and ran at 496 us per launch against
// 251 us for the bf16 role-split kernel: the mode that exists to be faster was slower (profiles/r03).  One s_barrier per step.
// LDS: two A tiles (2 x 16 KB) + two staging buffers (2 x 33 KB).
// ================================================================================================
struct Wsr8Cfg {
    static constexpr int TR = 32, ROWB = 512, TILEB = TR * ROWB, NR = 2, ES = 256 + 4, STG = TR * ES * 4;
    static constexpr int SMEM = NR * TILEB + 2 * STG;
};
typedef __attribute__((address_space(3))) void lds_void_f8_t;
typedef __attribute__((address_space(1))) const void gbl_void_f8_t;

__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_wsr512_f8_kernel(GemmF8 g) {
    using C = Wsr8Cfg;
    typedef int i32x8_t __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (row-range slot x, column slab y); the slabs of one x sit on one XCD (ids b, b + 8, ...): their A re-reads hit its L2
    const int ny = g.N / 256;
    const int b = blockIdx.x;
    const int y = (b >> 3) % ny, x = (b & 7) + 8 * (b / (8 * ny));
    const int gx = gridDim.x / ny;
    const int nb = y * 256;
    const int num_mt = (g.M + C::TR - 1) / C::TR;
    const int n = x < num_mt ? (num_mt - x + gx - 1) / gx : 0;      // steps of this workgroup: tiles x, x + gx, ...
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_f8_t*)smem;
    const uint32_t stg0 = lds0 + C::NR * C::TILEB;

    if (wave < 8) {
        // ================================================================ GEMM role
        const int gw = wave, r = lane & 15, q = lane >> 4;
        i32x8_t wf[2][4];      // [column tile][k-step]: lane (r, q) holds k = 128 ks + 32 q .. + 31 of W row nb + 32 gw + 16 j + r
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const char* wp = (const char*)g.B + (int64_t)(nb + 32 * gw + 16 * j + r) * g.ldb + 128 * ks + 32 * q;
                const u32x4 lo = *(const u32x4*)wp, hi = *(const u32x4*)(wp + 16);
                wf[j][ks] = (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            }
        // fragment address of k-step 0, first 16-byte half, row tile 0: row r, chunk slot (2 q) ^ (r & 15); second half = XOR 16,
        // k-step ks = XOR (ks << 7) (chunk bits 3, 4), row tile 1 = + 16 rows
        const uint32_t fr0 = (uint32_t)(r * C::ROWB + (((2 * q) ^ (r & 15)) << 4));
        const uint32_t sw0 = (uint32_t)((4 * q * C::ES + 32 * gw + r) * 4);      // staging address of acc[0][0][0]: row 4 q, column 32 gw + r
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // W fragments
        __builtin_amdgcn_s_barrier();
        for (int it = 0; it <= n; ++it) {
            if (it < n) {
                const uint32_t ab = lds0 + (uint32_t)((it & 1) * C::TILEB);
                f32x4 acc[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                u32x4 fa[2][2][2];      // [k-step parity][row tile][16-byte half]
                auto rd = [&](int ks) __attribute__((always_inline)) {
                    uint32_t f0 = fr0;
                    asm volatile("" : "+v"(f0));
                    const uint32_t ad = (f0 ^ (uint32_t)(ks << 7)) + ab, ad1 = ad ^ 16u;
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %5 offset:8192"
                                 : "=&v"(fa[ks & 1][0][0]), "=&v"(fa[ks & 1][0][1]), "=&v"(fa[ks & 1][1][0]), "=&v"(fa[ks & 1][1][1])
                                 : "v"(ad), "v"(ad1) : "memory");
                };
                rd(0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks + 1 < 4) {
                        rd(ks + 1);
                        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[ks & 1][0][0]), "+v"(fa[ks & 1][0][1]), "+v"(fa[ks & 1][1][0]), "+v"(fa[ks & 1][1][1]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[ks & 1][0][0]), "+v"(fa[ks & 1][0][1]), "+v"(fa[ks & 1][1][0]), "+v"(fa[ks & 1][1][1]));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const u32x4 lo = fa[ks & 1][i][0], hi = fa[ks & 1][i][1];
                        const i32x8_t af = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, wf[j][ks], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    }
                }
                // (inline-asm staging writes: the hazard recogniser does not put the wait states between an MFMA and an LDS instruction
                // that reads its result there; the s_nop is tied to the four accumulator tiles)
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]) :: "memory");
                const uint32_t sb = stg0 + (uint32_t)((it & 1) * C::STG) + sw0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t ad = sb + (uint32_t)(((16 * i + e) * C::ES + 16 * j) * 4);
                            asm volatile("ds_write_b32 %0, %1" :: "v"(ad), "v"(acc[i][j][e]) : "memory");
                        }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // staging tile in LDS before the barrier
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    // ==================================================================== epilogue role: 256 threads, 32 lanes per row, 8 rows per pass; also the DMA engine
    const int te = tid - 512, dw = wave - 8;
    const int erow = te >> 5, ecol = (te & 31) * 8;
    // LDS-DMA of one A tile: wave dw moves rows 8 dw .. 8 dw + 7 as four 1-KB pieces (two 512-byte rows each); LDS chunk slot `lane & 31` of a
    // row takes source chunk slot ^ (row & 15)
    auto dma = [&](int t, int slot) __attribute__((always_inline)) {
        const int mt = x + t * gx;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = 8 * dw + 2 * p + (lane >> 5);
            const int m = min(mt * C::TR + row, g.M - 1);
            const char* src = (const char*)g.A + (int64_t)m * g.lda + (((lane & 31) ^ (row & 15)) << 4);
            __builtin_amdgcn_global_load_lds((gbl_void_f8_t*)src, (lds_void_f8_t*)(smem + slot * C::TILEB + (8 * dw + 2 * p) * C::ROWB), 16, 0, 0);
        }
    };
    bf16* Cp = (bf16*)g.C;
    float bias[8], wsc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bias[e] = g.bias ? g.bias[nb + ecol + e] : 0.f;
        wsc[e] = g.b_row_scale ? g.b_row_scale[nb + ecol + e] : 1.f;
    }
    float pfx[4];      // dequantisation scales of this lane's four rows of the next tile
    auto load_xs = [&](int t) __attribute__((always_inline)) {
        const int mt = x + t * gx;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int m = min(mt * C::TR + erow + 8 * ps, g.M - 1);
            pfx[ps] = g.a_row_scale ? g.a_row_scale[m] : g.a_scale;
        }
    };
    load_xs(0);
    if (0 < n) dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it <= n; ++it) {
        if (it + 1 < n) dma(it + 1, (it + 1) & 1);      // first in the step: that slot held tile it - 1, every read of it drained before the last barrier
        bool full_tile = false;
        if (it >= 1) {
            const int tt = it - 1, mt = x + tt * gx;
            full_tile = mt * C::TR + C::TR <= g.M;
            const float* stage = (const float*)(smem + C::NR * C::TILEB + (tt & 1) * C::STG);
            const float xs[4] = {pfx[0], pfx[1], pfx[2], pfx[3]};
            load_xs(min(tt + 1, max(n - 1, 0)));      // unconditional (clamped): the next step's row scales travel under this step's arithmetic
            f32x4 sv[4][2];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                sv[ps][0] = *(const f32x4*)(stage + (erow + 8 * ps) * C::ES + ecol);
                sv[ps][1] = *(const f32x4*)(stage + (erow + 8 * ps) * C::ES + ecol + 4);
            }
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int m = mt * C::TR + erow + 8 * ps;
                if (m < g.M) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (bf16)(sv[ps][0][e] * (wsc[e] * xs[ps]) + bias[e]);
                        o[4 + e] = (bf16)(sv[ps][1][e] * (wsc[4 + e] * xs[ps]) + bias[4 + e]);
                    }
                    *(bf16x8*)(Cp + (int64_t)m * g.ldc + nb + ecol) = o;
                }
            }
        }
        // the tile of step it + 1 has landed for this wave: the counter retires in order, and a FULL tile's four row stores (every pass
        // stores, for every lane) were issued after the DMA pieces -- they may stay in flight (the row-scale loads only add to the margin)
        if (full_tile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

static bool f8_wsr512_ok(const GemmF8& g) {
    return !(g.opts & (OPT_TILE_GEMM | OPT_NO_ROLE_SPLIT_LN)) && g.K == 512 && g.N % 256 == 0 && g.N >= 256 && g.M >= 8192 && g.a_rows == nullptr && g.m_dev == nullptr &&
           g.lda % 16 == 0 && g.ldb % 16 == 0 && g.ldc % 8 == 0 && ((uintptr_t)g.C % 16) == 0 && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0;
}

static bool f8_big_ok(const GemmF8& g) {
    return g.a_rows == nullptr && g.m_dev == nullptr && g.M >= 4096 && g.N % 256 == 0 && g.K % 64 == 0 && g.K >= 128 && g.lda % 16 == 0 &&
           g.ldb % 16 == 0 && g.ldc % 8 == 0 && ((uintptr_t)g.C % 16) == 0;
}

int gemm_nt_f8(const GemmF8& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return 0;
    PMGT_CHECK(g.K > 0 && g.K % 16 == 0 && g.N % 4 == 0, -2, "gemm_nt_f8: K=%d must be a multiple of 16, N=%d of 4", g.K, g.N);
    PMGT_CHECK(g.lda % 16 == 0 && g.ldb % 16 == 0 && g.ldc % 4 == 0, -2, "gemm_nt_f8: leading dimensions must keep 16-byte rows");
    PMGT_CHECK(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0 && ((uintptr_t)g.C % 8) == 0, -2, "gemm_nt_f8: unaligned operands");
    PMGT_CHECK((g.bias == nullptr || ((uintptr_t)g.bias % 16) == 0) && (g.b_row_scale == nullptr || ((uintptr_t)g.b_row_scale % 16) == 0), -2,
               "gemm_nt_f8: bias / scale vectors must be 16-byte aligned");
    if (f8_wsr512_ok(g)) {
        PMGT_SMEM_ATTR((const void*)gemm_wsr512_f8_kernel, Wsr8Cfg::SMEM);
        const int ny = g.N / 256, num_mt = cdiv(g.M, Wsr8Cfg::TR);
        const int gx = std::max(8, std::min(256 / ny, num_mt) / 8 * 8);      // multiple of 8 row slots, one 12-wave workgroup per CU
        note_launch(LT_F8_WSR512);
        hipLaunchKernelGGL(gemm_wsr512_f8_kernel, dim3(gx * ny), dim3(768), Wsr8Cfg::SMEM, st, g);
        PMGT_LAUNCH_OK();
        return 0;
    }
    if (f8_big_ok(g) && !(g.opts & OPT_TILE_GEMM)) {
        constexpr int smem = 4 * (256 + 256) * 64;
        PMGT_SMEM_ATTR((const void*)gemm_nt_f8_big_kernel, smem);
        note_launch(LT_F8_BIG);
        hipLaunchKernelGGL(gemm_nt_f8_big_kernel, dim3(cdiv(cdiv(g.M, 256), 8) * 8 * (g.N / 256)), dim3(512), smem, st, g);
        PMGT_LAUNCH_OK();
        return 0;
    }
    const int num_n = cdiv(g.N, 128), num_m = cdiv(g.M, 128);
    note_launch(LT_F8_TILE);
    hipLaunchKernelGGL(gemm_nt_f8_kernel, dim3(8 * num_n * cdiv(num_m, 8)), dim3(256), 0, st, g);
    PMGT_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// quantisation
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void load8f(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8f<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <> __device__ __forceinline__ void load8f<bf16>(const bf16* p, float (&v)[8]) {
    const bf16x8 a = *(const bf16x8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
}

// one wave per row: pass 1 = absmax, pass 2 = scale + convert (the row comes back from L2)
template <typename T>
__device__ __forceinline__ void quant_row(const T* src, int cols, char* dst, float* scale_out, int lane) {
    float m = 0.f;
    for (int c0 = lane * 8; c0 < cols; c0 += 64 * 8) {
        float v[8];
        load8f<T>(src + c0, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
    }
    m = wave_max(m);
    const float inv = m > 0.f ? E4M3_MAX / m : 1.f;
    if (lane == 0) *scale_out = m > 0.f ? m / E4M3_MAX : 1.f;
    for (int c0 = lane * 8; c0 < cols; c0 += 64 * 8) {
        float v[8];
        load8f<T>(src + c0, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(v[e] * inv, -E4M3_MAX), E4M3_MAX);
        *(u32x2*)(dst + c0) = pack8_e4m3(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void quant_rows_kernel(const T* src, int64_t lds, int rows, int cols, char* dst, int64_t ldd, float* scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    quant_row<T>(src + (int64_t)row * lds, cols, dst + (int64_t)row * ldd, scale + row, threadIdx.x & 63);
}

template <typename T>
int quant_rows_e4m3(const T* src, int64_t lds, int rows, int cols, void* dst, int64_t ldd, float* scale, hipStream_t st) {
    if (rows <= 0) return 0;
    PMGT_CHECK(cols > 0 && cols % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0, -2, "quant_rows_e4m3: cols=%d and the leading dimensions must be multiples of 8", cols);
    PMGT_CHECK(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, -2, "quant_rows_e4m3: unaligned buffers");
    hipLaunchKernelGGL((quant_rows_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), 0, st, src, lds, rows, cols, (char*)dst, ldd, scale);
    PMGT_LAUNCH_OK();
    return 0;
}
template int quant_rows_e4m3<float>(const float*, int64_t, int, int, void*, int64_t, float*, hipStream_t);
template int quant_rows_e4m3<bf16>(const bf16*, int64_t, int, int, void*, int64_t, float*, hipStream_t);

__global__ __launch_bounds__(256) void quant_params_kernel(const float* params, const QuantDesc* desc, int n_desc, int total_rows, char* dst,
                                                           float* scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_rows) return;
    int k = 0;
    while (k + 1 < n_desc && row >= desc[k + 1].row_start) ++k;
    const QuantDesc d = desc[k];
    const int lr = row - d.row_start;
    quant_row<float>(params + d.src + (int64_t)lr * d.cols, d.cols, dst + d.dst + (int64_t)lr * d.cols, scale + d.scale + lr, threadIdx.x & 63);
}

int quant_params_e4m3(const float* params, const QuantDesc* desc_dev, int n_desc, int total_rows, void* dst, float* scale, hipStream_t st) {
    if (total_rows <= 0 || n_desc <= 0) return 0;
    hipLaunchKernelGGL(quant_params_kernel, dim3(cdiv(total_rows, 4)), dim3(256), 0, st, params, desc_dev, n_desc, total_rows, (char*)dst, scale);
    PMGT_LAUNCH_OK();
    return 0;
}

__global__ __launch_bounds__(256) void quant_tensor_kernel(const float* src, char* dst, int64_t n8, float inv) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8];
        load8f<float>(src + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(v[e] * inv, -E4M3_MAX), E4M3_MAX);
        *(u32x2*)(dst + i * 8) = pack8_e4m3(v);
    }
}
__global__ __launch_bounds__(256) void dequant_tensor_kernel(const char* src, float* dst, int64_t n8, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8];
        unpack8_e4m3(*(const u32x2*)(src + i * 8), v);
        f32x4 a = {v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale}, b = {v[4] * scale, v[5] * scale, v[6] * scale, v[7] * scale};
        *(f32x4*)(dst + i * 8) = a;
        *(f32x4*)(dst + i * 8 + 4) = b;
    }
}

int quant_tensor_e4m3(const float* src, void* dst, int64_t n, float inv_scale, hipStream_t st) {
    if (n <= 0) return 0;
    PMGT_CHECK(n % 8 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, -2, "quant_tensor_e4m3: n=%lld must be a multiple of 8, buffers aligned", (long long)n);
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(quant_tensor_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n8, 256), 4096)), dim3(256), 0, st, src, (char*)dst, n8, inv_scale);
    PMGT_LAUNCH_OK();
    return 0;
}
int dequant_tensor_e4m3(const void* src, float* dst, int64_t n, float scale, hipStream_t st) {
    if (n <= 0) return 0;
    PMGT_CHECK(n % 8 == 0 && ((uintptr_t)src % 8) == 0 && ((uintptr_t)dst % 16) == 0, -2, "dequant_tensor_e4m3: n=%lld must be a multiple of 8, buffers aligned", (long long)n);
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(dequant_tensor_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n8, 256), 4096)), dim3(256), 0, st, (const char*)src, dst, n8, scale);
    PMGT_LAUNCH_OK();
    return 0;
}

}  // namespace pmgt
