// Weight-stationary streaming GEMM for the d x d layers (bf16, K in {64, 128, 256, 512}).
//
//   C[M, N] = epi(A[M, K] * W[N, K]^T)   with optional fused LayerNorm over the full row (N == 256)
//
// At K = N = 256 these GEMMs have ~130 flop/byte: far below the MFMA/HBM ridge of MI355X, so the
// right structure is a STREAM of A, not a tiled matmul.  One 512-thread workgroup per CU keeps a
// 256-column slab of W in registers for its whole life (wave w owns output columns 32w..32w+31 as
// 2 x KS MFMA B-fragments = 64 VGPRs at K = 256) and loops over 64-row tiles of A:
//   global -> registers (two tiles in flight) -> XOR-swizzled LDS tile -> ds_read_b128 A fragments
//   shared by the 8 waves -> 64 MFMAs per wave -> fp32 staging in LDS -> row-contiguous epilogue
//   where 32 lanes own one output row: bias, GELU / GELU', dropout, residual and (optionally) the
//   LayerNorm of the row, all with 16-byte global accesses.
// A is read from HBM exactly once, W once per workgroup; column slabs of one row range (N > 256) are
// placed on the same XCD so their A re-reads hit that XCD's L2.
#include <type_traits>

#include "fp8.h"
#include "gemm.h"

namespace pmgt {

// NW = waves per workgroup.  8: one 512-thread workgroup per CU, 64-row tiles, wave w owns 32 output columns.  4: TWO 256-thread
// workgroups per CU (as qkvc_attn_fwd2_kernel), 32-row tiles, wave w owns 64 output columns (128 VGPRs of W at K = 256): in-kernel
// stamps show the 8-wave form marching through an MFMA phase (2.8k cycles per tile) and a VALU-bound epilogue phase (9k cycles with
// the fused LayerNorm) in lockstep between barriers; two workgroups drift out of phase, so one's epilogue overlaps the other's MFMAs.
template <int KS, int NW = 8> struct WsCfg {
    static constexpr int K = 32 * KS;
    static constexpr int NT = 64 * NW;                 // threads
    static constexpr int TR = 8 * NW;                  // tile rows
    static constexpr int IT = TR / 16;                 // 16-row MFMA tiles per wave
    static constexpr int JW = 16 / NW;                 // 16-column MFMA tiles per wave
    static constexpr int PR = NT / 32;                 // rows per epilogue pass (32 lanes per row): 4 passes per tile
    static constexpr int ROWB = K * 2;                 // A tile row bytes
    static constexpr int CPR = K / 8;                  // 16-byte chunks per row
    static constexpr int TILEB = TR * ROWB;
    static constexpr int LPT = TR * CPR / NT;          // chunks per thread per tile (K >= 64)
    static_assert(LPT * NT == TR * CPR, "tile must be a whole number of chunks per thread");
    static_assert(NW == 8 || NW == 4, "waves per workgroup");
    // K = 512 (KS = 16, the d = 512 shapes): the 256-column slab of W takes 128 VGPRs per lane in the 8-wave form and a 64-row A
    // tile 64 KB of LDS, so there is ONE LDS buffer and the next tile goes into it by LDS-DMA (no staging registers: a row is
    // 1 KB = one global_load_lds instruction, the chunk swizzle applied on the source side) right after the second barrier,
    // when every wave is done with its fragments; the epilogue phase covers its latency
    static constexpr bool DMA = KS == 16;
    static constexpr bool ONE = NW == 4 || KS == 16;     // one tile in flight
    static constexpr int NBUF = KS == 16 ? 1 : 2;
    static_assert(!(NW == 4 && KS == 16), "two workgroups per CU do not fit at K = 512");
    static constexpr int ES = 256 + 4;                 // staging row stride (floats)
    static constexpr int SMEM = NBUF * TILEB + TR * ES * 4 + 3 * 256 * 4;   // A ring + staging + bias/gamma/beta
    static constexpr int SWZ = CPR >= 16 ? 15 : CPR - 1;
};

// MODE (compile time, so every load in the loop is unconditional and the compiler can count vmcnt):
//   0 bias only | 1 bias + GELU, stores the pre-activation | 2 GELU' of a loaded pre-activation
//   3 bias + dropout + residual | 4 = 3 + fused LayerNorm of the row
enum { WS_PLAIN = 0, WS_GELU = 1, WS_GELU_GRAD = 2, WS_RES = 3, WS_RES_LN = 4 };

#ifdef PMGT_WS_PROF
// cycles per tile phase of gemm_ws_kernel: [block slot][wave][phase], phase 7 = tiles
__device__ unsigned int g_ws_prof[2][8][8];
#define WS_STAMP(k_) do { const unsigned long long n_ = __builtin_readcyclecounter(); pacc[k_] += (unsigned int)(n_ - plast); plast = n_; } while (0)
#else
#define WS_STAMP(k_) do { } while (0)
#endif

template <int KS, int MODE, int NW>
__device__ __forceinline__ void gemm_ws_body(const GemmWS& g_) {
    // device-side row count (compacted rows of the last-layer shortcut): the grid is sized for the host bound g_.M, the row loop for
    // min(M, *m_dev); a workgroup without rows leaves before its first barrier
    GemmWS g = g_;
    if (g_.m_dev) {
        g.M = min(g_.M, *g_.m_dev);
        if (g.M <= 0) return;
    }
    constexpr bool HAS_PF = MODE == WS_GELU_GRAD || MODE == WS_RES || MODE == WS_RES_LN;
    using C = WsCfg<KS, NW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    float* stage = (float*)(smem + C::NBUF * C::TILEB);
    float* cvec = stage + C::TR * C::ES;       // [3][256]: bias, LN gamma, LN beta of this column slab (LDS, so the
                                            // epilogue never waits on vmcnt for them)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;

    // block -> (row-range slot x, column slab y); slabs of one x sit on one XCD (ids b, b + 8, ...)
    const int ny = (g.N + 255) / 256;
    const int b = blockIdx.x;
    const int y = (b >> 3) % ny, x = (b & 7) + 8 * (b / (8 * ny));
    const int gx = gridDim.x / ny;
    const int nb = y * 256;
    const int num_mt = (g.M + C::TR - 1) / C::TR;

    // ---- resident W fragments: rows n = nb + 16 JW wave + 16 j + r, k = 32 ks + 8 q
    bf16x8 wf[C::JW][KS];
#pragma unroll
    for (int j = 0; j < C::JW; ++j) {
        const int n = nb + 16 * C::JW * wave + 16 * j + r;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (n < g.N) wf[j][ks] = *(const bf16x8*)((const bf16*)g.B + (int64_t)n * g.ldb + 32 * ks + 8 * q);
            else wf[j][ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    if (tid < 256) {      // (every thread of the 4-wave form)
        const int n = nb + tid;
        cvec[tid] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
        cvec[256 + tid] = (g.ln_out && n < g.N) ? g.ln_gamma[n] : 0.f;
        cvec[512 + tid] = (g.ln_out && n < g.N) ? g.ln_beta[n] : 0.f;
    }

    u32x4 ra[C::DMA ? 1 : (C::ONE ? 1 : 2)][C::DMA ? 1 : C::LPT];
    typedef __attribute__((address_space(3))) void lds_void_ws_t;
    typedef __attribute__((address_space(1))) const void gbl_void_ws_t;
    auto dma_tile = [&](int mt) {       // (KS == 16) 64 rows x 1 KB: wave w moves rows 8 w .. 8 w + 7, LDS slot `lane` <- source chunk lane ^ (row & 15)
        const int uw = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = 8 * uw + j;
            const int m = min(mt * C::TR + row, g.M - 1);
            const char* src = (const char*)g.A + ((int64_t)m * g.lda) * 2 + ((lane ^ (row & 15)) << 4);
            __builtin_amdgcn_global_load_lds((gbl_void_ws_t*)src, (lds_void_ws_t*)(sA + row * C::ROWB), 16, 0, 0);
        }
    };
    auto gload = [&](int mt, int set) {
#pragma unroll
        for (int i = 0; i < C::LPT; ++i) {
            const int idx = tid + C::NT * i;
            const int row = idx / C::CPR, ch = idx % C::CPR;
            const int m = min(mt * C::TR + row, g.M - 1);
            ra[set][i] = *(const u32x4*)((const char*)g.A + ((int64_t)m * g.lda) * 2 + ch * 16);
        }
    };
    auto sstore = [&](int buf, int set) {
#pragma unroll
        for (int i = 0; i < C::LPT; ++i) {
            const int idx = tid + C::NT * i;
            const int row = idx / C::CPR, ch = idx % C::CPR;
            *(u32x4*)(sA + buf * C::TILEB + row * C::ROWB + ((ch ^ (row & C::SWZ)) << 4)) = ra[set][i];
        }
    };

    const DropKey dk = make_drop_key(g.drop);
    bf16* Cp = (bf16*)g.C;
    const bf16* R = (const bf16*)g.res;
    bf16* AUX = (bf16*)g.aux;
    bf16* LNO = (bf16*)g.ln_out;
    // epilogue ownership: 32 consecutive lanes = one row, 8 consecutive columns per lane
    const int erow = tid >> 5, ecol = (tid & 31) * 8;
    const bf16* PF = (MODE == WS_GELU_GRAD) ? (const bf16*)AUX : R;      // prefetched epilogue operand
    const int64_t ldpf = (MODE == WS_GELU_GRAD) ? g.ldaux : g.ldr;

    // One tile: register set / LDS buffer index P is a compile-time constant (the loop below is unrolled by
    // two), so the two prefetch register sets never get copied into each other and the compiler can leave the
    // younger tile's loads in flight (counted vmcnt) while this tile is consumed.
#ifdef PMGT_WS_PROF
    unsigned int pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long plast = __builtin_readcyclecounter();
#endif
    auto tile_step = [&](auto Pc, int mt) {
        constexpr int P = decltype(Pc)::value;
#ifdef PMGT_WS_PROF
        ++pacc[7];
#endif
        WS_STAMP(0);
        // epilogue operand (residual, or the saved pre-activation for GELU') of this tile: issued first so it
        // is OLDER than the A prefetch below; the epilogue can then wait for it with a counted vmcnt and leave
        // the prefetch in flight.
        // (4-wave form with a residual: no prefetch -- 128 of the 256 VGPRs hold W; the operand is loaded in its pass and the
        // other workgroup of the CU covers the latency)
        constexpr bool PF_EARLY = HAS_PF && !(NW == 4 && (MODE == WS_RES || MODE == WS_RES_LN));
        bf16x8 pf[4];
        auto load_pf = [&]() {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {      // clamped, unconditional loads (values of padding rows are unused)
                const int m = min(mt * C::TR + erow + C::PR * ps, g.M - 1), n = min(nb + ecol, g.N - 8);
                pf[ps] = *(const bf16x8*)(PF + (int64_t)m * ldpf + n);
            }
        };
        // (K = 512 form: the barrier below waits for the LDS-DMA of this tile with vmcnt(0), which would expose these loads' latency
        // too -- they go out after it instead, under the MFMA phase)
        if constexpr (PF_EARLY && !C::DMA) load_pf();
        if constexpr (!C::ONE) {
            sstore(P, P);
            if (mt + 2 * gx < num_mt) gload(mt + 2 * gx, P);
        }
        WS_STAMP(1);
        // K = 512 form: this tile's LDS-DMA (issued behind the previous step's second barrier) must have landed for EVERY wave before
        // anyone reads fragments.  An explicit wait: __syncthreads() only orders what the memory model makes the compiler wait for, and
        // the unrolled loop's back edge reached this barrier with lgkmcnt(0) alone (rows of the incoming tile still in flight: wrong and
        // run-to-run different outputs on some boxes, tests/test_ops_gpu.py::test_role_split_k512_*).
        if constexpr (C::DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        WS_STAMP(2);
        // 4-wave form: ONE tile in flight in registers (the W fragments take 128 of the 256): loaded here, at the start of the
        // MFMA phase, parked in the other LDS buffer after the second barrier (that buffer was last read in the previous step's
        // MFMA phase)
        if constexpr (PF_EARLY && C::DMA) load_pf();
        const bool more4 = C::ONE && mt + gx < num_mt;
        if constexpr (C::ONE && !C::DMA) { if (more4) gload(mt + gx, 0); }
        f32x4 acc[C::IT][C::JW];
#pragma unroll
        for (int i = 0; i < C::IT; ++i)
#pragma unroll
            for (int j = 0; j < C::JW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* a_base = sA + (C::NBUF == 1 ? 0 : P) * C::TILEB;
        if constexpr (NW == 4) __builtin_amdgcn_s_setprio(3);      // the MFMA phase of this workgroup over the other one's VALU phase
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 fa[C::IT];
#pragma unroll
            for (int i = 0; i < C::IT; ++i) {
                const int row = 16 * i + r;
                fa[i] = *(const bf16x8*)(a_base + row * C::ROWB + (((4 * ks + q) ^ (row & C::SWZ)) << 4));
            }
#pragma unroll
            for (int i = 0; i < C::IT; ++i)
#pragma unroll
                for (int j = 0; j < C::JW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], wf[j][ks], acc[i][j], 0, 0, 0);
        }
        WS_STAMP(3);
        // ---- stage the TR x 256 fp32 tile (row = 16 i + 4 q + e, col = 16 JW wave + 16 j + r)
#pragma unroll
        for (int i = 0; i < C::IT; ++i)
#pragma unroll
            for (int j = 0; j < C::JW; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    stage[(16 * i + 4 * q + e) * C::ES + 16 * C::JW * wave + 16 * j + r] = acc[i][j][e];
        if constexpr (NW == 4) __builtin_amdgcn_s_setprio(0);
        WS_STAMP(4);
        __syncthreads();
        WS_STAMP(5);
        if constexpr (C::DMA) { if (more4) dma_tile(mt + gx); }
        else if constexpr (C::ONE) { if (more4) sstore(P ^ 1, 0); }
        // ---- row-contiguous epilogue: PR rows per pass, 4 passes
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = erow + C::PR * ps;
            const int m = mt * C::TR + row;
            const int n = nb + ecol;
            const bool ok = m < g.M && n < g.N;
            float v[8];
            if (ok) {
                const f32x4 s0 = *(const f32x4*)(stage + row * C::ES + ecol), s1 = *(const f32x4*)(stage + row * C::ES + ecol + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = s0[e]; v[4 + e] = s1[e]; }
                {
                    const f32x4 b0 = *(const f32x4*)(cvec + ecol), b1 = *(const f32x4*)(cvec + ecol + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                }
                if constexpr (MODE == WS_GELU) {
                    bf16x8 pre;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { pre[e] = (bf16)v[e]; v[e] = gelu_fast((float)pre[e]); }
                    *(bf16x8*)(AUX + (int64_t)m * g.ldaux + n) = pre;
                } else if constexpr (MODE == WS_GELU_GRAD) {
                    bf16x8 pre;
                    if constexpr (PF_EARLY) pre = pf[ps];
                    else pre = *(const bf16x8*)(PF + (int64_t)m * ldpf + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= gelu_fast_grad((float)pre[e]);
                }
                if ((MODE == WS_RES || MODE == WS_RES_LN) && dk.on) {
                    float d0[4], d1[4];
                    drop_mul4(dk, (uint32_t)m, (uint32_t)n >> 2, d0);
                    drop_mul4(dk, (uint32_t)m, ((uint32_t)n >> 2) + 1, d1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] *= d0[e]; v[4 + e] *= d1[e]; }
                }
                if constexpr (MODE == WS_RES || MODE == WS_RES_LN) {
                    bf16x8 rv;
                    if constexpr (PF_EARLY) rv = pf[ps];
                    else rv = *(const bf16x8*)(PF + (int64_t)m * ldpf + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) { o[e] = (bf16)v[e]; v[e] = (float)o[e]; }   // LN sees what backward re-reads
#ifdef PMGT_WS_ABL_NO_C
                if (g.M < 0)
#endif
                if (!(MODE == WS_RES_LN && g.skip_c)) *(bf16x8*)(Cp + (int64_t)m * g.ldc + n) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            if constexpr (MODE == WS_RES_LN) {      // fused LayerNorm over the row (host guarantees N == 256: 32 lanes x 8 columns)
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[e];
                s = sum_lanes32(s);
                const float mean = s * (1.f / 256.f);
                float ss = 0.f, tc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { tc[e] = v[e] - mean; ss = fmaf(tc[e], tc[e], ss); }
                ss = sum_lanes32(ss);
                // v_rsq_f32 (1 ulp) instead of the IEEE sqrt + divide sequences (~40 VALU instructions per row pass in an epilogue that
                // in-kernel stamps show VALU-bound: 5.4k / 9.1k cycles per tile on the older / younger wave of a SIMD)
                const float rstd = __builtin_amdgcn_rsqf(ss * (1.f / 256.f) + g.ln_eps);
                bf16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) {
#ifdef PMGT_WS_ABL_NO_STATS
                    if (g.M < 0)
#endif
                    if ((tid & 31) == 0) *(float2*)(g.ln_stats + 2 * (int64_t)m) = make_float2(mean, rstd);      // one 8-byte store (two store instructions cost this epilogue 3 %)
                    const f32x4 g0 = *(const f32x4*)(cvec + 256 + ecol), g1 = *(const f32x4*)(cvec + 256 + ecol + 4);
                    const f32x4 b0 = *(const f32x4*)(cvec + 512 + ecol), b1 = *(const f32x4*)(cvec + 512 + ecol + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (bf16)(tc[e] * rstd * g0[e] + b0[e]);
                        o[4 + e] = (bf16)(tc[4 + e] * rstd * g1[e] + b1[e]);
                    }
#ifdef PMGT_WS_ABL_NO_LNO
                    if (g.M < 0)
#endif
                    *(bf16x8*)(LNO + (int64_t)m * g.ldc + n) = o;
                }
                if (g.q8) {      // (uniform) the row as e4m3 for the next layer's fp8 projection; every lane joins the row maximum
                    float f[8], mx = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)o[e];      // the bf16 values that were stored: the quantisation of the STORED x
#pragma unroll
                    for (int e = 0; e < 8; ++e) mx = raw_max(mx, fabsf(f[e]));
                    mx = max_lanes32(mx);
                    const float inv = mx > 0.f ? E4M3_MAX / mx : 1.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(f[e] * inv, -E4M3_MAX, E4M3_MAX);
                    if (ok) {
                        *(u32x2*)((char*)g.q8 + (int64_t)m * g.N + n) = pack8_e4m3(f);
                        if ((tid & 31) == 0) g.q8_scale[m] = mx > 0.f ? mx / E4M3_MAX : 1.f;
                    }
                }
            }
            if constexpr (NW == 4) __builtin_amdgcn_sched_barrier(0);      // one pass at a time: 128 of the 256 VGPRs hold W
        }
        // the next tile's staging writes happen after its first barrier, which every wave reaches only after
        // finishing this epilogue; its A-tile write targets the other LDS buffer.
    };

    int mt = x;
    if constexpr (C::DMA) {
        if (mt < num_mt) dma_tile(mt);
    } else {
        if (mt < num_mt) gload(mt, 0);
        if constexpr (!C::ONE) { if (mt + gx < num_mt) gload(mt + gx, 1); }
        else { if (mt < num_mt) sstore(0, 0); }
    }
    while (mt < num_mt) {
        tile_step(std::integral_constant<int, 0>{}, mt);
        mt += gx;
        if (mt >= num_mt) break;
        tile_step(std::integral_constant<int, 1>{}, mt);
        mt += gx;
    }
#ifdef PMGT_WS_PROF
    WS_STAMP(6);
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133))
        for (int k_ = 0; k_ < 8; ++k_) g_ws_prof[blockIdx.x == 0 ? 0 : 1][wave][k_] = pacc[k_];
#endif
}

template <int KS, int MODE>
__global__ __launch_bounds__(512) void gemm_ws_kernel(GemmWS g) {
    gemm_ws_body<KS, MODE, 8>(g);
}
template <int KS, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_ws2_kernel(GemmWS g) {
    gemm_ws_body<KS, MODE, 4>(g);
}

static int ws_mode(const GemmWS& g) {
    const bool drop = g.drop.p > 0.f;
    if (g.epi == EPI_NONE && !g.res && !drop) return WS_PLAIN;
    if (g.epi == EPI_GELU && !g.res && !drop) return WS_GELU;
    if (g.epi == EPI_GELU_GRAD && !g.res && !drop) return WS_GELU_GRAD;
    if (g.epi == EPI_NONE && g.res) return (g.ln_out && g.N == 256 && !(g.opts & OPT_UNFUSED_LN)) ? WS_RES_LN : WS_RES;
    return -1;
}

bool gemm_ws_supported(const GemmWS& g) {
    return ws_mode(g) >= 0 && g.a_rows == nullptr && g.K % 32 == 0 && g.K >= 64 && g.K <= 512 && (g.K & (g.K - 1)) == 0 && g.N % 8 == 0 && g.N >= 8 &&
           g.M >= 64 && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.ldc % 8 == 0 && (g.res == nullptr || g.ldr % 8 == 0) &&
           (g.aux == nullptr || g.ldaux % 8 == 0);
}
bool gemm_ws_fuses_ln(const GemmWS& g) { return gemm_ws_supported(g) && ws_mode(g) == WS_RES_LN; }

// Which form runs (A/B on one box, c2 shapes, ms per step of the phase, 8-wave -> two 4-wave workgroups):
//   GELU (ffn1) 0.389 -> 0.362, GELU' (dgrad_ffn2) 0.467 -> 0.438, plain (dgrad_attn_out) 0.250 -> 0.243: VALU-heavy epilogues
//   with nothing but the A stream to prefetch -- the overlap pays;
//   residual (dgrad_ffn1) 0.333 -> 0.42, residual + LayerNorm (attn_out, ffn2) 0.56 -> 0.60 / 0.54 -> 0.58 (with the
//   residual loaded in its pass instead of prefetched, so that nothing spills next to the 128 VGPRs of W: 254 VGPRs; with
//   the prefetch 30 were spilled and the launches took 0.68 / 0.66): these launches are served from the Infinity Cache to
//   a large part, and one tile in flight per workgroup instead of two costs them memory-level parallelism.
// (the two-workgroup form needs K >= 128: chunks per thread)
static int ws_form(int mode) {
    return (mode == WS_GELU || mode == WS_GELU_GRAD || mode == WS_PLAIN) ? 2 : 1;
}

template <int KS, int MODE> static int launch_ws(const GemmWS& g, hipStream_t st) {
    const int ny = cdiv(g.N, 256);
    if constexpr (KS >= 4 && KS <= 8) {
        if (ws_form(MODE) == 2) {
            using C = WsCfg<KS, 4>;
            auto kern = gemm_ws2_kernel<KS, MODE>;
            PMGT_SMEM_ATTR((const void*)kern, C::SMEM);
            const int num_mt = cdiv(g.M, C::TR);
            const int gx = std::max(8, std::min(512 / ny, num_mt) / 8 * 8);      // two workgroups per CU
            note_launch(LT_GEMM_WS);
            hipLaunchKernelGGL(kern, dim3(gx * ny), dim3(C::NT), C::SMEM, st, g);
            PMGT_LAUNCH_OK();
            return 0;
        }
    }
    using C = WsCfg<KS, 8>;
    auto kern = gemm_ws_kernel<KS, MODE>;
    PMGT_SMEM_ATTR((const void*)kern, C::SMEM);
    const int num_mt = cdiv(g.M, 64);
    int gx = std::max(8, std::min(256 / ny, num_mt) / 8 * 8);      // multiple of 8 row slots, ~one workgroup per CU
    note_launch(LT_GEMM_WS);
    hipLaunchKernelGGL(kern, dim3(gx * ny), dim3(512), C::SMEM, st, g);
    PMGT_LAUNCH_OK();
    return 0;
}

template <int KS> static int launch_mode(const GemmWS& g, hipStream_t st) {
    switch (ws_mode(g)) {
        case WS_PLAIN: return launch_ws<KS, WS_PLAIN>(g, st);
        case WS_GELU: return launch_ws<KS, WS_GELU>(g, st);
        case WS_GELU_GRAD: return launch_ws<KS, WS_GELU_GRAD>(g, st);
        case WS_RES: return launch_ws<KS, WS_RES>(g, st);
        default: return launch_ws<KS, WS_RES_LN>(g, st);
    }
}

int gemm_ws(const GemmWS& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return 0;
    PMGT_CHECK(gemm_ws_supported(g), -2, "gemm_ws: unsupported shape/epilogue M=%d N=%d K=%d", g.M, g.N, g.K);
    PMGT_CHECK(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0 && ((uintptr_t)g.C % 16) == 0, -2, "gemm_ws: unaligned operands");
    if (ws_mode(g) == WS_RES_LN && gemm_wsr_ok(g)) return gemm_wsr(g, st);
    if (ws_mode(g) != WS_RES_LN && gemm_wsr512_ok(g)) return gemm_wsr512(g, st);
    switch (g.K) {
        case 64: return launch_mode<2>(g, st);
        case 128: return launch_mode<4>(g, st);
        case 512: return launch_mode<16>(g, st);
        default: return launch_mode<8>(g, st);
    }
}

}  // namespace pmgt

#ifdef PMGT_WS_PROF
extern "C" int pmgt_debug_ws_prof_read(unsigned int* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmgt::g_ws_prof), sizeof(pmgt::g_ws_prof));
}
#endif
