// Role-split weight-stationary GEMM with the residual + LayerNorm epilogue (bf16, K = N = 256): the attention-output and FFN2
// launches of the forward pass,  y = LayerNorm(dropout(A W^T + b) + res).
//
// gemm_ws_kernel<8, WS_RES_LN> runs its 64 MFMAs per wave and 64-row tile and then ~760 VALU instructions of epilogue per wave, in
// sequence inside one 8-wave workgroup per CU: 3.7 TB/s on 3U of traffic, vector-issue-bound.  Two workgroups per CU that drift out
// of phase would overlap the two phases, but a wave that holds 64 VGPRs of W next to the LayerNorm epilogue's temporaries does not fit
// the 128 registers sixteen waves leave (52 spilled: profiles/r03).  Here ONE 16-wave workgroup per CU splits the ROLES, as the fused
// attention kernels do -- a role needs only its own registers:
//   waves 0-7  (GEMM role):     W columns 32 g .. 32 g + 31 in 64 VGPRs; the 32-row A tile of step t arrives by LDS-DMA in a 4-slot ring
//                               (three steps ahead, counted vmcnt), 32 MFMAs, the fp32 result into staging buffer t & 1;
//   waves 8-15 (epilogue role): bias, dropout, residual (prefetched one step ahead), bf16 rounding, LayerNorm of the row (32 lanes per
//                               row), 16-byte stores -- of step t - 1, from staging buffer (t - 1) & 1, while the GEMM role computes step t.
// One s_barrier per step.  Results are bit-identical to gemm_ws_kernel<8, WS_RES_LN> (same arithmetic per element, same order).
#include "fp8.h"
#include "gemm.h"

namespace pmgt {

typedef __attribute__((address_space(3))) void lds_void_wsr_t;
typedef __attribute__((address_space(1))) const void gbl_void_wsr_t;

struct WsrCfg {
    static constexpr int TR = 32, ROWB = 512, TILEB = TR * ROWB, NR = 4, ES = 256 + 4, STG = TR * ES * 4;      // NR - 1 A tiles in flight
    static constexpr int SMEM = NR * TILEB + 2 * STG;
};

#ifdef PMGT_W5_PROF
__device__ unsigned long long g_w2_prof[16][8];
#define W2_STAMP(k_) do { const unsigned long long n_ = __builtin_readcyclecounter(); pacc[k_] += n_ - plast; plast = n_; } while (0)
#else
#define W2_STAMP(k_) do { } while (0)
#endif
// LNB = false: the forward form above.  LNB = true: the LayerNorm-BACKWARD epilogue (see the role below) on the same GEMM role.
template <bool LNB>
__global__ __launch_bounds__(1024) void gemm_wsr_kernel(GemmWS g) {
    using C = WsrCfg;
    if (g.m_dev) {      // device-side row count (see gemm_ws_body): uniform over the workgroup, taken before any barrier
        g.M = min(g.M, *g.m_dev);
        if (g.M <= 0) return;
    }
#ifdef PMGT_W5_PROF
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long plast = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = blockIdx.x, gx = gridDim.x;
    const int num_mt = (g.M + C::TR - 1) / C::TR;
    const int n = x < num_mt ? (num_mt - x + gx - 1) / gx : 0;      // steps of this workgroup: tiles x, x + gx, ...
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_wsr_t*)smem;
    const uint32_t stg0 = lds0 + C::NR * C::TILEB;

    if (wave < 8) {
        // ================================================================ GEMM role
        const int gw = wave, r = lane & 15, q = lane >> 4;
        bf16x8 wf[2][8];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                wf[j][ks] = *(const bf16x8*)((const bf16*)g.B + (int64_t)(32 * gw + 16 * j + r) * g.ldb + 32 * ks + 8 * q);
        // LDS-DMA of one A tile: wave gw moves rows 4 gw .. 4 gw + 3 as two 1-KB pieces (2 rows each); LDS slot `lane & 31` of a row
        // takes source chunk slot ^ (row & 15)
        auto dma = [&](int t, int slot) __attribute__((always_inline)) {
            const int mt = x + t * gx;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int row = 4 * gw + 2 * p + (lane >> 5);
                const int m = min(mt * C::TR + row, g.M - 1);
                const char* src = (const char*)g.A + (int64_t)m * g.lda * 2 + (((lane & 31) ^ (row & 15)) << 4);
                __builtin_amdgcn_global_load_lds((gbl_void_wsr_t*)src, (lds_void_wsr_t*)(smem + slot * C::TILEB + (4 * gw + 2 * p) * C::ROWB), 16, 0, 0);
            }
        };
        // fragment address of k-step 0 in row-tile 0: row r, chunk q ^ (r & 15); k-step ks = XOR (ks << 6); row-tile 1 = + 16 * ROWB
        const uint32_t fr0 = (uint32_t)(r * C::ROWB + ((((r >> 2) & 3) << 2 | (q ^ (r & 3))) << 4));
        // staging address of acc[0][0][0]: row 4 q, column 32 gw + r
        const uint32_t sw0 = (uint32_t)((4 * q * C::ES + 32 * gw + r) * 4);
        // counted wait: all but the `younger` most recently issued tiles (2 DMA pieces each) have landed
        auto wait_tiles = [&](int younger) __attribute__((always_inline)) {
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        static_assert(C::NR == 4, "wait_tiles: at most two younger tiles");
#pragma unroll
        for (int t = 0; t < C::NR - 1; ++t)
            if (t < n) dma(t, t);
        wait_tiles(min(C::NR - 2, n - 1));      // tile 0 has landed for this wave (the W loads are older and drain first)
        __builtin_amdgcn_s_barrier();
#ifdef PMGT_W5_PROF
        plast = __builtin_readcyclecounter();
#endif
        for (int it = 0; it <= n; ++it) {
            if (it < n) {
                const int slot = it % C::NR;
                if (it + C::NR - 1 < n) dma(it + C::NR - 1, (it + C::NR - 1) % C::NR);      // that slot held tile it - 1: every wave's reads of it drained before the last barrier
                W2_STAMP(0);
                const uint32_t ab = lds0 + (uint32_t)(slot * C::TILEB);
                f32x4 acc[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                u32x4 fa[2][2];      // [k-step parity][row tile]
                auto rd = [&](int ks) __attribute__((always_inline)) {
                    const uint32_t ad = (fr0 ^ (uint32_t)(ks << 6)) + ab;      // XOR inside the row, then the tile base
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192" : "=&v"(fa[ks & 1][0]), "=&v"(fa[ks & 1][1]) : "v"(ad) : "memory");
                };
                rd(0);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks + 1 < 8) {
                        rd(ks + 1);
                        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[ks & 1][0]), "+v"(fa[ks & 1][1]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[ks & 1][0]), "+v"(fa[ks & 1][1]));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[ks & 1][i]), wf[j][ks], acc[i][j], 0, 0, 0);
                }
                // (the staging writes below are inline asm: the compiler's hazard recogniser does not put the wait states between an MFMA
                // and an LDS instruction that reads its result there -- without them every row 4 q + 0 carried a stale value)
                // (tied to the four accumulator tiles: every MFMA is issued before it, every staging write after it)
                asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]) :: "memory");
                W2_STAMP(1);
                // fp32 tile -> staging buffer it & 1: element (row 16 i + 4 q + e, column 32 gw + 16 j + r)
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
                // the staging tile is in LDS before the barrier hands it to the epilogue role; the A tile of step it + 1 has landed
                // for this wave (issued a whole step ago; the pieces of step it + 2 stay in flight)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (it + 1 < n) wait_tiles(min(it + C::NR - 1, n - 1) - (it + 1));
                W2_STAMP(2);
            }
            __builtin_amdgcn_s_barrier();
            W2_STAMP(3);
        }
#ifdef PMGT_W5_PROF
        if (blockIdx.x == 40 && lane == 0) { for (int k_ = 0; k_ < 4; ++k_) g_w2_prof[wave][k_] = pacc[k_]; g_w2_prof[wave][7] = (unsigned long long)n; }
#endif
        if constexpr (LNB) __builtin_amdgcn_s_barrier();      // the epilogue role's column-sum exchange (one more barrier of the workgroup)
        return;
    }
    // ==================================================================== epilogue role: 512 threads, 32 lanes per row, 16 rows per pass
    const int te = tid - 512;
    const int erow = te >> 5, ecol = (te & 31) * 8;
    if constexpr (LNB) {
        // ---- LayerNorm backward of the row the GEMM role just produced: dy = A W^T + res is the gradient of a LayerNorm OUTPUT y the
        // forward kept (x^ = (y - beta) / gamma, rowops.h), so  dx = rstd (dy gamma - mean(dy gamma) - x^ mean(dy gamma x^))  leaves as
        // C (the residual branch) and, times the dropout mask of the dense layer in front of that LayerNorm, as lnb_dx_drop (what its
        // weight / data gradient GEMMs read); dy itself never reaches HBM.  dgamma | dbeta | dbias (column sums of dy x^, dy and the
        // bf16-rounded dx_drop) are carried in registers for the life of the workgroup: one [3][256] partial per workgroup.
        // (one 32-bit byte offset per row serves all four row-major operands: the host requires equal leading dimensions and < 4 GB
        // each, so every access is  scalar base + 32-bit lane offset  and no 64-bit lane addresses are kept alive)
        const char* R = (const char*)g.res;
        const char* Y = (const char*)g.lnb_y;
        char* DX = (char*)g.C;
        char* DXD = (char*)g.lnb_dx_drop;
        const uint32_t ldb2 = (uint32_t)g.ldc * 2u, ecol2 = (uint32_t)ecol * 2u;
        const DropKey dk = make_drop_key(g.lnb_drop);
        float gam[8], nbet[8], igam[8], dgam[8], dbet[8], dbia[8];      // x^ = y / gamma - beta / gamma
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gam[e] = g.lnb_gamma[ecol + e];
            igam[e] = gam[e] != 0.f ? __builtin_amdgcn_rcpf(gam[e]) : 0.f;      // (a dead channel: the host guard stores LayerNorm inputs instead, engine.py)
            nbet[e] = -g.lnb_beta[ecol + e] * igam[e];
            dgam[e] = dbet[e] = dbia[e] = 0.f;
        }
        // residual / y / rstd of this lane's two rows of a tile, loaded a whole step ahead: the registers of pass ps are refilled for
        // step t + 1 right after pass ps of step t has converted them (no second register set: sixteen waves leave 128 VGPRs each)
        bf16x8 pfr[2], pfy[2];
        float pfs[2];
        auto load_pf = [&](int t, int ps) __attribute__((always_inline)) {
            const uint32_t m = (uint32_t)min((x + t * gx) * C::TR + erow + 16 * ps, g.M - 1);
            const uint32_t off = m * ldb2 + ecol2;
            pfr[ps] = *(const bf16x8*)(R + off);
            pfy[ps] = *(const bf16x8*)(Y + off);
            pfs[ps] = *(const float*)((const char*)g.lnb_stats + (m * 8u + 4u));
        };
        load_pf(0, 0); load_pf(0, 1);      // (row indices are clamped: valid addresses even for a workgroup without tiles)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();      // the GEMM role's step 0 (this role trails it by one step: same number of barriers)
        for (int tt = 0; tt < n; ++tt) {
            {
                const int mt = x + tt * gx;
                const float* stage = (const float*)(smem + C::NR * C::TILEB + (tt & 1) * C::STG);
                const int tnext = min(tt + 1, n - 1);      // unconditional refill (the last step re-reads its own rows): no branch in the loop body
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int row = erow + 16 * ps;
                    const int m = mt * C::TR + row;
                    const bool ok = m < g.M;
                    const f32x4 s0 = *(const f32x4*)(stage + row * C::ES + ecol), s1 = *(const f32x4*)(stage + row * C::ES + ecol + 4);
                    float v[8], xh[8], gg[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = s0[e] + (float)pfr[ps][e]; v[4 + e] = s1[e] + (float)pfr[ps][4 + e]; }
                    if (!ok) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = 0.f;      // rows past M (clamped operands): no term in any sum, nothing stored
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) xh[e] = fmaf((float)pfy[ps][e], igam[e], nbet[e]);
                    const float rs = pfs[ps];
                    __builtin_amdgcn_sched_barrier(0);        // (the refill must not be scheduled above the conversions: it targets their registers)
                    load_pf(tnext, ps);                       // the next step's rows travel under the rest of this step
                    __builtin_amdgcn_sched_barrier(0);
                    float sg = 0.f, sgx = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
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
                    for (int e = 0; e < 8; ++e) { o[e] = (gg[e] - sg - xh[e] * sgx) * rs; ob[e] = (bf16)o[e]; }
                    const uint32_t off = (uint32_t)m * ldb2 + ecol2;
                    if (ok) *(bf16x8*)(DX + off) = ob;
                    if (DXD) {      // (uniform)
                        if (dk.on) {
                            float d0[4], d1[4];
                            drop_mul4(dk, (uint32_t)m, (uint32_t)ecol >> 2, d0);
                            drop_mul4(dk, (uint32_t)m, ((uint32_t)ecol >> 2) + 1, d1);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { o[e] *= d0[e]; o[4 + e] *= d1[e]; }
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) ob[e] = (bf16)o[e];
                        if (ok) *(bf16x8*)(DXD + off) = ob;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) dbia[e] += (float)ob[e];      // column sum of what the GEMMs behind it will read
                }
            }
            __builtin_amdgcn_s_barrier();
        }
        // ---- the sixteen row groups' column sums -> one [3][256] partial of this workgroup (the staging buffers are free: every
        // epilogue wave is past the loop's last barrier)
        float* red = (float*)(smem + C::NR * C::TILEB);      // [16][768]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[erow * 768 + ecol + e] = dgam[e];
            red[erow * 768 + 256 + ecol + e] = dbet[e];
            red[erow * 768 + 512 + ecol + e] = dbia[e];
        }
        __builtin_amdgcn_s_barrier();
        for (int idx = te; idx < 768; idx += 512) {
            float a = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) a += red[rr * 768 + idx];
            g.lnb_part[(int64_t)blockIdx.x * 768 + idx] = a;
        }
        return;
    }
    const DropKey dk = make_drop_key(g.drop);
    bf16* Cp = (bf16*)g.C;
    const bf16* R = (const bf16*)g.res;
    bf16* LNO = (bf16*)g.ln_out;
    float bias[8], gam[8], bet[8];      // this lane's eight columns, for the life of the workgroup
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bias[e] = g.bias ? g.bias[ecol + e] : 0.f;
        gam[e] = g.ln_gamma[ecol + e];
        bet[e] = g.ln_beta[ecol + e];
    }
    bf16x8 pf[2];
    auto load_pf = [&](int t) __attribute__((always_inline)) {
        const int mt = x + t * gx;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int m = min(mt * C::TR + erow + 16 * ps, g.M - 1);
            pf[ps] = *(const bf16x8*)(R + (int64_t)m * g.ldr + ecol);
        }
    };
    if (0 < n) load_pf(0);
    __builtin_amdgcn_s_barrier();
#ifdef PMGT_W5_PROF
    plast = __builtin_readcyclecounter();
#endif
    for (int it = 0; it <= n; ++it) {
        if (it >= 1) {
            const int tt = it - 1, mt = x + tt * gx;
            const float* stage = (const float*)(smem + C::NR * C::TILEB + (tt & 1) * C::STG);
            bf16x8 rv[2] = {pf[0], pf[1]};
            if (tt + 1 < n) load_pf(tt + 1);      // the next step's residual rows travel under this step's arithmetic
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int row = erow + 16 * ps;
                const int m = mt * C::TR + row;
                const bool ok = m < g.M;
                float v[8];
                const f32x4 s0 = *(const f32x4*)(stage + row * C::ES + ecol), s1 = *(const f32x4*)(stage + row * C::ES + ecol + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = s0[e] + bias[e]; v[4 + e] = s1[e] + bias[4 + e]; }
                if (dk.on) {
                    float d0[4], d1[4];
                    drop_mul4(dk, (uint32_t)m, (uint32_t)ecol >> 2, d0);
                    drop_mul4(dk, (uint32_t)m, ((uint32_t)ecol >> 2) + 1, d1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] *= d0[e]; v[4 + e] *= d1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv[ps][e];
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) { o[e] = (bf16)v[e]; v[e] = (float)o[e]; }   // LN sees what a stored C would hold: the same values whether or not it is stored (rows past M: clamped inputs, never stored)
                if (ok && !g.skip_c) *(bf16x8*)(Cp + (int64_t)m * g.ldc + ecol) = o;
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[e];
                s = sum_lanes32(s);
                const float mean = s * (1.f / 256.f);
                float ss = 0.f, tc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { tc[e] = v[e] - mean; ss = fmaf(tc[e], tc[e], ss); }
                ss = sum_lanes32(ss);
                const float rstd = __builtin_amdgcn_rsqf(ss * (1.f / 256.f) + g.ln_eps);
                if (ok) {
                    if ((te & 31) == 0) *(float2*)(g.ln_stats + 2 * (int64_t)m) = make_float2(mean, rstd);
                    bf16x8 y;
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = (bf16)(tc[e] * rstd * gam[e] + bet[e]);
                    *(bf16x8*)(LNO + (int64_t)m * g.ldc + ecol) = y;
                    if (g.q8) {      // (uniform) fp8 mode: the row additionally as e4m3 + one scale, for the next layer's fp8 projection (fp8.h contract)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (float)y[e];      // the quantisation of the STORED bf16 row
                    }
                }
                if (g.q8) {      // every lane of the row joins the maximum (rows past M: values unused)
                    float mx = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) mx = raw_max(mx, fabsf(v[e]));
                    mx = max_lanes32(mx);
                    const float inv = mx > 0.f ? E4M3_MAX / mx : 1.f;
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(v[e] * inv, -E4M3_MAX, E4M3_MAX);
                    if (ok) {
                        *(u32x2*)((char*)g.q8 + (int64_t)m * g.N + ecol) = pack8_e4m3(f);
                        if ((te & 31) == 0) g.q8_scale[m] = mx > 0.f ? mx / E4M3_MAX : 1.f;
                    }
                }
            }
        }
        W2_STAMP(0);
        __builtin_amdgcn_s_barrier();
        W2_STAMP(1);
    }
#ifdef PMGT_W5_PROF
    if (blockIdx.x == 40 && lane == 0) { for (int k_ = 0; k_ < 2; ++k_) g_w2_prof[wave][k_] = pacc[k_]; g_w2_prof[wave][7] = (unsigned long long)n; }
#endif
}

bool gemm_wsr_ok(const GemmWS& g) {
    return !(g.opts & (OPT_TILE_GEMM | OPT_NO_ROLE_SPLIT_LN | OPT_UNFUSED_LN)) && g.K == 256 && g.N == 256 && g.M >= 8192 && g.epi == EPI_NONE && g.res != nullptr && g.ln_out != nullptr && g.ln_stats != nullptr &&
           g.a_rows == nullptr && (g.q8 == nullptr || (g.q8_scale != nullptr && ((uintptr_t)g.q8 % 8) == 0)) && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.ldc % 8 == 0 && g.ldr % 8 == 0 &&
           ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0 && ((uintptr_t)g.C % 16) == 0 && ((uintptr_t)g.res % 16) == 0 &&
           ((uintptr_t)g.ln_out % 16) == 0 && ((uintptr_t)g.ln_stats % 8) == 0;
}

int gemm_wsr(const GemmWS& g, hipStream_t st) {
    PMGT_CHECK(gemm_wsr_ok(g), -2, "gemm_wsr: unsupported shape / epilogue M=%d N=%d K=%d", g.M, g.N, g.K);
    PMGT_SMEM_ATTR((const void*)gemm_wsr_kernel<false>, WsrCfg::SMEM);
    const int num_mt = cdiv(g.M, WsrCfg::TR);
    const int gx = std::max(8, std::min(256, num_mt) / 8 * 8);      // one 16-wave workgroup per CU
    note_launch(LT_GEMM_WSR);
    hipLaunchKernelGGL(gemm_wsr_kernel<false>, dim3(gx), dim3(1024), WsrCfg::SMEM, st, g);
    PMGT_LAUNCH_OK();
    return 0;
}

// ---- dy = A W^T + res followed by the LayerNorm backward of dy, in one launch (K = N = 256): the data-gradient GEMM in front of a
// LayerNorm site whose input the forward did not store.  One [3][256] partial (dgamma | dbeta | dbias) per workgroup.
int gemm_wsr_lnb_parts(int M) {
    const int num_mt = cdiv(M, WsrCfg::TR);
    return std::max(8, std::min(256, num_mt) / 8 * 8);
}
bool gemm_wsr_lnb_ok(const GemmWS& g) {
    return !(g.opts & (OPT_TILE_GEMM | OPT_NO_ROLE_SPLIT_LN | OPT_UNFUSED_LN_BWD)) && g.K == 256 && g.N == 256 && g.M >= 8192 && g.epi == EPI_NONE && g.bias == nullptr &&
           g.drop.p == 0.f && g.res != nullptr && g.lnb_y != nullptr && g.lnb_stats != nullptr && g.lnb_gamma != nullptr && g.lnb_beta != nullptr &&
           g.a_rows == nullptr && g.m_dev == nullptr && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.ldc % 8 == 0 && g.ldr == g.ldc && g.lnb_ldy == g.ldc &&
           (g.lnb_dx_drop == nullptr || g.lnb_lddx == g.ldc) && (int64_t)g.M * g.ldc * 2 < (int64_t)1 << 32 &&
           ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0 && ((uintptr_t)g.C % 16) == 0 && ((uintptr_t)g.res % 16) == 0 &&
           ((uintptr_t)g.lnb_y % 16) == 0 && ((uintptr_t)g.lnb_dx_drop % 16) == 0 && ((uintptr_t)g.lnb_stats % 8) == 0;
}
int gemm_wsr_lnb(const GemmWS& g, hipStream_t st) {
    PMGT_CHECK(gemm_wsr_lnb_ok(g) && g.lnb_part != nullptr, -2, "gemm_wsr_lnb: unsupported shape / epilogue M=%d N=%d K=%d", g.M, g.N, g.K);
    PMGT_SMEM_ATTR((const void*)gemm_wsr_kernel<true>, WsrCfg::SMEM);
    note_launch(LT_GEMM_WSR_LNB);
    hipLaunchKernelGGL(gemm_wsr_kernel<true>, dim3(gemm_wsr_lnb_parts(g.M)), dim3(1024), WsrCfg::SMEM, st, g);
    PMGT_LAUNCH_OK();
    return 0;
}


// ================================================================================================
// Role-split weight-stationary GEMM for K = 512 (the d = 512 shapes: Q|K|V|C projection, attention output, FFN1 / FFN2 and
// their data gradients), epilogues: bias | bias + GELU (pre-activation kept) | GELU' of a saved pre-activation | bias + dropout +
// residual.  gemm_ws_kernel<16, MODE> keeps a 256-column slab of W in 128 VGPRs per wave and alternates, in lockstep between
// barriers, an MFMA phase (~4 800 cycles per 64-row tile) and an epilogue phase (~4 000) with the LDS-DMA of the next tile exposed
// in between (in-kernel stamps: 10 000 cycles per tile, matrix pipe 37 - 41 % busy).  Here TWELVE waves per CU split the roles:
//   waves 0-7  (GEMM role):     W columns 32 g .. 32 g + 31 of the slab in 128 VGPRs; the 32-row A tile of step t + 1 travels by LDS-DMA
//                               (one 1-KB row per instruction, four per wave) while step t runs its 64 MFMAs per wave; the fp32 result
//                               goes to staging buffer t & 1;
//   waves 8-11 (epilogue role): the epilogue of step t - 1 from staging buffer (t - 1) & 1 (32 lanes per row, four passes of 8
//                               rows), its operand (residual / pre-activation) prefetched a step ahead.
// Three waves per SIMD leave 168 VGPRs each: the GEMM role fits with 32-row tiles (16 accumulator registers), which is also what
// lets two A tiles and two staging buffers share the LDS (64 + 66.5 KB).  One s_barrier per step.  Same arithmetic per element in
// the same order as gemm_ws_kernel<16, MODE> (bit-identical results).
// ================================================================================================
enum { W5_PLAIN = 0, W5_GELU = 1, W5_GELU_GRAD = 2, W5_RES = 3 };
#ifdef PMGT_W5_PROF
// cycles per step of workgroup 40: [wave][interval] (GEMM role: DMA issue | fragments + MFMA | staging + waits | barrier; epilogue role:
// work | barrier), slot 7 = steps
__device__ unsigned long long g_w5_prof[12][8];
#define W5_STAMP(k_) do { const unsigned long long n_ = __builtin_readcyclecounter(); pacc[k_] += n_ - plast; plast = n_; } while (0)
#else
#define W5_STAMP(k_) do { } while (0)
#endif

struct Wsr5Cfg {
    static constexpr int TR = 32, ROWB = 1024, TILEB = TR * ROWB, NR = 2, ES = 256 + 4, STG = TR * ES * 4;
    static constexpr int SMEM = NR * TILEB + 2 * STG;
};

template <int MODE>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_wsr512_kernel(GemmWS g) {
    using C = Wsr5Cfg;
    if (g.m_dev) {      // device-side row count (see gemm_ws_body)
        g.M = min(g.M, *g.m_dev);
        if (g.M <= 0) return;
    }
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
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_wsr_t*)smem;
    const uint32_t stg0 = lds0 + C::NR * C::TILEB;
#ifdef PMGT_W5_PROF
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long plast = 0;
#endif

    // Who moves the A tiles.  The plain epilogue leaves its four waves idle for 40 % of a step (in-kernel stamps: 2 050 cycles of
    // work against 3 550 on the GEMM role's critical path, 400 - 650 of which are the issue of its four DMA pieces): there the EPILOGUE
    // role issues the LDS-DMA of step t + 1 (eight rows per wave) and waits for it before the barrier.  The other epilogues are the
    // longer role themselves (4 000 - 4 900 cycles: dropout / GELU arithmetic on four waves) and leave the DMA to the GEMM role.
    constexpr bool EPI_DMA = MODE == W5_PLAIN;
    constexpr int DR = EPI_DMA ? 8 : 4;                     // rows per DMA-issuing wave
    const int dw = EPI_DMA ? wave - 8 : wave;              // its index among them
    // LDS-DMA of one A tile: wave dw moves rows DR dw .. DR dw + DR - 1, one 1-KB row per instruction; LDS chunk slot `lane` of a row
    // takes source chunk lane ^ (row & 15)
    // (the per-lane part of the source addresses is ONE register, lane ^ (DR dw & 15), XORed with p at the point of use: the
    // compiler otherwise keeps 64-bit addresses alive across the loop, spills them next to the 128 registers of W, and every
    // reload waits with vmcnt(0) -- for the previous DMA)
    const uint32_t l0 = (uint32_t)(lane ^ ((DR * dw) & 15));
    auto dma = [&](int t, int slot) __attribute__((always_inline)) {
        const int mt = x + t * gx;
        uint32_t lv = l0;
        asm volatile("" : "+v"(lv));
#pragma unroll
        for (int p = 0; p < DR; ++p) {
            const int row = DR * dw + p;
            const int m = min(mt * C::TR + row, g.M - 1);
            const char* rowp = (const char*)g.A + (int64_t)m * g.lda * 2;      // (wave-uniform)
            const char* src = rowp + ((lv ^ (uint32_t)p) << 4);
            __builtin_amdgcn_global_load_lds((gbl_void_wsr_t*)src, (lds_void_wsr_t*)(smem + slot * C::TILEB + row * C::ROWB), 16, 0, 0);
        }
    };

    if (wave < 8) {
        // ================================================================ GEMM role
        const int gw = wave, r = lane & 15, q = lane >> 4;
        bf16x8 wf[2][16];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                wf[j][ks] = *(const bf16x8*)((const bf16*)g.B + (int64_t)(nb + 32 * gw + 16 * j + r) * g.ldb + 32 * ks + 8 * q);
        // fragment address of k-step 0 in row tile 0: row r, chunk slot q ^ r; k-step ks = XOR (ks << 6); row tile 1 = + 16 rows
        const uint32_t fr0 = (uint32_t)(r * C::ROWB + ((q ^ r) << 4));
        // staging address of acc[0][0][0]: row 4 q, column 32 gw + r
        const uint32_t sw0 = (uint32_t)((4 * q * C::ES + 32 * gw + r) * 4);
        if constexpr (!EPI_DMA) { if (0 < n) dma(0, 0); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // W fragments and tile 0
        __builtin_amdgcn_s_barrier();
#ifdef PMGT_W5_PROF
        plast = __builtin_readcyclecounter();
#endif
        for (int it = 0; it <= n; ++it) {
            if (it < n) {
                if constexpr (!EPI_DMA) { if (it + 1 < n) dma(it + 1, (it + 1) & 1); }      // that slot held tile it - 1: every wave's reads of it drained before the last barrier
                W5_STAMP(0);
                const uint32_t ab = lds0 + (uint32_t)((it & 1) * C::TILEB);
                f32x4 acc[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                u32x4 fa[2][2];      // [k-step parity][row tile]: the reads of k-step ks + 1 travel under the MFMAs of k-step ks
                auto rd = [&](int ks) __attribute__((always_inline)) {
                    uint32_t f0 = fr0;
                    asm volatile("" : "+v"(f0));                               // recomputed per k-step, not sixteen addresses kept in registers
                    const uint32_t ad = (f0 ^ (uint32_t)(ks << 6)) + ab;       // XOR inside the row, then the tile base
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16384" : "=&v"(fa[ks & 1][0]), "=&v"(fa[ks & 1][1]) : "v"(ad) : "memory");
                };
                rd(0);
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    if (ks + 1 < 16) {
                        rd(ks + 1);
                        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[ks & 1][0]), "+v"(fa[ks & 1][1]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[ks & 1][0]), "+v"(fa[ks & 1][1]));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[ks & 1][i]), wf[j][ks], acc[i][j], 0, 0, 0);
                }
                // (inline-asm staging writes: the hazard recogniser does not put the wait states between an MFMA and an LDS instruction
                // that reads its result there; the s_nop is tied to the four accumulator tiles)
                asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]) :: "memory");
                W5_STAMP(1);
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
                // staging tile in LDS and the A tile of step it + 1 landed (for this wave) before the barrier
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                W5_STAMP(2);
            }
            __builtin_amdgcn_s_barrier();
            W5_STAMP(3);
        }
#ifdef PMGT_W5_PROF
        if (blockIdx.x == 40 && lane == 0) { for (int k_ = 0; k_ < 4; ++k_) g_w5_prof[wave][k_] = pacc[k_]; g_w5_prof[wave][7] = (unsigned long long)n; }
#endif
        return;
    }
    // ==================================================================== epilogue role: 256 threads, 32 lanes per row, 8 rows per pass
    const int te = tid - 512;
    const int erow = te >> 5, ecol = (te & 31) * 8;
    const DropKey dk = make_drop_key(g.drop);
    bf16* Cp = (bf16*)g.C;
    bf16* AUX = (bf16*)g.aux;
    const bf16* PF = (MODE == W5_GELU_GRAD) ? (const bf16*)g.aux : (const bf16*)g.res;      // prefetched epilogue operand
    const int64_t ldpf = (MODE == W5_GELU_GRAD) ? g.ldaux : g.ldr;
    constexpr bool HAS_PF = MODE == W5_GELU_GRAD || MODE == W5_RES;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = g.bias ? g.bias[nb + ecol + e] : 0.f;
    bf16x8 pf[4];
    auto load_pf = [&](int t) __attribute__((always_inline)) {
        const int mt = x + t * gx;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int m = min(mt * C::TR + erow + 8 * ps, g.M - 1);
            pf[ps] = *(const bf16x8*)(PF + (int64_t)m * ldpf + nb + ecol);
        }
    };
    if constexpr (HAS_PF) { if (0 < n) load_pf(0); }
    if constexpr (EPI_DMA) {
        if (0 < n) dma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#ifdef PMGT_W5_PROF
    plast = __builtin_readcyclecounter();
#endif
    for (int it = 0; it <= n; ++it) {
        if constexpr (EPI_DMA) { if (it + 1 < n) dma(it + 1, (it + 1) & 1); }      // first in the step: the stores below are younger
        bool full_tile = false;
        if (it >= 1) {
            const int tt = it - 1, mt = x + tt * gx;
            full_tile = mt * C::TR + C::TR <= g.M;
            const float* stage = (const float*)(smem + C::NR * C::TILEB + (tt & 1) * C::STG);
            bf16x8 rv[4];
            if constexpr (HAS_PF) {
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) rv[ps] = pf[ps];
                if (tt + 1 < n) load_pf(tt + 1);      // the next step's operand rows travel under this step's arithmetic
            }
            // the four passes' staging rows are read up front: ONE LDS round trip per step for a role that has a single wave per SIMD
            // (nothing else of its own to issue while a read is in flight)
            f32x4 sv[4][2];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                sv[ps][0] = *(const f32x4*)(stage + (erow + 8 * ps) * C::ES + ecol);
                sv[ps][1] = *(const f32x4*)(stage + (erow + 8 * ps) * C::ES + ecol + 4);
            }
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = erow + 8 * ps;
                const int m = mt * C::TR + row;
                const int ncol = nb + ecol;
                if (m < g.M) {
                    float v[8];
                    const f32x4 s0 = sv[ps][0], s1 = sv[ps][1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = s0[e]; v[4 + e] = s1[e]; }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bias[e];
                    if constexpr (MODE == W5_GELU) {
                        bf16x8 pre;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { pre[e] = (bf16)v[e]; v[e] = gelu_fast((float)pre[e]); }
                        *(bf16x8*)(AUX + (int64_t)m * g.ldaux + ncol) = pre;
                    } else if constexpr (MODE == W5_GELU_GRAD) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= gelu_fast_grad((float)rv[ps][e]);
                    }
                    if (MODE == W5_RES && dk.on) {
                        float d0[4], d1[4];
                        drop_mul4(dk, (uint32_t)m, (uint32_t)ncol >> 2, d0);
                        drop_mul4(dk, (uint32_t)m, ((uint32_t)ncol >> 2) + 1, d1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] *= d0[e]; v[4 + e] *= d1[e]; }
                    }
                    if constexpr (MODE == W5_RES) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rv[ps][e];
                    }
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                    *(bf16x8*)(Cp + (int64_t)m * g.ldc + ncol) = o;
                }
            }
        }
        if constexpr (EPI_DMA) {
            // the tile of step it + 1 has landed for this wave: the counter retires in order, and the four row stores of a FULL tile
            // (every pass stores, for every lane) were issued after the DMA pieces -- they may stay in flight
            if (full_tile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        W5_STAMP(0);
        __builtin_amdgcn_s_barrier();
        W5_STAMP(1);
    }
#ifdef PMGT_W5_PROF
    if (blockIdx.x == 40 && lane == 0) { for (int k_ = 0; k_ < 2; ++k_) g_w5_prof[wave][k_] = pacc[k_]; g_w5_prof[wave][7] = (unsigned long long)n; }
#endif
}

static int wsr5_mode(const GemmWS& g) {
    const bool drop = g.drop.p > 0.f;
    if (g.epi == EPI_NONE && !g.res && !drop) return W5_PLAIN;
    if (g.epi == EPI_GELU && !g.res && !drop) return W5_GELU;
    if (g.epi == EPI_GELU_GRAD && !g.res && !drop) return W5_GELU_GRAD;
    if (g.epi == EPI_NONE && g.res) return W5_RES;
    return -1;
}

bool gemm_wsr512_ok(const GemmWS& g) {
    const int mode = wsr5_mode(g);
    return mode >= 0 && !(g.opts & (OPT_TILE_GEMM | OPT_NO_ROLE_SPLIT_LN)) && g.K == 512 && g.N % 256 == 0 && g.N >= 256 && g.M >= 8192 && g.a_rows == nullptr &&
           // (q8: the e4m3 copy of the LayerNorm OUTPUT -- at N = 512 written by the LayerNorm launch behind this GEMM, not by its epilogue)
           g.lda % 8 == 0 && g.ldb % 8 == 0 && g.ldc % 8 == 0 && (g.res == nullptr || g.ldr % 8 == 0) && (g.aux == nullptr || g.ldaux % 8 == 0) &&
           ((mode != W5_GELU && mode != W5_GELU_GRAD) || g.aux != nullptr) &&
           ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0 && ((uintptr_t)g.C % 16) == 0 && ((uintptr_t)g.res % 16) == 0 && ((uintptr_t)g.aux % 16) == 0;
}

template <int MODE> static int launch_wsr512(const GemmWS& g, hipStream_t st) {
    auto kern = gemm_wsr512_kernel<MODE>;
    PMGT_SMEM_ATTR((const void*)kern, Wsr5Cfg::SMEM);
    const int ny = g.N / 256, num_mt = cdiv(g.M, Wsr5Cfg::TR);
    const int gx = std::max(8, std::min(256 / ny, num_mt) / 8 * 8);      // multiple of 8 row slots, one 12-wave workgroup per CU
    note_launch(LT_GEMM_WSR512);
    hipLaunchKernelGGL(kern, dim3(gx * ny), dim3(768), Wsr5Cfg::SMEM, st, g);
    PMGT_LAUNCH_OK();
    return 0;
}

int gemm_wsr512(const GemmWS& g, hipStream_t st) {
    PMGT_CHECK(gemm_wsr512_ok(g), -2, "gemm_wsr512: unsupported shape / epilogue M=%d N=%d K=%d", g.M, g.N, g.K);
    switch (wsr5_mode(g)) {
        case W5_PLAIN: return launch_wsr512<W5_PLAIN>(g, st);
        case W5_GELU: return launch_wsr512<W5_GELU>(g, st);
        case W5_GELU_GRAD: return launch_wsr512<W5_GELU_GRAD>(g, st);
        default: return launch_wsr512<W5_RES>(g, st);
    }
}

}  // namespace pmgt

#ifdef PMGT_W5_PROF
extern "C" int pmgt_debug_w5_prof_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmgt::g_w5_prof), sizeof(pmgt::g_w5_prof));
}
extern "C" int pmgt_debug_w2_prof_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pmgt::g_w2_prof), sizeof(pmgt::g_w2_prof));
}
#endif
