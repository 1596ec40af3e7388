// Full-row tile GEMM with the LayerNorm forward in its epilogue at N = 512 (bf16):
//     C = dropout(A W^T + bias) + res,   y = LayerNorm(C)        (BertSelfOutput / BertOutput of the reference,
//     pmgt/pmgt/modeling_pmgt.py:293-294,332 through transformers 4.11.2 modeling_bert.py BertSelfOutput / BertOutput)
// The streaming kernels of gemm_wsr.hip keep a 256-column slab of W in registers, so at N = 512 a row is split over two workgroups
// and LayerNorm ran as its own launch (1U read + 1U write per site, 12 sites per step of the d = 512 configurations).  Here one
// workgroup owns 128 WHOLE rows: tile 128 x 512, eight waves of 128 rows x 64 columns each (128 accumulator VGPRs, the wave tile of
// gemm_nt_big_kernel), A and W through a 3-stage LDS-DMA ring of 40 KB stages (k-step 32).  W is re-read from L2 per 128 rows
// (512 KB per tile against 128 KB of A from HBM): per k-step a CU takes in 40 KB for 4.2 MFLOP, about the ingest : MFMA balance of
// the 256 x 256 tiles.  Epilogue, all in registers: v_permlane16_swap gives every lane 8 consecutive columns (x 2) of one row; the
// residual rows are requested during the last two k-steps; x = bf16(dropout(acc + bias) + res) -- the rounding the two-launch form
// applies on its way through HBM, so both forms normalise the same values; row statistics as per-wave (mean, M2) pairs over 64 columns
// combined across the eight waves through 8 KB of LDS (Chan's formula: no E[x^2] - mean^2 cancellation); y = (x - mean) rstd gamma + beta.
// The pre-LayerNorm sum C is always stored: the d = 512 LayerNorm backward reads it (the x^-from-the-output form is a d = 256 path).
#include <type_traits>

#include "gemm.h"

namespace pmgt {

typedef __attribute__((address_space(3))) void rl_lds_void_t;
typedef __attribute__((address_space(1))) const void rl_gbl_void_t;

namespace {
constexpr int RL_BM = 128, RL_BN = 512, RL_ROWB = 64, RL_STAGE = (RL_BM + RL_BN) * RL_ROWB, RL_NST = 3;
constexpr int RL_EXCH = RL_NST * RL_STAGE;                 // [128 rows][8 waves] {mean, M2} of 64 columns
constexpr int RL_STAT = RL_EXCH + RL_BM * 8 * 8;           // [128 rows] {mean, rstd}
constexpr int RL_SMEM = RL_STAT + RL_BM * 8;
__device__ __forceinline__ int rl_swz(int row) { return (0 - (row >> 2)) & 3; }      // chunk XOR of the 64-byte stage rows (gemm.hip: nt_swz)
// sum over the four lanes r, r + 16, r + 32, r + 48
__device__ __forceinline__ float rl_qsum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = a + b;
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
}  // namespace

__global__ __launch_bounds__(512) void gemm_rowln512_kernel(GemmWS g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * RL_BM;
    if (g.m_dev) g.M = min(g.M, *g.m_dev);
    if (m0 >= g.M) return;

    // ---- LDS-DMA sources: one piece = 16 rows x 64 bytes; wave w brings A piece w and W pieces 4 w .. 4 w + 3 of every stage
    const char* asrc;
    const char* bsrc[4];
    {
        const int row = 16 * wave + (lane >> 2);
        const int ch = (lane & 3) ^ rl_swz(row);
        asrc = (const char*)g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda * 2 + ch * 16;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 16 * (4 * wave + j) + (lane >> 2);
        const int ch = (lane & 3) ^ rl_swz(row);
        bsrc[j] = (const char*)g.B + (int64_t)row * g.ldb * 2 + ch * 16;
    }
    auto issue = [&](int kt) __attribute__((always_inline)) {
        char* st = smem + (kt % RL_NST) * RL_STAGE;
        __builtin_amdgcn_global_load_lds((rl_gbl_void_t*)(asrc + (int64_t)kt * RL_ROWB), (rl_lds_void_t*)(st + 16 * wave * RL_ROWB), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((rl_gbl_void_t*)(bsrc[j] + (int64_t)kt * RL_ROWB),
                                             (rl_lds_void_t*)(st + RL_BM * RL_ROWB + 16 * (4 * wave + j) * RL_ROWB), 16, 0, 0);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint32_t lds_base = (uint32_t)(uintptr_t)(rl_lds_void_t*)smem;
    uint32_t offa[8], offb[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ra = i * 16 + r;
        offa[i] = (uint32_t)(ra * RL_ROWB + ((q ^ rl_swz(ra)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rb = wave * 64 + j * 16 + r;
        offb[j] = (uint32_t)(RL_BM * RL_ROWB + rb * RL_ROWB + ((q ^ rl_swz(rb)) << 4));
    }
    // post-swap ownership of a lane: row 16 i + r, columns ncol + 32 pr .. + 7 (pr = 0, 1)
    const int cb = ((q & 1) << 4) | ((q & 2) << 2);          // q = 0, 1, 2, 3 -> columns 0, 16, 8, 24 of the 32-column pair
    const int ncol = wave * 64 + cb;
    const bf16* R = (const bf16*)g.res;
    u32x4 rr[8][2];
    auto load_res = [&](int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i >= i0 && i < i1) {
                const int m = min(m0 + 16 * i + r, g.M - 1);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) rr[i][pr] = *(const u32x4*)(R + (int64_t)m * g.ldr + ncol + 32 * pr);
            }
    };

    const int nk = g.K / 32;      // >= 2 (host)
    issue(0);
    issue(1);
    // one k-step; TAIL = 0: steady state, 1: step nk - 2 (requests the first four residual row blocks behind its barrier), 2: step nk - 1.
    // The last two steps are peeled so that the residual registers are live there only (128 accumulator + 48 fragment VGPRs leave room
    // for half of the residual rows, not for all of them; the other half is requested when the fragments are dead).
    auto kstep = [&](int kt, auto TAILc) __attribute__((always_inline)) {
        constexpr int TAIL = decltype(TAILc)::value;
        // own DMAs of stage kt have landed (5 per stage and wave)
        if constexpr (TAIL == 2) {
            if (R) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // (the eight residual loads of step nk - 2 are younger than every DMA)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const uint32_t sbase = lds_base + (uint32_t)((kt % RL_NST) * RL_STAGE);
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
        // the DMA of stage kt + 2 goes into the slot stage kt - 1 left: every wave has passed this step's barrier, i.e. finished its reads of it
        if constexpr (TAIL == 0) issue(kt + 2);
        if constexpr (TAIL == 1) { if (R) load_res(0, 4); }
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
            // operands swapped (D = W_frag x A_frag^T): acc[i][j][e] = out[16 i + r][64 wave + 16 j + 4 q + e]
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, t[8 + j]), __builtin_bit_cast(bf16x8, t[i]),
                                                                    acc[i][j], 0, 0, 0);
        }
    };
    for (int kt = 0; kt < nk - 2; ++kt) kstep(kt, std::integral_constant<int, 0>{});
    kstep(nk - 2, std::integral_constant<int, 1>{});
    kstep(nk - 1, std::integral_constant<int, 2>{});
    if (R) load_res(4, 8);
#ifdef PMGT_RL_MAIN_ONLY      // (ablation build: the main loop and one store per lane that keeps the accumulators alive)
    {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (R) sacc += (float)rr[0][0][0] + (float)rr[7][1][3];
        ((float*)g.ln_out)[(int64_t)blockIdx.x * 512 + tid] = sacc;
        return;
    }
#endif

    // ---- epilogue phase A: x = bf16(dropout(acc + bias) + res), kept as fp32 in the accumulator registers in the post-swap layout
    const DropKey dk = make_drop_key(g.drop);
    bf16* Cp = (bf16*)g.C;
    float bias[2][8];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 bv = g.bias ? *(const f32x4*)(g.bias + ncol + 32 * pr + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[pr][4 * h + e] = bv[e];
        }
    // (no explicit vmcnt wait here: every DMA has landed -- the last k-step waited for them -- and the compiler counts the residual /
    //  parameter loads below itself, so rows 0 .. 3 are worked on while rows 4 .. 7 are still in flight)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + 16 * i + r;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            f32x4 a = acc[i][2 * pr], b = acc[i][2 * pr + 1];
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                         "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a[e] + bias[pr][e]; v[4 + e] = b[e] + bias[pr][4 + e]; }
            if (dk.on) {
                float d0[4], d1[4];
                const uint32_t n = (uint32_t)(ncol + 32 * pr);
                drop_mul4(dk, (uint32_t)m, n >> 2, d0);
                drop_mul4(dk, (uint32_t)m, (n >> 2) + 1, d1);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] *= d0[e]; v[4 + e] *= d1[e]; }
            }
            if (R) {
                const bf16x8 rv8 = __builtin_bit_cast(bf16x8, rr[i][pr]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv8[e];
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = (bf16)v[e]; v[e] = (float)o[e]; }
#ifndef PMGT_RL_NO_C
            if (m < g.M) *(bf16x8*)(Cp + (int64_t)m * g.ldc + ncol + 32 * pr) = o;
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = v[e]; b[e] = v[4 + e]; }
            acc[i][2 * pr] = a;
            acc[i][2 * pr + 1] = b;
        }
    }
    // (gamma / beta: requested here, when the residual registers are dead; they land under phase B and its barriers)
    float gam[2][8], bet[2][8];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 gv = *(const f32x4*)(g.ln_gamma + ncol + 32 * pr + 4 * h), bv = *(const f32x4*)(g.ln_beta + ncol + 32 * pr + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gam[pr][4 * h + e] = gv[e]; bet[pr][4 * h + e] = bv[e]; }
        }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B: per-wave {mean, M2} of each row over this wave's 64 columns -> LDS
    float2* exch = (float2*)(smem + RL_EXCH);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
        const float mw = rl_qsum(s) * (1.f / 64.f);
        float m2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dlt = acc[i][j][e] - mw; m2 = fmaf(dlt, dlt, m2); }
        m2 = rl_qsum(m2);
        if (q == 0) exch[(16 * i + r) * 8 + wave] = make_float2(mw, m2);
    }
    __syncthreads();
    // wave w combines rows 16 w .. 16 w + 15: lane (r, q) takes the pairs of waves 2 q, 2 q + 1
    float2* stat = (float2*)(smem + RL_STAT);
    {
        const int row = 16 * wave + r;
        const float2 p0 = exch[row * 8 + 2 * q], p1 = exch[row * 8 + 2 * q + 1];
        const float mean = rl_qsum(p0.x + p1.x) * (1.f / 8.f);
        const float d0 = p0.x - mean, d1 = p1.x - mean;
        const float m2 = rl_qsum((p0.y + p1.y) + 64.f * (d0 * d0 + d1 * d1));
        const float rstd = 1.f / sqrtf(m2 * (1.f / (float)RL_BN) + g.ln_eps);
        if (q == 0) {
            stat[row] = make_float2(mean, rstd);
            const int m = m0 + row;
            if (m < g.M) { g.ln_stats[2 * (int64_t)m] = mean; g.ln_stats[2 * (int64_t)m + 1] = rstd; }
        }
    }
    __syncthreads();
    // ---- phase C: y = (x - mean) rstd gamma + beta
    bf16* Y = (bf16*)g.ln_out;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + 16 * i + r;
        const float2 ms = stat[16 * i + r];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = acc[i][2 * pr + (e >> 2)][e & 3];
                o[e] = (bf16)((x - ms.x) * ms.y * gam[pr][e] + bet[pr][e]);
            }
            if (m < g.M) *(bf16x8*)(Y + (int64_t)m * g.N + ncol + 32 * pr) = o;
        }
    }
}

bool gemm_rowln_ok(const GemmWS& g) {
    return !(g.opts & (OPT_TILE_GEMM | OPT_UNFUSED_LN)) && g.N == RL_BN && g.K % 32 == 0 && g.K >= 64 && g.M >= 4096 && g.a_rows == nullptr &&
           !g.res_gather && g.epi == EPI_NONE && g.ln_out != nullptr && g.ln_stats != nullptr && g.ln_gamma != nullptr && g.ln_beta != nullptr &&
           g.q8 == nullptr && !g.skip_c && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.C != nullptr && g.ldc % 8 == 0 && ((uintptr_t)g.C % 16) == 0 &&
           (g.res == nullptr || (g.ldr % 8 == 0 && ((uintptr_t)g.res % 16) == 0)) && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0 &&
           ((uintptr_t)g.ln_out % 16) == 0 && (g.bias == nullptr || ((uintptr_t)g.bias % 16) == 0) && ((uintptr_t)g.ln_gamma % 16) == 0 &&
           ((uintptr_t)g.ln_beta % 16) == 0 && ((uintptr_t)g.ln_stats % 8) == 0;
}

int gemm_rowln(const GemmWS& g, hipStream_t st) {
    if (g.M <= 0) return 0;
    PMGT_CHECK(gemm_rowln_ok(g), -2, "gemm_rowln: unsupported shape / epilogue M=%d N=%d K=%d", g.M, g.N, g.K);
    PMGT_SMEM_ATTR((const void*)gemm_rowln512_kernel, RL_SMEM);
    note_launch(LT_GEMM_ROWLN);
    hipLaunchKernelGGL(gemm_rowln512_kernel, dim3(cdiv(g.M, RL_BM)), dim3(512), RL_SMEM, st, g);
    PMGT_LAUNCH_OK();
    return 0;
}

}  // namespace pmgt
