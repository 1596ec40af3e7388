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
#include "gemm.h"

namespace pmgt {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

template <int KS> struct QaCfg {
    static constexpr int K = 32 * KS;
    static constexpr int ROWB = K * 2;
    static constexpr int CPR = K / 8;
    static constexpr int TILEB = 64 * ROWB;
    static constexpr int LPT = 64 * CPR / 512;
    static_assert(LPT * 512 == 64 * CPR, "tile must be a whole number of chunks per thread");
    static constexpr int SWZ = CPR >= 16 ? 15 : CPR - 1;
    static constexpr int QT_ROWB = 512;                       // projection tile: 64 rows x 256 bf16
    static constexpr int QTB = 64 * QT_ROWB;
    static constexpr int SMEM = 2 * TILEB + QTB + 8 * 64 * 4; // A ring + projection tile + per-wave {rho[32], madd[32]}
};

// reduction over the 4 lanes l, l^16, l^32, l^48 in the VALU (v_permlane16/32_swap, see attention_mfma.hip)
__device__ __forceinline__ float qred(float v, bool is_max) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = is_max ? fmaxf(a, b) : a + b;
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return is_max ? fmaxf(a, b) : a + b;
}

// byte address of 16-byte chunk `ch` (0..31) of row `row` inside the swizzled projection tile
__device__ __forceinline__ int qt_addr(int row, int ch) { return row * 512 + ((ch ^ (row & 15)) << 4); }

template <int KS>
__global__ __launch_bounds__(512) void qkvc_attn_fwd_kernel(QkvcAttn a) {
    using C = QaCfg<KS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* qt = smem + 2 * C::TILEB;
    float* wl = (float*)(qt + C::QTB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int d = a.H * 32, M = a.Tseq * 32;

    const int ny = a.H / 2;
    const int b = blockIdx.x;
    const int y = (b >> 3) % ny, x = (b & 7) + 8 * (b / (8 * ny));
    const int gx = gridDim.x / ny;
    const int num_mt = (M + 63) / 64;
    // local column c of the slab -> column of q | k | v | c
    auto gcol = [&](int c) { return (c >> 6) * d + (2 * y + ((c >> 5) & 1)) * 32 + (c & 31); };
    // ... and the column it is STORED at: the same, or head-major (head, matrix, w): 256 contiguous bytes per (row, head)
    auto ocol = [&](int c) { return a.hm ? ((2 * y + ((c >> 5) & 1)) * 4 + (c >> 6)) * 32 + (c & 31) : gcol(c); };

    // ---- resident W fragments (A operand now): rows n = gcol(32 wave + 16 j + r), k = 32 ks + 8 q
    bf16x8 wf[2][KS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = gcol(32 * wave + 16 * j + r);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[j][ks] = *(const bf16x8*)((const bf16*)a.W + (int64_t)n * a.ldw + 32 * ks + 8 * q);
    }
    // bias of the 4 consecutive output columns this lane owns in each of its two 16-column blocks
    f32x4 bj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = gcol(32 * wave + 16 * j + 4 * q);
        bj[j] = a.bias ? *(const f32x4*)(a.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    u32x4 ra[2][C::LPT];
    auto gload = [&](int mt, int set) {
#pragma unroll
        for (int i = 0; i < C::LPT; ++i) {
            const int idx = tid + 512 * i;
            const int row = idx / C::CPR, ch = idx % C::CPR;
            const int m = min(mt * 64 + row, M - 1);
            ra[set][i] = *(const u32x4*)((const char*)a.X + ((int64_t)m * a.ldx) * 2 + ch * 16);
        }
    };
    auto sstore = [&](int buf, int set) {
#pragma unroll
        for (int i = 0; i < C::LPT; ++i) {
            const int idx = tid + 512 * i;
            const int row = idx / C::CPR, ch = idx % C::CPR;
            *(u32x4*)(sA + buf * C::TILEB + row * C::ROWB + ((ch ^ (row & C::SWZ)) << 4)) = ra[set][i];
        }
    };

    const DropKey k1 = make_drop_key(a.drop1), k2 = make_drop_key(a.drop2);
    const float beta = a.beta, omb = 1.f - a.beta;
    const float isq = 0.17677669529663687f;     // 1 / sqrt(32)
    // attention role of this wave
    const int us = wave >> 2, uh = (wave >> 1) & 1, it = wave & 1;
    const int h = 2 * y + uh;
    float* rho = wl + wave * 64;
    float* madd = rho + 32;
    // 16-byte chunk index of (matrix mtx, this wave's head) at k-chunk q inside a projection-tile row
    auto blk = [&](int mtx) { return 4 * (2 * mtx + uh) + q; };
    bf16* QKVC = (bf16*)a.qkvc;
    bf16* CTX = (bf16*)a.ctx;
    const int erow = tid >> 5, ech = tid & 31;

    auto tile_step = [&](auto Pc, int mt) {
        constexpr int P = decltype(Pc)::value;
        sstore(P, P);
        if (mt + 2 * gx < num_mt) gload(mt + 2 * gx, P);
        __syncthreads();                 // A tile visible; every wave is past the previous tile's attention phase
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
        const char* a_base = sA + P * C::TILEB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * i + r;
                fa[i] = *(const bf16x8*)(a_base + row * C::ROWB + (((4 * ks + q) ^ (row & C::SWZ)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {      // D[n = 4 q + e][m = r]: acc[i][j][e] = out[16 i + r][16 j + 4 q + e]
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], fa[i], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], fa[i], acc[i][1], 0, 0, 0);
            }
        }
        // ---- + bias, bf16, into the projection tile (8 bytes per lane per block)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 16 * i + r;
                const f32x4 v = acc[i][j] + bj[j];
                bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                *(bf16x4*)(qt + qt_addr(row, 4 * wave + 2 * j + (q >> 1)) + 8 * (q & 1)) = o;
            }
        __syncthreads();
        // ---- (a) projection tile -> HBM, 16 bytes per lane, rows contiguous inside each 64-byte head block
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = erow + 16 * ps;
            const int m = mt * 64 + row;
            if (m < M) *(u32x4*)(QKVC + (int64_t)m * a.ldq + ocol(8 * ech)) = *(const u32x4*)(qt + qt_addr(row, ech));
        }
        // ---- (b) attention of (sequence 2 mt + us, head h, queries 16 it .. 16 it + 15)
        const int t = 2 * mt + us;
        const bool act = t < a.Tseq;
        if (it == 1 && t < a.cls_only_seqs) return;      // (wave-uniform) only query row 0 of this sequence is ever read
        const int R0 = 32 * us;
        bf16x8 fq, fk[2], fc[2];
        fq = *(const bf16x8*)(qt + qt_addr(R0 + 16 * it + r, blk(0)));
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            fk[jt] = *(const bf16x8*)(qt + qt_addr(R0 + 16 * jt + r, blk(1)));
            fc[jt] = *(const bf16x8*)(qt + qt_addr(R0 + 16 * jt + r, blk(3)));
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float c = (float)fc[jt][e]; ss = fmaf(c, c, ss); }
            ss = qred(ss, false);
            if (q == 0) rho[16 * jt + r] = rsqrtf(ss);       // 1 / |c_row|
        }
        if (lane < 32) madd[lane] = (act && a.mask) ? (1.f - a.mask[(int64_t)t * 32 + lane]) * -10000.f : 0.f;
        // rho / madd are private to this wave: LDS operations of one wave execute in order, no barrier needed
        __builtin_amdgcn_wave_barrier();
        f32x4 a1[2], a2[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            a1[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[jt], it == 0 ? fc[0] : fc[1], z, 0, 0, 0);
            a2[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[jt], fq, z, 0, 0, 0);
        }
        const int i = 16 * it + r;
        {
            const float rho_i = rho[i];
            float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 16 * jt + 4 * q + e;
                    const float v1 = 1.f - a1[jt][e] * (rho_i * rho[j]) + (i == j ? 1.f : 0.f) + madd[j];
                    const float v2 = a2[jt][e] * isq + madd[j];
                    a1[jt][e] = v1;
                    a2[jt][e] = v2;
                    m1 = fmaxf(m1, v1);
                    m2 = fmaxf(m2, v2);
                }
            m1 = qred(m1, true);
            m2 = qred(m2, true);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float e1 = __expf(a1[jt][e] - m1), e2 = __expf(a2[jt][e] - m2);
                    a1[jt][e] = e1;
                    a2[jt][e] = e2;
                    s1 += e1;
                    s2 += e2;
                }
            s1 = qred(s1, false);
            s2 = qred(s2, false);
            const float i1 = __frcp_rn(s1), i2 = __frcp_rn(s2);
            a1[0] *= i1; a1[1] *= i1;
            a2[0] *= i2; a2[1] *= i2;
        }
        // mix + dropout -> P^T, packed as the B operand (k slot e of lane (r, q): key 16 (e >> 2) + 4 q + (e & 3))
        bf16x8 pb;
        {
            const uint64_t hbase = ((uint64_t)t * a.H + h) * 32;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                float d1[4] = {1.f, 1.f, 1.f, 1.f}, d2[4] = {1.f, 1.f, 1.f, 1.f};
                if (k1.on) {
                    drop_mul4(k1, (uint32_t)(hbase + i), (uint32_t)(4 * jt + q), d1);
                    drop_mul4(k2, (uint32_t)(hbase + i), (uint32_t)(4 * jt + q), d2);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) pb[4 * jt + e] = (bf16)(beta * d1[e] * a1[jt][e] + omb * d2[e] * a2[jt][e]);
            }
        }
        // O^T[c][i] = sum_j V[j][c] P[i][j]: A operand = transposed read of the V block (rows = keys)
        f32x4 o[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int row_lo = R0 + 4 * q + (r >> 2), row_hi = row_lo + 16;
            const int cb = (2 * 2 + uh) * 64 + (16 * ct + 4 * (r & 3)) * 2;      // byte column inside the row: V block of this head
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(qt + qt_addr(row_lo, cb >> 4) + (cb & 15)));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(qt + qt_addr(row_hi, cb >> 4) + (cb & 15)));
            const bf16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            o[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb, z, 0, 0, 0);
        }
        // one 16-byte store per lane: the 64 bytes of this (row, head) leave as one run (common.h: store_row32)
        store_row32(CTX + ((int64_t)min(t, a.Tseq - 1) * 32 + i) * a.ldc + h * 32, o[0], o[1], q, act);
        // The next tile_step's first barrier orders this phase's LDS reads before the next projection-tile
        // writes (which come after that step's second barrier anyway).
    };

    int mt = x;
    if (mt < num_mt) gload(mt, 0);
    if (mt + gx < num_mt) gload(mt + gx, 1);
    while (mt < num_mt) {
        tile_step(std::integral_constant<int, 0>{}, mt);
        mt += gx;
        if (mt >= num_mt) break;
        tile_step(std::integral_constant<int, 1>{}, mt);
        mt += gx;
    }
}

bool qkvc_attn_supported(const QkvcAttn& a) {
    const int d = a.H * 32;
    return a.S == 32 && a.dh == 32 && (d == 256 || d == 128) && a.H % 2 == 0 && a.Tseq >= 2 && a.ldx % 8 == 0 && a.ldw % 8 == 0 &&
           a.ldq % 8 == 0 && a.ldc % 4 == 0 && ((uintptr_t)a.X % 16) == 0 && ((uintptr_t)a.W % 16) == 0 &&
           ((uintptr_t)a.qkvc % 16) == 0 && ((uintptr_t)a.ctx % 8) == 0 && (a.bias == nullptr || ((uintptr_t)a.bias % 16) == 0);
}

template <int KS> static int launch_qa(const QkvcAttn& a, hipStream_t st) {
    using C = QaCfg<KS>;
    auto kern = qkvc_attn_fwd_kernel<KS>;
    static bool attr_done = false;
    if (!attr_done) {
        PMGT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM));
        attr_done = true;
    }
    const int ny = a.H / 2, num_mt = cdiv(a.Tseq * 32, 64);
    const int gx = std::max(8, std::min(256 / ny, num_mt) / 8 * 8);
    hipLaunchKernelGGL(kern, dim3(gx * ny), dim3(512), C::SMEM, st, a);
    PMGT_LAUNCH_OK();
    return 0;
}

int qkvc_attn_fwd(const QkvcAttn& a, hipStream_t st) {
    if (a.Tseq <= 0) return 0;
    PMGT_CHECK(qkvc_attn_supported(a), -2, "qkvc_attn_fwd: unsupported shape S=%d dh=%d H=%d", a.S, a.dh, a.H);
    return a.H * 32 == 256 ? launch_qa<8>(a, st) : launch_qa<4>(a, st);
}

}  // namespace pmgt
