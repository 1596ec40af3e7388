// PMGT dual-softmax ("diversity promoting") self-attention, forward and backward
// (pmgt/pmgt/modeling_pmgt.py:420-534 of the reference):
//   A1 = softmax(1 - C C^T / (|C_i||C_j|) + I + mask),  A2 = softmax(Q K^T / sqrt(dh) + mask)
//   O  = (beta * drop(A1) + (1 - beta) * drop(A2)) V
//
// Round-1 kernel: fp32 VALU math for both storage modes.  A lane group of GS lanes (16/32/64,
// the smallest >= S) owns one (sequence, head); lane i owns query row i (q_i, c_i, o_i in registers),
// K/V/C tiles sit in LDS and are read as wave-wide broadcasts, the two SxS score matrices live in
// LDS as [key][query] with an odd stride so both the row phase (lane = query) and the column phase
// of the backward (lane = key) are bank-conflict free.  Nothing SxS ever reaches HBM unless the
// caller asks for the mixed probabilities (output_attentions).  Dropout masks are regenerated from
// the counter RNG in backward.  Padded keys get the additive -10000 mask of transformers 4.11.2.
#include "attention.h"

namespace pmgt {

template <typename T> struct ChunkT;
template <> struct ChunkT<float> {
    static constexpr int CH = 4;
    static __device__ __forceinline__ void ld(const void* p, float* f) {
        f32x4 v = *(const f32x4*)p;
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    }
    static __device__ __forceinline__ void st(void* p, const float* f) { *(f32x4*)p = (f32x4){f[0], f[1], f[2], f[3]}; }
};
template <> struct ChunkT<bf16> {
    static constexpr int CH = 8;
    static __device__ __forceinline__ void ld(const void* p, float* f) {
        bf16x8 v = *(const bf16x8*)p;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
    }
    static __device__ __forceinline__ void st(void* p, const float* f) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)f[e];
        *(bf16x8*)p = v;
    }
};

template <typename T, int DH> __device__ __forceinline__ void load_row(const T* p, float* f) {
    constexpr int CH = ChunkT<T>::CH;
#pragma unroll
    for (int c = 0; c < DH / CH; ++c) ChunkT<T>::ld(p + c * CH, f + c * CH);
}
template <typename T, int DH> __device__ __forceinline__ void store_row(T* p, const float* f) {
    constexpr int CH = ChunkT<T>::CH;
#pragma unroll
    for (int c = 0; c < DH / CH; ++c) ChunkT<T>::st(p + c * CH, f + c * CH);
}
// dot(reg[DH], LDS row) with the LDS row read as 16-byte broadcasts
template <typename T, int DH> __device__ __forceinline__ float dot_row(const float* reg, const char* row) {
    constexpr int CH = ChunkT<T>::CH;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < DH / CH; ++c) {
        float f[CH];
        ChunkT<T>::ld(row + c * 16, f);
#pragma unroll
        for (int e = 0; e < CH; ++e) acc = fmaf(reg[c * CH + e], f[e], acc);
    }
    return acc;
}
// reg[DH] += s * LDS row
template <typename T, int DH> __device__ __forceinline__ void axpy_row(float* reg, float s, const char* row) {
    constexpr int CH = ChunkT<T>::CH;
#pragma unroll
    for (int c = 0; c < DH / CH; ++c) {
        float f[CH];
        ChunkT<T>::ld(row + c * 16, f);
#pragma unroll
        for (int e = 0; e < CH; ++e) reg[c * CH + e] = fmaf(s, f[e], reg[c * CH + e]);
    }
}

// stride (floats) of the [key][query] score matrices: odd, >= the number of queries.  Groups of 128 lanes serve
// 64 < S <= 100 (max_position_embeddings of the reference, pmgt/pmgt/configuration_pmgt.py:23) with stride 101, which
// keeps S = 100 / head size 32 / fp32 inside the 160 KiB of a CU (stride 129 would not).
template <int GS> struct ScoreStride { static constexpr int V = GS == 128 ? 101 : GS + 1; };
static inline int score_stride(int gs) { return gs == 128 ? 101 : gs + 1; }

static inline size_t attn_group_bytes(int S, int dh, int gs, int esize, bool bwd) {
    size_t tiles = (size_t)(bwd ? 5 : 3) * S * dh * esize;
    size_t sc = (size_t)2 * S * score_stride(gs) * 4;
    size_t misc = (size_t)2 * gs * 4;
    return (tiles + sc + misc + 15) / 16 * 16;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <typename T, int DH, int GS>
__global__ void attn_fwd_kernel(AttnArgs a, int group_bytes) {
    constexpr int CH = ChunkT<T>::CH, RS = DH * (int)sizeof(T), NCHK = DH / CH, ST = ScoreStride<GS>::V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ng = blockDim.x / GS;
    const int grp = threadIdx.x / GS, li = threadIdx.x % GS;
    const int gidx = blockIdx.x * ng + grp;
    const int S = a.S, H = a.H, d = H * DH;
    const bool gact = gidx < a.Tseq * H;
    const int t = gact ? gidx / H : 0, h = gact ? gidx % H : 0;
    char* base = smem + (size_t)grp * group_bytes;
    char* sK = base;
    char* sV = sK + S * RS;
    char* sC = sV + S * RS;
    float* sc1 = (float*)(sC + S * RS);
    float* sc2 = sc1 + S * ST;
    float* rho = sc2 + S * ST;
    float* madd = rho + GS;
    const T* X = (const T*)a.qkvc + (int64_t)t * S * 4 * d + h * DH;

    if (gact) {
        for (int idx = li; idx < S * NCHK; idx += GS) {
            const int s = idx / NCHK, c = idx % NCHK;
            const T* row = X + (int64_t)s * 4 * d + c * CH;
            *(u32x4*)(sK + s * RS + c * 16) = *(const u32x4*)(row + d);
            *(u32x4*)(sV + s * RS + c * 16) = *(const u32x4*)(row + 2 * d);
            *(u32x4*)(sC + s * RS + c * 16) = *(const u32x4*)(row + 3 * d);
        }
    }
    const bool ract = gact && li < S;
    float q[DH], cc[DH];
    if (ract) {
        load_row<T, DH>(X + (int64_t)li * 4 * d, q);
        load_row<T, DH>(X + (int64_t)li * 4 * d + 3 * d, cc);
    } else {
#pragma unroll
        for (int e = 0; e < DH; ++e) { q[e] = 0.f; cc[e] = 0.f; }
    }
    float r2 = 0.f;
#pragma unroll
    for (int e = 0; e < DH; ++e) r2 = fmaf(cc[e], cc[e], r2);
    const float rho_i = sqrtf(r2);
    rho[li] = rho_i;
    madd[li] = (ract && a.mask) ? (1.f - a.mask[(int64_t)t * S + li]) * -10000.f : 0.f;
    __syncthreads();

    const float sq = sqrtf((float)DH);
    float mx1 = -INFINITY, mx2 = -INFINITY;
    // beta in {0, 1}: one branch is multiplied by exactly 0 (pmgt/pmgt/modeling_pmgt.py:519-521) -- its scores, softmax and dropout draws are
    // skipped (uniform branches); `no_beta_skip` (OPT_NO_BETA_SKIP) keeps the general arithmetic, which gives the same zeros the long way
    const bool dead1 = a.beta == 0.f && !(a.opts & OPT_NO_BETA_SKIP), dead2 = a.beta == 1.f && !(a.opts & OPT_NO_BETA_SKIP);
    if (ract) {
        for (int j = 0; j < S; ++j) {
            if (!dead1) {
                const float d1 = dot_row<T, DH>(cc, sC + j * RS);
                const float s1 = 1.f - d1 / (rho_i * rho[j]) + (j == li ? 1.f : 0.f) + madd[j];
                sc1[j * ST + li] = s1;
                mx1 = fmaxf(mx1, s1);
            }
            if (!dead2) {
                const float d2 = dot_row<T, DH>(q, sK + j * RS);
                const float s2 = d2 / sq + madd[j];
                sc2[j * ST + li] = s2;
                mx2 = fmaxf(mx2, s2);
            }
        }
        float sum1 = 0.f, sum2 = 0.f;
        for (int j = 0; j < S; ++j) {
            if (!dead1) {
                const float e1 = expf(sc1[j * ST + li] - mx1);
                sc1[j * ST + li] = e1;
                sum1 += e1;
            }
            if (!dead2) {
                const float e2 = expf(sc2[j * ST + li] - mx2);
                sc2[j * ST + li] = e2;
                sum2 += e2;
            }
        }
        const DropKey k1 = make_drop_key(a.drop1), k2 = make_drop_key(a.drop2);
        const float w1 = dead1 ? 0.f : a.beta / sum1, w2 = dead2 ? 0.f : (1.f - a.beta) / sum2;
        float o[DH];
#pragma unroll
        for (int e = 0; e < DH; ++e) o[e] = 0.f;
        const uint64_t pbase = (((uint64_t)t * H + h) * S + li) * S;
        const uint32_t prow = (uint32_t)(((uint64_t)t * H + h) * S + li);
        for (int j = 0; j < S; ++j) {
            float p1 = dead1 ? 0.f : sc1[j * ST + li] * w1, p2 = dead2 ? 0.f : sc2[j * ST + li] * w2;
            if (k1.on) {
                if (!dead1) p1 *= drop_mul1(k1, prow, (uint32_t)j);
                if (!dead2) p2 *= drop_mul1(k2, prow, (uint32_t)j);
            }
            const float p = p1 + p2;
            if (a.probs) a.probs[pbase + j] = p;
            axpy_row<T, DH>(o, p, sV + j * RS);
        }
        store_row<T, DH>((T*)a.ctx + ((int64_t)t * S + li) * d + h * DH, o);
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
template <typename T, int DH, int GS>
__global__ void attn_bwd_kernel(AttnArgs a, int group_bytes) {
    constexpr int CH = ChunkT<T>::CH, RS = DH * (int)sizeof(T), NCHK = DH / CH, ST = ScoreStride<GS>::V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ng = blockDim.x / GS;
    const int grp = threadIdx.x / GS, li = threadIdx.x % GS;
    const int gidx = blockIdx.x * ng + grp;
    const int S = a.S, H = a.H, d = H * DH;
    const bool gact = gidx < a.Tseq * H;
    const int t = gact ? gidx / H : 0, h = gact ? gidx % H : 0;
    char* base = smem + (size_t)grp * group_bytes;
    char* sK = base;
    char* sV = sK + S * RS;
    char* sC = sV + S * RS;
    char* sQ = sC + S * RS;
    char* sO = sQ + S * RS;                       // dO tile
    float* A1 = (float*)(sO + S * RS);            // [key][query], stride GS + 1
    float* A2 = A1 + S * ST;
    float* rho = A2 + S * ST;
    float* madd = rho + GS;
    const T* X = (const T*)a.qkvc + (int64_t)t * S * 4 * d + h * DH;
    const T* DO = (const T*)a.dctx + (int64_t)t * S * d + h * DH;
    T* DX = (T*)a.dqkvc + (int64_t)t * S * 4 * d + h * DH;

    if (gact) {
        for (int idx = li; idx < S * NCHK; idx += GS) {
            const int s = idx / NCHK, c = idx % NCHK;
            const T* row = X + (int64_t)s * 4 * d + c * CH;
            *(u32x4*)(sQ + s * RS + c * 16) = *(const u32x4*)(row);
            *(u32x4*)(sK + s * RS + c * 16) = *(const u32x4*)(row + d);
            *(u32x4*)(sV + s * RS + c * 16) = *(const u32x4*)(row + 2 * d);
            *(u32x4*)(sC + s * RS + c * 16) = *(const u32x4*)(row + 3 * d);
            *(u32x4*)(sO + s * RS + c * 16) = *(const u32x4*)(DO + (int64_t)s * d + c * CH);
        }
    }
    const bool ract = gact && li < S;
    const float sq = sqrtf((float)DH);
    const DropKey k1 = make_drop_key(a.drop1), k2 = make_drop_key(a.drop2);
    const float beta = a.beta, omb = 1.f - a.beta;
    // beta in {0, 1}: the dead branch (see the forward) -- its probabilities are never formed, dS = 0, and dQ = dK = 0 (beta = 1) or dC = 0 (beta = 0)
    // are stored as the exact zeros autograd reports
    const bool dead1 = a.beta == 0.f && !(a.opts & OPT_NO_BETA_SKIP), dead2 = a.beta == 1.f && !(a.opts & OPT_NO_BETA_SKIP);
    float rho_i = 0.f;
    {
        float cc[DH];
        if (ract) load_row<T, DH>(X + (int64_t)li * 4 * d + 3 * d, cc);
        else {
#pragma unroll
            for (int e = 0; e < DH; ++e) cc[e] = 0.f;
        }
        float r2 = 0.f;
#pragma unroll
        for (int e = 0; e < DH; ++e) r2 = fmaf(cc[e], cc[e], r2);
        rho_i = sqrtf(r2);
        rho[li] = rho_i;
        madd[li] = (ract && a.mask) ? (1.f - a.mask[(int64_t)t * S + li]) * -10000.f : 0.f;
        __syncthreads();
        // ---- recompute the two probability matrices (normalised, before dropout)
        if (ract) {
            float q[DH];
            load_row<T, DH>(X + (int64_t)li * 4 * d, q);
            float mx1 = -INFINITY, mx2 = -INFINITY;
            for (int j = 0; j < S; ++j) {
                float s1 = 0.f, s2 = 0.f;
                if (!dead1) {
                    const float d1 = dot_row<T, DH>(cc, sC + j * RS);
                    s1 = 1.f - d1 / (rho_i * rho[j]) + (j == li ? 1.f : 0.f) + madd[j];
                    mx1 = fmaxf(mx1, s1);
                }
                if (!dead2) {
                    const float d2 = dot_row<T, DH>(q, sK + j * RS);
                    s2 = d2 / sq + madd[j];
                    mx2 = fmaxf(mx2, s2);
                }
                A1[j * ST + li] = s1;
                A2[j * ST + li] = s2;
            }
            float sum1 = 0.f, sum2 = 0.f;
            for (int j = 0; j < S; ++j) {
                const float e1 = dead1 ? 0.f : expf(A1[j * ST + li] - mx1);
                const float e2 = dead2 ? 0.f : expf(A2[j * ST + li] - mx2);
                A1[j * ST + li] = e1;
                A2[j * ST + li] = e2;
                sum1 += e1;
                sum2 += e2;
            }
            const float i1 = dead1 ? 0.f : 1.f / sum1, i2 = dead2 ? 0.f : 1.f / sum2;      // a dead branch: all-zero probabilities, hence dS = 0 below
            for (int j = 0; j < S; ++j) {
                A1[j * ST + li] *= i1;
                A2[j * ST + li] *= i2;
            }
        }
    }
    __syncthreads();
    const uint64_t hbase = ((uint64_t)t * H + h) * S;   // prob index = (hbase + i) * S + j

    // ---- column pass 0 (lane = key j): dV_j = sum_i P_ij dO_i
    if (ract) {
        const int j = li;
        float dv[DH];
#pragma unroll
        for (int e = 0; e < DH; ++e) dv[e] = 0.f;
        for (int i = 0; i < S; ++i) {
            float p1 = beta * A1[j * ST + i], p2 = omb * A2[j * ST + i];
            if (k1.on) { p1 *= drop_mul1(k1, (uint32_t)(hbase + i), (uint32_t)j); p2 *= drop_mul1(k2, (uint32_t)(hbase + i), (uint32_t)j); }
            axpy_row<T, DH>(dv, p1 + p2, sO + i * RS);
        }
        store_row<T, DH>(DX + (int64_t)j * 4 * d + 2 * d, dv);
    }
    __syncthreads();   // A1/A2 are overwritten below

    // ---- row passes (lane = query i)
    float dch[DH];     // d(c_hat_x), x = this lane; row part now, column part later
#pragma unroll
    for (int e = 0; e < DH; ++e) dch[e] = 0.f;
    if (ract) {
        const int i = li;
        float doi[DH];
        load_row<T, DH>(DO + (int64_t)i * d, doi);
        float rd1 = 0.f, rd2 = 0.f;
        for (int j = 0; j < S; ++j) {
            const float dp = dot_row<T, DH>(doi, sV + j * RS);
            float g1 = beta * dp, g2 = omb * dp;
            if (k1.on) { g1 *= drop_mul1(k1, (uint32_t)(hbase + i), (uint32_t)j); g2 *= drop_mul1(k2, (uint32_t)(hbase + i), (uint32_t)j); }
            rd1 = fmaf(A1[j * ST + i], g1, rd1);
            rd2 = fmaf(A2[j * ST + i], g2, rd2);
        }
        float dq[DH];
#pragma unroll
        for (int e = 0; e < DH; ++e) dq[e] = 0.f;
        for (int j = 0; j < S; ++j) {
            const float dp = dot_row<T, DH>(doi, sV + j * RS);
            float g1 = beta * dp, g2 = omb * dp;
            if (k1.on) { g1 *= drop_mul1(k1, (uint32_t)(hbase + i), (uint32_t)j); g2 *= drop_mul1(k2, (uint32_t)(hbase + i), (uint32_t)j); }
            const float ds1 = A1[j * ST + i] * (g1 - rd1);
            const float ds2 = A2[j * ST + i] * (g2 - rd2);
            A1[j * ST + i] = ds1;
            A2[j * ST + i] = ds2;
            axpy_row<T, DH>(dq, ds2 / sq, sK + j * RS);
            axpy_row<T, DH>(dch, -ds1 / rho[j], sC + j * RS);     // dN_ij * c_hat_j
        }
        store_row<T, DH>(DX + (int64_t)i * 4 * d, dq);
    }
    __syncthreads();

    // ---- column passes (lane = key j): dK_j = sum_i ds2_ij Q_i / sqrt(dh);  dch_j += sum_i dN_ij c_hat_i
    if (ract) {
        const int j = li;
        float dk[DH];
#pragma unroll
        for (int e = 0; e < DH; ++e) dk[e] = 0.f;
        for (int i = 0; i < S; ++i) axpy_row<T, DH>(dk, A2[j * ST + i] / sq, sQ + i * RS);
        store_row<T, DH>(DX + (int64_t)j * 4 * d + d, dk);
        for (int i = 0; i < S; ++i) axpy_row<T, DH>(dch, -A1[j * ST + i] / rho[i], sC + i * RS);
        // through c_hat = c / |c|:  dc = (dch - (dch . c_hat) c_hat) / |c|
        float cc[DH];
        load_row<T, DH>(X + (int64_t)j * 4 * d + 3 * d, cc);
        float dt = 0.f;
#pragma unroll
        for (int e = 0; e < DH; ++e) dt = fmaf(dch[e], cc[e], dt);
        dt /= (rho_i * rho_i);
        float dc[DH];
#pragma unroll
        for (int e = 0; e < DH; ++e) dc[e] = (dch[e] - dt * cc[e]) / rho_i;
        store_row<T, DH>(DX + (int64_t)j * 4 * d + 3 * d, dc);
    }
}

// ------------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------------
template <typename T, int DH, int GS> static int launch(const AttnArgs& a, bool bwd, hipStream_t st) {
    const size_t gb = attn_group_bytes(a.S, DH, GS, sizeof(T), bwd);
    const size_t lds_cap = 160 * 1024;
    PMGT_CHECK(gb <= lds_cap, -3, "attention: S=%d head_dim=%d needs %zu bytes of LDS per (sequence, head), more than a CU has",
               a.S, DH, gb);
    // groups per workgroup: fill <= ~78 KiB so two workgroups fit a CU; at least one wave of lanes if possible
    int ng = (int)std::min<size_t>(4, std::max<size_t>(1, (78 * 1024) / gb));
    const size_t shmem = gb * ng;
    auto kern = bwd ? attn_bwd_kernel<T, DH, GS> : attn_fwd_kernel<T, DH, GS>;
    if (shmem > 64 * 1024) {
        PMGT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    }
    const int groups = a.Tseq * a.H;
    hipLaunchKernelGGL(kern, dim3(cdiv(groups, ng)), dim3(ng * GS), shmem, st, a, (int)gb);
    PMGT_LAUNCH_OK();
    return 0;
}

template <typename T, int DH> static int launch_gs(const AttnArgs& a, bool bwd, hipStream_t st) {
    if (a.S <= 16) return launch<T, DH, 16>(a, bwd, st);
    if (a.S <= 32) return launch<T, DH, 32>(a, bwd, st);
    if (a.S <= 64) return launch<T, DH, 64>(a, bwd, st);
    return launch<T, DH, 128>(a, bwd, st);       // 64 < S <= 100: two waves per (sequence, head), one query row per lane
}

template <typename T> static int dispatch(const AttnArgs& a, bool bwd, hipStream_t st) {
    if (a.Tseq <= 0) return 0;
    if constexpr (sizeof(T) == 2) {
        if (!(a.opts & OPT_VALU_ATTENTION) && attn_mfma_supported(a)) return attn_mfma(a, bwd, st);   // bf16 perf path
    }
    PMGT_CHECK(a.S >= 1 && a.S <= 100, -3,
               "attention: sequence length %d exceeds the reference's max_position_embeddings default of 100 "
               "(pmgt/pmgt/configuration_pmgt.py:23)", a.S);
    switch (a.dh) {
        case 16: return launch_gs<T, 16>(a, bwd, st);
        case 32: return launch_gs<T, 32>(a, bwd, st);
        case 64: return launch_gs<T, 64>(a, bwd, st);
        case 128: return launch_gs<T, 128>(a, bwd, st);
        default:
            PMGT_CHECK(false, -3, "attention: head size %d not supported by the HIP path (16, 32, 64, 128)", a.dh);
    }
    return 0;
}

template <typename T> int attn_fwd(const AttnArgs& a, hipStream_t st) { return dispatch<T>(a, false, st); }
template <typename T> int attn_bwd(const AttnArgs& a, hipStream_t st) { return dispatch<T>(a, true, st); }
template int attn_fwd<float>(const AttnArgs&, hipStream_t);
template int attn_fwd<bf16>(const AttnArgs&, hipStream_t);
template int attn_bwd<float>(const AttnArgs&, hipStream_t);
template int attn_bwd<bf16>(const AttnArgs&, hipStream_t);

}  // namespace pmgt
