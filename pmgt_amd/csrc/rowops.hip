// Row-wise fused kernels: LayerNorm fwd/bwd and the PMGTEmbeddings modality mix fwd/bwd.
// One wave (64 lanes) owns one row; a lane holds 4-element chunks `lane + 64*i` of the row in
// registers (d <= 1024), so every row is read once and reduced with wave shuffles.  These kernels
// are HBM-bound: algorithmic bytes per row are listed in DESIGN.md.
#include "fp8.h"
#include "rowops.h"

namespace pmgt {

template <int NCH> __device__ __forceinline__ bool ch_ok(int lane, int i, int nch) { return lane + 64 * i < nch; }

__device__ __forceinline__ float sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// ------------------------------------------------------------------------------------------------
// LayerNorm forward
// ------------------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, float* __restrict__ stats,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int M, int d, float eps, DropCfg drop, const int* m_dev, char* __restrict__ q8,
                                                     float* __restrict__ q8_scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m_dev) M = min(M, *m_dev);
    if (m >= M) return;
    const int nch = d >> 2;
    const T* xr = x + (int64_t)m * d;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        v[i] = ch < nch ? load4<T>(xr + 4 * ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
        s += sum4(v[i]);
    }
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (lane + 64 * i < nch) {
            f32x4 t = v[i] - mean;
            ss += sum4(t * t);
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)d + eps);
    if (lane == 0) { stats[2 * (int64_t)m] = mean; stats[2 * (int64_t)m + 1] = rstd; }
    const DropKey dk = make_drop_key(drop);
    f32x4 oq[NCH];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        oq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ch < nch) {
            f32x4 g = *(const f32x4*)(gamma + 4 * ch), b = *(const f32x4*)(beta + 4 * ch);
            f32x4 o = (v[i] - mean) * rstd * g + b;
            if (dk.on) {
{ float dm[4]; drop_mul4(dk, (uint32_t)m, (uint32_t)ch, dm);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] *= dm[e]; }
            }
            store4<T>(y + (int64_t)m * d + 4 * ch, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) { oq[i][e] = to_f<T>(from_f<T>(o[e])); mx = fmaxf(mx, fabsf(oq[i][e])); }
        }
    }
    if (q8) {      // (uniform) fp8 mode: the stored row once more as per-row e4m3 + scale (fp8.h contract) for the next projection
        mx = wave_max(mx);
        const float inv = mx > 0.f ? E4M3_MAX / mx : 1.f;
        if (lane == 0) q8_scale[m] = mx > 0.f ? mx / E4M3_MAX : 1.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                int w = 0;
                w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(oq[i][0] * inv, -E4M3_MAX, E4M3_MAX),
                                                    __builtin_amdgcn_fmed3f(oq[i][1] * inv, -E4M3_MAX, E4M3_MAX), w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(oq[i][2] * inv, -E4M3_MAX, E4M3_MAX),
                                                    __builtin_amdgcn_fmed3f(oq[i][3] * inv, -E4M3_MAX, E4M3_MAX), w, true);
                *(int*)(q8 + (int64_t)m * d + 4 * ch) = w;
            }
        }
    }
}

template <typename T>
int ln_fwd(const T* x, T* y, float* stats, const float* gamma, const float* beta, int M, int d, float eps,
           DropCfg out_drop, hipStream_t st, const int* m_dev, void* q8v, float* q8_scale) {
    char* q8 = (char*)q8v;
    if (M <= 0) return 0;
    PMGT_CHECK(d % 4 == 0 && d <= 1024, -2, "ln_fwd: hidden size %d must be a multiple of 4 and <= 1024", d);
    dim3 grid(cdiv(M, 4)), block(256);
    if (d <= 256) hipLaunchKernelGGL((ln_fwd_kernel<T, 1>), grid, block, 0, st, x, y, stats, gamma, beta, M, d, eps, out_drop, m_dev, q8, q8_scale);
    else if (d <= 512) hipLaunchKernelGGL((ln_fwd_kernel<T, 2>), grid, block, 0, st, x, y, stats, gamma, beta, M, d, eps, out_drop, m_dev, q8, q8_scale);
    else hipLaunchKernelGGL((ln_fwd_kernel<T, 4>), grid, block, 0, st, x, y, stats, gamma, beta, M, d, eps, out_drop, m_dev, q8, q8_scale);
    PMGT_LAUNCH_OK();
    return 0;
}
template int ln_fwd<float>(const float*, float*, float*, const float*, const float*, int, int, float, DropCfg, hipStream_t, const int*, void*, float*);
template int ln_fwd<bf16>(const bf16*, bf16*, float*, const float*, const float*, int, int, float, DropCfg, hipStream_t, const int*, void*, float*);

// ------------------------------------------------------------------------------------------------
// LayerNorm backward.  64 rows per block (16 per wave, two at a time); dgamma/dbeta partials per block.
// ------------------------------------------------------------------------------------------------
template <typename T, int NCH, bool FROMY = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     T* __restrict__ dx, T* __restrict__ dx_drop, float* __restrict__ part,
                                                     int M, int d, DropCfg in_drop, DropCfg out_drop, const int* m_dev, int rpb,
                                                     const float* __restrict__ beta_y = nullptr) {
    __shared__ float red[3 * 1024];
    if (m_dev) M = min(M, *m_dev);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = d >> 2;
    const float inv_d = 1.f / (float)d;
    const DropKey ik = make_drop_key(in_drop), ok = make_drop_key(out_drop);
    f32x4 gam[NCH], dgam[NCH], dbet[NCH], dbia[NCH];
    f32x4 bety[FROMY ? NCH : 1], igam[FROMY ? NCH : 1];      // FROMY: x^ = (y - beta) * (1 / gamma)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        gam[i] = ch < nch ? *(const f32x4*)(gamma + 4 * ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
        dgam[i] = dbet[i] = dbia[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (FROMY) {
            bety[i] = ch < nch ? *(const f32x4*)(beta_y + 4 * ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) igam[i][e] = gam[i][e] != 0.f ? 1.f / gam[i][e] : 0.f;
        }
    }
    // one row of the wave: dx, and the row's terms of the three column sums (rows are taken in increasing order, so the sums
    // do not depend on how many rows are in flight)
    auto row = [&](int m, const f32x4 (&dyl)[NCH], const f32x4 (&xl)[NCH], float mean, float rstd) __attribute__((always_inline)) {
        f32x4 g[NCH], xh[NCH];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                f32x4 dyv = dyl[i];
                if (ik.on) {
{ float dm[4]; drop_mul4(ik, (uint32_t)m, (uint32_t)ch, dm);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dyv[e] *= dm[e]; }
                }
                if constexpr (FROMY) xh[i] = (xl[i] - bety[i]) * igam[i];
                else xh[i] = (xl[i] - mean) * rstd;
                g[i] = dyv * gam[i];
                dgam[i] += dyv * xh[i];
                dbet[i] += dyv;
                sg += sum4(g[i]);
                sgx += sum4(g[i] * xh[i]);
            } else {
                g[i] = xh[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        sg = wave_sum(sg) * inv_d;
        sgx = wave_sum(sgx) * inv_d;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                f32x4 o = (g[i] - sg - xh[i] * sgx) * rstd;
                store4<T>(dx + (int64_t)m * d + 4 * ch, o);
                if (dx_drop) {
                    if (ok.on) {
{ float dm[4]; drop_mul4(ok, (uint32_t)m, (uint32_t)ch, dm);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] *= dm[e]; }
                    }
                    store4<T>(dx_drop + (int64_t)m * d + 4 * ch, o);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dbia[i][e] += to_f<T>(from_f<T>(o[e]));   // column sum of what the GEMMs will read
            }
        }
    };
    // TWO rows of the wave in flight (their loads are issued before the first row's reductions): with one, the 32 waves of a
    // CU keep 32 KB of reads in flight, short of what the HBM latency needs at full rate
    for (int it = 0; it < rpb / 4; it += 2) {
        const int m0 = blockIdx.x * rpb + it * 4 + wave;
        if (m0 >= M) break;
        const bool two = it + 1 < rpb / 4 && m0 + 4 < M;      // (wave-uniform)
        const int m1 = two ? m0 + 4 : m0;
        f32x4 dy0[NCH], x0[NCH], dy1[NCH], x1[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = min(lane + 64 * i, nch - 1);      // unconditional loads (a predicate here serialises them); row() ignores the extras
            dy0[i] = load4<T>(dy + (int64_t)m0 * d + 4 * ch);
            x0[i] = load4<T>(x + (int64_t)m0 * d + 4 * ch);
            dy1[i] = load4<T>(dy + (int64_t)m1 * d + 4 * ch);
            x1[i] = load4<T>(x + (int64_t)m1 * d + 4 * ch);
        }
        const float mean0 = stats[2 * (int64_t)m0], rstd0 = stats[2 * (int64_t)m0 + 1];
        const float mean1 = stats[2 * (int64_t)m1], rstd1 = stats[2 * (int64_t)m1 + 1];
        row(m0, dy0, x0, mean0, rstd0);
        if (two) row(m1, dy1, x1, mean1, rstd1);
    }
    // cross-wave reduction of the partials, one wave at a time into LDS
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nch) {
                    f32x4 a = dgam[i], b = dbet[i], c = dbia[i];
                    if (w > 0) { a += *(f32x4*)(red + 4 * ch); b += *(f32x4*)(red + d + 4 * ch); c += *(f32x4*)(red + 2 * d + 4 * ch); }
                    *(f32x4*)(red + 4 * ch) = a;
                    *(f32x4*)(red + d + 4 * ch) = b;
                    *(f32x4*)(red + 2 * d + 4 * ch) = c;
                }
            }
        }
        __syncthreads();
    }
    float* out = part + (int64_t)blockIdx.x * 3 * d;
    for (int i = threadIdx.x; i < 3 * d; i += 256) out[i] = red[i];
}

template <typename T>
int ln_bwd(const T* dy, const T* x, const float* stats, const float* gamma, T* dx, T* dx_drop, float* part, int M,
           int d, DropCfg in_drop, DropCfg out_drop, hipStream_t st, const int* m_dev, const float* beta_y) {
    if (M <= 0) return 0;
    PMGT_CHECK(d % 4 == 0 && d <= 1024, -2, "ln_bwd: hidden size %d must be a multiple of 4 and <= 1024", d);
    dim3 grid(ln_bwd_parts(M)), block(256);
    const int rpb = ln_bwd_rows(M);
#define PMGT_LNB(NCH_)                                                                                                                     \
    do {                                                                                                                                   \
        if (beta_y) hipLaunchKernelGGL((ln_bwd_kernel<T, NCH_, true>), grid, block, 0, st, dy, x, stats, gamma, dx, dx_drop, part, M, d, in_drop, out_drop, m_dev, rpb, beta_y); \
        else hipLaunchKernelGGL((ln_bwd_kernel<T, NCH_, false>), grid, block, 0, st, dy, x, stats, gamma, dx, dx_drop, part, M, d, in_drop, out_drop, m_dev, rpb, beta_y);       \
    } while (0)
    if (d <= 256) PMGT_LNB(1);
    else if (d <= 512) PMGT_LNB(2);
    else PMGT_LNB(4);
#undef PMGT_LNB
    PMGT_LAUNCH_OK();
    return 0;
}
template int ln_bwd<float>(const float*, const float*, const float*, const float*, float*, float*, float*, int, int, DropCfg, DropCfg, hipStream_t, const int*, const float*);
template int ln_bwd<bf16>(const bf16*, const bf16*, const float*, const float*, bf16*, bf16*, float*, int, int, DropCfg, DropCfg, hipStream_t, const int*, const float*);

// ------------------------------------------------------------------------------------------------
// Embedding mix forward: a = softmax(Wa tanh([e_0; e_1; ...]) + ba); x = sum_k a_k e_k + pos[s] + role[s>0];
// h0 = dropout(LN(x)).   (pmgt/pmgt/modeling_pmgt.py:199-208; NF = len(feat_hidden_sizes))
// ------------------------------------------------------------------------------------------------
template <typename T, int NCH, int PHASE, int NF>
__global__ __launch_bounds__(256) void embed_mix_fwd_kernel(EmbedMix p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m >= p.M) return;
    const int d = p.d, nch = d >> 2;
    const int s = m % p.S;
    const T* E = (const T*)p.E + (p.e_rows ? p.e_rows[m] : (int64_t)m) * (PHASE == 2 ? d : NF * d);
    constexpr int NE = PHASE == 2 ? 1 : NF;      // token phase of the table mode: one row, already mixed
    f32x4 ev[NE][NCH];
    float a[NE];
    a[0] = 1.f;
    if (PHASE != 2) {
        float z[NF];
#pragma unroll
        for (int k = 0; k < NF; ++k) z[k] = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                f32x4 tv[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    ev[f][i] = load4<T>(E + f * d + 4 * ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) tv[f][e] = tanh_act<T>(ev[f][i][e]);
                }
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    float acc = sum4(tv[0] * *(const f32x4*)(p.Wa + (int64_t)k * NF * d + 4 * ch));
#pragma unroll
                    for (int f = 1; f < NF; ++f) acc += sum4(tv[f] * *(const f32x4*)(p.Wa + ((int64_t)k * NF + f) * d + 4 * ch));
                    z[k] += acc;
                }
            } else {
#pragma unroll
                for (int f = 0; f < NF; ++f) ev[f][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        float zm = -INFINITY;
#pragma unroll
        for (int k = 0; k < NF; ++k) { z[k] = wave_sum(z[k]) + p.ba[k]; zm = fmaxf(zm, z[k]); }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < NF; ++k) { z[k] = expf(z[k] - zm); den += z[k]; }
#pragma unroll
        for (int k = 0; k < NF; ++k) a[k] = z[k] / den;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NF; ++k) p.a[NF * (int64_t)m + k] = a[k];
        }
        if (PHASE == 1) {         // per-node mix only: F[n] = sum_k a_k e_k
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nch) {
                    f32x4 v = ev[0][i] * a[0];
#pragma unroll
                    for (int f = 1; f < NF; ++f) v += ev[f][i] * a[f];
                    store4<T>((T*)p.pre + (int64_t)m * d + 4 * ch, v);
                }
            }
            return;
        }
    } else {                        // token phase: the mixed feature of the node, already weighted
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            ev[0][i] = ch < nch ? load4<T>(E + 4 * ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }

    f32x4 x[NCH];
    float sum = 0.f;
    const float* pos = p.pos + (int64_t)s * d;
    const float* role = p.role + (s > 0 ? d : 0);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            f32x4 v = ev[0][i] * a[0];
#pragma unroll
            for (int f = 1; f < NE; ++f) v += ev[f][i] * a[f];
            v = v + *(const f32x4*)(pos + 4 * ch) + *(const f32x4*)(role + 4 * ch);
            store4<T>((T*)p.pre + (int64_t)m * d + 4 * ch, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = to_f<T>(from_f<T>(v[e]));   // LN sees what backward will re-read
            x[i] = v;
            sum += sum4(v);
        } else {
            x[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(sum) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (lane + 64 * i < nch) { f32x4 t = x[i] - mean; ss += sum4(t * t); }
    const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)d + p.eps);
    if (lane == 0) { p.stats[2 * (int64_t)m] = mean; p.stats[2 * (int64_t)m + 1] = rstd; }
    const DropKey dk = make_drop_key(p.drop);
    f32x4 oq[NCH];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        oq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ch < nch) {
            f32x4 o = (x[i] - mean) * rstd * *(const f32x4*)(p.gamma + 4 * ch) + *(const f32x4*)(p.beta + 4 * ch);
            if (dk.on) {
                float dm[4];
                drop_mul4(dk, (uint32_t)m, (uint32_t)ch, dm);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] *= dm[e];
            }
            store4<T>((T*)p.h0 + (int64_t)m * d + 4 * ch, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) { oq[i][e] = to_f<T>(from_f<T>(o[e])); mx = fmaxf(mx, fabsf(oq[i][e])); }
        }
    }
    if (p.q8) {      // (uniform) the stored row once more as e4m3 + scale (fp8.h contract)
        mx = wave_max(mx);
        const float inv = mx > 0.f ? E4M3_MAX / mx : 1.f;
        if (lane == 0) p.q8_scale[m] = mx > 0.f ? mx / E4M3_MAX : 1.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                int w = 0;
                w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(oq[i][0] * inv, -E4M3_MAX, E4M3_MAX),
                                                    __builtin_amdgcn_fmed3f(oq[i][1] * inv, -E4M3_MAX, E4M3_MAX), w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(oq[i][2] * inv, -E4M3_MAX, E4M3_MAX),
                                                    __builtin_amdgcn_fmed3f(oq[i][3] * inv, -E4M3_MAX, E4M3_MAX), w, true);
                *(int*)((char*)p.q8 + (int64_t)m * d + 4 * ch) = w;
            }
        }
    }
}

// Token phase of the table-mode embedding at hidden size 256 (bf16): x = F_all[ids[m]] + pos[s] + role[s > 0] -> LayerNorm -> dropout, and its
// backward.  A row is 32 lanes x 16 bytes; a wave task = positions (2 pp, 2 pp + 1) of TOK_R consecutive sequences: the position and role rows
// (fp32, 2 KB per token when read per row -- four times the bytes of the gathered row, all through L2) are read once per task, and the TOK_R
// row pairs' loads are in flight together.  `pre` == NULL (what the engine passes): the pre-LayerNorm sum is not stored -- the backward
// recomputes it from the same three rows (F_all is 3.7 MB and stays in L2) with the same operations, so both passes see the same bf16 values
// (1U of writes and 1U of reads less per step).
constexpr int TOK_R = 4;
__device__ __forceinline__ float half_sum(float v) {       // sum over the 32 lanes of this half of the wave
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float half_max(float v) {
    v = dpp_max<0xB1>(v);
    v = dpp_max<0x4E>(v);
    v = dpp_max<0x141>(v);
    v = dpp_max<0x140>(v);
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
struct TokTask {
    int q0, s;          // first sequence, position of this half-wave
    bool sv;            // the position exists (odd S: the last pair has one)
};
__host__ __device__ inline int tok_tasks(int M, int S) { return ((S + 1) >> 1) * ((M / S + TOK_R - 1) / TOK_R); }
__device__ __forceinline__ TokTask tok_task(int w, int S, int hs) {
    const int npp = (S + 1) >> 1;
    TokTask t;
    t.q0 = (w / npp) * TOK_R;
    t.s = 2 * (w % npp) + hs;
    t.sv = t.s < S;
    if (!t.sv) t.s = S - 1;
    return t;
}
// the pre-LayerNorm sum of one row as the forward rounds it (x: 8 gathered bf16; ps / rl: position and role rows, fp32)
__device__ __forceinline__ void tok_sum(const bf16x8& x, const f32x4 (&ps)[2], const f32x4 (&rl)[2], float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)x[j] * 1.f + ps[j >> 2][j & 3] + rl[j >> 2][j & 3];
}

template <typename T>
__global__ __launch_bounds__(256) void embed_tok8_fwd_kernel(EmbedMix p) {
    static_assert(sizeof(T) == 2, "bf16 only");
    constexpr int R = TOK_R, d = 256;
    const int lane = threadIdx.x & 63, hl = lane & 31, hs = lane >> 5;
    const int S = p.S, Tq = p.M / S;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= tok_tasks(p.M, S)) return;
    const TokTask tk = tok_task(w, S, hs);
    int64_t er[R];
#pragma unroll
    for (int k = 0; k < R; ++k) er[k] = p.e_rows[(int64_t)min(tk.q0 + k, Tq - 1) * S + tk.s];
    bf16x8 x[R];
#pragma unroll
    for (int k = 0; k < R; ++k) x[k] = *(const bf16x8*)((const bf16*)p.E + er[k] * d + 8 * hl);
    f32x4 ps[2], rl[2], gam[2], bet[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        ps[c] = *(const f32x4*)(p.pos + (int64_t)tk.s * d + 8 * hl + 4 * c);
        rl[c] = *(const f32x4*)(p.role + (tk.s > 0 ? d : 0) + 8 * hl + 4 * c);
        gam[c] = *(const f32x4*)(p.gamma + 8 * hl + 4 * c);
        bet[c] = *(const f32x4*)(p.beta + 8 * hl + 4 * c);
    }
    const DropKey dk = make_drop_key(p.drop);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const bool ok = tk.sv && tk.q0 + k < Tq;           // (uniform over the half-wave; every lane runs the reductions)
        const int64_t m = (int64_t)min(tk.q0 + k, Tq - 1) * S + tk.s;
        float v[8];
        tok_sum(x[k], ps, rl, v);
        if (p.pre && ok) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
            *(bf16x8*)((bf16*)p.pre + m * d + 8 * hl) = o;
        }
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = (float)(bf16)v[j]; sm += v[j]; }
        const float mean = half_sum(sm) / (float)d;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] -= mean; sq += v[j] * v[j]; }
        const float rstd = 1.f / sqrtf(half_sum(sq) / (float)d + p.eps);
        if (hl == 0 && ok) { p.stats[2 * m] = mean; p.stats[2 * m + 1] = rstd; }
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = v[j] * rstd * gam[j >> 2][j & 3] + bet[j >> 2][j & 3];
        if (dk.on) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float dm[4];
                drop_mul4(dk, (uint32_t)m, (uint32_t)(2 * hl + c), dm);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * c + e] *= dm[e];
            }
        }
        bf16x8 ob;
#pragma unroll
        for (int j = 0; j < 8; ++j) ob[j] = (bf16)o[j];
        if (ok) *(bf16x8*)((bf16*)p.h0 + m * d + 8 * hl) = ob;
        if (p.q8) {
            float oq[8], mx = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { oq[j] = (float)ob[j]; mx = fmaxf(mx, fabsf(oq[j])); }
            mx = half_max(mx);
            const float inv = mx > 0.f ? E4M3_MAX / mx : 1.f;
            if (hl == 0 && ok) p.q8_scale[m] = mx > 0.f ? mx / E4M3_MAX : 1.f;
            int wq[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                int w_ = 0;
                w_ = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(oq[4 * c] * inv, -E4M3_MAX, E4M3_MAX),
                                                     __builtin_amdgcn_fmed3f(oq[4 * c + 1] * inv, -E4M3_MAX, E4M3_MAX), w_, false);
                w_ = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(oq[4 * c + 2] * inv, -E4M3_MAX, E4M3_MAX),
                                                     __builtin_amdgcn_fmed3f(oq[4 * c + 3] * inv, -E4M3_MAX, E4M3_MAX), w_, true);
                wq[c] = w_;
            }
            if (ok) *(int2*)((char*)p.q8 + m * d + 8 * hl) = make_int2(wq[0], wq[1]);
        }
    }
}

// LayerNorm backward of the same rows: dF = LN'(dh0 * dropout mask), dgamma / dbeta partials per workgroup ([embed_part_elems]: the dWa / dba
// part is zero, as in the generic token phase).  Tasks are dealt round-robin over the grid of embed_bwd_parts(M) workgroups.
template <typename T>
__global__ __launch_bounds__(256) void embed_tok8_bwd_kernel(EmbedMix p) {
    static_assert(sizeof(T) == 2, "bf16 only");
    extern __shared__ float red[];       // embed_part_elems(d, NF) floats
    constexpr int R = TOK_R, d = 256;
    const int lane = threadIdx.x & 63, hl = lane & 31, hs = lane >> 5, wave = threadIdx.x >> 6;
    const int S = p.S, Tq = p.M / S;
    const int ntask = tok_tasks(p.M, S);
    const DropKey ik = make_drop_key(p.drop);
    f32x4 gam[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) gam[c] = *(const f32x4*)(p.gamma + 8 * hl + 4 * c);
    float dgam[8], dbet[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dgam[j] = dbet[j] = 0.f;
    for (int w = blockIdx.x * 4 + wave; w < ntask; w += gridDim.x * 4) {
        const TokTask tk = tok_task(w, S, hs);
        int64_t er[R];
#pragma unroll
        for (int k = 0; k < R; ++k) er[k] = p.e_rows[(int64_t)min(tk.q0 + k, Tq - 1) * S + tk.s];
        bf16x8 x[R], dy[R];
        float mean[R], rstd[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int64_t m = (int64_t)min(tk.q0 + k, Tq - 1) * S + tk.s;
            x[k] = *(const bf16x8*)((const bf16*)p.E + er[k] * d + 8 * hl);
            dy[k] = *(const bf16x8*)((const bf16*)p.dh0 + m * d + 8 * hl);
            mean[k] = p.stats[2 * m];
            rstd[k] = p.stats[2 * m + 1];
        }
        f32x4 ps[2], rl[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ps[c] = *(const f32x4*)(p.pos + (int64_t)tk.s * d + 8 * hl + 4 * c);
            rl[c] = *(const f32x4*)(p.role + (tk.s > 0 ? d : 0) + 8 * hl + 4 * c);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool ok = tk.sv && tk.q0 + k < Tq;
            const int64_t m = (int64_t)min(tk.q0 + k, Tq - 1) * S + tk.s;
            float v[8], g[8], dyv[8];
            tok_sum(x[k], ps, rl, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) dyv[j] = ok ? (float)dy[k][j] : 0.f;
            if (ik.on) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float dm[4];
                    drop_mul4(ik, (uint32_t)m, (uint32_t)(2 * hl + c), dm);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dyv[4 * c + e] *= dm[e];
                }
            }
            float sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = ((float)(bf16)v[j] - mean[k]) * rstd[k];      // x^
                g[j] = dyv[j] * gam[j >> 2][j & 3];
                dgam[j] += dyv[j] * v[j];
                dbet[j] += dyv[j];
                sg += g[j];
                sgx += g[j] * v[j];
            }
            sg = half_sum(sg) * (1.f / (float)d);
            sgx = half_sum(sgx) * (1.f / (float)d);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)((g[j] - sg - v[j] * sgx) * rstd[k]);
            if (ok) *(bf16x8*)((bf16*)p.dF + m * d + 8 * hl) = o;
        }
    }
    // workgroup sums through LDS, half-wave after half-wave (fixed order): dgamma | dbeta
    for (int hw = 0; hw < 8; ++hw) {
        if (wave == (hw >> 1) && hs == (hw & 1)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float* dg = red + 8 * hl + j;
                dg[0] = (hw > 0 ? dg[0] : 0.f) + dgam[j];
                dg[d] = (hw > 0 ? dg[d] : 0.f) + dbet[j];
            }
        }
        __syncthreads();
    }
    const int n_out = embed_part_elems(d, p.nf);
    float* out = p.part + (int64_t)blockIdx.x * n_out;
    for (int i = threadIdx.x; i < n_out; i += 256) out[i] = i < 2 * d ? red[i] : 0.f;
}

template <typename T> int embed_mix_fwd(const EmbedMix& e, hipStream_t st) {
    if (e.M <= 0) return 0;
    PMGT_CHECK(e.d % 4 == 0 && e.d <= 1024, -2, "embed_mix_fwd: hidden size %d must be a multiple of 4 and <= 1024", e.d);
    if constexpr (sizeof(T) == 2) {
        if (e.phase == 2 && e.d == 256 && e.e_rows != nullptr) {
            PMGT_CHECK(e.S > 0 && e.M % e.S == 0, -2, "embed_mix_fwd: %d tokens are not whole sequences of %d", e.M, e.S);
            const int tasks = tok_tasks(e.M, e.S);
            note_launch(LT_EMBED_TOK8);
            hipLaunchKernelGGL((embed_tok8_fwd_kernel<T>), dim3(cdiv(tasks, 4)), dim3(256), 0, st, e);
            PMGT_LAUNCH_OK();
            return 0;
        }
    }
    PMGT_CHECK(e.phase != 2 || e.pre != nullptr, -2, "embed_mix_fwd: the token phase at hidden size %d stores the pre-LayerNorm sum", e.d);
    PMGT_CHECK(e.nf >= 1 && e.nf <= 4, -2, "embed_mix_fwd: %d modalities (1 .. 4 are built)", e.nf);
    dim3 grid(cdiv(e.M, 4)), block(256);
#define PMGT_EMB_FWD_NF(PH, NF_)                                                                                  \
    do {                                                                                                          \
        if (e.d <= 256) hipLaunchKernelGGL((embed_mix_fwd_kernel<T, 1, PH, NF_>), grid, block, 0, st, e);         \
        else if (e.d <= 512) hipLaunchKernelGGL((embed_mix_fwd_kernel<T, 2, PH, NF_>), grid, block, 0, st, e);    \
        else hipLaunchKernelGGL((embed_mix_fwd_kernel<T, 4, PH, NF_>), grid, block, 0, st, e);                    \
    } while (0)
#define PMGT_EMB_FWD(PH)                                                                                          \
    do {                                                                                                          \
        if (e.nf == 2 || PH == 2) PMGT_EMB_FWD_NF(PH, 2);                                                         \
        else if (e.nf == 1) PMGT_EMB_FWD_NF(PH, 1);                                                               \
        else if (e.nf == 3) PMGT_EMB_FWD_NF(PH, 3);                                                               \
        else PMGT_EMB_FWD_NF(PH, 4);                                                                              \
    } while (0)
    if (e.phase == 1) PMGT_EMB_FWD(1);
    else if (e.phase == 2) PMGT_EMB_FWD(2);
    else PMGT_EMB_FWD(0);
#undef PMGT_EMB_FWD
#undef PMGT_EMB_FWD_NF
    PMGT_LAUNCH_OK();
    return 0;
}
template int embed_mix_fwd<float>(const EmbedMix&, hipStream_t);
template int embed_mix_fwd<bf16>(const EmbedMix&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// Embedding mix backward.  Partials per block: dgamma[d] | dbeta[d] | dWa[NF][NF d] | dba[NF] (padded to 4 floats)
// ------------------------------------------------------------------------------------------------
template <typename T, int NCH, int PHASE, int NF>
__global__ __launch_bounds__(256) void embed_mix_bwd_kernel(EmbedMix p) {
    extern __shared__ float red[];       // embed_part_elems(d, NF) floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = p.d, nch = d >> 2;
    const float inv_d = 1.f / (float)d;
    const DropKey ik = make_drop_key(p.drop);
    constexpr int NW = PHASE == 2 ? 1 : NF;      // token phase of the table mode: LayerNorm backward only
    f32x4 gam[NCH], dgam[NCH], dbet[NCH], dw[NW][NW][NCH], wa[NW][NW][NCH];     // [k][f]: row k of Wa, columns of modality f
    float dba[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) dba[k] = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        const bool ok = ch < nch;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        gam[i] = ok ? *(const f32x4*)(p.gamma + 4 * ch) : z;
        dgam[i] = dbet[i] = z;
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int f = 0; f < NW; ++f) {
                wa[k][f][i] = (ok && PHASE != 2) ? *(const f32x4*)(p.Wa + ((int64_t)k * NF + f) * d + 4 * ch) : z;
                dw[k][f][i] = z;
            }
    }
    const int rpb = ln_bwd_rows(p.M);
    for (int it = 0; it < rpb / 4; ++it) {
        const int m = blockIdx.x * rpb + it * 4 + wave;
        if (m >= p.M) break;
        f32x4 df[NCH];
        // the projected feature rows of token / node m: requested FIRST, so that they travel together with the LayerNorm operands
        // instead of after the two reductions that depend on those (one memory round trip per row instead of two)
        f32x4 ev[NW][NCH];
        if constexpr (PHASE != 2) {
            const T* E = (const T*)p.E + (p.e_rows ? p.e_rows[m] : (int64_t)m) * NF * d;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + 64 * i;
#pragma unroll
                for (int f = 0; f < NF; ++f) ev[f][i] = ch < nch ? load4<T>(E + f * d + 4 * ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (PHASE != 1) {     // LayerNorm backward of token m -> df (gradient wrt the pre-LN sum)
            const float mean = p.stats[2 * (int64_t)m], rstd = p.stats[2 * (int64_t)m + 1];
            f32x4 g[NCH], xh[NCH];
            float sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nch) {
                    f32x4 dyv = load4<T>((const T*)p.dh0 + (int64_t)m * d + 4 * ch);
                    if (ik.on) {
                        float dm[4];
                        drop_mul4(ik, (uint32_t)m, (uint32_t)ch, dm);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dyv[e] *= dm[e];
                    }
                    xh[i] = (load4<T>((const T*)p.pre + (int64_t)m * d + 4 * ch) - mean) * rstd;
                    g[i] = dyv * gam[i];
                    dgam[i] += dyv * xh[i];
                    dbet[i] += dyv;
                    sg += sum4(g[i]);
                    sgx += sum4(g[i] * xh[i]);
                } else {
                    g[i] = xh[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            sg = wave_sum(sg) * inv_d;
            sgx = wave_sum(sgx) * inv_d;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nch) {
                    df[i] = (g[i] - sg - xh[i] * sgx) * rstd;
                    store4<T>((T*)p.dF + (int64_t)m * d + 4 * ch, df[i]);
                } else {
                    df[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        } else {                // node phase: df = sum of the token gradients of node m
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + 64 * i;
                df[i] = ch >= nch ? (f32x4){0.f, 0.f, 0.f, 0.f}
                                  : (p.dF_f32 ? *(const f32x4*)((const float*)p.dF + (int64_t)m * d + 4 * ch)
                                              : load4<T>((const T*)p.dF + (int64_t)m * d + 4 * ch));
            }
        }
        if constexpr (PHASE != 2) {      // (token phase of the table mode: the mix is differentiated per node)
            // through f = sum_k a_k e_k
            float a[NF], da[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) { a[f] = p.a[NF * (int64_t)m + f]; da[f] = 0.f; }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + 64 * i;
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    if (ch < nch) da[f] += sum4(df[i] * ev[f][i]);
            }
            float dot = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f) { da[f] = wave_sum(da[f]); dot = f == 0 ? a[0] * da[0] : dot + a[f] * da[f]; }
            float dz[NF];
#pragma unroll
            for (int k = 0; k < NF; ++k) { dz[k] = a[k] * (da[k] - dot); dba[k] += dz[k]; }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nch) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        f32x4 tv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) tv[e] = tanh_act<T>(ev[f][i][e]);
                        f32x4 back = wa[0][f][i] * dz[0];
#pragma unroll
                        for (int k = 0; k < NF; ++k) {
                            dw[k][f][i] += tv * dz[k];
                            if (k > 0) back += wa[k][f][i] * dz[k];
                        }
                        store4<T>((T*)p.dE + (int64_t)m * NF * d + f * d + 4 * ch, df[i] * a[f] + (1.f - tv * tv) * back);
                    }
                }
            }
        }
    }
    // workgroup sums through LDS, wave after wave: dgamma | dbeta | dWa[k][f] | dba
    constexpr int NV = 2 + NW * NW;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nch) {
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        f32x4 v = k == 0 ? dgam[i] : (k == 1 ? dbet[i] : dw[(k - 2) / NW][(k - 2) % NW][i]);
                        float* dst = red + k * d + 4 * ch;
                        if (w > 0) v += *(f32x4*)dst;
                        *(f32x4*)dst = v;
                    }
                }
            }
            if (lane == 0) {   // dz is wave-uniform, so lane 0 carries the row sums
#pragma unroll
                for (int k = 0; k < 4; ++k) red[NV * d + k] = k < NW ? (w > 0 ? red[NV * d + k] : 0.f) + dba[k < NW ? k : 0] : 0.f;
            }
        }
        __syncthreads();
    }
    // token phase of the table mode: the dWa / dba part of the partial is zero (the node phase adds its own)
    const int n_out = embed_part_elems(d, NF), n_red = NV * d + 4;
    float* out = p.part + (int64_t)blockIdx.x * n_out;
    for (int i = threadIdx.x; i < n_out; i += 256) {
        float v;
        if (PHASE == 2 && NF != 1) v = i < 2 * d ? red[i] : 0.f;
        else v = red[i];
        out[i] = v;
    }
    (void)n_red;
}

template <typename T> int embed_mix_bwd(const EmbedMix& e, hipStream_t st) {
    if (e.M <= 0) return 0;
    PMGT_CHECK(e.d % 4 == 0 && e.d <= 1024, -2, "embed_mix_bwd: hidden size %d must be a multiple of 4 and <= 1024", e.d);
    PMGT_CHECK(e.nf >= 1 && e.nf <= 4, -2, "embed_mix_bwd: %d modalities (1 .. 4 are built)", e.nf);
    dim3 grid(embed_bwd_parts(e.M)), block(256);
    const size_t lds = (size_t)embed_part_elems(e.d, e.nf) * sizeof(float);
    if (e.phase == 2 && e.pre == nullptr) {      // the pre-LayerNorm sum is recomputed (what embed_tok8_fwd_kernel's callers pass)
        if constexpr (sizeof(T) == 2) {
            PMGT_CHECK(e.d == 256 && e.e_rows != nullptr && e.E != nullptr && e.pos != nullptr && e.role != nullptr && e.S > 0 && e.M % e.S == 0, -2,
                       "embed_mix_bwd: recomputed pre-LayerNorm sum needs the table-mode token phase at hidden size 256 (d=%d)", e.d);
            note_launch(LT_EMBED_TOK8);
            hipLaunchKernelGGL((embed_tok8_bwd_kernel<T>), grid, block, lds, st, e);
            PMGT_LAUNCH_OK();
            return 0;
        }
        PMGT_CHECK(false, -2, "embed_mix_bwd: fp32 stores the pre-LayerNorm sum");
    }
#define PMGT_EMB_BWD_NF(PH, NF_)                                                                                  \
    do {                                                                                                          \
        if (e.d <= 256) hipLaunchKernelGGL((embed_mix_bwd_kernel<T, 1, PH, NF_>), grid, block, lds, st, e);       \
        else if (e.d <= 512) hipLaunchKernelGGL((embed_mix_bwd_kernel<T, 2, PH, NF_>), grid, block, lds, st, e);  \
        else hipLaunchKernelGGL((embed_mix_bwd_kernel<T, 4, PH, NF_>), grid, block, lds, st, e);                  \
    } while (0)
#define PMGT_EMB_BWD(PH)                                                                                          \
    do {                                                                                                          \
        if (e.nf == 2) PMGT_EMB_BWD_NF(PH, 2);                                                                    \
        else if (e.nf == 1) PMGT_EMB_BWD_NF(PH, 1);                                                               \
        else if (e.nf == 3) PMGT_EMB_BWD_NF(PH, 3);                                                               \
        else PMGT_EMB_BWD_NF(PH, 4);                                                                              \
    } while (0)
    if (e.phase == 1) PMGT_EMB_BWD(1);
    else if (e.phase == 2) PMGT_EMB_BWD(2);
    else PMGT_EMB_BWD(0);
#undef PMGT_EMB_BWD
#undef PMGT_EMB_BWD_NF
    PMGT_LAUNCH_OK();
    return 0;
}
template int embed_mix_bwd<float>(const EmbedMix&, hipStream_t);
template int embed_mix_bwd<bf16>(const EmbedMix&, hipStream_t);

// grid (d / 256, max_pos): one position row per workgroup row (the serial 100-position loop of one workgroup took 18 us at every
// batch size); the two role rows are summed by row 0 over the S positions with independent loads, in increasing s (fixed order)
__global__ void pos_role_finish_kernel(const float* __restrict__ possum, int S, int d, int max_pos,
                                       float* __restrict__ dpos, float* __restrict__ drole, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (c >= d) return;
    const float v = s < S ? possum[(int64_t)s * d + c] : 0.f;
    float* o = dpos + (int64_t)s * d + c;
    *o = accumulate ? *o + v : v;
    if (s == 0) {
        float r1 = 0.f;
#pragma unroll 8
        for (int t = 1; t < S; ++t) r1 += possum[(int64_t)t * d + c];
        drole[c] = accumulate ? drole[c] + v : v;
        drole[d + c] = accumulate ? drole[d + c] + r1 : r1;
    }
}

int pos_role_finish(const float* possum, int S, int d, int max_pos, float* dpos, float* drole, bool accumulate,
                    hipStream_t st) {
    hipLaunchKernelGGL(pos_role_finish_kernel, dim3(cdiv(d, 256), max_pos), dim3(256), 0, st, possum, S, d, max_pos, dpos,
                       drole, accumulate ? 1 : 0);
    PMGT_LAUNCH_OK();
    return 0;
}

}  // namespace pmgt
