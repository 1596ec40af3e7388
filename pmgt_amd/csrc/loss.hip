// Loss heads of PMGT pre-training and their gradients wrt the encoder output:
//   GSR  (graph structure reconstruction): pmgt/pmgt/modeling_pmgt.py:537-546 + loop pmgt/pmgt/models.py:106-126
//   NFR  (masked node feature reconstruction): pmgt/pmgt/models.py:129-162, pmgt/pmgt/modeling_pmgt.py:549-569
// All row counts that depend on the random mask stay on the device (no host sync): kernels take a
// device-side count and tiles beyond it exit.
#include "loss.h"

namespace pmgt {

__device__ __forceinline__ float sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// ------------------------------------------------------------------------------------------------
// small single-block utilities
// ------------------------------------------------------------------------------------------------
// off[i] = sum_{k<i} num_pairs[k], off[B] = total
__global__ __launch_bounds__(1024) void pair_offsets_kernel(const int64_t* __restrict__ num_pairs, int B, int* __restrict__ off) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (B + 1023) / 1024;
    const int b0 = tid * per, b1 = min(B, b0 + per);
    int s = 0;
    for (int i = b0; i < b1; ++i) s += (int)num_pairs[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid > 0 ? part[tid - 1] : 0;
    for (int i = b0; i < b1; ++i) { off[i] = run; run += (int)num_pairs[i]; }
    if (tid == 1023) off[B] = part[1023];
}

// Several small device-to-device copies in ONE launch (the collated batch -> the workspace layout of a step: ids and masks of the
// target, pair and masked sequences): six hipMemcpyAsync calls cost ~6 us each on the step's critical path at small batch sizes.
struct MultiCopyArgs {
    struct J { const uint32_t* src; uint32_t* dst; int64_t n4; } j[8];
    int blk0[9];
    int njobs;
};
__global__ __launch_bounds__(256) void multi_copy_kernel(MultiCopyArgs a) {
    int k = 0;
    while (k + 1 < a.njobs && (int)blockIdx.x >= a.blk0[k + 1]) ++k;       // (uniform)
    const MultiCopyArgs::J& jb = a.j[k];
    const int64_t i0 = ((int64_t)(blockIdx.x - a.blk0[k]) * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (i0 + e < jb.n4) jb.dst[i0 + e] = jb.src[i0 + e];
}
int multi_copy(const CopyJob* jobs, int njobs, hipStream_t st) {
    MultiCopyArgs a;
    int cnt = 0, blocks = 0;
    for (int q = 0; q < njobs; ++q) {
        if (jobs[q].bytes <= 0) continue;
        PMGT_CHECK(cnt < 8 && jobs[q].bytes % 4 == 0 && ((uintptr_t)jobs[q].src % 4) == 0 && ((uintptr_t)jobs[q].dst % 4) == 0, -2,
                   "multi_copy: at most 8 jobs of 4-byte granularity");
        a.j[cnt].src = (const uint32_t*)jobs[q].src; a.j[cnt].dst = (uint32_t*)jobs[q].dst; a.j[cnt].n4 = jobs[q].bytes / 4;
        a.blk0[cnt] = blocks;
        blocks += (int)cdiv64(a.j[cnt].n4, 1024);
        ++cnt;
    }
    if (cnt == 0) return 0;
    a.blk0[cnt] = blocks;
    a.njobs = cnt;
    hipLaunchKernelGGL(multi_copy_kernel, dim3(blocks), dim3(256), 0, st, a);
    PMGT_LAUNCH_OK();
    return 0;
}

int pair_offsets(const int64_t* num_pairs, int B, int* off, hipStream_t st) {
    hipLaunchKernelGGL(pair_offsets_kernel, dim3(1), dim3(1024), 0, st, num_pairs, B, off);
    PMGT_LAUNCH_OK();
    return 0;
}

// NFR masking with the device RNG (pmgt/pmgt/models.py:132-151): writes masked ids and, per position,
// the id to reconstruct (-1 where not masked).
__global__ void nfr_generate_kernel(const int64_t* __restrict__ ids, int B, int S, int n_nodes, float random_ratio,
                                    float mask_ratio, const uint64_t* rng, int64_t* __restrict__ masked_ids,
                                    int64_t* __restrict__ tgt_full) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * S) return;
    const int s = idx % S;
    int64_t id = ids[idx];
    int64_t tgt = -1;
    if (s > 0 && id != 0) {
        DropCfg c1 = {rng, 0.5f, SITE_NFR1}, c2 = {rng, 0.5f, SITE_NFR2};
        const DropKey k1 = make_drop_key(c1), k2 = make_drop_key(c2);
        auto u32 = [](const DropKey& k, uint64_t i) {
            uint32_t x = fmix32(((uint32_t)i ^ k.k0) * 0x9E3779B1u + (uint32_t)(i >> 32));
            return fmix32(x + k.k1);
        };
        const float r1 = (float)(u32(k1, 2 * (uint64_t)idx) >> 8) * (1.f / 16777216.f);
        if (r1 < random_ratio) id = 2 + (int64_t)(u32(k1, 2 * (uint64_t)idx + 1) % (uint32_t)n_nodes);
        const float r2 = (float)(u32(k2, (uint64_t)idx) >> 8) * (1.f / 16777216.f);
        if (r2 < mask_ratio) { tgt = id; id = 1; }
    }
    masked_ids[idx] = id;
    tgt_full[idx] = tgt;
}

int nfr_generate(const int64_t* ids, int B, int S, int n_nodes, float random_ratio, float mask_ratio,
                 const uint64_t* rng, int64_t* masked_ids, int64_t* tgt_full, hipStream_t st) {
    hipLaunchKernelGGL(nfr_generate_kernel, dim3(cdiv(B * S, 256)), dim3(256), 0, st, ids, B, S, n_nodes, random_ratio,
                       mask_ratio, rng, masked_ids, tgt_full);
    PMGT_LAUNCH_OK();
    return 0;
}

// Row-major compaction of the masked positions: rows[k] = token row (seq_off + b) * S + s,
// tids[k] = id to reconstruct, *count = number of masked positions.
// One 1024-thread block walks the B * S positions in slabs of 1024 consecutive elements (coalesced 8-byte loads): rank inside
// the slab = popcount of the wave's ballot below the lane + the counts of the lower waves (16 counters in LDS), slab base carried
// in a register.  (The first form gave every thread 32 CONSECUTIVE positions -- 256-byte strided loads, twice, around a
// 20-barrier Hillis-Steele scan: 46 us at B = 1024 on the critical path in front of the encoder.)
__global__ __launch_bounds__(1024) void nfr_compact_kernel(const int64_t* __restrict__ tgt_full, int B, int S, int seq_off,
                                                           int64_t* __restrict__ rows, int64_t* __restrict__ tids,
                                                           int* __restrict__ count) {
    __shared__ int wcnt[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = B * S;
    int base = 0;
    // eight slabs of 1024 positions are requested together: with one load per slab iteration the kernel was 32 serialised memory round trips
    // (28 us at B = 1024, at the head of every step)
    for (int g0 = 0, it = 0; g0 < n; g0 += 8 * 1024) {
        int64_t tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = g0 + 1024 * u + tid;
            tv[u] = i < n ? tgt_full[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u, ++it) {
            const int s0 = g0 + 1024 * u;
            if (s0 >= n) break;                   // (uniform)
            const int i = s0 + tid;
            const int64_t t = tv[u];
            const bool hit = t >= 0;
            const unsigned long long m = __ballot(hit);
            const int below = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[it & 1][wave] = __popcll(m);
            __syncthreads();                      // (double-buffered counters: one barrier per slab)
            int woff = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const int c = wcnt[it & 1][w];
                woff += w < wave ? c : 0;
                tot += c;
            }
            if (hit) {
                const int k = base + woff + below;
                rows[k] = (int64_t)seq_off * S + i;
                tids[k] = t;
            }
            base += tot;
        }
    }
    if (tid == 0) *count = base;
}

int nfr_compact(const int64_t* tgt_full, int B, int S, int seq_off, int64_t* rows, int64_t* tids, int* count,
                hipStream_t st) {
    hipLaunchKernelGGL(nfr_compact_kernel, dim3(1), dim3(1024), 0, st, tgt_full, B, S, seq_off, rows, tids, count);
    PMGT_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// GSR: one wave per target node
// ------------------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ __launch_bounds__(256) void gsr_kernel(GsrArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= g.B) return;
    const int d = g.d, nch = d >> 2;
    const int64_t rs = g.cls_stride ? g.cls_stride : (int64_t)g.S * d;   // CLS row stride
    const T* H = (const T*)g.h;
    T* DH = (T*)g.dh;
    f32x4 zt[NCH], dzt[NCH];
    float n2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        zt[c] = ch < nch ? load4<T>(H + (int64_t)i * rs + 4 * ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
        dzt[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        n2 += sum4(zt[c] * zt[c]);
    }
    const float nt = fmaxf(sqrtf(wave_sum(n2)), 1e-12f);
#pragma unroll
    for (int c = 0; c < NCH; ++c) zt[c] = zt[c] / nt;     // z_hat_t
    const int j0 = g.off[i], j1 = g.off[i + 1];
    const float inv_cnt = 1.f / (float)(j1 - j0);
    float loss = 0.f;
    for (int j = j0; j < j1; ++j) {
        f32x4 zp[NCH];
        float p2 = 0.f, dot = 0.f;
        const T* row = H + (int64_t)(g.B + j) * rs;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + 64 * c;
            zp[c] = ch < nch ? load4<T>(row + 4 * ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
            p2 += sum4(zp[c] * zp[c]);
        }
        const float np = fmaxf(sqrtf(wave_sum(p2)), 1e-12f);
#pragma unroll
        for (int c = 0; c < NCH; ++c) { zp[c] = zp[c] / np; dot += sum4(zp[c] * zt[c]); }
        const float l = wave_sum(dot);
        const float y = g.labels[j];
        loss += fmaxf(l, 0.f) - l * y + log1pf(expf(-fabsf(l)));       // BCEWithLogits
        if (lane == 0 && g.logits) g.logits[j] = l;
        if (DH) {
            const float dl = (1.f / (1.f + expf(-l)) - y) * inv_cnt / (float)g.B;
            // d z_hat_p = dl * z_hat_t ; through the normalisation: (g - (g . z_hat) z_hat) / |z|
            const float gz = dl * l;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = lane + 64 * c;
                if (ch < nch) store4<T>(DH + (int64_t)(g.B + j) * rs + 4 * ch, (zt[c] * dl - zp[c] * gz) / np);
                dzt[c] += zp[c] * dl;
            }
        }
    }
    if (lane == 0) g.loss_part[i] = loss * inv_cnt / (float)g.B;
    if (DH) {
        float gz = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) gz += sum4(dzt[c] * zt[c]);
        gz = wave_sum(gz);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) store4<T>(DH + (int64_t)i * rs + 4 * ch, (dzt[c] - zt[c] * gz) / nt);
        }
    }
}

template <typename T> int gsr_fwd_bwd(const GsrArgs& g, hipStream_t st) {
    if (g.B <= 0) return 0;
    PMGT_CHECK(g.d % 4 == 0 && g.d <= 1024, -2, "gsr: hidden size %d must be a multiple of 4 and <= 1024", g.d);
    dim3 grid(cdiv(g.B, 4)), block(256);
    if (g.d <= 256) hipLaunchKernelGGL((gsr_kernel<T, 1>), grid, block, 0, st, g);
    else if (g.d <= 512) hipLaunchKernelGGL((gsr_kernel<T, 2>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gsr_kernel<T, 4>), grid, block, 0, st, g);
    PMGT_LAUNCH_OK();
    return 0;
}
template int gsr_fwd_bwd<float>(const GsrArgs&, hipStream_t);
template int gsr_fwd_bwd<bf16>(const GsrArgs&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// NFR: pred (from the GEMM) -> dpred in place + squared-error partials per modality
// ------------------------------------------------------------------------------------------------
// 4 consecutive table elements as fp32: storage type T, or (F8) e4m3 bytes times the table's scale
template <typename T, bool F8>
__device__ __forceinline__ f32x4 table_load4(const void* table, int64_t elem, float scale) {
    if constexpr (F8) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const int w = *(const int*)((const char*)table + elem);
        const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
        return (f32x4){lo[0] * scale, lo[1] * scale, hi[0] * scale, hi[1] * scale};
    } else {
        return load4<T>((const T*)table + elem);
    }
}

template <typename T, bool F8>
__global__ __launch_bounds__(256) void nfr_diff_kernel(NfrDiffArgs a) {
    __shared__ float red[4 * MAX_FEATS];
    const int n = *a.count;
    const int r0 = blockIdx.x * 8;
    float ss[MAX_FEATS] = {0.f, 0.f, 0.f, 0.f};
    if (r0 < n) {
        int F = 0, c0[MAX_FEATS + 1];
        float cm[MAX_FEATS];
#pragma unroll
        for (int f = 0; f < MAX_FEATS; ++f) {
            c0[f] = F;
            if (f < a.nf) F += a.F[f];
            // d (mean over modalities of mse_f) / d pred = (2 / nf) diff / (n F_f)   (nf = 2: diff / (n F_f))
            cm[f] = f < a.nf ? (2.f / (float)a.nf) / ((float)n * (float)a.F[f]) : 0.f;
        }
        c0[MAX_FEATS] = F;
        const int r1 = min(n, r0 + 8);
        for (int r = r0; r < r1; ++r) {
            const int64_t tid = a.tids[r];
            T* p = (T*)a.pred + (int64_t)r * F;
            for (int c4 = threadIdx.x * 4; c4 < F; c4 += 1024) {
                // (a group of 4 columns never straddles two modalities: every F_f is a multiple of 4)
                const int f = (c4 >= c0[1]) + (c4 >= c0[2] && a.nf > 2) + (c4 >= c0[3] && a.nf > 3);
                f32x4 tgt, diff;
                float cf;
                if (f == 0) { tgt = table_load4<T, F8>(a.table[0], tid * a.F[0] + c4, a.scale[0]); cf = cm[0]; }
                else if (f == 1) { tgt = table_load4<T, F8>(a.table[1], tid * a.F[1] + (c4 - c0[1]), a.scale[1]); cf = cm[1]; }
                else if (f == 2) { tgt = table_load4<T, F8>(a.table[2], tid * a.F[2] + (c4 - c0[2]), a.scale[2]); cf = cm[2]; }
                else { tgt = table_load4<T, F8>(a.table[3], tid * a.F[3] + (c4 - c0[3]), a.scale[3]); cf = cm[3]; }
                diff = load4<T>(p + c4) - tgt;
                const float s4 = sum4(diff * diff);
                ss[0] += f == 0 ? s4 : 0.f;
                ss[1] += f == 1 ? s4 : 0.f;
                ss[2] += f == 2 ? s4 : 0.f;
                ss[3] += f == 3 ? s4 : 0.f;
                store4<T>(p + c4, diff * cf);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int f = 0; f < MAX_FEATS; ++f) {
        ss[f] = wave_sum(ss[f]);
        if (lane == 0) red[4 * f + wave] = ss[f];
    }
    __syncthreads();
    if (threadIdx.x < MAX_FEATS) {
        const int f = threadIdx.x;
        a.sse_part[MAX_FEATS * (int64_t)blockIdx.x + f] = (red[4 * f] + red[4 * f + 1]) + (red[4 * f + 2] + red[4 * f + 3]);
    }
}

template <typename T> int nfr_diff(const NfrDiffArgs& a, hipStream_t st) {
    if (a.cap <= 0) return 0;
    PMGT_CHECK(a.nf >= 1 && a.nf <= MAX_FEATS, -2, "nfr_diff: %d modalities (1 .. %d are built)", a.nf, MAX_FEATS);
    for (int f = 0; f < a.nf; ++f) PMGT_CHECK(a.F[f] > 0 && a.F[f] % 4 == 0, -2, "nfr_diff: feature sizes must be multiples of 4");
    if (a.tables_f8) {
        if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((nfr_diff_kernel<T, true>), dim3(nfr_diff_parts(a.cap)), dim3(256), 0, st, a);
        else PMGT_CHECK(false, -2, "nfr_diff: e4m3 tables belong to the fp8 mode (bf16 activations)");
    } else {
        hipLaunchKernelGGL((nfr_diff_kernel<T, false>), dim3(nfr_diff_parts(a.cap)), dim3(256), 0, st, a);
    }
    PMGT_LAUNCH_OK();
    return 0;
}
template int nfr_diff<float>(const NfrDiffArgs&, hipStream_t);
template int nfr_diff<bf16>(const NfrDiffArgs&, hipStream_t);

// dst[rows[k], :] = src[k, :] (or += with `add`; the rows are distinct) for k < *count
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const T* __restrict__ src, const int64_t* __restrict__ rows,
                                                           const int* __restrict__ count, int d, T* __restrict__ dst, int add) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= *count) return;
    const int lane = threadIdx.x & 63;
    const int64_t r = rows[k];
    for (int c4 = lane * 4; c4 < d; c4 += 256) {
        f32x4 v = load4<T>(src + (int64_t)k * d + c4);
        if (add) v += load4<T>(dst + r * d + c4);
        store4<T>(dst + r * d + c4, v);
    }
}
template <typename T>
int scatter_rows(const T* src, const int64_t* rows, const int* count, int cap, int d, T* dst, hipStream_t st, bool add) {
    if (cap <= 0) return 0;
    hipLaunchKernelGGL((scatter_rows_kernel<T>), dim3(cdiv(cap, 4)), dim3(256), 0, st, src, rows, count, d, dst, add ? 1 : 0);
    PMGT_LAUNCH_OK();
    return 0;
}
template int scatter_rows<float>(const float*, const int64_t*, const int*, int, int, float*, hipStream_t, bool);
template int scatter_rows<bf16>(const bf16*, const int64_t*, const int*, int, int, bf16*, hipStream_t, bool);

__global__ void build_need_rows_kernel(int B, int P, int S, const int64_t* __restrict__ nfr_rows,
                                       const int* __restrict__ nfr_count, int64_t* __restrict__ rows, int* __restrict__ count,
                                       int* __restrict__ inv) {
    const int n = *nfr_count;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int64_t row = -1;
    if (i < B + P) row = (int64_t)i * S;              // CLS token of sequence i (targets, then pairs)
    else if (i < B + P + n) row = nfr_rows[i - B - P];
    if (row >= 0) {
        rows[i] = row;
        if (inv) inv[row] = i;                            // token row -> compact row (the caller filled `inv` with -1)
    }
    if (i == 0) *count = B + P + n;
}
int build_need_rows(int B, int P, int S, const int64_t* nfr_rows, const int* nfr_count, int64_t* rows, int* count,
                    hipStream_t st, int* inv, int64_t n_tokens) {
    const int cap = B + P + B * (S - 1 > 0 ? S - 1 : 1);
    if (inv) PMGT_HIP(hipMemsetAsync(inv, 0xFF, (size_t)n_tokens * sizeof(int), st));
    hipLaunchKernelGGL(build_need_rows_kernel, dim3(cdiv(cap, 256)), dim3(256), 0, st, B, P, S, nfr_rows, nfr_count, rows, count, inv);
    PMGT_LAUNCH_OK();
    return 0;
}

// out = {loss, gsr, nfr};  nfr = mean over modalities of sse_f / (n F_f)  (torch.stack(loss).mean(), modeling_pmgt.py:565-569);
// n == 0 gives NaN like the reference
__global__ __launch_bounds__(64) void loss_finish_kernel(const float* __restrict__ gsr_part, int B,
                                                         const float* __restrict__ sse_part, int nparts,
                                                         const int* __restrict__ count, FeatSizes fs, int with_nfr,
                                                         float* __restrict__ out, int* __restrict__ count_out) {
    const int lane = threadIdx.x;
    float g = 0.f;
    for (int i = lane; i < B; i += 64) g += gsr_part[i];
    g = wave_sum(g);
    float nfr = 0.f;
    if (with_nfr) {
        const int n = *count;
        const int used = (n + 7) / 8;
        float ss[MAX_FEATS] = {0.f, 0.f, 0.f, 0.f};
        for (int i = lane; i < used && i < nparts; i += 64) {
#pragma unroll
            for (int f = 0; f < MAX_FEATS; ++f) ss[f] += sse_part[MAX_FEATS * i + f];
        }
        float tot = 0.f;
#pragma unroll
        for (int f = 0; f < MAX_FEATS; ++f) {
            ss[f] = wave_sum(ss[f]);
            if (f < fs.nf) {
                const float mse = ss[f] / ((float)n * (float)fs.F[f]);
                tot = f == 0 ? mse : tot + mse;
            }
        }
        nfr = (1.f / (float)fs.nf) * tot;
    }
    if (lane == 0) {
        out[0] = g + nfr; out[1] = g; out[2] = nfr;
        if (count_out && with_nfr) *count_out = *count;      // the caller's copy of the masked-row count (was a 4-byte memcpy launch)
    }
}

int loss_finish(const float* gsr_part, int B, const float* sse_part, int nparts, const int* count, FeatSizes fs,
                bool with_nfr, float* out, hipStream_t st, int* count_out) {
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, gsr_part, B, sse_part, nparts, count, fs,
                       with_nfr ? 1 : 0, out, count_out);
    PMGT_LAUNCH_OK();
    return 0;
}

}  // namespace pmgt
