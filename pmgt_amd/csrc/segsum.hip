// Segment sums by node id (table mode of the feature projection, backward):
//
//     dE_all[n, :] = sum over tokens m with ids[m] == n of dE[m, :]
//
// so that dW_m = dE_all^T x table_m is a GEMM over N+2 rows instead of one over all M tokens (54x fewer flops at
// C2 / B = 1024).  Deterministic: tokens are ordered by a STABLE radix sort of (id, token index) (rocPRIM: index
// preparation, not arithmetic), every segment is summed in that order -- runs inside a 256-position chunk by one
// workgroup in registers, segments that span chunks through per-chunk partial rows added in chunk order.
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "segsum.h"

namespace pmgt {

static constexpr int SEG_CH = 64;       // sorted positions per WAVE (a chunk); a 256-thread workgroup walks four chunks

__global__ void seg_keys_kernel(const int64_t* __restrict__ ids, int M, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) { keys[m] = (uint32_t)ids[m]; vals[m] = (uint32_t)m; }
}

// seg_off[n] = first sorted position whose key is >= n  (n = 0 .. n_rows)
__global__ void seg_bounds_kernel(const uint32_t* __restrict__ skeys, int M, int n_rows, int* __restrict__ seg_off) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n > n_rows) return;
    int lo = 0, hi = M;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (skeys[mid] < (uint32_t)n) lo = mid + 1; else hi = mid;
    }
    seg_off[n] = lo;
}

// One WAVE per chunk of SEG_CH sorted positions; lane l owns columns 4l .. 4l+3 (+ 256 g): a row is read with 8-byte
// (bf16) / 16-byte (fp32) accesses by all 64 lanes, eight rows in flight.
// A run [a, b) of equal keys is COMPLETE when it is the whole segment: written to out[key].  Otherwise its partial sum
// goes to part[chunk][slot]: slot 0 if the run starts at the chunk start, else slot 1 (then it ends at the chunk end).
template <typename T, typename TO, int CG>
__global__ __launch_bounds__(256) void seg_sum_kernel(const T* __restrict__ src, int64_t lds_, const uint32_t* __restrict__ skeys,
                                                      const uint32_t* __restrict__ perm, const int* __restrict__ seg_off, int M,
                                                      int cols, TO* __restrict__ out, float* __restrict__ part) {
    __shared__ uint32_t sk_all[4][SEG_CH], sp_all[4][SEG_CH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x * 4 + wave;
    const int p0 = chunk * SEG_CH, p1 = min(M, p0 + SEG_CH);
    if (p0 >= M) return;
    uint32_t* sk = sk_all[wave];
    uint32_t* sp = sp_all[wave];
    {
        const int p = min(p0 + lane, M - 1);
        sk[lane] = skeys[p];
        sp[lane] = perm[p];
    }
    __builtin_amdgcn_wave_barrier();
    f32x4 acc[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int a = p0;
    uint32_t cur = sk[0];
    auto flush = [&](int b) {
        const bool complete = a == seg_off[cur] && b == seg_off[cur + 1];
#pragma unroll
        for (int g = 0; g < CG; ++g) {
            const int c = 4 * (lane + 64 * g);
            if (c < cols) {
                if (complete) store4<TO>(out + (int64_t)cur * cols + c, acc[g]);
                else *(f32x4*)(part + ((int64_t)chunk * 2 + (a == p0 ? 0 : 1)) * cols + c) = acc[g];
            }
            acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    const int n = p1 - p0;
    for (int i0 = 0; i0 < n; i0 += 8) {
        f32x4 v[8][CG];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u, n - 1);
            const T* row = src + (int64_t)sp[i] * lds_;
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                const int c = 4 * (lane + 64 * g);
                v[u][g] = c < cols ? load4<T>(row + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u;
            if (i < n) {
                if (sk[i] != cur) {
                    flush(p0 + i);
                    a = p0 + i;
                    cur = sk[i];
                }
#pragma unroll
                for (int g = 0; g < CG; ++g) acc[g] += v[u][g];
            }
        }
    }
    flush(p1);
}

// Segments that span several chunks (and empty ones): one workgroup per node adds the partial rows in chunk order.
template <typename T, int CG>
__global__ __launch_bounds__(256) void seg_fix_kernel(const int* __restrict__ seg_off, int n_rows, int cols, const float* __restrict__ part,
                                                      T* __restrict__ out) {
    const int k = blockIdx.x, tid = threadIdx.x;
    const int s = seg_off[k], e = seg_off[k + 1];
    const int c0 = s / SEG_CH, c1 = e > s ? (e - 1) / SEG_CH : c0;
    if (e > s && c0 == c1) return;          // a complete run inside one chunk: already written
    float acc[CG][2];
#pragma unroll
    for (int g = 0; g < CG; ++g) acc[g][0] = acc[g][1] = 0.f;
    if (e > s) {
        for (int c = c0; c <= c1; ++c) {
            const int slot = (c == c0 && s > c0 * SEG_CH) ? 1 : 0;
            const float* pr = part + ((int64_t)c * 2 + slot) * cols;
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                const int col = 2 * (tid + 256 * g);
                if (col < cols) { acc[g][0] += pr[col]; acc[g][1] += pr[col + 1]; }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < CG; ++g) {
        const int col = 2 * (tid + 256 * g);
        if (col < cols) {
            out[(int64_t)k * cols + col] = (T)acc[g][0];
            out[(int64_t)k * cols + col + 1] = (T)acc[g][1];
        }
    }
}

int64_t seg_sort_temp_bytes(int M) {
    size_t bytes = 0;
    uint32_t* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, (size_t)M, 0, 32, (hipStream_t)0);
    return (int64_t)bytes + 256;
}

int seg_sort(const int64_t* ids, int M, int n_rows, uint32_t* keys, uint32_t* vals, uint32_t* skeys, uint32_t* perm, int* seg_off,
             void* temp, int64_t temp_bytes, hipStream_t st) {
    PMGT_CHECK(M > 0 && n_rows > 0, -2, "seg_sort: empty input");
    hipLaunchKernelGGL(seg_keys_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, ids, M, keys, vals);
    PMGT_LAUNCH_OK();
    int bits = 1;
    while ((1ll << bits) < (int64_t)n_rows) ++bits;
    size_t need = 0;
    PMGT_HIP(rocprim::radix_sort_pairs(nullptr, need, keys, skeys, vals, perm, (size_t)M, 0, bits, st));
    PMGT_CHECK((int64_t)need <= temp_bytes, -4, "seg_sort: temporary storage too small (%lld > %lld)", (long long)need, (long long)temp_bytes);
    PMGT_HIP(rocprim::radix_sort_pairs(temp, need, keys, skeys, vals, perm, (size_t)M, 0, bits, st));
    hipLaunchKernelGGL(seg_bounds_kernel, dim3(cdiv(n_rows + 1, 256)), dim3(256), 0, st, skeys, M, n_rows, seg_off);
    PMGT_LAUNCH_OK();
    return 0;
}

int64_t seg_part_elems(int M, int cols) { return (int64_t)cdiv(M, SEG_CH) * 2 * cols; }

template <typename T, typename TO>
int seg_sum(const T* src, int64_t ld, const uint32_t* skeys, const uint32_t* perm, const int* seg_off, int M, int n_rows, int cols,
            TO* out, float* part, hipStream_t st) {
    PMGT_CHECK(cols % 4 == 0 && cols <= 1024 && ld % 4 == 0, -2, "seg_sum: cols=%d must be a multiple of 4 and <= 1024", cols);
    const int chunks = cdiv(cdiv(M, SEG_CH), 4);      // workgroups of four wave-chunks
    const int cg = cdiv(cols, 256);
    switch (cg) {
        case 1:
            hipLaunchKernelGGL((seg_sum_kernel<T, TO, 1>), dim3(chunks), dim3(256), 0, st, src, ld, skeys, perm, seg_off, M, cols, out, part);
            hipLaunchKernelGGL((seg_fix_kernel<TO, 1>), dim3(n_rows), dim3(256), 0, st, seg_off, n_rows, cols, part, out);
            break;
        case 2:
            hipLaunchKernelGGL((seg_sum_kernel<T, TO, 2>), dim3(chunks), dim3(256), 0, st, src, ld, skeys, perm, seg_off, M, cols, out, part);
            hipLaunchKernelGGL((seg_fix_kernel<TO, 2>), dim3(n_rows), dim3(256), 0, st, seg_off, n_rows, cols, part, out);
            break;
        default:
            hipLaunchKernelGGL((seg_sum_kernel<T, TO, 4>), dim3(chunks), dim3(256), 0, st, src, ld, skeys, perm, seg_off, M, cols, out, part);
            hipLaunchKernelGGL((seg_fix_kernel<TO, 4>), dim3(n_rows), dim3(256), 0, st, seg_off, n_rows, cols, part, out);
            break;
    }
    PMGT_LAUNCH_OK();
    return 0;
}
template int seg_sum<float, float>(const float*, int64_t, const uint32_t*, const uint32_t*, const int*, int, int, int, float*, float*, hipStream_t);
template int seg_sum<bf16, bf16>(const bf16*, int64_t, const uint32_t*, const uint32_t*, const int*, int, int, int, bf16*, float*, hipStream_t);
template int seg_sum<bf16, float>(const bf16*, int64_t, const uint32_t*, const uint32_t*, const int*, int, int, int, float*, float*, hipStream_t);

}  // namespace pmgt
