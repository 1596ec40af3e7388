// Segment sums by node id (table mode of the feature projection, backward):
//
//     dE_all[n, :] = sum over tokens m with ids[m] == n of dE[m, :]
//
// so that dW_m = dE_all^T x table_m is a GEMM over N+2 rows instead of one over all M tokens (54x fewer flops at
// C2 / B = 1024).  Deterministic: tokens are ordered by a STABLE sort of (id, token index) (index preparation, not
// arithmetic), every segment is summed in that order -- runs inside a 64-position chunk by one wave in registers,
// segments that span chunks through per-chunk partial rows added in chunk order.
//
// The sort is a hand-written least-significant-digit radix sort sized for what the keys are: node ids < n_rows <= M / 2
// (13 bits at the benchmark graph), i.e. ceil(bits / 8) counting passes -- two at C2 -- of
//     per-tile digit histogram ([tile][digit] count table) -> stable scatter (each tile derives its own output bases from the table; above
//     256 tiles a column-scan launch in between turns the table into prefixes once instead of once per tile).
// Inside a tile (16 rounds x 4 waves x 64 lanes, in index order) the rank of an element among its
// equals is: equals in earlier (round, wave) slots -- a prefix over 64 LDS counters per digit -- plus equals in lower lanes of
// its own wave -- eight ballots.  The result is THE stable order, the same permutation any stable sort produces.
// (Rounds 1 - 4 called rocPRIM here: 21 launches, 0.44 ms of GPU time per step next to the forward pass.)
#include <string.h>

#include "segsum.h"

namespace pmgt {

static constexpr int SEG_CH = 64;       // sorted positions per WAVE (a chunk); a 256-thread workgroup walks four chunks

static constexpr int RS_ROUNDS = 8, RS_TILE = 256 * RS_ROUNDS, RS_SLOTS = 4 * RS_ROUNDS, RS_MAXBINS = 256;

// digit counts of every tile of the current order, hist[tile][digit] (pass 0 reads the int64 ids, later passes the keys of the previous scatter)
template <bool FROM_IDS>
__global__ __launch_bounds__(256) void rs_hist_kernel(const int64_t* __restrict__ ids, const uint32_t* __restrict__ kin, int M, int shift, uint32_t mask,
                                                      int ntiles, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[RS_MAXBINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int j = 0; j < RS_ROUNDS; ++j) {
        const int idx = base + j * 256 + threadIdx.x;
        if (idx < M) atomicAdd(&h[((FROM_IDS ? (uint32_t)ids[idx] : kin[idx]) >> shift) & mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= mask) hist[(int64_t)blockIdx.x * (mask + 1u) + threadIdx.x] = h[threadIdx.x];      // [tile][digit]
}

// Large token counts (more than RS_FOLD_TILES tiles): the folded form below has EVERY tile walk the whole [tile][digit] table for its bases --
// O(tiles^2 x digits) reads behind a tiles-deep dependent loop per workgroup (768 tiles at B = 4 096: 0.3 GB of L2 reads per pass).  Here the
// column prefixes are computed ONCE per pass, in place: wave = one digit, lane l = a strip of ceil(tiles / 64) consecutive tiles (strip sums ->
// wave scan -> exclusive prefixes written over the counts); tot[digit] = the column total.  The scatter then reads one table row.
__global__ __launch_bounds__(256) void rs_colscan_kernel(uint32_t* __restrict__ hist, int ntiles, uint32_t nb, uint32_t* __restrict__ tot) {
    const int lane = threadIdx.x & 63;
    const uint32_t dg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (dg >= nb) return;                 // (wave-uniform)
    const int per = (ntiles + 63) / 64, t0 = lane * per, t1 = min(ntiles, t0 + per);
    uint32_t s = 0;
    for (int t = t0; t < t1; ++t) s += hist[(int64_t)t * nb + dg];
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    uint32_t run = inc - s;
    for (int t = t0; t < t1; ++t) {
        const uint32_t c = hist[(int64_t)t * nb + dg];
        hist[(int64_t)t * nb + dg] = run;
        run += c;
    }
    if (lane == 63) tot[dg] = inc;
}
static constexpr int RS_FOLD_TILES = 256;

// One counting pass: element (key, val) of tile t goes to base[digit][t] + (equals before it inside the tile).  FROM_IDS: pass 0
// reads the int64 ids (val = token index).  (A first form also counted the NEXT pass's digit per destination tile with global atomics, to
// save the later histogram launches: 393 216 device-scope atomics made that scatter 81 us; a histogram launch is 8.)
template <bool FROM_IDS, bool SCANNED = false>
__global__ __launch_bounds__(256) void rs_scatter_kernel(const int64_t* __restrict__ ids, const uint32_t* __restrict__ kin,
                                                         const uint32_t* __restrict__ vin, int M, int shift, uint32_t mask, int ntiles,
                                                         const uint32_t* __restrict__ base /* [tile][digit] counts (SCANNED: column prefixes) */,
                                                         const uint32_t* __restrict__ coltot /* SCANNED: [digit] column totals */,
                                                         uint32_t* __restrict__ kout, uint32_t* __restrict__ vout) {
    __shared__ uint16_t cnt[RS_MAXBINS][RS_SLOTS + 2];     // RS_SLOTS + 2 entries = an odd number of dwords per row: the per-digit prefix walks rows conflict-free
    __shared__ uint32_t tbase[RS_MAXBINS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x;
    for (int i = tid; i < RS_MAXBINS * (RS_SLOTS + 2) / 2; i += 256) ((uint32_t*)cnt)[i] = 0u;
    {   // first output position of (digit, this tile) = keys with a smaller digit anywhere + keys with this digit in earlier tiles: digit `tid` walks its
        // column of the [tile][digit] count table (coalesced across digits: the whole table is a few hundred KB in L2), then one exclusive scan
        // of the column totals over the digits.  (A first form ran that scan as a launch of its own, one workgroup: two launches of pure latency.)
        __shared__ uint32_t wtot[4];
        const uint32_t nb = mask + 1u;
        uint32_t tot = 0, pre = 0;
        if (SCANNED) {
            if ((uint32_t)tid < nb) { pre = base[(int64_t)tile * nb + tid]; tot = coltot[tid]; }
        } else if ((uint32_t)tid < nb) {
            uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            int t = 0;
            for (; t + 4 <= tile; t += 4) {
                t0 += base[(int64_t)t * nb + tid]; t1 += base[(int64_t)(t + 1) * nb + tid];
                t2 += base[(int64_t)(t + 2) * nb + tid]; t3 += base[(int64_t)(t + 3) * nb + tid];
            }
            for (; t < tile; ++t) t0 += base[(int64_t)t * nb + tid];
            pre = t0 + t1 + t2 + t3;
            t0 = t1 = t2 = t3 = 0;
            for (; t + 4 <= ntiles; t += 4) {
                t0 += base[(int64_t)t * nb + tid]; t1 += base[(int64_t)(t + 1) * nb + tid];
                t2 += base[(int64_t)(t + 2) * nb + tid]; t3 += base[(int64_t)(t + 3) * nb + tid];
            }
            for (; t < ntiles; ++t) t0 += base[(int64_t)t * nb + tid];
            tot = pre + t0 + t1 + t2 + t3;
        }
        uint32_t inc = tot;                 // inclusive scan over the wave's lanes, then over the four waves
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w_ = 0; w_ < wave; ++w_) woff += wtot[w_];
        if ((uint32_t)tid < nb) tbase[tid] = woff + inc - tot + pre;
    }
    __syncthreads();
    uint32_t key[RS_ROUNDS], val[RS_ROUNDS];
    uint32_t rank[RS_ROUNDS];
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; ++j) {
        const int idx = tile * RS_TILE + j * 256 + wave * 64 + lane;
        const bool valid = idx < M;
        const int ic = valid ? idx : M - 1;
        key[j] = FROM_IDS ? (uint32_t)ids[ic] : kin[ic];
        val[j] = FROM_IDS ? (uint32_t)ic : vin[ic];
        const uint32_t dg = (key[j] >> shift) & mask;
        uint64_t m = __builtin_amdgcn_ballot_w64(valid);            // lanes holding the same digit as this one
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dg >> b) & 1u;
            const uint64_t bal = __builtin_amdgcn_ballot_w64(bit);
            m &= bit ? bal : ~bal;
        }
        rank[j] = (uint32_t)__popcll(m & below);
        if (valid && rank[j] == 0) cnt[dg][j * 4 + wave] = (uint16_t)__popcll(m);
        if (!valid) rank[j] = 0xffffffffu;
    }
    __syncthreads();
    if ((uint32_t)tid <= mask) {           // exclusive prefix over the (round, wave) slots of digit `tid`
        uint32_t run = 0;
        for (int sidx = 0; sidx < RS_SLOTS; ++sidx) {
            const uint32_t c = cnt[tid][sidx];
            cnt[tid][sidx] = (uint16_t)run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; ++j) {
        if (rank[j] != 0xffffffffu) {
            const uint32_t dg = (key[j] >> shift) & mask;
            const uint32_t pos = tbase[dg] + cnt[dg][j * 4 + wave] + rank[j];
            kout[pos] = key[j];
            vout[pos] = val[j];
        }
    }
}

// seg_off[n] = first sorted position whose key is >= n  (n = 0 .. n_rows): one pass over the sorted keys -- position p writes the entries of
// the ids in (skeys[p - 1], skeys[p]] (more than one where ids in between do not occur), the last position also the tail up to n_rows
__global__ void seg_bounds_kernel(const uint32_t* __restrict__ skeys, int M, int n_rows, int* __restrict__ seg_off) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M) return;
    const int k = (int)skeys[p];
    const int lo = p == 0 ? 0 : (int)skeys[p - 1] + 1;
    for (int n = lo; n <= min(k, n_rows); ++n) seg_off[n] = p;
    if (p == M - 1)
        for (int n = k + 1; n <= n_rows; ++n) seg_off[n] = M;
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

static inline int rs_tiles(int M) { return cdiv(M, RS_TILE); }

int64_t seg_sort_temp_bytes(int M) {       // one [tiles][256] count table + [256] column totals (the scanned form of large token counts)
    return (int64_t)RS_MAXBINS * rs_tiles(M) * 4 + 256 + RS_MAXBINS * 4;
}

int seg_sort(const int64_t* ids, int M, int n_rows, uint32_t* keys, uint32_t* vals, uint32_t* skeys, uint32_t* perm, int* seg_off,
             void* temp, int64_t temp_bytes, hipStream_t st) {
    PMGT_CHECK(M > 0 && n_rows > 0, -2, "seg_sort: empty input");
    PMGT_CHECK(temp_bytes >= seg_sort_temp_bytes(M), -4, "seg_sort: temporary storage too small (%lld > %lld)", (long long)seg_sort_temp_bytes(M),
               (long long)temp_bytes);
    int bits = 1;
    while ((1ll << bits) < (int64_t)n_rows) ++bits;
    const int passes = cdiv(bits, 8), w = cdiv(bits, passes);       // digits of equal width (13 bits: 7 + 7)
    const int ntiles = rs_tiles(M);
    uint32_t* hist[1] = {(uint32_t*)temp};
    uint32_t* coltot = (uint32_t*)((char*)temp + (int64_t)RS_MAXBINS * ntiles * 4 + 256);
    const uint32_t mask = (1u << w) - 1u;
    const uint32_t *kin = nullptr, *vin = nullptr;
    for (int p = 0; p < passes; ++p) {
        // the last pass lands in (skeys, perm); the buffers alternate backwards from there
        uint32_t* kout = ((passes - 1 - p) & 1) ? keys : skeys;
        uint32_t* vout = ((passes - 1 - p) & 1) ? vals : perm;
        const int shift = p * w;
        if (p == 0) hipLaunchKernelGGL(rs_hist_kernel<true>, dim3(ntiles), dim3(256), 0, st, ids, kin, M, shift, mask, ntiles, hist[0]);
        else hipLaunchKernelGGL(rs_hist_kernel<false>, dim3(ntiles), dim3(256), 0, st, ids, kin, M, shift, mask, ntiles, hist[0]);
        if (ntiles > RS_FOLD_TILES) {
            hipLaunchKernelGGL(rs_colscan_kernel, dim3(cdiv((int)mask + 1, 4)), dim3(256), 0, st, hist[0], ntiles, mask + 1u, coltot);
            if (p == 0) hipLaunchKernelGGL((rs_scatter_kernel<true, true>), dim3(ntiles), dim3(256), 0, st, ids, kin, vin, M, shift, mask, ntiles, hist[0], coltot, kout, vout);
            else hipLaunchKernelGGL((rs_scatter_kernel<false, true>), dim3(ntiles), dim3(256), 0, st, ids, kin, vin, M, shift, mask, ntiles, hist[0], coltot, kout, vout);
        } else {
            if (p == 0) hipLaunchKernelGGL((rs_scatter_kernel<true, false>), dim3(ntiles), dim3(256), 0, st, ids, kin, vin, M, shift, mask, ntiles, hist[0], coltot, kout, vout);
            else hipLaunchKernelGGL((rs_scatter_kernel<false, false>), dim3(ntiles), dim3(256), 0, st, ids, kin, vin, M, shift, mask, ntiles, hist[0], coltot, kout, vout);
        }
        kin = kout; vin = vout;
    }
    PMGT_LAUNCH_OK();
    hipLaunchKernelGGL(seg_bounds_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, skeys, M, n_rows, seg_off);
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
