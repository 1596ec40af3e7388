// Fused global-norm clip + AdamW over the flat parameter buffer, and the weight "mirror"
// (bf16 / transposed copies the GEMMs read).
//   clip:  torch.nn.utils.clip_grad_norm_ semantics (pmgt/base_trainer.py:314 via PL gradient_clip_val)
//   AdamW: DenseSparseAdamW dense branch, pmgt/optimizers.py:256-270
// Step count, bias corrections and the clip coefficient live in device memory so the whole step can
// be captured in a hipGraph and replayed.
#include "optim.h"

namespace pmgt {

__global__ __launch_bounds__(256) void sqnorm_part_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
    float s = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            f32x4 v = *(const f32x4*)(g + i);
            s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        } else {
            for (int64_t k = i; k < n; ++k) s += g[k] * g[k];
        }
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// scal: [0] = clip coefficient, [1] = lr / bc1, [2] = 1 / sqrt(bc2), [3] = total grad norm (pre-clip)
__global__ __launch_bounds__(64) void adam_prepare_kernel(const float* __restrict__ part, int nparts, float max_norm,
                                                          float lr, float b1, float b2, int64_t* __restrict__ step,
                                                          float* __restrict__ scal) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += (double)part[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) {
        const int64_t t = step[0] + 1;
        step[0] = t;
        const double norm = sqrt(s);
        double coef = 1.0;
        if (max_norm > 0.f) coef = fmin((double)max_norm / (norm + 1e-6), 1.0);
        const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
        scal[0] = (float)coef;
        scal[1] = (float)((double)lr / bc1);
        scal[2] = (float)(1.0 / sqrt(bc2));
        scal[3] = (float)norm;
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const uint8_t* __restrict__ decay, int64_t n,
                                                    float lr, float wd, float b1, float b2, float eps,
                                                    const float* __restrict__ scal) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float coef = scal[0], step_size = scal[1], inv_sqrt_bc2 = scal[2];
    const int cnt = (int)min((int64_t)4, n - i);
    for (int k = 0; k < cnt; ++k) {
        const int64_t j = i + k;
        const float gg = g[j] * coef;
        float pp = p[j] * (1.f - lr * (decay[j] ? wd : 0.f));
        const float mm = m[j] * b1 + gg * (1.f - b1);
        const float vv = v[j] * b2 + gg * gg * (1.f - b2);
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        pp -= step_size * (mm / denom);
        p[j] = pp; m[j] = mm; v[j] = vv;
    }
}

int adamw_step(const AdamArgs& a, hipStream_t st) {
    if (a.n <= 0) return 0;
    const int nparts = (int)std::min<int64_t>(1024, cdiv64(a.n, 1024));
    hipLaunchKernelGGL(sqnorm_part_kernel, dim3(nparts), dim3(256), 0, st, a.g, a.n, a.part);
    PMGT_LAUNCH_OK();
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, st, a.part, nparts, a.max_norm, a.lr, a.b1, a.b2, a.step, a.scal);
    PMGT_LAUNCH_OK();
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)cdiv64(cdiv64(a.n, 4), 256)), dim3(256), 0, st, a.p, a.g, a.m, a.v,
                       a.decay, a.n, a.lr, a.wd, a.b1, a.b2, a.eps, a.scal);
    PMGT_LAUNCH_OK();
    return 0;
}

__global__ void advance_counter_kernel(uint64_t* rng) { rng[1] += 1; }
int advance_rng(uint64_t* rng, hipStream_t st) {
    hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(1), 0, st, rng);
    PMGT_LAUNCH_OK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// mirror: for each 2-D weight W[R,C] (fp32 master) write W as T and/or W^T as T
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mirror_kernel(const float* __restrict__ params, T* __restrict__ mirror,
                                                     const MirrorDesc* __restrict__ desc, int ndesc) {
    __shared__ float tile[32][33];
    int k = 0;
    while (k + 1 < ndesc && (int)blockIdx.x >= desc[k + 1].tile_start) ++k;
    const MirrorDesc dsc = desc[k];
    const int tl = blockIdx.x - dsc.tile_start;
    const int tc = (dsc.cols + 31) / 32;
    const int r0 = (tl / tc) * 32, c0 = (tl % tc) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const float* W = params + dsc.src;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        const float v = (r < dsc.rows && c < dsc.cols) ? W[(int64_t)r * dsc.cols + c] : 0.f;
        tile[i][tx] = v;
        if (dsc.dst >= 0 && r < dsc.rows && c < dsc.cols) mirror[dsc.dst + (int64_t)r * dsc.cols + c] = from_f<T>(v);
    }
    __syncthreads();
    if (dsc.dst_t >= 0) {
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i, r = r0 + tx;    // output row = original column
            if (c < dsc.cols && r < dsc.rows) mirror[dsc.dst_t + (int64_t)c * dsc.rows + r] = from_f<T>(tile[tx][i]);
        }
    }
    if (dsc.dst_t_hm >= 0) {
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i, r = r0 + tx;
            if (c < dsc.cols && r < dsc.rows) {
                const int m = r / dsc.hm_d, h = (r % dsc.hm_d) / dsc.hm_dh, w = r % dsc.hm_dh;
                const int rh = (h * 4 + m) * dsc.hm_dh + w;
                mirror[dsc.dst_t_hm + (int64_t)c * dsc.rows + rh] = from_f<T>(tile[tx][i]);
            }
        }
    }
}

template <typename T>
int build_mirror(const float* params, T* mirror, const MirrorDesc* desc_dev, int ndesc, int total_tiles, hipStream_t st) {
    if (total_tiles <= 0) return 0;
    hipLaunchKernelGGL((mirror_kernel<T>), dim3(total_tiles), dim3(256), 0, st, params, mirror, desc_dev, ndesc);
    PMGT_LAUNCH_OK();
    return 0;
}
template int build_mirror<float>(const float*, float*, const MirrorDesc*, int, int, hipStream_t);
template int build_mirror<bf16>(const float*, bf16*, const MirrorDesc*, int, int, hipStream_t);

// fp32 -> T cast of a flat array (feature tables, materialised inputs)
template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) store4<T>(dst + i, *(const f32x4*)(src + i));
        else for (int64_t k = i; k < n; ++k) dst[k] = from_f<T>(src[k]);
    }
}
template <typename T> int cast_f32(const float* src, T* dst, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    const int blocks = (int)std::min<int64_t>(8192, cdiv64(cdiv64(n, 4), 256));
    hipLaunchKernelGGL((cast_kernel<T>), dim3(blocks), dim3(256), 0, st, src, dst, n);
    PMGT_LAUNCH_OK();
    return 0;
}
template int cast_f32<float>(const float*, float*, int64_t, hipStream_t);
template int cast_f32<bf16>(const float*, bf16*, int64_t, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void uncast_kernel(const T* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = to_f<T>(src[i]);
}
template <typename T> int cast_to_f32(const T* src, float* dst, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    const int blocks = (int)std::min<int64_t>(8192, cdiv64(n, 256));
    hipLaunchKernelGGL((uncast_kernel<T>), dim3(blocks), dim3(256), 0, st, src, dst, n);
    PMGT_LAUNCH_OK();
    return 0;
}
template int cast_to_f32<float>(const float*, float*, int64_t, hipStream_t);
template int cast_to_f32<bf16>(const bf16*, float*, int64_t, hipStream_t);

// Clock probe (measurement plumbing of bench.py): one wave per workgroup writes {shader-cycle counter, 100 MHz wall counter, XCC id}.
// Two probes on the stream around a timed region give the shader clock the region sustained, per XCC (the cycle counters of
// different XCCs need not share an origin: samples are paired by XCC id).
__global__ void clock_probe_kernel(uint64_t* __restrict__ out) {
    if (threadIdx.x != 0) return;
    const uint64_t cyc = __builtin_readcyclecounter();
    const uint64_t wall = __builtin_amdgcn_s_memrealtime();
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;      // hwreg(HW_REG_XCC_ID, 0, 4)
    out[blockIdx.x * 4 + 0] = cyc;
    out[blockIdx.x * 4 + 1] = wall;
    out[blockIdx.x * 4 + 2] = xcc;
    out[blockIdx.x * 4 + 3] = 1;
}
int clock_probe(uint64_t* out, int blocks, hipStream_t st) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(64), 0, st, out);
    PMGT_LAUNCH_OK();
    return 0;
}

}  // namespace pmgt
