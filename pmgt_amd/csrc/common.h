// Shared device helpers for the PMGT gfx950 kernels: storage types (fp32 parity mode / bf16 perf
// mode), wave64 reductions, the counter-based dropout RNG and launch/error plumbing.
// Written for CDNA4 only (wavefront = 64, MFMA, 160 KiB LDS); no CUDA / multi-backend paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace pmgt {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int WAVE = 64;

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define PMGT_CHECK(cond, code, ...)                                                               \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            ::pmgt::set_error(__VA_ARGS__);                                                       \
            return (code);                                                                        \
        }                                                                                         \
    } while (0)
#define PMGT_HIP(expr)                                                                            \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            ::pmgt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,   \
                              __LINE__);                                                          \
            return -100;                                                                          \
        }                                                                                         \
    } while (0)
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: set once per (call site, device) -- one bit per
// device ordinal; two threads racing on the first launch both set it (idempotent)
#define PMGT_SMEM_ATTR(kern, bytes)                                                               \
    do {                                                                                          \
        static std::atomic<uint64_t> done__{0};                                                   \
        int dev__ = 0;                                                                            \
        PMGT_HIP(hipGetDevice(&dev__));                                                           \
        const uint64_t bit__ = 1ull << (dev__ & 63);                                              \
        if (!(done__.load(std::memory_order_acquire) & bit__)) {                                  \
            PMGT_HIP(hipFuncSetAttribute((kern), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
            done__.fetch_or(bit__, std::memory_order_release);                                    \
        }                                                                                         \
    } while (0)
#define PMGT_LAUNCH_OK()                                                                          \
    do {                                                                                          \
        hipError_t e__ = hipGetLastError();                                                       \
        if (e__ != hipSuccess) {                                                                  \
            ::pmgt::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__),         \
                              __FILE__, __LINE__);                                                \
            return -101;                                                                          \
        }                                                                                         \
    } while (0)

// ---- which kernel families a call chain launched (test instrumentation: per host thread, no effect on the launches) ---------
// The dispatchers pick a kernel by shape; a parity test at a given size is only a test of the kernel it believes it covers if that
// kernel actually ran.  Each dispatcher notes the family it launches; tests read the counts through pmgt_launch_trace_count().
enum LaunchTag {
    LT_GEMM_WSR = 0, LT_GEMM_WSR_LNB, LT_GEMM_WSR512, LT_GEMM_WS, LT_NT_BIG, LT_NT_BIG_GATHER, LT_NT_BIG_128, LT_NT_LNB, LT_NT_TILE,
    LT_TN_BIG, LT_TN_BIG_GATHER, LT_TN_DMA, LT_TN_DMA_GATHER, LT_TN_TILE, LT_ATTN_TILES_FWD, LT_ATTN_TILES_BWD, LT_QKVC_ATTN_FWD,
    LT_ATTN_BWD_WGRAD, LT_F8_BIG, LT_F8_TILE, LT_F8_WSR512, LT_GEMM_ROWLN, LT_NT_LNF, LT_EMBED_TOK8, LT_QKVC_ATTN_FWD_VC, LT_ATTN_BWD_WGRAD_VC, LT_NT_VC, LT_ATTN_BWD_WGRAD_VC2, LT_COUNT
};
void note_launch(int tag);

// ---- scalar conversions -----------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f(T x);
template <> __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }   // RNE, NaN-safe

// Load / store 4 consecutive elements as fp32 (16 B for float, 8 B for bf16).
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
    bf16x4 v = *(const bf16x4*)p;
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, f32x4 v) {
    bf16x4 r = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *(bf16x4*)p = r;
}

// Two adjacent 16-column MFMA output blocks of one row-per-lane result (C/D layout transposed products: lane (r, q)
// holds columns 4q..4q+3 of block 0 in v0 and of block 1 in v1, for row r): v_permlane16_swap exchanges the odd
// 16-lane rows of v0 with the even rows of v1, after which every lane owns 8 CONSECUTIVE columns -> one 16-byte store
// per lane, 64 contiguous bytes per row, instead of two 8-byte stores per lane that put 16 x 32-byte segments on the
// write path (the attention backward spent 40 % of its time in such stores).  `row_c0` = address of column 0 of block 0
// in this lane's row (16-byte aligned); all 64 lanes must call (the predicate covers the store only).
__device__ __forceinline__ void store_row32(bf16* row_c0, f32x4 v0, f32x4 v1, int q, bool pred) {
    const bf16x4 p0 = {(bf16)v0[0], (bf16)v0[1], (bf16)v0[2], (bf16)v0[3]};
    const bf16x4 p1 = {(bf16)v1[0], (bf16)v1[1], (bf16)v1[2], (bf16)v1[3]};
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t a = __builtin_bit_cast(u32x2_t, p0), b = __builtin_bit_cast(u32x2_t, p1);
    uint32_t ax = a[0], ay = a[1], bx = b[0], by = b[1];
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(ax), "+v"(bx), "+v"(ay), "+v"(by));
    const int cb = ((q & 1) << 4) | ((q & 2) << 2);          // q = 0, 1, 2, 3 -> columns 0, 16, 8, 24
    if (pred) *(u32x4*)(row_c0 + cb) = (u32x4){ax, ay, bx, by};
}

// ---- dropout RNG --------------------------------------------------------------------------------
// Counter-based: keep(element) = hash(seed, step, site, element index) >= p * 2^32.  Nothing is
// stored; backward kernels regenerate the same mask from the same (seed, step, site, index).
// `rng` points at device memory {seed, step}: the step is advanced by the optimizer kernel, so a
// captured hipGraph replays with fresh masks.
struct DropCfg {
    const uint64_t* rng;   // device: [0] = seed, [1] = step counter
    float p;               // drop probability; 0 disables
    uint32_t site;         // unique per dropout site (layer * 8 + kind)
};
struct DropKey {
    uint32_t k0, k1, thr;
    float scale;
    bool on;
};
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ DropKey make_drop_key(const DropCfg& c) {
    DropKey k;
    k.on = c.p > 0.f;
    k.k0 = k.k1 = k.thr = 0; k.scale = 1.f;
    if (k.on) {
        // {seed, step} is written by an EARLIER kernel only: read it through the constant address space, so that the
        // compiler emits one scalar s_load_dwordx4 (scalar cache, shared by the CU) instead of a vector global load
        // followed by s_waitcnt vmcnt(0) -- a full memory round trip that drains every other load of the wave and,
        // in kernels that build the key mid-way (GEMM epilogues, attention backward), sat on the critical path twice
        typedef const __attribute__((address_space(4))) uint64_t* rng_cptr;
        const rng_cptr rp = (rng_cptr)(uintptr_t)c.rng;
        const uint64_t seed = rp[0], step = rp[1];
        k.k0 = fmix32((uint32_t)seed ^ (c.site * 0x9E3779B1u));
        k.k1 = fmix32((uint32_t)(seed >> 32) + (uint32_t)step * 0x7FEB352Du + (uint32_t)(step >> 32) + c.site);
        double t = (double)c.p * 4294967296.0;
        k.thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
        k.scale = 1.f / (1.f - c.p);
    }
    return k;
}
// One hash pair decides 4 neighbouring elements (row, columns 4*cg .. 4*cg+3) with 16 bits each, so a
// kept/dropped decision costs ~4 integer ops per element instead of ~14.  Every kernel that touches a
// dropout site (forward epilogue, backward regeneration) indexes it by the same (row, column).
__device__ __forceinline__ void drop_mul4(const DropKey& k, uint32_t row, uint32_t cg, float (&m)[4]) {
    // one multiply-xorshift round per 32 bits (12 integer ops for four elements instead of 21 with two full
    // murmur finalisers: the attention kernels are VALU-issue-bound and draw 2 x S^2 masks per (sequence, head));
    // keep rate, row/column means and neighbour correlations checked against the full finaliser (|corr| < 1e-3)
    uint32_t x = row * 0x9E3779B1u + cg * 0x85EBCA77u + k.k0;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    uint32_t y = x * 0x297A2D39u + k.k1;
    y ^= y >> 15;
    const uint32_t t = k.thr >> 16;
    m[0] = (x & 0xFFFFu) >= t ? k.scale : 0.f;
    m[1] = (x >> 16) >= t ? k.scale : 0.f;
    m[2] = (y & 0xFFFFu) >= t ? k.scale : 0.f;
    m[3] = (y >> 16) >= t ? k.scale : 0.f;
}
// the same decisions as drop_mul4, as predicates (the caller folds k.scale into another factor)
__device__ __forceinline__ void drop_keep4(const DropKey& k, uint32_t row, uint32_t cg, bool (&keep)[4]) {
    uint32_t x = row * 0x9E3779B1u + cg * 0x85EBCA77u + k.k0;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    uint32_t y = x * 0x297A2D39u + k.k1;
    y ^= y >> 15;
    const uint32_t t = k.thr >> 16;
    keep[0] = (x & 0xFFFFu) >= t;
    keep[1] = (x >> 16) >= t;
    keep[2] = (y & 0xFFFFu) >= t;
    keep[3] = (y >> 16) >= t;
}
__device__ __forceinline__ float drop_mul1(const DropKey& k, uint32_t row, uint32_t col) {
    float m[4];
    drop_mul4(k, row, col >> 2, m);
    const uint32_t e = col & 3u;
    return e == 0 ? m[0] : (e == 1 ? m[1] : (e == 2 ? m[2] : m[3]));
}
enum DropSite { SITE_EMB = 0, SITE_A1 = 1, SITE_A2 = 2, SITE_AO = 3, SITE_FO = 4, SITE_NFR1 = 5, SITE_NFR2 = 6 };
__host__ __device__ inline uint32_t site_id(int layer, int kind) { return (uint32_t)(layer + 1) * 8u + (uint32_t)kind; }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    // d/dx [x Phi(x)] = Phi(x) + x phi(x)
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
// bf16 mode: erfc(|x| / sqrt 2) = 2^(-a p(a)), a = min(|x|, 5.75), p a degree-5 polynomial fitted to an absolute error of
// 2.9e-7 (tools/fit_gelu.py) -- 1000x below the bf16 rounding of the result, and branch-free: 11 VALU operations per
// element instead of erff's two divergent branches (~38).  gelu(x) = max(x, 0) - |x| / 2 * erfc(|x| / sqrt 2) keeps the
// RELATIVE accuracy of the negative tail.  The fp32 parity mode keeps erff.
__device__ __forceinline__ float erfc_abs_fast(float ax) {
    const float a = fminf(ax, 5.75f);
    float p = -1.775498822e-05f;
    p = fmaf(p, a, 6.477578427e-04f);
    p = fmaf(p, a, -7.724046707e-03f);
    p = fmaf(p, a, 5.292674527e-02f);
    p = fmaf(p, a, 4.590827227e-01f);
    p = fmaf(p, a, 1.151116848e+00f);
    return __builtin_amdgcn_exp2f(-(p * a));
}
__device__ __forceinline__ float gelu_fast(float x) { return fmaf(-0.5f * fabsf(x), erfc_abs_fast(fabsf(x)), fmaxf(x, 0.f)); }
// gelu'(x) = Phi(x) + x phi(x).  gelu' - 1/2 is odd: t P(t^2) on t = clamp(x, -4, 4) / 4, eight coefficients (tools/fit_gelu.py, grad mode):
// |error| <= 2.7e-4 everywhere (the clamp included: gelu'(+-4) is within 5e-4 of its limit) -- a tenth of the bf16 rounding of the product it
// enters.  11 VALU operations and no transcendental; the erfc / exp form it replaces took 18 with two (the GELU' epilogue of the streaming
// GEMM is vector-issue-bound: 143 us per launch against 111 us for the residual epilogue with the same bytes).
__device__ __forceinline__ float gelu_fast_grad(float x) {
    const float t = __builtin_amdgcn_fmed3f(x, -4.f, 4.f) * 0.25f, t2 = t * t;
    float p = -1.763080788e+01f;
    p = fmaf(p, t2, 8.145783234e+01f);
    p = fmaf(p, t2, -1.613120728e+02f);
    p = fmaf(p, t2, 1.802627869e+02f);
    p = fmaf(p, t2, -1.259512939e+02f);
    p = fmaf(p, t2, 5.725682831e+01f);
    p = fmaf(p, t2, -1.676991463e+01f);
    p = fmaf(p, t2, 3.186886549e+00f);
    return fmaf(p, t, 0.5f);
}
// bf16 mode: tanh(x) = 1 - 2 / (e^(2x) + 1) on the exp2 / rcp units (5 VALU operations, absolute error ~1e-7: far below the bf16
// rounding of what it feeds) instead of libm's tanhf (~40 with two divergent branches): the modality-mix kernels evaluate it on every
// element of every projected feature row and were VALU-bound on it in token mode (d = 512: 308 us forward for 0.8 GB).  The fp32
// parity mode keeps tanhf.
__device__ __forceinline__ float tanh_fast(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}
template <typename T> __device__ __forceinline__ float tanh_act(float x) {
    if constexpr (sizeof(T) == 2) return tanh_fast(x); else return tanhf(x);
}
template <typename T> __device__ __forceinline__ float gelu_fwd(float x) {
    if constexpr (sizeof(T) == 2) return gelu_fast(x); else return gelu_erf(x);
}
template <typename T> __device__ __forceinline__ float gelu_bwd(float x) {
    if constexpr (sizeof(T) == 2) return gelu_fast_grad(x); else return gelu_erf_grad(x);
}

// Sum over the 32 consecutive lanes [0, 32) / [32, 64) of a wave, result in every lane: four DPP row rotations (VALU, no
// LDS round trip as ds_bpermute would need) for the 16-lane rows, one v_permlane16_swap for the neighbouring row.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum_lanes32(float v) {
    v = dpp_add<0x121>(v);      // row_ror:1
    v = dpp_add<0x122>(v);      // row_ror:2
    v = dpp_add<0x124>(v);      // row_ror:4
    v = dpp_add<0x128>(v);      // row_ror:8
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// v_max_f32 without the canonicalising self-max clang adds in front of fmaxf for values of unknown origin
__device__ __forceinline__ float raw_max(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// max over the 32 consecutive lanes [0, 32) / [32, 64) of a wave, result in every lane (cf. sum_lanes32)
__device__ __forceinline__ float max_lanes32(float v) {
    v = raw_max(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false)));
    v = raw_max(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false)));
    v = raw_max(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false)));
    v = raw_max(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));
    float x = v, y = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return raw_max(x, y);
}

// ---- wave64 reductions ------------------------------------------------------------------------
// Result in every lane, bit-identical across the lanes (every stage pairs lane i with a partner p(i), p an involution, and both
// compute a + b): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror as DPP modifiers of the add itself, then
// v_permlane16_swap / v_permlane32_swap for the neighbouring 16-lane row and the other half.  ~12 VALU instructions and no LDS
// round trip; the ds_bpermute butterfly this replaces cost six dependent LDS latencies and ~45 instructions per value (the
// LayerNorm backward spent a quarter of its issue slots in it).
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = a + b;
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
template <int CTRL> __device__ __forceinline__ float dpp_max(float v) {
    return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false)));
}
__device__ __forceinline__ float wave_max(float v) {
    v = dpp_max<0xB1>(v);
    v = dpp_max<0x4E>(v);
    v = dpp_max<0x141>(v);
    v = dpp_max<0x140>(v);
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = fmaxf(a, b);
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

// ---- path options ---------------------------------------------------------------------------------
// Per-engine state (pmgt_engine_set_option; keys in include/pmgt_ops.h), carried into the kernel dispatchers by the `opts` field
// of their argument structs: nothing here is a process global, two engines in one process do not see each other's choices.
// 0 = the product path; a set bit takes an alternative (parity A/B of a fused form against the plain one).
enum PathOpt : uint32_t {
    OPT_TILE_GEMM = 1u << 0,            // register-staged tiled GEMMs everywhere (no streaming kernel, no LDS-DMA, no 256 x 256 tiles)
    OPT_VALU_ATTENTION = 1u << 1,       // bf16 attention on the generic fp32-VALU kernel
    OPT_WAVE_ATTENTION_BWD = 1u << 2,   // MFMA attention backward: one wave per (sequence, head) instead of cooperating waves
    OPT_NO_SHORTCUT = 1u << 3,          // last layer on every token even when last_hidden is not requested
    OPT_NO_FUSED_QKVC_ATTENTION = 1u << 4,
    OPT_NO_HEAD_MAJOR = 1u << 5,        // Q|K|V|C stays q | k | v | c between the fused forward and the backward
    OPT_NO_TABLE_PROJECTION = 1u << 6,  // feature projection per token even on small graphs
    OPT_NO_SEGMENT_SUM = 1u << 7,       // table mode keeps the per-token weight-gradient GEMM of the feature projection
    OPT_CONSUMER_QUANT = 1u << 8,       // fp8 mode: layer inputs quantised by their consumer instead of their producer
    OPT_NO_FUSED_ATTENTION_BWD = 1u << 9,
    OPT_STORE_LN_INPUT = 1u << 10,      // every LayerNorm site stores its input (no x^ from the LayerNorm output)
    OPT_EAGER_REDUCE = 1u << 11,        // partial sums reduced by a launch per producer instead of one per gradient bucket
    OPT_SIDE_STREAM_REDUCE = 1u << 12,  // ... on the engine's side stream (implies the per-producer launches)
    OPT_UNFUSED_LN = 1u << 13,          // LayerNorm as its own launch after the streaming GEMM
    OPT_ONE_BUCKET = 1u << 14,          // gradient-ready callback once per backward pass instead of per layer
    OPT_SMALL_ARENA = 1u << 15,         // test: the partial-sum arena holds ONE producer's regions, so every take flushes the previous ones
    OPT_NO_ROLE_SPLIT_LN = 1u << 16,    // streaming GEMMs stay on the 8-wave lockstep kernels (no role-split forms: K = N = 256 residual + LayerNorm, K = 512)
    OPT_NO_TILE_ATTENTION = 1u << 17,   // S = 64 / head size 64 attention: the cooperative kernels (per-wave fragment loads from global memory) instead of the tile forms
    OPT_UNFUSED_LN_BWD = 1u << 18,      // LayerNorm backward as its own launch behind the data-gradient GEMM that produces its dy
    OPT_LOCKSTEP_ATTENTION_BWD = 1u << 19,      // fused attention backward: both pairs of a step in the same phase (round 3's schedule) instead of one interval apart
    OPT_NO_CLS_ONLY_ATTENTION_BWD = 1u << 21,   // fused attention backward of the shortcut layer: the attention waves of query rows 16 .. 31 run their softmax phases for CLS-only sequences too (d ctx is zero there: identical results)
    OPT_NO_BETA_SKIP = 1u << 22,                // beta == 1: the fused kernels keep projecting / differentiating Q and K and run the (dead) dot-product branch
    OPT_NO_VC2_ATTENTION_BWD = 1u << 23,        // beta == 1: the one-head-per-step vc_only backward instead of the two-heads-per-step form (A/B)
    OPT_SIDE_STREAM_WGRAD = 1u << 20,   // the dense weight-gradient GEMMs of a layer on the engine's side stream, next to the data-gradient chain (opt-in: measured slower at every batch size, profiles/r04/NOTES.md section 9)
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace pmgt
