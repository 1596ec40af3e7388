// OCP e4m3 (gfx950: e4m3fn, NOT the MI300 fnuz encoding) pieces of the fp8 mode: quantisation kernels and the fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales)
// GEMM of the feature projection / Q|K|V|C projection (config C5 of BASELINE.json: "fp8 (e4m3) feature-projection + QKV
// MFMA path").  Everything else of the fp8 mode is the bf16 engine.
//
// Quantisation contract (restated bit for bit by oracle/pmgt_oracle.py::fake_quant_rows):
//     amax = max |x_row| ;  scale = amax / 448 (1 if amax == 0) ;  inv = 448 / amax ;  q = e4m3_rne(x * inv)
// all in IEEE fp32; the dequantised value is q * scale.
#pragma once
#include "common.h"

namespace pmgt {

constexpr float E4M3_MAX = 448.f;

// 8 fp32 -> 8 e4m3 bytes (round to nearest even, inputs already inside +-448)
__device__ __forceinline__ u32x2 pack8_e4m3(const float (&v)[8]) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    return (u32x2){(uint32_t)lo, (uint32_t)hi};
}
__device__ __forceinline__ void unpack8_e4m3(u32x2 p, float (&v)[8]) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)p[0], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)p[0], true);
    f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8((int)p[1], false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)p[1], true);
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1]; v[4] = c[0]; v[5] = c[1]; v[6] = d[0]; v[7] = d[1];
}

// C[M,N] (bf16) = (A8[M,K] * B8[N,K]^T) * sa[m] * sb[n] + bias[n];  A8 / B8 are e4m3 with the reduction index contiguous.
struct GemmF8 {
    const void* A = nullptr; int64_t lda = 0;      // bytes == elements
    const int64_t* a_rows = nullptr;               // optional row gather on A (feature table rows by node id)
    const float* a_row_scale = nullptr;            // [M] dequantisation scale per logical row, or
    float a_scale = 1.f;                           // one scale for the whole operand (feature tables)
    const void* B = nullptr; int64_t ldb = 0;
    const float* b_row_scale = nullptr;            // [N] per output channel (nullptr = 1)
    void* C = nullptr; int64_t ldc = 0;            // bf16
    int M = 0, N = 0, K = 0;
    const float* bias = nullptr;
    const int* m_dev = nullptr;
    uint32_t opts = 0;                             // PathOpt bits (OPT_TILE_GEMM: 128 x 128 tile only)
};
int gemm_nt_f8(const GemmF8& g, hipStream_t st);

// Per-row absmax quantisation of rows x cols (cols % 8 == 0) from fp32 or bf16.
template <typename T>
int quant_rows_e4m3(const T* src, int64_t lds, int rows, int cols, void* dst, int64_t ldd, float* scale, hipStream_t st);
// Several fp32 matrices of the flat parameter buffer in one launch (the fp8 weight mirror of a step).
struct QuantDesc {
    int64_t src;        // offset into the fp32 parameter buffer
    int64_t dst;        // byte offset into the e4m3 mirror
    int64_t scale;      // offset into the scale buffer
    int rows, cols;
    int row_start;      // first global row index of this matrix in the launch
};
int quant_params_e4m3(const float* params, const QuantDesc* desc_dev, int n_desc, int total_rows, void* dst, float* scale,
                      hipStream_t st);
// Whole tensor with one given inverse scale (frozen feature tables, quantised once): dst = e4m3(clamp(src * inv_scale)).
int quant_tensor_e4m3(const float* src, void* dst, int64_t n, float inv_scale, hipStream_t st);
int dequant_tensor_e4m3(const void* src, float* dst, int64_t n, float scale, hipStream_t st);

}  // namespace pmgt
