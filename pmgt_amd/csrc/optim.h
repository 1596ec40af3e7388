// Optimizer + weight mirror (declarations); see optim.hip.
#pragma once
#include "common.h"

namespace pmgt {

struct AdamArgs {
    float* p = nullptr; const float* g = nullptr; float* m = nullptr; float* v = nullptr;
    const uint8_t* decay = nullptr;    // per element: 1 = weight decay applies
    int64_t n = 0;
    float lr = 1e-3f, wd = 1e-2f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    float max_norm = 0.f;              // <= 0: no clipping
    int64_t* step = nullptr;           // device: optimizer step counter
    float* scal = nullptr;             // device [4]: clip coef, lr/bc1, 1/sqrt(bc2), grad norm
    float* part = nullptr;             // device [1024] scratch
};
int adamw_step(const AdamArgs& a, hipStream_t st);
int advance_rng(uint64_t* rng, hipStream_t st);
int clock_probe(uint64_t* out, int blocks, hipStream_t st);      // out: [blocks][4] = {shader cycles, 100 MHz wall ticks, XCC id, 1}

struct MirrorDesc {
    int64_t src;      // offset (floats) of W[rows, cols] in the flat parameter buffer
    int rows, cols;
    int64_t dst;      // offset (elements) of the same-layout copy in the mirror, or -1
    int64_t dst_t;    // offset of the transposed copy [cols, rows], or -1
    int64_t dst_t_hm; // offset of a second transposed copy whose COLUMNS (= rows r of W = q|k|v|c x head x w) are in
                      // head-major order (head, matrix, w), or -1; hm_d / hm_dh = hidden and head size of that order
    int hm_d, hm_dh;
    int tile_start;   // first 32x32 tile index of this tensor in the launch
};
template <typename T>
int build_mirror(const float* params, T* mirror, const MirrorDesc* desc_dev, int ndesc, int total_tiles, hipStream_t st);
template <typename T> int cast_f32(const float* src, T* dst, int64_t n, hipStream_t st);
template <typename T> int cast_to_f32(const T* src, float* dst, int64_t n, hipStream_t st);

}  // namespace pmgt
