// Issue cost of 32-bit integer multiplies on gfx950 (the dropout hash draws two per four elements): eight independent chains per lane,
// one wave per SIMD and four waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/prof/imul_rate.hip -o /tmp/imul_rate && /tmp/imul_rate
// Measured (counter units per chain step and wave, same at 1 and 4 waves per SIMD): v_mul_lo_u32 9.0 - 9.2, 24-bit multiply-add 8.5 - 9.2, shift + xor + add
// 11.5, v_perm + add 12.5: a 32-bit multiply costs about two simple integer operations, and the 24-bit form is no cheaper -- the two multiplies of the
// dropout hash (common.h: drop_keep4) cannot be replaced by anything shorter that still mixes 32 bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void k(uint32_t* out, int iters, unsigned long long* cyc) {
    uint32_t x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 2654435761u + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) x[i] = x[i] * 0x2C1B3C6Du;                       // v_mul_lo_u32
            else if (MODE == 1) x[i] = __umul24(x[i], 0x3C6Du) + 0x9E37u;   // v_mad_u32_u24 / v_mul_u32_u24
            else if (MODE == 2) x[i] = (x[i] ^ (x[i] >> 15)) + 0x9E3779B1u; // shift-xor + add
            else x[i] = __builtin_amdgcn_perm(x[i], x[i] + 0x9E3779B1u, 0x02010003u);   // v_perm_b32 byte shuffle + add
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 4096;
    const char* names[4] = {"v_mul_lo_u32", "24-bit mul(+add)", "xor-shift + add (2 ops)", "v_perm + add (2 ops)"};
    for (int wpsimd : {1, 4}) {
        for (int m = 0; m < 4; ++m) {
            unsigned long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                dim3 g(256), b(64 * 4 * wpsimd);
                if (m == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, iters, cyc);
                if (m == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, iters, cyc);
                if (m == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, iters, cyc);
                if (m == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, iters, cyc);
                hipDeviceSynchronize();
                hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            }
            printf("%d wave(s) per SIMD  %-26s %6.2f cycles per chain step per wave (8 independent chains)\n", wpsimd, names[m], (double)h / iters / 8.0);
        }
    }
    return 0;
}
