#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float p16(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float p32(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__global__ void k(float* out) {
    float v = (float)(1 << (threadIdx.x >> 4)) + 0.001f * (threadIdx.x & 15);
    float a = p16(v);
    float b = p32(v);
    float c = p32(p16(v));
    float s16 = v + __shfl_xor(v, 16, 64);
    float s32 = v + __shfl_xor(v, 32, 64);
    float sb = s16 + __shfl_xor(s16, 32, 64);
    out[threadIdx.x * 6 + 0] = a; out[threadIdx.x * 6 + 1] = s16;
    out[threadIdx.x * 6 + 2] = b; out[threadIdx.x * 6 + 3] = s32;
    out[threadIdx.x * 6 + 4] = c; out[threadIdx.x * 6 + 5] = sb;
}
int main() {
    float* d; hipMalloc(&d, 64 * 6 * 4);
    k<<<1, 64>>>(d);
    float h[64 * 6]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 8) printf("lane %2d: p16 %.3f shfl16 %.3f | p32 %.3f shfl32 %.3f | both %.3f %.3f\n", l, h[l*6], h[l*6+1], h[l*6+2], h[l*6+3], h[l*6+4], h[l*6+5]);
    return 0;
}
