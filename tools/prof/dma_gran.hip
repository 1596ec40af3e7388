// LDS-DMA streaming rate by request granularity: every workgroup (512 threads, one per CU) streams 256-row x K bf16 row blocks
// of a [M, K] matrix into a 4-slot LDS ring, exactly as the 256x256 GEMM tiles do, with rows cut into RB-byte pieces per k-step:
//   RB = 64  : one instruction = 16 rows x 64 B   (K-step 32, what gemm_nt_big_kernel / gemm_tn_big_kernel issue)
//   RB = 128 : one instruction =  8 rows x 128 B  (K-step 64)
//   RB = 256 : one instruction =  4 rows x 256 B  (K-step 128)
// No MFMA, no LDS reads: the achievable HBM rate of the access pattern alone.  hipcc --offload-arch=gfx950 -O3 dma_gran.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

template <int RB>
__global__ __launch_bounds__(512) void stream_kernel(const char* A, int M, int Kbytes, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = 256, STAGE = ROWS * RB, NST = 4, LPR = RB / 16, RPI = 64 / LPR;   // lanes per row, rows per instruction
    constexpr int PER = ROWS / (8 * RPI);                                                  // instructions per wave per stage
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = Kbytes / RB;
    for (int tile = blockIdx.x; tile * ROWS < M; tile += gridDim.x) {
        const char* base = A + (size_t)tile * ROWS * Kbytes;
        auto issue = [&](int kt) {
            char* st = smem + (kt % NST) * STAGE;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int row = RPI * (PER * wave + j) + lane / LPR;
                const char* src = base + (size_t)row * Kbytes + (size_t)kt * RB + (lane % LPR) * 16;
                __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(st + RPI * (PER * wave + j) * RB), 16, 0, 0);
            }
        };
        for (int kt = 0; kt < NST - 1 && kt < nk; ++kt) issue(kt);
        for (int kt = 0; kt < nk; ++kt) {
            // keep NST - 1 stages in flight: wait for the oldest, barrier (the consumers would read here), issue the next
            if (kt + NST - 1 < nk) {
                if constexpr (PER == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if constexpr (PER == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (kt + NST - 1 < nk) issue(kt + NST - 1);
        }
        __builtin_amdgcn_s_barrier();
    }
    if (sink && tid == 0 && smem[0] == 123) sink[0] = 1;
}

template <int RB> static void run(const char* A, int M, int Kbytes, int* sink) {
    constexpr int smem = 4 * 256 * RB;
    hipFuncSetAttribute((const void*)stream_kernel<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(stream_kernel<RB>, dim3(256), dim3(512), smem, 0, A, M, Kbytes, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep == 2) printf("RB %3d B per row piece: %.1f us, %.2f TB/s\n", RB, ms * 1e3, (double)M * Kbytes / ms / 1e9);
    }
}
template <int RB>
__global__ __launch_bounds__(512) void stream_l2_kernel(const char* A, int reps, int Kbytes, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = 256, STAGE = ROWS * RB, NST = 4, LPR = RB / 16, RPI = 64 / LPR;
    constexpr int PER = ROWS / (8 * RPI);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = Kbytes / RB;
    for (int rep = 0; rep < reps; ++rep) {
        auto issue = [&](int kt) {
            char* st = smem + (kt % NST) * STAGE;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int row = RPI * (PER * wave + j) + lane / LPR;
                const char* src = A + (size_t)row * Kbytes + (size_t)kt * RB + (lane % LPR) * 16;
                __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(st + RPI * (PER * wave + j) * RB), 16, 0, 0);
            }
        };
        for (int kt = 0; kt < NST - 1 && kt < nk; ++kt) issue(kt);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + NST - 1 < nk) {
                if constexpr (PER == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (kt + NST - 1 < nk) issue(kt + NST - 1);
        }
        __builtin_amdgcn_s_barrier();
    }
    if (sink && tid == 0 && smem[0] == 123) sink[0] = 1;
}
template <int RB> static void run_l2(const char* A, int Kbytes, int* sink) {
    constexpr int smem = 4 * 256 * RB;
    const int reps = 6;
    hipFuncSetAttribute((const void*)stream_l2_kernel<RB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(stream_l2_kernel<RB>, dim3(256), dim3(512), smem, 0, A, reps, Kbytes, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = 256.0 * reps * 256 * Kbytes;
        if (rep == 2) printf("L2-resident source, RB %3d: %.1f us, %.2f TB/s aggregate = %.1f GB/s per CU\n", RB, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    }
}
int main() {
    const int M = 12 * 1024 * 32, Kbytes = 2048;      // the dQKVC operand of dgrad_qkvc: [393216, 1024] bf16 = 805 MB
    char* A; int* sink;
    hipMalloc(&A, (size_t)M * Kbytes); hipMemset(A, 1, (size_t)M * Kbytes); hipMalloc(&sink, 4);
    char* B; hipMalloc(&B, (size_t)M * Kbytes); hipMemset(B, 2, (size_t)M * Kbytes);     // evict the infinity cache between runs
    run<64>(A, M, Kbytes, sink);
    run<128>(B, M, Kbytes, sink);
    run<64>(B, M, Kbytes, sink);
    run<128>(A, M, Kbytes, sink);
    // the same stream from an L2-resident source (every workgroup re-reads ONE 256-row block = 512 KB, as the W operand of the
    // GEMM tiles is): the per-CU ceiling of the LDS-DMA path itself
    run_l2<64>(A, Kbytes, sink);
    run_l2<128>(A, Kbytes, sink);
    return 0;
}
