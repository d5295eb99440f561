// mfma_i8_rate_probe.hip -- what does v_mfma_i32_16x16x64_i8 sustain on the whole chip?  Every wave issues ITER x 32
// independent MFMAs (32 accumulators, operands in registers, nothing else in the loop); grid = CUs x waves per CU.
// Also the K = 32 form and the f16 16x16x32 form for comparison.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_i8_rate_probe tools/probes/mfma_i8_rate_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x)                                                                  \
    do {                                                                        \
        hipError_t e = (x);                                                     \
        if (e != hipSuccess) {                                                  \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                            \
        }                                                                       \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k_rate(int iters, int *out) {
    v4i acc[32];
    v4f facc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (v4i){0, 0, 0, 0}, facc[i] = (v4f){0, 0, 0, 0};
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {(int)blockIdx.x, 5, 6, 7};
    long a8 = threadIdx.x, b8 = blockIdx.x + 1;
    v8h ah, bh;
#pragma unroll
    for (int i = 0; i < 8; ++i) ah[i] = (_Float16)(float)(threadIdx.x + i), bh[i] = (_Float16)(float)(blockIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
            if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, acc[i], 0, 0, 0);
            if (KIND == 2) facc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, facc[i], 0, 0, 0);
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3] + (int)facc[i][0];
    if (s == 0x7fffffff) out[0] = s;
}

template <int KIND>
static void run(const char *name, double macs_per_mfma, int wg_per_cu, int threads) {
    int *out;
    CHK(hipMalloc(&out, 4));
    const int iters = 2000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(threads), 0, 0, iters, out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(threads), 0, 0, iters, out);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)grid * threads / 64, mfmas = waves * iters * 32;
    const double per_simd = mfmas / 1024.0;
    printf("%-28s %d WG/CU x %3d threads: %.3f ms, %.2f Pop/s (2 ops per MAC), %.1f ns per MFMA per SIMD\n", name, wg_per_cu, threads, ms,
           mfmas * macs_per_mfma * 2 / (ms * 1e-3) / 1e15, ms * 1e6 / per_simd);
    CHK(hipFree(out));
}

int main() {
    run<0>("v_mfma_i32_16x16x64_i8", 16.0 * 16 * 64, 1, 256);
    run<0>("v_mfma_i32_16x16x64_i8", 16.0 * 16 * 64, 1, 512);
    run<0>("v_mfma_i32_16x16x64_i8", 16.0 * 16 * 64, 2, 512);
    run<1>("v_mfma_i32_16x16x32_i8", 16.0 * 16 * 32, 1, 256);
    run<1>("v_mfma_i32_16x16x32_i8", 16.0 * 16 * 32, 1, 512);
    run<2>("v_mfma_f32_16x16x32_f16", 16.0 * 16 * 32, 1, 256);
    run<2>("v_mfma_f32_16x16x32_f16", 16.0 * 16 * 32, 1, 512);
    return 0;
}
