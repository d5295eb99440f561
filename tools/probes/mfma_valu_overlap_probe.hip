// mfma_valu_overlap_probe.hip -- how many VALU instructions hide in the shadow of one v_mfma_f32_16x16x32_f16 on gfx950?
// Each wave runs a loop of 32 independent MFMAs (8 accumulators x 4), each followed by K plain VALU ops (v_fma_f32 on private
// registers) or K transcendental ops (v_exp_f32), at one or two waves per SIMD.  Prints SIMD cycles per MFMA (s_memtime based
// clock estimate via wall time x 2.1 GHz is avoided: the kernel reports ns per MFMA per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_valu_overlap_probe tools/probes/mfma_valu_overlap_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x)                                                                   \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int K, int EXP>
__global__ __launch_bounds__(256, 2) void k_probe(int steps, float *out) {
    const int tid = threadIdx.x;
    v4f acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    v8h a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(tid * 0.001f + i), b[i] = (_Float16)(1.0f - i * 0.01f);
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = tid * 0.01f + i;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float &x = f[(m * K + k) & 7];
                if (EXP)
                    x = __builtin_amdgcn_exp2f(x);
                else
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(0.999f));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][3] + f[i];
    if (sum == 1234.5f) out[0] = sum;
}

// the same loop with R conflict-free ds_read_b128 per group of 4 MFMAs feeding the A operand (read two groups ahead through a
// 4-slot ring, as k_prefill_attn does), K VALU riders per MFMA, and optionally one workgroup barrier per 64 MFMAs
template <int R, int K, int BAR>
__global__ __launch_bounds__(256, 2) void k_probe_lds(int steps, float *out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    for (int i = tid; i < 8192; i += 256) reinterpret_cast<unsigned *>(lds)[i] = 0x3c003c00u;
    __syncthreads();
    v4f acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    v8h b;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)(1.0f - i * 0.01f);
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = tid * 0.01f + i;
    const unsigned ro = (unsigned)(c * 256 + ((g ^ c) * 16));
    v8h ring[4][4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[sl][i] = *reinterpret_cast<const v8h *>(lds + (ro ^ (unsigned)(64 * sl)) + 4096 * i);
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < R) ring[(grp + 2) & 3][i] = *reinterpret_cast<const v8h *>(lds + (ro ^ (unsigned)(64 * ((grp + 2) & 3))) + 4096 * i + 16384 * (grp & 1));
                acc[(grp & 1) * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[grp & 3][i], b, acc[(grp & 1) * 4 + i], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float &x = f[(i * K + k) & 7];
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(0.999f));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (BAR) __syncthreads();
    }
    float sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][3] + f[i];
    if (sum == 1234.5f) out[0] = sum;
}

template <int R, int K, int BAR>
static void run_lds(int wg_per_cu, float *out) {
    const int steps = 2000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe_lds<R, K, BAR>), dim3(grid), dim3(256), 0, 0, steps, out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe_lds<R, K, BAR>), dim3(grid), dim3(256), 0, 0, steps, out);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%d ds_read_b128 per 4 MFMAs, %d v_fma per MFMA, %s, %d wave(s) per SIMD: %6.2f ns per MFMA per SIMD\n", R, K, BAR ? "barrier per 32 MFMAs" : "no barrier",
           wg_per_cu, ms * 1e6 / ((double)wg_per_cu * steps * 32));
}

template <int K, int EXP>
static void run(int wg_per_cu, float *out) {
    const int steps = 2000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe<K, EXP>), dim3(grid), dim3(256), 0, 0, steps, out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe<K, EXP>), dim3(grid), dim3(256), 0, 0, steps, out);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_simd = (double)wg_per_cu * steps * 32;
    printf("%d %s per MFMA, %d wave(s) per SIMD: %6.2f ns per MFMA per SIMD\n", K, EXP ? "v_exp_f32" : "v_fma_f32", wg_per_cu, ms * 1e6 / mfma_per_simd);
}

int main() {
    float *out;
    CHK(hipMalloc(&out, 4));
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>(w, out);
        run<1, 0>(w, out);
        run<2, 0>(w, out);
        run<3, 0>(w, out);
        run<4, 0>(w, out);
        run<6, 0>(w, out);
        run<8, 0>(w, out);
        run<1, 1>(w, out);
        run<2, 1>(w, out);
        run<4, 1>(w, out);
    }
    for (int w = 1; w <= 2; ++w) {
        run_lds<0, 0, 0>(w, out);
        run_lds<1, 0, 0>(w, out);
        run_lds<2, 0, 0>(w, out);
        run_lds<4, 0, 0>(w, out);
        run_lds<2, 2, 0>(w, out);
        run_lds<4, 2, 0>(w, out);
        run_lds<2, 2, 1>(w, out);
        run_lds<4, 2, 1>(w, out);
    }
    return 0;
}
