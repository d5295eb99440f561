// gemm_loop_probe.hip -- which ingredient of k_gemm_mfma's K step keeps the int8 matrix pipe half idle?  (rocprofv3 on the real kernel:
// SQ_VALU_MFMA_BUSY_CYCLES = 50.6 % of the SIMD cycles, SQ_WAIT_INST_ANY 43-48 % of the wave cycles, no LDS stall to speak of.)
// The real loop, per wave and K step: 128 v_mfma_i32_16x16x64_i8 on 32 accumulators (4 row tiles x 8 B tiles), 4 x 4 weight dwords
// expanded by 176 VALU instructions (3 shifts, 4 ands, 4 v_perm per dword), 32 ds_read_b128 of the B operand, one barrier.
// This probe rebuilds the step from nothing, one ingredient at a time (template flags), at one or two waves per SIMD:
//   F_ROT   the A / B operand registers change from MFMA to MFMA (4 A registers sets, a fresh B per group of 4)
//   F_LDS   the B operand comes from LDS (ds_read_b128 per group, swizzle-free reads of a 32 KB tile)
//   F_DEC   the A operands are expanded from packed 2-bit codes (the real decode: 11 VALU per operand, 4 operands per m step)
//   F_BAR   one s_barrier per K step
// and prints ns per MFMA per SIMD for each combination.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/gemm_loop_probe tools/probes/gemm_loop_probe.hip
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

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4i dec16(unsigned w, unsigned lut) {
    v4i a;
    a[0] = (int)__builtin_amdgcn_perm(0u, lut, w & 0x03030303u);
    a[1] = (int)__builtin_amdgcn_perm(0u, lut, (w >> 2) & 0x03030303u);
    a[2] = (int)__builtin_amdgcn_perm(0u, lut, (w >> 4) & 0x03030303u);
    a[3] = (int)__builtin_amdgcn_perm(0u, lut, (w >> 6) & 0x03030303u);
    return a;
}

template <int F_ROT, int F_LDS, int F_DEC, int F_BAR>
__global__ __launch_bounds__(256, 2) void k_loop(int steps, const v4u *__restrict__ wsrc, unsigned lut, int *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    for (int i = tid; i < 8192; i += 256) reinterpret_cast<unsigned *>(lds)[i] = (unsigned)(i * 2654435761u);
    __syncthreads();
    v4i acc[4][8];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[rt][ct] = (v4i){0, 0, 0, 0};
    v4u wv[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) wv[rt] = wsrc[(blockIdx.x * 4 + rt) * 64 + lane];
    v4i afix[4], bfix = {tid, 5, 6, 7};
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) afix[rt] = (v4i){tid + rt, 1, 2, 3};
    const unsigned char *bread = lds + c * 256 + (((4 * g) ^ ((c & 3) | (c & 8))) * 16);
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            v4i a[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                if (F_DEC) {
                    const unsigned wd = m == 0 ? wv[rt][0] : m == 1 ? wv[rt][1] : m == 2 ? wv[rt][2] : wv[rt][3];
                    a[rt] = dec16(wd + (unsigned)s, lut);
                } else if (F_ROT) {
                    a[rt] = afix[rt];
                    a[rt][0] += m;
                } else {
                    a[rt] = afix[0];
                }
            }
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                v4i b = bfix;
                if (F_LDS) b = *reinterpret_cast<const v4i *>(bread + ct * 4096 + ((m ^ (c & 3)) - (0 ^ (c & 3))) * 16);
                else if (F_ROT) b[1] += ct + m;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rt], b, acc[rt][ct], 0, 0, 0);
            }
        }
        if (F_BAR) __syncthreads();
    }
    int sum = 0;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) sum += acc[rt][ct][0] + acc[rt][ct][3];
    if (sum == 0x7fffffff) out[0] = sum;
}

template <int F_ROT, int F_LDS, int F_DEC, int F_BAR>
static void run(const char *name, int wg_per_cu, const v4u *w, int *out) {
    const int steps = 400, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto k = k_loop<F_ROT, F_LDS, F_DEC, F_BAR>;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 32768, 0, steps, w, 0xff000100u, out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 32768, 0, steps, w, 0xff000100u, out);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_simd = (double)grid * 4 * steps * 128 / 1024.0;
    printf("%-52s %d wave(s) per SIMD: %7.3f ms  %5.2f ns per MFMA per SIMD  (%.2f Pop/s)\n", name, wg_per_cu, ms, ms * 1e6 / mfma_per_simd,
           (double)grid * 4 * steps * 128 * 16 * 16 * 64 * 2 / (ms * 1e-3) / 1e15);
}

int main() {
    v4u *w;
    int *out;
    CHK(hipMalloc(&w, 512 * 4 * 64 * 16));
    CHK(hipMemset(w, 0x5a, 512 * 4 * 64 * 16));
    CHK(hipMalloc(&out, 4));
    for (int wpc = 1; wpc <= 2; ++wpc) {
        run<0, 0, 0, 0>("MFMAs alone, fixed operands", wpc, w, out);
        run<1, 0, 0, 0>("+ operands change per MFMA", wpc, w, out);
        run<1, 1, 0, 0>("+ B operand from LDS", wpc, w, out);
        run<1, 0, 1, 0>("+ A operands expanded from 2-bit codes (no LDS)", wpc, w, out);
        run<1, 1, 1, 0>("+ both", wpc, w, out);
        run<1, 1, 1, 1>("+ both + barrier per step", wpc, w, out);
    }
    return 0;
}
