// mfma_power_probe.hip -- what does the matrix pipe sustain on THIS kernel family's operand data?
// The rate probe next to this file (mfma_i8_rate_probe.hip) feeds constant, small operands: 7.3-7.9 ns per v_mfma_i32_16x16x64_i8 per SIMD.
// MI355X lowers its clock under load (MI355X_MICROARCH.md, DVFS give-back), and how far depends on how many operand bits toggle: the prefill
// matmul on all-zero operands runs 23 % faster than on random ones (same instruction stream).  This probe issues the same bare MFMA stream
// (32 independent accumulators per wave, two waves per SIMD, every CU, ~40 ms per run so the power controller has settled) on
//   trivial   constant small operands (the rate probe's)
//   planes    A = expanded 2-bit weights (uniform in {-2,-1,1,2}), B = activation digit planes: low digit uniform int8, high digit of a
//             unit-normal row scaled to 15 bits (what k_quant_rows writes); 8 operand register sets in rotation
//   f16       A = {+-1,+-2} x an f16 block scale, B = unit-normal f16 activations; 8 sets in rotation
// and prints ns per MFMA per SIMD and the implied op rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_power_probe tools/probes/mfma_power_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

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

constexpr int NSET = 8;

// ops: [set][A | B][lane 64][4 dwords]
template <int KIND>
__global__ __launch_bounds__(512) void k_power(int iters, const v4i *ops, int *out) {
    v4i acc[32];
    v4f facc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (v4i){0, 0, 0, 0}, facc[i] = (v4f){0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    v4i a[NSET], b[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s) a[s] = ops[(s * 2 + 0) * 64 + lane], b[s] = ops[(s * 2 + 1) * 64 + lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int s = i % NSET, t = (i / 4) % NSET;  // A changes every MFMA, B every four (one B tile feeds four row tiles)
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], b[t], acc[i], 0, 0, 0);
            if (KIND == 1) facc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a[s]), __builtin_bit_cast(v8h, b[t]), facc[i], 0, 0, 0);
        }
    }
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) sum += acc[i][0] + acc[i][3] + (int)facc[i][0];
    if (sum == 0x7fffffff) out[0] = sum;
}

template <int KIND>
static void run(const char *name, const std::vector<int> &host_ops, double macs_per_mfma) {
    int *out;
    v4i *ops;
    CHK(hipMalloc(&out, 4));
    CHK(hipMalloc(&ops, host_ops.size() * 4));
    CHK(hipMemcpy(ops, host_ops.data(), host_ops.size() * 4, hipMemcpyHostToDevice));
    const int iters = 60000, grid = 256, threads = 512;  // two waves per SIMD on every CU
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_power<KIND>, dim3(grid), dim3(threads), 0, 0, iters, ops, out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_power<KIND>, dim3(grid), dim3(threads), 0, 0, iters, ops, out);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)grid * threads / 64, mfmas = waves * iters * 32, per_simd = mfmas / 1024.0;
    printf("%-44s %8.2f ms  %6.2f ns per MFMA per SIMD  %6.2f Pop/s\n", name, ms, ms * 1e6 / per_simd, mfmas * macs_per_mfma * 2 / (ms * 1e-3) / 1e15);
    fflush(stdout);
    CHK(hipFree(out));
    CHK(hipFree(ops));
}

static uint32_t pack4(int a, int b, int c, int d) { return (uint32_t)(a & 255) | ((uint32_t)(b & 255) << 8) | ((uint32_t)(c & 255) << 16) | ((uint32_t)(d & 255) << 24); }
static uint32_t pack2h(float a, float b) {
    _Float16 x = (_Float16)a, y = (_Float16)b;
    uint16_t ux, uy;
    std::memcpy(&ux, &x, 2), std::memcpy(&uy, &y, 2);
    return (uint32_t)ux | ((uint32_t)uy << 16);
}

int main() {
    std::mt19937 rng(42);
    std::normal_distribution<float> nd(0.f, 1.f);
    const int wv[4] = {-2, -1, 1, 2};
    const size_t n = (size_t)NSET * 2 * 64 * 4;
    std::vector<int> trivial(n), lo(n), hi(n), f16(n), zero(n, 0);
    for (int s = 0; s < NSET; ++s)
        for (int ab = 0; ab < 2; ++ab)
            for (int l = 0; l < 64; ++l)
                for (int d = 0; d < 4; ++d) {
                    const size_t i = (((size_t)s * 2 + ab) * 64 + l) * 4 + d;
                    trivial[i] = ab == 0 ? (d == 0 ? l : d) : 4 + d;
                    if (ab == 0) {
                        lo[i] = hi[i] = (int)pack4(wv[rng() & 3], wv[rng() & 3], wv[rng() & 3], wv[rng() & 3]);
                        f16[i] = (int)pack2h(0.0625f * wv[rng() & 3], 0.0625f * wv[rng() & 3]);
                    } else {
                        int lob[4], hib[4];
                        for (int j = 0; j < 4; ++j) {
                            // a unit-normal element of a row whose maximum is ~4: q = round(x / 4 * 2^13), balanced base-256 digits
                            int q = (int)lrintf(nd(rng) / 4.0f * 8192.0f);
                            q = q > 16383 ? 16383 : q < -16383 ? -16383 : q;
                            int d0 = ((q + 128) & 255) - 128;
                            lob[j] = d0, hib[j] = (q - d0) / 256;
                        }
                        lo[i] = (int)pack4(lob[0], lob[1], lob[2], lob[3]);
                        hi[i] = (int)pack4(hib[0], hib[1], hib[2], hib[3]);
                        f16[i] = (int)pack2h(nd(rng), nd(rng));
                    }
                }
    printf("bare MFMA streams, 32 accumulators per wave, two waves per SIMD, 256 CUs, ~40 ms each\n");
    run<0>("i8 16x16x64  all-zero operands", zero, 16.0 * 16 * 64);
    run<0>("i8 16x16x64  trivial constant operands", trivial, 16.0 * 16 * 64);
    run<0>("i8 16x16x64  weights x LOW digit plane", lo, 16.0 * 16 * 64);
    run<0>("i8 16x16x64  weights x HIGH digit plane", hi, 16.0 * 16 * 64);
    run<1>("f16 16x16x32 all-zero operands", zero, 16.0 * 16 * 32);
    run<1>("f16 16x16x32 scaled weights x normal acts", f16, 16.0 * 16 * 32);
    run<0>("i8 16x16x64  weights x LOW digit plane (again)", lo, 16.0 * 16 * 64);
    return 0;
}
