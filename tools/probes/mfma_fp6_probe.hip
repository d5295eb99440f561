// mfma_fp6_probe.hip -- can the block-scaled fp6 x fp4 matrix instruction carry the prefill matmul?
//
// v_mfma_scale_f32_16x16x128_f8f6f4 with A = fp4 (e2m1: every value of an I2_S code map {-2..2} is exact) and B = fp6 (e2m3: the
// integers -16..16 over 8 are exact, i.e. a balanced base-32 digit) does K = 128 in the cycles the f16 form needs for K = 32 and
// the int8 form for K = 64.  A 15-bit activation is three base-32 digits whose weights 1, 32, 1024 ride in the instruction's E8M0
// block scale, so the three digit products accumulate into ONE f32 accumulator: 3 MFMAs per 128 columns instead of 4 (two int8
// digits x two K = 64 steps), half the accumulator registers, and far narrower multipliers (MI355X lowers its clock by switching
// activity: EXPERIMENTS 4.5).
//
// Part 1 checks the operand maps with exact integer data against a host loop:
//   lane l = (r = l & 15, g = l >> 4) holds A[row r][k = 32 g + j] / B[k = 32 g + j][col r], j = 0..31, element j at bits
//   [4j, 4j+4) of v[0:3] (fp4) or [6j, 6j+6) of v[0:5] (fp6); the lane's scale byte (op_sel picks the byte) is the E8M0 scale of
//   exactly those 32 elements; C/D as every other 16x16 form (col = l & 15, row = 4 (l >> 4) + reg).
// Part 2 times bare MFMA streams (32 independent accumulators per wave, two waves per SIMD, every CU, ~40 ms) on zeros and on
// random weights x random digits, next to the int8 and f16 forms of mfma_power_probe.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_fp6_probe tools/probes/mfma_fp6_probe.hip
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
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

// ---- host encoders ---------------------------------------------------------------------------------------------------------
// fp4 e2m1 of an integer in {-2..2} (1 = 0b0010, 2 = 0b0100), fp6 e2m3 of n / 8 for an integer n in -16..16: the subnormals and
// the first two binades are contiguous, so the magnitude code IS |n|.
static unsigned fp4_of(int w) { return (w < 0 ? 8u : 0u) | (w == 0 ? 0u : (abs(w) == 1 ? 2u : 4u)); }
static unsigned fp6_of(int n) { return (n < 0 ? 32u : 0u) | (unsigned)abs(n); }

static void put_bits(uint32_t *dst, int bit, int width, unsigned v) {
    for (int b = 0; b < width; ++b)
        if (v >> b & 1) dst[(bit + b) >> 5] |= 1u << ((bit + b) & 31);
}

// ---- part 1: layout ----------------------------------------------------------------------------------------------------------
__global__ void k_layout(const int *a8, const int *b8, const int *sa, const int *sb, float *out) {
    const int lane = threadIdx.x;
    v8i a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = a8[lane * 8 + i], b[i] = b8[lane * 8 + i];
    v4f c = {0, 0, 0, 0};
    // cbsz = 4: A is fp4; blgp = 2: B is fp6 e2m3; scale bytes: op_sel 0 = byte 0 of the lane's scale dword
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 2, 0, sa[lane], 0, sb[lane]);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
}

static int check_layout() {
    std::mt19937 rng(7);
    int A[16][128], B[128][16], ea[16][4], eb[16][4];  // integer values; block exponents per (row, 32-group) / (col, 32-group)
    for (int r = 0; r < 16; ++r)
        for (int k = 0; k < 128; ++k) A[r][k] = (int)(rng() % 5) - 2;
    for (int k = 0; k < 128; ++k)
        for (int c = 0; c < 16; ++c) B[k][c] = (int)(rng() % 33) - 16;
    for (int r = 0; r < 16; ++r)
        for (int g = 0; g < 4; ++g) ea[r][g] = (int)(rng() % 3), eb[r][g] = (int)(rng() % 11);
    std::vector<uint32_t> a8(64 * 8, 0), b8(64 * 8, 0);
    std::vector<int> sa(64), sb(64);
    for (int l = 0; l < 64; ++l) {
        const int r = l & 15, g = l >> 4;
        for (int j = 0; j < 32; ++j) {
            put_bits(&a8[l * 8], 4 * j, 4, fp4_of(A[r][32 * g + j]));
            put_bits(&b8[l * 8], 6 * j, 6, fp6_of(B[32 * g + j][r]));
        }
        sa[l] = 127 + ea[r][g];
        sb[l] = 127 + eb[r][g];
    }
    int *da, *db, *dsa, *dsb;
    float *dout;
    CHK(hipMalloc(&da, 64 * 8 * 4));
    CHK(hipMalloc(&db, 64 * 8 * 4));
    CHK(hipMalloc(&dsa, 256));
    CHK(hipMalloc(&dsb, 256));
    CHK(hipMalloc(&dout, 1024));
    CHK(hipMemcpy(da, a8.data(), 64 * 8 * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(db, b8.data(), 64 * 8 * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
    CHK(hipDeviceSynchronize());
    float out[256];
    CHK(hipMemcpy(out, dout, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const int col = l & 15, row = 4 * (l >> 4) + i;
            double want = 0;
            for (int k = 0; k < 128; ++k) want += (double)A[row][k] * B[k][col] / 8.0 * std::ldexp(1.0, ea[row][k >> 5] + eb[col][k >> 5]);
            if ((double)out[l * 4 + i] != want) {
                if (bad < 8) printf("  mismatch row %d col %d: got %.4f want %.4f\n", row, col, out[l * 4 + i], want);
                ++bad;
            }
        }
    printf("layout check (fp4 A x fp6 B, per-lane E8M0 scales): %s (%d of 256 wrong)\n", bad ? "FAILED" : "exact", bad);
    return bad;
}

// ---- part 2: rate on data ------------------------------------------------------------------------------------------------------
constexpr int NSET = 8;

// ops: [set][A | B][lane 64][8 dwords]
template <int KIND>
__global__ __launch_bounds__(512) void k_power(int iters, const v8i *ops, int *out) {
    v4i acc[32];
    v4f facc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (v4i){0, 0, 0, 0}, facc[i] = (v4f){0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    v8i a[NSET], b[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s) a[s] = ops[(s * 2 + 0) * 64 + lane], b[s] = ops[(s * 2 + 1) * 64 + lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int s = i % NSET, t = (i / 4) % NSET;  // A changes every MFMA, B every four (one B tile feeds four row tiles)
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s].lo, b[t].lo, acc[i], 0, 0, 0);
            if (KIND == 1) facc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a[s].lo), __builtin_bit_cast(v8h, b[t].lo), facc[i], 0, 0, 0);
            if (KIND == 2) facc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[s], b[t], facc[i], 4, 2, 0, 127, 0, 127 + 5 * (i % 3));
            if (KIND == 3) facc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[s], b[t], facc[i], 4, 4, 0, 127, 0, 127);
            if (KIND == 4) facc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[s], b[t], facc[i], 0, 0, 0, 127, 0, 127);
            if (KIND == 5) facc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[s], b[t], facc[i], 4, 0, 0, 127, 0, 127);
        }
    }
    float sum = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) sum += (float)(acc[i][0] + acc[i][3]) + facc[i][0];
    if (sum == 123456.789f) out[0] = (int)sum;
}

template <int KIND>
static void run(const char *name, const std::vector<uint32_t> &host_ops, double macs_per_mfma) {
    int *out;
    v8i *ops;
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
    printf("%-52s %8.2f ms  %6.2f ns per MFMA per SIMD  %6.2f Pop/s\n", name, ms, ms * 1e6 / per_simd, mfmas * macs_per_mfma * 2 / (ms * 1e-3) / 1e15);
    fflush(stdout);
    CHK(hipFree(out));
    CHK(hipFree(ops));
}

int main() {
    if (check_layout()) return 1;
    std::mt19937 rng(42);
    std::normal_distribution<float> nd(0.f, 1.f);
    const int wv[4] = {-2, -1, 1, 2};
    const size_t n = (size_t)NSET * 2 * 64 * 8;
    std::vector<uint32_t> zero(n, 0), i8(n, 0), f16(n, 0), f6(n, 0), f4(n, 0), f8(n, 0);
    for (int s = 0; s < NSET; ++s)
        for (int ab = 0; ab < 2; ++ab)
            for (int l = 0; l < 64; ++l) {
                uint32_t *pi8 = &i8[(((size_t)s * 2 + ab) * 64 + l) * 8], *pf16 = &f16[(((size_t)s * 2 + ab) * 64 + l) * 8];
                uint32_t *pf6 = &f6[(((size_t)s * 2 + ab) * 64 + l) * 8], *pf4 = &f4[(((size_t)s * 2 + ab) * 64 + l) * 8];
                uint32_t *pf8 = &f8[(((size_t)s * 2 + ab) * 64 + l) * 8];
                for (int j = 0; j < 32; ++j) {
                    if (ab == 0) {
                        const int w = wv[rng() & 3];
                        if (j < 16) put_bits(pi8, 8 * j, 8, (unsigned)w & 255);
                        if (j < 8) {
                            _Float16 h = (_Float16)(0.0625f * w);
                            uint16_t u;
                            std::memcpy(&u, &h, 2);
                            put_bits(pf16, 16 * j, 16, u);
                        }
                        put_bits(pf6, 4 * j, 4, fp4_of(w));
                        put_bits(pf4, 4 * j, 4, fp4_of(w));
                        put_bits(pf8, 4 * j, 4, fp4_of(w));
                    } else {
                        // a unit-normal element of a row whose maximum is ~4 as a 15-bit fixed-point value: low digit of each form
                        int q = (int)lrintf(nd(rng) / 4.0f * 8192.0f);
                        q = q > 16383 ? 16383 : q < -16383 ? -16383 : q;
                        const int d8 = ((q + 128) & 255) - 128, d5 = ((q + 16) & 31) - 16;
                        if (j < 16) put_bits(pi8, 8 * j, 8, (unsigned)d8 & 255);
                        if (j < 8) {
                            _Float16 h = (_Float16)nd(rng);
                            uint16_t u;
                            std::memcpy(&u, &h, 2);
                            put_bits(pf16, 16 * j, 16, u);
                        }
                        put_bits(pf6, 6 * j, 6, fp6_of(d5));
                        put_bits(pf4, 4 * j, 4, rng() & 15);
                        put_bits(pf8, 8 * j, 8, rng() & 0x77);  // finite e4m3 magnitudes
                    }
                }
            }
    printf("bare MFMA streams, 32 accumulators per wave, two waves per SIMD, 256 CUs, ~40 ms each\n");
    run<0>("i8 16x16x64    all-zero operands", zero, 16.0 * 16 * 64);
    run<0>("i8 16x16x64    weights x LOW base-256 digit", i8, 16.0 * 16 * 64);
    run<1>("f16 16x16x32   scaled weights x normal acts", f16, 16.0 * 16 * 32);
    run<2>("f8f6f4 16x16x128 fp4 x fp6, all-zero operands", zero, 16.0 * 16 * 128);
    run<2>("f8f6f4 16x16x128 fp4 weights x fp6 base-32 digit", f6, 16.0 * 16 * 128);
    run<3>("f8f6f4 16x16x128 fp4 weights x fp4 random", f4, 16.0 * 16 * 128);
    run<5>("f8f6f4 16x16x128 fp4 weights x fp8 random", f8, 16.0 * 16 * 128);
    run<4>("f8f6f4 16x16x128 fp8 x fp8, all-zero operands", zero, 16.0 * 16 * 128);
    run<0>("i8 16x16x64    weights x LOW base-256 digit (again)", i8, 16.0 * 16 * 64);
    return 0;
}
