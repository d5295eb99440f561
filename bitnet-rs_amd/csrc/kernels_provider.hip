// kernels_provider.hip -- the KernelProvider trait ops at speed, and the QuantizedLinear composite above them.
//
//   KernelProvider::matmul_i2s(a: &[i8], b: &[u8], c: &mut [f32], m, n, k)      K/lib.rs:44-52, K/cpu/fallback.rs:39-83,
//                                                                                K/cpu/x86.rs:417-517 (AVX2: madd_epi16 blocks)
//   QuantizedLinear::quantized_matmul_i2s                                        crates/bitnet-inference/src/layers/quantized_linear.rs:704-802
//       = quantize_input_i2s (:1762-1773: clamp(x, -2, 1).round() as i8)  ->  weights unpacked to RAW codes 0..3
//         (:769-776: (code - 2) + 2)  ->  provider.matmul_i2s  ->  per-output scale (:779-802, input_scale = 1)
//
// C = A_i8 . B_u8 is integer work: the tiled kernel keeps it in integers.  gfx950 has v_dot4_u32_u8 (unsigned x unsigned) and
// v_dot4_i32_i8 (signed x signed) but no mixed form, so A goes in biased: (a + 128) is a byte XOR, and
//     sum_l a_l b_l  =  sum_l (a_l + 128) b_l  -  128 * sum_l b_l
// with both sums exact in 32 bits (k <= 65,536: 255 * 255 * k < 2^32).  B is [k, n] row-major (the trait's layout), i.e. the
// FOUR k a dot4 wants are four different rows: a thread takes 4 consecutive columns, loads 4 rows as dwords (coalesced along n)
// and transposes the 4 x 4 bytes with v_perm_b32.  The f32 result is the exact integer rounded ONCE: identical to every
// reference implementation while their own f32 partial sums are exact (< 2^24 -- always for the I2_S value ranges this op
// carries, |a| <= 2, b <= 3); beyond that the reference's scalar, AVX2 and AVX-512 forms differ among themselves.
// BITNET_HIP_KERNEL_EXACT keeps the one-thread-per-output kernel in the scalar reference's summation order.
#include "common.hpp"

namespace bitnet_hip {

namespace {

constexpr int kMR = 8;  // rows of A a thread accumulates at once

__device__ __forceinline__ void transpose4x4(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3) {
    // rows r_i = bytes (col 0..3) of k + i  ->  c_j = bytes (k .. k + 3) of col j
    const uint32_t lo01 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), hi01 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
    const uint32_t lo23 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), hi23 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
    c0 = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
    c1 = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
    c2 = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
    c3 = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
}

// grid (ceil(n / 4 / 64), ceil(m / kMR), K parts); requires n % 4 == 0, k % 4 == 0, 4-byte aligned a rows (k % 4 == 0) and b.
// One thread = 4 adjacent output columns x kMR rows over one K part [z kc, z kc + kc): with a single K part it stores f32,
// otherwise it adds its exact int32 partial sums into c (zeroed beforehand, read as int32) and k_i32_to_f32 converts in
// place -- integer addition commutes, so the result does not depend on the order the parts arrive in.  (One K part only
// would leave a 1 x 2560 x 2560 product on 10 waves.)
template <bool SPLIT>
__global__ __launch_bounds__(64) void k_matmul_i2s_tiled(const int8_t *__restrict__ a, const uint8_t *__restrict__ b, float *__restrict__ c, int m,
                                                         int n, int k, int kc) {
    const int j4 = blockIdx.x * 64 + threadIdx.x;  // column group
    if (4 * j4 >= n) return;
    const int i0 = blockIdx.y * kMR;
    const int k0 = SPLIT ? blockIdx.z * kc : 0, k1 = SPLIT ? (k0 + kc < k ? k0 + kc : k) : k;
    const uint32_t *b32 = reinterpret_cast<const uint32_t *>(b) + j4;
    const size_t ldb = (size_t)n / 4;
    uint32_t acc[kMR][4] = {};
    uint32_t bs[4] = {0, 0, 0, 0};  // column sums of b
    for (int l = k0; l < k1; l += 4) {
        const uint32_t r0 = b32[(size_t)l * ldb], r1 = b32[(size_t)(l + 1) * ldb], r2 = b32[(size_t)(l + 2) * ldb], r3 = b32[(size_t)(l + 3) * ldb];
        uint32_t col[4];
        transpose4x4(r0, r1, r2, r3, col[0], col[1], col[2], col[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) bs[j] = __builtin_amdgcn_udot4(col[j], 0x01010101u, bs[j], false);
#pragma unroll
        for (int i = 0; i < kMR; ++i) {
            const int row = i0 + i < m ? i0 + i : m - 1;  // clamped: surplus rows recompute the last one, never stored
            const uint32_t av = *reinterpret_cast<const uint32_t *>(a + (size_t)row * k + l) ^ 0x80808080u;  // a + 128 per byte
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_udot4(av, col[j], acc[i][j], false);
        }
    }
#pragma unroll
    for (int i = 0; i < kMR; ++i) {
        if (i0 + i >= m) break;
        float *dst = c + (size_t)(i0 + i) * n + 4 * j4;
        if (SPLIT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(reinterpret_cast<int *>(dst) + j, (int)(acc[i][j] - 128u * bs[j]));
        } else {
            float4 o;
            o.x = (float)(int32_t)(acc[i][0] - 128u * bs[0]);
            o.y = (float)(int32_t)(acc[i][1] - 128u * bs[1]);
            o.z = (float)(int32_t)(acc[i][2] - 128u * bs[2]);
            o.w = (float)(int32_t)(acc[i][3] - 128u * bs[3]);
            *reinterpret_cast<float4 *>(dst) = o;
        }
    }
}
__global__ void k_i32_to_f32(float *__restrict__ c, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) c[i] = (float)reinterpret_cast<const int *>(c)[i];
}

// quantize_input_i2s (quantized_linear.rs:1762-1773): clamp(x, -2, 1).round() as i8 -- f32::round is half away from zero,
// NaN survives the clamp and `as i8` turns it into 0
__global__ void k_quant_input_i2s(const float *__restrict__ x, int8_t *__restrict__ q, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    v = v < -2.0f ? -2.0f : (v > 1.0f ? 1.0f : v);
    q[i] = v != v ? (int8_t)0 : (int8_t)roundf(v);
}

// prepare_quantized_weights_i2s (:769-776): 2-bit fields, LSB first, as raw codes 0..3 ((code - 2) + 2)
__global__ void k_unpack_codes_u8(const uint8_t *__restrict__ packed, uint8_t *__restrict__ out, size_t numel) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one packed byte
    if (4 * i >= numel) return;
    const uint8_t bv = packed[i];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (4 * i + s < numel) out[4 * i + s] = (bv >> (2 * s)) & 3u;
}

// apply_quantization_scales (:779-802): scale index = col when there is one scale per output feature, else
// min((col * in_features) / block_size, n_scales - 1); input_scale = 1.0
__global__ void k_apply_scales(float *__restrict__ out, size_t m, size_t n, const float *__restrict__ scales, size_t n_scales, size_t in_features,
                               size_t block_size) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * n) return;
    const size_t col = i % n;
    size_t idx = col;
    if (n_scales != n) {
        idx = (col * in_features) / block_size;
        idx = idx < n_scales - 1 ? idx : n_scales - 1;
    }
    const float s = n_scales ? scales[idx] : 1.0f;  // scales.get(idx).unwrap_or(1.0)
    out[i] *= 1.0f * s;
}

// KernelProvider::quantize I2S (K/cpu/fallback.rs:102-159) with one thread per OUTPUT BYTE (4 elements): coalesced float4
// loads, the 32-element block's maximum over its 8 threads by DPP-free shuffles, byte OR-pack as the reference does
__global__ __launch_bounds__(256) void k_quantize_i2s_fast(const float *__restrict__ in, size_t n, uint8_t *__restrict__ out, size_t out_len,
                                                           float *__restrict__ scales) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;  // elements 4 t .. 4 t + 3; n % 32 == 0 on this path
    float4 v = {0.f, 0.f, 0.f, 0.f};
    const bool in_range = 4 * t < n;
    if (in_range) v = *reinterpret_cast<const float4 *>(in + 4 * t);
    float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));  // fmaxf ignores NaN like f32::max
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    mx = fmaxf(mx, __shfl_xor(mx, 4));
    if (!in_range) return;
    const float scale = mx > 1e-8f ? mx / 1.5f : 1.0f;
    if ((t & 7) == 0) scales[t >> 3] = scale;
    const float e[4] = {v.x / scale, v.y / scale, v.z / scale, v.w / scale};
    uint8_t q = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) q |= (uint8_t)((e[s] > 0.5f ? 1 : (e[s] < -0.5f ? 3 : 0)) << (2 * s));
    if (t < out_len) out[t] |= q;
}

}  // namespace

hipError_t launch_matmul_i2s_tiled(const int8_t *a, const uint8_t *b, float *c, size_t m, size_t n, size_t k, hipStream_t stream) {
    if (m == 0 || n == 0) return hipSuccess;
    if (n % 4 != 0 || k % 4 != 0 || k == 0 || k > 65536 || ((uintptr_t)a & 3) || ((uintptr_t)b & 3) || ((uintptr_t)c & 15)) return hipErrorInvalidValue;
    const size_t gx = div_ceil(n / 4, 64), gy = div_ceil(m, kMR);
    // enough K parts for ~4 waves per SIMD-quarter of the chip, each at least 64 deep
    size_t parts = div_ceil((size_t)2048, gx * gy);
    parts = parts > k / 64 ? k / 64 : parts;
    if (parts <= 1) {
        hipLaunchKernelGGL(k_matmul_i2s_tiled<false>, dim3((unsigned)gx, (unsigned)gy), dim3(64), 0, stream, a, b, c, (int)m, (int)n, (int)k, (int)k);
        return hipGetLastError();
    }
    const size_t kc = div_ceil(div_ceil(k, parts), 4) * 4;
    parts = div_ceil(k, kc);
    hipError_t e = hipMemsetAsync(c, 0, m * n * sizeof(float), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_matmul_i2s_tiled<true>, dim3((unsigned)gx, (unsigned)gy, (unsigned)parts), dim3(64), 0, stream, a, b, c, (int)m, (int)n, (int)k, (int)kc);
    hipLaunchKernelGGL(k_i32_to_f32, dim3((unsigned)div_ceil(m * n, 256)), dim3(256), 0, stream, c, m * n);
    return hipGetLastError();
}
bool matmul_i2s_tiled_ok(size_t m, size_t n, size_t k) { return m > 0 && n % 4 == 0 && k % 4 == 0 && k > 0 && k <= 65536; }

hipError_t launch_quant_input_i2s(const float *x, int8_t *q, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_quant_input_i2s, dim3((unsigned)div_ceil(n, 256)), dim3(256), 0, stream, x, q, n);
    return hipGetLastError();
}
hipError_t launch_unpack_codes_u8(const uint8_t *packed, uint8_t *out, size_t numel, hipStream_t stream) {
    if (numel == 0) return hipSuccess;
    hipLaunchKernelGGL(k_unpack_codes_u8, dim3((unsigned)div_ceil(div_ceil(numel, 4), 256)), dim3(256), 0, stream, packed, out, numel);
    return hipGetLastError();
}
hipError_t launch_apply_scales(float *out, size_t m, size_t n, const float *scales, size_t n_scales, size_t in_features, size_t block_size,
                               hipStream_t stream) {
    if (m * n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_apply_scales, dim3((unsigned)div_ceil(m * n, 256)), dim3(256), 0, stream, out, m, n, scales, n_scales, in_features, block_size);
    return hipGetLastError();
}
hipError_t launch_quantize_i2s_fast(const float *in, size_t n, uint8_t *out, size_t out_len, float *scales, hipStream_t stream) {
    if (n == 0 || n % 32 != 0 || ((uintptr_t)in & 15)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_quantize_i2s_fast, dim3((unsigned)div_ceil(n / 4, 256)), dim3(256), 0, stream, in, n, out, out_len, scales);
    return hipGetLastError();
}

}  // namespace bitnet_hip
