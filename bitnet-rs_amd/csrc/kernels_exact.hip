// kernels_exact.hip -- reference-order kernels (gfx950).
//
// One thread per output element, accumulating left to right in f32 exactly as the
// reference's scalar loops do, so results are bit-identical to the scalar CPU path.
// They are the fallback for shapes the streaming kernels do not take (odd k, tiny
// blocks) and the anchor that proves the streaming kernels' decode.
//
// This file MUST be compiled with -ffp-contract=off: the reference (Rust) never
// fuses `acc += a * w` into an FMA.
#include "common.hpp"

namespace bitnet_hip {

__device__ __forceinline__ float lut_value(uint32_t lut, uint32_t code) {
    return (float)(int8_t)(lut >> (8u * code));
}

// y[mi, row] = sum_j (lut[code(row,j)] [* scale(row, j / bs)]) * x[mi, j]
// Summation order: Q/i2s_qk256.rs:218-271 (no scale) and
// K/cpu/quantized_matmul.rs:74-93 (scaled: w = t * scale; acc += a * w).
__global__ void k_gemv_exact(const uint8_t *__restrict__ codes, size_t row_stride,
                             const float *__restrict__ scales, size_t nblk, size_t block_size,
                             uint32_t lut, const float *__restrict__ x, float *__restrict__ y,
                             size_t rows, size_t cols, size_t m) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * m) return;
    size_t row = gid % rows, mi = gid / rows;
    const uint8_t *wr = codes + row * row_stride;
    const float *xr = x + mi * cols;
    float acc = 0.0f;
    if (scales == nullptr) {
        for (size_t j = 0; j < cols; ++j) {
            uint32_t code = (wr[j >> 2] >> ((j & 3) * 2)) & 3u;
            float w = lut_value(lut, code);
            acc += w * xr[j];
        }
    } else {
        for (size_t blk = 0; blk < nblk; ++blk) {
            size_t j0 = blk * block_size;
            size_t j1 = j0 + block_size < cols ? j0 + block_size : cols;
            float scale = scales[row * nblk + blk];
            for (size_t j = j0; j < j1; ++j) {
                uint32_t code = (wr[j >> 2] >> ((j & 3) * 2)) & 3u;
                float w = lut_value(lut, code) * scale;
                acc += xr[j] * w;
            }
        }
    }
    y[mi * rows + row] = acc;
}

hipError_t launch_gemv_exact(const Weights &w, const float *x, float *y, size_t m, hipStream_t stream) {
    size_t total = w.rows * m;
    if (total == 0) return hipSuccess;
    unsigned block = 64;  // small blocks: spread the few threads over many CUs
    unsigned grid = (unsigned)div_ceil(total, block);
    hipLaunchKernelGGL(k_gemv_exact, dim3(grid), dim3(block), 0, stream, w.codes, w.row_stride_bytes,
                       w.scales, w.nblk, w.block_size, w.lut, x, y, w.rows, w.cols, m);
    return hipGetLastError();
}

// KernelProvider::matmul_i2s: C = A_i8 . B_u8 (K/cpu/fallback.rs:68-80), sums are
// small integers, exact in f32 in any order up to 2^24; keep the reference order.
__global__ void k_matmul_i2s_u8(const int8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                float *__restrict__ c, size_t m, size_t n, size_t k) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= m * n) return;
    size_t i = gid / n, j = gid % n;
    float sum = 0.0f;
    for (size_t l = 0; l < k; ++l) sum += (float)a[i * k + l] * (float)b[l * n + j];
    c[gid] = sum;
}

hipError_t launch_matmul_i2s_u8(const int8_t *a, const uint8_t *b, float *c, size_t m, size_t n,
                                size_t k, hipStream_t stream) {
    size_t total = m * n;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_matmul_i2s_u8, dim3((unsigned)div_ceil(total, 64)), dim3(64), 0, stream, a, b,
                       c, m, n, k);
    return hipGetLastError();
}

// KernelProvider::quantize, I2S: K/cpu/fallback.rs:126-156.  One thread per
// 32-element block; a block owns whole output bytes (32 elems = 8 bytes), so the
// OR-pack needs no atomics.  `out` holds whatever the caller put there (the
// reference ORs into it).
__global__ void k_quantize_i2s(const float *__restrict__ in, size_t n, uint8_t *__restrict__ out,
                               size_t out_len, float *__restrict__ scales, size_t num_blocks) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= num_blocks) return;
    size_t start = b * 32, end = start + 32 < n ? start + 32 : n;
    float max_val = 0.0f;
    for (size_t i = start; i < end; ++i) max_val = fmaxf(max_val, fabsf(in[i]));
    float scale = max_val > 1e-8f ? max_val / 1.5f : 1.0f;
    scales[b] = scale;
    for (size_t i = start; i < end; ++i) {
        float normalized = in[i] / scale;
        uint8_t q = normalized > 0.5f ? 1 : (normalized < -0.5f ? 3 : 0);
        size_t byte_idx = i >> 2;
        if (byte_idx < out_len) out[byte_idx] |= (uint8_t)(q << ((i & 3) * 2));
    }
}

hipError_t launch_quantize_i2s(const float *in, size_t n, uint8_t *out, size_t out_len, float *scales,
                               hipStream_t stream) {
    size_t num_blocks = div_ceil(n, 32);
    if (num_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_quantize_i2s, dim3((unsigned)div_ceil(num_blocks, 64)), dim3(64), 0, stream,
                       in, n, out, out_len, scales, num_blocks);
    return hipGetLastError();
}

__device__ __forceinline__ float f16_bits_to_f32(uint16_t h) {
    _Float16 f;
    __builtin_memcpy(&f, &h, sizeof(f));
    return (float)f;  // v_cvt_f32_f16: exact, subnormals included
}

// Block dequant with inline f16 scale: M/quant/i2s.rs:66-140 per block, walked as
// :292-346 does.  A row's blocks are packed back to back; a tail block of n < bs
// elements stores ceil(n/4) code bytes + 2 scale bytes (:305-321).  One thread per
// element.
__global__ void k_dequant_i2s(const uint8_t *__restrict__ bytes, size_t rows, size_t cols,
                              size_t block, size_t row_bytes, int inv, float k, int transposed,
                              float *__restrict__ out) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * cols) return;
    size_t r = gid / cols, c = gid % cols;
    size_t b = c / block, i = c % block;
    const uint8_t *blk = bytes + r * row_bytes + b * (block / 4 + 2);
    size_t n = cols - b * block < block ? cols - b * block : block;
    size_t qlen = (n + 3) / 4;
    uint16_t sb = (uint16_t)(blk[qlen] | (blk[qlen + 1] << 8));
    float s = fabsf(f16_bits_to_f32(sb));
    if (inv) s = s < 1e-8f ? 1.0f : 1.0f / s;
    s *= k;
    s = s < 1e-3f ? 1e-3f : (s > 1e3f ? 1e3f : s);  // f32::clamp; NaN stays NaN
    uint32_t code = (blk[i >> 2] >> ((i & 3) * 2)) & 3u;
    float v = s * lut_value(LUT_QK256, code);  // I2SMapping::Sym, M/quant/i2s.rs:46
    out[transposed ? c * rows + r : r * cols + c] = v;
}

hipError_t launch_dequant_i2s(const uint8_t *bytes, size_t rows, size_t cols, size_t block, int inv,
                              float k, int transposed, float *out, hipStream_t stream) {
    size_t total = rows * cols;
    if (total == 0) return hipSuccess;
    size_t bpr = div_ceil(cols, block);
    size_t ntail = cols - (bpr - 1) * block;
    size_t row_bytes = (bpr - 1) * (block / 4 + 2) + (ntail + 3) / 4 + 2;
    hipLaunchKernelGGL(k_dequant_i2s, dim3((unsigned)div_ceil(total, 256)), dim3(256), 0, stream, bytes,
                       rows, cols, block, row_bytes, inv, k, transposed, out);
    return hipGetLastError();
}

}  // namespace bitnet_hip
