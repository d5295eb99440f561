// kernels_valu.hip -- wave-per-row streaming GEMV on the vector ALUs (gfx950).
//
// The shape BASELINE.json's north_star sketches: bit-packed weight rows streamed
// with coalesced reads (lane l takes dword l, l+64, ... of the row: 256 B
// contiguous per wave instruction), activations staged once per workgroup in LDS,
// 2-bit codes expanded in registers, f32 FMA, wave-64 shuffle reduction.
//
// Kept as the baseline the MFMA kernel is measured against (the VALU
// budget of this formulation is ~2.75 vector ops per weight, which puts it above
// the HBM roofline time on MI355X; see profiles/).
#include "common.hpp"

namespace bitnet_hip {

// Expands the 16 codes of one dword against 16 activations.
// t_i = (w >> 2i) & 0x03030303 puts code (4b+i) in byte b; v_perm_b32 then maps
// the four code bytes through the 4-entry int8 LUT in one instruction.
__device__ __forceinline__ float dot16(uint32_t w, const float *xs, uint32_t lut) {
    const float4 *xp = reinterpret_cast<const float4 *>(xs);
    float4 x0 = xp[0], x1 = xp[1], x2 = xp[2], x3 = xp[3];
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t t = (w >> (2 * i)) & 0x03030303u;
        uint32_t p = __builtin_amdgcn_perm(0u, lut, t);
        float xi0 = i == 0 ? x0.x : i == 1 ? x0.y : i == 2 ? x0.z : x0.w;
        float xi1 = i == 0 ? x1.x : i == 1 ? x1.y : i == 2 ? x1.z : x1.w;
        float xi2 = i == 0 ? x2.x : i == 1 ? x2.y : i == 2 ? x2.z : x2.w;
        float xi3 = i == 0 ? x3.x : i == 1 ? x3.y : i == 2 ? x3.z : x3.w;
        acc += (float)(int8_t)(p & 0xff) * xi0;
        acc += (float)(int8_t)((p >> 8) & 0xff) * xi1;
        acc += (float)(int8_t)((p >> 16) & 0xff) * xi2;
        acc += (float)(int8_t)(p >> 24) * xi3;
    }
    return acc;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool HAS_SCALE>
__global__ __launch_bounds__(256) void k_gemv_valu(const uint32_t *__restrict__ codes, int stride_dw,
                                                   const float *__restrict__ scales, int nblk,
                                                   int bs_dw, uint32_t lut,
                                                   const float *__restrict__ x,
                                                   float *__restrict__ y, int rows, int cols,
                                                   int kpad) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    for (int i = threadIdx.x; i < kpad; i += 256) xs[i] = i < cols ? x[i] : 0.0f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const uint32_t *wr = codes + (size_t)row * stride_dw;
        float acc = 0.0f;
        for (int d = lane; d < stride_dw; d += 64) {
            float part = dot16(wr[d], xs + d * 16, lut);
            if (HAS_SCALE) part *= scales[(size_t)row * nblk + d / bs_dw];
            acc += part;
        }
        acc = wave_sum(acc);
        if (lane == 0) y[row] = acc;
    }
}

bool valu_supported(const Weights &w) {
    if (w.row_stride_bytes % 4 != 0) return false;
    if (w.row_stride_bytes * 4 > 16384) return false;  // LDS staging: 64 KiB of f32 at most
    if (w.scaled && (w.block_size % 16 != 0 || w.cols % 16 != 0)) return false;
    return w.rows > 0 && w.cols > 0;
}

hipError_t launch_gemv_valu(const Weights &w, const float *x, float *y, size_t m, hipStream_t stream) {
    const int stride_dw = (int)(w.row_stride_bytes / 4);
    const int kpad = stride_dw * 16;
    const size_t lds = (size_t)kpad * sizeof(float);
    unsigned grid = (unsigned)div_ceil(w.rows, 4);
    if (grid > 2048) grid = 2048;
    for (size_t mi = 0; mi < m; ++mi) {
        const float *xr = x + mi * w.cols;
        float *yr = y + mi * w.rows;
        if (w.scaled)
            hipLaunchKernelGGL(k_gemv_valu<true>, dim3(grid), dim3(256), lds, stream,
                               (const uint32_t *)w.codes, stride_dw, w.scales, (int)w.nblk,
                               (int)(w.block_size / 16), w.lut, xr, yr, (int)w.rows, (int)w.cols, kpad);
        else
            hipLaunchKernelGGL(k_gemv_valu<false>, dim3(grid), dim3(256), lds, stream,
                               (const uint32_t *)w.codes, stride_dw, nullptr, 0, 1, w.lut, xr, yr,
                               (int)w.rows, (int)w.cols, kpad);
    }
    return hipGetLastError();
}

}  // namespace bitnet_hip
