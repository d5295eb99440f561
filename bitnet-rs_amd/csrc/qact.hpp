// qact.hpp -- "QAct": an activation vector quantised ONCE by the kernel that produces it, in the form the
// streaming I2_S GEMV (kernels_gemvq.hip) feeds straight to the matrix cores.
//
// north_star asks for "2-bit weight unpack x f16 activation dot product".  QAct is the f16-class activation of
// this library: per 16 consecutive elements one power-of-two scale and a 15-bit fixed-point value per element,
//     x_k  ~=  as[k / 16] * (256 * d1[k] + d0[k]),      d0, d1 int8 (balanced base-256 digits),
//     as = 2^(E - 13), E = exponent of the group's absolute maximum  =>  |256 d1 + d0| <= 2^14,
// i.e. every element is held to 2^-15 of its 16-group's maximum (f16 holds 2^-12 of the ELEMENT; for the dot
// products of this path the group-relative bound is what matters, and it is ~8x tighter than an f16 row with one
// shared scale would be).  The digits are int8 planes, so v_mfma_i32_16x16x64_i8 multiplies them with the expanded
// 2-bit weights and every 16-element partial sum is an EXACT integer; scales are applied once per 32 weights.
//
// Why the producer quantises: in round 1 every one of the 1,728 waves of a gate|up launch
// converted its own K range (218 VALU instructions per wave, 96 KB of f32 activation + gamma loads per workgroup
// through the CU's 64 B/clk vector-memory path) -- half of the kernel's issue slots.  A producer workgroup
// owns 16 output rows = one 16-group: it quantises them in its epilogue (~25 instructions on ONE wave).
//
// Layout in HBM, per 256 columns one 576-byte record (a consumer wave copies its K range into LDS with flat 16-byte
// loads):   [0, 256) d0 plane   [256, 512) d1 plane   [512, 576) sixteen f32 group scales
// Scale slot of 16-group t (t = k / 16 inside the record): lane group g = t / 4 of the MFMA's K = 64, MFMA m = t % 4
// of the record's four; m = 2 p + h (p = which 32-block of the lane group, h = which half of it) -> slot 4 g + 2 h + p,
// so the four scales a consumer lane needs for one record are 16 contiguous bytes.
//
// LayerNorm rides along as in round 1 (applied after the product): the producer multiplies by the consumer's gamma
// before quantising and leaves one (sum, sum of squares) pair per 16 rows (f32 sums stored as f64); the consumer adds the
// pairs up in f64.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace bitnet_hip {

constexpr int kQRec = 576;  // bytes per 256-column record

__host__ __device__ inline size_t qact_bytes(size_t cols) { return ((cols + 255) / 256) * (size_t)kQRec; }

// ---- reductions inside the 16-lane DPP rows (lanes 16 t .. 16 t + 15 of a wave) ----------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t qdpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
// maximum of NON-NEGATIVE floats over the row of 16 (bit patterns order like unsigned integers)
__device__ __forceinline__ float row16_max_abs(float v) {
    uint32_t u = __float_as_uint(v) & 0x7fffffffu, o;
    o = qdpp_u<0xB1>(u), u = o > u ? o : u;   // quad_perm [1,0,3,2]
    o = qdpp_u<0x4E>(u), u = o > u ? o : u;   // quad_perm [2,3,0,1]
    o = qdpp_u<0x141>(u), u = o > u ? o : u;  // row_half_mirror
    o = qdpp_u<0x140>(u), u = o > u ? o : u;  // row_mirror
    return __uint_as_float(u);
}
template <int CTRL>
__device__ __forceinline__ double qdpp_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = qdpp_u<CTRL>((unsigned)u), hi = qdpp_u<CTRL>((unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double row16_sum_d(double v) {
    v += qdpp_d<0xB1>(v);
    v += qdpp_d<0x4E>(v);
    v += qdpp_d<0x141>(v);
    v += qdpp_d<0x140>(v);
    return v;
}

template <int CTRL>
__device__ __forceinline__ float qdpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum_f(float v) {  // four v_add_f32 with a DPP operand
    v += qdpp_f<0xB1>(v);
    v += qdpp_f<0x4E>(v);
    v += qdpp_f<0x141>(v);
    v += qdpp_f<0x140>(v);
    return v;
}

__device__ __forceinline__ int qcvt_rpi(float x) {  // floor(x + 1/2), one instruction
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// One thread per element, the 16 threads of a DPP row = the 16 elements of group `group` (rows 16 group .. + 15 of
// the producing matrix = columns of the consumer), r = element inside the group.  v = the f32 value (what the
// reference holds), u = what the consumer multiplies its weights with (v, or v * gamma for a LayerNorm consumer).
// EVERY lane of the row must call this (DPP reductions).  stats (nullable): [group][2] f64 (sum v, sum v^2).
__device__ __forceinline__ void qact_emit(uint8_t *__restrict__ qout, double *__restrict__ stats, int group, int r, float v, float u) {
    if (stats) {
        // f32 sums over the 16 values (8 instructions on the ONE wave that runs a producer's epilogue; the f64 form was 24
        // with their DPP hazards), widened to f64 for the consumer, which adds the pairs of the whole row up in f64: the
        // statistics stay ~1e-7 accurate, two orders below the activation format itself
        const float s1 = row16_sum_f(v), s2 = row16_sum_f(v * v);
        if (r == 0) *reinterpret_cast<double2 *>(stats + 2 * (size_t)group) = double2{(double)s1, (double)s2};
    }
    const float am = row16_max_abs(u);
    int be = (int)(__float_as_uint(am) >> 23);  // biased exponent of the group maximum (sign already cleared)
    be = be < 32 ? 32 : be;                      // keeps both scales normal floats; groups below 2^-95 quantise to zero
    be = be > 254 ? 254 : be;                    // Inf / NaN input: garbage in, finite scale out
    const float sc = __uint_as_float((uint32_t)(267 - be) << 23);   // 2^(13 - E)
    const float as = __uint_as_float((uint32_t)(be - 13) << 23);    // 2^(E - 13)
    const int q = qcvt_rpi(u * sc);                                  // |q| <= 2^14
    const uint32_t t = ((uint32_t)q + 0x80u) ^ 0x80u;                // byte 0 = d0, byte 1 = d1 (balanced digits)
    const int rec = group >> 4, tp = group & 15;
    uint8_t *base = qout + (size_t)rec * kQRec;
    base[16 * tp + r] = (uint8_t)t;
    base[256 + 16 * tp + r] = (uint8_t)(t >> 8);
    if (r == 0) {
        const int g = tp >> 2, m = tp & 3;
        reinterpret_cast<float *>(base + 512)[4 * g + 2 * (m & 1) + (m >> 1)] = as;
    }
}

}  // namespace bitnet_hip
