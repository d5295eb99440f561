// kernels_gemm.hip -- I2_S / QK256 matmul for MANY activation rows (prefill) on gfx950.
//
// Same arithmetic as the batch-1 GEMV (kernels_mfma.hip): every activation row is turned
// into fixed point with one power-of-two scale per row, q = rint(x * 2^(S-E)), S = 8*NDIG-3,
// split into NDIG balanced base-256 digits; each digit plane is an int8 matrix, the 2-bit
// codes expand to int8 through the 4-entry code map (v_perm_b32), and
// v_mfma_i32_16x16x64_i8 produces exact int32 sums per digit.  The epilogue recombines
//     y[t, r] = 2^(E_t - S) * sum_d 256^d * D_d[r, t]        (only this line rounds)
// NDIG = 4 is the 30-bit form the GEMV uses; NDIG = 3 (22 bits, about f32's own mantissa)
// and NDIG = 2 (14 bits, finer than f16's 11) trade digits for matrix-core time.
//
//   k_quant_rows   one workgroup per activation row: [LayerNorm ->] max -> digits.
//                  planes[row/16][digit][row%16][Kp] int8, Kp = nblk*256 (zero padded):
//                  16 consecutive plane rows = one MFMA B tile (16 tokens of one digit),
//                  so all digits of a token land in the SAME lane of different
//                  accumulators and recombine in registers.
//   k_gemm_mfma    workgroup tile 256 weight rows x (32*TTW) tokens, 8 waves as 4 (rows) x 2
//                  (tokens); wave tile 64 rows x TTW*NDIG B tiles; K advances one 256-column
//                  QK256 block per step: the activation tile goes global -> registers -> LDS
//                  (next step's loads are in flight during this step's MFMAs), the weight
//                  tiles are the GEMV's 1-KiB lane-ordered tiles, straight into registers.
// Weights are read once per 32*TTW tokens (2 bits each: cheap); the int8 activation tile is
// the larger stream and is shared by the 4 row-waves through LDS.
//
// Later forms in this file (each explained at its definition; DESIGN.md 4.2 is the map):
//   k_gemm_f16a / k_gemm_f16h   BitNet32-F16 (and the hybrid o / down of QK256) on v_mfma_f32_16x16x32_f16: f16 rows handed from epilogue to epilogue,
//                               LayerNorm after the product (f16_chain_epilogue, ChainLnStats); f16h = 64 x 128 wave tile for the wide launches
//   k_gemm_fp6 / k_gemm_fp6w    the 2-digit product as three base-32 fp6 digits x fp4 weights on v_mfma_scale_f32_16x16x128_f8f6f4 (bit-identical to the
//                               int8 planes); resident fp4 image (k_retile_fp4); fp6w = 2 x 2 wave arrangement, LDS-DMA staging, buffer-loaded weights
//   k_quant_rows_w              the 2-digit quantiser with one wave per row (long launches of rows <= 2560 columns)
//   QB32 (k_rows_to_qb32, qb32_pack_unit, k_gemm_fp6<.., EPI = 1>)   producer-quantised block-scaled rows: no quantiser launch (opt-in)
#include <mutex>
#include <unordered_set>

#include "common.hpp"

namespace bitnet_hip {

typedef int v4i __attribute__((ext_vector_type(4)));

// LDS image of an activation column (256 bytes = sixteen 16-byte units; unit 4 g + m is what lane group g reads for MFMA m):
// the K = 32 form (WS == 3) pads the column to 272 bytes; every other form keeps 256 bytes and stores unit u of column c at
// position u ^ t(c), t(c) = (c & 3) | (c & 8) -- with that XOR the four 16-lane service groups of a ds_read_b128 (lanes
// {0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) each touch sixteen different units.  The padded stride the first
// version used (272) left every operand read 2-way conflicted (rocprofv3 SQ_LDS_BANK_CONFLICT = the reads' own cycles again).
constexpr int kColStride = 272;
__device__ __forceinline__ int col_swz(int c) { return (c & 3) | (c & 8); }

struct GemmArgs {
    const uint8_t *tiles;  // [n_tiles][nblk][64][16]
    int rows, cols, nblk;
    uint32_t lut;
    const int8_t *planes;    // [m_pad/16][NDIG][16][nblk*256]
    const float *inv_scale;  // [m_pad] 2^(E-S) per activation row (0 for padding rows)
    float *y;                // [m, rows] ([m, rows/2] with silu_mul)
    int m;
    const float *residual;  // optional [m, rows]
    const float *wscale;    // optional f32 per (row, 256-block) / (row, 32-block), row-major
    const uint16_t *stiles_h;  // WS == 3: f16 32-block scale tiles [tile][blk][kg][row][p] (k_retile_scales_h)
    int silu_mul;
    // ---- the f16 activation chain (k_gemm_f16a<.., EPI = 1>): `planes` = raw f16 rows [m_pad][cols] written by the PRODUCER of the
    // activations (inv_scale null), LayerNorm applied after the product, outputs handed on as f16 ----------------------------------
    const float *stats_in = nullptr;  // LayerNorm of the input: (sum, sum of squares) partials [n_stats][stats_stride] float2 over its columns
    int n_stats = 0, stats_stride = 0;
    float ln_eps = 0.0f;
    const float *ln_g = nullptr;       // g_r = W[r, :] . gamma (bitnet_hip_weights_bind_ln)
    _Float16 *yh = nullptr;            // f16 output rows [m_pad][rows] ([m_pad][rows / 2] with silu_mul), optionally x gamma_out[row]
    const float *gamma_out = nullptr;
    float *stats_out = nullptr;        // (sum, sum of squares) of the f32 outputs per (64-row slab, token): [rows / 64][stats_stride] float2
    unsigned long long *stamps = nullptr;  // diagnostic build (BH_STAMPS): per wave 8 x u64 of phase cycles, tools/stamp_f16a.py
    // ---- QB32 activations (round 5; k_gemm_fp6<.., EPI = 1> consumes, f16_chain_epilogue<.., QB = true> / k_rows_to_qb32 produce): `planes` = the digit
    // records [m_pad][nblk][592]: 576 bytes of digits (the fp6 form's layout) + the block's eight E8M0-ready exponent bytes (one per 32-column unit) + 8 pad ----
    uint8_t *qb_out = nullptr;         // producer: records of the OUTPUT rows [m_pad][qb_nblk_out][592] (gamma_out * y, the next matmul's K = rows)
    int qb_nblk_out = 0;
    const uint8_t *tiles4 = nullptr;  // k_gemm_fp6<.., RES = 1>: the resident fp4 image [n_tiles][nblk][m 2][64][16] (k_retile_fp4)
    int bx_off = 0;  // first row block of this launch (the wide gate|up launch is split in two: k_gemm_f16h on the first blocks, k_gemm_f16a on the rest)
    int wgroup = 0;  // > 0: an XCD walks `wgroup` weight row blocks x all token tiles before the next group (gridDim.x % wgroup == 0): gemm_weight_group
};

struct QuantArgs {
    const float *x;  // [m, cols] f32 -- or f16 when x_f16 (rows the producing kernel already rounded: attention output, silu * up)
    int x_f16 = 0;
    int m, m_pad, cols, kp;
    const float *ln_gamma;
    float ln_eps;
    int8_t *planes;
    float *inv_scale;
};

template <int CTRL>
__device__ __forceinline__ float qdpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float qwave_max(float v) {
    v = fmaxf(v, qdpp_f<0xB1>(v));
    v = fmaxf(v, qdpp_f<0x4E>(v));
    v = fmaxf(v, qdpp_f<0x141>(v));
    v = fmaxf(v, qdpp_f<0x140>(v));
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ double qwave_sum_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- activation rows -> digit planes --------------------------------------------------------
// 256 threads per row, NV float4 per thread (cols <= 1024 * NV).
// FP6 = 1 (NDIG = 2 only): the SAME 15-bit integer q = rint(x 2^(13 - E)) written as three balanced base-32 digits in fp6 (e2m3)
// for k_gemm_fp6 below -- q = d0 + 32 d1 + 1024 d2, d0, d1 in [-16, 15], |d2| <= 16; the fp6 code of n / 8 is sign | |n| (the
// subnormals and the first two binades of e2m3 are contiguous), so a digit is stored as a 6-bit sign-magnitude integer.  Row layout:
// [256-block][lane group g 4][digit 3][MFMA m 2][chunk of 8 columns 4][6 bytes] = 576 bytes per block; inside a chunk the column at
// offset o sits at k-slot 2 (o & 3) + (o >> 2) -- the order in which the weight expansion leaves its fp4 nibbles (expand16_fp4).
template <int NDIG, int NV, int FP6 = 0>
__global__ __launch_bounds__(256) void k_quant_rows(QuantArgs p) {
    static_assert(!FP6 || NDIG == 2, "the fp6 form carries the 2-digit form's integer");
    constexpr int S = 8 * NDIG - 3;
    __shared__ double stat[8];
    __shared__ float smax[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nvec = p.cols >> 2, kvec = p.kp >> 2;
    const bool live = row < p.m;
    const int rr = live ? row : p.m - 1;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + 256 * i, ci = idx < nvec ? idx : nvec - 1;
        if (p.x_f16) {  // (wave-uniform)
            typedef _Float16 qh4 __attribute__((ext_vector_type(4)));
            const qh4 hv = *reinterpret_cast<const qh4 *>(reinterpret_cast<const _Float16 *>(p.x) + (size_t)rr * p.cols + 4 * ci);
            v[i] = float4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        } else {
            v[i] = *reinterpret_cast<const float4 *>(p.x + (size_t)rr * p.cols + 4 * ci);
        }
        if (idx >= nvec || !live) v[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    if (p.ln_gamma) {
        // LayerNorm without bias, with mean subtraction (T:67-100), one pass in f64 like the GEMV prologue
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const double a = v[i].x, b = v[i].y, c = v[i].z, d = v[i].w;
            s1 += (a + b) + (c + d);
            s2 += (a * a + b * b) + (c * c + d * d);
        }
        s1 = qwave_sum_d(s1);
        s2 = qwave_sum_d(s2);
        if (lane == 0) {
            stat[2 * wave] = s1;
            stat[2 * wave + 1] = s2;
        }
        __syncthreads();
        s1 = (stat[0] + stat[2]) + (stat[4] + stat[6]);
        s2 = (stat[1] + stat[3]) + (stat[5] + stat[7]);
        const double mean_d = s1 / (double)p.cols, var_d = s2 / (double)p.cols - mean_d * mean_d;
        const float mean = (float)mean_d, denom = sqrtf((float)(var_d > 0.0 ? var_d : 0.0) + p.ln_eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < nvec && live) {
                const float4 g = *reinterpret_cast<const float4 *>(p.ln_gamma + 4 * idx);
                v[i].x = (v[i].x - mean) / denom * g.x;
                v[i].y = (v[i].y - mean) / denom * g.y;
                v[i].z = (v[i].z - mean) / denom * g.z;
                v[i].w = (v[i].w - mean) / denom * g.w;
            }
        }
    }
    float am = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) am = fmaxf(fmaxf(am, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
    am = qwave_max(am);
    if (lane == 0) smax[wave] = am;
    __syncthreads();
    am = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    int be = (int)((__float_as_uint(am) >> 23) & 0xffu);
    be = be < 32 ? 32 : be;
    const float sc = __uint_as_float((uint32_t)(254 + S - be) << 23);     // 2^(S - E)
    const float inv_s = __uint_as_float((uint32_t)(be - S) << 23);        // 2^(E - S)
    if (tid == 0) p.inv_scale[row] = live ? inv_s : 0.0f;
    if (FP6) {
        // q = rint(x 2^(S - E)) as f32 goes through LDS so that ONE lane owns the 32 columns of an MFMA operand (lane group g, MFMA m
        // of a 256-block): columns 32 u .. 32 u + 31 live in units 9 u .. 9 u + 7 of 16 bytes (the ninth unit of every 144 bytes is
        // padding: with it the sixteen lanes of a ds_read_b128 service group touch sixteen different units)
        extern __shared__ __attribute__((aligned(16))) float qrow[];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            if (idx >= kvec) continue;
            *reinterpret_cast<float4 *>(qrow + ((idx >> 3) * 9 + (idx & 7)) * 4) =
                float4{__builtin_rintf(v[i].x * sc), __builtin_rintf(v[i].y * sc), __builtin_rintf(v[i].z * sc), __builtin_rintf(v[i].w * sc)};
        }
        __syncthreads();
        // balanced base-32 digits in f32: r1 = rint(q / 32), d0 = q - 32 r1 (|d0| <= 16: a tie rounds to even, +-16 is an fp6 value),
        // likewise d1 from r1, d2 = rint(r1 / 32) (|d2| <= 16 since |q| <= 2^14).  v_cvt_scalef32_2xpk16_fp6_f32 packs 32 values / 8 into
        // 24 bytes, k-slot 2 i from its first operand's element i and 2 i + 1 from its second's (tools/probes/cvt_fp6_probe.hip): with
        // first = columns 8 q + 0 .. 3, second = columns 8 q + 4 .. 7 that IS the slot order of expand16_fp4's nibbles
        typedef float qv16f __attribute__((ext_vector_type(16)));
        typedef unsigned qv6u __attribute__((ext_vector_type(6)));
        uint8_t *rbase = reinterpret_cast<uint8_t *>(p.planes) + (size_t)row * (size_t)(p.kp >> 8) * 576;
        uint8_t *orow = reinterpret_cast<uint8_t *>(qrow) + (size_t)(p.kp >> 5) * 144;
        // one work item = one 32-column unit, ALL THREE digits (round 5: a work item per (digit, unit) computed the four digit
        // instructions of every element three times over and kept one: 1,280 wave-instructions per row against ~100 of the int8
        // split -- the fp6 quantiser took 20 / 40-46 us where the int8 one took 11-17 / 25; now 4 VALU per element, once)
        const int n_units = p.kp >> 5;
        for (int u = tid; u < n_units; u += 256) {
            qv16f lo[3], hi[3];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 f = *reinterpret_cast<const float4 *>(qrow + (u * 9 + k) * 4);
                const float xs[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r1 = __builtin_rintf(xs[e] * 0.03125f), d0 = __builtin_fmaf(r1, -32.0f, xs[e]);
                    const float r2 = __builtin_rintf(r1 * 0.03125f), d1 = __builtin_fmaf(r2, -32.0f, r1);
                    const int ix = 4 * (k >> 1) + e;
                    if (k & 1) hi[0][ix] = d0, hi[1][ix] = d1, hi[2][ix] = r2;
                    else lo[0][ix] = d0, lo[1][ix] = d1, lo[2][ix] = r2;
                }
            }
            // the packed row is assembled in LDS (behind the integers) and leaves in whole 16-byte pieces, lane after lane
            uint8_t *ob = orow + (size_t)(u >> 3) * 576 + ((u >> 1) & 3) * 144 + (u & 1) * 24;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const qv6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(lo[d], hi[d], 8.0f);
                uint2 *o = reinterpret_cast<uint2 *>(ob + d * 48);
                o[0] = uint2{r[0], r[1]}, o[1] = uint2{r[2], r[3]}, o[2] = uint2{r[4], r[5]};
            }
        }
        __syncthreads();
        for (int i = tid; i < (p.kp >> 8) * 36; i += 256) reinterpret_cast<uint4 *>(rbase)[i] = reinterpret_cast<const uint4 *>(orow)[i];
        return;
    }
    int8_t *base = p.planes + ((size_t)(row >> 4) * NDIG * 16 + (row & 15)) * p.kp;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + 256 * i;
        if (idx >= kvec) continue;
        // balanced digits: bytes of (q + B) ^ B with B = 0x80 in the NDIG-1 low bytes (kernels_mfma.hip digits4),
        // then a 4 x 4 byte transpose to one dword per digit plane (byte b = element b)
        constexpr uint32_t B = NDIG == 4 ? 0x00808080u : NDIG == 3 ? 0x00008080u : 0x00000080u;
        const uint32_t e0 = ((uint32_t)__float2int_rn(v[i].x * sc) + B) ^ B, e1 = ((uint32_t)__float2int_rn(v[i].y * sc) + B) ^ B;
        const uint32_t e2 = ((uint32_t)__float2int_rn(v[i].z * sc) + B) ^ B, e3 = ((uint32_t)__float2int_rn(v[i].w * sc) + B) ^ B;
        const uint32_t lo01 = __builtin_amdgcn_perm(e1, e0, 0x05010400u), hi01 = __builtin_amdgcn_perm(e1, e0, 0x07030602u);
        const uint32_t lo23 = __builtin_amdgcn_perm(e3, e2, 0x05010400u), hi23 = __builtin_amdgcn_perm(e3, e2, 0x07030602u);
        uint32_t dg[4];
        dg[0] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
        dg[1] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
        dg[2] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
        dg[3] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
#pragma unroll
        for (int d = 0; d < NDIG; ++d) *reinterpret_cast<uint32_t *>(base + (size_t)d * 16 * p.kp + 4 * idx) = dg[d];
    }
}

// f64 sum over the wave on the VALU: DPP steps inside a row of 16 lanes (quad_perm, row_half_mirror, row_mirror: each lane adds a partner
// that holds the other half of its group), v_permlane16_swap / v_permlane32_swap across rows -- no ds_bpermute round trips.  Every
// lane ends with the same value (a fixed order of additions).
template <int CTRL>
__device__ __forceinline__ double qdpp_d(double v) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double qswap_d(double v, bool rows32) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t l = (uint32_t)u, h = (uint32_t)(u >> 32);
    // with both operands the same register the swap leaves (own, partner) in some order in the two results: their sum is what is wanted
    const auto a = rows32 ? __builtin_amdgcn_permlane32_swap(l, l, false, false) : __builtin_amdgcn_permlane16_swap(l, l, false, false);
    const auto b = rows32 ? __builtin_amdgcn_permlane32_swap(h, h, false, false) : __builtin_amdgcn_permlane16_swap(h, h, false, false);
    return __builtin_bit_cast(double, ((uint64_t)b[0] << 32) | a[0]) + __builtin_bit_cast(double, ((uint64_t)b[1] << 32) | a[1]);
}
__device__ __forceinline__ double qwave_sum_d_valu(double v) {
    v += qdpp_d<0xB1>(v);
    v += qdpp_d<0x4E>(v);
    v += qdpp_d<0x141>(v);
    v += qdpp_d<0x140>(v);
    v = qswap_d(v, false);
    return qswap_d(v, true);
}

// ---- the 2-digit quantiser for rows of up to 2560 columns (round 5): ONE WAVE PER ROW ----------------
// k_quant_rows<2, 3, FP6> is instruction bound, not memory bound (ISA count: ~2,600 wave-instructions per 2560-column row -- an IEEE
// division per element for the LayerNorm, f64 butterflies over ds_bpermute, four workgroup barriers, the fp6 packing on 80 of 256
// threads -- = 20 us for 4096 rows where the bytes need 12).  Here a wave owns a row (ten float4 per lane, all in flight at once; no
// barrier before the row's integers exist), the LayerNorm folds into the quantiser's own multiplier:
//     q = rint(((x - mean) * gamma) * k),  k = 2^(S - E) / denom,  E = exponent of max |(x - mean) * gamma| / denom
// (one division per ROW; the int8 and the fp6 forms call the same code, so they still carry the same integer bit for bit), and the four
// rows of a workgroup pack their 4 x 80 units on all 256 threads.  Same layouts as k_quant_rows.
template <int FP6>
__global__ __launch_bounds__(256, 4) void k_quant_rows_w(QuantArgs p) {
    constexpr int S = 13, NV = 10;
    extern __shared__ __attribute__((aligned(16))) uint8_t qrow_all[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = (int)blockIdx.x * 4 + wave;
    const int nvec = p.cols >> 2, kvec = p.kp >> 2, n_units = p.kp >> 5, nblk = p.kp >> 8;
    const bool live = row < p.m;
    const int rr = live ? row : p.m - 1;
    float4 v[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = lane + 64 * i, ci = idx < nvec ? idx : nvec - 1;
        if (p.x_f16) {  // (uniform)
            typedef _Float16 qh4 __attribute__((ext_vector_type(4)));
            const qh4 hv = *reinterpret_cast<const qh4 *>(reinterpret_cast<const _Float16 *>(p.x) + (size_t)rr * p.cols + 4 * ci);
            v[i] = float4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        } else {
            v[i] = *reinterpret_cast<const float4 *>(p.x + (size_t)rr * p.cols + 4 * ci);
        }
        gm[i] = p.ln_gamma ? *reinterpret_cast<const float4 *>(p.ln_gamma + 4 * ci) : float4{1.f, 1.f, 1.f, 1.f};
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i >= nvec || !live) v[i] = float4{0.f, 0.f, 0.f, 0.f}, gm[i] = float4{0.f, 0.f, 0.f, 0.f};
    float denom = 1.0f;  // (1 / denom is never formed: denom divides the row's maximum and the multiplier, once each)
    if (p.ln_gamma) {
        // LayerNorm without bias, with mean subtraction (T:67-100): sums in f64 over the exact f32 values
        // (two independent chains each, one add / one fma per element: the f64 pipe runs at half rate and is this kernel's busiest unit)
        double s1a = 0.0, s1b = 0.0, s2a = 0.0, s2b = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const double a = v[i].x, b = v[i].y, c = v[i].z, d = v[i].w;
            s1a += a, s1b += b, s1a += c, s1b += d;
            s2a = __builtin_fma(a, a, s2a), s2b = __builtin_fma(b, b, s2b), s2a = __builtin_fma(c, c, s2a), s2b = __builtin_fma(d, d, s2b);
        }
        double s1 = qwave_sum_d_valu(s1a + s1b);
        double s2 = qwave_sum_d_valu(s2a + s2b);
        const double mean_d = s1 / (double)p.cols, var_d = s2 / (double)p.cols - mean_d * mean_d;
        const float mean = (float)mean_d;
        denom = sqrtf((float)(var_d > 0.0 ? var_d : 0.0) + p.ln_eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i].x = (v[i].x - mean) * gm[i].x;
            v[i].y = (v[i].y - mean) * gm[i].y;
            v[i].z = (v[i].z - mean) * gm[i].z;
            v[i].w = (v[i].w - mean) * gm[i].w;
        }
    }
    float am = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) am = fmaxf(fmaxf(am, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
    am = qwave_max(am) / denom;
    int be = (int)((__float_as_uint(am) >> 23) & 0xffu);
    be = be < 32 ? 32 : be;
    const float sc = __uint_as_float((uint32_t)(254 + S - be) << 23) / denom;  // 2^(S - E) / denom
    const float inv_s = __uint_as_float((uint32_t)(be - S) << 23);             // 2^(E - S)
    if (lane == 0) p.inv_scale[row] = live ? inv_s : 0.0f;
    if (!FP6) {
        int8_t *base = p.planes + ((size_t)(row >> 4) * 2 * 16 + (row & 15)) * p.kp;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = lane + 64 * i;
            if (idx >= kvec) continue;
            // balanced base-256 digits: bytes of (q + 0x80) ^ 0x80, then a 4 x 4 byte transpose to one dword per digit plane (k_quant_rows)
            constexpr uint32_t B = 0x00000080u;
            const uint32_t e0 = ((uint32_t)__float2int_rn(v[i].x * sc) + B) ^ B, e1 = ((uint32_t)__float2int_rn(v[i].y * sc) + B) ^ B;
            const uint32_t e2 = ((uint32_t)__float2int_rn(v[i].z * sc) + B) ^ B, e3 = ((uint32_t)__float2int_rn(v[i].w * sc) + B) ^ B;
            const uint32_t lo01 = __builtin_amdgcn_perm(e1, e0, 0x05010400u), lo23 = __builtin_amdgcn_perm(e3, e2, 0x05010400u);
            *reinterpret_cast<uint32_t *>(base + 4 * idx) = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
            *reinterpret_cast<uint32_t *>(base + (size_t)16 * p.kp + 4 * idx) = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
        }
        return;
    }
    // the row's integers go through LDS as int16 (|q| <= 2^14), so that ONE work item owns the 32 columns of an MFMA operand: a unit's 64
    // bytes sit in 80 (five 16-byte granules per unit: the sixteen lanes of a ds_read_b128 service group touch sixteen different granules).
    // 25.6 KB per workgroup: all 1024 workgroups of a 4096-row launch are resident at once (the f32 image + a staged copy of the packed rows
    // took 69 KB: two workgroups per CU, two rounds: 16.7 us; the packed units now leave straight from the registers)
    const int row_q = n_units * 80;  // bytes per row
    uint8_t *qall = reinterpret_cast<uint8_t *>(qrow_all), *qrow = qall + (size_t)wave * row_q;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = lane + 64 * i;
        if (idx >= kvec) continue;
        const auto a = __builtin_amdgcn_cvt_pk_i16(__float2int_rn(v[i].x * sc), __float2int_rn(v[i].y * sc));
        const auto b = __builtin_amdgcn_cvt_pk_i16(__float2int_rn(v[i].z * sc), __float2int_rn(v[i].w * sc));
        *reinterpret_cast<uint2 *>(qrow + (idx >> 3) * 80 + (idx & 7) * 8) = uint2{__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b)};
    }
    __syncthreads();
    typedef float qv16f __attribute__((ext_vector_type(16)));
    typedef unsigned qv6u __attribute__((ext_vector_type(6)));
    const size_t row_b = (size_t)nblk * 576;  // packed bytes per row
    for (int j = tid; j < 4 * n_units; j += 256) {
        const int r = j / n_units, u = j - r * n_units;
        const uint8_t *qr = qall + (size_t)r * row_q + u * 80;
        qv16f lo[3], hi[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // 16-byte granule k: columns 8 k .. 8 k + 7 of the unit
            const uint4 w = *reinterpret_cast<const uint4 *>(qr + 16 * k);
            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = (float)(int16_t)(ws[e >> 1] >> (16 * (e & 1)));
                const float r1 = __builtin_rintf(x * 0.03125f), d0 = __builtin_fmaf(r1, -32.0f, x);
                const float r2 = __builtin_rintf(r1 * 0.03125f), d1 = __builtin_fmaf(r2, -32.0f, r1);
                const int ix = 4 * k + (e & 3);  // k-slot 8 k + 2 (e & 3) + (e >> 2): first operand = columns 8 k + 0 .. 3, second = 8 k + 4 .. 7
                if (e & 4) hi[0][ix] = d0, hi[1][ix] = d1, hi[2][ix] = r2;
                else lo[0][ix] = d0, lo[1][ix] = d1, lo[2][ix] = r2;
            }
        }
        uint8_t *ob = reinterpret_cast<uint8_t *>(p.planes) + ((size_t)blockIdx.x * 4 + r) * row_b + (size_t)(u >> 3) * 576 + ((u >> 1) & 3) * 144 + (u & 1) * 24;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const qv6u c6 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(lo[d], hi[d], 8.0f);
            uint2 *o = reinterpret_cast<uint2 *>(ob + d * 48);
            o[0] = uint2{c6[0], c6[1]}, o[1] = uint2{c6[2], c6[3]}, o[2] = uint2{c6[4], c6[5]};
        }
    }
}

__device__ __forceinline__ v4i gdecode16(uint32_t w, uint32_t lut) {
    v4i a;
    a[0] = (int)__builtin_amdgcn_perm(0u, lut, w & 0x03030303u);
    a[1] = (int)__builtin_amdgcn_perm(0u, lut, (w >> 2) & 0x03030303u);
    a[2] = (int)__builtin_amdgcn_perm(0u, lut, (w >> 4) & 0x03030303u);
    a[3] = (int)__builtin_amdgcn_perm(0u, lut, (w >> 6) & 0x03030303u);
    return a;
}

// the same value from ONE conversion where the integer fits (a 32-block's digit sums are < 2^14 each): exact integer
// d0 + 256 d1 (+ 65536 d2), rounded once by the conversion -- what the f32 additions of exact terms give too
template <int NDIG>
__device__ __forceinline__ float combine_digits_i(const v4i *acc, int j) {
    if (NDIG == 2) return (float)(int)(((uint32_t)acc[1][j] << 8) + (uint32_t)acc[0][j]);
    if (NDIG == 3) return (float)(int)(((((uint32_t)acc[2][j] << 8) + (uint32_t)acc[1][j]) << 8) + (uint32_t)acc[0][j]);
    return (float)(int)(((uint32_t)acc[1][j] << 8) + (uint32_t)acc[0][j]) + (65536.0f * (float)acc[2][j] + 16777216.0f * (float)acc[3][j]);
}
__device__ __forceinline__ float gfma_mix_lo(float a, uint32_t h, float c) {  // a * f16(h[15:0]) + c
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ float gfma_mix_hi(float a, uint32_t h, float c) {  // a * f16(h[31:16]) + c
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(h), "v"(c));
    return r;
}

template <int NDIG>
__device__ __forceinline__ float combine_digits(const v4i *acc, int j) {
    // ((d0 + 256 d1) + (65536 d2 + 2^24 d3)): the GEMV's order
    if (NDIG == 2) return (float)acc[0][j] + 256.0f * (float)acc[1][j];
    if (NDIG == 3) return ((float)acc[0][j] + 256.0f * (float)acc[1][j]) + 65536.0f * (float)acc[2][j];
    return ((float)acc[0][j] + 256.0f * (float)acc[1][j]) + (65536.0f * (float)acc[2][j] + 16777216.0f * (float)acc[3][j]);
}

// WS: 0 no weight scales; 1 one f32 scale per (row, 256-block); 2 one per (row, 32-block): the four
// lane groups of an MFMA's K = 64 belong to four different 32-blocks, so B is masked to one lane group at a
// time (4x the MFMAs) and every pair of MFMAs is folded into f32 with its block's scale.
// WS == 3: 32-block scales on v_mfma_i32_16x16x32_i8 -- one MFMA = one 32-weight block, no masking (the K = 64 form has four
// different blocks in its four lane groups: WS == 2 masks B to one lane group at a time, 4x the MFMAs).  The K = 32 operand
// wants lane group g to hold columns 8 g .. 8 g + 7 of a 32-block; in the GEMV's tiles those are eight 2-bit fields of ONE
// dword of another lane (lane (r, i) for blocks 2 i and 2 i + 1): the wave passes its tiles through a private LDS area and
// reads them back in that deal -- no second copy of the codes in HBM (fetching the 8 dwords per lane straight from global
// memory instead cost 40 % of the prefill: 32 load instructions per K step).  Scales come as f16 tiles.
// MINW = 4 waves per SIMD = two workgroups per CU (<= 128 registers): the narrow-token-tile form for launches whose wide-tile grid would
// leave most CUs idle in its last round (a 2560-row matrix x 4096 tokens = 320 wide workgroups on 256 CUs).
typedef unsigned gv4u __attribute__((ext_vector_type(4)));
typedef float gv4f __attribute__((ext_vector_type(4)));

// Output rows are written once and read by the NEXT kernel: stored non-temporal, they stop evicting the activation tiles and
// weight slabs the K loops of the other workgroups are re-reading from L2 (4096 x 13824 outputs = 226 MB through 8 x 4 MB of L2).
// Measured (tools/perf_gemm_ksweep.py): the launch's time over K is a line whose intercept is the stores' -- 80 -> 59 us at 13824
// rows, 15.5 -> 10.3 at 2560 -- gate|up 299 -> 275 us, q|k|v 98 -> 93, o 78 -> 73, down 177 -> 172 (quantiser included).
__device__ __forceinline__ void store_out4(float *dst, float a, float b, float c, float d) {
    __builtin_nontemporal_store((gv4f){a, b, c, d}, reinterpret_cast<gv4f *>(dst));
}

// f16 hand-over rows are clamped to +-65504 (an f16 element carries its own exponent: no row scale).  A clamp is COUNTED, not silent: the
// decoder reads the counter behind a prompt forward and repeats the prompt on the row-scaled forms when it is not zero (Decoder::prefill).
__device__ unsigned int g_f16_saturations = 0;
__device__ __forceinline__ _Float16 gclamp_f16(float v, bool &sat) {
    sat = sat || !(fabsf(v) <= 65504.0f);  // (also true for NaN)
    return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
}
__device__ __forceinline__ void gflush_sat(bool sat) {
    if (sat) atomicAdd(&g_f16_saturations, 1u);  // (rare: one atomic per lane that clamped, none otherwise)
}
unsigned long long f16_saturations(bool reset) {
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_f16_saturations), sizeof(v)) != hipSuccess) return ~0ull;  // (synchronises the device)
    if (reset && v) {
        const unsigned int z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_f16_saturations), &z, sizeof(z));
    }
    return v;
}

// silu(gate) * up (FeedForward::forward T:756-781, silu(v) = v / (1 + exp(-v))) as the decode GEMV's epilogue computes it (kernels_gemvq.hip):
// v_exp_f32 + v_rcp_f32, 6 VALU per output.  The IEEE division and libm expf of the first version were ~22 VALU per output: 700 of the ~1400
// VALU instructions of a gate|up wave's epilogue, which no MFMA of that wave overlaps.  |error| ~ 1e-6 relative (the f16 hand-over rounds at 5e-4).
__device__ __forceinline__ float gsilu_mul(float gv, float uv) { return gv * __builtin_amdgcn_rcpf(1.0f + __expf(-gv)) * uv; }

// The epilogue's stores.  A lane (c = token of its 16-token group, g) holds val[rt][j] = output row 16 (tile0 + rt) + 4 g + j of its
// token: one row tile is 64 contiguous bytes per token, so a store instruction per row tile writes HALF lines (non-temporal, not
// merged in L2: WRITE_SIZE read 158 MB for 113 MB of gate|up outputs).  Row tiles are therefore stored in PAIRS: the odd tile's values
// move to the lane 8 tokens away (DPP row_ror:8, same g), and each of the pair's two instructions writes whole 128-byte lines --
// tokens 0-7 (lanes c < 8: their even tile; lanes c >= 8: the odd tile of token c - 8), then tokens 8-15 likewise.
__device__ __forceinline__ float gdpp_ror8(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));
}
// a / b: this lane's four values of rows ra .. ra + 3 / ra + 16 .. ra + 19 (ra includes 4 g) of token tok0 + c; ld = row length of y
__device__ __forceinline__ void store_tile_pair(float *y, const float *residual, int ld, int m, int tok0, int c, int ra, const float (&a)[4], const float (&b)[4]) {
    float rot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rot[j] = gdpp_ror8(b[j]);
    const bool lo = c < 8;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {  // tokens 0-7, then 8-15 of the group
        const bool own = lo == (hh == 0);
        const int token = tok0 + 8 * hh + (c & 7), row = own ? ra : ra + 16;
        if (token < m && row + 3 < ld) {
            const size_t off = (size_t)token * ld + row;
            float o0 = own ? a[0] : rot[0], o1 = own ? a[1] : rot[1], o2 = own ? a[2] : rot[2], o3 = own ? a[3] : rot[3];
            if (residual) {
                const float4 r = *reinterpret_cast<const float4 *>(residual + off);
                o0 += r.x, o1 += r.y, o2 += r.z, o3 += r.w;
            }
            store_out4(y + off, o0, o1, o2, o3);
        }
    }
}
// the whole epilogue of one 16-token group of a wave: val = the lane's 4 row tiles x 4 rows, already scaled; tile0 = first row tile
__device__ __forceinline__ void store_wave_tiles(const GemmArgs &p, float (&val)[4][4], int tok0, int c, int g, int tile0) {
    const int token = tok0 + c;
    if (!p.silu_mul) {
        if ((p.rows & 3) == 0) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) store_tile_pair(p.y, p.residual, p.rows, p.m, tok0, c, 16 * (tile0 + 2 * pp) + 4 * g, val[2 * pp], val[2 * pp + 1]);
        } else if (token < p.m) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const int row0 = 16 * (tile0 + rt) + 4 * g;
                const size_t off = (size_t)token * p.rows + row0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (row0 + j < p.rows) p.y[off + j] = val[rt][j] + (p.residual ? p.residual[off + j] : 0.0f);
            }
        }
    } else {
        // row tiles alternate (gate, up): FeedForward::forward T:756-781, silu(v) = v / (1 + exp(-v))
        const int half_rows = p.rows >> 1;
        float r[2][4];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gv = val[2 * pr][j], uv = val[2 * pr + 1][j];
                r[pr][j] = gsilu_mul(gv, uv);
            }
        const int ra = 16 * (tile0 >> 1) + 4 * g;  // tile0 / 2 = the wave's first tile of the [m, rows / 2] output
        if (p.yh) {  // f16 rows for the down-projection's quantiser: half the bytes written here and read there (half_rows % 4 == 0 checked by the launcher)
            typedef _Float16 sh4 __attribute__((ext_vector_type(4)));
            if (token < p.m) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    sh4 o;
                    bool sat = false;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = gclamp_f16(r[pr][j], sat);
                    gflush_sat(sat);
                    *reinterpret_cast<sh4 *>(p.yh + (size_t)token * half_rows + ra + 16 * pr) = o;  // (non-temporal: no gain, 257-275 vs 261-291 us same box)
                }
            }
        } else if ((half_rows & 3) == 0) {
            store_tile_pair(p.y, nullptr, half_rows, p.m, tok0, c, ra, r[0], r[1]);
        } else if (token < p.m) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ra + 16 * pr + j < half_rows) p.y[(size_t)token * half_rows + ra + 16 * pr + j] = r[pr][j];
        }
    }
}

// CW = wave columns: 2 (8 waves, 256 rows x 2 TTW token tiles) or 1 (4 waves, 256 rows x TTW token tiles; with MINW = 2 two such
// workgroups share a CU with independent barriers at the full register budget).
template <int NDIG, int TTW, int WS, int MINW = 1, int CW = 2>
__global__ __launch_bounds__(256 * CW, MINW) void k_gemm_mfma(GemmArgs p) {
    constexpr int CT = NDIG * TTW;          // B tiles per wave
    constexpr int WG_COLS = CW * CT * 16;           // plane rows per workgroup
    constexpr int NB = WG_COLS * 16 / (256 * CW);   // uint4 per thread per K step
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4, rw = wave & 3, cw = wave >> 2;
    const int n_tiles = (p.rows + 15) >> 4;
    const int kp = p.nblk * 256;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Renumber them so that one XCD
    // works through CONSECUTIVE logical ids: the workgroups that share an activation tile (same by, all bx)
    // then find it in ONE L2 instead of pulling it into eight.  Placement only changes speed, never results.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int gx = gridDim.x, total = gx * gridDim.y, id = by * gx + bx;
        if ((total & 7) == 0) {
            const int l = (id & 7) * (total >> 3) + (id >> 3);
            bx = l % gx;
            by = l / gx;
            if (p.wgroup > 0) {  // weight-stationary walk: groups of `wgroup` row blocks, inside a group the token tiles, inside a tile the group's blocks
                const int per = (int)gridDim.y * p.wgroup, grp = l / per, r = l - grp * per;
                by = r / p.wgroup;
                bx = grp * p.wgroup + (r - by * p.wgroup);
            }
        }
    }

    // this wave's four weight row tiles (clamped: surplus tiles recompute the last one, never stored)
    const uint8_t *wptr[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        int t = bx * 16 + rw * 4 + rt;
        t = t < n_tiles ? t : n_tiles - 1;
        wptr[rt] = p.tiles + ((size_t)t * p.nblk * 64 + lane) * 16;
    }
    // this thread's share of the activation tile
    constexpr int CSTR = WS == 3 ? kColStride : 256;
    const int8_t *bsrc[NB];
    int bdst[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + 256 * CW * i, col = idx >> 4, seg = idx & 15;
        bsrc[i] = p.planes + (size_t)(by * WG_COLS + col) * kp + seg * 16;
        bdst[i] = col * CSTR + (WS == 3 ? seg : seg ^ col_swz(col & 15)) * 16;
    }
    constexpr int kBuf = WG_COLS * CSTR;  // one activation tile in LDS; the unscaled variant keeps two
    gv4u wn[4], bn[NB];  // native vectors: arrays of HIP's uint4 struct are kept in scratch memory by this compiler
    if (!WS) {
        // prologue: activation tile 0 -> LDS buffer 0, tile 1's loads in flight; weights of step 0 in flight
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) wn[rt] = *reinterpret_cast<const gv4u *>(wptr[rt]);
#pragma unroll
        for (int i = 0; i < NB; ++i) bn[i] = *reinterpret_cast<const gv4u *>(bsrc[i]);
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<gv4u *>(lds + bdst[i]) = bn[i];
        const int n1 = p.nblk > 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) bn[i] = *reinterpret_cast<const gv4u *>(bsrc[i] + (size_t)n1 * 256);
        __syncthreads();
    }
    v4i acc[4][CT];
    float facc[WS ? 4 : 1][WS ? TTW : 1][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (v4i){0, 0, 0, 0};
    if (WS) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int tt = 0; tt < TTW; ++tt)
#pragma unroll
                for (int j = 0; j < 4; ++j) facc[rt][tt][j] = 0.0f;
    }
    const uint8_t *bread = lds + (cw * CT * 16 + c) * CSTR + (WS == 3 ? 64 * g : 0);
    // swizzled forms: byte offset of unit 4 g + m inside this lane's column, m = 0..3
    int moff[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) moff[m] = ((4 * g + m) ^ col_swz(c)) * 16;
    // 32-block scales: element offset of (this lane's output row, block 0) in the [rows, cols / 32] scale array
    int soff[WS == 2 ? 4 : 1][4];
    if (WS == 2) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            int t = bx * 16 + rw * 4 + rt;
            t = t < n_tiles ? t : n_tiles - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int row = 16 * t + 4 * g + j;
                row = row < p.rows ? row : p.rows - 1;
                soff[rt][j] = row * (p.nblk * 8);
            }
        }
    }

    // K = 32 form: this lane's f16 scale tiles, one 16-byte load per (row tile, 64 columns)
    const uint16_t *sptr[WS == 3 ? 4 : 1];
    gv4u scn[WS == 3 ? 4 : 1];
    const int sh3 = 4 * (g & 1);  // this lane's columns are the low (g even) or high nibble half of each source byte
    uint8_t *wst = lds + kBuf + wave * 4096;                                       // WS == 3: this wave's tile staging area
    const uint8_t *wrd = wst + c * 16 + 4 * (g >> 1);                              //          and where this lane reads it back
    if (WS == 3) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            int t = bx * 16 + rw * 4 + rt;
            t = t < n_tiles ? t : n_tiles - 1;
            sptr[rt] = p.stiles_h + (size_t)t * p.nblk * 128 + 8 * g;
            scn[rt] = *reinterpret_cast<const gv4u *>(sptr[rt]);
        }
    }

    for (int blk = 0; blk < p.nblk; ++blk) {
        const uint8_t *bcur = bread;
        if (WS) {  // scaled variant: registers go to the f32 accumulators: single buffer, no prefetch
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) wn[rt] = *reinterpret_cast<const gv4u *>(wptr[rt] + (size_t)blk * 1024);
#pragma unroll
            for (int i = 0; i < NB; ++i) bn[i] = *reinterpret_cast<const gv4u *>(bsrc[i] + (size_t)blk * 256);
            __syncthreads();  // the previous step's LDS reads are done
#pragma unroll
            for (int i = 0; i < NB; ++i) *reinterpret_cast<gv4u *>(lds + bdst[i]) = bn[i];
            if (WS == 3) {
                // the K = 32 operand of lane (r, g) is eight 2-bit fields of ONE dword of lane (r, i)'s 16 bytes (blocks 2 i,
                // 2 i + 1): the wave's four 1-KiB tiles pass through a wave-private LDS area and are read back in that deal
                // (the barrier below also orders these stores before the reads)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) *reinterpret_cast<gv4u *>(wst + rt * 1024 + lane * 16) = wn[rt];
            }
            __syncthreads();
        } else {
            // tile blk is in buffer blk & 1 (everyone passed the barrier that ended step blk - 1, so nobody still
            // reads the other buffer): stage tile blk + 1 there now, then start the loads of tile blk + 2
            bcur = bread + (blk & 1) * kBuf;
            uint8_t *nxt = lds + ((blk + 1) & 1) * kBuf;
#if defined(BH_ABLATE) && (BH_ABLATE & 4)  // developer build: no LDS staging stores
            if (blk + 1 < p.nblk && p.m < 0) {
#else
            if (blk + 1 < p.nblk) {
#endif
#pragma unroll
                for (int i = 0; i < NB; ++i) *reinterpret_cast<gv4u *>(nxt + bdst[i]) = bn[i];
            }
        }
        gv4u wc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) wc[rt] = wn[rt];
        if (!WS) {
            const int n1 = blk + 1 < p.nblk ? blk + 1 : p.nblk - 1, n2 = blk + 2 < p.nblk ? blk + 2 : p.nblk - 1;
#if !(defined(BH_ABLATE) && (BH_ABLATE & 32))  // developer build: no weight loads in the loop
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) wn[rt] = *reinterpret_cast<const gv4u *>(wptr[rt] + (size_t)n1 * 1024);  // next step's weights
#endif
#if !(defined(BH_ABLATE) && (BH_ABLATE & 8))  // developer build: no activation tile loads
#pragma unroll
            for (int i = 0; i < NB; ++i) bn[i] = *reinterpret_cast<const gv4u *>(bsrc[i] + (size_t)n2 * 256);
#endif
        }
        if (WS == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // dword i of the lane = 32-blocks 2 i, 2 i + 1 of this 256-block
                // scales of step (blk, i): 16 bytes per row tile = rows 4 g .. 4 g + 3 x (block 2 i, 2 i + 1); the next step's
                // are requested now and used one iteration later
                gv4u scc[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) scc[rt] = scn[rt];
                {
                    int nx = blk * 4 + i + 1;
                    nx = nx < p.nblk * 4 ? nx : p.nblk * 4 - 1;
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) scn[rt] = *reinterpret_cast<const gv4u *>(sptr[rt] + 32 * nx);
                }
                v4i a[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        // dword 2 hb + (g >> 1) of lane (r, i): its elements 8 h .. 8 h + 7 (fields are transposed 4 x 4: a nibble half)
                        const uint32_t wd = *reinterpret_cast<const uint32_t *>(wrd + rt * 1024 + i * 256 + 8 * hb) >> sh3;
                        a[rt][2 * hb] = (int)__builtin_amdgcn_perm(0u, p.lut, wd & 0x03030303u);
                        a[rt][2 * hb + 1] = (int)__builtin_amdgcn_perm(0u, p.lut, (wd >> 2) & 0x03030303u);
                    }
                }
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    // one MFMA = one 32-block: results start from the inline constant 0 (nothing to clear afterwards)
                    const v4i zero = {0, 0, 0, 0};
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        // bread points at column byte 64 g: this lane's eight k of 32-block 2 i + hb sit at 32 (2 i + hb) + 8 g
                        const long b = *reinterpret_cast<const long *>(bcur + ct * 16 * CSTR + 64 * i + 32 * hb - 56 * g);
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt) {
                            const long av = (long)(uint32_t)a[rt][2 * hb] | ((long)(uint32_t)a[rt][2 * hb + 1] << 32);
                            acc[rt][ct] = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, b, zero, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        const uint32_t sw[4] = {scc[rt].x, scc[rt].y, scc[rt].z, scc[rt].w};  // dword j = row 4 g + j: (block 2 i, 2 i + 1)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
#pragma unroll
                            for (int tt = 0; tt < TTW; ++tt) {
                                const float cv = combine_digits_i<NDIG>(&acc[rt][tt * NDIG], j);
                                facc[rt][tt][j] = hb ? gfma_mix_hi(cv, sw[j], facc[rt][tt][j]) : gfma_mix_lo(cv, sw[j], facc[rt][tt][j]);
                            }
                        }
                    }
                }
            }
        } else if (WS != 2) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                v4i a[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const uint32_t wd = m == 0 ? wc[rt].x : m == 1 ? wc[rt].y : m == 2 ? wc[rt].z : wc[rt].w;
#if defined(BH_ABLATE) && (BH_ABLATE & 2)  // developer build: no code expansion
                    a[rt] = (v4i){(int)wd, (int)(wd >> 1), (int)(wd >> 2), (int)(wd >> 3)};
#else
                    a[rt] = gdecode16(wd, p.lut);
#endif
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#if defined(BH_ABLATE) && (BH_ABLATE & 1)  // developer build: no LDS operand reads
                    const v4i b = (v4i){ct + blk, m, ct, 1};
#else
                    const v4i b = *reinterpret_cast<const v4i *>(bcur + ct * 16 * CSTR + moff[m]);
#endif
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rt], b, acc[rt][ct], 0, 0, 0);
                }
            }
        } else {
            // k-slot (lane group kg, dword m, byte j) = column 64 kg + 16 m + j of the block: 32-block 2 kg + (m >> 1)
#pragma unroll 1
            for (int kg = 0; kg < 4; ++kg) {
                const int live = g == kg ? -1 : 0;
#pragma unroll 1
                for (int mp = 0; mp < 2; ++mp) {
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        const int m = 2 * mp + mm;
                        v4i a[4];
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt) {
                            const uint32_t lo = mm == 0 ? wc[rt].x : wc[rt].y, hi = mm == 0 ? wc[rt].z : wc[rt].w;
                            a[rt] = gdecode16(mp == 0 ? lo : hi, p.lut);  // dword m = 2 mp + mm
                        }
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) {
                            v4i b = *reinterpret_cast<const v4i *>(bcur + ct * 16 * CSTR + ((4 * g + m) ^ col_swz(c)) * 16);  // (m is a run-time index here)
                            b[0] &= live, b[1] &= live, b[2] &= live, b[3] &= live;
#pragma unroll
                            for (int rt = 0; rt < 4; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rt], b, acc[rt][ct], 0, 0, 0);
                        }
                    }
                    const int sb = 8 * blk + 2 * kg + mp;  // this pair's 32-block
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float sv = p.wscale[soff[rt][j] + sb];
#pragma unroll
                            for (int tt = 0; tt < TTW; ++tt) facc[rt][tt][j] += combine_digits<NDIG>(&acc[rt][tt * NDIG], j) * sv;
                        }
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (v4i){0, 0, 0, 0};
                    }
                }
            }
        }
#if !(defined(BH_ABLATE) && (BH_ABLATE & 16))  // developer build: no barrier per K step
        if (!WS) __syncthreads();  // every wave is done with buffer blk & 1
#endif
        if (WS == 1) {
            // one f32 weight scale per (row, 256-block): fold this block's exact sums into f32
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                int t = bx * 16 + rw * 4 + rt;
                t = t < n_tiles ? t : n_tiles - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int row = 16 * t + 4 * g + j;
                    row = row < p.rows ? row : p.rows - 1;
                    const float s = p.wscale[(size_t)row * p.nblk + blk];
#pragma unroll
                    for (int tt = 0; tt < TTW; ++tt) facc[rt][tt][j] += combine_digits<NDIG>(&acc[rt][tt * NDIG], j) * s;
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (v4i){0, 0, 0, 0};
            }
        }
    }

    // ---- epilogue: digits -> f32, x 2^(E_t - S), [residual | silu*mul], store (whole lines: store_wave_tiles) -----------------------
#pragma unroll
    for (int tt = 0; tt < TTW; ++tt) {
        const int tok0 = (by * CW * TTW + cw * TTW + tt) * 16;
        const float is = p.inv_scale[tok0 + c];  // (padding rows: 0)
        float val[4][4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) val[rt][j] = (WS ? facc[rt][tt][j] : combine_digits<NDIG>(&acc[rt][tt * NDIG], j)) * is;
        store_wave_tiles(p, val, tok0, c, g, bx * 16 + rw * 4);
    }
}

// ================================================================================================================================
// BitNet32-F16 (ternary codes x one f16 scale per 32 weights) on the f16 matrix cores: "2-bit weight unpack x f16 activation dot
// product, per-block scale" as north_star words it.  The int8 digit form above has to fold every 32-block's integer sums into f32
// with that block's scale -- 3 VALU per (row, token, block), 768 of the 980 VALU instructions of a K step (EXPERIMENTS 4.3) -- which
// made the headline storage format prefill at HALF the QK256 rate.  Here the block scale is folded into the WEIGHT instead:
// the A operand of v_mfma_f32_16x16x32_f16 is (+-s or 0) as f16 -- exact, s is an f16 value -- accumulating straight into the f32
// accumulator of the whole K loop: no fold, no per-block epilogue, and one MFMA per (32 columns, row tile, token tile) instead of two
// digit MFMAs.  Activations are f16 with one power-of-two scale per row (row maximum in [1, 2): 11-bit mantissa per ELEMENT, 2^-12
// relative rounding each).  Kernel: k_gemm_f16a below.

// activation rows -> f16 planes [row][kp] (zero padded), inv_scale[row] = 2^E (0 for padding rows)
template <int NV>
__global__ __launch_bounds__(256) void k_quant_rows_f16(QuantArgs p) {
    __shared__ double stat[8];
    __shared__ float smax[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nvec = p.cols >> 2, kvec = p.kp >> 2;
    const bool live = row < p.m;
    const int rr = live ? row : p.m - 1;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + 256 * i, ci = idx < nvec ? idx : nvec - 1;
        if (p.x_f16) {  // (wave-uniform) f16 rows from the producing kernel (BITNET_HIP_FUSE_X_F16), as k_quant_rows reads them:
                        // without this branch the QK256 route of BITNET_HIP_GEMM_F16A=1 read the f16 buffer as f32 rows
            typedef _Float16 qh4 __attribute__((ext_vector_type(4)));
            const qh4 hv = *reinterpret_cast<const qh4 *>(reinterpret_cast<const _Float16 *>(p.x) + (size_t)rr * p.cols + 4 * ci);
            v[i] = float4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        } else {
            v[i] = *reinterpret_cast<const float4 *>(p.x + (size_t)rr * p.cols + 4 * ci);
        }
        if (idx >= nvec || !live) v[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    if (p.ln_gamma) {  // LayerNorm without bias, with mean subtraction (T:67-100): the same arithmetic as k_quant_rows
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const double a = v[i].x, b = v[i].y, c = v[i].z, d = v[i].w;
            s1 += (a + b) + (c + d);
            s2 += (a * a + b * b) + (c * c + d * d);
        }
        s1 = qwave_sum_d(s1);
        s2 = qwave_sum_d(s2);
        if (lane == 0) {
            stat[2 * wave] = s1;
            stat[2 * wave + 1] = s2;
        }
        __syncthreads();
        s1 = (stat[0] + stat[2]) + (stat[4] + stat[6]);
        s2 = (stat[1] + stat[3]) + (stat[5] + stat[7]);
        const double mean_d = s1 / (double)p.cols, var_d = s2 / (double)p.cols - mean_d * mean_d;
        const float mean = (float)mean_d, denom = sqrtf((float)(var_d > 0.0 ? var_d : 0.0) + p.ln_eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < nvec && live) {
                const float4 g = *reinterpret_cast<const float4 *>(p.ln_gamma + 4 * idx);
                v[i].x = (v[i].x - mean) / denom * g.x;
                v[i].y = (v[i].y - mean) / denom * g.y;
                v[i].z = (v[i].z - mean) / denom * g.z;
                v[i].w = (v[i].w - mean) / denom * g.w;
            }
        }
    }
    float am = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) am = fmaxf(fmaxf(am, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
    am = qwave_max(am);
    if (lane == 0) smax[wave] = am;
    __syncthreads();
    am = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    int be = (int)((__float_as_uint(am) >> 23) & 0xffu);
    be = be < 32 ? 32 : be > 253 ? 253 : be;
    const float sc = __uint_as_float((uint32_t)(254 - be) << 23);  // 2^-E: the row maximum lands in [1, 2)
    const float inv_s = __uint_as_float((uint32_t)be << 23);       // 2^E
    if (tid == 0) p.inv_scale[row] = live ? inv_s : 0.0f;
    _Float16 *base = reinterpret_cast<_Float16 *>(p.planes) + (size_t)row * p.kp;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + 256 * i;
        if (idx >= kvec) continue;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 o = {(_Float16)(v[i].x * sc), (_Float16)(v[i].y * sc), (_Float16)(v[i].z * sc), (_Float16)(v[i].w * sc)};
        *reinterpret_cast<h4 *>(base + 4 * idx) = o;
    }
}

typedef _Float16 gh2 __attribute__((ext_vector_type(2)));
typedef _Float16 gh8 __attribute__((ext_vector_type(8)));

// ================================================================================================================================
// k_gemm_f16a: both storage formats on the f16 matrix cores.  Round 3's form (k_gemm_f16w) made one MFMA = one 32-block, for which each
// wave dealt its four 1-KiB tiles through a private LDS area (4 ds_write_b128 + 32 ds_read_b32 per K step) and fetched four scale dwords
// per row tile; this form needs neither -- measured, same box, 4096 tokens, quantiser included: gate|up 390 -> 366 us, down 236 -> 211,
// o 94 -> 85, q|k|v 122 -> 113.  A lane (row r, group g)
// of a streaming tile holds the 64 codes of columns 64 g .. 64 g + 63 of the 256-block (dword m = columns 16 m .. 16 m + 15 of them, the
// sixteen 2-bit fields transposed 4 x 4: kernels_mfma.hip k_retile).  The K = 32 MFMA (m, h) takes from every lane the EIGHT codes of
// columns 64 g + 16 m + 8 h .. + 7 -- its own registers, no exchange -- i.e. its 32 k-slots are four runs of 8 columns, one per lane
// group; the B operand only has to agree: lane (token c, group g) reads the 8 halves of token c at column 64 g + 16 m + 8 h, one 16-byte
// unit of the token's row.  The 32-element block of those columns is 2 g + (m >> 1): the lane's own k-group in the f16 scale tiles
// ([tile][256-block][k-group][row][2]: ONE dword per row tile and K step instead of four).
// Activation tile in LDS: [token 16 TTW][256 halves], unpadded, 16-byte unit u of token c stored at u ^ fswz(c), fswz(c) = (c & 3) |
// ((c & 8) >> 1): with it the four 16-lane service groups of a ds_read_b128 each touch sixteen different units (brute-forced over the
// lane groups of MI355X_MICROARCH.md's LDS table); double-buffered, one barrier per K step, the next step's loads in flight.
// FMT 0: QK256 (code map values as f16, no scale); FMT 1: BitNet32-F16 (value x the block's f16 scale, v_pk_mul_f16).
__device__ __forceinline__ int fswz(int c) { return (c & 3) | ((c & 8) >> 1); }

template <int FMT>
__device__ __forceinline__ gh8 expand8_f16(uint32_t w, int h, uint32_t lut_hi, gh2 s2) {
    const uint32_t ca = (w >> (4 * h)) & 0x03030303u, cb = (w >> (4 * h + 2)) & 0x03030303u;
    uint32_t t[4];
    t[0] = __builtin_amdgcn_perm(0u, lut_hi, __builtin_amdgcn_perm(0x0C0C0C0Cu, ca, 0x01040004u));
    t[1] = __builtin_amdgcn_perm(0u, lut_hi, __builtin_amdgcn_perm(0x0C0C0C0Cu, ca, 0x03040204u));
    t[2] = __builtin_amdgcn_perm(0u, lut_hi, __builtin_amdgcn_perm(0x0C0C0C0Cu, cb, 0x01040004u));
    t[3] = __builtin_amdgcn_perm(0u, lut_hi, __builtin_amdgcn_perm(0x0C0C0C0Cu, cb, 0x03040204u));
    gh2 w2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w2[q] = FMT == 1 ? __builtin_bit_cast(gh2, t[q]) * s2 : __builtin_bit_cast(gh2, t[q]);
    return (gh8){w2[0][0], w2[0][1], w2[1][0], w2[1][1], w2[2][0], w2[2][1], w2[3][0], w2[3][1]};
}

// ================================================================================================================================
// QB32: block-scaled fixed-point activation rows for the fp6 x fp4 prompt matmul, written by the PRODUCER of the activations (round 5).
// The row quantiser of the digit forms needs the row maximum -- a value no producing workgroup has -- so it stayed a launch of its own
// (two per layer, 20 us each at 4096 tokens: 6 % of the QK256 prompt).  The scaled MFMA takes one E8M0 scale per lane = per 32 K-slots of
// one token, so the scale can be LOCAL: per 32-column unit u of a token, E_u = exponent of the unit's largest |v|, q = rint(v 2^(13 - E_u))
// (|q| <= 2^14: the same 15-bit integer class as the 2-digit planes and as the decode path's QAct, which scales per 16 columns), written as
// three balanced base-32 digits in fp6 exactly as k_quant_rows<2, NV, 1> writes them; the unit's byte s_u = E_u + 117 + 5 d is the MFMA's
// scale operand for digit d (2^(s - 127) * n / 8 = 2^(E_u - 13 + 5 d) n).  Products are exact; the f32 accumulator rounds across units of
// different exponents (2^-24 relative per addition, as every f32-accumulating form here).  LayerNorm is applied AFTER the product, as on the
// f16 chain: the producer multiplies by the consumer's gamma and leaves (sum, sum of squares) partials of the exact f32 row.
// One work item packs one unit: 32 f32 values from LDS (eight 16-byte pieces of a 144-byte slot: the ninth piece pads the slot so that
// sixteen lanes reading sixteen slots touch sixteen different bank groups) -> 3 x 24 bytes of digits at dst + 48 d, returns s_u (d = 0).
__device__ __forceinline__ uint32_t qb32_pack_unit(const float *src, uint8_t *dst) {
    typedef float qv16f __attribute__((ext_vector_type(16)));
    typedef unsigned qv6u __attribute__((ext_vector_type(6)));
    float4 f[8];
    float am = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f[k] = *reinterpret_cast<const float4 *>(src + 4 * k);
        am = fmaxf(fmaxf(am, fmaxf(fabsf(f[k].x), fabsf(f[k].y))), fmaxf(fabsf(f[k].z), fabsf(f[k].w)));
    }
    uint32_t be = (__float_as_uint(am) >> 23) & 0xffu;  // biased exponent of the unit's maximum (E_u = be - 127)
    be = be < 24u ? 24u : be > 240u ? 240u : be;
    const float sc = __uint_as_float((267u - be) << 23);  // 2^(13 - E_u)
    qv16f lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float xs[4] = {f[k].x, f[k].y, f[k].z, f[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float q = __builtin_rintf(xs[e] * sc);
            const float r1 = __builtin_rintf(q * 0.03125f), d0 = __builtin_fmaf(r1, -32.0f, q);
            const float r2 = __builtin_rintf(r1 * 0.03125f), d1 = __builtin_fmaf(r2, -32.0f, r1);
            const int ix = 4 * (k >> 1) + e;
            if (k & 1) hi[0][ix] = d0, hi[1][ix] = d1, hi[2][ix] = r2;
            else lo[0][ix] = d0, lo[1][ix] = d1, lo[2][ix] = r2;
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const qv6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(lo[d], hi[d], 8.0f);  // (k-slot order: k_quant_rows explains)
        uint2 *o = reinterpret_cast<uint2 *>(dst + d * 48);
        o[0] = uint2{r[0], r[1]}, o[1] = uint2{r[2], r[3]}, o[2] = uint2{r[4], r[5]};
    }
    return be - 10u;
}
// where unit U (32 columns) of a token's row lives: records [token][blk = U / 8] of kQbRec = 592 bytes = digits [g = (U / 2) % 4][digit][m = U % 2][24 bytes]
// (576 bytes, the 36 sixteen-byte units the matmul stages per token and K step) + exponent byte U % 8 at 576 (unit 36: staged with them -- as a
// separate array the bytes were four 2-byte gathers per lane and K step, which cost the vector-memory path 0.5 us per K step: 40 us per layer)
constexpr int kQbRec = 592;
__device__ __forceinline__ uint8_t *qb32_unit_ptr(uint8_t *planes, size_t token, int nblk, int U) {
    return planes + ((size_t)token * nblk + (U >> 3)) * kQbRec + ((U >> 1) & 3) * 144 + (U & 1) * 24;
}
__device__ __forceinline__ uint8_t *qb32_exp_ptr(uint8_t *planes, size_t token, int nblk, int U) {
    return planes + ((size_t)token * nblk + (U >> 3)) * kQbRec + 576 + (U & 7);
}

// f32 rows -> QB32 of gamma * x (gamma nullable) + the row's (sum, sum of squares) as partial 0 of the consumer's LayerNorm statistics:
// the QB32 chain's entry (the embedding rows), one workgroup per row -- the twin of k_rows_to_f16.
__global__ __launch_bounds__(256) void k_rows_to_qb32(const float *__restrict__ x, const float *__restrict__ gamma, int m, int cols, uint8_t *__restrict__ planes,
                                                      float *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float qb_row[];
    __shared__ double red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool live = row < m;
    double s1 = 0.0, s2 = 0.0;
    for (int i = tid; i < cols / 4; i += 256) {
        float4 v = live ? *reinterpret_cast<const float4 *>(x + (size_t)row * cols + 4 * i) : float4{0.f, 0.f, 0.f, 0.f};
        s1 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        s2 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        if (gamma) {
            const float4 gm = *reinterpret_cast<const float4 *>(gamma + 4 * i);
            v.x *= gm.x, v.y *= gm.y, v.z *= gm.z, v.w *= gm.w;
        }
        *reinterpret_cast<float4 *>(qb_row + ((i >> 3) * 9 + (i & 7)) * 4) = v;
    }
    s1 = qwave_sum_d(s1), s2 = qwave_sum_d(s2);
    if (lane == 0) red[2 * wave] = s1, red[2 * wave + 1] = s2;
    __syncthreads();
    if (tid == 0 && stats) {
        s1 = (red[0] + red[2]) + (red[4] + red[6]);
        s2 = (red[1] + red[3]) + (red[5] + red[7]);
        *reinterpret_cast<float2 *>(stats + 2 * (size_t)row) = float2{(float)s1, (float)s2};
    }
    const int n_units = cols >> 5, nblk = cols >> 8;
    for (int u = tid; u < n_units; u += 256) *qb32_exp_ptr(planes, row, nblk, u) = (uint8_t)qb32_pack_unit(qb_row + u * 36, qb32_unit_ptr(planes, row, nblk, u));
}

// ---- the f16 chain's epilogues (k_gemm_f16a<.., EPI = 1>; NW = 8: the ring form of tools/probes/gemm_f16_ring.patch): NW waves of RT row tiles x TTW token tiles ------
// LayerNorm after the product (T:67-100 applied to the INPUT): W . LN(x) = (W . (gamma * x) - mean g) / denom, g_r = W[r, :] . gamma; mean / denom of
// the workgroup's WG_TOK tokens from the producer's per-slab partial sums of the exact f32 x, added up in a fixed order (f64): deterministic.
// Runs in the PROLOGUE of the consuming kernel (round 5), while the first tiles' loads are in flight: at the start of the epilogue it was 10
// dependent L2 round trips (later one batch of 10 loads) plus two barriers that no MFMA of the workgroup overlapped -- 7 us of a 40 us gate|up
// workgroup.  `red` = 16 * NW * 64 bytes of scratch that nobody else touches before the next barrier; mu_rs = WG_TOK float2 that live to the epilogue.
// Two halves: the loads of the first ten partials per thread are ISSUED before the kernel's first tile loads (their round trip -- the partials were
// written by the previous kernel on other XCDs -- then runs beside the tile's), the sums are FINISHED behind the tile's staging.
template <int WG_TOK, int NW>
struct ChainLnStats {
    static constexpr int NG = NW * 64 / WG_TOK;
    float2 v[10];
    const float *sp;
    int pg;
    __device__ __forceinline__ void issue(const GemmArgs &p, int by, int tid) {
        const int tk = tid % WG_TOK;
        pg = tid / WG_TOK;
        sp = p.stats_in + 2 * ((size_t)by * WG_TOK + tk);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            // UNCONDITIONAL loads (index clamped, value selected): behind a per-load condition hipcc put `s_waitcnt vmcnt(0)` + the f64
            // conversion inside every branch -- ten serial round trips, 4.5 us per workgroup, 1.5 ms of a 4096-token prompt
            const int i = pg + k * NG, ic = i < p.n_stats ? i : p.n_stats - 1;
            v[k] = *reinterpret_cast<const float2 *>(sp + 2 * (size_t)ic * p.stats_stride);  // (finish() zeroes the clamped ones: no use of the value here,
        }                                                                                  //  so no wait ahead of the tile loads that follow)
    }
    // sum(): this thread's partials added up and left in `red` (16 * NW * 64 bytes of LDS nobody else touches); NO barrier here: the caller's next
    // barrier (the one that ends its prologue) orders it before final(), which the first WG_TOK threads run behind that barrier while the other
    // waves go on into the K loop -- mu_rs is first read in the epilogue, behind every K step's barrier.  (As finish() with its own barrier
    // in the prologue, every workgroup paid two barriers and a serial f64 section before its first MFMA.)
    __device__ __forceinline__ void sum(const GemmArgs &p, double *red, int tid) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const bool in = pg + k * NG < p.n_stats;
            s1 += (double)(in ? v[k].x : 0.0f), s2 += (double)(in ? v[k].y : 0.0f);  // (+ 0.0 past the end: the sum is unchanged)
        }
        for (int i0 = pg + 10 * NG; i0 < p.n_stats; i0 += 10 * NG) {  // more than 10 NG partials (K > 2560): further batches, same fixed order
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const int i = i0 + k * NG, ic = i < p.n_stats ? i : p.n_stats - 1;
                const float2 t = *reinterpret_cast<const float2 *>(sp + 2 * (size_t)ic * p.stats_stride);
                v[k] = float2{i < p.n_stats ? t.x : 0.0f, i < p.n_stats ? t.y : 0.0f};
            }
#pragma unroll
            for (int k = 0; k < 10; ++k) s1 += (double)v[k].x, s2 += (double)v[k].y;
        }
        red[2 * tid] = s1, red[2 * tid + 1] = s2;
    }
    __device__ __forceinline__ void final(const GemmArgs &p, const double *red, float2 *mu_rs, int tid) {
        if (tid < WG_TOK) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int q = 0; q < NG; ++q) s1 += red[2 * (q * WG_TOK + tid)], s2 += red[2 * (q * WG_TOK + tid) + 1];
            const double mean_d = s1 / (double)p.cols, var_d = s2 / (double)p.cols - mean_d * mean_d;
            const float denom = sqrtf((float)(var_d > 0.0 ? var_d : 0.0) + p.ln_eps);
            mu_rs[tid] = float2{(float)mean_d, 1.0f / denom};
        }
    }
};

// QB = true (TTW = 4 only): the outputs also leave as QB32 of gamma_out * y for the next fp6-form matmul (GemmArgs::qb_out): a token's NW RT / 2 units of
// 32 output rows are staged in LDS half a token tile (32 tokens) at a time and packed one unit per work item (qb32_pack_unit).
template <int RT, int TTW, int NW, bool QB = false>
__device__ __forceinline__ void f16_chain_epilogue(const GemmArgs &p, gv4f (&acc)[RT][TTW], uint8_t *lds, const float2 *mu_rs, int bx, int by, int rw, int c, int g, int tid) {
    static_assert(!QB || TTW == 4, "the QB32 hand-over is staged in two halves of a 64-token tile");
    constexpr int WG_TOK = TTW * 16;
    const int tile0 = bx * (NW * RT) + rw * RT;
    float lng[RT][4], gout[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        int row0 = 16 * (tile0 + rt) + 4 * g;
        row0 = row0 + 3 < p.rows ? row0 : p.rows - 4;  // (rows % 256 == 0 on this path: never taken)
        const float4 lg = p.stats_in ? *reinterpret_cast<const float4 *>(p.ln_g + row0) : float4{0.f, 0.f, 0.f, 0.f};
        const float4 go = p.gamma_out ? *reinterpret_cast<const float4 *>(p.gamma_out + row0) : float4{1.f, 1.f, 1.f, 1.f};
        lng[rt][0] = lg.x, lng[rt][1] = lg.y, lng[rt][2] = lg.z, lng[rt][3] = lg.w;
        gout[rt][0] = go.x, gout[rt][1] = go.y, gout[rt][2] = go.z, gout[rt][3] = go.w;
    }
    typedef _Float16 gh4 __attribute__((ext_vector_type(4)));
    bool sat = false;  // an f16 hand-over value of a live token was clamped (counted once per lane at the end)
#pragma unroll
    for (int tt = 0; tt < TTW; ++tt) {
        const int tok0 = (by * TTW + tt) * 16, token = tok0 + c;
        const bool live = token < p.m;
        float val[RT][4];
        float2 mr = float2{0.f, 1.f};
        if (p.stats_in) mr = mu_rs[tt * 16 + c];
        const float is = p.inv_scale ? p.inv_scale[token] : 1.0f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) val[rt][j] = (acc[rt][tt][j] * is - mr.x * lng[rt][j]) * mr.y;
        if (RT == 4 && p.silu_mul) {
            // row tiles alternate (gate, up): FeedForward::forward T:756-781; the product goes out as f16 rows for the down-projection
            const int half_rows = p.rows >> 1, ra = 16 * (tile0 >> 1) + 4 * g;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gv = val[2 * pr][j], uv = val[2 * pr + 1][j];
                    r[j] = gsilu_mul(gv, uv);
                }
                if (live) {
                    if (p.yh) {
                        gh4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = gclamp_f16(r[j], sat);
                        *reinterpret_cast<gh4 *>(p.yh + (size_t)token * half_rows + ra + 16 * pr) = o;
                    }
                    if (p.y) store_out4(p.y + (size_t)token * half_rows + ra + 16 * pr, r[0], r[1], r[2], r[3]);
                }
            }
            continue;
        }
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row0 = 16 * (tile0 + rt) + 4 * g;
            const size_t off = (size_t)(live ? token : 0) * p.rows + row0;
            if (p.residual) {  // x = x + W h (in place: y aliases the residual), T:1073
                const float4 rv = *reinterpret_cast<const float4 *>(p.residual + off);
                val[rt][0] += rv.x, val[rt][1] += rv.y, val[rt][2] += rv.z, val[rt][3] += rv.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) s1 += val[rt][j], s2 += val[rt][j] * val[rt][j];
            if (live) {
                if (p.y) *reinterpret_cast<float4 *>(p.y + off) = float4{val[rt][0], val[rt][1], val[rt][2], val[rt][3]};
                if (p.yh) {
                    gh4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = gclamp_f16(val[rt][j] * gout[rt][j], sat);
                    *reinterpret_cast<gh4 *>(p.yh + off) = o;
                }
            }
            if (QB) {  // stage gamma_out * y: token (tt & 1) 16 + c of this half, columns 16 (rw RT + rt) + 4 g .. + 3 of the workgroup's NW RT 16
                constexpr int NU = NW * RT / 2;
                float *st = reinterpret_cast<float *>(lds + 8192);  // (behind the LayerNorm scratch of this function)
                const int col = 16 * (rw * RT + rt) + 4 * g;
                *reinterpret_cast<float4 *>(st + ((((tt & 1) * 16 + c) * NU + (col >> 5)) * 9) * 4 + (col & 31)) =
                    float4{val[rt][0] * gout[rt][0], val[rt][1] * gout[rt][1], val[rt][2] * gout[rt][2], val[rt][3] * gout[rt][3]};
            }
        }
        if (QB && (tt & 1)) {
            constexpr int NU = NW * RT / 2;
            const float *st = reinterpret_cast<const float *>(lds + 8192);
            __syncthreads();
            for (int i = tid; i < 32 * NU; i += NW * 64) {
                const int tl = i / NU, u = i - tl * NU, U = bx * NU + u;
                const size_t tok = (size_t)by * WG_TOK + (tt >> 1) * 32 + tl;  // (rows up to m_pad exist in every QB32 buffer)
                *qb32_exp_ptr(p.qb_out, tok, p.qb_nblk_out, U) = (uint8_t)qb32_pack_unit(st + (tl * NU + u) * 36, qb32_unit_ptr(p.qb_out, tok, p.qb_nblk_out, U));
            }
            __syncthreads();
        }
        if (p.stats_out) {  // this wave's 16 RT rows of the token: the consumer's LayerNorm adds the slabs up
            s1 += __shfl_xor(s1, 16), s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
            if (g == 0) *reinterpret_cast<float2 *>(p.stats_out + 2 * ((size_t)(bx * NW + rw) * p.stats_stride + token)) = live ? float2{s1, s2} : float2{0.f, 0.f};
            // RT == 5: rows / 80 partials fill the first slabs of the [rows / 64] array the interface names; the first NW / 4 waves of
            // every workgroup clear the rows / 320 surplus ones (behind the NW gx real ones), so the consumer still adds rows / 64 entries up
            if (RT == 5 && rw < NW / 4 && g == 0)
                *reinterpret_cast<float2 *>(p.stats_out + 2 * ((size_t)(NW * gridDim.x + bx * (NW / 4) + rw) * p.stats_stride + token)) = float2{0.f, 0.f};
        }
    }
    gflush_sat(sat);
}

// RT = row tiles per wave: 4 (256-row workgroups), or 5 (320-row workgroups, chain form only): a 2560-row matrix x 4096 tokens is then
// 8 x 64 = 512 workgroups -- exactly one round of the 512 slots (two 4-wave workgroups per CU) instead of 640 in two rounds.
template <int FMT, int TTW, int EPI = 0, int RT = 4>
__global__ __launch_bounds__(256, 2) void k_gemm_f16a(GemmArgs p, uint32_t lut_hi) {
    static_assert(RT == 4 || EPI >= 1, "the 5-tile wave exists for the chain's epilogue only");  // EPI 2 = chain epilogue + QB32 hand-over
    constexpr int WG_TOK = TTW * 16, NB = WG_TOK * 32 / 256, ROWB = 512, kBuf = WG_TOK * ROWB;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4, rw = wave;
    const int n_tiles = (p.rows + 15) >> 4;
    const int kp = p.nblk * 256;
    int bx = blockIdx.x, by = blockIdx.y;
    {  // one XCD works through consecutive logical ids (k_gemm_mfma explains)
        const int gx = gridDim.x, total = gx * gridDim.y, id = by * gx + bx;
        if ((total & 7) == 0) {
            const int l = (id & 7) * (total >> 3) + (id >> 3);
            bx = l % gx;
            by = l / gx;
            if (p.wgroup > 0) {  // weight-stationary walk: groups of `wgroup` row blocks, inside a group the token tiles, inside a tile the group's blocks
                const int per = (int)gridDim.y * p.wgroup, grp = l / per, r = l - grp * per;
                by = r / p.wgroup;
                bx = grp * p.wgroup + (r - by * p.wgroup);
            }
        }
    }
    bx += p.bx_off;
    const uint8_t *wptr[RT];
    const uint32_t *sptr[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        int t = bx * (4 * RT) + rw * RT + rt;
        t = t < n_tiles ? t : n_tiles - 1;
        wptr[rt] = p.tiles + ((size_t)t * p.nblk * 64 + lane) * 16;
        sptr[rt] = FMT == 1 ? reinterpret_cast<const uint32_t *>(p.stiles_h) + (size_t)t * p.nblk * 64 + g * 16 + c : nullptr;
    }
    const _Float16 *planes = reinterpret_cast<const _Float16 *>(p.planes);
    const uint8_t *bsrc[NB];
    int bdst[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + 256 * i, tok = idx >> 5, seg = idx & 31;
        bsrc[i] = reinterpret_cast<const uint8_t *>(planes + (size_t)(by * WG_TOK + tok) * kp) + seg * 16;
        bdst[i] = tok * ROWB + (seg ^ fswz(tok & 15)) * 16;
    }
    int boff[8];  // this lane's unit of MFMA (m, h): 8 g + ((2 m + h) ^ fswz(c))
#pragma unroll
    for (int u = 0; u < 8; ++u) boff[u] = c * ROWB + 128 * g + ((u ^ fswz(c)) * 16);
    gv4f acc[RT][TTW];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < TTW; ++ct) acc[rt][ct] = (gv4f){0.f, 0.f, 0.f, 0.f};
    ChainLnStats<WG_TOK, 4> lnst;
    const bool ln_in = EPI && p.stats_in;
    if (ln_in) lnst.issue(p, by, tid);
    gv4u wn[RT], bn[NB];
    uint32_t sn[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) wn[rt] = *reinterpret_cast<const gv4u *>(wptr[rt]), sn[rt] = 0;
    if (FMT == 1) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) sn[rt] = sptr[rt][0];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) bn[i] = *reinterpret_cast<const gv4u *>(bsrc[i]);
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<gv4u *>(lds + bdst[i]) = bn[i];
    {
        const int n1 = p.nblk > 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) bn[i] = *reinterpret_cast<const gv4u *>(bsrc[i] + (size_t)n1 * 512);
    }
    // the input LayerNorm's (mean, 1 / denom) per token and the reduction's scratch, behind the tile buffers
    float2 *mu_rs = reinterpret_cast<float2 *>(lds + 2 * kBuf);
    double *ln_red = reinterpret_cast<double *>(lds + 2 * kBuf + WG_TOK * 8);
    if (ln_in) lnst.sum(p, ln_red, tid);
    __syncthreads();
    if (ln_in) lnst.final(p, ln_red, mu_rs, tid);

#ifdef BH_STAMPS
    // phase cycles of this wave (s_memtime), summed over the K steps: 0 staging stores (incl. their wait for the tile's loads), 1 issuing
    // the next loads, 2 expansion + operand reads + MFMAs, 3 the barrier; 4 the prologue, 5 the epilogue (written by the caller side below)
    unsigned long long ph[4] = {0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    unsigned long long t_prev = t_begin;
#define F16A_PHASE(i)                                                    \
    do {                                                                 \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();   \
        ph[i] += t_now - t_prev;                                         \
        t_prev = t_now;                                                  \
    } while (0)
#else
#define F16A_PHASE(i) \
    do {              \
    } while (0)
#endif
#ifdef BH_STAMPS
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
    t_prev = t_loop;
#endif
    for (int blk = 0; blk < p.nblk; ++blk) {
        const uint8_t *bcur = lds + (blk & 1) * kBuf;
        uint8_t *nxt = lds + ((blk + 1) & 1) * kBuf;
        if (blk + 1 < p.nblk) {
#pragma unroll
            for (int i = 0; i < NB; ++i) *reinterpret_cast<gv4u *>(nxt + bdst[i]) = bn[i];
        }
        F16A_PHASE(0);
        gv4u wc[RT];
        uint32_t sc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) wc[rt] = wn[rt], sc[rt] = sn[rt];
        {
            const int n1 = blk + 1 < p.nblk ? blk + 1 : p.nblk - 1, n2 = blk + 2 < p.nblk ? blk + 2 : p.nblk - 1;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) wn[rt] = *reinterpret_cast<const gv4u *>(wptr[rt] + (size_t)n1 * 1024);
            if (FMT == 1) {
#pragma unroll
#if defined(BH_ABLATE) && (BH_ABLATE & 4)  // developer build: no scale loads in the loop
                for (int rt = 0; rt < RT; ++rt) sn[rt] = 0x3C003C00u + (uint32_t)n1;
#else
                for (int rt = 0; rt < RT; ++rt) sn[rt] = sptr[rt][(size_t)n1 * 64];
#endif
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) bn[i] = *reinterpret_cast<const gv4u *>(bsrc[i] + (size_t)n2 * 512);
        }
        F16A_PHASE(1);
#ifdef BH_IGLP
        if (RT == 4) __builtin_amdgcn_iglp_opt(BH_IGLP);
#endif
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                gh8 a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const uint32_t wd = m == 0 ? wc[rt].x : m == 1 ? wc[rt].y : m == 2 ? wc[rt].z : wc[rt].w;
                    // (s, s) of this lane's 32-block 2 g + (m >> 1): low / high half of its scale dword
                    const gh2 s2 = __builtin_bit_cast(gh2, __builtin_amdgcn_perm(0u, sc[rt], (m >> 1) ? 0x03020302u : 0x01000100u));
#if defined(BH_ABLATE) && (BH_ABLATE & 2)  // developer build: no code expansion
                    a[rt] = __builtin_bit_cast(gh8, (gv4u){wd, wd >> (h + 1), sc[rt], wd ^ lut_hi});
#else
                    a[rt] = expand8_f16<FMT>(wd, h, lut_hi, s2);
#endif
                }
#pragma unroll
                for (int ct = 0; ct < TTW; ++ct) {
#if defined(BH_ABLATE) && (BH_ABLATE & 1)  // developer build: no LDS operand reads
                    const gh8 b = __builtin_bit_cast(gh8, (gv4u){(unsigned)(ct + blk), (unsigned)m, (unsigned)h, 1u});
#else
                    const gh8 b = *reinterpret_cast<const gh8 *>(bcur + ct * 16 * ROWB + boff[2 * m + h]);
#endif
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], b, acc[rt][ct], 0, 0, 0);
                }
            }
        }
        F16A_PHASE(2);
        __syncthreads();
        F16A_PHASE(3);
    }
#ifdef BH_STAMPS
    if (p.stamps && lane == 0) {
        unsigned long long *o = p.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        o[0] = ph[0], o[1] = ph[1], o[2] = ph[2], o[3] = ph[3];
        o[4] = t_loop - t_begin;
        o[6] = t_begin;
        o[7] = __builtin_amdgcn_s_memtime();
    }
#endif
#undef F16A_PHASE
    if (EPI == 0) {
#pragma unroll
        for (int tt = 0; tt < TTW; ++tt) {
            const int tok0 = (by * TTW + tt) * 16;
            const float is = p.inv_scale[tok0 + c];
            float val[4][4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)  // (RT == 4 here)
#pragma unroll
                for (int j = 0; j < 4; ++j) val[rt][j] = acc[rt][tt][j] * is;
            store_wave_tiles(p, val, tok0, c, g, bx * 16 + rw * 4);
        }
        return;
    }
    f16_chain_epilogue<RT, TTW, 4, EPI == 2>(p, acc, lds, mu_rs, bx, by, rw, c, g, tid);
}

// ================================================================================================================================
// k_gemm_f16h: k_gemm_f16a's arithmetic on a 64-row x 128-TOKEN wave tile (round 5).  The code expansion is a fixed cost per
// weight and K step (17 VALU per 8 weights: 544 per wave and step against 128 MFMAs at 64 tokens -- the VALU issue port was the unit closest to
// full, EXPERIMENTS 4.6); at 128 tokens per wave the same expansion feeds 256 MFMAs.  128 f32 accumulators + the 128-token tile only fit with
// HALF-K LDS steps: a step stages the columns {64 g + 32 hs .. + 31 : g} (four runs of 32: what the MFMAs (m, h), m = 2 hs, 2 hs + 1, take from every
// lane group) -- 256 bytes per token, 32 KiB per buffer, two buffers, two workgroups per CU as before; the weight / scale tiles are fetched once per
// 256-column block and used by both halves.  LDS image of a token: 16 units of 16 bytes, unit q = 4 g + k (k = 2 (m - 2 hs) + h) stored at
// q ^ ((c & 3) | (c & 8)) -- conflict-free for the four 16-lane service groups of ds_read_b128 (brute-forced).  Chain epilogue only (EPI = 1 of
// k_gemm_f16a), four row tiles per wave.  The launcher pairs it with k_gemm_f16a: 128-token workgroups on as many row blocks as fill whole rounds of
// the 512 slots, 64-token workgroups on the rest (13824 rows x 4096 tokens: 48 x 32 = 1536 = 3 rounds, then 6 x 64 = 384 half-length ones).
// LDS-DMA: 64 lanes x 16 B from (scalar base + per-lane byte offset) to LDS bytes [lds_dst, lds_dst + 1024) -- no VGPR staging (kernels_prefill_attn.hip
// uses the same form).  Invisible to hipcc's vmcnt bookkeeping: the kernel waits for its pieces with an explicit s_waitcnt vmcnt(0) ahead of the barrier
// that publishes the tile; hipcc's own waits for the weight loads only ever over-wait (loads return in order).
__device__ __forceinline__ void gdma1k(unsigned lane_off, const void *sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// Register loads as BUFFER loads (SGPR resource descriptor + scalar offset + one 32-bit lane offset): written as pointer arithmetic hipcc folds base + offset
// into eight 64-bit VGPR pointers and spills them around k_gemm_f16h's 128 accumulators; as buffer loads they cost one VGPR and stay visible to hipcc's
// vmcnt bookkeeping.  (As inline-asm global loads they were 10 % faster still -- and wrong: hipcc is free to copy an asm output register before the data
// has landed.)  Matrices below 2 GiB of tiles (the launcher checks).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gbuf_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);  // raw buffer, dword format (gfx94x / gfx950 word 3)
}

template <int FMT>
__global__ __launch_bounds__(256, 2) void k_gemm_f16h(GemmArgs p, uint32_t lut_hi) {
    constexpr int RT = 4, TTW = 8, WG_TOK = TTW * 16, ROWB = 256, kBuf = WG_TOK * ROWB;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the tile bases stay in SGPRs)
    const int c = lane & 15, g = lane >> 4, rw = wave;
    const int n_tiles = (p.rows + 15) >> 4;
    const int kp = p.nblk * 256;
    int bx = blockIdx.x, by = blockIdx.y;
    {  // one XCD works through consecutive logical ids (k_gemm_mfma explains)
        const int gx = gridDim.x, total = gx * gridDim.y, id = by * gx + bx;
        if ((total & 7) == 0) {
            const int l = (id & 7) * (total >> 3) + (id >> 3);
            bx = l % gx;
            by = l / gx;
            if (p.wgroup > 0) {
                const int per = (int)gridDim.y * p.wgroup, grp = l / per, r = l - grp * per;
                by = r / p.wgroup;
                bx = grp * p.wgroup + (r - by * p.wgroup);
            }
        }
    }
    bx += p.bx_off;
    // wave-uniform bases (SGPR pairs) + 32-bit lane offsets: 128 accumulators leave no room for 64-bit per-lane pointers -- nor for a register-staged
    // activation tile (32 registers: hipcc spilled 144-228 bytes), so the tile goes global -> LDS by LDS-DMA, one half-K step ahead of its MFMAs
    const __amdgpu_buffer_rsrc_t rs_w = gbuf_rsrc(p.tiles), rs_s = gbuf_rsrc(FMT == 1 ? (const void *)p.stiles_h : (const void *)p.tiles);
    int wptr[RT], sptr[RT];  // scalar byte offsets of this wave's row tiles in the code / scale tile arrays
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        int t = bx * (4 * RT) + rw * RT + rt;
        t = t < n_tiles ? t : n_tiles - 1;
        wptr[rt] = t * p.nblk * 1024;
        sptr[rt] = t * p.nblk * 256;
    }
    const uint32_t woff = (uint32_t)lane * 16u, soff = (uint32_t)(g * 16 + c) * 4u;
    const uint32_t row_b = (uint32_t)kp * 2u;                                                                    // bytes of a token's f16 row
    const uint8_t *abase = reinterpret_cast<const uint8_t *>(p.planes) + (size_t)(by * WG_TOK + 32 * wave) * row_b;  // uniform: this wave stages tokens 32 w .. 32 w + 31
    // A half-step's tile is 32 pieces of 1 KiB (4 tokens x 16 units); wave w moves pieces 8 w .. 8 w + 7.  Lane l of piece i lands at slot (token 4 i + (l >> 4),
    // position l & 15): it fetches the unit that belongs there, q = (l & 15) ^ col_swz(token & 15) = columns 64 (q / 4) + 32 hs + 8 (q % 4) of the token's row.
    // token & 15 = 4 (i & 3) + (l >> 4): four lane-offset patterns; pieces i and i + 4 differ by 16 tokens (a scalar).
    uint32_t dsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tl = 4 * i + (lane >> 4), q = (lane & 15) ^ col_swz(tl);
        dsrc[i] = (uint32_t)tl * row_b + (uint32_t)((q >> 2) * 128 + (q & 3) * 16);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    auto stage = [&](int s, int buf) {  // half-step s (block s / 2, half s % 2) -> buffer buf
        const uint8_t *src = abase + (size_t)(s >> 1) * 512 + (size_t)(s & 1) * 64;
        const unsigned dst = lds0 + (unsigned)buf * kBuf + (unsigned)(8 * wave) * 1024u;
#pragma unroll
        for (int i = 0; i < 8; ++i) gdma1k(dsrc[i & 3], src + (size_t)(i >> 2) * 16 * row_b, dst + 1024u * i);
    };
    int boff[4];  // this lane's unit of MFMA k = 2 (m - 2 hs) + h
#pragma unroll
    for (int k = 0; k < 4; ++k) boff[k] = c * ROWB + (((4 * g + k) ^ col_swz(c)) * 16);
    ChainLnStats<WG_TOK, 4> lnst;
    const bool ln_in = p.stats_in != nullptr;
    if (ln_in) lnst.issue(p, by, tid);
    gv4u wn[RT];
    uint32_t sn[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) wn[rt] = __builtin_bit_cast(gv4u, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff, wptr[rt], 0)), sn[rt] = 0;
    if (FMT == 1) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) sn[rt] = __builtin_amdgcn_raw_buffer_load_b32(rs_s, soff, sptr[rt], 0);
    }
    const int n_steps = 2 * p.nblk;
    stage(0, 0);
    float2 *mu_rs = reinterpret_cast<float2 *>(lds + 2 * kBuf);
    double *ln_red = reinterpret_cast<double *>(lds + 2 * kBuf + WG_TOK * 8);
    if (ln_in) lnst.sum(p, ln_red, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ln_in) lnst.final(p, ln_red, mu_rs, tid);
    gv4f acc[RT][TTW];  // (declared behind the statistics' twenty registers: with both live the prologue spilled)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < TTW; ++ct) acc[rt][ct] = (gv4f){0.f, 0.f, 0.f, 0.f};
    for (int blk = 0; blk < p.nblk; ++blk) {
        // this block's weights / scales become current; the next block's are requested
        gv4u wc[RT];
        uint32_t sc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) wc[rt] = wn[rt], sc[rt] = sn[rt];
        {
            const int n1 = blk + 1 < p.nblk ? blk + 1 : p.nblk - 1;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) wn[rt] = __builtin_bit_cast(gv4u, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff, wptr[rt] + n1 * 1024, 0));
            if (FMT == 1) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) sn[rt] = __builtin_amdgcn_raw_buffer_load_b32(rs_s, soff, sptr[rt] + n1 * 256, 0);
            }
        }
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {  // (compile-time after unrolling: the half tile of step s = 2 blk + hs sits in buffer hs)
            const int s = 2 * blk + hs;
            const uint8_t *bcur = lds + hs * kBuf;
            if (s + 1 < n_steps) stage(s + 1, hs ^ 1);  // (every wave is past the barrier that ended step s - 1: nobody reads that buffer)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    gh8 a[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        // dword m = 2 hs + mm of the lane's 16 bytes; (s, s) of its 32-block 2 g + hs: low / high half of the scale dword
                        const uint32_t wd = hs == 0 ? (mm == 0 ? wc[rt].x : wc[rt].y) : (mm == 0 ? wc[rt].z : wc[rt].w);
                        const gh2 s2v = __builtin_bit_cast(gh2, __builtin_amdgcn_perm(0u, sc[rt], hs ? 0x03020302u : 0x01000100u));
                        a[rt] = expand8_f16<FMT>(wd, h, lut_hi, s2v);
                    }
#pragma unroll
                    for (int ct = 0; ct < TTW; ++ct) {
                        const gh8 b = *reinterpret_cast<const gh8 *>(bcur + ct * 16 * ROWB + boff[2 * mm + h]);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], b, acc[rt][ct], 0, 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next half tile have landed
            __syncthreads();
        }
    }
    f16_chain_epilogue<RT, TTW, 4>(p, acc, lds, mu_rs, bx, by, rw, c, g, tid);
}

// ================================================================================================================================
// k_gemm_fp6: the 2-digit form's arithmetic on the block-scaled fp6 x fp4 matrix instruction (round 4).
// v_mfma_scale_f32_16x16x128_f8f6f4 with A = fp4 (e2m1: every value of a code map in -2 .. 2 is exact) and B = fp6 (e2m3: n / 8 is
// exact for |n| <= 16, a balanced base-32 digit) multiplies K = 128 in the cycles the int8 form needs for K = 64, and its E8M0 block
// scales are free powers of two: the three digits of q = d0 + 32 d1 + 1024 d2 enter with scales 2^3, 2^8, 2^13 (the 2^3 undoes the
// n / 8) and accumulate into ONE f32 accumulator -- every product and every partial sum is an integer, exact in f32 while it stays
// below 2^24 (|w q| <= 2^15 per term: a sum only leaves that range when more than 512 terms line up at full scale; past it the
// accumulator rounds like any f32 sum, 2^-24 relative).  So this form computes the SAME integer as k_gemm_mfma<2, ...> from the SAME
// quantised q (k_quant_rows<2, NV, 1>) with 3 MFMAs per 128 columns instead of 4, half the accumulator registers, and multipliers a
// fraction of the int8 ones' size (the int8 loop runs at a power limit: EXPERIMENTS 4.5; tools/probes/mfma_fp6_probe.hip).
// Operands: lane (row r, group g) of a streaming tile holds the 64 codes of columns 64 g .. 64 g + 63; MFMA m = 0, 1 takes dwords
// 2 m, 2 m + 1 expanded to 32 fp4 nibbles (expand16_fp4: k-slot 8 q + n = column 64 g + 32 m + 8 q + 4 (n & 1) + (n >> 1)); the
// B operand of lane (token c, group g) is the 24 bytes the quantiser wrote for (g, digit, m) in that slot order.
// Activation tile in LDS: [g 4][token 16 TTW][144 bytes = digit 3 x m 2 x 24]: a lane's nine 16-byte units are contiguous, and the
// sixteen lanes of every ds_read_b128 service group (MI355X_MICROARCH.md, LDS: all 16 tokens, mixed g) touch units 9 c + const
// (mod 16): sixteen different ones, since 9 is odd and a group's slab is a multiple of 16 units -- conflict-free without padding.
typedef int gv8i __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void expand16_fp4(uint32_t w, uint32_t lut4, int &o0, int &o1) {
    const uint32_t p0 = __builtin_amdgcn_perm(0u, lut4, w & 0x03030303u), p1 = __builtin_amdgcn_perm(0u, lut4, (w >> 2) & 0x03030303u);
    const uint32_t p2 = __builtin_amdgcn_perm(0u, lut4, (w >> 4) & 0x03030303u), p3 = __builtin_amdgcn_perm(0u, lut4, (w >> 6) & 0x03030303u);
    o0 = (int)((p1 << 4) | p0);  // byte b: low nibble = element b, high nibble = element 4 + b of the dword's sixteen
    o1 = (int)((p3 << 4) | p2);  //         elements 8 + b, 12 + b
}

// The resident fp4 image (round 5): exactly the nibbles expand16_fp4 produces, stored once per matrix -- [row tile][256-block][MFMA m 2]
// [lane 64][16 B] = 2 KiB per (tile, block), twice the 2-bit streaming tile (which stays the decode path's copy).  k_gemm_fp6<.., RES = 1>
// loads its A operands straight from it: no code expansion in the K loop (13 VALU per 16 weights and lane before).
__global__ void k_retile_fp4(const uint8_t *__restrict__ tiles, uint8_t *__restrict__ tiles4, uint32_t lut4, size_t total16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte lane segment of the streaming tiles each
    if (i >= total16) return;
    const gv4u w = *reinterpret_cast<const gv4u *>(tiles + i * 16);
    int e[8];
    expand16_fp4(w.x, lut4, e[0], e[1]);
    expand16_fp4(w.y, lut4, e[2], e[3]);
    expand16_fp4(w.z, lut4, e[4], e[5]);
    expand16_fp4(w.w, lut4, e[6], e[7]);
    const size_t tb = i >> 6, lane = i & 63;
    *reinterpret_cast<v4i *>(tiles4 + ((tb * 2 + 0) * 64 + lane) * 16) = (v4i){e[0], e[1], e[2], e[3]};
    *reinterpret_cast<v4i *>(tiles4 + ((tb * 2 + 1) * 64 + lane) * 16) = (v4i){e[4], e[5], e[6], e[7]};
}

// RT = row tiles per wave: 4, or 5 (320-row workgroups: gemm_five_tiles; no silu * mul pairing with it).
// RES = 1: A operands from the resident fp4 image (GemmArgs::tiles4) instead of expanding the 2-bit tiles in the loop.
// EPI = 1: QB32 activations in (per-unit exponent bytes beside the digit records: the B scale operand is per lane) and the f16 chain's epilogue out
// (LayerNorm after the product from the producer's statistics partials, residual, silu * up as f16 rows, f32 rows).
template <int TTW, int RT = 4, int RES = 0, int EPI = 0>
__global__ __launch_bounds__(256, (TTW >= 4 || EPI) ? 2 : TTW == 2 ? 3 : 4) void k_gemm_fp6(GemmArgs p, uint32_t lut4) {  // (the chain epilogue's narrow forms spill under 3 / 4 waves per SIMD)
    // EPI: QB32 records (kQbRec = 592 bytes): unit 36 of a token's record holds the block's eight exponent bytes and is staged behind the digit tile
    // (the digit tile is staged as 36 sixteen-byte units per token by all threads; the exponent bytes as ONE 8-byte piece per token by the first T threads:
    //  as a 37th unit they made every thread's tenth load / store per K step -- 11 % more staging for 1.4 % more payload, and the launch 10 % slower)
    constexpr int T = 16 * TTW, REC = EPI ? kQbRec : 576, UNITS = T * 36, NB = (UNITS + 255) / 256, kDig = T * 576, kBuf = kDig + (EPI ? T * 8 : 0);
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the tile bases below stay in SGPRs)
    const int c = lane & 15, g = lane >> 4, rw = wave;
    const int n_tiles = (p.rows + 15) >> 4;
    int bx = blockIdx.x, by = blockIdx.y;
    {  // one XCD works through consecutive logical ids (k_gemm_mfma explains)
        const int gx = gridDim.x, total = gx * gridDim.y, id = by * gx + bx;
        if ((total & 7) == 0) {
            const int l = (id & 7) * (total >> 3) + (id >> 3);
            bx = l % gx;
            by = l / gx;
            if (p.wgroup > 0) {  // weight-stationary walk: groups of `wgroup` row blocks, inside a group the token tiles, inside a tile the group's blocks
                const int per = (int)gridDim.y * p.wgroup, grp = l / per, r = l - grp * per;
                by = r / p.wgroup;
                bx = grp * p.wgroup + (r - by * p.wgroup);
            }
        }
    }
    constexpr int NM = RES ? 2 : 1;            // 16-byte weight loads per row tile and K step
    constexpr size_t WSTEP = RES ? 2048 : 1024;  // bytes of one (tile, 256-block) in the layout read
    // Addresses are a wave-uniform base (SGPR pair) + a 32-bit lane offset (round 5: 64-bit per-lane pointers for 4-5 weight tiles, 9 staging
    // units and 4 exponent rows took 40 registers and a v_mad_i64 per load; the QB32 form spilled on them)
    const uint8_t *wptr[RT];  // uniform: this wave's row tiles
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        int t = bx * (4 * RT) + rw * RT + rt;
        t = t < n_tiles ? t : n_tiles - 1;
        wptr[rt] = RES ? p.tiles4 + (size_t)t * p.nblk * 2048 : p.tiles + (size_t)t * p.nblk * 1024;
    }
    const uint32_t woff = (uint32_t)lane * 16u;
    // this thread's 16-byte units of the activation tile: unit u of a token's 576 bytes = (g = u / 9, k = u % 9)
    const uint32_t row_bytes = (uint32_t)p.nblk * (uint32_t)REC;
    const uint8_t *abase = reinterpret_cast<const uint8_t *>(p.planes) + (size_t)by * T * row_bytes;  // uniform
    uint32_t bsrc[NB];
    int bdst[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int idx = tid + 256 * i;
        idx = idx < UNITS ? idx : UNITS - 1;  // (surplus threads of the last round repeat its last unit: same bytes, same place)
        const int tok = idx / 36, u = idx % 36, gg = u / 9, k = u % 9;
        bsrc[i] = (uint32_t)tok * row_bytes + (uint32_t)u * 16u;
        bdst[i] = ((gg * T + tok) * 9 + k) * 16;
    }
    const bool e_thr = EPI && tid < T;  // this thread stages token tid's eight exponent bytes
    const uint32_t esrc = (uint32_t)(tid < T ? tid : 0) * row_bytes + 576u;
    typedef unsigned gv2u __attribute__((ext_vector_type(2)));
    gv2u exn = {0u, 0u};
    ChainLnStats<T, 4> lnst;
    const bool ln_in = EPI && p.stats_in;
    if (ln_in) lnst.issue(p, by, tid);
    gv4f acc[RT][TTW];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < TTW; ++ct) acc[rt][ct] = (gv4f){0.f, 0.f, 0.f, 0.f};
    gv4u wn[NM][RT], bn[NB];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int m = 0; m < NM; ++m) wn[m][rt] = *reinterpret_cast<const gv4u *>(wptr[rt] + m * 1024 + woff);
#pragma unroll
    for (int i = 0; i < NB; ++i) bn[i] = *reinterpret_cast<const gv4u *>(abase + bsrc[i]);
    if (e_thr) exn = *reinterpret_cast<const gv2u *>(abase + esrc);
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<gv4u *>(lds + bdst[i]) = bn[i];
    if (e_thr) *reinterpret_cast<gv2u *>(lds + kDig + tid * 8) = exn;
    {
        const int n1 = p.nblk > 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) bn[i] = *reinterpret_cast<const gv4u *>(abase + (size_t)n1 * REC + bsrc[i]);
        if (e_thr) exn = *reinterpret_cast<const gv2u *>(abase + (size_t)n1 * REC + esrc);
    }
    // (EPI) the input LayerNorm's (mean, 1 / denom) per token and the reduction's scratch, behind the tile buffers
    float2 *mu_rs = reinterpret_cast<float2 *>(lds + 2 * kBuf);
    double *ln_red = reinterpret_cast<double *>(lds + 2 * kBuf + T * 8);
    if (ln_in) lnst.sum(p, ln_red, tid);
    __syncthreads();
    if (ln_in) lnst.final(p, ln_red, mu_rs, tid);
    const int bread = (g * T + c) * 144;
    // QB32: this lane's two exponent bytes (units m = 0, 1 of lane group g) per token tile and K step, requested one step ahead
    const int eread = kDig + c * 8 + 2 * g;  // QB32: this lane's two exponent bytes (units m = 0, 1 of lane group g) of token tile ct: + 128 ct
    uint32_t ec[TTW];
#pragma unroll
    for (int ct = 0; ct < TTW; ++ct) ec[ct] = 0u;

    for (int blk = 0; blk < p.nblk; ++blk) {
        const uint8_t *bcur = lds + (blk & 1) * kBuf + bread;
        uint8_t *nxt = lds + ((blk + 1) & 1) * kBuf;
        if (blk + 1 < p.nblk) {
#pragma unroll
            for (int i = 0; i < NB; ++i) *reinterpret_cast<gv4u *>(nxt + bdst[i]) = bn[i];
            if (e_thr) *reinterpret_cast<gv2u *>(nxt + kDig + tid * 8) = exn;
        }
        gv4u wc[NM][RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int m = 0; m < NM; ++m) wc[m][rt] = wn[m][rt];
        {
            const int n1 = blk + 1 < p.nblk ? blk + 1 : p.nblk - 1, n2 = blk + 2 < p.nblk ? blk + 2 : p.nblk - 1;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int m = 0; m < NM; ++m) wn[m][rt] = *reinterpret_cast<const gv4u *>(wptr[rt] + (size_t)n1 * WSTEP + m * 1024 + woff);
#pragma unroll
            for (int i = 0; i < NB; ++i) bn[i] = *reinterpret_cast<const gv4u *>(abase + (size_t)n2 * REC + bsrc[i]);
            if (e_thr) exn = *reinterpret_cast<const gv2u *>(abase + (size_t)n2 * REC + esrc);
        }
        if (EPI) {
#pragma unroll
            for (int ct = 0; ct < TTW; ++ct) ec[ct] = *reinterpret_cast<const uint16_t *>(lds + (blk & 1) * kBuf + eread + ct * 128);
        }
        // B operands one group (token tile, digit) ahead of their MFMAs: hipcc otherwise waits for every group's reads right after
        // issuing them (EXPERIMENTS 4.5, VAR2); group 0 is requested ahead of the code expansion
        // (the wide tile only: the 5-tile and the narrow forms have no registers to spare for the second operand set)
        constexpr int NG = TTW * 3;
        constexpr bool PIPE = TTW == 4 && RT == 4;
        v4i rr[2][3];
        if (PIPE) {
#pragma unroll
            for (int q = 0; q < 3; ++q) rr[0][q] = *reinterpret_cast<const v4i *>(bcur + 16 * q);
        }
        v4i a[2][RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (RES) {  // the image holds the operands as they are
                a[0][rt] = __builtin_bit_cast(v4i, wc[0][rt]);
                a[1][rt] = __builtin_bit_cast(v4i, wc[NM - 1][rt]);
                continue;
            }
            int e[8];
            expand16_fp4(wc[0][rt].x, lut4, e[0], e[1]);
            expand16_fp4(wc[0][rt].y, lut4, e[2], e[3]);
            expand16_fp4(wc[0][rt].z, lut4, e[4], e[5]);
            expand16_fp4(wc[0][rt].w, lut4, e[6], e[7]);
            a[0][rt] = (v4i){e[0], e[1], e[2], e[3]};
            a[1][rt] = (v4i){e[4], e[5], e[6], e[7]};
        }
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            const int ct = grp / 3, d = grp % 3;
            if (PIPE && grp + 1 < NG) {
                const uint8_t *bp = bcur + ((grp + 1) / 3) * 16 * 144 + ((grp + 1) % 3) * 48;
#pragma unroll
                for (int q = 0; q < 3; ++q) rr[(grp + 1) & 1][q] = *reinterpret_cast<const v4i *>(bp + 16 * q);
            }
            if (!PIPE) {
#pragma unroll
                for (int q = 0; q < 3; ++q) rr[grp & 1][q] = *reinterpret_cast<const v4i *>(bcur + ct * 16 * 144 + d * 48 + 16 * q);
            }
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
            const v4i r0 = rr[grp & 1][0], r1 = rr[grp & 1][1], r2 = rr[grp & 1][2];
            const gv8i b0 = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, -1, -1);
            const gv8i b1 = __builtin_shufflevector(r1, r2, 2, 3, 4, 5, 6, 7, -1, -1);
            // B scale byte (byte 0 of the operand): the digit's weight 2^(3 + 5 d) -- times the unit's own 2^(E_u - 13) on QB32 rows (s_u + 5 d)
            const int sb0 = EPI ? (int)(ec[ct] & 0xffu) + 5 * d : 130 + 5 * d, sb1 = EPI ? (int)(ec[ct] >> 8) + 5 * d : 130 + 5 * d;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[rt][ct] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_shufflevector(a[0][rt], a[0][rt], 0, 1, 2, 3, -1, -1, -1, -1), b0, acc[rt][ct], 4, 2, 0,
                                                                               127, 0, sb0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[rt][ct] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_shufflevector(a[1][rt], a[1][rt], 0, 1, 2, 3, -1, -1, -1, -1), b1, acc[rt][ct], 4, 2, 0,
                                                                               127, 0, sb1);
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    if (EPI) {
        f16_chain_epilogue<RT, TTW, 4>(p, acc, lds, mu_rs, bx, by, rw, c, g, tid);
        return;
    }
#pragma unroll
    for (int tt = 0; tt < TTW; ++tt) {
        const int tok0 = (by * TTW + tt) * 16;
        const float is = p.inv_scale[tok0 + c];  // (padding rows: 0)
        float val[4][4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) val[rt][j] = acc[rt][tt][j] * is;
        store_wave_tiles(p, val, tok0, c, g, bx * (4 * RT) + rw * RT);
        if (RT == 5) {  // the fifth tile: 64 contiguous bytes per token
            const int token = tok0 + c, row0 = 16 * (bx * 20 + rw * 5 + 4) + 4 * g;
            if (token < p.m && row0 + 3 < p.rows) {
                const size_t off = (size_t)token * p.rows + row0;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = acc[4][tt][j] * is;
                if (p.residual) {
                    const float4 r = *reinterpret_cast<const float4 *>(p.residual + off);
                    o[0] += r.x, o[1] += r.y, o[2] += r.z, o[3] += r.w;
                }
                store_out4(p.y + off, o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// k_gemm_fp6w (round 5): the same product, operands, k-order per accumulator and results as k_gemm_fp6<4, 4, 1, 0> on a 2 x 2 wave arrangement --
// wave (wr, wt) owns rows 128 wr .. + 127 (EIGHT row tiles) x tokens 32 wt .. + 31 (two token tiles) of the 256 x 64 workgroup tile, so a B operand read from
// LDS feeds eight MFMAs instead of four: half the LDS operand bytes per MFMA (the unit an ablation named: EXPERIMENTS 8.7), paid for with twice the weight
// loads per wave (the two waves of a row half fetch the same 2 KiB pieces: L1 / L2 hits).  128 weight registers (this step's + the next step's) leave no room for a
// register-staged activation tile: it goes global -> LDS by LDS-DMA (wave w moves lane-group slab g = w: 576 units of 16 bytes = nine 1 KiB pieces, each lane
// fetching the unit that belongs at its slot), requested a whole K step ahead; weights by buffer loads (visible to hipcc's vmcnt bookkeeping).
__global__ __launch_bounds__(256, 2) void k_gemm_fp6w(GemmArgs p) {
    constexpr int RT = 8, TTW = 2, T = 64, kBuf = T * 576;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4, wr = wave >> 1, wt = wave & 1;
    const int n_tiles = (p.rows + 15) >> 4;
    int bx = blockIdx.x, by = blockIdx.y;
    {  // one XCD works through consecutive logical ids (k_gemm_mfma explains)
        const int gx = gridDim.x, total = gx * gridDim.y, id = by * gx + bx;
        if ((total & 7) == 0) {
            const int l = (id & 7) * (total >> 3) + (id >> 3);
            bx = l % gx;
            by = l / gx;
            if (p.wgroup > 0) {
                const int per = (int)gridDim.y * p.wgroup, grp = l / per, r = l - grp * per;
                by = r / p.wgroup;
                bx = grp * p.wgroup + (r - by * p.wgroup);
            }
        }
    }
    const __amdgpu_buffer_rsrc_t rs_w = gbuf_rsrc(p.tiles4);
    int wptr[RT];  // scalar byte offsets of this wave's row tiles in the fp4 image
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        int t = bx * 16 + wr * 8 + rt;
        t = t < n_tiles ? t : n_tiles - 1;
        wptr[rt] = t * p.nblk * 2048;
    }
    const uint32_t woff = (uint32_t)lane * 16u;
    const uint32_t row_bytes = (uint32_t)p.nblk * 576u;
    const uint8_t *abase = reinterpret_cast<const uint8_t *>(p.planes) + (size_t)by * T * row_bytes + (size_t)wave * 144;  // uniform: slab g = wave of every token's record
    uint32_t dsrc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int r = i * 64 + lane, tok = r / 9, k = r - tok * 9;
        dsrc[i] = (uint32_t)tok * row_bytes + (uint32_t)k * 16u;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    auto stage = [&](int blk, int buf) {
        const uint8_t *src = abase + (size_t)blk * 576;
        const unsigned dst = lds0 + (unsigned)buf * kBuf + (unsigned)wave * 9216u;
#pragma unroll
        for (int i = 0; i < 9; ++i) gdma1k(dsrc[i], src, dst + 1024u * i);
    };
    gv4f acc[RT][TTW];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < TTW; ++ct) acc[rt][ct] = (gv4f){0.f, 0.f, 0.f, 0.f};
    v4i wn[2][RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int m = 0; m < 2; ++m) wn[m][rt] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff, wptr[rt] + m * 1024, 0));
    stage(0, 0);
    // (the builtin form: hipcc's own bookkeeping then knows its weight loads have retired and puts no counted wait in front of their first use -- a
    //  counted wait would, in hardware, also wait for the DMA requests it cannot see)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    asm volatile("" ::: "memory");
    __syncthreads();
    const int bread = (g * T + wt * 32 + c) * 144;
    for (int blk = 0; blk < p.nblk; ++blk) {
        const uint8_t *bcur = lds + (blk & 1) * kBuf + bread;
#if defined(BH_ABLATE) && (BH_ABLATE & 128)
        if (blk + 1 < p.nblk && p.m == 12345) stage(blk + 1, (blk + 1) & 1);
#else
        if (blk + 1 < p.nblk) stage(blk + 1, (blk + 1) & 1);  // (every wave is past the barrier that ended step blk - 1: nobody reads that buffer)
#endif
        v4i a[2][RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m][rt] = wn[m][rt];
        {
            const int n1 = blk + 1 < p.nblk ? blk + 1 : p.nblk - 1;
#if defined(BH_ABLATE) && (BH_ABLATE & 256)
            if (p.m == 12345)
#endif
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int m = 0; m < 2; ++m) wn[m][rt] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff, wptr[rt] + n1 * 2048 + m * 1024, 0));
        }
        constexpr int NG = TTW * 3;
        v4i rr[2][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) rr[0][q] = *reinterpret_cast<const v4i *>(bcur + 16 * q);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            const int ct = grp / 3, d = grp % 3;
            if (grp + 1 < NG) {
                const uint8_t *bp = bcur + ((grp + 1) / 3) * 16 * 144 + ((grp + 1) % 3) * 48;
#pragma unroll
                for (int q = 0; q < 3; ++q) rr[(grp + 1) & 1][q] = *reinterpret_cast<const v4i *>(bp + 16 * q);
            }
            __builtin_amdgcn_sched_barrier(0);
            const v4i r0 = rr[grp & 1][0], r1 = rr[grp & 1][1], r2 = rr[grp & 1][2];
            const gv8i b0 = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, -1, -1);
            const gv8i b1 = __builtin_shufflevector(r1, r2, 2, 3, 4, 5, 6, 7, -1, -1);
            const int sb = 130 + 5 * d;  // B scale: the digit's weight 2^(3 + 5 d)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[rt][ct] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_shufflevector(a[0][rt], a[0][rt], 0, 1, 2, 3, -1, -1, -1, -1), b0, acc[rt][ct], 4, 2, 0,
                                                                               127, 0, sb);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[rt][ct] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_shufflevector(a[1][rt], a[1][rt], 0, 1, 2, 3, -1, -1, -1, -1), b1, acc[rt][ct], 4, 2, 0,
                                                                               127, 0, sb);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of the next tile have landed, and the next step's weights
        asm volatile("" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int tt = 0; tt < TTW; ++tt) {
        const int tok0 = (by * 4 + wt * 2 + tt) * 16;
        const float is = p.inv_scale[tok0 + c];  // (padding rows: 0)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            float val[4][4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int j = 0; j < 4; ++j) val[rt][j] = acc[4 * hf + rt][tt][j] * is;
#if defined(BH_ABLATE) && (BH_ABLATE & 64)
            if (val[0][0] == 1234567.f) store_wave_tiles(p, val, tok0, c, g, bx * 16 + wr * 8 + 4 * hf);
#else
            store_wave_tiles(p, val, tok0, c, g, bx * 16 + wr * 8 + 4 * hf);
#endif
        }
    }
}

// f32 rows -> the f16 chain's first input: xh = f16(gamma * x) (nullable gamma: 1) and the row's (sum, sum of squares) as partial 0
// of the consumer's LayerNorm statistics.  One workgroup per row; used once per prompt (the embedding rows); every later hand-over is
// an epilogue of the kernel that produced the rows.
__global__ __launch_bounds__(256) void k_rows_to_f16(const float *__restrict__ x, const float *__restrict__ gamma, int m, int m_pad, int cols,
                                                     _Float16 *__restrict__ xh, float *__restrict__ stats, int stats_stride) {
    __shared__ double red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool live = row < m;
    double s1 = 0.0, s2 = 0.0;
    for (int i = tid; i < cols / 4; i += 256) {
        float4 v = live ? *reinterpret_cast<const float4 *>(x + (size_t)row * cols + 4 * i) : float4{0.f, 0.f, 0.f, 0.f};
        s1 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        s2 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        if (gamma) {
            const float4 gm = *reinterpret_cast<const float4 *>(gamma + 4 * i);
            v.x *= gm.x, v.y *= gm.y, v.z *= gm.z, v.w *= gm.w;
        }
        typedef _Float16 gh4 __attribute__((ext_vector_type(4)));
        bool sat = false;
        const gh4 o = {gclamp_f16(v.x, sat), gclamp_f16(v.y, sat), gclamp_f16(v.z, sat), gclamp_f16(v.w, sat)};
        gflush_sat(sat && live);
        *reinterpret_cast<gh4 *>(xh + (size_t)row * cols + 4 * i) = o;
    }
    s1 = qwave_sum_d(s1), s2 = qwave_sum_d(s2);
    if (lane == 0) red[2 * wave] = s1, red[2 * wave + 1] = s2;
    __syncthreads();
    if (tid == 0 && stats) {
        s1 = (red[0] + red[2]) + (red[4] + red[6]);
        s2 = (red[1] + red[3]) + (red[5] + red[7]);
        *reinterpret_cast<float2 *>(stats + 2 * (size_t)row) = float2{(float)s1, (float)s2};
    }
}

// ---- host side ---------------------------------------------------------------------------------
constexpr size_t kGemmCUs = 256;
thread_local GemmTileChoice g_last_gemm_tile;  // what the last launch on this thread ran (bitnet_hip_matmul_last_tile)
// Token tiles (of 16) per wave of the 4-wave 2-digit / f16 forms: 4 (64 tokens) unless
//  (a) the grid would not cover the chip twice (a short prompt, or one rank's share of a token-parallel prefill: 1024 rows x 2560
//      output rows are 160 such tiles for 512 slots): narrower tiles until it does;
//  (b) its last round would leave most slots idle: 2560 output rows x 4096 tokens are 640 tiles on 512 slots -- two rounds for
//      1.25 rounds of work.  32-token tiles run THREE to a CU (132 registers, 32 KiB of LDS): 1280 tiles on 768 slots.  Measured,
//      same box: o 74 -> 68 us, down 173 -> 160 us (quantiser included).  int8 form only: the f16 kernel's 32-token form (its staging
//      and scale work per MFMA double) LOSES 2 % of the BitNet32-F16 prefill under the same rule.
//  The f16 forms (cover = one workgroup per CU): their code expansion is a fixed cost per K step whatever the token tile, so a half-wide
//  tile is barely shorter and only pays where it fills CUs that would idle -- measured, BitNet32-F16 chain, same box: 1024 tokens 9.52 ->
//  8.42 ms, 2048 tokens 13.8 -> 13.3 ms with the 64- / 32-token tiles this rule takes (256 tokens: 4.9 ms either way).
static int gemm_token_tiles(size_t gx0, size_t m_pad, bool tail_rule, size_t cover = 2 * kGemmCUs) {
    int ttw = 4;
    while (ttw > 1 && gx0 * (m_pad / (16 * (size_t)ttw)) < cover) ttw >>= 1;
    if (ttw == 4 && tail_rule) {
        const size_t tiles = gx0 * (m_pad / 64), slots = 2 * kGemmCUs, rounds = div_ceil(tiles, slots);
        if (4 * tiles < 3 * rounds * slots) ttw = 2;  // the last round under a quarter full on average: < 75 % of the slots used
    }
    return ttw;
}
// Five row tiles per wave (320-row workgroups of 64 tokens) instead of four, for the forms whose accumulators leave the registers
// (f32 accumulators: the f16 chain, the fp6 form): taken when it needs fewer workgroup rounds x rows per workgroup on the 512 slots.
// 2560 output rows x 4096 tokens: 640 workgroups of 256 rows = two rounds (cost 2 x 4), 512 of 320 rows = exactly one (cost 5).
static bool gemm_five_tiles(size_t rows, size_t m_pad) {
    if (rows % 320 != 0) return false;
    const size_t slots = 2 * kGemmCUs, tb = m_pad / 64;
    return div_ceil(rows / 320 * tb, slots) * 5 < div_ceil(div_ceil(rows, 256) * tb, slots) * 4;
}
// Weight-stationary walk of a launch (EXPERIMENTS 4.6).  In the plain order an XCD takes a band of token tiles and walks ALL row blocks for each: the
// activation tile stays in its L2, the whole weight matrix streams through once per token tile (gate|up: 8.85 MB x 64 tiles = 566 MB of L2 misses
// per launch, served by the Infinity Cache).  Grouped, an XCD takes `wgroup` row blocks -- the largest divisor of the row-block count whose codes
// (+ scales) stay within 1.5 MiB of its 4 MiB L2 -- walks every token tile for them, then the next group: the weights are fetched about once, the
// activation tiles once per group.  Measured (rocprofv3 --pmc FETCH_SIZE, tools/pmc_fetch_wgroup.sh): gate|up 589 -> 158 MB (QK256, groups of 9
// row blocks), 752 -> 255 MB (BitNet32-F16); the 2560-row launches have few row blocks and wide activation bands and get WORSE (117 -> 188 MB),
// so only launches of at least 24 row blocks take it.  Time: neutral (the re-reads were never what the K step waits for); results: identical.
// BITNET_HIP_GEMM_WGROUP: -1 automatic (default), 0 off, n a fixed group.
static int gemm_weight_group(size_t gx, size_t row_block_rows, size_t cols, bool f16_scales, size_t budget = (size_t)3 << 19) {
    static const int mode = [] { const char *e = getenv("BITNET_HIP_GEMM_WGROUP"); return e ? atoi(e) : -1; }();
    if (mode == 0 || gx < 2) return 0;
    if (mode > 0) return gx % (size_t)mode == 0 ? mode : 0;
    if (gx < 24) return 0;
    const size_t block_bytes = row_block_rows * cols / 4 + (f16_scales ? row_block_rows * cols / 16 : 0);
    int best = 0;
    for (size_t g = 2; g <= gx; ++g)
        if (gx % g == 0 && g * block_bytes <= budget) best = (int)g;  // <= 1.5 MiB per group (2 MiB of a resident fp4 image: kFp4GroupBudget)
    return best;
}
// The resident fp4 image is twice the bytes per row block: under the 1.5 MiB rule gate|up's groups shrink from 9 row blocks to 3 and every group pass
// re-reads the 23.6 MB of fp6 planes -- rocprofv3 --pmc FETCH_SIZE per gate|up launch (fabric requests: the Infinity Cache serves them), groups of
// 0 / 2 / 3 / 6 / 9 / 18 / 27 row blocks: 1179 / 713 / 541 / 372 / 420 / 710 / 1001 MB (algorithmic: 41 MB; eight L2s each need the planes once per
// group they walk); prompt 18.80 (3) / 18.73 (6) / 18.64 (9) ms same box.  2 MiB admits 6.
constexpr size_t kFp4GroupBudget = (size_t)2 << 20;
static int gemm_ttw(int ndig, int ws) { return ws >= 2 ? (ndig <= 3 ? 2 : 1) : (ndig == 2 && !ws) ? 4 : 2; }

// 32-block scales that are f16 values go through the K = 32 path and its f16 scale tiles; the others read row-major f32 scales
static bool gemm_k32(const Weights &w) { return w.scaled && w.block_size == 32 && w.scales_f16 && w.scale_tiles_h && w.cols % 256 == 0; }
bool gemm_needs_row_major_scales(const Weights &w) { return w.scaled && !gemm_k32(w); }

bool gemm_supported(const Weights &w) {
    if (!w.tiles || w.cols % 4 != 0 || w.cols > 8192) return false;
    if (w.row_stride_bytes != div_ceil(w.cols, 256) * 64) return false;
    if (w.scaled && w.block_size != 256 && !(w.block_size == 32 && w.cols % 256 == 0)) return false;
    return true;
}

size_t gemm_workspace_bytes(size_t m, size_t cols, int ndig) {
    const size_t wg_tokens = 128, m_pad = div_ceil(m, wg_tokens) * wg_tokens;
    const size_t kp = div_ceil(cols, 256) * 256;
    const size_t plane_bytes = ndig == 2 ? kp / 256 * 576 : ndig * kp;  // 2 digits: room for the fp6 form's three base-32 digits (k_gemm_fp6)
    return m_pad * plane_bytes + div_ceil(m_pad * sizeof(float), 256) * 256 + 256;
}

// 2 digits, rows of up to 2560 columns: the wave-per-row quantiser (both digit forms: the int8 planes and the fp6 form stay bit-identical)
static bool quant_rows_w_applies(const QuantArgs &q) {
    static const int mode = [] { const char *e = getenv("BITNET_HIP_QUANT_W"); return e ? atoi(e) : 1; }();
    // from 2048 rows on: four rows per workgroup are 512 workgroups then, two per CU; below, the workgroup-per-row kernel fills more of the chip
    // (4096-token prompt 18.6 -> 18.4 ms with it, 512-token prompt 4.94 -> 5.08 ms: one box each)
    return mode != 0 && q.kp <= 2560 && q.m_pad % 4 == 0 && q.m_pad >= 2048;
}
static void launch_quant_rows_w(const QuantArgs &q, bool fp6, hipStream_t stream) {
    if (fp6) {
        hipLaunchKernelGGL(k_quant_rows_w<1>, dim3(q.m_pad / 4), dim3(256), (size_t)4 * (size_t)(q.kp / 32) * 80, stream, q);
    } else {
        hipLaunchKernelGGL(k_quant_rows_w<0>, dim3(q.m_pad / 4), dim3(256), 0, stream, q);
    }
}

template <int NDIG, int TTW>
static hipError_t launch_gemm_t(const Weights &w, const QuantArgs &q, const GemmArgs &a, hipStream_t stream) {
    const int nv = (int)div_ceil((size_t)q.kp / 4, 256);
    void (*qk)(QuantArgs) = nv <= 3 ? k_quant_rows<NDIG, 3> : k_quant_rows<NDIG, 8>;
    if (NDIG == 2 && quant_rows_w_applies(q)) launch_quant_rows_w(q, false, stream);
    else hipLaunchKernelGGL(qk, dim3(q.m_pad), dim3(256), 0, stream, q);
    constexpr int TTWS = 2;  // scaled variant: narrower token tile (f32 accumulators take the registers)
    const bool k32 = a.stiles_h != nullptr;
    const bool bs32 = (a.wscale && w.block_size == 32) || k32;  // 32-block scales: one token tile per wave (registers)
    constexpr int TT32 = NDIG <= 3 ? 2 : 1;  // 32-block scales: token tiles per wave that still fit the registers
    void (*gk)(GemmArgs) = k32 ? k_gemm_mfma<NDIG, TT32, 3> : !a.wscale ? k_gemm_mfma<NDIG, TTW, 0> : bs32 ? k_gemm_mfma<NDIG, TT32, 2> : k_gemm_mfma<NDIG, TTWS, 1>;
    int ttw = (!a.wscale && !k32) ? TTW : bs32 ? TT32 : TTWS, cw = 2;
    const bool scaled_variant = a.wscale || k32;
    // 4-wave workgroups (256 rows x TTW token tiles) bounded to two waves per SIMD: two workgroups share a CU with independent
    // barriers (one stages its next tile while the other multiplies) at the full register budget, and a 2560-row matrix x
    // 4096 tokens is 640 workgroups on 512 slots instead of 320 wide ones on 256 CUs.  Measured against the 8-wave tile (and
    // the spilling 128-register half-width form that used to take the 2560-row launches), 4096 tokens, 2 digits, quantiser
    // included: gate|up 370 -> 346 us, q|k|v 105 -> 97, o 104 -> 76, down 257 -> 188.
    if (!scaled_variant || k32) {
        gk = k32 ? k_gemm_mfma<NDIG, TT32, 3, 2, 1> : k_gemm_mfma<NDIG, TTW, 0, 2, 1>;
        cw = 1;
        if (NDIG == 2 && !scaled_variant) {
            // few activation rows (a short prompt, or one rank's share of a token-parallel prefill: 1024 rows x 2560 output rows
            // is 160 of these tiles for 512 slots): narrower token tiles until the grid covers the chip
            const size_t gx0 = div_ceil(div_ceil(w.rows, 16), 16);
            static const size_t cover = [] { const char *e = getenv("BITNET_HIP_I8_COVER"); return e ? (size_t)atoi(e) : 3 * kGemmCUs / 2; }();  // 384: swept 512 / 384 / 256 / 128 at 256 .. 2048 tokens, QK256 prompt -1 .. -3.5 % against 512
            ttw = gemm_token_tiles(gx0, q.m_pad, true, cover);
            if (ttw == 2) gk = k_gemm_mfma<2, 2, 0, 2, 1>;
            if (ttw == 1) gk = k_gemm_mfma<2, 1, 0, 2, 1>;
        }
    }
    const size_t lds = (size_t)cw * NDIG * ttw * 16 * (k32 ? kColStride : 256) * (scaled_variant ? 1 : 2) + (k32 ? (size_t)4 * cw * 4096 : 0);  // unscaled: double-buffered; K = 32: + tile staging
    {
        // once per kernel; entry points may run concurrently (Send + Sync), so the set is guarded like its twin in kernels_mfma.hip
        static std::mutex raised_mu;
        static std::unordered_set<const void *> raised;
        std::lock_guard<std::mutex> lk(raised_mu);
        if (!raised.count((const void *)gk)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            raised.insert((const void *)gk);
        }
    }
    g_last_gemm_tile = GemmTileChoice{NDIG, 16 * ttw, 4 * cw, k32 ? 3 : !a.wscale ? 0 : bs32 ? 2 : 1};
    const unsigned gx = (unsigned)div_ceil(div_ceil(w.rows, 16), 16), gy = (unsigned)(q.m_pad / (16 * cw * ttw));
    GemmArgs aw = a;
    if (!scaled_variant) aw.wgroup = gemm_weight_group(gx, 256, w.cols, false);
    hipLaunchKernelGGL(gk, dim3(gx, gy), dim3(256 * cw), lds, stream, aw);
    return hipGetLastError();
}

// k_gemm_f16a builds (code value) x (block scale) in f16: only code maps whose four values are in -2 .. 2 (every map of the
// reference: their f16 images have a zero low byte, lut_f16_hi) and scales whose double is finite in f16 take it; any other
// matrix (bitnet_hip_weights_upload_coded accepts an arbitrary int8 code map) keeps the exact int8 digit form.
static bool lut_fits_f16w(uint32_t lut) {
    for (int c = 0; c < 4; ++c) {
        const int v = (int)(int8_t)((lut >> (8 * c)) & 0xffu);
        if (v < -2 || v > 2) return false;
    }
    return true;
}
// code map values -> the high bytes of their f16 images (every value of the reference's maps, -2 .. 2, has a zero low byte)
static uint32_t lut_f16_hi(uint32_t lut) {
    uint32_t out = 0;
    for (int c = 0; c < 4; ++c) {
        const int v = (int)(int8_t)((lut >> (8 * c)) & 0xffu);
        const uint32_t hi = v == 0 ? 0x00u : v == 1 ? 0x3Cu : v == -1 ? 0xBCu : v == 2 ? 0x40u : v == -2 ? 0xC0u : 0xFFu;
        out |= hi << (8 * c);
    }
    return out;
}

// f16 activations on the f16 matrix cores (k_gemm_f16a): BitNet32-F16 at 2 digits; QK256 only on request (BITNET_HIP_GEMM_F16A=1:
// measured 5-13 % slower than the int8 digit form there -- twice the expansion VALU for the same MFMA count)
static hipError_t launch_gemm_f16(const Weights &w, const QuantArgs &q, const GemmArgs &a, hipStream_t stream) {
    const int nv = (int)div_ceil((size_t)q.kp / 4, 256);
    void (*qk)(QuantArgs) = nv <= 3 ? k_quant_rows_f16<3> : k_quant_rows_f16<8>;
    hipLaunchKernelGGL(qk, dim3(q.m_pad), dim3(256), 0, stream, q);
    // token tile: 64 (TTW 4) while the grid still covers the chip twice over, else narrower (short prompts, one rank's share)
    const size_t gx0 = div_ceil(div_ceil(w.rows, 16), 16);
    const int ttw = gemm_token_tiles(gx0, q.m_pad, false, kGemmCUs);
    const bool fmt1 = a.stiles_h != nullptr;
    void (*fk)(GemmArgs, uint32_t) = fmt1 ? (ttw == 4 ? k_gemm_f16a<1, 4> : ttw == 2 ? k_gemm_f16a<1, 2> : k_gemm_f16a<1, 1>)
                                          : (ttw == 4 ? k_gemm_f16a<0, 4> : ttw == 2 ? k_gemm_f16a<0, 2> : k_gemm_f16a<0, 1>);
    {
        static std::mutex f_mu;
        static std::unordered_set<const void *> f_raised;
        std::lock_guard<std::mutex> lk(f_mu);
        if (!f_raised.count((const void *)fk)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            f_raised.insert((const void *)fk);
        }
    }
    g_last_gemm_tile = GemmTileChoice{2, 16 * ttw, 4, fmt1 ? 4 : 5};
    hipLaunchKernelGGL(fk, dim3((unsigned)gx0, (unsigned)(q.m_pad / (16 * ttw))), dim3(256), (size_t)2 * ttw * 16 * 512, stream, a, lut_f16_hi(w.lut));
    return hipGetLastError();
}

// the fp6 form (k_gemm_fp6) takes unscaled matrices whose code map fits fp4 (every map of the reference: values in -2 .. 2)
static uint32_t lut_fp4(uint32_t lut) {
    uint32_t out = 0;
    for (int c = 0; c < 4; ++c) {
        const int v = (int)(int8_t)((lut >> (8 * c)) & 0xffu);
        const uint32_t nib = v == 0 ? 0x0u : v == 1 ? 0x2u : v == -1 ? 0xAu : v == 2 ? 0x4u : 0xCu;  // e2m1: 1.0 = 0b0010, 2.0 = 0b0100
        out |= nib << (8 * c);
    }
    return out;
}
int gemm_fp6_mode() {
    static const int mode = [] { const char *e = getenv("BITNET_HIP_GEMM_FP6"); return e ? atoi(e) : 0; }();
    return mode;
}
bool gemm_fp6_supported(const Weights &w) { return gemm_supported(w) && !w.scaled && lut_fits_f16w(w.lut); }

// ---- the resident fp4 image of a matrix (Weights::tiles4): built once (at upload when the host asks: bitnet_hip_weights_fp4_image; else by the
// first fp6-form launch), under the handle's mutex; 4 bits per weight beside the 2-bit streaming tiles, which stay the decode path's copy.
// BITNET_HIP_FP4_RESIDENT=0 keeps the in-loop expansion (no image is ever built).
bool gemm_fp4_resident_enabled() {
    static const int mode = [] { const char *e = getenv("BITNET_HIP_FP4_RESIDENT"); return e ? atoi(e) : 1; }();
    return mode != 0;
}
hipError_t ensure_fp4_image(Weights &w, hipStream_t stream) {
    if (!gemm_fp6_supported(w)) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(*w.mu);
    if (w.tiles4) return hipSuccess;
    const size_t n_tiles = div_ceil(w.rows, 16), nblk = div_ceil(w.cols, 256), total16 = n_tiles * nblk * 64;
    uint8_t *img = nullptr;
    hipError_t e = hipMalloc((void **)&img, total16 * 32);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_retile_fp4, dim3((unsigned)div_ceil(total16, 256)), dim3(256), 0, stream, w.tiles, img, lut_fp4(w.lut), total16);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);  // other threads / streams may launch on the image the moment the lock is gone
    if (e != hipSuccess) {
        (void)hipFree(img);
        return e;
    }
    w.tiles4 = img;
    return hipSuccess;
}
void drop_fp4_image(Weights &w) {
    std::lock_guard<std::mutex> lk(*w.mu);
    if (w.tiles4) (void)hipFree(w.tiles4);
    w.tiles4 = nullptr;
}
// would launch_gemm_mfma send this call to k_gemm_fp6? (the ABI asks before the launch, to build the image outside it)
bool gemm_takes_fp6(const Weights &w, const GemvFusion &fu, int ndig) {
    if (fu.int8_form || ndig != 2 || !gemm_fp6_supported(w)) return false;
    if (gemm_k32(w)) return false;
    static const int f16a_mode = [] { const char *e = getenv("BITNET_HIP_GEMM_F16A"); return e ? atoi(e) : 0; }();
    if (f16a_mode && w.cols % 256 == 0) return false;
    return fu.fp6_form || gemm_fp6_mode();
}

static hipError_t launch_gemm_fp6(const Weights &w, const QuantArgs &q, const GemmArgs &a, hipStream_t stream) {
    const int nv = (int)div_ceil((size_t)q.kp / 4, 256);
    void (*qk)(QuantArgs) = nv <= 3 ? k_quant_rows<2, 3, 1> : k_quant_rows<2, 8, 1>;
    if (quant_rows_w_applies(q)) launch_quant_rows_w(q, true, stream);
    else hipLaunchKernelGGL(qk, dim3(q.m_pad), dim3(256), (size_t)(q.kp / 32) * 144 + (size_t)(q.kp / 256) * 576, stream, q);  // LDS: the row's integers (144 bytes per 32 columns) + its packed image
    size_t gx0 = div_ceil(div_ceil(w.rows, 16), 16);
    static const size_t cover = [] { const char *e = getenv("BITNET_HIP_FP6_COVER"); return e ? (size_t)atoi(e) : 2 * kGemmCUs; }();  // (developer sweep of the token tile)
    int ttw = gemm_token_tiles(gx0, q.m_pad, false, cover);
    const bool rt5 = ttw == 4 && !a.silu_mul && w.rows % 4 == 0 && gemm_five_tiles(w.rows, q.m_pad);
    if (rt5) gx0 = w.rows / 320;
    else ttw = gemm_token_tiles(gx0, q.m_pad, true, cover);
    const bool res = a.tiles4 != nullptr;
    void (*fk)(GemmArgs, uint32_t) = res ? (rt5 ? k_gemm_fp6<4, 5, 1> : ttw == 4 ? k_gemm_fp6<4, 4, 1> : ttw == 2 ? k_gemm_fp6<2, 4, 1> : k_gemm_fp6<1, 4, 1>)
                                         : (rt5 ? k_gemm_fp6<4, 5> : ttw == 4 ? k_gemm_fp6<4> : ttw == 2 ? k_gemm_fp6<2> : k_gemm_fp6<1>);
    {
        static std::mutex f_mu;
        static std::unordered_set<const void *> f_raised;
        std::lock_guard<std::mutex> lk(f_mu);
        if (!f_raised.count((const void *)fk)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            f_raised.insert((const void *)fk);
        }
    }
    g_last_gemm_tile = GemmTileChoice{2, 16 * ttw, 4, 6, rt5 ? 80 : 64, res ? 1 : 0};
    GemmArgs aw = a;
    aw.wgroup = gemm_weight_group(gx0, rt5 ? 320 : 256, res ? 2 * w.cols : w.cols, false, res ? kFp4GroupBudget : (size_t)3 << 19);  // (the image is twice the bytes per row block)
    static const int fp6w_mode = [] { const char *e = getenv("BITNET_HIP_GEMM_FP6W"); return e ? atoi(e) : 1; }();
    if (fp6w_mode && res && ttw == 4 && !rt5 && w.rows % 256 == 0 && (size_t)div_ceil(w.rows, 16) * (w.cols / 256) * 2048 < ((size_t)1 << 31)) {
        static const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_fp6w), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  // (72 KiB of tile buffers; once, thread-safe)
        if (raised != hipSuccess) return raised;
        g_last_gemm_tile.wave_rows = 128;  // the 2 x 2 arrangement: a wave owns 128 rows x 32 tokens of the 256 x 64 tile
        hipLaunchKernelGGL(k_gemm_fp6w, dim3((unsigned)gx0, (unsigned)(q.m_pad / 64)), dim3(256), (size_t)2 * 64 * 576, stream, aw);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(fk, dim3((unsigned)gx0, (unsigned)(q.m_pad / (16 * ttw))), dim3(256), (size_t)2 * ttw * 16 * 576, stream, aw, lut_fp4(w.lut));
    return hipGetLastError();
}

// ---- the f16 activation chain: every projection input is an f16 matrix its PRODUCER wrote (no quantiser kernel between the launches) ----
bool gemm_f16_chain_supported(const Weights &w) {
    if (!w.tiles || w.cols % 256 != 0 || w.rows % 256 != 0 || w.cols > 8192) return false;
    if (w.row_stride_bytes != div_ceil(w.cols, 256) * 64 || !lut_fits_f16w(w.lut)) return false;
    if (w.scaled) return gemm_k32(w) && w.scales_f16_x2_finite;
    return true;
}

hipError_t launch_rows_to_f16(const float *x, const float *gamma, size_t m, size_t cols, void *xh, float *stats, hipStream_t stream) {
    if (cols % 4 != 0 || m == 0) return hipErrorInvalidValue;
    const size_t m_pad = div_ceil(m, 64) * 64;
    hipLaunchKernelGGL(k_rows_to_f16, dim3((unsigned)m_pad), dim3(256), 0, stream, x, gamma, (int)m, (int)m_pad, (int)cols, static_cast<_Float16 *>(xh), stats,
                       (int)m_pad);
    return hipGetLastError();
}

hipError_t launch_gemm_f16_chain(const Weights &w, const GemmF16Io &io, size_t m, hipStream_t stream) {
    if (!gemm_f16_chain_supported(w) || m == 0 || !io.xh) return hipErrorInvalidValue;
    if (io.stats_in && !(w.ln_g && io.n_stats > 0)) return hipErrorInvalidValue;
    if (io.silu_mul && (!w.paired || io.residual)) return hipErrorInvalidValue;
    const size_t m_pad = div_ceil(m, 64) * 64;
    GemmArgs a;
    a.tiles = w.tiles;
    a.stiles_h = w.scaled ? w.scale_tiles_h : nullptr;
    a.rows = (int)w.rows;
    a.cols = (int)w.cols;
    a.nblk = (int)(w.cols / 256);
    a.lut = w.lut;
    a.planes = static_cast<const int8_t *>(io.xh);
    a.inv_scale = nullptr;
    a.y = io.y;
    a.m = (int)m;
    a.residual = io.residual;
    a.wscale = nullptr;
    a.silu_mul = io.silu_mul ? 1 : 0;
    a.stats_in = io.stats_in;
    a.n_stats = io.n_stats;
    a.stats_stride = (int)m_pad;
    a.ln_eps = io.ln_eps;
    a.ln_g = w.ln_g;
    a.yh = static_cast<_Float16 *>(io.yh);
    a.gamma_out = io.gamma_out;
    a.stats_out = io.stats_out;
#ifdef BH_STAMPS
    a.stamps = g_mfma_stamps;
#endif
    size_t gx0 = w.rows / 256;
    const int ttw = gemm_token_tiles(gx0, m_pad, false, kGemmCUs);
    const bool fmt1 = w.scaled;
    const bool rt5 = ttw == 4 && !io.silu_mul && gemm_five_tiles(w.rows, m_pad);
    if (rt5) gx0 = w.rows / 320;
    void (*fk)(GemmArgs, uint32_t) = rt5    ? (fmt1 ? k_gemm_f16a<1, 4, 1, 5> : k_gemm_f16a<0, 4, 1, 5>)
                                     : fmt1 ? (ttw == 4 ? k_gemm_f16a<1, 4, 1> : ttw == 2 ? k_gemm_f16a<1, 2, 1> : k_gemm_f16a<1, 1, 1>)
                                            : (ttw == 4 ? k_gemm_f16a<0, 4, 1> : ttw == 2 ? k_gemm_f16a<0, 2, 1> : k_gemm_f16a<0, 1, 1>);
    if (io.qb_out) {  // the outputs also leave as QB32 rows (the next fp6-form matmul's input): 64-token tiles only, no silu pairing, K' = rows % 256 == 0
        if (ttw != 4 || io.silu_mul || w.rows % 256 != 0) return hipErrorInvalidValue;
        fk = rt5 ? (fmt1 ? k_gemm_f16a<1, 4, 2, 5> : k_gemm_f16a<0, 4, 2, 5>) : (fmt1 ? k_gemm_f16a<1, 4, 2> : k_gemm_f16a<0, 4, 2>);
        a.qb_nblk_out = (int)(w.rows / 256);
        a.qb_out = static_cast<uint8_t *>(io.qb_out);
    }
    {
        static std::mutex f_mu;
        static std::unordered_set<const void *> f_raised;
        std::lock_guard<std::mutex> lk(f_mu);
        if (!f_raised.count((const void *)fk)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            f_raised.insert((const void *)fk);
        }
    }
    g_last_gemm_tile = GemmTileChoice{2, 16 * ttw, 4, fmt1 ? 4 : 5, rt5 ? 80 : 64};
    const size_t lds = (size_t)2 * ttw * 16 * 512 + (size_t)ttw * 16 * 8 + 4096;  // two tile buffers + the tokens' (mean, 1 / denom) + the statistics scratch
    // Wide launches (gate|up: 54 row blocks): 128-token workgroups (k_gemm_f16h: half the code expansion per MFMA) on as many row blocks as fill WHOLE rounds
    // of the 512 slots, this kernel's 64-token workgroups on the rest -- 13824 rows x 4096 tokens: 48 x 32 = 1536 workgroups = 3 rounds, then 6 x 64 = 384
    // half-length ones (as one 128-token grid it would be 3.375 rounds: the 16 % idle tail would cost more than the expansion saves).  Same arithmetic per
    // output element whichever kernel computes it (k-order, f32 accumulation): bit-identical to the one-kernel launch.  BITNET_HIP_GEMM_F16H=0: off.
    static const int f16h_mode = [] { const char *e = getenv("BITNET_HIP_GEMM_F16H"); return e ? atoi(e) : 1; }();
    const size_t slots = 2 * kGemmCUs;
    if (f16h_mode && ttw == 4 && !rt5 && !io.qb_out && m_pad % 128 == 0 && w.rows * w.cols / 4 < ((size_t)1 << 31)) {
        const size_t t128 = m_pad / 128;
        size_t n_a = (gx0 * t128 / slots) * slots / t128;  // row blocks whose 128-token workgroups fill whole rounds
        // a launch whose 128-token workgroups fit ONE round that is at least three quarters full (q|k|v: 15 x 32 = 480 of 512 slots, against 960 = 1.9 rounds of
        // 64-token ones: a 128-token workgroup takes 1.85 x a 64-token one) goes to k_gemm_f16h whole
        if (gx0 * t128 <= slots && 4 * gx0 * t128 >= 3 * slots) n_a = gx0;
        if (gx0 < 24 && n_a != gx0) n_a = 0;
        if (n_a >= gx0 / 2 && n_a > 0) {
            void (*hk)(GemmArgs, uint32_t) = fmt1 ? k_gemm_f16h<1> : k_gemm_f16h<0>;
            {
                static std::mutex h_mu;
                static std::unordered_set<const void *> h_raised;
                std::lock_guard<std::mutex> lk(h_mu);
                if (!h_raised.count((const void *)hk)) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(hk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e != hipSuccess) return e;
                    h_raised.insert((const void *)hk);
                }
            }
            GemmArgs aa = a;
            aa.wgroup = gemm_weight_group(n_a, 256, w.cols, fmt1);
            hipLaunchKernelGGL(hk, dim3((unsigned)n_a, (unsigned)t128), dim3(256), (size_t)2 * 128 * 256 + 128 * 8 + 4096, stream, aa, lut_f16_hi(w.lut));
            g_last_gemm_tile = GemmTileChoice{2, 128, 4, fmt1 ? 4 : 5, 64};
            if (n_a == gx0) return hipGetLastError();
            a.bx_off = (int)n_a;
            a.wgroup = 0;
            hipLaunchKernelGGL(fk, dim3((unsigned)(gx0 - n_a), (unsigned)(m_pad / 64)), dim3(256), lds, stream, a, lut_f16_hi(w.lut));
            return hipGetLastError();
        }
    }
    a.wgroup = gemm_weight_group(gx0, rt5 ? 320 : 256, w.cols, fmt1);
    hipLaunchKernelGGL(fk, dim3((unsigned)gx0, (unsigned)(m_pad / (16 * ttw))), dim3(256), lds, stream, a, lut_f16_hi(w.lut));
    return hipGetLastError();
}

size_t qb32_bytes(size_t m, size_t cols) {
    const size_t m_pad = div_ceil(m, 64) * 64, nblk = div_ceil(cols, 256);
    return m_pad * nblk * kQbRec + 256;
}

hipError_t launch_rows_to_qb32(const float *x, const float *gamma, size_t m, size_t cols, void *qb, float *stats, hipStream_t stream) {
    if (cols % 256 != 0 || m == 0 || cols > 8192) return hipErrorInvalidValue;
    const size_t m_pad = div_ceil(m, 64) * 64, nblk = cols / 256;
    uint8_t *planes = static_cast<uint8_t *>(qb);
    (void)nblk;
    hipLaunchKernelGGL(k_rows_to_qb32, dim3((unsigned)m_pad), dim3(256), (cols / 32) * 144, stream, x, gamma, (int)m, (int)cols, planes, stats);
    return hipGetLastError();
}

bool gemm_qb32_supported(const Weights &w) { return gemm_fp6_supported(w) && w.rows % 256 == 0 && w.cols % 256 == 0; }

hipError_t launch_gemm_qb32(const Weights &w, const GemmF16Io &io, size_t m, hipStream_t stream) {
    if (!gemm_qb32_supported(w) || m == 0 || !io.xh || io.qb_out) return hipErrorInvalidValue;
    if (io.stats_in && !(w.ln_g && io.n_stats > 0)) return hipErrorInvalidValue;
    if (io.silu_mul && (!w.paired || io.residual)) return hipErrorInvalidValue;
    const size_t m_pad = div_ceil(m, 64) * 64, nblk = w.cols / 256;
    GemmArgs a;
    a.tiles = w.tiles;
    a.tiles4 = gemm_fp4_resident_enabled() ? w.tiles4 : nullptr;
    a.stiles_h = nullptr;
    a.rows = (int)w.rows;
    a.cols = (int)w.cols;
    a.nblk = (int)nblk;
    a.lut = w.lut;
    a.planes = static_cast<const int8_t *>(io.xh);
    a.inv_scale = nullptr;
    a.y = io.y;
    a.m = (int)m;
    a.residual = io.residual;
    a.wscale = nullptr;
    a.silu_mul = io.silu_mul ? 1 : 0;
    a.stats_in = io.stats_in;
    a.n_stats = io.n_stats;
    a.stats_stride = (int)m_pad;
    a.ln_eps = io.ln_eps;
    a.ln_g = w.ln_g;
    a.yh = static_cast<_Float16 *>(io.yh);
    a.gamma_out = io.gamma_out;
    a.stats_out = io.stats_out;
    size_t gx0 = w.rows / 256;
    const int ttw = gemm_token_tiles(gx0, m_pad, false);
    const bool rt5 = false;  // (the 320-row form of this variant does not fit 256 registers: hipcc spills 80 bytes; the model's QB32 consumers -- 3840 and 13824 rows -- take four tiles anyway)
    const bool res = a.tiles4 != nullptr;
    void (*fk)(GemmArgs, uint32_t) = res ? (ttw == 4 ? k_gemm_fp6<4, 4, 1, 1> : ttw == 2 ? k_gemm_fp6<2, 4, 1, 1> : k_gemm_fp6<1, 4, 1, 1>)
                                         : (ttw == 4 ? k_gemm_fp6<4, 4, 0, 1> : ttw == 2 ? k_gemm_fp6<2, 4, 0, 1> : k_gemm_fp6<1, 4, 0, 1>);
    {
        static std::mutex f_mu;
        static std::unordered_set<const void *> f_raised;
        std::lock_guard<std::mutex> lk(f_mu);
        if (!f_raised.count((const void *)fk)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            f_raised.insert((const void *)fk);
        }
    }
    g_last_gemm_tile = GemmTileChoice{2, 16 * ttw, 4, 8, rt5 ? 80 : 64, res ? 1 : 0};  // scale_mode 8: the fp6 form on QB32 rows
    a.wgroup = gemm_weight_group(gx0, rt5 ? 320 : 256, res ? 2 * w.cols : w.cols, false, res ? kFp4GroupBudget : (size_t)3 << 19);
    hipLaunchKernelGGL(fk, dim3((unsigned)gx0, (unsigned)(m_pad / (16 * ttw))), dim3(256), (size_t)2 * ttw * 16 * (576 + 8) + (size_t)ttw * 16 * 8 + 4096, stream, a, lut_fp4(w.lut));
    return hipGetLastError();
}

hipError_t launch_gemm_mfma(const Weights &w, const float *x, float *y, size_t m, const GemvFusion &fu, int ndig,
                            void *workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!gemm_supported(w) || (ndig != 2 && ndig != 3 && ndig != 4)) return hipErrorInvalidValue;
    if (fu.fp6_form && (fu.int8_form || ndig != 2 || !gemm_fp6_supported(w))) return hipErrorInvalidValue;
    if (workspace_bytes < gemm_workspace_bytes(m, w.cols, ndig) || !workspace) return hipErrorInvalidValue;
    const bool k32 = gemm_k32(w);
    const int ws_mode = !w.scaled ? 0 : w.block_size == 32 ? 2 : 1;
    if (gemm_needs_row_major_scales(w) && !w.scales) return hipErrorInvalidValue;  // the caller materialises them (ensure_reference)
    const size_t wg_tokens = (size_t)32 * gemm_ttw(ndig, ws_mode), m_pad = div_ceil(m, wg_tokens) * wg_tokens;
    QuantArgs q;
    q.x = x;
    q.m = (int)m;
    q.m_pad = (int)m_pad;
    q.cols = (int)w.cols;
    q.kp = (int)(div_ceil(w.cols, 256) * 256);
    q.ln_gamma = fu.ln_gamma;
    q.ln_eps = fu.ln_eps;
    q.x_f16 = fu.x_f16 ? 1 : 0;
    uint8_t *ws = static_cast<uint8_t *>(workspace);
    ws = reinterpret_cast<uint8_t *>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    q.inv_scale = reinterpret_cast<float *>(ws);
    q.planes = reinterpret_cast<int8_t *>(ws + div_ceil(m_pad * 4, 256) * 256);
    GemmArgs a;
    a.tiles = w.tiles;
    a.stiles_h = k32 ? w.scale_tiles_h : nullptr;
    a.rows = (int)w.rows;
    a.cols = (int)w.cols;
    a.nblk = (int)div_ceil(w.cols, 256);
    a.lut = w.lut;
    a.planes = q.planes;
    a.inv_scale = q.inv_scale;
    a.y = y;
    a.m = (int)m;
    a.residual = fu.residual;
    a.wscale = k32 ? nullptr : w.scales;  // per 256-block or per 32-block (row-major [rows, cols / block])
    a.silu_mul = fu.silu_mul ? 1 : 0;
    a.tiles4 = gemm_fp4_resident_enabled() && !fu.fp6_expand ? w.tiles4 : nullptr;  // (set before the launch by the ABI: ensure_fp4_image)
    const bool takes_f16 = !fu.int8_form && ndig == 2 && k32 && lut_fits_f16w(w.lut) && w.scales_f16_x2_finite;
    if (fu.x_f16 || fu.y_f16) {  // f16 hand-over: the int8 digit form's quantiser reads f16 rows; silu * up goes out as f16 rows
        const bool f16_form = takes_f16;
        if (f16_form || (fu.y_f16 && (!fu.silu_mul || ((w.rows >> 1) & 3) != 0))) return hipErrorInvalidValue;
        if (fu.y_f16) a.yh = reinterpret_cast<_Float16 *>(y), a.y = nullptr;
    }
    if (takes_f16) return launch_gemm_f16(w, q, a, stream);  // BitNet32-F16 at f16 activation precision: the f16 matrix cores
    if (!fu.int8_form) {
        static const int f16a_mode = [] { const char *e = getenv("BITNET_HIP_GEMM_F16A"); return e ? atoi(e) : 0; }();
        if (f16a_mode && ndig == 2 && !w.scaled && lut_fits_f16w(w.lut) && w.cols % 256 == 0)
            return launch_gemm_f16(w, q, a, stream);
        // unscaled matrices at 2 digits: the same integer on the fp6 x fp4 MFMA, on request (flag) or by BITNET_HIP_GEMM_FP6=1
        if (gemm_takes_fp6(w, fu, ndig)) return launch_gemm_fp6(w, q, a, stream);
    }
    if (ndig == 2) return launch_gemm_t<2, 4>(w, q, a, stream);
    if (ndig == 3) return launch_gemm_t<3, 2>(w, q, a, stream);
    return launch_gemm_t<4, 2>(w, q, a, stream);
}

}  // namespace bitnet_hip
