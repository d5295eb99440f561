// kernels_mfma.hip -- the streaming I2_S / QK256 GEMV for MI355X (gfx950).
//
// Why matrix cores for a batch-1 GEMV (DESIGN.md "VALU budget"): at 2 bits per
// weight the HBM roofline delivers 32 weights per byte, i.e. ~57 weights per clock
// per CU at 8 TB/s, while a CU retires 128 vector lane-ops per clock.  Unpack +
// convert + FMA on the vector ALUs costs >= 2.75 lane-ops per weight
// (kernels_valu.hip) -- above the roofline time.  Here the vector ALUs only expand
// 2-bit codes to int8 (11 instructions per 16 weights, v_perm_b32 as a 4-entry LUT)
// and v_mfma_i32_16x16x64_i8 does every multiply-add.
//
// Exactness: activations are converted once per launch to 30-bit fixed point with
// one power-of-two scale per wave's K range (q = floor(x * 2^(29-E) + 1/2), |q| <= 2^30)
// and split into four balanced base-256 digits d0..d3 in [-128,127].  The four
// digit planes are four B-matrix columns of the MFMA, so
//     sum_k w[r,k] * x[k]  =  2^(E-29) * sum_d 256^d * (sum_k w[r,k] * d_d[k])
// with every inner sum an exact int32.  Only the final 4-term combine rounds
// (<= 2 ulp): closer to the real-number result than the reference's own f32 loops,
// and bit-reproducible for any tiling / K split / sharding.  Elements more than 2^6
// below the row maximum are rounded at 2^-30 of that maximum.
//
// Layout: the codes are re-tiled once at upload into [row tile][256-col block]
// tiles of 1 KiB = one coalesced global_load_dwordx4 per wave (lane l: row l&15,
// 16-byte segment l>>4 of the 64-byte QK256 block), feeding four MFMAs.  Inside
// each dword the sixteen 2-bit fields are transposed 4x4 so that
// (w >> 2i) & 0x03030303 yields elements 4i..4i+3 in byte order, i.e. the A operand
// comes out in natural K order and the activation digit planes are stored in
// natural order too.
//
// Launch latency decides this kernel, not bandwidth ("Latency work"): one round of workgroups
// (<= 256), every load of a wave requested up front (the whole K range of a wave lives in registers: RING
// 1-KiB tiles), the wave index in an SGPR so that index math runs on the scalar unit, ~650 VALU instructions
// per wave between "activations arrived" and "MFMA loop done", long K ranges spread over the SIMDs.
#include <cstdlib>
#include <mutex>
#include <unordered_set>

#include "common.hpp"
#include "qact.hpp"

namespace bitnet_hip {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4fl __attribute__((ext_vector_type(4)));

// Weight bytes are read ONCE per launch by ONE CU: non-temporal loads keep them from displacing
// the activation vectors in L2 / Infinity Cache and land sooner (MI355X_MICROARCH.md, nt-weights).
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint4 load_nt16(const void *p) {
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
    return uint4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ float4 load_nt16f(const float *p) {
    const v4fl v = __builtin_nontemporal_load(reinterpret_cast<const v4fl *>(p));
    return float4{v[0], v[1], v[2], v[3]};
}

// In-kernel time stamps (s_memrealtime, 100 MHz) for the diagnostic build only; the
// production library is compiled without BH_STAMPS and carries none of this.
#ifdef BH_STAMPS
#define BH_STAMP(i)                                                                    \
    do {                                                                               \
        if (p.stamps && threadIdx.x == 0) {                                            \
            p.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
            if ((i) == 0) p.stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memtime(); \
            if ((i) == 5) p.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime(); \
        }                                                                              \
    } while (0)
#else
#define BH_STAMP(i) \
    do {            \
    } while (0)
#endif

constexpr int kRing = 5;   // weight tiles (1 KiB each) a wave keeps in flight
constexpr int kNVMAX = 4;  // float4 LayerNorm-statistics vectors per thread (512 threads): K <= 8192

struct MfmaArgs {
    const uint8_t *tiles;  // [n_tiles][nblk][64 lanes][16 B], fields transposed (k_retile)
    int rows, cols, nblk;  // nblk = ceil(cols / 256)
    uint32_t lut;
    int ksplit;            // 1, 2, 4 or 8 K ranges per row tile (waves of one workgroup)
    int ks_log2;           // log2(ksplit): the index math ahead of the first load uses shifts, not divisions
    double inv_cols;       // 1.0 / cols (LayerNorm statistics)
    const float *x;        // [mt, cols]
    float *y;              // [mt, rows]  (silu_mul: [mt, rows/2]); activation row blockIdx.y
    int out_rows;          // row stride of y
    const float *ln_gamma; // optional LayerNorm prologue (T:67-100 semantics)
    const float *ln_g;     // LN == 2: g_r = W[r,:] . gamma  (bitnet_hip_weights_bind_ln)
    float ln_eps;
    const float *residual; // optional: y = residual + W x
    const float *wscale;   // optional f32 scale per (row, 256-block)
    const float *stiles;   // optional f32 scales per (row, 32-block), tiled [tile][blk][kg][row][p] (k_retile_scales)
    const uint16_t *stiles_h;  // the same as f16 when every scale is an f16 value (template BS32 == 2)
    int silu_mul;          // rows are (gate tile, up tile) pairs: y = silu(gate) * up
    // MERGE: x is the decode attention output, merged here from its chunk records (GemvFusion::attn_rec)
    const float *attn_rec;
    const int *attn_pos;
    int attn_chunks_max, attn_group_log2, attn_chunk_log2;
    // the output as a QAct for the next GEMV (qact.hpp; rows % 16 == 0): records, the consumer's LayerNorm weight, statistics pairs
    uint8_t *qout;
    const float *gamma_out;
    double *stats_out;
    unsigned long long *stamps;  // diagnostic builds only
};

// ---- wave-64 reductions on DPP (no LDS crossbar): 4 row steps + 4 readlanes -------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
// maximum of a NON-NEGATIVE float over the wave: on the bit patterns as unsigned integers (same order, no NaN
// canonicalisation instructions), four DPP steps inside the rows of 16, row_bcast 15 / 31 across them, lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_max_u(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o > v ? o : v;
}
__device__ __forceinline__ float wave_max_f(float v) {
    uint32_t u = __float_as_uint(v);
    u = dpp_max_u<0xB1, 0xf>(u);
    u = dpp_max_u<0x4E, 0xf>(u);
    u = dpp_max_u<0x141, 0xf>(u);
    u = dpp_max_u<0x140, 0xf>(u);
    u = dpp_max_u<0x142, 0xa>(u);  // row_bcast:15 into rows 1 and 3
    u = dpp_max_u<0x143, 0xc>(u);  // row_bcast:31 into rows 2 and 3
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)u, 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    double r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 16 * i);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 16 * i);
        r[i] = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// One dword = 16 codes (already field-transposed) -> the A operand of one MFMA:
// register i, byte b <- LUT[code of element 4i+b].
__device__ __forceinline__ v4i decode16(uint32_t w, uint32_t lut) {
    v4i a;
    a[0] = (int)__builtin_amdgcn_perm(0u, lut, w & 0x03030303u);
    a[1] = (int)__builtin_amdgcn_perm(0u, lut, (w >> 2) & 0x03030303u);
    a[2] = (int)__builtin_amdgcn_perm(0u, lut, (w >> 4) & 0x03030303u);
    a[3] = (int)__builtin_amdgcn_perm(0u, lut, (w >> 6) & 0x03030303u);
    return a;
}

// 30-bit fixed point -> four balanced base-256 digits d0..d3 in [-128, 127], q = sum d_i 256^i.
// Adding 0x808080 turns the three low digits into the UNSIGNED bytes of the sum (d_i + 128, the carries are the
// adder's own), the top byte is already d3; flipping the three added bits back gives the signed digits:
//     bytes of (q + 0x00808080) ^ 0x00808080  =  d0, d1, d2, d3        (two instructions per element)
// round(v * sc) to the nearest integer in ONE instruction: v_cvt_rpi_i32_f32 = floor(x + 0.5) (ties go up
// instead of to even; |error| <= 0.5 either way, and the kernel is VALU bound)
__device__ __forceinline__ int cvt_rpi(float x) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// max(|a|, |b|, m) in one instruction (the source modifiers do the fabs)
__device__ __forceinline__ float max3_abs(float a, float b, float m) {
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(m));
    return r;
}
// a * (f16 half of h) + c in f32
__device__ __forceinline__ float fma_mix_lo(float a, uint32_t h, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ float fma_mix_hi(float a, uint32_t h, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t digits4(float v, float sc) {
    return ((uint32_t)cvt_rpi(v * sc) + 0x00808080u) ^ 0x00808080u;
}
// 4 x 4 byte transpose: element dwords e0..e3 (byte i = digit i) -> plane dwords d0..d3 (byte b = element b)
__device__ __forceinline__ void digit_planes(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, uint32_t &d0, uint32_t &d1,
                                             uint32_t &d2, uint32_t &d3) {
    const uint32_t lo01 = __builtin_amdgcn_perm(e1, e0, 0x05010400u), hi01 = __builtin_amdgcn_perm(e1, e0, 0x07030602u);
    const uint32_t lo23 = __builtin_amdgcn_perm(e3, e2, 0x05010400u), hi23 = __builtin_amdgcn_perm(e3, e2, 0x07030602u);
    d0 = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
    d1 = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
    d2 = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
    d3 = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
}

// NW waves per workgroup, RING = 256-column blocks per wave (all in flight at once), NV =
// float4 LayerNorm-statistics vectors per thread, LN = fused LayerNorm prologue, BS32 =
// 32-element block scales.  They are template parameters, and every global load below is
// UNCONDITIONAL (indices clamped, values masked afterwards), so that hipcc can count the
// outstanding loads: with a load inside any branch it falls back to s_waitcnt vmcnt(0) at the
// first use of the activations, i.e. waits for the whole weight stream before the prologue.
// LN: 0 none; 1 LayerNorm prologue on the activations; 2 the same LayerNorm applied AFTER the
// product: with g_r = sum_k W[r,k] gamma[k] precomputed once per (matrix, gamma)
// (bitnet_hip_weights_bind_ln),  sum_k W[r,k] (x_k - mean)/denom gamma_k = (sum_k W[r,k] gamma_k x_k - mean g_r) / denom,
// so the waves quantise gamma*x of their own K range straight away and the row statistics are only
// needed in the epilogue, behind the barrier that is there anyway (saves the two prologue barriers,
// the normalised row's LDS round trip and ~1.4 us per launch).
// MERGE (LN == 0): the activation row is the decode attention's output and is assembled here from the per-chunk
// records (m, l, un-normalised P.V) that k_attn_partial left in the scratch buffer -- softmax merge of up to 4
// chunks per lane, for the 4 columns the lane quantises anyway -- instead of a separate combine launch (a kernel
// boundary + a cross-XCD hand-over + its own chain: 3.2 us per layer).  Every workgroup needs the whole row, so
// every workgroup reads every live record: n_chunks x 10 KB through the CU's 64 B/clk vector-memory path, which is
// why this is for short contexts only (+3.3 % tokens/s at 3-4 chunks, break-even at 6-7).
template <int NW, int RING, int NV, int LN, int BS32, int MERGE = 0>
__global__ __launch_bounds__(NW * 64) void k_gemv_mfma(MfmaArgs p) {
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform: in an SGPR, so the tile / K-range index math and the load bases run on the scalar unit
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    // several activation rows (prompt rows of a format the tiled matmul does not take): grid.y walks them
    const float *px = p.x + (size_t)blockIdx.y * p.cols;
    float *py = p.y + (size_t)blockIdx.y * p.out_rows;
    const float *pres = p.residual ? p.residual + (size_t)blockIdx.y * p.rows : nullptr;
    constexpr int ps = RING * 256 + 16;  // plane stride: +16 B makes the B reads conflict-free
    constexpr int NP = BS32 ? 5 : 4;  // 32-block mode: a fifth, all-zero plane per wave (see the main loop)
    uint8_t *planes = lds + wave * NP * ps;                          // this wave's [NP][ps]
    double *stat = reinterpret_cast<double *>(lds + NW * NP * ps);   // [NW][2] LayerNorm sums
    float *part = reinterpret_cast<float *>(stat + 2 * NW);         // [NW][16]
    float *vbuf = part + NW * 16;                                   // [cols] normalised row (LN only)
    BH_STAMP(0);
    if (BS32) {  // zero plane, written while the first loads are in flight
#pragma unroll
        for (int j = 0; j < RING; ++j) *reinterpret_cast<uint32_t *>(planes + 4 * ps + 256 * j + 4 * lane) = 0u;
    }

    // ---- wave -> (row tile, K range of at most RING 256-column blocks) ------------------
    const int tiles_per_wg = NW >> p.ks_log2;
    const int n_tiles = (p.rows + 15) >> 4;
    int tile = blockIdx.x * tiles_per_wg + (wave >> p.ks_log2);
    tile = tile < n_tiles ? tile : n_tiles - 1;  // surplus waves redo the last tile; never stored
    // K split 8 over a block count that 8 does not divide gives ranges of n and n + 1 blocks, the longer ones at
    // kparts 3 and 7 -- waves 3 and 7 share SIMD 3 (wave i runs on SIMD i % 4).  Waves 4..7 therefore take
    // kparts 7..4: every SIMD gets at most one long range.  The partial sums are stored by kpart, so the value
    // (and its summation order) does not change.
    int kpart = wave & (p.ksplit - 1);
    if (p.ks_log2 == 3 && wave >= 4) kpart = 11 - wave;
    const int b0 = (kpart * p.nblk) >> p.ks_log2, b1 = ((kpart + 1) * p.nblk) >> p.ks_log2;
    const int nvec = p.cols >> 2;  // cols % 4 == 0

    // ---- 1. activations first (vmcnt retires in order; these come from L2) ---------------
    // byte offsets in 32 bits from a uniform base (scalar base + one VGPR offset per load); clamped to the
    // row's last float4 so that a short range / ragged row re-reads valid memory
    const uint32_t last_vec = 16u * (uint32_t)(nvec - 1), xo0 = 1024u * (uint32_t)b0 + 16u * (uint32_t)lane;
    float4 xr[RING];
    float4 sx[NV], sg[NV], gr[LN == 2 ? RING : 1];
    const float *mrec[MERGE ? RING : 1];
    int mo[MERGE ? RING : 1];
    float2 mml[MERGE ? RING : 1][4];
    float4 mov[MERGE ? RING : 1][4];
    if (LN == 1) {
        // LayerNorm needs the whole row: the workgroup reads x and gamma ONCE (each thread its
        // share) and hands the normalised row to the waves through LDS.
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + NT * i;
            const int ci = idx < nvec ? idx : nvec - 1;
            sx[i] = *reinterpret_cast<const float4 *>(px + 4 * ci);
            sg[i] = *reinterpret_cast<const float4 *>(p.ln_gamma + 4 * ci);
        }
    } else if (LN == 2) {
        // own K range of x and gamma (they gate the quantisation) and this thread's share of the whole row
        // for the statistics (only needed in the epilogue); all ahead of the weight stream
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            const uint32_t o = umin32(xo0 + 1024u * j, last_vec);
            xr[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(px) + o);
            gr[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(p.ln_gamma) + o);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
            sx[i] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(px) + umin32(16u * (uint32_t)(tid + NT * i), last_vec));
    } else if (MERGE) {
        // chunks 0..3 of this lane's head are requested before the position (hence the live chunk count) is known;
        // dead records hold zeros or an earlier token's values (finite): only their m is masked below
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            if (b0 + j >= b1) continue;  // wave-uniform: a K range shorter than RING blocks (10 blocks over 8 waves) skips the slot
            const uint32_t kf = umin32(xo0 + 1024u * j, last_vec) >> 2;  // first of this lane's 4 columns
            const uint32_t h = kf >> 7, d = kf & 127u, kvh = h >> p.attn_group_log2, g = h & ((1u << p.attn_group_log2) - 1u);
            mrec[j] = p.attn_rec + (size_t)kvh * p.attn_chunks_max * kAttnRecFloats + 2 * g;
            mo[j] = 8 - 2 * g + g * 128 + d;  // from the (m, l) pair to the lane's 4 P.V values
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cc = c < p.attn_chunks_max ? c : p.attn_chunks_max - 1;
                mml[j][c] = *reinterpret_cast<const float2 *>(mrec[j] + (size_t)cc * kAttnRecFloats);
                mov[j][c] = *reinterpret_cast<const float4 *>(mrec[j] + (size_t)cc * kAttnRecFloats + mo[j]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < RING; ++j)
            xr[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(px) + umin32(xo0 + 1024u * j, last_vec));
    }

    // ---- 2. weight tiles: RING 1-KiB tiles in flight per wave ------------------------------
    const uint8_t *wbase = p.tiles + ((size_t)tile * p.nblk * 64 + lane) * 16;
    // 32-element block scales ride along, layout [tile][256-block][kg][row c][p] (one layout for this kernel and
    // kernels_gemvq.hip): this lane (rows 4g..4g+3 of the tile, k-group kg = r16 >> 2) needs rows 4g+i, 32-blocks
    // 2 kg + p = 8 consecutive values [i][p], shared by the 4 digit lanes
    const size_t sidx = ((size_t)tile * p.nblk * 16 + (size_t)((r16 >> 2) * 4 + g)) * 8;
    const float *sbase = BS32 == 1 ? p.stiles + sidx : nullptr;
    const uint16_t *sbase_h = BS32 == 2 ? p.stiles_h + sidx : nullptr;
    uint4 wt[RING];
    float4 s_lo[RING], s_hi[RING];
    uint4 s_h[RING];
#pragma unroll
    for (int j = 0; j < RING; ++j) {
        const int blk = b0 + j < b1 ? b0 + j : b1 - 1;  // clamped: a short range re-reads its last tile
        wt[j] = load_nt16(wbase + (size_t)blk * 1024);
        if (BS32 == 1) {
            s_lo[j] = load_nt16f(sbase + (size_t)blk * 128);
            s_hi[j] = load_nt16f(sbase + (size_t)blk * 128 + 4);
        }
        if (BS32 == 2) s_h[j] = load_nt16(sbase_h + (size_t)blk * 128);  // 8 halves = 16 B
    }
    int m_chunks = 0;
    if (MERGE) m_chunks = (*p.attn_pos + (1 << p.attn_chunk_log2)) >> p.attn_chunk_log2;  // live records, 1..4 (the caller switches to the combine kernel beyond that)
    // Every load of this wave is now requested.  Without the fence hipcc moves the activation
    // arithmetic (and its s_waitcnt) up between the weight loads, so that half of the weight stream is
    // only requested once the activations have arrived (~1 us later).
    __builtin_amdgcn_sched_barrier(0);
    BH_STAMP(1);

    if (MERGE) {
        // softmax merge of the chunk records: out = sum_c e^(m_c - M) o_c / sum_c e^(m_c - M) l_c  (k_attn_combine's value)
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            if (b0 + j >= b1) {  // skipped slot: zero digits
                xr[j] = float4{0.0f, 0.0f, 0.0f, 0.0f};
                continue;
            }
            float M = -INFINITY;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                mml[j][c].x = c < m_chunks ? mml[j][c].x : -INFINITY;
                M = fmaxf(M, mml[j][c].x);
            }
            float L = 0.0f;
            float4 a = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float e = __expf(mml[j][c].x - M);
                L += e * mml[j][c].y;
                a.x += e * mov[j][c].x, a.y += e * mov[j][c].y, a.z += e * mov[j][c].z, a.w += e * mov[j][c].w;
            }
            const float rl = 1.0f / L;
            xr[j] = float4{a.x * rl, a.y * rl, a.z * rl, a.w * rl};
        }
    }
    // ---- 3. prologue: [LayerNorm] -> fixed point -> this wave's digit planes --------------
    if (LN == 2) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            xr[j].x *= gr[j].x;
            xr[j].y *= gr[j].y;
            xr[j].z *= gr[j].z;
            xr[j].w *= gr[j].w;
        }
    }
    if (LN == 1) {
        // LayerNorm without bias, WITH mean subtraction (T:89-97; candle LayerNorm slow path):
        // (x - mean) / sqrt(mean((x - mean)^2) + eps) * gamma.  One pass: sum and sum of
        // squares in f64 (products of f32 are exact in f64).
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool in = tid + NT * i < nvec;
            const double a = in ? sx[i].x : 0.0f, b = in ? sx[i].y : 0.0f, c = in ? sx[i].z : 0.0f, d = in ? sx[i].w : 0.0f;
            s1 += (a + b) + (c + d);
            s2 += (a * a + b * b) + (c * c + d * d);
        }
        s1 = wave_sum_d(s1);
        s2 = wave_sum_d(s2);
        if (lane == 0) {
            stat[2 * wave] = s1;
            stat[2 * wave + 1] = s2;
        }
        __syncthreads();
        s1 = 0.0;
        s2 = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            s1 += stat[2 * w];
            s2 += stat[2 * w + 1];
        }
        const double mean_d = s1 / (double)p.cols;
        const double var_d = s2 / (double)p.cols - mean_d * mean_d;
        const float mean = (float)mean_d;
        const float denom = sqrtf((float)(var_d > 0.0 ? var_d : 0.0) + p.ln_eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + NT * i;
            if (idx < nvec) {
                float4 v;
                v.x = (sx[i].x - mean) / denom * sg[i].x;
                v.y = (sx[i].y - mean) / denom * sg[i].y;
                v.z = (sx[i].z - mean) / denom * sg[i].z;
                v.w = (sx[i].w - mean) / denom * sg[i].w;
                *reinterpret_cast<float4 *>(vbuf + 4 * idx) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            const int idx = (b0 + j) * 64 + lane;
            xr[j] = *reinterpret_cast<const float4 *>(vbuf + 4 * (idx < nvec ? idx : nvec - 1));
        }
    }
    float am = 0.0f;
    // wave-uniform: a full range of whole 256-column blocks needs no masking at all
    const bool ragged = b1 - b0 < RING || b1 * 64 > nvec;
    if (ragged) {  // one scalar branch, not one per block
#pragma unroll
        for (int j = 0; j < RING; ++j)
            if (!(b0 + j < b1 && (b0 + j) * 64 + lane < nvec)) xr[j] = float4{0.0f, 0.0f, 0.0f, 0.0f};  // outside the range / row
    }
#pragma unroll
    for (int j = 0; j < RING; ++j) am = max3_abs(xr[j].z, xr[j].w, max3_abs(xr[j].x, xr[j].y, am));
    am = wave_max_f(am);  // this wave's K range only: the scale is per wave
    BH_STAMP(2);
    // scale = 2^(29 - E), E = unbiased exponent of the maximum (clamped so the scale stays a
    // normal float); |x * scale| < 2^30
    int be = (int)((__float_as_uint(am) >> 23) & 0xffu);
    be = be < 32 ? 32 : be;
    const float sc = __uint_as_float((uint32_t)(283 - be) << 23);
    const float inv_s = __uint_as_float((uint32_t)(be - 29) << 23);
#pragma unroll
    for (int j = 0; j < RING; ++j) {
        uint32_t d0, d1, d2, d3;
        digit_planes(digits4(xr[j].x, sc), digits4(xr[j].y, sc), digits4(xr[j].z, sc), digits4(xr[j].w, sc), d0, d1, d2, d3);
        const int pos = 256 * j + 4 * lane;
        *reinterpret_cast<uint32_t *>(planes + 0 * ps + pos) = d0;
        *reinterpret_cast<uint32_t *>(planes + 1 * ps + pos) = d1;
        *reinterpret_cast<uint32_t *>(planes + 2 * ps + pos) = d2;
        *reinterpret_cast<uint32_t *>(planes + 3 * ps + pos) = d3;
    }
    BH_STAMP(3);

    // ---- 4. main loop: decode -> MFMA.  The planes are this wave's own: LDS executes one
    //         wave's accesses in order, no barrier needed.  Slots past the wave's range hold
    //         zero digits (and a re-read tile), so they add exact zeros. ----------------------
    // B operand of lane (col c = r16, k-group g): 16 bytes of plane c & 3.
    // 32-block mode: column c = 4*kg + d is live for k-group kg only; the other lanes read the zero plane, so the
    // loop carries no masking instructions (the kernel is VALU-issue bound)
    const uint8_t *bbase = planes + ((!BS32 || (r16 >> 2) == g) ? (r16 & 3) : 4) * ps + 64 * g;
    v4i acc = {0, 0, 0, 0};
    float facc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < RING; ++j) {
        if (j > 0 && b0 + j >= b1) continue;  // wave-uniform: a slot past this wave's range would only add exact zeros
        const uint32_t wd[4] = {wt[j].x, wt[j].y, wt[j].z, wt[j].w};
        if (BS32) {
            // 32-element blocks: column c = 4*kg + d carries digit d of k-group kg only (B is
            // zero elsewhere), so D[row][c] is the exact integer sum over 16 k's; MFMAs
            // m = 0,1 complete 32-block 2*kg, m = 2,3 complete 2*kg + 1.
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const v4i a = decode16(wd[m], p.lut);
                const v4i b = *reinterpret_cast<const v4i *>(bbase + 256 * j + 16 * m);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
                if (m & 1) {
                    float4 sv;
                    if (BS32 == 2) {
                        // f16 scales go into the fma as they are (v_fma_mix_f32: f32 * f16 + f32), no conversions;
                        // dword i = row 4g+i: (32-block 2 kg, 32-block 2 kg + 1)
                        if (m == 1) {
                            facc[0] = fma_mix_lo((float)acc[0], s_h[j].x, facc[0]);
                            facc[1] = fma_mix_lo((float)acc[1], s_h[j].y, facc[1]);
                            facc[2] = fma_mix_lo((float)acc[2], s_h[j].z, facc[2]);
                            facc[3] = fma_mix_lo((float)acc[3], s_h[j].w, facc[3]);
                        } else {
                            facc[0] = fma_mix_hi((float)acc[0], s_h[j].x, facc[0]);
                            facc[1] = fma_mix_hi((float)acc[1], s_h[j].y, facc[1]);
                            facc[2] = fma_mix_hi((float)acc[2], s_h[j].z, facc[2]);
                            facc[3] = fma_mix_hi((float)acc[3], s_h[j].w, facc[3]);
                        }
                        acc = (v4i){0, 0, 0, 0};
                        continue;
                    } else {
                        sv = m == 1 ? float4{s_lo[j].x, s_lo[j].z, s_hi[j].x, s_hi[j].z} : float4{s_lo[j].y, s_lo[j].w, s_hi[j].y, s_hi[j].w};
                    }
                    facc[0] += (float)acc[0] * sv.x;
                    facc[1] += (float)acc[1] * sv.y;
                    facc[2] += (float)acc[2] * sv.z;
                    facc[3] += (float)acc[3] * sv.w;
                    acc = (v4i){0, 0, 0, 0};
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const v4i a = decode16(wd[m], p.lut);
                const v4i b = *reinterpret_cast<const v4i *>(bbase + 256 * j + 16 * m);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
            }
            if (p.wscale) {
                // one f32 weight scale per (row, 256-block): fold this block's exact integer
                // sums into f32 accumulators (rows 4g+j of the tile)
                const int blk = b0 + j < b1 ? b0 + j : b1 - 1;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    int row = 16 * tile + 4 * g + jj;
                    row = row < p.rows ? row : p.rows - 1;
                    facc[jj] += (float)acc[jj] * p.wscale[(size_t)row * p.nblk + blk];
                }
                acc = (v4i){0, 0, 0, 0};
            }
        }
    }
    BH_STAMP(4);

    if (LN == 2) {
        // row statistics, off the critical path: their loads were the first ones issued, the sums are only
        // read behind the epilogue's barrier
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool in = tid + NT * i < nvec;
            const double a = in ? sx[i].x : 0.0f, b = in ? sx[i].y : 0.0f, c = in ? sx[i].z : 0.0f, d = in ? sx[i].w : 0.0f;
            s1 += (a + b) + (c + d);
            s2 += (a * a + b * b) + (c * c + d * d);
        }
        s1 = wave_sum_d(s1);
        s2 = wave_sum_d(s2);
        if (lane == 0) {
            stat[2 * wave] = s1;
            stat[2 * wave + 1] = s2;
        }
    }
    // ---- 5. epilogue: digits -> f32 (x this wave's 2^(E-29)), K-range reduction, store -----
    const float cw = (BS32 || r16 < 4) ? __uint_as_float((uint32_t)(127 + 8 * (r16 & 3)) << 23) * inv_s : 0.0f;
    float f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[j] = ((p.wscale || BS32) ? facc[j] : (float)acc[j]) * cw;
        f[j] += dpp_f<0xB1>(f[j]);  // lanes c ^ 1
        f[j] += dpp_f<0x4E>(f[j]);  // lanes c ^ 2: the four digit columns
        if (BS32) {                 // the four k-groups live in columns 4*kg + d
            f[j] += dpp_f<0x141>(f[j]);
            f[j] += dpp_f<0x140>(f[j]);
        }
    }
    if (r16 == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) part[(((wave >> p.ks_log2) << p.ks_log2) + kpart) * 16 + 4 * g + j] = f[j];
    }
    __syncthreads();
    // only the storing threads (the first tiles_per_wg * 16, i.e. the first wave or two) go on: the other
    // waves would repeat the row statistics' f64 arithmetic on the same SIMDs for nothing
    if (wave * 64 >= tiles_per_wg * 16) return;
    double ln_mean = 0.0, ln_rdenom = 1.0;
    if (LN == 2) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            s1 += stat[2 * w];
            s2 += stat[2 * w + 1];
        }
        const double mean_d = s1 * p.inv_cols;  // 1/cols from the host: no f64 division sequences in the kernel
        const double var_d = s2 * p.inv_cols - mean_d * mean_d;
        ln_mean = (double)(float)mean_d;                                              // the f32 mean the prologue form subtracts
        const double denom = (double)sqrtf((float)(var_d > 0.0 ? var_d : 0.0) + p.ln_eps);
        double r = __builtin_amdgcn_rcp(denom);  // v_rcp_f64 + two Newton steps: 1/denom to the last bit or so
        r = r * (2.0 - denom * r);
        ln_rdenom = r * (2.0 - denom * r);
    }
    if (!p.silu_mul) {
        if (tid < tiles_per_wg * 16) {
            const int tl = tid >> 4, r = tid & 15;
            const int row = 16 * (blockIdx.x * tiles_per_wg + tl) + r;
            if (row < p.rows) {
                float v = 0.0f;
                for (int kp = 0; kp < p.ksplit; ++kp) v += part[((tl << p.ks_log2) + kp) * 16 + r];
                if (LN == 2) v = (float)(((double)v - ln_mean * (double)p.ln_g[row]) * ln_rdenom);
                if (pres) v += pres[row];
                py[row] = v;
                if (p.qout) qact_emit(p.qout, p.stats_out, row >> 4, r, v, p.gamma_out ? v * p.gamma_out[row] : v);  // rows % 16 == 0: whole 16-lane rows get here
            }
        }
    } else {
        // rows come in (gate tile, up tile) pairs; FeedForward::forward T:756-781:
        // hidden = silu(gate) * up, silu(v) = v / (1 + exp(-v))
        const int pairs_per_wg = tiles_per_wg / 2, half_rows = p.rows / 2;
        if (tid < pairs_per_wg * 16) {
            const int pl = tid >> 4, r = tid & 15;
            const int row = 16 * (blockIdx.x * pairs_per_wg + pl) + r;
            if (row < half_rows) {
                float gv = 0.0f, uv = 0.0f;
                for (int kp = 0; kp < p.ksplit; ++kp) {
                    gv += part[(((2 * pl) << p.ks_log2) + kp) * 16 + r];
                    uv += part[(((2 * pl + 1) << p.ks_log2) + kp) * 16 + r];
                }
                if (LN == 2) {  // stored rows: (gate tile, up tile) pairs
                    const int t0 = blockIdx.x * tiles_per_wg + 2 * pl;
                    gv = (float)(((double)gv - ln_mean * (double)p.ln_g[16 * t0 + r]) * ln_rdenom);
                    uv = (float)(((double)uv - ln_mean * (double)p.ln_g[16 * (t0 + 1) + r]) * ln_rdenom);
                }
                const float hv = gv / (1.0f + expf(-gv)) * uv;
                py[row] = hv;
                if (p.qout) qact_emit(p.qout, p.stats_out, row >> 4, r, hv, p.gamma_out ? hv * p.gamma_out[row] : hv);
            }
        }
    }
    BH_STAMP(5);
}

unsigned long long *g_mfma_stamps = nullptr;  // set through bitnet_hip_debug_set_stamps (diagnostic build)

bool mfma_supported(const Weights &w) {
    if (w.cols == 0 || w.rows == 0) return false;
    if (w.cols > (size_t)kNVMAX * 2048) return false;                    // prologue register budget
    if (w.cols % 4 != 0) return false;                                   // float4 activation loads
    if (w.row_stride_bytes != div_ceil(w.cols, 256) * 64) return false;  // QK256-shaped rows
    if (w.scaled && w.block_size != 256 && w.block_size != 32) return false;  // f32 scales per 256- or 32-block
    if (w.scaled && w.block_size == 32 && w.cols % 256 != 0) return false;
    if (div_ceil(div_ceil(w.cols, 256), (size_t)(w.paired ? 4 : 8)) > (size_t)kRing) return false;  // K range per wave
    return true;
}

int mfma_pick_ksplit(size_t rows, size_t cols, bool paired, int nw) {
    const size_t n_tiles = div_ceil(rows, 16), nblk = div_ceil(cols, 256);
    const int ks_max = paired ? 4 : 8;  // K ranges per row tile (<= waves per workgroup)
    // One round of identical workgroups: the largest K split whose grid still fits the 256 CUs with
    // one workgroup each (a second workgroup on some CUs makes those the tail: 432 workgroups
    // for gate|up took 7.4 us against 5.9 us for 216), K range per wave <= kRing blocks.
    for (int ks = ks_max; ks >= 1; ks >>= 1) {
        if ((size_t)ks > nblk || div_ceil(nblk, (size_t)ks) > (size_t)kRing) continue;
        if (div_ceil(n_tiles * ks, (size_t)nw) <= 256) return ks;
    }
    // larger matrices need several rounds anyway: spread over the CUs without leaving a wave
    // fewer than ~2 tiles
    int ks = 1;
    while (ks < ks_max && (n_tiles * ks < 8 * 256 || div_ceil(nblk, ks) > (size_t)kRing) && (size_t)ks * 2 <= nblk) ks *= 2;
    return ks;
}

hipError_t launch_gemv_mfma(const Weights &w, const float *x, float *y, size_t m, const GemvFusion &fu,
                            hipStream_t stream) {
    if (!w.tiles) return hipErrorInvalidValue;
    MfmaArgs a;
    a.tiles = w.tiles;
    a.rows = (int)w.rows;
    a.cols = (int)w.cols;
    a.nblk = (int)div_ceil(w.cols, 256);
    a.lut = w.lut;
    const int nw = 8;  // (16-wave workgroups were a round-1 tuning knob that never paid: removed)
    a.ksplit = mfma_pick_ksplit(w.rows, w.cols, fu.silu_mul, nw);
    a.ks_log2 = a.ksplit == 8 ? 3 : a.ksplit == 4 ? 2 : a.ksplit == 2 ? 1 : 0;
    a.inv_cols = 1.0 / (double)w.cols;
    a.ln_gamma = fu.ln_gamma;
    a.ln_eps = fu.ln_eps;
    a.ln_g = (fu.ln_gamma && w.ln_g && w.ln_gamma_bound == fu.ln_gamma) ? w.ln_g : nullptr;
    a.wscale = (w.scaled && w.block_size == 256) ? w.scales : nullptr;
    if (w.scaled && w.block_size == 256 && !w.scales) return hipErrorInvalidValue;
    const bool bs32_any = w.scaled && w.block_size == 32;
    a.stiles = (bs32_any && !w.scales_f16) ? w.scale_tiles : nullptr;
    a.stiles_h = (bs32_any && w.scales_f16) ? w.scale_tiles_h : nullptr;
    a.silu_mul = fu.silu_mul ? 1 : 0;
    a.attn_rec = fu.attn_rec;
    a.attn_pos = fu.attn_pos;
    a.attn_chunks_max = fu.attn_chunks_max;
    a.attn_group_log2 = fu.attn_group_log2;
    a.attn_chunk_log2 = fu.attn_chunk_log2;
    a.qout = static_cast<uint8_t *>(fu.qout);
    a.gamma_out = fu.gamma_out;
    a.stats_out = fu.stats_out;
    if (fu.qout && (w.rows % 16 != 0 || m != 1 || (fu.silu_mul && w.rows % 32 != 0))) return hipErrorInvalidValue;
    a.stamps = g_mfma_stamps;
    const int tiles_per_wg = nw / a.ksplit;
    const unsigned grid = (unsigned)div_ceil(div_ceil(w.rows, 16), tiles_per_wg);
    const size_t out_rows = fu.silu_mul ? w.rows / 2 : w.rows;
    // template selection: RING = blocks per wave, NV = statistics float4 per thread
    const int ring = (int)div_ceil((size_t)a.nblk, (size_t)a.ksplit);
    const bool ln = fu.ln_gamma != nullptr, ln2 = a.ln_g != nullptr;
    const int bs32 = a.stiles ? 1 : a.stiles_h ? 2 : 0;
    if (bs32_any && !bs32) return hipErrorInvalidValue;
    const int nv = ln ? (int)div_ceil(w.cols / 4, (size_t)nw * 64) : 1;
    if (ring > kRing || nv > kNVMAX) return hipErrorInvalidValue;
    void (*kfn)(MfmaArgs) = nullptr;
#define BH_PICK(NWv, RINGv, NVv)                                                                          \
    if (nw == NWv && ring <= RINGv && nv <= NVv && !kfn)                                                  \
        kfn = ln2 ? (bs32 == 2 ? k_gemv_mfma<NWv, RINGv, NVv, 2, 2> : bs32 == 1 ? k_gemv_mfma<NWv, RINGv, NVv, 2, 1> : k_gemv_mfma<NWv, RINGv, NVv, 2, 0>) \
              : ln ? (bs32 == 2 ? k_gemv_mfma<NWv, RINGv, NVv, 1, 2> : bs32 == 1 ? k_gemv_mfma<NWv, RINGv, NVv, 1, 1> : k_gemv_mfma<NWv, RINGv, NVv, 1, 0>) \
                   : (bs32 == 2 ? k_gemv_mfma<NWv, RINGv, 1, 0, 2> : bs32 == 1 ? k_gemv_mfma<NWv, RINGv, 1, 0, 1> : k_gemv_mfma<NWv, RINGv, 1, 0, 0>);
    if (fu.attn_rec) {  // x merged from the decode attention's chunk records: the o-projection shape only
        if (ln || m != 1 || nw != 8 || ring > 2 || w.cols % 128 != 0 || !fu.attn_pos || fu.attn_chunks_max < 1) return hipErrorInvalidValue;
        kfn = bs32 == 2 ? k_gemv_mfma<8, 2, 1, 0, 2, 1> : bs32 == 1 ? k_gemv_mfma<8, 2, 1, 0, 1, 1> : k_gemv_mfma<8, 2, 1, 0, 0, 1>;
    }
    BH_PICK(8, 2, 2) BH_PICK(8, 2, 4) BH_PICK(8, 3, 2) BH_PICK(8, 3, 4) BH_PICK(8, 4, 2) BH_PICK(8, 4, 4) BH_PICK(8, 5, 2) BH_PICK(8, 5, 4)
#undef BH_PICK
    if (!kfn) return hipErrorInvalidValue;
    const int ring_t = ring <= 2 ? 2 : ring;  // the instantiated RING (LDS plane stride)
    const size_t lds = (size_t)nw * (bs32 ? 5 : 4) * (ring_t * 256 + 16) + 2 * nw * sizeof(double) + nw * 16 * sizeof(float) +
                       ((ln && !ln2) ? w.cols * sizeof(float) : 0);
    if (lds > 64 * 1024) {
        static std::mutex raised_mu;                     // launches may come from several host threads (Send + Sync)
        static std::unordered_set<const void *> raised;  // raised once per kernel, outside any capture
        std::lock_guard<std::mutex> lk(raised_mu);
        if (!raised.count((const void *)kfn)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            raised.insert((const void *)kfn);
        }
    }
    // one activation row per workgroup row of the grid (forward_qk256's row loop, T:683-691, in one launch)
    a.out_rows = (int)out_rows;
    for (size_t m0 = 0; m0 < m; m0 += 65535) {
        const size_t mc = m - m0 < 65535 ? m - m0 : 65535;
        a.x = x + m0 * w.cols;
        a.y = y + m0 * out_rows;
        a.residual = fu.residual ? fu.residual + m0 * w.rows : nullptr;
        hipLaunchKernelGGL(kfn, dim3(grid, (unsigned)mc), dim3(nw * 64), lds, stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- tiled layout: [row tile][256-col block][lane][16 B], 2-bit fields transposed -----
// dword in : field f = 4b + i (bits 2f..2f+1) holds element 4b + i  (byte b, slot i)
// dword out: field 4b + i holds element 4i + b, so that (w >> 2i) & 0x03030303 puts
//            elements 4i .. 4i+3 into bytes 0..3.
__device__ __forceinline__ uint32_t transpose_fields(uint32_t w) {
    uint32_t o = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) o |= ((w >> (2 * (4 * i + b))) & 3u) << (2 * (4 * b + i));
    return o;
}

__global__ void k_retile(const uint8_t *__restrict__ codes, size_t row_stride, int rows, int nblk,
                         uint8_t *__restrict__ tiles, size_t total16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte segment each
    if (i >= total16) return;
    const int lane = (int)(i & 63);
    const size_t tb = i >> 6;
    const int blk = (int)(tb % nblk);
    const size_t tile = tb / nblk;
    const int row = (int)(16 * tile + (lane & 15));
    uint4 v = {0, 0, 0, 0};
    if (row < rows) v = *reinterpret_cast<const uint4 *>(codes + (size_t)row * row_stride + 64 * blk + 16 * (lane >> 4));
    v.x = transpose_fields(v.x);
    v.y = transpose_fields(v.y);
    v.z = transpose_fields(v.z);
    v.w = transpose_fields(v.w);
    *reinterpret_cast<uint4 *>(tiles + i * 16) = v;
}

// scales [rows, cols/32] -> [tile][256-block][kg][row c][p]: value of (row 16 tile + c, 32-block 8 blk + 2 kg + p).
// A lane of kernels_gemvq.hip (kg, c) reads its two values as one dword (f16) / one float2; a lane of k_gemv_mfma
// (rows 4g..4g+3, k-group kg) reads 8 consecutive values.
__global__ void k_retile_scales(const float *__restrict__ scales, int rows, int nblk, float *__restrict__ out,
                                size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i & 1), c = (int)((i >> 1) & 15), kg = (int)((i >> 5) & 3);
    const size_t tb = i >> 7;
    const int blk = (int)(tb % nblk);
    const size_t tile = tb / nblk;
    const int row = (int)(16 * tile + c);
    out[i] = row < rows ? scales[(size_t)row * nblk * 8 + 8 * blk + 2 * kg + p] : 0.0f;
}

__global__ void k_retile_scales_h(const float *__restrict__ scales, int rows, int nblk, uint16_t *__restrict__ out, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i & 1), c = (int)((i >> 1) & 15), kg = (int)((i >> 5) & 3);
    const size_t tb = i >> 7;
    const int blk = (int)(tb % nblk);
    const size_t tile = tb / nblk;
    const int row = (int)(16 * tile + c);
    const _Float16 h = (_Float16)(row < rows ? scales[(size_t)row * nblk * 8 + 8 * blk + 2 * kg + p] : 0.0f);  // exact: checked at upload
    out[i] = __builtin_bit_cast(uint16_t, h);
}

// ---- the inverse permutations: tiles -> reference layout (transposing the 4 x 4 fields twice is the identity) ----
__global__ void k_untile(const uint8_t *__restrict__ tiles, size_t row_stride, int rows, int nblk, uint8_t *__restrict__ codes, size_t total16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total16) return;
    const int lane = (int)(i & 63);
    const size_t tb = i >> 6;
    const int blk = (int)(tb % nblk);
    const size_t tile = tb / nblk;
    const int row = (int)(16 * tile + (lane & 15));
    if (row >= rows) return;
    uint4 v = *reinterpret_cast<const uint4 *>(tiles + i * 16);
    v.x = transpose_fields(v.x);
    v.y = transpose_fields(v.y);
    v.z = transpose_fields(v.z);
    v.w = transpose_fields(v.w);
    const size_t off = (size_t)row * row_stride + 64 * blk + 16 * (lane >> 4);
    if (off + 16 <= (size_t)rows * row_stride) *reinterpret_cast<uint4 *>(codes + off) = v;  // row_stride = nblk * 64 on this path
}
template <class T>
__global__ void k_untile_scales(const T *__restrict__ tiles, int rows, int nblk, float *__restrict__ scales, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i & 1), c = (int)((i >> 1) & 15), kg = (int)((i >> 5) & 3);
    const size_t tb = i >> 7;
    const int blk = (int)(tb % nblk);
    const size_t tile = tb / nblk;
    const int row = (int)(16 * tile + c);
    if (row < rows) scales[(size_t)row * nblk * 8 + 8 * blk + 2 * kg + p] = (float)tiles[i];
}

size_t weights_device_bytes(const Weights &w) {
    const size_t n_tiles = div_ceil(w.rows, 16), nblk = div_ceil(w.cols, 256);
    size_t b = 0;
    if (w.codes) b += w.rows * w.row_stride_bytes;
    if (w.scales) b += w.rows * w.nblk * sizeof(float);
    if (w.tiles) b += n_tiles * nblk * 1024;
    if (w.tiles4) b += n_tiles * nblk * 2048;
    if (w.scale_tiles) b += n_tiles * nblk * 128 * sizeof(float);
    if (w.scale_tiles_h) b += n_tiles * nblk * 128 * sizeof(uint16_t);
    if (w.ln_g) b += w.rows * sizeof(float);
    return b;
}

void trim_reference(Weights &w) {
    std::lock_guard<std::mutex> lk(*w.mu);
    if (w.ref_pins->load(std::memory_order_acquire) > 0) return;  // a launch on the copies is being enqueued (ReferencePin): they stay for now
    if (!w.tiles || w.row_stride_bytes != div_ceil(w.cols, 256) * 64) return;  // only what ensure_reference can rebuild
    if (w.codes) {
        (void)hipFree(w.codes);
        w.codes = nullptr;
    }
    if (w.scales && w.scaled && w.block_size == 32 && (w.scale_tiles || w.scale_tiles_h)) {
        (void)hipFree(w.scales);
        w.scales = nullptr;
    }
}

hipError_t ensure_reference(Weights &w, hipStream_t stream, bool pin, bool scales_only) {
    std::lock_guard<std::mutex> lk(*w.mu);
    struct PinOnSuccess {  // the pin is taken under the lock, on every successful return
        Weights &w;
        bool pin;
        hipError_t *e;
        ~PinOnSuccess() {
            if (pin && *e == hipSuccess) w.ref_pins->fetch_add(1, std::memory_order_acq_rel);
        }
    };
    hipError_t result = hipSuccess;
    PinOnSuccess guard{w, pin, &result};
    const size_t n_tiles = div_ceil(w.rows, 16), nblk = div_ceil(w.cols, 256);
    bool built = false;
    if (!w.codes && !scales_only) {
        if (!w.tiles) return result = hipErrorInvalidValue;
        const size_t bytes = w.rows * w.row_stride_bytes + 16;
        hipError_t e = hipMalloc((void **)&w.codes, bytes);
        if (e != hipSuccess) return result = e;
        const size_t total16 = n_tiles * nblk * 64;
        hipLaunchKernelGGL(k_untile, dim3((unsigned)div_ceil(total16, 256)), dim3(256), 0, stream, w.tiles, w.row_stride_bytes, (int)w.rows, (int)nblk,
                           w.codes, total16);
        built = true;
    }
    if (w.scaled && !w.scales) {
        if (w.block_size != 32 || !(w.scale_tiles || w.scale_tiles_h)) return result = hipErrorInvalidValue;
        hipError_t e = hipMalloc((void **)&w.scales, w.rows * w.nblk * sizeof(float));
        if (e != hipSuccess) return result = e;
        const size_t total = n_tiles * nblk * 128;
        if (w.scale_tiles_h)
            hipLaunchKernelGGL(k_untile_scales<_Float16>, dim3((unsigned)div_ceil(total, 256)), dim3(256), 0, stream,
                               reinterpret_cast<const _Float16 *>(w.scale_tiles_h), (int)w.rows, (int)nblk, w.scales, total);
        else
            hipLaunchKernelGGL(k_untile_scales<float>, dim3((unsigned)div_ceil(total, 256)), dim3(256), 0, stream, w.scale_tiles, (int)w.rows, (int)nblk,
                               w.scales, total);
        built = true;
    }
    if (!built) return result = hipSuccess;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);  // other threads / streams may use the pointers the moment the lock is gone
    return result = e;
}

hipError_t build_tiles(Weights &w, hipStream_t stream) {
    if (w.scales && w.block_size == 32 && w.scales_f16 && !w.scale_tiles_h) {
        const size_t n_tiles = div_ceil(w.rows, 16), nblk = div_ceil(w.cols, 256);
        const size_t total = n_tiles * nblk * 128;
        hipError_t e = hipMalloc((void **)&w.scale_tiles_h, total * sizeof(uint16_t));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_retile_scales_h, dim3((unsigned)div_ceil(total, 256)), dim3(256), 0, stream, w.scales, (int)w.rows,
                           (int)nblk, w.scale_tiles_h, total);
    }
    if (w.scales && w.block_size == 32 && !w.scales_f16 && !w.scale_tiles) {
        const size_t n_tiles = div_ceil(w.rows, 16), nblk = div_ceil(w.cols, 256);
        const size_t total = n_tiles * nblk * 128;
        hipError_t e = hipMalloc((void **)&w.scale_tiles, total * sizeof(float));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_retile_scales, dim3((unsigned)div_ceil(total, 256)), dim3(256), 0, stream, w.scales,
                           (int)w.rows, (int)nblk, w.scale_tiles, total);
    }
    if (w.tiles) return hipSuccess;
    if (!w.codes) return hipErrorInvalidValue;
    const size_t n_tiles = div_ceil(w.rows, 16), nblk = div_ceil(w.cols, 256);
    const size_t total16 = n_tiles * nblk * 64;
    hipError_t e = hipMalloc((void **)&w.tiles, total16 * 16);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_retile, dim3((unsigned)div_ceil(total16, 256)), dim3(256), 0, stream, w.codes,
                       w.row_stride_bytes, (int)w.rows, (int)nblk, w.tiles, total16);
    w.n_row_tiles = n_tiles;
    w.n_kblocks = nblk;
    return hipGetLastError();
}

}  // namespace bitnet_hip
