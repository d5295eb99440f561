// common.hpp -- shared host-side plumbing for libbitnet_hip.so (gfx950 only).
//
// Error convention follows the reference's C bridge (K/ffi/cpp_bridge.cpp:19-27):
// thread-local last-error string, int return codes, nothing thrown across the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>

#include "bitnet_hip.h"

namespace bitnet_hip {

extern thread_local std::string g_last_error;

inline int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
inline int set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define BH_HIP_TRY(expr)                                                                       \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::bitnet_hip::set_error(BITNET_HIP_ERR_GPU, "HIP error %d (%s) at %s:%d: %s", \
                                           (int)_e, hipGetErrorName(_e), __FILE__, __LINE__,   \
                                           #expr);                                             \
    } while (0)

inline size_t div_ceil(size_t a, size_t b) { return (a + b - 1) / b; }

// Code -> value maps used by the reference (SURVEY.md 8a "code-map summary"),
// packed as four int8 in one u32 (byte c = value of code c).
constexpr uint32_t pack_lut(int v0, int v1, int v2, int v3) {
    return (uint32_t)(uint8_t)(int8_t)v0 | ((uint32_t)(uint8_t)(int8_t)v1 << 8) |
           ((uint32_t)(uint8_t)(int8_t)v2 << 16) | ((uint32_t)(uint8_t)(int8_t)v3 << 24);
}
constexpr uint32_t LUT_QK256 = pack_lut(-2, -1, 1, 2);    // Q/i2s_qk256.rs:139-146
constexpr uint32_t LUT_TERNARY = pack_lut(0, 1, 0, -1);   // K/cpu/quantized_matmul.rs:19-27

// Device-resident weight matrix behind a bitnet_hip_weights_t handle.
struct Weights {
    size_t rows = 0;              // output features (n)
    size_t cols = 0;              // input features (k)
    size_t row_stride_bytes = 0;  // code bytes per row in `codes`
    size_t block_size = 0;        // scale block along k (0 = no scales)
    size_t nblk = 0;              // scale blocks per row
    uint32_t lut = 0;             // pack_lut(...)
    bool scaled = false;          // the matrix has block scales (whichever copies of them are materialised)
    // Reference layout.  Dropped once the streaming layout below exists (the fast kernels never read it: keeping both doubled
    // the device memory of a model) and rebuilt on demand by ensure_reference() for the reference-order kernels, under `mu`.
    uint8_t *codes = nullptr;     // [rows, row_stride_bytes] or null
    float *scales = nullptr;      // [rows, nblk] f32 or null (always kept for 256-element blocks: 4 B per 256 weights)
    std::shared_ptr<std::mutex> mu = std::make_shared<std::mutex>();
    // calls that are about to launch a kernel on `codes` / `scales` (ReferencePin): trim_reference leaves them alone meanwhile
    std::shared_ptr<std::atomic<int>> ref_pins = std::make_shared<std::atomic<int>>(0);
    size_t algorithmic_bytes = 0; // code bytes + scale bytes
    int device = 0;
    // MFMA layout (built lazily by the MFMA path): 16-row x 256-col tiles,
    // 1 KiB each, lane-ordered.  See kernels_mfma.hip.
    uint8_t *tiles = nullptr;
    // resident fp4 image for the prompt matmul's fp6 x fp4 form (kernels_gemm.hip k_retile_fp4 / ensure_fp4_image): optional, 4 bits per weight
    uint8_t *tiles4 = nullptr;
    float *scale_tiles = nullptr;  // 32-block scales in tile order (kernels_mfma.hip k_retile_scales)
    // Every scale is exactly an f16 value (BitNet32-F16 files store them so): the streaming layout
    // keeps them as f16 (2 bytes per 32 weights instead of 4) -- same numbers, fewer bytes.
    bool scales_f16 = false;
    // ... and +-2 s stays finite in f16 (|s| <= 32752): the prefill's k_gemm_f16w multiplies the code value into the scale in f16
    bool scales_f16_x2_finite = false;
    uint16_t *scale_tiles_h = nullptr;
    size_t n_row_tiles = 0, n_kblocks = 0;
    bool paired = false;  // rows are interleaved (gate tile, up tile) pairs (weights_concat interleave16)
    // bitnet_hip_weights_bind_ln: g_r = W[r,:] . gamma for ONE LayerNorm weight vector (device pointer it was
    // computed from): lets the fused LayerNorm -> GEMV apply the normalisation after the product
    float *ln_g = nullptr;
    const float *ln_gamma_bound = nullptr;
};

// Optional work fused around a GEMV launch (MFMA kernel only).
struct GemvFusion {
    const float *ln_gamma = nullptr;  // LayerNorm(no bias, mean-subtracting) prologue on x
    float ln_eps = 0.0f;
    const float *residual = nullptr;  // y = residual + W x
    bool silu_mul = false;            // rows are (gate tile, up tile) pairs: y = silu(gate) * up
    bool x_f16 = false, y_f16 = false;  // prefill matmul only: x rows / the silu * up output are f16 (BITNET_HIP_FUSE_X_F16 / _Y_F16)
    bool int8_form = false;             // prefill matmul only: keep the int8 digit planes (BITNET_HIP_FUSE_INT8_DIGITS: no fp6 / f16 form)
    bool fp6_form = false;              // prefill matmul only: the fp6 x fp4 form of the 2-digit product (BITNET_HIP_FUSE_FP6_DIGITS)
    bool fp6_expand = false;            // ... expanding the 2-bit tiles in its K loop instead of reading the resident fp4 image (BITNET_HIP_FUSE_FP6_EXPAND)
    // x = the decode attention's output, merged from its chunk records by the GEMV itself (no combine launch):
    // records of launch_attn_decode(..., combine = false); contexts of at most 4 records
    const float *attn_rec = nullptr;
    const int *attn_pos = nullptr;    // *attn_pos + 1 keys
    int attn_chunks_max = 0;          // records per KV head in the buffer
    int attn_chunk_log2 = 6;          // positions per record = 1 << attn_chunk_log2
    int attn_group_log2 = 0;          // query heads per KV head = 1 << attn_group_log2
    // the output also goes out as a QAct (qact.hpp) for the next GEMV: qout records, the consumer's LayerNorm weight
    // (nullable) and per-16-row (sum, sum of squares) pairs (nullable); rows % 16 == 0
    void *qout = nullptr;
    const float *gamma_out = nullptr;
    double *stats_out = nullptr;
};

// Inputs / outputs of the GEMV on pre-quantised activations (kernels_gemvq.hip).
struct GemvQIo {
    const void *qin = nullptr;         // QAct records of the activation vector [cols]
    const double *stats_in = nullptr;  // with ln_gamma: (sum, sum of squares) per 16 columns of the un-normalised vector
    const float *ln_gamma = nullptr;   // LayerNorm of the consumer, applied after the product: must be the bound gamma
    float ln_eps = 0.0f;
    const float *residual = nullptr;
    bool silu_mul = false;
    float *y = nullptr;                // f32 output (nullable)
    void *qout = nullptr;              // QAct output (nullable) ...
    const float *gamma_out = nullptr;  // ... multiplied by the NEXT LayerNorm's weight first (nullable)
    double *stats_out = nullptr;       // ... with its row statistics (nullable)
    // instead of qin: the decode attention's chunk records, merged by the GEMV itself (short contexts, <= 4 records)
    const float *attn_rec = nullptr;
    const int *attn_pos = nullptr;
    int attn_chunks_max = 0, attn_group_log2 = 0, attn_chunk_log2 = 6;  // positions per record = 1 << attn_chunk_log2
};

// ---- kernel launchers (kernels_*.hip) -------------------------------------
// All launchers are asynchronous on `stream` and return hipGetLastError().

hipError_t launch_gemv_exact(const Weights &w, const float *x, float *y, size_t m, hipStream_t stream);
hipError_t launch_gemv_valu(const Weights &w, const float *x, float *y, size_t m, hipStream_t stream);
bool valu_supported(const Weights &w);
hipError_t launch_gemv_mfma(const Weights &w, const float *x, float *y, size_t m, const GemvFusion &fu,
                            hipStream_t stream);
bool mfma_supported(const Weights &w);
int mfma_pick_ksplit(size_t rows, size_t cols, bool paired, int nw);
bool gemvq_supported(const Weights &w);
hipError_t launch_gemv_q(const Weights &w, const GemvQIo &io, hipStream_t stream);
hipError_t launch_quant_act(const float *x, const float *gamma, size_t n, void *qout, double *stats, hipStream_t stream);
hipError_t launch_embed_q(const void *table, const int *tokens, const int *offset_ptr, int hidden, int vocab, float *x_out,
                          const float *gamma, void *qout, double *stats, hipStream_t stream);
hipError_t build_tiles(Weights &w, hipStream_t stream);
// codes / scales in the reference layout, rebuilt from the tiles if they were dropped (exact inverse permutation);
// synchronises `stream` when it had to rebuild.  trim_reference drops them again when the tiles can stand in.
// scales_only: the caller reads `scales` alone (the prefill matmul's row-major block scales): the trimmed row-major CODES are
// not re-materialised -- one copy of the codes stays on the device, and when `scales` exists already nothing is allocated,
// launched or synchronised (safe under stream capture).
hipError_t ensure_reference(Weights &w, hipStream_t stream, bool pin = false, bool scales_only = false);
void trim_reference(Weights &w);
// Holds the reference-layout copies from "rebuilt if missing" until the launch that reads them has been ENQUEUED: another thread's
// trim_reference (bitnet_hip_weights_trim, weights_concat) skips a pinned matrix; once the pin is gone a trim's hipFree waits for
// the device, so the enqueued kernel still sees its operands (the lock used to be dropped before the launch).
struct ReferencePin {
    Weights &w;
    hipError_t status;
    ReferencePin(Weights &w_, hipStream_t stream, bool scales_only = false) : w(w_), status(ensure_reference(w_, stream, true, scales_only)) {}
    ~ReferencePin() {
        if (status == hipSuccess) w.ref_pins->fetch_sub(1, std::memory_order_acq_rel);
    }
    ReferencePin(const ReferencePin &) = delete;
    ReferencePin &operator=(const ReferencePin &) = delete;
};
size_t weights_device_bytes(const Weights &w);
// many-row (prefill) matmul, kernels_gemm.hip: ndig in {2,3,4} fixed-point digits per activation
bool gemm_supported(const Weights &w);
bool gemm_fp6_supported(const Weights &w);
bool gemm_fp4_resident_enabled();                            // BITNET_HIP_FP4_RESIDENT != 0 (default on)
hipError_t ensure_fp4_image(Weights &w, hipStream_t stream);  // builds Weights::tiles4 once (under the handle's mutex)
void drop_fp4_image(Weights &w);
bool gemm_takes_fp6(const Weights &w, const GemvFusion &fu, int ndig);  // would launch_gemm_mfma run k_gemm_fp6 for this call?  // unscaled, code map in -2 .. 2: the fp6 x fp4 form of the 2-digit product (k_gemm_fp6)
bool gemm_needs_row_major_scales(const Weights &w);  // 256-block scales and non-f16 32-block scales: the others read the scale tiles
size_t gemm_workspace_bytes(size_t m, size_t cols, int ndig);
// The f16 activation chain of the prompt forward (kernels_gemm.hip k_gemm_f16a<.., 1>): the input is an f16 matrix [m_pad][cols]
// (m_pad = m rounded up to 64 rows) written by the kernel that PRODUCED the activations; LayerNorm of the input is applied after the
// product from per-slab (sum, sum of squares) partials of the exact f32 rows (needs bitnet_hip_weights_bind_ln); outputs: f32 rows
// (optionally residual + W x), and / or f16 rows (optionally x gamma_out: the next LayerNorm's weight) + the next partials.
struct GemmF16Io {
    const void *xh = nullptr;
    const float *stats_in = nullptr;  // float2 [n_stats][m_pad]
    int n_stats = 0;
    float ln_eps = 0.0f;
    float *y = nullptr;               // f32 [m][rows] (rows / 2 with silu_mul), nullable
    const float *residual = nullptr;  // f32 [m][rows], may alias y
    bool silu_mul = false;
    void *yh = nullptr;               // f16 [m_pad][rows] (rows / 2 with silu_mul), nullable
    const float *gamma_out = nullptr; // [rows]: yh = f16(gamma_out * y)
    float *stats_out = nullptr;       // float2 [rows / 64][m_pad]
    void *qb_out = nullptr;           // QB32 buffer (qb32_bytes(m, rows)) receiving gamma_out * y (launch_gemm_f16_chain only; not with silu_mul)
};
// QB32 activation rows (kernels_gemm.hip): digit records [m_pad][cols / 256][576] then one exponent byte per (token, 32 columns); m_pad = m up to 64
size_t qb32_bytes(size_t m, size_t cols);
hipError_t launch_rows_to_qb32(const float *x, const float *gamma, size_t m, size_t cols, void *qb, float *stats, hipStream_t stream);
bool gemm_qb32_supported(const Weights &w);
// the fp6 x fp4 form on QB32 rows (io.xh = the QB32 buffer) with the f16 chain's epilogue (LayerNorm after the product, residual, silu * up -> f16 rows)
hipError_t launch_gemm_qb32(const Weights &w, const GemmF16Io &io, size_t m, hipStream_t stream);
bool gemm_f16_chain_supported(const Weights &w);
unsigned long long f16_saturations(bool reset);  // f16 hand-over values clamped to +-65504 since the last reset (lanes, not elements); synchronises the device
hipError_t launch_gemm_f16_chain(const Weights &w, const GemmF16Io &io, size_t m, hipStream_t stream);
hipError_t launch_rows_to_f16(const float *x, const float *gamma, size_t m, size_t cols, void *xh, float *stats, hipStream_t stream);
hipError_t launch_gemm_mfma(const Weights &w, const float *x, float *y, size_t m, const GemvFusion &fu, int ndig,
                            void *workspace, size_t workspace_bytes, hipStream_t stream);
// the tile form launch_gemm_mfma chose on this thread's last call: digits, tokens per wave tile (16 x TTW), waves per workgroup,
// weight-scale mode (0 none, 1 per 256-block, 2 masked K = 64 per 32-block, 3 K = 32 MFMA with f16 scale tiles)
struct GemmTileChoice {
    int digits = 0, wave_tokens = 0, waves = 0, scale_mode = 0, wave_rows = 64;
    int resident_fp4 = 0;  // the fp6 form read its weight operands from the resident fp4 image
};
extern thread_local GemmTileChoice g_last_gemm_tile;
extern unsigned long long *g_mfma_stamps;  // diagnostic build only
hipError_t launch_matmul_i2s_u8(const int8_t *a, const uint8_t *b, float *c, size_t m, size_t n,
                                size_t k, hipStream_t stream);
hipError_t launch_quantize_i2s(const float *in, size_t n, uint8_t *out, size_t out_len, float *scales,
                               hipStream_t stream);
// kernels_provider.hip: the trait ops at speed + the pieces of QuantizedLinear::quantized_matmul_i2s
bool matmul_i2s_tiled_ok(size_t m, size_t n, size_t k);
hipError_t launch_matmul_i2s_tiled(const int8_t *a, const uint8_t *b, float *c, size_t m, size_t n, size_t k, hipStream_t stream);
hipError_t launch_quantize_i2s_fast(const float *in, size_t n, uint8_t *out, size_t out_len, float *scales, hipStream_t stream);
hipError_t launch_quant_input_i2s(const float *x, int8_t *q, size_t n, hipStream_t stream);
hipError_t launch_unpack_codes_u8(const uint8_t *packed, uint8_t *out, size_t numel, hipStream_t stream);
hipError_t launch_apply_scales(float *out, size_t m, size_t n, const float *scales, size_t n_scales, size_t in_features, size_t block_size,
                               hipStream_t stream);
hipError_t launch_dequant_i2s(const uint8_t *bytes, size_t rows, size_t cols, size_t block, int inv,
                              float k, int transposed, float *out, hipStream_t stream);

hipError_t launch_embed_f16(const void *table, const int *tokens, const int *offset_ptr, int n, int hidden,
                            int vocab, float *out, hipStream_t stream);
hipError_t launch_advance_pos(int *pos_ptr, hipStream_t stream);
hipError_t launch_add(const float *a, const float *b, float *out, size_t n, hipStream_t stream);
// out[i] = silu(gate[.]) * up[.]; tile == 0: gate[i], up[i]; tile > 0: both live in ONE buffer of alternating
// tile-row groups (gate tile, up tile, ...), gate = the buffer, up = gate + tile (weights_concat interleave16)
hipError_t launch_silu_mul(const float *gate, const float *up, float *out, size_t n, size_t tile, hipStream_t stream);
hipError_t launch_norm_rows(const float *x, const float *gamma, float *out, int rows, int hidden, float eps,
                            bool rms, hipStream_t stream);
hipError_t launch_attn_decode(const float *qkv, const float *rope_sin, const float *rope_cos, float *kcache,
                              float *vcache, int n_heads, int n_kv, int D, int max_pos, const int *pos_ptr,
                              float *scratch, float *out, hipStream_t stream, bool combine = true, int halves = 1, void *qout = nullptr,
                              int kv_f16 = 0);
// decode attention chunk record (one per KV head and 64- or 128-position chunk) in the scratch buffer:
// (m, l) per head of the query group [4][2], then the un-normalised P.V partial [4][128]
constexpr int kAttnRecFloats = 8 + 4 * 128;
size_t attn_scratch_floats(int n_kv, int max_pos);
size_t attn_prefill_workspace_bytes(int n_heads, int n_kv, int nq, int T);
hipError_t launch_attn_generic(const float *q, const float *k, const float *v, float *out, int n_heads, int seq, int causal,
                               float scale, void *workspace, size_t workspace_bytes, hipStream_t stream);
hipError_t launch_attn_prefill(const float *q, int ld_q, const int *q_block_pos, int nq, const float *kv, int ld_kv, int T,
                               const float *rope_sin, const float *rope_cos, float *kcache, float *vcache, int n_heads, int n_kv,
                               int D, int max_pos, void *workspace, size_t workspace_bytes, float *out, hipStream_t stream,
                               int zz_world = 0, int kv_f16 = 0, int cache_f16 = 0, int phase = 0);
hipError_t launch_pack_cols(const float *src, size_t ld, size_t col0, size_t ncols, size_t rows, void *dst, int f16, hipStream_t stream);
hipError_t launch_stream_read(const void *buf, size_t bytes, unsigned *sink, hipStream_t stream);
hipError_t launch_logits_f16(const void *table, const float *x, const float *gamma, float eps, int hidden, int vocab,
                             float *logits, float *best_val, int *best_idx, int n_wg, hipStream_t stream);
hipError_t launch_argmax_final(const float *best_val, const int *best_idx, int n, int *token_out, int *pos_ptr,
                               int *history, const int *n_forced, hipStream_t stream);
hipError_t launch_argmax(const float *v, int n, float *best_val, int *best_idx, int n_wg, int *token_out,
                         hipStream_t stream);

}  // namespace bitnet_hip
