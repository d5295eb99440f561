/*
 * bitnet_hip.h -- C ABI of libbitnet_hip.so: the MI355X (gfx950) drop-in for the
 * I2_S / QK256 hot path of EffortlessMetrics/BitNet-rs.
 *
 * This is the boundary the reference's `bitnet-kernels` ROCm provider would bind
 * (`extern "C"`, plain pointers and sizes).  Each entry point names the reference
 * interface it replaces (path:line under the reference checkout).  Prefixes:
 *   K/ = crates/bitnet-kernels/src/   Q/ = crates/bitnet-quantization/src/
 *   M/ = crates/bitnet-models/src/    T  = crates/bitnet-transformer/src/lib.rs
 *
 * Conventions (identical to the reference's existing C bridge, K/ffi/bridge.rs:17-39
 * and K/ffi/cpp_bridge.cpp:19-27,76-86,112-118):
 *   - every fallible call returns int: 0 = success, non-zero = error;
 *   - the error text is kept in a thread-local string, valid until the next call
 *     on the same thread, read with bitnet_hip_get_last_error();
 *   - the caller owns every buffer; nothing is retained past return (handles own
 *     their device copies); outputs are fully overwritten;
 *   - no exception crosses the boundary; entry points are re-entrant.
 *
 * Two families:
 *   1. host-pointer drop-ins: argument lists mirror the Rust slices 1:1
 *      (pointer + length), synchronous and self-contained (H2D, launch, D2H),
 *      like the reference GPU provider does per call (K/gpu/cuda.rs:287-336);
 *   2. device-resident API: upload weights once -> opaque handle; activations
 *      and outputs are device pointers; optional hipStream_t (void*).  This is the
 *      measured path.
 *
 * There is NO CPU fallback behind any symbol: without a HIP device every compute
 * entry point fails with BITNET_HIP_ERR_GPU.
 *
 * ONE SKU.  The launch heuristics -- token-tile and row-tile choices of the tiled matmuls, the
 * weight-stationary group size, the grid covers, the attention's key splits, the GEMV's workgroup
 * counts -- are constants tuned for MI355X: 256 CUs in 8 XCDs, 4 MiB of L2 per XCD, 160 KiB of
 * LDS per CU, 512 registers per SIMD lane.  Results do not depend on them (placement and tiling
 * change speed only); on another gfx950 part they would want re-tuning (kGemmCUs, the cover
 * thresholds, gemm_weight_group's 1.5 MiB in kernels_gemm.hip; Decoder::hybrid_applies' 400
 * workgroups in host/decoder.cpp).  Environment switches that steer them: INTEGRATION.md.
 */
#ifndef BITNET_HIP_H
#define BITNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Return codes.  Mapping to crates/bitnet-common/src/error.rs:80-97 KernelError:
 *   INVALID_ARGUMENT -> InvalidArguments, GPU -> GpuError,
 *   UNSUPPORTED -> UnsupportedHardware, EXECUTION -> ExecutionFailed. */
#define BITNET_HIP_OK 0
#define BITNET_HIP_ERR_INVALID_ARGUMENT (-1) /* same value cpp_bridge.cpp returns for null/size errors */
#define BITNET_HIP_ERR_GPU (-2)
#define BITNET_HIP_ERR_UNSUPPORTED (-3)
#define BITNET_HIP_ERR_EXECUTION (-4)

/* QuantizationType ints, K/ffi.rs:40-44 */
#define BITNET_HIP_QTYPE_I2S 0
#define BITNET_HIP_QTYPE_TL1 1
#define BITNET_HIP_QTYPE_TL2 2

/* ------------------------------------------------------------------------- */
/* lifecycle  (K/ffi/bridge.rs:18-20,38: bitnet_cpp_init/cleanup/is_available/ */
/*             get_last_error;  K/rocm/mod.rs:125-137: is_rocm_available,      */
/*             rocm_device_count)                                              */
/* ------------------------------------------------------------------------- */

/* Idempotent.  Selects `device` for the calling thread and creates the library's
 * internal workspace.  device < 0 means "current device". */
int bitnet_hip_init(int device);
void bitnet_hip_cleanup(void);
/* Non-zero iff at least one HIP device is visible.  The BITNET_ENABLE_ROCM=1
 * opt-in (K/rocm/mod.rs:76-81) stays on the host side of the boundary. */
int bitnet_hip_is_available(void);
int bitnet_hip_device_count(void);
/* NULL when no error is recorded on this thread (cpp_bridge.cpp:281-283). */
const char *bitnet_hip_get_last_error(void);

/* K/rocm/mod.rs:33-53 RocmDeviceInfo */
typedef struct bitnet_hip_device_info {
    int32_t device_id;
    char name[128];
    char gcn_arch[64];
    uint64_t total_memory;
    int32_t compute_unit_count;
    int32_t max_wavefront_size;
    uint64_t max_shared_memory_per_workgroup;
    int32_t supports_fp16;
    int32_t supports_bf16;
} bitnet_hip_device_info;
int bitnet_hip_get_device_info(int device, bitnet_hip_device_info *out);

/* ------------------------------------------------------------------------- */
/* 1. host-pointer drop-ins                                                    */
/* ------------------------------------------------------------------------- */

/* gemv_qk256(qs_data:&[u8], x:&[f32], y_out:&mut [f32], rows, cols, row_stride_bytes)
 * Q/i2s_qk256.rs:346-353 (live call sites T:684, T:924,
 * crates/bitnet-inference/src/layers/quantized_linear.rs:572).
 * y[r] = sum_j LUT[code(r,j)] * x[j], LUT {-2,-1,+1,+2}, LSB-first 2-bit codes,
 * a ragged tail (cols % 256 != 0) is summed like the reference's `take(cols)`.  Error text keeps the substrings the reference's tests
 * assert ("y_out length", "x length", "too short"; Q/i2s_qk256.rs:701-739). */
int bitnet_hip_gemv_qk256(const uint8_t *qs_data, size_t qs_len, const float *x, size_t x_len,
                          float *y_out, size_t y_len, size_t rows, size_t cols,
                          size_t row_stride_bytes);

/* i2s_matmul_f32(activations, weights_packed, scales, out, m, n, k, block_size)
 * K/cpu/quantized_matmul.rs:57-66 == i2s_matmul_forward(act,w,scales,out,&I2sMatmulConfig)
 * K/cuda/quantized_matmul.rs:309-326.
 * out[r,c] = sum_blk sum_{i in blk} act[r,i] * (t(code(c,i)) * scales[c*nb+blk]),
 * t: 0->0, 1->+1, 3->-1, 2->0; weights [n, ceil(k/4)] bytes; scales f32 [n, ceil(k/bs)]. */
int bitnet_hip_i2s_matmul_f32(const float *activations, size_t act_len,
                              const uint8_t *weights_packed, size_t w_len, const float *scales,
                              size_t scales_len, float *out, size_t out_len, size_t m, size_t n,
                              size_t k, size_t block_size);

/* qk256_gemv_hip(weights, scales, input, output, m, n, k, &Qk256GemvConfig)
 * K/rocm/qk256_gemv.rs:52-65 -- the stub this library fills.  Semantics are those
 * the CUDA twin documents (K/cuda/qk256_gemv.rs:1-16, K/cuda/quantized_matmul.rs:7-9):
 * ternary codes, one f32 scale per 256-element block: input [m,k], weights
 * [n, k/4], scales [n, k/256], output [m,n].  == i2s_matmul_f32 with block 256. */
int bitnet_hip_qk256_gemv(const uint8_t *weights, size_t w_len, const float *scales,
                          size_t scales_len, const float *input, size_t in_len, float *output,
                          size_t out_len, size_t m, size_t n, size_t k);

/* KernelProvider::matmul_i2s(a:&[i8], b:&[u8], c:&mut [f32], m, n, k)
 * K/lib.rs:44-52; arithmetic of K/cpu/fallback.rs:39-83 (C = A_i8 . B_u8, B is
 * UNPACKED u8 row-major [k,n], no scale); FFI twin bitnet_cpp_matmul_i2s
 * K/ffi/bridge.rs:21-28. */
int bitnet_hip_matmul_i2s(const int8_t *a, size_t a_len, const uint8_t *b, size_t b_len, float *c,
                          size_t c_len, size_t m, size_t n, size_t k);

/* QuantizedLinear::quantized_matmul_i2s(input, provider)  crates/bitnet-inference/src/layers/quantized_linear.rs:704-802 --
 * the composite the alternative layer runs above the trait: input [m, k] f32 -> i8 by clamp(x, -2, 1).round()
 * (quantize_input_i2s :1762-1773); weights_packed = the layer's 2-bit data, unpacked LSB-first to RAW codes 0..3 and handed
 * to matmul_i2s as its u8 [k, n] row-major operand exactly as the reference does (:769-776, :722-731); then every output
 * column times scales[col] (one scale per output feature) or scales[min(col * k / block_size, len - 1)] (:779-802,
 * input_scale = 1).  Lossy by construction in the reference too; restated quirk for quirk (oracle: bo_quantized_matmul_i2s). */
int bitnet_hip_quantized_matmul_i2s(const float *input, size_t in_len, const uint8_t *weights_packed, size_t w_len,
                                    const float *scales, size_t scales_len, size_t block_size, float *output,
                                    size_t out_len, size_t m, size_t n, size_t k);

/* KernelProvider::quantize(input, output, scales, qtype)  K/lib.rs:53-58;
 * I2S arithmetic of K/cpu/fallback.rs:102-159 (block 32, scale = absmax/1.5,
 * >0.5 -> 1, <-0.5 -> 3, else 0, OR-packed LSB-first into `output`, which the
 * caller zeroes); FFI twin bitnet_cpp_quantize K/ffi/bridge.rs:29-37.
 * Only qtype I2S is on the hot path; TL1/TL2 return BITNET_HIP_ERR_UNSUPPORTED. */
int bitnet_hip_quantize(const float *input, size_t input_len, uint8_t *output, size_t output_len,
                        float *scales, size_t scales_len, int qtype);

/* dequantize_to_f32[_transposed][_with_cfg](bytes, shape, cfg)
 * M/quant/i2s.rs:237, :448, :591, :774 -- blocks of ceil(bs/4) code bytes + 2 B LE
 * f16 scale, value = clamp(|f16|[^-1]*k, 1e-3, 1e3) * {-2,-1,+1,+2}[code]; block size
 * inferred from the byte count among {256,128,64,32}.  Unlike the reference's
 * lenient partial-data path, a byte count matching no block size is an error. */
int bitnet_hip_dequant_i2s(const uint8_t *bytes, size_t bytes_len, size_t rows, size_t cols,
                           int inv_scale, float k, int transposed, float *out, size_t out_len);

/* ------------------------------------------------------------------------- */
/* 2. device-resident API                                                      */
/* ------------------------------------------------------------------------- */

typedef uint64_t bitnet_hip_weights_t; /* opaque; 0 is never a valid handle */

/* Kernel selection for the device GEMV/matmul (bitnet_hip_set_kernel). */
#define BITNET_HIP_KERNEL_AUTO 0
#define BITNET_HIP_KERNEL_EXACT 1 /* one thread per output, reference summation order: bit-exact */
#define BITNET_HIP_KERNEL_VALU 2  /* wave-per-row, f32 FMA, shuffle reduction */
#define BITNET_HIP_KERNEL_MFMA 3  /* i8 MFMA on exact fixed-point activation digits, reference row-major codes */
#define BITNET_HIP_KERNEL_MFMA_TILED 4 /* same, codes re-tiled at upload into 1-KiB lane-ordered tiles */
int bitnet_hip_set_kernel(int kernel);
int bitnet_hip_get_kernel(void);

/* I2SQk256NoScale {rows, cols, row_stride_bytes, qs}  Q/i2s_qk256.rs:66-106:
 * accepts qs_len within +-128 B of rows*row_stride (the reference's alignment
 * slack); map {-2,-1,+1,+2}, no scale. */
int bitnet_hip_weights_upload_qk256(const uint8_t *qs_data, size_t qs_len, size_t rows,
                                    size_t cols, size_t row_stride_bytes,
                                    bitnet_hip_weights_t *out);
/* Ternary weights of K/cpu/quantized_matmul.rs:47-56: [n, ceil(k/4)] code bytes +
 * f32 scales [n, ceil(k/block_size)], block_size 32 (BitNet32-F16) or 256. */
int bitnet_hip_weights_upload_i2s(const uint8_t *weights_packed, size_t w_len, const float *scales,
                                  size_t scales_len, size_t n, size_t k, size_t block_size,
                                  bitnet_hip_weights_t *out);
/* Same storage with an explicit 4-entry code map (value of code 0..3), e.g. {-2,-1,0,+1}
 * x f32 block scale = I2SQuantizer::dequantize_tensor, the form the GGUF loader gives the
 * 32-element flavours (Q/utils.rs:76-91, Q/i2s.rs:181-237, M/gguf_simple.rs:1260-1285). */
int bitnet_hip_weights_upload_coded(const uint8_t *weights_packed, size_t w_len, const float *scales,
                                    size_t scales_len, size_t n, size_t k, size_t block_size,
                                    const int8_t *code_map, bitnet_hip_weights_t *out);
/* BitNet32-F16 on-disk blocks: per 32 elements 8 code bytes + 2 bytes LE f16 scale, blocks
 * running row-major over [n, k] (k % 32 == 0).  scale_mode 0: scale = f16 as stored
 * (M/gguf_simple.rs:1217-1233); scale_mode 1: scale = clamp(|f16|, 1e-3, 1e3)
 * (M/quant/i2s.rs:66-100, Sym map, k = 1). */
int bitnet_hip_weights_upload_inline_f16(const uint8_t *blocks, size_t len, size_t n, size_t k,
                                         const int8_t *code_map, int scale_mode, bitnet_hip_weights_t *out);
int bitnet_hip_weights_free(bitnet_hip_weights_t w);
/* rows (n), cols (k), algorithmic bytes one GEMV reads from this handle
 * (code bytes + scale bytes). */
int bitnet_hip_weights_info(bitnet_hip_weights_t w, size_t *rows, size_t *cols,
                            size_t *algorithmic_bytes);

/* Device memory behind a handle.  The streaming layout (1-KiB code tiles + scale tiles) is the only copy kept: the
 * reference-layout copy the reference-order kernels (BITNET_HIP_KERNEL_EXACT / _VALU) and the tiled matmul's 32-element
 * scales read is rebuilt on their first use (exact inverse permutation; that call synchronises its stream once) and stays
 * until bitnet_hip_weights_trim drops it again.  Nothing a host does concurrently on OTHER handles is affected. */
size_t bitnet_hip_weights_device_bytes(bitnet_hip_weights_t w);
int bitnet_hip_weights_trim(bitnet_hip_weights_t w);

/* y_dev[rows] = W . x_dev[cols]   (one activation row: batch-1 decode) */
int bitnet_hip_gemv_dev(bitnet_hip_weights_t w, const float *x_dev, float *y_dev, void *stream);
/* Y_dev[m, rows] = X_dev[m, cols] . W^T   (forward_qk256's per-row loop T:683-691,
 * batched; row-major, leading dimensions cols / rows) */
int bitnet_hip_matmul_dev(bitnet_hip_weights_t w, const float *x_dev, float *y_dev, size_t m,
                          void *stream);
/* The same with the kernel named per call (BITNET_HIP_KERNEL_*) instead of the process-wide default of
 * bitnet_hip_set_kernel: what a multi-threaded host uses (the trait is Send + Sync, K/lib.rs:39), and how
 * bench.py / the tests run an UNFUSED step on the bit-exact reference-order kernel next to the fast one. */
int bitnet_hip_matmul_kernel_dev(bitnet_hip_weights_t w, const float *x_dev, float *y_dev, size_t m,
                                 int kernel, void *stream);
/* Many activation rows at once (prefill; forward_qk256's per-row loop T:683-691 as ONE tiled
 * matmul on the matrix cores).  Same fusions as gemv_fused_dev, per row.  `digits` = base-256
 * fixed-point digits per activation (4: the GEMV's 30 bits; 3: 22 bits; 2: 14 bits -- on BitNet32-F16 matrices (32-element blocks
 * with f16 scales) digits = 2 means f16 activations with one power-of-two scale per row, on the f16 matrix cores).
 * The int8 digit planes live in a caller-owned device workspace.
 * flags (beside BITNET_HIP_FUSE_SILU_MUL), int8 digit form only -- f16 hand-over between the prompt forward's launches:
 *   BITNET_HIP_FUSE_X_F16: x_dev holds f16 rows [m][cols] (the attention's f16 output, the f16 silu * up rows); no LayerNorm with it
 *   BITNET_HIP_FUSE_Y_F16: with FUSE_SILU_MUL, y_dev receives the product as f16 rows [m][rows / 2]
 *   BITNET_HIP_FUSE_INT8_DIGITS: keep the int8 base-256 digit planes at digits = 2 on every matrix (without it BitNet32-F16 matrices
 *     take f16 activations on the f16 matrix cores, above)
 *   BITNET_HIP_FUSE_FP6_DIGITS: digits = 2 on an unscaled matrix (QK256) whose code map lies in -2..2: the SAME 15-bit integer per
 *     activation as three base-32 digits on the block-scaled fp6 x fp4 MFMA (k_gemm_fp6: same products, f32 accumulation of exact
 *     integers -- bit-identical to the int8 form while every partial sum stays below 2^24); also the process default with
 *     BITNET_HIP_GEMM_FP6=1.  Measured: 10-14 % faster per gate|up / down launch under sustained load, its quantiser 9-20 us slower
 *     per launch: about even inside a prompt, hence opt-in (EXPERIMENTS 4.6).  Round 5: the weight operands come from the resident fp4 image
 *     (bitnet_hip_weights_fp4_image, built on first use) unless
 *   BITNET_HIP_FUSE_FP6_EXPAND is set too: the fp6 form expanding the 2-bit streaming tiles in its K loop (no image is built or read) */
#define BITNET_HIP_FUSE_X_F16 2
#define BITNET_HIP_FUSE_Y_F16 4
#define BITNET_HIP_FUSE_INT8_DIGITS 8
#define BITNET_HIP_FUSE_FP6_DIGITS 16
#define BITNET_HIP_FUSE_FP6_EXPAND 32
size_t bitnet_hip_matmul_workspace_bytes(size_t m, size_t k, int digits);
int bitnet_hip_matmul_fused_dev(bitnet_hip_weights_t w, const float *x_dev, float *y_dev, size_t m,
                                const float *ln_gamma_dev, float ln_eps, const float *residual_dev,
                                int flags, int digits, void *workspace_dev, size_t workspace_bytes,
                                void *stream);
/* Which tile form the calling thread's last bitnet_hip_matmul_[fused_]dev launch ran (any pointer may be null): digits,
 * tokens per wave tile (16 / 32 / 64), waves per workgroup (4 / 8), weight-scale mode (0 none, 1 per 256-block, 2 per
 * 32-block on the masked K = 64 MFMA, 3 per 32-block on the K = 32 int8 MFMA with f16 scale tiles, 4 the f16 MFMA with the block
 * scale folded into f16 weights and f16 activations: BitNet32-F16 at digits = 2).  The parity tests assert
 * that the instance bench.py times (2 digits, 64-token tile) is the one they compared with the oracle. */
int bitnet_hip_matmul_last_tile(int *digits, int *wave_tokens, int *waves, int *scale_mode);
/* Output rows per wave of that launch: 64 (four 16-row tiles, 256-row workgroups) or 80 (five: the 320-row workgroups the forms with
 * f32 accumulators take when they need fewer rounds of the chip -- 2560 output rows x 4096 tokens = 512 workgroups, one round);
 * or 128 (the fp6 x fp4 form's 2 x 2 wave arrangement, k_gemm_fp6w: a wave owns 128 rows x 32 tokens of the 256 x 64 workgroup tile);
 * 0 before the thread's first tiled matmul.  scale_mode 5 = the f16 MFMA on an unscaled matrix, 6 = the fp6 x fp4 form (k_gemm_fp6). */
int bitnet_hip_matmul_last_wave_rows(void);
/* 1 when that launch was the fp6 x fp4 form reading its weight operands from the resident fp4 image (below), else 0. */
int bitnet_hip_matmul_last_resident_fp4(void);
/* The resident fp4 image of an unscaled matrix whose code map lies in -2..2 (QK256: every value is exact in fp4 e2m1): 4 bits per
 * weight in the fp6 x fp4 MFMA's A-operand order, kept BESIDE the 2-bit streaming tiles (which remain the decode path's copy:
 * bitnet_hip_weights_device_bytes counts both; QK256 2B-4T: q|k|v + gate|up of 30 layers = 678 MB, all seven projections 1.04 GB --
 * HBM is 288 GB).  With it bitnet_hip_matmul_fused_dev's BITNET_HIP_FUSE_FP6_DIGITS form loads its weight operands directly: no code
 * expansion in the K loop; the integers multiplied are the same, so the results stay bit-identical to the int8 digit planes.
 * enable = 1 builds it (idempotent; an allocation + one kernel, synchronises `stream`), 0 frees it.  Without this call the first
 * fp6-form launch on the handle builds it (not under stream capture); BITNET_HIP_FP4_RESIDENT=0 disables the image altogether.
 * Replaces nothing in the reference: its loader keeps one packed copy (crates/bitnet-quantization/src/i2s_qk256.rs:66-128). */
int bitnet_hip_weights_fp4_image(bitnet_hip_weights_t h, int enable, void *stream);

/* The prompt forward's f16 ACTIVATION CHAIN (north_star: "2-bit weight unpack x f16 activation dot product"; replaces the reference's
 * per-row loop T:683-691 / T:924 over many activation rows, K/cpu/quantized_matmul.rs:57-96 for the scaled format): every projection
 * reads an f16 matrix xh [m_pad][cols] (m_pad = m rounded up to 64; rows >= m are never stored but must be readable) that the kernel
 * which PRODUCED those activations wrote, so no conversion / quantisation launch sits between two projections.
 *   ln_gamma != NULL: LayerNorm (no bias, mean-subtracting, T:67-100) of the INPUT, applied after the product: xh must hold
 *                     f16(gamma * x); stats_in = float2 (sum, sum of squares) partials [n_stats][m_pad] of the exact f32 x over its
 *                     columns; the matrix must be bound to this gamma (bitnet_hip_weights_bind_ln).
 *   y (nullable): f32 rows [m][rows]; residual (nullable, may alias y): y = residual + W x (T:1073, T:1125).
 *   BITNET_HIP_FUSE_SILU_MUL: (gate, up) interleaved handle, outputs have rows / 2 columns (T:756-781).
 *   yh (nullable): the output as f16 rows [m_pad][rows or rows / 2], multiplied by gamma_out[row] first when given (the NEXT
 *                  LayerNorm's weight); stats_out (nullable): float2 [rows / 64][m_pad] partials of the f32 outputs for that LayerNorm:
 *                  (sum, sum of squares) over disjoint row ranges that together cover the row -- one per wave of the producing launch
 *                  (64 rows, or 80 in 320-row workgroups: then the first rows / 80 entries are used and the others are written as zero);
 *                  the consumer adds all rows / 64 entries up (n_stats = rows / 64).
 * bitnet_hip_matmul_f16_supported: rows % 256 == 0, cols % 256 == 0, code map values in -2..2, no scales or f16 32-block scales.
 * bitnet_hip_rows_to_f16_dev: the chain's entry (the embedding rows): xh = f16(gamma * x) (gamma nullable) + stats partial 0. */
/* f16 hand-over rows carry no row scale: a value beyond +-65504 is clamped -- and COUNTED.  Returns the number of lanes (up to four elements each)
 * that clamped a value of a live token since the last reset, in any of the chain's writers (bitnet_hip_rows_to_f16_dev, the yh outputs of
 * bitnet_hip_matmul_f16_dev / _matmul_qb32_dev / FUSE_Y_F16); reset != 0 clears it.  Synchronises the device.  A host that cannot rule such
 * activations out (outlier channels x gamma beyond the f16 range) checks it behind a prompt and repeats the prompt on the row-scaled forms
 * (bitnet_hip_matmul_fused_dev), as Decoder::prefill does. */
unsigned long long bitnet_hip_f16_saturations(int reset);
int bitnet_hip_matmul_f16_supported(bitnet_hip_weights_t w);
int bitnet_hip_rows_to_f16_dev(const float *x_dev, const float *gamma_dev, size_t m, size_t cols, void *xh_dev, float *stats_dev, void *stream);
int bitnet_hip_matmul_f16_dev(bitnet_hip_weights_t w, const void *xh_dev, size_t m, const float *stats_in_dev, size_t n_stats,
                              const float *ln_gamma_dev, float ln_eps, float *y_dev, const float *residual_dev, int flags, void *yh_dev,
                              const float *gamma_out_dev, float *stats_out_dev, void *stream);

/* QB32 ACTIVATIONS (round 5): the producer-quantised input of the fp6 x fp4 prompt matmul -- what QAct is to the decode GEMV.  The digit forms'
 * row quantiser needs the row maximum, which no producing workgroup has, so it stayed a launch of its own (two per layer, 6 % of the QK256
 * prompt).  The block-scaled MFMA takes one E8M0 scale per 32 K-slots of a token, so a QB32 row scales LOCALLY: per 32-column unit, E = exponent
 * of the unit's largest |v|, q = rint(v 2^(13 - E)) as three balanced base-32 fp6 digits (the 15-bit integer class of the 2-digit planes), one
 * exponent byte per unit.  Buffer: records [m_pad][cols / 256] of 592 bytes (576 of digits, the block's 8 exponent bytes, 8 of padding), m_pad = m
 * rounded up to 64 (bitnet_hip_qb32_bytes).  LayerNorm is applied after the product from the producer's statistics partials, as on the f16 chain.
 *   bitnet_hip_rows_to_qb32_dev: f32 rows -> QB32 of gamma * x (gamma nullable) + stats partial 0: the chain's entry (the embedding rows).
 *   bitnet_hip_matmul_f16_dev(..., flags | BITNET_HIP_FUSE_YH_QB32, yh_dev = a QB32 buffer of [m][rows], gamma_out, stats_out): the o- / down-
 *     projection's epilogue leaves gamma_out * y as QB32 rows (64-token tiles: m_pad / 64 * rows / 256 >= 256 workgroups; no FUSE_SILU_MUL).
 *   bitnet_hip_matmul_qb32_dev: bitnet_hip_matmul_f16_dev's arguments with a QB32 buffer as the input: y / residual / FUSE_SILU_MUL / yh (f16
 *     rows) / gamma_out / stats_out as there; matrices: unscaled, code map in -2..2, rows % 256 == 0, cols % 256 == 0 (reads the resident fp4
 *     image, building it on first use).  Replaces the reference's per-row loop T:683-691 / T:924 over many activation rows. */
#define BITNET_HIP_FUSE_YH_QB32 64
size_t bitnet_hip_qb32_bytes(size_t m, size_t cols);
int bitnet_hip_rows_to_qb32_dev(const float *x_dev, const float *gamma_dev, size_t m, size_t cols, void *qb_dev, float *stats_dev, void *stream);
int bitnet_hip_matmul_qb32_supported(bitnet_hip_weights_t w);
int bitnet_hip_matmul_qb32_dev(bitnet_hip_weights_t w, const void *qb_dev, size_t m, const float *stats_in_dev, size_t n_stats,
                               const float *ln_gamma_dev, float ln_eps, float *y_dev, const float *residual_dev, int flags, void *yh_dev,
                               const float *gamma_out_dev, float *stats_out_dev, void *stream);

/* Several uploaded matrices with the same cols / code map / block size as ONE
 * launch: rows concatenated (q|k|v share their input: T:288-290).  interleave16 != 0
 * (exactly two matrices of equal rows % 16 == 0) alternates 16-row tiles a0,b0,a1,b1..
 * so a (gate, up) pair meets in one workgroup (BITNET_HIP_FUSE_SILU_MUL). */
int bitnet_hip_weights_concat(const bitnet_hip_weights_t *parts, size_t n_parts, int interleave16,
                              bitnet_hip_weights_t *out);

/* GEMV with the neighbouring decode-step work fused in (MFMA kernel; a matrix shape that kernel does not take -- e.g. 32-element
 * scales with cols % 256 != 0 -- gets the same result from separate device launches in the reference's op order):
 *   ln_gamma != NULL : x <- LayerNorm(x) first -- no bias, WITH mean subtraction,
 *                      (x-mean)/sqrt(mean((x-mean)^2)+eps)*gamma  (T:67-100, T:1015, T:1104)
 *   residual != NULL : y = residual + W x                        (T:1073, T:1125)
 *   BITNET_HIP_FUSE_SILU_MUL (handle from concat(..., interleave16=1)):
 *                      y[r] = silu(gate[r]) * up[r], y has rows/2   (T:756-781) */
#define BITNET_HIP_FUSE_SILU_MUL 1
int bitnet_hip_gemv_fused_dev(bitnet_hip_weights_t w, const float *x_dev, float *y_dev, size_t m,
                              const float *ln_gamma_dev, float ln_eps, const float *residual_dev,
                              int flags, void *stream);
/* Optional, once per (matrix, LayerNorm weight) pair: precomputes g_r = W[r,:] . gamma so that
 * gemv_fused_dev called with THIS ln_gamma_dev pointer applies the LayerNorm after the product,
 * (W (gamma*x) - mean g) / denom -- same value to f32 rounding, ~1.4 us less per launch (no
 * whole-row pass before the weights can be used).  Re-bind if the gamma buffer's contents change. */
int bitnet_hip_weights_bind_ln(bitnet_hip_weights_t w, const float *ln_gamma_dev, void *stream);

/* ------------------------------------------------------------------------- */
/* 3. decode-step operators (device pointers; K/rocm/rmsnorm.rs, attention.rs) */
/* ------------------------------------------------------------------------- */

/* rmsnorm_hip(input, gamma, output, num_rows, &HipRmsNormConfig{hidden_dim, eps})
 * K/rocm/rmsnorm.rs:50-60, host pointers: out = x / sqrt(mean(x^2)+eps) * gamma. */
int bitnet_hip_rmsnorm(const float *input, size_t in_len, const float *gamma, size_t gamma_len,
                       float *output, size_t out_len, size_t num_rows, size_t hidden_dim, float eps);
/* fused_attention_hip(q, k, v, output, seq_len, &HipAttentionConfig{num_heads, head_dim, causal, scale})
 * K/rocm/attention.rs:54-65: all four tensors [batch, num_heads, seq_len, head_dim] row-major f32
 * (batch = len / (num_heads*seq_len*head_dim)); softmax(scale * q k^T [+ causal mask]) v per head.
 * head_dim 128 (f16 operands on the matrix cores, f32 accumulate and softmax). */
int bitnet_hip_attention(const float *q, size_t q_len, const float *k, size_t k_len, const float *v,
                         size_t v_len, float *output, size_t out_len, size_t seq_len, size_t num_heads,
                         size_t head_dim, int causal, float scale);
/* qk256_gemv_hip_batch(&[GemvBatchItem], &cfg)  K/rocm/qk256_gemv.rs:67-82: items processed in order,
 * each with the arguments of bitnet_hip_qk256_gemv; stops at the first error. */
typedef struct bitnet_hip_gemv_item {
    const uint8_t *weights; size_t weights_len;
    const float *scales;    size_t scales_len;
    const float *input;     size_t input_len;
    float *output;          size_t output_len;
    size_t m, n, k;
} bitnet_hip_gemv_item;
int bitnet_hip_qk256_gemv_batch(const bitnet_hip_gemv_item *items, size_t n_items);
/* Device rows: rms != 0 -> RMSNorm (above); rms == 0 -> the LayerNorm the transformer
 * actually uses (layer_norm_with_optional_bias T:67-100: no bias, mean subtracted). */
int bitnet_hip_norm_rows_dev(const float *x_dev, const float *gamma_dev, float *out_dev, size_t rows,
                             size_t hidden, float eps, int rms, void *stream);
/* TransformerModel::embed (T:1390-1426), row gather from the f16 table [vocab, hidden];
 * tokens_dev: int32[]; reads tokens_dev[*offset_dev + i], i < n (offset_dev NULL = 0), so a
 * captured graph can walk a token history by position. */
int bitnet_hip_embed_f16_dev(const void *table_f16_dev, const int32_t *tokens_dev,
                             const int32_t *offset_dev, size_t n, size_t hidden, size_t vocab,
                             float *out_dev, void *stream);
/* The two elementwise steps of the reference's UNFUSED block (the fast path fuses them into GEMV epilogues):
 * out = a + b (residual add, T:1073, T:1125); out[i] = silu(gate[.]) * up[.] (T:765-781) -- tile == 0: separate
 * vectors; tile > 0: gate_dev is ONE vector of alternating `tile`-row groups (gate, up, gate, ...) as a plain GEMV
 * on a weights_concat(..., interleave16 = 1) handle produces it (tile = 16), up_dev = gate_dev + tile. */
int bitnet_hip_add_dev(const float *a_dev, const float *b_dev, float *out_dev, size_t n, void *stream);
int bitnet_hip_silu_mul_dev(const float *gate_dev, const float *up_dev, float *out_dev, size_t n, size_t tile,
                            void *stream);
/* *pos_dev += 1 (prompt positions whose logits nobody reads). */
int bitnet_hip_advance_pos_dev(int32_t *pos_dev, void *stream);
/* One new token through MultiHeadAttention::forward's core (T:373-540): RoPE on q,k with
 * the split-half layout (T:134-163) at position *pos_dev, append k,v to the f32 cache
 * (T:1171-1202; n_kv*ceil(max_pos/64)*64*head_dim floats each, layout private to this library: K is kept
 * transposed), GQA softmax attention over pos+1 keys.  head_dim 128, n_heads/n_kv <= 4.
 * The caches must be ZERO-FILLED before their first use (hipMemset once after allocation): the kernel
 * reads whole 64-position tiles and gives slots past the context an exact zero weight, which only cancels
 * finite bit patterns (slots left over from an earlier, longer sequence are fine).
 * qkv_dev: [n_heads*D | n_kv*D | n_kv*D] raw projections; out_dev: [n_heads*D].
 * rope_sin/cos_dev: [max_pos, D/2] (crates/bitnet-rope/src/lib.rs:59-93). */
int bitnet_hip_attention_decode_dev(const float *qkv_dev, const float *rope_sin_dev,
                                    const float *rope_cos_dev, float *kcache_dev, float *vcache_dev,
                                    size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos,
                                    const int32_t *pos_dev, float *scratch_dev, float *out_dev,
                                    void *stream);
/* The same operator with workgroups of 512 threads that cover 128 positions each (half as many workgroups and chunk
 * records).  Slower than bitnet_hip_attention_decode_dev while n_kv_heads * ceil((pos + 1) / 64) fits the 256 CUs
 * (+1.5 us per layer at 0.4k..2.5k keys), faster beyond (4k keys, 5 KV heads: 320 -> 160 workgroups, 11.9 -> 10.9 us). */
int bitnet_hip_attention_decode_wide_dev(const float *qkv_dev, const float *rope_sin_dev,
                                         const float *rope_cos_dev, float *kcache_dev, float *vcache_dev,
                                         size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos,
                                         const int32_t *pos_dev, float *scratch_dev, float *out_dev,
                                         void *stream);
/* Short contexts (at most bitnet_hip_attention_merge_max_keys() keys = 4 records of 64 positions): the attention in ONE launch plus an output projection that
 * merges the chunk results itself.  bitnet_hip_attention_decode_partial_dev = the first of the two kernels of
 * bitnet_hip_attention_decode_dev (RoPE, KV append, per-chunk softmax pieces into scratch_dev; same arguments,
 * no out_dev); bitnet_hip_gemv_attn_merge_dev(w, scratch, ...) = bitnet_hip_gemv_fused_dev(w, attention output,
 * y, m = 1, no LayerNorm, residual) with the attention output assembled from those records on the fly (every
 * workgroup of the projection reads every live record, n_chunks x 10 KB -- which is why this is for short
 * contexts only; the caller falls back to bitnet_hip_attention_decode_dev + bitnet_hip_gemv_fused_dev once
 * *pos_dev + 1 exceeds that).  w: cols == n_heads * 128; query group 1, 2 or 4.  scratch_dev must be zero-filled
 * before its first use (records of chunks past the context are read and given zero weight). */
int bitnet_hip_attention_decode_partial_dev(const float *qkv_dev, const float *rope_sin_dev, const float *rope_cos_dev,
                                            float *kcache_dev, float *vcache_dev, size_t n_heads, size_t n_kv_heads,
                                            size_t head_dim, size_t max_pos, const int32_t *pos_dev, float *scratch_dev,
                                            void *stream);
size_t bitnet_hip_attention_merge_max_keys(void); /* largest *pos_dev + 1 the merging projection takes (4 records) */
int bitnet_hip_gemv_attn_merge_dev(bitnet_hip_weights_t w, const float *attn_scratch_dev, size_t n_heads,
                                   size_t n_kv_heads, size_t max_pos, const int32_t *pos_dev, float *y_dev,
                                   const float *residual_dev, void *stream);
/* ---- the decode step on activations quantised by their PRODUCER ("QAct") ---------------------------------------
 * north_star's kernel is "2-bit weight unpack x f16 activation dot product".  The f16-class activation of this
 * library is a QAct: per 16 consecutive elements one power-of-two scale and a 15-bit fixed-point value per element
 * (two int8 digit planes), i.e. every element to 2^-15 of its 16-group's maximum; 576 bytes per 256 elements
 * (bitnet_hip_qact_bytes; layout private to the library, csrc/qact.hpp).  The kernel that PRODUCES a vector
 * writes it in this form (16 output rows of a GEMV = one group), so the consuming GEMV feeds the digit planes to the
 * matrix cores without touching them -- in round 1 every wave of every GEMV re-quantised its K range.  Exact f32
 * activations stay available through gemv_dev / gemv_fused_dev.
 * LayerNorm between producer and consumer (T:1015, T:1104) is applied after the product as with weights_bind_ln:
 * the producer multiplies by the consumer's gamma (gamma_out_dev) before quantising and leaves one (sum, sum of
 * squares) f64 pair per 16 rows (stats_out, bitnet_hip_qact_stats_bytes); the consumer passes them as stats_in together
 * with the bound ln_gamma_dev.  Shapes: cols % 256 == 0, rows % 16 == 0, no scales or 32-element block scales
 * (bitnet_hip_gemv_q_supported); everything else stays on gemv_fused_dev. */
size_t bitnet_hip_qact_bytes(size_t cols);
size_t bitnet_hip_qact_stats_bytes(size_t cols);
/* any f32 device vector -> QAct [* gamma] [+ statistics] (what a producing kernel does in its epilogue) */
int bitnet_hip_quantize_act_dev(const float *x_dev, const float *gamma_dev, size_t cols, void *qact_out,
                                double *stats_out, void *stream);
/* bitnet_hip_embed_f16_dev for ONE token, also leaving the first block's QAct (gamma_dev = its attention_norm) */
int bitnet_hip_embed_q_dev(const void *table_f16_dev, const int32_t *tokens_dev, const int32_t *offset_dev,
                           size_t hidden, size_t vocab, float *x_out_dev, const float *gamma_dev, void *qact_out,
                           double *stats_out, void *stream);
int bitnet_hip_gemv_q_supported(bitnet_hip_weights_t w);
/* bitnet_hip_gemv_fused_dev on a QAct input: y = [residual +] W . act  [LayerNorm: ln_gamma_dev bound + stats_in]
 * [FUSE_SILU_MUL]; outputs: y_dev f32 (nullable) and / or qact_out [* gamma_out_dev] [+ stats_out]. */
int bitnet_hip_gemv_q_dev(bitnet_hip_weights_t w, const void *qact_in, const double *stats_in,
                          const float *ln_gamma_dev, float ln_eps, const float *residual_dev, int flags,
                          float *y_dev, void *qact_out, const float *gamma_out_dev, double *stats_out, void *stream);
/* bitnet_hip_attention_decode[_wide]_dev whose combine step also (or only: out_dev NULL) leaves the QAct of the
 * attention output for the o-projection */
#define BITNET_HIP_ATTN_WIDE 1    /* 128-position workgroups (bitnet_hip_attention_decode_wide_dev) */
#define BITNET_HIP_ATTN_KV_F16 2  /* the caches hold f16: HALF the bytes of the long-context stream.  Opt-in: the reference's cache is f32
                                   * (T:1171-1202); values are rounded once, from the exact f32 k / v, when appended.  Same element
                                   * counts as the f32 caches, half the allocation; zero-fill before first use likewise.  A cache is
                                   * f16 or f32 for its whole life: fill it with bitnet_hip_attention_prefill_kv16_dev / cache_f16 = 1. */
#define BITNET_HIP_ATTN_PARTIAL 4 /* chunk records only (bitnet_hip_attention_decode_partial_dev): out_dev and qact_out unused */
int bitnet_hip_attention_decode_q_dev(const float *qkv_dev, const float *rope_sin_dev, const float *rope_cos_dev,
                                      void *kcache_dev, void *vcache_dev, size_t n_heads, size_t n_kv_heads,
                                      size_t head_dim, size_t max_pos, const int32_t *pos_dev, float *scratch_dev,
                                      int flags, float *out_dev, void *qact_out, void *stream);
/* bitnet_hip_attention_prefill_dev filling f16 decode caches */
int bitnet_hip_attention_prefill_kv16_dev(const float *qkv_dev, const float *rope_sin_dev, const float *rope_cos_dev,
                                          void *kcache_f16_dev, void *vcache_f16_dev, size_t n_heads, size_t n_kv_heads,
                                          size_t head_dim, size_t max_pos, size_t seq_len, void *workspace_dev,
                                          size_t workspace_bytes, float *out_dev, void *stream);
/* The same call with a flag word: BITNET_HIP_ATTN_CACHE_F16 = the f16 caches of _kv16_dev; BITNET_HIP_ATTN_OUT_F16 = `out` receives
 * f16 rows [seq_len][n_heads * head_dim] (the input format of bitnet_hip_matmul_f16_dev: the o-projection reads them as they are). */
#define BITNET_HIP_ATTN_CACHE_F16 1
#define BITNET_HIP_ATTN_OUT_F16 2
int bitnet_hip_attention_prefill_flags_dev(const float *qkv, const float *rope_sin, const float *rope_cos, void *kcache, void *vcache, size_t n_heads,
                                           size_t n_kv_heads, size_t head_dim, size_t max_pos, size_t seq_len, void *workspace, size_t workspace_bytes,
                                           void *out, int flags, void *stream);
/* bitnet_hip_gemv_attn_merge_dev (short contexts) with the QAct outputs of gemv_q_dev */
int bitnet_hip_gemv_attn_merge_q_dev(bitnet_hip_weights_t w, const float *attn_scratch_dev, size_t n_heads,
                                     size_t n_kv_heads, size_t max_pos, const int32_t *pos_dev, float *y_dev,
                                     const float *residual_dev, void *qact_out, const float *gamma_out_dev,
                                     double *stats_out, void *stream);
/* The same attention for a whole prompt of seq_len tokens on a FRESH cache (positions
 * 0..seq_len-1): RoPE, cache append, causal GQA softmax attention (T:398-543 with the causal
 * mask T:452-470).  qkv_dev: [seq_len, n_heads*D + 2*n_kv*D]; out_dev: [seq_len, n_heads*D].
 * q, k, v and the probabilities go through the matrix cores in f16 (f32 accumulate); the
 * cache receives the exact f32 k, v.  Semantics of the reference's stub
 * fused_attention_hip(q,k,v,output,seq_len,&cfg{num_heads,head_dim,causal=true,scale=1/sqrt(d)})
 * (K/rocm/attention.rs:54-65) with grouped KV heads. */
size_t bitnet_hip_attention_prefill_workspace_bytes(size_t n_heads, size_t n_kv_heads, size_t seq_len);
int bitnet_hip_attention_prefill_dev(const float *qkv_dev, const float *rope_sin_dev,
                                     const float *rope_cos_dev, float *kcache_dev, float *vcache_dev,
                                     size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos,
                                     size_t seq_len, void *workspace_dev, size_t workspace_bytes,
                                     float *out_dev, void *stream);
/* Token-parallel form of the call above (long prompts split over several GPUs): this GPU holds
 * n_q of the prompt's query rows, q_dev [n_q, ld_q] (head h at columns [128 h, 128 h+128)), in
 * 64-row blocks whose absolute first positions are q_block_pos_dev[b] (multiples of 64; only
 * the last block may be partial), and the raw k|v rows of the WHOLE context gathered from all
 * ranks in absolute order, kv_dev [n_ctx, ld_kv] (k heads, then v heads).  Every query attends
 * to positions <= its own; the cache receives all n_ctx positions.  Needs no collective itself:
 * the caller gathers k|v (RCCL all-gather) between the projection and this call. */
size_t bitnet_hip_attention_prefill_sharded_workspace_bytes(size_t n_heads, size_t n_kv_heads, size_t n_q, size_t n_ctx);
int bitnet_hip_attention_prefill_sharded_dev(const float *q_dev, size_t ld_q, const int32_t *q_block_pos_dev,
                                             size_t n_q, const float *kv_dev, size_t ld_kv, size_t n_ctx,
                                             const float *rope_sin_dev, const float *rope_cos_dev,
                                             float *kcache_dev, float *vcache_dev, size_t n_heads,
                                             size_t n_kv_heads, size_t head_dim, size_t max_pos,
                                             void *workspace_dev, size_t workspace_bytes, float *out_dev,
                                             void *stream);
/* The same with the k|v rows exactly as an all-gather over `world` ranks left them: rank r's block holds its two zigzag
 * chunks (chunk c of 2 * world equal chunks belongs to rank c or 2 * world - 1 - c: every rank gets the same amount of
 * causal attention), each row = k heads then v heads, f32 or -- kv_is_f16 -- f16 (half the bytes on the wire; the f32 decode
 * cache then holds the f16-rounded k, v of the prompt).  No scatter pass on the host side: the position -> row map is
 * applied where the rows are read.  n_ctx % (2 * world * 64) == 0.  bitnet_hip_pack_cols_dev builds the send buffer: the
 * columns [col0, col0 + ncols) of `rows` f32 rows, compact, as f32 or f16. */
int bitnet_hip_attention_prefill_gathered_dev(const float *q_dev, size_t ld_q, const int32_t *q_block_pos_dev, size_t n_q,
                                              const void *kv_gathered_dev, size_t n_ctx, size_t world, int kv_is_f16,
                                              const float *rope_sin_dev, const float *rope_cos_dev, void *kcache_dev,
                                              void *vcache_dev, int cache_f16, size_t n_heads, size_t n_kv_heads,
                                              size_t head_dim, size_t max_pos, void *workspace_dev, size_t workspace_bytes,
                                              float *out_dev, void *stream);
/* The same in two steps, so that the all-gather of k|v can run on ANOTHER stream beside the query-side work: phase 1 prepares the
 * query slabs only (RoPE, f16; kv_gathered_dev is not read), phase 2 does the k / v slabs, the cache fill and the attention on
 * the workspace phase 1 left; phase 0 = both (the call above).  cache_f16 is a flag word here: BITNET_HIP_ATTN_CACHE_F16 (bit 0, as
 * above) | BITNET_HIP_ATTN_OUT_F16 (bit 1: out_dev receives f16 rows, the o-projection's FUSE_X_F16 / f16-chain input). */
int bitnet_hip_attention_prefill_gathered_phase_dev(const float *q_dev, size_t ld_q, const int32_t *q_block_pos_dev, size_t n_q,
                                                    const void *kv_gathered_dev, size_t n_ctx, size_t world, int kv_is_f16,
                                                    const float *rope_sin_dev, const float *rope_cos_dev, void *kcache_dev,
                                                    void *vcache_dev, int cache_f16, size_t n_heads, size_t n_kv_heads,
                                                    size_t head_dim, size_t max_pos, void *workspace_dev, size_t workspace_bytes,
                                                    float *out_dev, int phase, void *stream);
int bitnet_hip_pack_cols_dev(const float *src_dev, size_t ld, size_t col0, size_t ncols, size_t rows, void *dst_dev,
                             int as_f16, void *stream);
/* bytes of scratch_dev attention_decode_dev needs (per-chunk softmax partials) */
size_t bitnet_hip_attention_scratch_bytes(size_t n_kv_heads, size_t max_pos);
/* TransformerModel::logits, tied embeddings (T:1599-1630): logits = LN(x) . E^T, E the f16
 * table [vocab, hidden], f32 accumulate; gamma_dev == NULL skips the final norm (T:1589).
 * scratch_dev: >= 8 * n_workgroups bytes (argmax partials); token_dev (nullable) receives
 * the greedy token (argmax, lowest index on ties, NaN -> -inf:
 * crates/bitnet-cli/src/sampling.rs:45-49,189-202); with p = *pos_dev (nullable):
 * history_dev[p+1] = token unless p+1 < *n_forced_dev (a prompt token already sits there),
 * then *pos_dev = p+1. */
int bitnet_hip_logits_f16_dev(const void *table_f16_dev, const float *x_dev, const float *gamma_dev,
                              float eps, size_t hidden, size_t vocab, float *logits_dev,
                              void *scratch_dev, size_t n_workgroups, int32_t *token_dev,
                              int32_t *pos_dev, int32_t *history_dev, const int32_t *n_forced_dev,
                              void *stream);
/* argmax over a device vector with the same tie/NaN rules; scratch as above. */
int bitnet_hip_argmax_dev(const float *v_dev, size_t n, void *scratch_dev, size_t n_workgroups,
                          int32_t *token_dev, void *stream);

/* ---- measurement aid -------------------------------------------------------
 * Measured HBM read ceiling of the device (quote the roofline fraction against the vendor
 * figure AND a measured stream ceiling): a read-only streaming kernel (non-temporal 16-byte loads, one
 * workgroup round per 2 MiB) over `bytes` of freshly written device memory (bytes >= 1 GiB keeps L2 and
 * the 256 MB MALL out of it), `iters` timed passes with HIP events on the given stream.  Writes the
 * best pass in GB/s (1e9 B/s) to *best_gbs and the mean to *mean_gbs.  Allocates and frees its buffer. */
int bitnet_hip_hbm_read_ceiling(size_t bytes, int iters, double *best_gbs, double *mean_gbs, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BITNET_HIP_H */
