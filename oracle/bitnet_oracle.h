/*
 * bitnet_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's (EffortlessMetrics/BitNet-rs) CPU
 * arithmetic for the I2_S hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker.
 * Nothing under bitnet-rs_amd/ links, loads or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Prefixes:
 *   Q/ = crates/bitnet-quantization/src/      K/ = crates/bitnet-kernels/src/
 *   M/ = crates/bitnet-models/src/            T  = crates/bitnet-transformer/src/lib.rs
 *
 * Pinning: tests/test_oracle_kat.py replays every known-answer
 * test the reference holds for these functions; oracle/_ref (the reference's
 * own vendored ggml-quants.c, compiled where it lies) pins the LUT, the
 * LSB-first extraction and the f16 scale conversion.
 *
 * All functions return 0 on success, non-zero on error, and write a message
 * into err (if err != NULL) carrying the same substrings the reference's
 * error text carries.
 */
#ifndef BITNET_ORACLE_H
#define BITNET_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BO_QK256_BLOCK 256
#define BO_QK256_PACKED_BYTES 64
#define BO_ERRLEN 256

/* ---- QK256 ("GgmlQk256NoScale") : Q/i2s_qk256.rs ---------------------- */

/* Q/i2s_qk256.rs:159-168  64 B -> 256 codes, LSB-first */
void bo_unpack_qk256_block(const uint8_t qs64[64], uint8_t out_codes256[256]);
/* Q/i2s_qk256.rs:139-146  LUT {-2,-1,+1,+2} */
float bo_code_to_f32(uint8_t code);
/* Q/i2s_qk256.rs:196-274  one row, left-to-right f32 accumulate, tail via take */
float bo_gemv_qk256_row(const uint8_t *qs_row, const float *x, size_t cols);
/* Q/i2s_qk256.rs:293-321  scalar multi-row GEMV with the reference's checks */
int bo_gemv_qk256_scalar(const uint8_t *qs, size_t qs_len, const float *x, size_t x_len,
                         float *y, size_t y_len, size_t rows, size_t cols,
                         size_t row_stride_bytes, char *err);
/* Q/i2s_qk256_avx2.rs:254-295 (+:81-229 row kernel, :42-61 decode) -- same
 * intrinsic sequence.  Returns 2 if the host CPU lacks AVX2+FMA. */
int bo_gemv_qk256_avx2(const uint8_t *qs, size_t qs_len, const float *x, size_t x_len,
                       float *y, size_t y_len, size_t rows, size_t cols,
                       size_t row_stride_bytes, char *err);
/* Q/i2s_qk256.rs:346-372  runtime dispatch: AVX2 if present else scalar */
int bo_gemv_qk256(const uint8_t *qs, size_t qs_len, const float *x, size_t x_len,
                  float *y, size_t y_len, size_t rows, size_t cols,
                  size_t row_stride_bytes, char *err);
int bo_have_avx2(void);
/* Test-infrastructure thread pool: tasks 0 .. n_tasks-1 run on at most n_threads threads (the
 * caller included); workers persist for the life of the process (created on first use). */
typedef void (*bo_task_fn)(void *arg, int task);
void bo_parallel_for(int n_tasks, int n_threads, bo_task_fn fn, void *arg);
int bo_pool_workers(void);
/* Row-partitioned over n_threads pool threads, each running the AVX2 row kernel
 * ("reference kernel, parallelised" -- BASELINE.md section 3 mode (b)). */
int bo_gemv_qk256_avx2_mt(const uint8_t *qs, size_t qs_len, const float *x, size_t x_len,
                          float *y, size_t y_len, size_t rows, size_t cols,
                          size_t row_stride_bytes, int n_threads, char *err);
/* f64-accumulated truth for error budgeting (not a reference function). */
void bo_gemv_qk256_f64(const uint8_t *qs, const float *x, double *y, size_t rows,
                       size_t cols, size_t row_stride_bytes);
/* Q/i2s_qk256.rs:85-106  I2SQk256NoScale::new size check (+-128 B slack).
 * On success writes row_stride_bytes. */
int bo_i2s_qk256_new(size_t rows, size_t cols, size_t qs_len, size_t *row_stride_bytes,
                     char *err);
/* Q/qk256_dispatch.rs:41-99  legacy scalar kernel: map {-1,0,+1,-1} x f32 block scale */
int bo_qk256_dispatch_gemv_scalar(float *output, size_t rows, size_t cols,
                                  const uint8_t *packed, size_t packed_len,
                                  const float *scales, size_t scales_len,
                                  const float *activations, char *err);

/* ---- ternary I2_S family : K/cpu/quantized_matmul.rs ------------------ */

/* K/cpu/quantized_matmul.rs:19-27  0->0, 1->+1, 3->-1, 2->0 */
int8_t bo_decode_i2s(uint8_t bits);
/* K/cpu/quantized_matmul.rs:30-41 */
uint8_t bo_pack_i2s(const int8_t vals[4]);
/* K/cpu/quantized_matmul.rs:57-96 (+ validation :204-256) */
int bo_i2s_matmul_f32(const float *act, size_t act_len, const uint8_t *w, size_t w_len,
                      const float *scales, size_t scales_len, float *out, size_t out_len,
                      size_t m, size_t n, size_t k, size_t block_size, char *err);
/* K/cpu/quantized_matmul.rs:155-200 */
int bo_i2s_matmul_blocked(const float *act, size_t act_len, const uint8_t *w, size_t w_len,
                          const float *scales, size_t scales_len, float *out, size_t out_len,
                          size_t m, size_t n, size_t k, size_t block_size, char *err);
/* K/cpu/quantized_matmul.rs:105-148 */
int bo_dequantize_and_matmul(const float *act, size_t act_len, const uint8_t *w, size_t w_len,
                             const float *scales, size_t scales_len, float *out,
                             size_t out_len, size_t m, size_t n, size_t k, size_t block_size,
                             char *err);

/* ---- KernelProvider (FallbackKernel) : K/cpu/fallback.rs -------------- */

/* K/cpu/fallback.rs:39-83   C = A_i8 . B_u8, B unpacked row-major [k,n] */
int bo_quantized_matmul_i2s(const float *input, size_t in_len, const uint8_t *packed, size_t packed_len, const float *scales,
                            size_t n_scales, size_t block_size, float *out, size_t out_len, size_t m, size_t n, size_t k,
                            char *err);
int bo_matmul_i2s(const int8_t *a, size_t a_len, const uint8_t *b, size_t b_len, float *c,
                  size_t c_len, size_t m, size_t n, size_t k, char *err);
/* K/cpu/fallback.rs:102-159  block 32, scale = absmax/1.5, OR-packs into output */
int bo_quantize_i2s(const float *input, size_t input_len, uint8_t *output, size_t output_len,
                    float *scales, size_t scales_len, char *err);

/* ---- block dequant with inline f16 scale : M/quant/i2s.rs ------------- */

/* f16 bits -> f32 (half::f16::to_f32; exact) */
float bo_f16_to_f32(uint16_t h);
/* M/quant/i2s.rs:66-140 / :144-200  one block, Sym LUT, s = clamp(|f16|[^-1] * k, 1e-3, 1e3) */
void bo_i2s_dequant_block(float *dst, const uint8_t *qbits, size_t n, uint16_t scale_bits,
                          int inv_scale, float k);
/* M/quant/i2s.rs:205-214 */
size_t bo_i2s_expected_bytes(size_t rows, size_t cols, size_t block);
size_t bo_i2s_infer_block_size(size_t bytes, size_t rows, size_t cols); /* 0 = none */
/* M/quant/i2s.rs:237-274 (+:276-348, :350-434) and _with_cfg twins :591-769.
 * out has rows*cols floats.  transposed!=0 follows :448-585 / :774-902
 * (output logical shape [cols, rows]). */
int bo_i2s_dequantize_to_f32(const uint8_t *bytes, size_t bytes_len, size_t rows, size_t cols,
                             int inv, float k, int transposed, float *out, char *err);

/* ---- 2-bit pack/unpack with c-2 mapping : Q/utils.rs ------------------ */

/* Q/utils.rs:57-74 */
void bo_pack_2bit_values(const int8_t *values, size_t n, uint8_t *packed);
/* Q/utils.rs:76-91 */
void bo_unpack_2bit_values(const uint8_t *packed, size_t packed_len, size_t output_len,
                           int8_t *values);
/* Q/simd_ops.rs:170-238 scalar semantics: out[i] = q[i] as f32 * scales[i / block] */
void bo_dequantize_blocks(const int8_t *q, size_t n, const float *scales, size_t block_size,
                          float *out);

/* ---- decode step around the GEMVs : transformer_oracle.c ---------------- */

typedef struct bo_model_cfg {
    int hidden, n_layers, n_heads, n_kv_heads, head_dim, ffn, vocab, max_pos;
    float eps, rope_theta;
} bo_model_cfg;

/* crates/bitnet-rope/src/lib.rs:59-93 */
void bo_rope_build_tables(int dim, int max_seq_len, float base, float *sin_out, float *cos_out);
/* T:134-163 (split halves), one head vector */
void bo_rope_apply(float *x, int dim, const float *sin_row, const float *cos_row);
/* T:67-100: LayerNorm without bias, with mean subtraction */
void bo_layernorm(const float *x, const float *w, float eps, int n, float *out);
/* K/rocm/rmsnorm.rs:1-12 formula */
void bo_rmsnorm(const float *x, const float *w, float eps, int n, float *out);
float bo_silu(float v);
/* crates/bitnet-cli/src/sampling.rs:189-202 */
int bo_argmax(const float *logits, size_t n);
/* T:398-543 for one query row against t_k cached positions */
void bo_attention_decode(const float *q, const float *kc, const float *vc, int n_heads,
                         int n_kv_heads, int dim, int max_pos, int t_k, float *out);

/* Model = pointers to caller-owned arrays (nothing is copied). */
void *bo_model_create(const bo_model_cfg *cfg, int n_threads);
void bo_model_destroy(void *m);
int bo_model_set_layer(void *m, int layer, const float *attn_norm, const float *ffn_norm,
                       const uint8_t *q, const uint8_t *k, const uint8_t *v, const uint8_t *o,
                       const uint8_t *gate, const uint8_t *up, const uint8_t *down);
/* projections as dense f32 [out, in] (32-element I2_S flavours are dequantised at load: M/gguf_simple.rs:1260-1285) */
int bo_model_set_layer_dense(void *m, int layer, const float *attn_norm, const float *ffn_norm, const float *const *w7);
/* the same dense matrices held as ternary codes + one f32 scale per (row, block): W = t(code) * scale, multiplied on the fly in
 * the dense loop's order (= i2s_matmul_f32, K/cpu/quantized_matmul.rs:57-96) */
int bo_model_set_layer_ternary(void *m, int layer, const float *attn_norm, const float *ffn_norm,
                               const uint8_t *const *codes7, const float *const *scales7, int block);
void bo_model_set_globals(void *m, const uint16_t *embed_f16, const float *final_norm);
void *bo_kv_create(void *m);
void bo_kv_destroy(void *kv);
void bo_kv_reset(void *kv);
int bo_kv_len(void *kv);
/* T:1599-1630 */
void bo_logits(void *m, const float *hidden, float *logits);
/* one token: T:1482-1504 body (embed -> 30 blocks -> final norm -> logits) */
int bo_model_step(void *m, void *kv, int token, float *hidden_out, float *logits_out, float *trace);

#ifdef __cplusplus
}
#endif
#endif /* BITNET_ORACLE_H */
