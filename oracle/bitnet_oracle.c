/*
 * bitnet_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See bitnet_oracle.h for scope, citations and the pinning story.
 *
 * Build: see oracle/Makefile.  -ffp-contract=off is mandatory: the reference
 * is Rust, which never contracts `acc += a * w` into an FMA, so neither may
 * this restatement (the AVX2 path uses explicit _mm256_fmadd_ps exactly where
 * the reference does).
 */
#include "bitnet_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int fail(char *err, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#include <stdarg.h>
static int fail(char *err, const char *fmt, ...) {
    if (err) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, BO_ERRLEN, fmt, ap);
        va_end(ap);
    }
    return 1;
}

static size_t div_ceil(size_t a, size_t b) { return (a + b - 1) / b; }
static size_t min_sz(size_t a, size_t b) { return a < b ? a : b; }

/* ======================================================================= */
/* QK256                                                                   */
/* ======================================================================= */

/* Q/i2s_qk256.rs:159-168 */
void bo_unpack_qk256_block(const uint8_t qs64[64], uint8_t out[256]) {
    for (size_t i = 0; i < 64; ++i) {
        uint8_t b = qs64[i];
        size_t base = i * 4;
        out[base] = b & 0x03;
        out[base + 1] = (b >> 2) & 0x03;
        out[base + 2] = (b >> 4) & 0x03;
        out[base + 3] = (b >> 6) & 0x03;
    }
}

/* Q/i2s_qk256.rs:139-146 */
float bo_code_to_f32(uint8_t code) {
    static const float LUT[4] = {-2.0f, -1.0f, 1.0f, 2.0f};
    return LUT[code & 3];
}

/* Q/i2s_qk256.rs:196-274 (the BITNET_QUANT_SANITY probe :216,229-259 only
 * prints; it does not change the result and is not restated). */
float bo_gemv_qk256_row(const uint8_t *qs_row, const float *x, size_t cols) {
    size_t blocks = div_ceil(cols, BO_QK256_BLOCK);
    float acc = 0.0f;
    uint8_t codes[BO_QK256_BLOCK];
    size_t col = 0;
    for (size_t b = 0; b < blocks; ++b) {
        bo_unpack_qk256_block(qs_row + b * BO_QK256_PACKED_BYTES, codes);
        size_t take = min_sz(BO_QK256_BLOCK, cols - col);
        for (size_t j = 0; j < take; ++j) {
            float w = bo_code_to_f32(codes[j]);
            acc += w * x[col + j];
        }
        col += take;
        if (col >= cols) break;
    }
    return acc;
}

/* Q/i2s_qk256.rs:293-321 */
int bo_gemv_qk256_scalar(const uint8_t *qs, size_t qs_len, const float *x, size_t x_len,
                         float *y, size_t y_len, size_t rows, size_t cols,
                         size_t row_stride_bytes, char *err) {
    if (y_len != rows) return fail(err, "I2S_QK256: y_out length %zu != rows %zu", y_len, rows);
    if (x_len < cols) return fail(err, "I2S_QK256: x length %zu < cols %zu", x_len, cols);
    size_t expected_total = rows * row_stride_bytes;
    if (qs_len < expected_total)
        return fail(err, "I2S_QK256: data too short: %zu < %zu", qs_len, expected_total);
    /* debug_assert in gemv_qk256_row (:200-207): row slice length must equal
     * ceil(cols/256)*64.  Surfaced here as an error instead of a panic. */
    if (row_stride_bytes != div_ceil(cols, BO_QK256_BLOCK) * BO_QK256_PACKED_BYTES)
        return fail(err, "I2S_QK256: row bytes mismatch: got %zu, expected %zu for %zu cols",
                    row_stride_bytes, div_ceil(cols, BO_QK256_BLOCK) * BO_QK256_PACKED_BYTES,
                    cols);
    for (size_t r = 0; r < rows; ++r)
        y[r] = bo_gemv_qk256_row(qs + r * row_stride_bytes, x, cols);
    return 0;
}

void bo_gemv_qk256_f64(const uint8_t *qs, const float *x, double *y, size_t rows, size_t cols,
                       size_t row_stride_bytes) {
    for (size_t r = 0; r < rows; ++r) {
        const uint8_t *row = qs + r * row_stride_bytes;
        double acc = 0.0;
        for (size_t j = 0; j < cols; ++j) {
            uint8_t code = (row[j / 4] >> (2 * (j % 4))) & 3;
            acc += (double)bo_code_to_f32(code) * (double)x[j];
        }
        y[r] = acc;
    }
}

int bo_have_avx2(void) {
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
}

/* Q/i2s_qk256_avx2.rs:42-61 */
__attribute__((target("avx2,fma"))) static inline __m256
decode_8_weights_avx2(uint8_t byte0, uint8_t byte1, __m256i shifts, __m256i mask_03,
                      __m256i two, __m256i one) {
    int32_t packed = (int32_t)byte0 | ((int32_t)byte1 << 16);
    __m256i broadcast = _mm256_set1_epi32(packed);
    __m256i codes = _mm256_and_si256(_mm256_srlv_epi32(broadcast, shifts), mask_03);
    __m256i shifted = _mm256_sub_epi32(codes, two);
    __m256i correction = _mm256_and_si256(_mm256_srli_epi32(codes, 1), one);
    return _mm256_cvtepi32_ps(_mm256_add_epi32(shifted, correction));
}

/* Q/i2s_qk256_avx2.rs:81-229 */
__attribute__((target("avx2,fma"))) static float
gemv_qk256_row_avx2(const uint8_t *qs_row, const float *x, size_t cols) {
    size_t blocks_needed = div_ceil(cols, BO_QK256_BLOCK);
    const __m256i shifts = _mm256_setr_epi32(0, 2, 4, 6, 16, 18, 20, 22);
    const __m256i mask_03 = _mm256_set1_epi32(0x03);
    const __m256i two = _mm256_set1_epi32(2);
    const __m256i one = _mm256_set1_epi32(1);
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
    __m256 acc2 = _mm256_setzero_ps(), acc3 = _mm256_setzero_ps();
    float scalar_acc = 0.0f;
    size_t col = 0;
    for (size_t blk_idx = 0; blk_idx < blocks_needed; ++blk_idx) {
        const uint8_t *blk = qs_row + blk_idx * BO_QK256_PACKED_BYTES;
        size_t take = min_sz(BO_QK256_BLOCK, cols - col);
        if (blk_idx + 1 < blocks_needed) {
            _mm_prefetch((const char *)(blk + BO_QK256_PACKED_BYTES), _MM_HINT_T0);
            _mm_prefetch((const char *)(x + col + BO_QK256_BLOCK), _MM_HINT_T0);
            _mm_prefetch((const char *)(x + col + BO_QK256_BLOCK + 16), _MM_HINT_T0);
        }
        size_t j = 0;
        while (j + 32 <= take) {
            size_t pi = j / 4;
            __m256 w0 = decode_8_weights_avx2(blk[pi], blk[pi + 1], shifts, mask_03, two, one);
            __m256 w1 = decode_8_weights_avx2(blk[pi + 2], blk[pi + 3], shifts, mask_03, two, one);
            __m256 w2 = decode_8_weights_avx2(blk[pi + 4], blk[pi + 5], shifts, mask_03, two, one);
            __m256 w3 = decode_8_weights_avx2(blk[pi + 6], blk[pi + 7], shifts, mask_03, two, one);
            size_t xj = col + j;
            __m256 x0 = _mm256_loadu_ps(x + xj);
            __m256 x1 = _mm256_loadu_ps(x + xj + 8);
            __m256 x2 = _mm256_loadu_ps(x + xj + 16);
            __m256 x3 = _mm256_loadu_ps(x + xj + 24);
            acc0 = _mm256_fmadd_ps(w0, x0, acc0);
            acc1 = _mm256_fmadd_ps(w1, x1, acc1);
            acc2 = _mm256_fmadd_ps(w2, x2, acc2);
            acc3 = _mm256_fmadd_ps(w3, x3, acc3);
            j += 32;
        }
        while (j + 8 <= take) {
            size_t pi = j / 4;
            __m256 w = decode_8_weights_avx2(blk[pi], blk[pi + 1], shifts, mask_03, two, one);
            __m256 xv = _mm256_loadu_ps(x + col + j);
            acc0 = _mm256_fmadd_ps(w, xv, acc0);
            j += 8;
        }
        while (j < take) {
            uint8_t packed_byte = blk[j / 4];
            unsigned shift = (unsigned)(j % 4) * 2;
            uint8_t code = (packed_byte >> shift) & 0x03;
            float w = code == 0 ? -2.0f : code == 1 ? -1.0f : code == 2 ? 1.0f : 2.0f;
            scalar_acc += w * x[col + j];
            j += 1;
        }
        col += take;
        if (col >= cols) break;
    }
    __m256 sum01 = _mm256_add_ps(acc0, acc1);
    __m256 sum23 = _mm256_add_ps(acc2, acc3);
    __m256 acc = _mm256_add_ps(sum01, sum23);
    __m128 hi = _mm256_extractf128_ps(acc, 1);
    __m128 lo = _mm256_castps256_ps128(acc);
    __m128 sum128 = _mm_add_ps(hi, lo);
    __m128 sum64 = _mm_hadd_ps(sum128, sum128);
    __m128 sum32 = _mm_hadd_ps(sum64, sum64);
    return _mm_cvtss_f32(sum32) + scalar_acc;
}

static int avx2_checks(size_t qs_len, size_t x_len, size_t y_len, size_t rows, size_t cols,
                       size_t row_stride_bytes, char *err) {
    if (y_len != rows) return fail(err, "AVX2: y_out length %zu != rows %zu", y_len, rows);
    if (x_len < cols) return fail(err, "AVX2: x length %zu < cols %zu", x_len, cols);
    size_t expected_total = rows * row_stride_bytes;
    if (qs_len < expected_total)
        return fail(err, "AVX2: data too short: %zu < %zu", qs_len, expected_total);
    if (row_stride_bytes != div_ceil(cols, BO_QK256_BLOCK) * BO_QK256_PACKED_BYTES)
        return fail(err, "AVX2: row bytes mismatch: got %zu, expected %zu for %zu cols",
                    row_stride_bytes, div_ceil(cols, BO_QK256_BLOCK) * BO_QK256_PACKED_BYTES,
                    cols);
    return 0;
}

/* Q/i2s_qk256_avx2.rs:254-295 */
__attribute__((target("avx2,fma"))) static void
gemv_rows_avx2(const uint8_t *qs, const float *x, float *y, size_t r0, size_t r1, size_t rows,
               size_t cols, size_t row_stride_bytes) {
    for (size_t r = r0; r < r1; ++r) {
        if (r + 1 < rows)
            _mm_prefetch((const char *)(qs + (r + 1) * row_stride_bytes), _MM_HINT_T0);
        y[r] = gemv_qk256_row_avx2(qs + r * row_stride_bytes, x, cols);
    }
}

int bo_gemv_qk256_avx2(const uint8_t *qs, size_t qs_len, const float *x, size_t x_len,
                       float *y, size_t y_len, size_t rows, size_t cols,
                       size_t row_stride_bytes, char *err) {
    if (!bo_have_avx2()) {
        fail(err, "AVX2 implementation only available on x86_64 with AVX2+FMA");
        return 2;
    }
    int rc = avx2_checks(qs_len, x_len, y_len, rows, cols, row_stride_bytes, err);
    if (rc) return rc;
    gemv_rows_avx2(qs, x, y, 0, rows, rows, cols, row_stride_bytes);
    return 0;
}

/* ---- persistent host thread pool (test infrastructure) ------------------------------------
 * The threaded entry points used to create and join their threads per call (210 GEMVs per
 * token x n threads).  bo_parallel_for keeps up to 63 workers alive for the life of the process;
 * a call deals tasks 0 .. n_tasks-1 to at most n_threads participants (the caller included).
 * A task is a fixed partition of independent outputs (row range / head range), so results do
 * not depend on which thread runs it.  One parallel region at a time (callers are serialised). */
#define BO_POOL_MAX 63
static struct {
    pthread_mutex_t mu, call_mu;
    pthread_cond_t go, done;
    pthread_t th[BO_POOL_MAX];
    int n_workers;
    unsigned long gen;
    bo_task_fn fn;
    void *arg;
    int n_tasks, next, finished, max_extra, joined;
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, NULL, NULL, 0, 0, 0, 0, 0};

static void pool_drain(void) { /* g_pool.mu held on entry and on exit */
    while (g_pool.next < g_pool.n_tasks) {
        int t = g_pool.next++;
        bo_task_fn fn = g_pool.fn;
        void *arg = g_pool.arg;
        pthread_mutex_unlock(&g_pool.mu);
        fn(arg, t);
        pthread_mutex_lock(&g_pool.mu);
        if (++g_pool.finished == g_pool.n_tasks) pthread_cond_broadcast(&g_pool.done);
    }
}
static void *pool_worker(void *p) {
    (void)p;
    unsigned long seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.gen == seen) pthread_cond_wait(&g_pool.go, &g_pool.mu);
        seen = g_pool.gen;
        if (g_pool.joined < g_pool.max_extra) { /* at most n_threads - 1 workers beside the caller */
            ++g_pool.joined;
            pool_drain();
        }
    }
    return NULL;
}
int bo_pool_workers(void) { return g_pool.n_workers; }
void bo_parallel_for(int n_tasks, int n_threads, bo_task_fn fn, void *arg) {
    if (n_tasks <= 0) return;
    if (n_threads > n_tasks) n_threads = n_tasks;
    if (n_threads > BO_POOL_MAX + 1) n_threads = BO_POOL_MAX + 1;
    if (n_threads < 2) {
        for (int t = 0; t < n_tasks; ++t) fn(arg, t);
        return;
    }
    pthread_mutex_lock(&g_pool.call_mu);
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.n_workers < n_threads - 1) {
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        if (pthread_create(&g_pool.th[g_pool.n_workers], &at, pool_worker, NULL) != 0) {
            pthread_attr_destroy(&at);
            break; /* fewer helpers than asked for: the caller drains what is left */
        }
        pthread_attr_destroy(&at);
        ++g_pool.n_workers;
    }
    g_pool.fn = fn, g_pool.arg = arg, g_pool.n_tasks = n_tasks, g_pool.next = 0, g_pool.finished = 0;
    g_pool.max_extra = n_threads - 1, g_pool.joined = 0;
    ++g_pool.gen;
    pthread_cond_broadcast(&g_pool.go);
    pool_drain();
    while (g_pool.finished < g_pool.n_tasks) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    g_pool.n_tasks = 0; /* a late waker finds nothing to take */
    pthread_mutex_unlock(&g_pool.mu);
    pthread_mutex_unlock(&g_pool.call_mu);
}

struct mt_arg {
    const uint8_t *qs;
    const float *x;
    float *y;
    size_t per, rows, cols, stride;
};
static void mt_task(void *p, int t) {
    struct mt_arg *a = (struct mt_arg *)p;
    size_t r0 = min_sz(a->rows, a->per * (size_t)t), r1 = min_sz(a->rows, r0 + a->per);
    if (r0 < r1) gemv_rows_avx2(a->qs, a->x, a->y, r0, r1, a->rows, a->cols, a->stride);
}

/* gemv_qk256_avx2 (Q/i2s_qk256_avx2.rs:254-295) with the rows partitioned over n_threads host threads (the reference's row loop
 * is single-threaded: this is the "all cores" baseline of SURVEY 8d).  Rows are independent dot products: bit-identical to one thread. */
int bo_gemv_qk256_avx2_mt(const uint8_t *qs, size_t qs_len, const float *x, size_t x_len,
                          float *y, size_t y_len, size_t rows, size_t cols,
                          size_t row_stride_bytes, int n_threads, char *err) {
    if (!bo_have_avx2()) {
        fail(err, "AVX2 implementation only available on x86_64 with AVX2+FMA");
        return 2;
    }
    int rc = avx2_checks(qs_len, x_len, y_len, rows, cols, row_stride_bytes, err);
    if (rc) return rc;
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > rows) n_threads = (int)(rows ? rows : 1);
    struct mt_arg a = {qs, x, y, div_ceil(rows, (size_t)n_threads), rows, cols, row_stride_bytes};
    bo_parallel_for(n_threads, n_threads, mt_task, &a);
    return 0;
}

/* Q/i2s_qk256.rs:346-372 */
int bo_gemv_qk256(const uint8_t *qs, size_t qs_len, const float *x, size_t x_len, float *y,
                  size_t y_len, size_t rows, size_t cols, size_t row_stride_bytes, char *err) {
    if (bo_have_avx2())
        return bo_gemv_qk256_avx2(qs, qs_len, x, x_len, y, y_len, rows, cols, row_stride_bytes,
                                  err);
    return bo_gemv_qk256_scalar(qs, qs_len, x, x_len, y, y_len, rows, cols, row_stride_bytes,
                                err);
}

/* Q/i2s_qk256.rs:85-106 */
int bo_i2s_qk256_new(size_t rows, size_t cols, size_t qs_len, size_t *row_stride_bytes,
                     char *err) {
    size_t blocks_per_row = div_ceil(cols, BO_QK256_BLOCK);
    size_t stride = blocks_per_row * BO_QK256_PACKED_BYTES;
    size_t expected = rows * stride;
    const size_t TOLERANCE = 128;
    size_t diff = qs_len > expected ? qs_len - expected : expected - qs_len;
    if (diff > TOLERANCE)
        return fail(err,
                    "I2SQk256NoScale: data size mismatch: got %zu bytes, expected %zu for "
                    "%zux%zu matrix. Check tensor orientation: QK256 requires [out_dim, in_dim] "
                    "layout.",
                    qs_len, expected, rows, cols);
    if (row_stride_bytes) *row_stride_bytes = stride;
    return 0;
}

/* Q/qk256_dispatch.rs:41-99 */
int bo_qk256_dispatch_gemv_scalar(float *output, size_t rows, size_t cols,
                                  const uint8_t *packed, size_t packed_len,
                                  const float *scales, size_t scales_len,
                                  const float *activations, char *err) {
    if (cols % BO_QK256_BLOCK != 0)
        return fail(err, "Cols must be multiple of QK256=%d", BO_QK256_BLOCK);
    size_t blocks_per_row = cols / BO_QK256_BLOCK;
    if (packed_len != rows * cols / 4) return fail(err, "Packed weight size mismatch");
    if (scales_len != rows * blocks_per_row) return fail(err, "Scales length mismatch");
    for (size_t row = 0; row < rows; ++row) {
        float row_sum = 0.0f;
        for (size_t b = 0; b < blocks_per_row; ++b) {
            size_t global_block = row * blocks_per_row + b;
            float scale = scales[global_block];
            size_t byte_offset = global_block * BO_QK256_BLOCK / 4;
            size_t act_offset = b * BO_QK256_BLOCK;
            float block_sum = 0.0f;
            for (size_t e = 0; e < BO_QK256_BLOCK; ++e) {
                uint8_t two_bit = (packed[byte_offset + e / 4] >> ((e % 4) * 2)) & 3;
                float v = two_bit == 0 ? -1.0f : two_bit == 1 ? 0.0f : two_bit == 2 ? 1.0f : -1.0f;
                block_sum += v * activations[act_offset + e];
            }
            row_sum += block_sum * scale;
        }
        output[row] = row_sum;
    }
    return 0;
}

/* ======================================================================= */
/* ternary I2_S                                                            */
/* ======================================================================= */

/* K/cpu/quantized_matmul.rs:19-27 */
int8_t bo_decode_i2s(uint8_t bits) {
    switch (bits & 0x03) {
    case 0: return 0;
    case 1: return 1;
    case 3: return -1;
    default: return 0;
    }
}

/* K/cpu/quantized_matmul.rs:30-41 */
uint8_t bo_pack_i2s(const int8_t vals[4]) {
    uint8_t byte = 0;
    for (int i = 0; i < 4; ++i) {
        uint8_t code = vals[i] == 1 ? 1 : vals[i] == -1 ? 3 : 0;
        byte |= (uint8_t)(code << (i * 2));
    }
    return byte;
}

/* K/cpu/quantized_matmul.rs:204-256 */
static int validate_matmul_args(size_t act_len, size_t w_len, size_t scales_len, size_t out_len,
                                size_t m, size_t n, size_t k, size_t block_size, char *err) {
    if (block_size == 0) return fail(err, "block_size must be > 0");
    if (m == 0 || n == 0 || k == 0)
        return fail(err, "dimensions must be > 0: m=%zu, n=%zu, k=%zu", m, n, k);
    size_t packed_k = div_ceil(k, 4), nbk = div_ceil(k, block_size);
    if (act_len < m * k)
        return fail(err, "activations too small: expected %zu, got %zu", m * k, act_len);
    if (w_len < packed_k * n)
        return fail(err, "weights_packed too small: expected %zu, got %zu", packed_k * n, w_len);
    if (scales_len < n * nbk)
        return fail(err, "scales too small: expected %zu, got %zu", n * nbk, scales_len);
    if (out_len < m * n)
        return fail(err, "output too small: expected %zu, got %zu", m * n, out_len);
    return 0;
}

/* K/cpu/quantized_matmul.rs:57-96 */
int bo_i2s_matmul_f32(const float *act, size_t act_len, const uint8_t *w, size_t w_len,
                      const float *scales, size_t scales_len, float *out, size_t out_len,
                      size_t m, size_t n, size_t k, size_t block_size, char *err) {
    int rc = validate_matmul_args(act_len, w_len, scales_len, out_len, m, n, k, block_size, err);
    if (rc) return rc;
    size_t packed_k = div_ceil(k, 4), nbk = div_ceil(k, block_size);
    for (size_t i = 0; i < out_len; ++i) out[i] = 0.0f;
    for (size_t row = 0; row < m; ++row) {
        const float *a_row = act + row * k;
        for (size_t col = 0; col < n; ++col) {
            float acc = 0.0f;
            for (size_t blk = 0; blk < nbk; ++blk) {
                size_t blk_start = blk * block_size;
                size_t blk_end = min_sz(blk_start + block_size, k);
                float scale = scales[col * nbk + blk];
                for (size_t idx = blk_start; idx < blk_end; ++idx) {
                    uint8_t bits = (w[col * packed_k + idx / 4] >> ((idx % 4) * 2)) & 0x03;
                    float wv = (float)bo_decode_i2s(bits) * scale;
                    acc += a_row[idx] * wv;
                }
            }
            out[row * n + col] = acc;
        }
    }
    return 0;
}

/* K/cpu/quantized_matmul.rs:155-200 */
int bo_i2s_matmul_blocked(const float *act, size_t act_len, const uint8_t *w, size_t w_len,
                          const float *scales, size_t scales_len, float *out, size_t out_len,
                          size_t m, size_t n, size_t k, size_t block_size, char *err) {
    int rc = validate_matmul_args(act_len, w_len, scales_len, out_len, m, n, k, block_size, err);
    if (rc) return rc;
    if (block_size > 256) return fail(err, "block_size %zu exceeds w_blk[256]", block_size);
    size_t packed_k = div_ceil(k, 4), nbk = div_ceil(k, block_size);
    for (size_t i = 0; i < out_len; ++i) out[i] = 0.0f;
    for (size_t blk = 0; blk < nbk; ++blk) {
        size_t blk_start = blk * block_size;
        size_t blk_end = min_sz(blk_start + block_size, k);
        for (size_t col = 0; col < n; ++col) {
            float scale = scales[col * nbk + blk];
            int8_t w_blk[256];
            memset(w_blk, 0, sizeof(w_blk));
            for (size_t idx = blk_start; idx < blk_end; ++idx) {
                uint8_t bits = (w[col * packed_k + idx / 4] >> ((idx % 4) * 2)) & 0x03;
                w_blk[idx - blk_start] = bo_decode_i2s(bits);
            }
            for (size_t row = 0; row < m; ++row) {
                float acc = 0.0f;
                const float *a_row = act + row * k + blk_start;
                for (size_t i = 0; i < blk_end - blk_start; ++i)
                    acc += a_row[i] * (float)w_blk[i];
                out[row * n + col] += acc * scale;
            }
        }
    }
    return 0;
}

/* K/cpu/quantized_matmul.rs:105-148 */
int bo_dequantize_and_matmul(const float *act, size_t act_len, const uint8_t *w, size_t w_len,
                             const float *scales, size_t scales_len, float *out,
                             size_t out_len, size_t m, size_t n, size_t k, size_t block_size,
                             char *err) {
    int rc = validate_matmul_args(act_len, w_len, scales_len, out_len, m, n, k, block_size, err);
    if (rc) return rc;
    size_t packed_k = div_ceil(k, 4), nbk = div_ceil(k, block_size);
    float *wf = (float *)calloc(k * n, sizeof(float));
    if (!wf) return fail(err, "oracle: out of memory");
    for (size_t col = 0; col < n; ++col)
        for (size_t blk = 0; blk < nbk; ++blk) {
            size_t blk_start = blk * block_size;
            size_t blk_end = min_sz(blk_start + block_size, k);
            float scale = scales[col * nbk + blk];
            for (size_t idx = blk_start; idx < blk_end; ++idx) {
                uint8_t bits = (w[col * packed_k + idx / 4] >> ((idx % 4) * 2)) & 0x03;
                wf[idx * n + col] = (float)bo_decode_i2s(bits) * scale;
            }
        }
    for (size_t i = 0; i < out_len; ++i) out[i] = 0.0f;
    for (size_t row = 0; row < m; ++row)
        for (size_t col = 0; col < n; ++col) {
            float acc = 0.0f;
            for (size_t idx = 0; idx < k; ++idx) acc += act[row * k + idx] * wf[idx * n + col];
            out[row * n + col] = acc;
        }
    free(wf);
    return 0;
}

/* ======================================================================= */
/* KernelProvider fallback                                                 */
/* ======================================================================= */

/* K/cpu/fallback.rs:39-83 */
int bo_matmul_i2s(const int8_t *a, size_t a_len, const uint8_t *b, size_t b_len, float *c,
                  size_t c_len, size_t m, size_t n, size_t k, char *err) {
    if (a_len != m * k)
        return fail(err, "Matrix A dimension mismatch: expected %zu, got %zu", m * k, a_len);
    if (b_len != k * n)
        return fail(err, "Matrix B dimension mismatch: expected %zu, got %zu", k * n, b_len);
    if (c_len != m * n)
        return fail(err, "Matrix C dimension mismatch: expected %zu, got %zu", m * n, c_len);
    for (size_t i = 0; i < c_len; ++i) c[i] = 0.0f;
    for (size_t i = 0; i < m; ++i)
        for (size_t j = 0; j < n; ++j) {
            float sum = 0.0f;
            for (size_t l = 0; l < k; ++l) sum += (float)a[i * k + l] * (float)b[l * n + j];
            c[i * n + j] = sum;
        }
    return 0;
}

/* QuantizedLinear::quantized_matmul_i2s, crates/bitnet-inference/src/layers/quantized_linear.rs:704-744 with its helpers:
 *   quantize_input_i2s      :1762-1773   x.clamp(-2.0, 1.0).round() as i8  (f32::round: half away from zero; NaN as i8 = 0)
 *   prepare_quantized_weights_i2s :769-776 + unpack_2bit_values :1738-1759   (code - 2) + 2 = the RAW code 0..3, LSB first,
 *                                                     `numel` = k * n values, handed to matmul_i2s as its [k, n] operand
 *   provider.matmul_i2s     :722-731      FallbackKernel, K/cpu/fallback.rs:39-83 (above)
 *   apply_quantization_scales :779-802    scale index = col if scales.len() == out_features, else
 *                                         min((col * in_features) / block_size, scales.len() - 1); input_scale = 1.0;
 *                                         scales.get(idx).unwrap_or(1.0) */
int bo_quantized_matmul_i2s(const float *input, size_t in_len, const uint8_t *packed, size_t packed_len, const float *scales,
                            size_t n_scales, size_t block_size, float *out, size_t out_len, size_t m, size_t n, size_t k,
                            char *err) {
    if (in_len != m * k) return fail(err, "Matrix A dimension mismatch: expected %zu, got %zu", m * k, in_len);
    size_t numel = k * n;
    int8_t *a = (int8_t *)malloc(in_len ? in_len : 1);
    uint8_t *b = (uint8_t *)malloc(numel ? numel : 1);
    size_t nb = 0;
    for (size_t i = 0; i < in_len; ++i) {
        float x = input[i];
        float c = x < -2.0f ? -2.0f : (x > 1.0f ? 1.0f : x); /* f32::clamp: NaN stays NaN */
        a[i] = (c != c) ? (int8_t)0 : (int8_t)roundf(c);
    }
    for (size_t i = 0; i < packed_len && nb < numel; ++i)
        for (int shift = 0; shift < 8 && nb < numel; shift += 2) b[nb++] = (uint8_t)((((packed[i] >> shift) & 3) - 2) + 2);
    int rc = bo_matmul_i2s(a, in_len, b, nb, out, out_len, m, n, k, err); /* a short weight buffer fails B's length check */
    free(a);
    free(b);
    if (rc) return rc;
    for (size_t row = 0; row < m; ++row)
        for (size_t col = 0; col < n; ++col) {
            size_t idx = col;
            if (n_scales != n) {
                size_t weight_idx = col * k;
                idx = weight_idx / block_size;
                if (n_scales == 0 || idx > n_scales - 1) idx = n_scales ? n_scales - 1 : 0;
            }
            float scale = idx < n_scales ? scales[idx] : 1.0f;
            out[row * n + col] *= 1.0f * scale;
        }
    return 0;
}

/* K/cpu/fallback.rs:102-159.  NB: OR-packs into `output` without clearing it
 * first (:153) -- callers pass a zeroed buffer; restated as-is. */
int bo_quantize_i2s(const float *input, size_t input_len, uint8_t *output, size_t output_len,
                    float *scales, size_t scales_len, char *err) {
    const size_t BLOCK = 32;
    size_t num_blocks = div_ceil(input_len, BLOCK);
    if (output_len < input_len / 4)
        return fail(err, "Output buffer too small for I2_S: expected %zu, got %zu",
                    input_len / 4, output_len);
    if (scales_len < num_blocks)
        return fail(err, "Scales buffer too small: expected %zu, got %zu", num_blocks,
                    scales_len);
    for (size_t b = 0; b < num_blocks; ++b) {
        size_t start = b * BLOCK, end = min_sz(start + BLOCK, input_len);
        float max_val = 0.0f;
        for (size_t i = start; i < end; ++i) {
            float a = fabsf(input[i]);
            /* f32::max: NaN-ignoring max */
            max_val = fmaxf(max_val, a);
        }
        float scale = max_val > 1e-8f ? max_val / 1.5f : 1.0f;
        scales[b] = scale;
        for (size_t i = start; i < end; ++i) {
            float normalized = input[i] / scale;
            uint8_t q = normalized > 0.5f ? 1 : normalized < -0.5f ? 3 : 0;
            size_t byte_idx = i / 4;
            unsigned bit_offset = (unsigned)(i % 4) * 2;
            if (byte_idx < output_len) output[byte_idx] |= (uint8_t)(q << bit_offset);
        }
    }
    return 0;
}

/* ======================================================================= */
/* block dequant with inline f16 scale (M/quant/i2s.rs)                    */
/* ======================================================================= */

float bo_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t mant = h & 0x3ffu;
    uint32_t f;
    if (exp == 0) {
        if (mant == 0) {
            f = sign;
        } else {
            int e = -1;
            do {
                mant <<= 1;
                e++;
            } while ((mant & 0x400u) == 0);
            mant &= 0x3ffu;
            f = sign | ((uint32_t)(127 - 15 - e) << 23) | (mant << 13);
        }
    } else if (exp == 31) {
        f = sign | 0x7f800000u | (mant << 13);
    } else {
        f = sign | ((exp + 127 - 15) << 23) | (mant << 13);
    }
    float r;
    memcpy(&r, &f, sizeof(r));
    return r;
}

static float clampf_rust(float v, float lo, float hi) {
    /* f32::clamp: NaN stays NaN */
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

/* M/quant/i2s.rs:66-140 (env form) == :144-200 (cfg form) */
void bo_i2s_dequant_block(float *dst, const uint8_t *qbits, size_t n, uint16_t scale_bits,
                          int inv_scale, float k) {
    static const float LUT[4] = {-2.0f, -1.0f, 1.0f, 2.0f}; /* I2SMapping::Sym :46 */
    float s = fabsf(bo_f16_to_f32(scale_bits));
    if (inv_scale) s = s < 1e-8f ? 1.0f : 1.0f / s;
    s *= k;
    s = clampf_rust(s, 1e-3f, 1e3f);
    float scaled[4] = {s * LUT[0], s * LUT[1], s * LUT[2], s * LUT[3]};
    size_t chunks = n / 4, rem = n % 4;
    for (size_t c = 0; c < chunks; ++c) {
        uint8_t b = qbits[c];
        dst[c * 4] = scaled[b & 3];
        dst[c * 4 + 1] = scaled[(b >> 2) & 3];
        dst[c * 4 + 2] = scaled[(b >> 4) & 3];
        dst[c * 4 + 3] = scaled[(b >> 6) & 3];
    }
    if (rem) {
        uint8_t b = qbits[chunks];
        for (size_t i = 0; i < rem; ++i) dst[chunks * 4 + i] = scaled[(b >> (i * 2)) & 3];
    }
}

/* M/quant/i2s.rs:205-209 */
size_t bo_i2s_expected_bytes(size_t rows, size_t cols, size_t block) {
    size_t bpr = div_ceil(cols, block);
    size_t qbits = div_ceil(block, 4);
    return (rows * bpr) * (qbits + 2);
}

/* M/quant/i2s.rs:211-214 */
size_t bo_i2s_infer_block_size(size_t bytes, size_t rows, size_t cols) {
    static const size_t cands[4] = {256, 128, 64, 32};
    for (int i = 0; i < 4; ++i)
        if (bo_i2s_expected_bytes(rows, cols, cands[i]) == bytes) return cands[i];
    return 0;
}

/* Walks blocks exactly as :292-346 (full) / :372-431 (partial) do, writing
 * either row-major or (transposed) column-major. */
static int dequant_walk(const uint8_t *bytes, size_t bytes_len, size_t rows, size_t cols,
                        size_t block, int inv, float k, int transposed, int partial,
                        size_t available_blocks, float *out, char *err) {
    size_t bpr = div_ceil(cols, block);
    size_t off = 0, processed = 0;
    float scratch[256];
    for (size_t r = 0; r < rows; ++r) {
        size_t c = 0;
        for (size_t b = 0; b < bpr; ++b) {
            if (partial && processed == available_blocks) return 0;
            size_t n = min_sz(cols - c, block);
            if (n == 0) break;
            size_t qlen = (n + 3) / 4;
            /* :563 / :880: the transposed partial walker tests a whole
             * block's bytes; the row-major ones (:391, :308) test qlen + 2. */
            size_t need = (partial && transposed) ? div_ceil(block, 4) + 2 : qlen + 2;
            if (off + need > bytes_len) {
                if (partial) return 0;
                return fail(err, "I2_S: buffer bounds exceeded at offset %zu", off);
            }
            const uint8_t *q = bytes + off;
            off += qlen;
            uint16_t sb = (uint16_t)(bytes[off] | (bytes[off + 1] << 8));
            off += 2;
            bo_i2s_dequant_block(scratch, q, n, sb, inv, k);
            if (transposed)
                for (size_t i = 0; i < n; ++i) out[(c + i) * rows + r] = scratch[i];
            else
                memcpy(out + r * cols + c, scratch, n * sizeof(float));
            c += n;
            if (transposed) {
                /* :580 / :897 count every block */
                processed += 1;
            } else {
                /* :425-429: the row-major partial walker breaks out of the
                 * block loop when the row completes BEFORE counting that
                 * block -- restated as-is. */
                if (c >= cols) break;
                processed += 1;
            }
        }
    }
    return 0;
}

int bo_i2s_dequantize_to_f32(const uint8_t *bytes, size_t bytes_len, size_t rows, size_t cols,
                             int inv, float k, int transposed, float *out, char *err) {
    for (size_t i = 0; i < rows * cols; ++i) out[i] = 0.0f;
    size_t block;
    if (!transposed) {
        /* :239-273 */
        block = 256;
        if (bytes_len != bo_i2s_expected_bytes(rows, cols, block)) {
            size_t b = bo_i2s_infer_block_size(bytes_len, rows, cols);
            if (b) {
                block = b;
            } else {
                size_t per_block = div_ceil(block, 4) + 2;
                return dequant_walk(bytes, bytes_len, rows, cols, block, inv, k, 0, 1,
                                    bytes_len / per_block, out, err);
            }
        }
    } else {
        /* :448-483: infer first, default 256 */
        block = bo_i2s_infer_block_size(bytes_len, rows, cols);
        if (!block) {
            block = 256;
            size_t per_block = div_ceil(block, 4) + 2;
            return dequant_walk(bytes, bytes_len, rows, cols, block, inv, k, 1, 1,
                                bytes_len / per_block, out, err);
        }
    }
    if (bytes_len != rows * div_ceil(cols, block) * (div_ceil(block, 4) + 2))
        return fail(err, "I2_S: internal size mismatch for block=%zu", block);
    return dequant_walk(bytes, bytes_len, rows, cols, block, inv, k, transposed, 0, 0, out, err);
}

/* ======================================================================= */
/* Q/utils.rs 2-bit pack/unpack (c-2 map) + block scale                    */
/* ======================================================================= */

/* Q/utils.rs:57-74 */
void bo_pack_2bit_values(const int8_t *values, size_t n, uint8_t *packed) {
    size_t nbytes = div_ceil(n, 4);
    for (size_t b = 0; b < nbytes; ++b) {
        uint8_t byte = 0;
        for (size_t i = 0; i < 4 && b * 4 + i < n; ++i) {
            int v = values[b * 4 + i];
            if (v < -2) v = -2;
            if (v > 1) v = 1;
            byte |= (uint8_t)((uint8_t)(v + 2) << (i * 2));
        }
        packed[b] = byte;
    }
}

/* Q/utils.rs:76-91 */
void bo_unpack_2bit_values(const uint8_t *packed, size_t packed_len, size_t output_len,
                           int8_t *values) {
    size_t cnt = 0;
    for (size_t b = 0; b < packed_len; ++b)
        for (int i = 0; i < 4; ++i) {
            if (cnt >= output_len) break;
            values[cnt++] = (int8_t)((int)((packed[b] >> (i * 2)) & 3) - 2);
        }
}

/* Q/simd_ops.rs:170-238,336-365: out = q as f32 * scale[block] */
void bo_dequantize_blocks(const int8_t *q, size_t n, const float *scales, size_t block_size,
                          float *out) {
    for (size_t i = 0; i < n; ++i) out[i] = (float)q[i] * scales[i / block_size];
}
