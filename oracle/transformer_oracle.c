/*
 * transformer_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference's decode step around the I2_S GEMVs
 * (SURVEY.md 8a row a15, 8f rank 1-2).  T = crates/bitnet-transformer/src/lib.rs.
 *
 *   LayerNorm (no bias, WITH mean subtraction)   T:67-100 (candle_nn::LayerNorm
 *       slow path: xc = x - mean; xc / sqrt(mean(xc^2) + eps) * weight)
 *   RoPE tables                                  crates/bitnet-rope/src/lib.rs:59-93
 *   RoPE apply, split halves                     T:134-163
 *   KV cache append                              T:1171-1202
 *   GQA attention, 1/sqrt(d) scale, causal mask,
 *       max-subtracted softmax, P.V              T:398-543, T:704-719
 *   SiLU-gated FFN                               T:751-798
 *   block: pre-norm attn + residual, pre-norm FFN + residual   T:977-1134
 *   per-token stepping with a KV cache           T:1435-1551 (forward_full), T:1557-1597
 *   tied-embedding logits                        T:1599-1630
 *   greedy argmax, lowest index on ties          crates/bitnet-cli/src/sampling.rs:189-202
 *   projections through gemv_qk256 per row       T:589-702, T:829-942
 *
 * Sums are plain left-to-right f32 (candle's own reduction order is not
 * specified; parity is by tolerance / cosine).
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "bitnet_oracle.h"

typedef struct {
    const float *attn_norm, *ffn_norm;
    const uint8_t *q, *k, *v, *o, *gate, *up, *down; /* QK256 bytes, [out, ceil(in/256)*64] */
    /* dense f32 [out, in] per projection (q,k,v,o,gate,up,down) when the loader dequantised
     * the tensor at load time (32-element I2_S flavours, M/gguf_simple.rs:1260-1285; then the
     * plain candle Linear path T:546-588) */
    const float *dense[7];
    /* the same dense matrix held as the file holds it: ternary codes (K/cuda/quantized_matmul.rs:19-27: 0 -> 0, 1 -> +1,
     * 2 -> 0, 3 -> -1), four per byte LSB-first, each output row's codes contiguous, and one f32 scale per (row, block):
     * W[r, c] = t(code) * scale[r, c / block].  proj() multiplies W[r, c] on the fly in the dense loop's order -- the loop of
     * i2s_matmul_f32, K/cpu/quantized_matmul.rs:57-96 -- so a full-size model needs 26 MB per layer instead of 278 MB of f32. */
    const uint8_t *tern[7];
    const float *tern_scales[7];
    int tern_block;
} bo_layer;

typedef struct {
    bo_model_cfg cfg;
    bo_layer *layers;
    const uint16_t *embed_f16; /* [vocab, hidden] */
    const float *final_norm;
    float *rope_sin, *rope_cos; /* [max_pos, head_dim/2] */
    int n_threads;
} bo_model;

typedef struct {
    float *k, *v; /* [layers][kv_heads][max_pos][head_dim] */
    int seq_len;
} bo_kv;

/* crates/bitnet-rope/src/lib.rs:59-93 */
void bo_rope_build_tables(int dim, int max_seq_len, float base, float *sin_out, float *cos_out) {
    int half = dim / 2;
    float *inv_freq = (float *)malloc(sizeof(float) * (size_t)half);
    for (int i = 0; i < half; ++i) inv_freq[i] = 1.0f / powf(base, (2.0f * (float)i) / (float)dim);
    for (int pos = 0; pos < max_seq_len; ++pos)
        for (int i = 0; i < half; ++i) {
            float angle = (float)pos * inv_freq[i];
            sin_out[(size_t)pos * half + i] = sinf(angle);
            cos_out[(size_t)pos * half + i] = cosf(angle);
        }
    free(inv_freq);
}

/* T:67-100 */
void bo_layernorm(const float *x, const float *w, float eps, int n, float *out) {
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) sum += x[i];
    float mean = sum / (float)n;
    float ss = 0.0f;
    for (int i = 0; i < n; ++i) {
        float d = x[i] - mean;
        ss += d * d;
    }
    float denom = sqrtf(ss / (float)n + eps);
    for (int i = 0; i < n; ++i) out[i] = (x[i] - mean) / denom * w[i];
}

/* K/rocm/rmsnorm.rs:1-12: rms = sqrt(mean(x^2) + eps); out = x / rms * gamma */
void bo_rmsnorm(const float *x, const float *w, float eps, int n, float *out) {
    float ss = 0.0f;
    for (int i = 0; i < n; ++i) ss += x[i] * x[i];
    float rms = sqrtf(ss / (float)n + eps);
    for (int i = 0; i < n; ++i) out[i] = x[i] / rms * w[i];
}

/* T:134-163, one head vector of `dim` at `pos` */
void bo_rope_apply(float *x, int dim, const float *sin_row, const float *cos_row) {
    int half = dim / 2;
    for (int i = 0; i < half; ++i) {
        float x0 = x[i], x1 = x[half + i];
        x[i] = x0 * cos_row[i] - x1 * sin_row[i];
        x[half + i] = x0 * sin_row[i] + x1 * cos_row[i];
    }
}

float bo_silu(float v) { return v / (1.0f + expf(-v)); }

/* crates/bitnet-cli/src/sampling.rs:189-202 (NaN -> -inf first, :45-49) */
int bo_argmax(const float *logits, size_t n) {
    size_t best = 0;
    float best_val = -INFINITY;
    for (size_t i = 0; i < n; ++i) {
        float v = isnan(logits[i]) ? -INFINITY : logits[i];
        if (v > best_val || (v == best_val && i < best)) {
            best_val = v;
            best = i;
        }
    }
    return (int)best;
}

/* one query vector against a cache of t_k positions: T:426-533 for seq_len 1.
 * q: [n_heads*dim] (already rotated), kc/vc: [kv_heads][max_pos][dim]. */
void bo_attention_decode(const float *q, const float *kc, const float *vc, int n_heads,
                         int n_kv_heads, int dim, int max_pos, int t_k, float *out) {
    int group = n_heads / n_kv_heads;
    float scale = 1.0f / sqrtf((float)dim);
    float *scores = (float *)malloc(sizeof(float) * (size_t)t_k);
    for (int h = 0; h < n_heads; ++h) {
        int kvh = h / group;
        const float *qh = q + (size_t)h * dim;
        const float *kh = kc + (size_t)kvh * max_pos * dim;
        const float *vh = vc + (size_t)kvh * max_pos * dim;
        float mx = -INFINITY;
        for (int j = 0; j < t_k; ++j) {
            float s = 0.0f;
            for (int d = 0; d < dim; ++d) s += qh[d] * kh[(size_t)j * dim + d];
            s = s * scale; /* affine(scale, 0) T:449; mask is all zeros for the last row */
            scores[j] = s;
            if (s > mx) mx = s;
        }
        float sum = 0.0f;
        for (int j = 0; j < t_k; ++j) {
            scores[j] = expf(scores[j] - mx);
            sum += scores[j];
        }
        for (int d = 0; d < dim; ++d) {
            float acc = 0.0f;
            for (int j = 0; j < t_k; ++j) acc += (scores[j] / sum) * vh[(size_t)j * dim + d];
            out[(size_t)h * dim + d] = acc;
        }
    }
    free(scores);
}

/* The same arithmetic with the HEADS dealt to threads: a head's scores, softmax and weighted sum are computed by one thread in
 * the order above, so every output element is bit-identical to bo_attention_decode (heads are independent: T:426-533 loops
 * over them).  Test infrastructure for long contexts (4096-token oracle runs), not a different algorithm. */
struct attn_arg {
    const float *q, *kc, *vc;
    int n_heads, n_kv_heads, dim, max_pos, t_k, h0, h1;
    float *out;
};
static void attn_heads(const struct attn_arg *a) {
    int group = a->n_heads / a->n_kv_heads, dim = a->dim, t_k = a->t_k;
    float scale = 1.0f / sqrtf((float)dim);
    float *scores = (float *)malloc(sizeof(float) * (size_t)t_k);
    for (int h = a->h0; h < a->h1; ++h) {
        int kvh = h / group;
        const float *qh = a->q + (size_t)h * dim;
        const float *kh = a->kc + (size_t)kvh * a->max_pos * dim;
        const float *vh = a->vc + (size_t)kvh * a->max_pos * dim;
        float mx = -INFINITY;
        for (int j = 0; j < t_k; ++j) {
            float s = 0.0f;
            for (int d = 0; d < dim; ++d) s += qh[d] * kh[(size_t)j * dim + d];
            s = s * scale;
            scores[j] = s;
            if (s > mx) mx = s;
        }
        float sum = 0.0f;
        for (int j = 0; j < t_k; ++j) {
            scores[j] = expf(scores[j] - mx);
            sum += scores[j];
        }
        for (int d = 0; d < dim; ++d) {
            float acc = 0.0f;
            for (int j = 0; j < t_k; ++j) acc += (scores[j] / sum) * vh[(size_t)j * dim + d];
            a->out[(size_t)h * dim + d] = acc;
        }
    }
    free(scores);
}
struct attn_job {
    struct attn_arg base;
    int per;
};
static void attn_task(void *p, int t) {
    struct attn_job *j = (struct attn_job *)p;
    struct attn_arg a = j->base;
    a.h0 = t * j->per;
    a.h1 = a.h0 + j->per > a.n_heads ? a.n_heads : a.h0 + j->per;
    if (a.h0 < a.h1) attn_heads(&a);
}
void bo_attention_decode_mt(const float *q, const float *kc, const float *vc, int n_heads, int n_kv_heads, int dim,
                            int max_pos, int t_k, float *out, int n_threads) {
    if (n_threads < 2 || t_k < 128) {
        bo_attention_decode(q, kc, vc, n_heads, n_kv_heads, dim, max_pos, t_k, out);
        return;
    }
    int nt = n_threads > 64 ? 64 : n_threads;
    if (nt > n_heads) nt = n_heads;
    struct attn_job job = {{q, kc, vc, n_heads, n_kv_heads, dim, max_pos, t_k, 0, 0, out}, (n_heads + nt - 1) / nt};
    bo_parallel_for(nt, nt, attn_task, &job);
}

/* dense f32 projection, rows dealt to threads: each row's dot product is the loop of proj() below, bit for bit */
struct dense_arg {
    const float *W, *x;
    const uint8_t *codes; /* ternary kind: codes + scales instead of W */
    const float *scales;
    int block;
    float *y;
    int per, rows, cols;
};
/* K/cuda/quantized_matmul.rs:19-27 */
static inline float tern_value(unsigned bits) { return bits == 1 ? 1.0f : bits == 3 ? -1.0f : 0.0f; }
static void dense_rows(const struct dense_arg *a, int r0, int r1) {
    if (a->W) {
        for (int r = r0; r < r1; ++r) {
            float acc = 0.0f;
            for (int c = 0; c < a->cols; ++c) acc += a->x[c] * a->W[(size_t)r * a->cols + c];
            a->y[r] = acc;
        }
        return;
    }
    /* the loop of i2s_matmul_f32 (K/cpu/quantized_matmul.rs:57-96; bo_i2s_matmul_f32): wv = t(code) * scale, acc += a * wv,
     * columns left to right -- the dense loop above on W[r, c] = t * scale, value for value */
    const size_t packed_k = ((size_t)a->cols + 3) / 4, nbk = ((size_t)a->cols + (size_t)a->block - 1) / (size_t)a->block;
    static const float TV[4] = {0.0f, 1.0f, 0.0f, -1.0f};
    int r = r0;
    /* four rows at a time: four independent accumulator chains (each row's own sum is still strictly left to right; rows are
     * independent outputs), which is what lets a host core run the 2 G weights of a full-size token in a fraction of a second */
    for (; r + 4 <= r1; r += 4) {
        const uint8_t *w0 = a->codes + (size_t)r * packed_k, *w1 = w0 + packed_k, *w2 = w1 + packed_k, *w3 = w2 + packed_k;
        const float *s0 = a->scales + (size_t)r * nbk, *s1 = s0 + nbk, *s2 = s1 + nbk, *s3 = s2 + nbk;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (size_t blk = 0; blk < nbk; ++blk) {
            size_t c0 = blk * (size_t)a->block, c1 = c0 + (size_t)a->block > (size_t)a->cols ? (size_t)a->cols : c0 + (size_t)a->block;
            const float k0 = s0[blk], k1 = s1[blk], k2 = s2[blk], k3 = s3[blk];
            for (size_t c = c0; c < c1; ++c) {
                const unsigned sh = (unsigned)(c & 3) * 2;
                const float xv = a->x[c];
                a0 += xv * (TV[(w0[c >> 2] >> sh) & 3u] * k0);
                a1 += xv * (TV[(w1[c >> 2] >> sh) & 3u] * k1);
                a2 += xv * (TV[(w2[c >> 2] >> sh) & 3u] * k2);
                a3 += xv * (TV[(w3[c >> 2] >> sh) & 3u] * k3);
            }
        }
        a->y[r] = a0, a->y[r + 1] = a1, a->y[r + 2] = a2, a->y[r + 3] = a3;
    }
    for (; r < r1; ++r) {
        const uint8_t *wr = a->codes + (size_t)r * packed_k;
        const float *sr = a->scales + (size_t)r * nbk;
        float acc = 0.0f;
        for (size_t blk = 0; blk < nbk; ++blk) {
            size_t c0 = blk * (size_t)a->block, c1 = c0 + (size_t)a->block > (size_t)a->cols ? (size_t)a->cols : c0 + (size_t)a->block;
            float scale = sr[blk];
            for (size_t c = c0; c < c1; ++c) {
                float wv = tern_value((wr[c / 4] >> ((c % 4) * 2)) & 3u) * scale;
                acc += a->x[c] * wv;
            }
        }
        a->y[r] = acc;
    }
}
static void dense_task(void *p, int t) {
    const struct dense_arg *a = (const struct dense_arg *)p;
    int r0 = t * a->per, r1 = r0 + a->per > a->rows ? a->rows : r0 + a->per;
    if (r0 < r1) dense_rows(a, r0, r1);
}
static void dense_mt(struct dense_arg *a, int n_threads) {
    int nt = n_threads > 64 ? 64 : n_threads;
    if (nt < 2 || a->rows < 256) {
        dense_rows(a, 0, a->rows);
        return;
    }
    a->per = (a->rows + nt - 1) / nt;
    bo_parallel_for(nt, nt, dense_task, a);
}

void *bo_model_create(const bo_model_cfg *cfg, int n_threads) {
    bo_model *m = (bo_model *)calloc(1, sizeof(bo_model));
    m->cfg = *cfg;
    m->layers = (bo_layer *)calloc((size_t)cfg->n_layers, sizeof(bo_layer));
    int half = cfg->head_dim / 2;
    m->rope_sin = (float *)malloc(sizeof(float) * (size_t)cfg->max_pos * half);
    m->rope_cos = (float *)malloc(sizeof(float) * (size_t)cfg->max_pos * half);
    bo_rope_build_tables(cfg->head_dim, cfg->max_pos, cfg->rope_theta, m->rope_sin, m->rope_cos);
    m->n_threads = n_threads < 1 ? 1 : n_threads;
    return m;
}

void bo_model_destroy(void *mp) {
    bo_model *m = (bo_model *)mp;
    if (!m) return;
    free(m->layers);
    free(m->rope_sin);
    free(m->rope_cos);
    free(m);
}

int bo_model_set_layer(void *mp, int layer, const float *attn_norm, const float *ffn_norm,
                       const uint8_t *q, const uint8_t *k, const uint8_t *v, const uint8_t *o,
                       const uint8_t *gate, const uint8_t *up, const uint8_t *down) {
    bo_model *m = (bo_model *)mp;
    if (layer < 0 || layer >= m->cfg.n_layers) return 1;
    bo_layer *L = &m->layers[layer];
    L->attn_norm = attn_norm;
    L->ffn_norm = ffn_norm;
    L->q = q;
    L->k = k;
    L->v = v;
    L->o = o;
    L->gate = gate;
    L->up = up;
    L->down = down;
    return 0;
}

int bo_model_set_layer_dense(void *mp, int layer, const float *attn_norm, const float *ffn_norm,
                             const float *const *w7) {
    bo_model *m = (bo_model *)mp;
    if (layer < 0 || layer >= m->cfg.n_layers) return 1;
    bo_layer *L = &m->layers[layer];
    L->attn_norm = attn_norm;
    L->ffn_norm = ffn_norm;
    for (int i = 0; i < 7; ++i) L->dense[i] = w7[i];
    return 0;
}

int bo_model_set_layer_ternary(void *mp, int layer, const float *attn_norm, const float *ffn_norm,
                               const uint8_t *const *codes7, const float *const *scales7, int block) {
    bo_model *m = (bo_model *)mp;
    if (layer < 0 || layer >= m->cfg.n_layers || block < 1) return 1;
    bo_layer *L = &m->layers[layer];
    L->attn_norm = attn_norm;
    L->ffn_norm = ffn_norm;
    for (int i = 0; i < 7; ++i) L->tern[i] = codes7[i], L->tern_scales[i] = scales7[i];
    L->tern_block = block;
    return 0;
}

void bo_model_set_globals(void *mp, const uint16_t *embed_f16, const float *final_norm) {
    bo_model *m = (bo_model *)mp;
    m->embed_f16 = embed_f16;
    m->final_norm = final_norm;
}

void *bo_kv_create(void *mp) {
    bo_model *m = (bo_model *)mp;
    bo_kv *kv = (bo_kv *)calloc(1, sizeof(bo_kv));
    size_t n = (size_t)m->cfg.n_layers * m->cfg.n_kv_heads * m->cfg.max_pos * m->cfg.head_dim;
    kv->k = (float *)calloc(n, sizeof(float));
    kv->v = (float *)calloc(n, sizeof(float));
    return kv;
}
void bo_kv_destroy(void *p) {
    bo_kv *kv = (bo_kv *)p;
    if (!kv) return;
    free(kv->k);
    free(kv->v);
    free(kv);
}
void bo_kv_reset(void *p) { ((bo_kv *)p)->seq_len = 0; }
int bo_kv_len(void *p) { return ((bo_kv *)p)->seq_len; }

/* forward_qk256 (T:589-702): y = gemv_qk256(W, x) -- the reference dispatches to
 * AVX2 when present (Q/i2s_qk256.rs:355-368). */
static int proj(const bo_model *m, const bo_layer *L, int which, const uint8_t *w, const float *x, float *y, int rows,
                int cols) {
    if (L->dense[which] || L->tern[which]) { /* x . W^T, one f32 dot product per output */
        struct dense_arg a = {L->dense[which], x, L->tern[which], L->tern_scales[which], L->tern_block, y, 0, rows, cols};
        dense_mt(&a, m->n_threads);
        return 0;
    }
    size_t stride = (size_t)((cols + 255) / 256) * 64;
    char err[BO_ERRLEN];
    if (bo_have_avx2() && m->n_threads > 1)
        return bo_gemv_qk256_avx2_mt(w, (size_t)rows * stride, x, (size_t)cols, y, (size_t)rows,
                                     (size_t)rows, (size_t)cols, stride, m->n_threads, err);
    return bo_gemv_qk256(w, (size_t)rows * stride, x, (size_t)cols, y, (size_t)rows, (size_t)rows,
                         (size_t)cols, stride, err);
}

struct logit_arg {
    const bo_model *m;
    const float *h;
    float *out;
    int per;
};
static void logit_task(void *p, int t) {
    struct logit_arg *a = (struct logit_arg *)p;
    int H = a->m->cfg.hidden, V = a->m->cfg.vocab;
    int v0 = t * a->per, v1 = v0 + a->per > V ? V : v0 + a->per;
    for (int v = v0; v < v1; ++v) {
        const uint16_t *row = a->m->embed_f16 + (size_t)v * H;
        float acc = 0.0f;
        for (int k = 0; k < H; ++k) acc += a->h[k] * bo_f16_to_f32(row[k]);
        a->out[v] = acc;
    }
}

/* T:1599-1630: logits = hidden . E^T with E the (f16-sourced) embedding matrix */
void bo_logits(void *mp, const float *hidden, float *logits) {
    bo_model *m = (bo_model *)mp;
    int nt = m->n_threads, V = m->cfg.vocab;
    if (nt > 64) nt = 64;
    if (nt < 1) nt = 1;
    struct logit_arg a = {m, hidden, logits, (V + nt - 1) / nt};
    bo_parallel_for(nt, nt, logit_task, &a);
}

/* One decode step (T:1482-1504 body): embed -> blocks -> final norm [-> logits].
 * hidden_out (nullable): final-normed hidden [hidden]; logits_out (nullable): [vocab].
 * trace (nullable): per-layer residual stream after each block, [n_layers, hidden]. */
int bo_model_step(void *mp, void *kvp, int token, float *hidden_out, float *logits_out, float *trace) {
    bo_model *m = (bo_model *)mp;
    bo_kv *kv = (bo_kv *)kvp;
    const bo_model_cfg *c = &m->cfg;
    int H = c->hidden, D = c->head_dim, NH = c->n_heads, NKV = c->n_kv_heads, F = c->ffn;
    int pos = kv->seq_len;
    if (pos >= c->max_pos) return 2; /* "KV cache overflow" T:1190-1194 */
    if (token < 0 || token >= c->vocab) return 3;
    float *x = (float *)malloc(sizeof(float) * (size_t)H);
    float *xn = (float *)malloc(sizeof(float) * (size_t)H);
    float *q = (float *)malloc(sizeof(float) * (size_t)NH * D);
    float *kx = (float *)malloc(sizeof(float) * (size_t)NKV * D);
    float *vx = (float *)malloc(sizeof(float) * (size_t)NKV * D);
    float *att = (float *)malloc(sizeof(float) * (size_t)NH * D);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)H);
    float *gate = (float *)malloc(sizeof(float) * (size_t)F);
    float *up = (float *)malloc(sizeof(float) * (size_t)F);
    /* embed: row gather (T:1415-1424), f16 -> f32 */
    for (int i = 0; i < H; ++i) x[i] = bo_f16_to_f32(m->embed_f16[(size_t)token * H + i]);
    const float *sin_row = m->rope_sin + (size_t)pos * (D / 2);
    const float *cos_row = m->rope_cos + (size_t)pos * (D / 2);
    size_t layer_stride = (size_t)NKV * c->max_pos * D;
    int rc = 0;
    for (int l = 0; l < c->n_layers && !rc; ++l) {
        const bo_layer *L = &m->layers[l];
        bo_layernorm(x, L->attn_norm, c->eps, H, xn);
        rc |= proj(m, L, 0, L->q, xn, q, NH * D, H);
        rc |= proj(m, L, 1, L->k, xn, kx, NKV * D, H);
        rc |= proj(m, L, 2, L->v, xn, vx, NKV * D, H);
        for (int h = 0; h < NH; ++h) bo_rope_apply(q + (size_t)h * D, D, sin_row, cos_row);
        for (int h = 0; h < NKV; ++h) bo_rope_apply(kx + (size_t)h * D, D, sin_row, cos_row);
        float *kc = kv->k + (size_t)l * layer_stride, *vc = kv->v + (size_t)l * layer_stride;
        for (int h = 0; h < NKV; ++h) {
            memcpy(kc + ((size_t)h * c->max_pos + pos) * D, kx + (size_t)h * D, sizeof(float) * (size_t)D);
            memcpy(vc + ((size_t)h * c->max_pos + pos) * D, vx + (size_t)h * D, sizeof(float) * (size_t)D);
        }
        bo_attention_decode_mt(q, kc, vc, NH, NKV, D, c->max_pos, pos + 1, att, m->n_threads);
        rc |= proj(m, L, 3, L->o, att, tmp, H, NH * D);
        for (int i = 0; i < H; ++i) x[i] = tmp[i] + x[i]; /* x + residual T:1073 */
        bo_layernorm(x, L->ffn_norm, c->eps, H, xn);
        rc |= proj(m, L, 4, L->gate, xn, gate, F, H);
        rc |= proj(m, L, 5, L->up, xn, up, F, H);
        for (int i = 0; i < F; ++i) gate[i] = bo_silu(gate[i]) * up[i];
        rc |= proj(m, L, 6, L->down, gate, tmp, H, F);
        for (int i = 0; i < H; ++i) x[i] = tmp[i] + x[i];
        if (trace) memcpy(trace + (size_t)l * H, x, sizeof(float) * (size_t)H);
    }
    bo_layernorm(x, m->final_norm, c->eps, H, xn);
    if (hidden_out) memcpy(hidden_out, xn, sizeof(float) * (size_t)H);
    if (logits_out) bo_logits(m, xn, logits_out);
    kv->seq_len = pos + 1;
    free(x);
    free(xn);
    free(q);
    free(kx);
    free(vx);
    free(att);
    free(tmp);
    free(gate);
    free(up);
    return rc;
}
