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
static void *attn_worker(void *p) {
    struct attn_arg *a = (struct attn_arg *)p;
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
    return NULL;
}
void bo_attention_decode_mt(const float *q, const float *kc, const float *vc, int n_heads, int n_kv_heads, int dim,
                            int max_pos, int t_k, float *out, int n_threads) {
    if (n_threads < 2 || t_k < 128) {
        bo_attention_decode(q, kc, vc, n_heads, n_kv_heads, dim, max_pos, t_k, out);
        return;
    }
    pthread_t th[64];
    struct attn_arg args[64];
    int nt = n_threads > 64 ? 64 : n_threads;
    if (nt > n_heads) nt = n_heads;
    int per = (n_heads + nt - 1) / nt;
    int used = 0;
    for (int t = 0; t < nt; ++t) {
        int h0 = t * per, h1 = h0 + per > n_heads ? n_heads : h0 + per;
        if (h0 >= h1) break;
        args[t] = (struct attn_arg){q, kc, vc, n_heads, n_kv_heads, dim, max_pos, t_k, h0, h1, out};
        pthread_create(&th[t], NULL, attn_worker, &args[t]);
        ++used;
    }
    for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
}

/* dense f32 projection, rows dealt to threads: each row's dot product is the loop of proj() below, bit for bit */
struct dense_arg {
    const float *W, *x;
    float *y;
    int r0, r1, cols;
};
static void *dense_worker(void *p) {
    struct dense_arg *a = (struct dense_arg *)p;
    for (int r = a->r0; r < a->r1; ++r) {
        float acc = 0.0f;
        for (int c = 0; c < a->cols; ++c) acc += a->x[c] * a->W[(size_t)r * a->cols + c];
        a->y[r] = acc;
    }
    return NULL;
}
static void dense_mt(const float *W, const float *x, float *y, int rows, int cols, int n_threads) {
    pthread_t th[64];
    struct dense_arg args[64];
    int nt = n_threads > 64 ? 64 : n_threads;
    int per = (rows + nt - 1) / nt, used = 0;
    for (int t = 0; t < nt; ++t) {
        int r0 = t * per, r1 = r0 + per > rows ? rows : r0 + per;
        if (r0 >= r1) break;
        args[t] = (struct dense_arg){W, x, y, r0, r1, cols};
        pthread_create(&th[t], NULL, dense_worker, &args[t]);
        ++used;
    }
    for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
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
    if (L->dense[which]) { /* x . W^T, one f32 dot product per output */
        const float *W = L->dense[which];
        if (m->n_threads > 1 && rows >= 256) {
            dense_mt(W, x, y, rows, cols, m->n_threads);
            return 0;
        }
        for (int r = 0; r < rows; ++r) {
            float acc = 0.0f;
            for (int c = 0; c < cols; ++c) acc += x[c] * W[(size_t)r * cols + c];
            y[r] = acc;
        }
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
    int v0, v1;
};
static void *logit_worker(void *p) {
    struct logit_arg *a = (struct logit_arg *)p;
    int H = a->m->cfg.hidden;
    for (int v = a->v0; v < a->v1; ++v) {
        const uint16_t *row = a->m->embed_f16 + (size_t)v * H;
        float acc = 0.0f;
        for (int k = 0; k < H; ++k) acc += a->h[k] * bo_f16_to_f32(row[k]);
        a->out[v] = acc;
    }
    return NULL;
}

/* T:1599-1630: logits = hidden . E^T with E the (f16-sourced) embedding matrix */
void bo_logits(void *mp, const float *hidden, float *logits) {
    bo_model *m = (bo_model *)mp;
    int nt = m->n_threads, V = m->cfg.vocab;
    pthread_t th[64];
    struct logit_arg args[64];
    if (nt > 64) nt = 64;
    int per = (V + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        int v0 = t * per, v1 = v0 + per > V ? V : v0 + per;
        if (v0 > V) v0 = V;
        args[t] = (struct logit_arg){m, hidden, logits, v0, v1};
        pthread_create(&th[t], NULL, logit_worker, &args[t]);
    }
    for (int t = 0; t < nt; ++t) pthread_join(th[t], NULL);
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
