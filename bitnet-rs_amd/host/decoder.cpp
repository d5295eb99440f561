// decoder.cpp -- see decoder.hpp.  Host orchestration only; every device operation is a
// bitnet_hip_* call (include/bitnet_hip.h) or plain HIP memory / stream / graph plumbing.
#include "decoder.hpp"

#include "blake3.hpp"

#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>

namespace bitnet_host {

namespace {
struct Event {  // RAII: an error return between create and destroy must not leak the event
    hipEvent_t e = nullptr;
    ~Event() {
        if (e) hipEventDestroy(e);
    }
};
}  // namespace

#define HCHK(expr)                                                                     \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            char _b[256];                                                              \
            snprintf(_b, sizeof(_b), "HIP error %s at %s:%d (%s)", hipGetErrorName(_e), __FILE__, __LINE__, #expr); \
            err_ = _b;                                                                 \
            return BITNET_HIP_ERR_GPU;                                                 \
        }                                                                              \
    } while (0)
#define BCHK(expr)                       \
    do {                                 \
        int _rc = (expr);                \
        if (_rc != 0) return fail(#expr); \
    } while (0)

int Decoder::fail_arg(const char *what) {
    err_ = what;
    return BITNET_HIP_ERR_INVALID_ARGUMENT;
}

int Decoder::fail(const char *what) {
    const char *e = bitnet_hip_get_last_error();
    err_ = std::string(what) + ": " + (e ? e : "error");
    return BITNET_HIP_ERR_EXECUTION;
}

template <class T>
static hipError_t dalloc(T **p, size_t n) {
    return hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T) ? n * sizeof(T) : 1);
}

static std::string config_problem(const Config &c) {
    char b[160];
    if (c.hidden <= 0 || c.n_layers <= 0 || c.n_heads <= 0 || c.n_kv_heads <= 0 || c.head_dim <= 0 || c.ffn <= 0 || c.vocab <= 0 || c.max_pos <= 1) {
        snprintf(b, sizeof(b), "config: every dimension must be positive (hidden %d, layers %d, heads %d/%d, head_dim %d, ffn %d, vocab %d, max_pos %d)",
                 c.hidden, c.n_layers, c.n_heads, c.n_kv_heads, c.head_dim, c.ffn, c.vocab, c.max_pos);
        return b;
    }
    if (c.n_heads % c.n_kv_heads != 0) return "config: num_heads must be divisible by num_key_value_heads";  // T:215-220
    if (c.head_dim != 128) return "config: head_dim must be 128 (attention kernels)";
    if (c.n_heads / c.n_kv_heads > 4) return "config: at most 4 query heads per KV head (attention kernels)";
    if (c.hidden % 512 != 0 || c.hidden > 8192) return "config: hidden must be a multiple of 512, <= 8192 (logits kernel)";
    if (c.n_layers > 4096 || c.max_pos > (1 << 20)) return "config: n_layers / max_pos out of range";
    return "";
}

Decoder::Decoder(const Config &cfg) : c_(cfg), layers_(cfg.n_layers > 0 && cfg.n_layers <= 4096 ? (size_t)cfg.n_layers : 0) {
    // A rejected configuration or a failed allocation leaves a DEAD object: error() says why, every other entry point
    // returns an error without touching a member (c_ used to keep the rejected n_layers while layers_ was empty).
    dead_ = true;
    err_ = config_problem(cfg);
    if (!err_.empty()) {
        layers_.clear();
        return;
    }
    if (const char *e = getenv("BITNET_HOST_LOGITS_WGS")) logits_wgs_ = atoi(e) > 0 ? atoi(e) : logits_wgs_;  // tuning knob
    if (const char *e = getenv("BITNET_TRACE_DIR")) trace_dir_ = e;  // the reference's switch (crates/bitnet-trace/src/lib.rs:113-117)
    if (const char *e = getenv("BITNET_HOST_KV16")) kv_f16_ = atoi(e) != 0;  // opt-in f16 KV cache (long contexts)
    if (const char *e = getenv("BITNET_HOST_ACT")) act_mode_ = atoi(e) != 0 ? 1 : 0;  // 0: exact f32 activations between the kernels
    if (const char *e = getenv("BITNET_HOST_PREFILL_CHAIN")) prefill_chain_ = atoi(e) != 0 ? 1 : 0;  // the prompt forward's f16 activation chain
    if (bitnet_hip_init(-1) != 0) {
        const char *e = bitnet_hip_get_last_error();
        err_ = e ? e : "bitnet_hip_init failed";
        return;
    }
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
        err_ = "hipStreamCreate failed";
        return;
    }
    stream_ = s;
    const size_t H = c_.hidden, D = c_.head_dim, half = D / 2;
    {
        // bitnet_hip_gemv_attn_merge_dev's shape limits (include/bitnet_hip.h)
        const int group = c_.n_kv_heads > 0 ? c_.n_heads / c_.n_kv_heads : 0;
        merge_ok_ = c_.head_dim == 128 && (group == 1 || group == 2 || group == 4) && c_.n_heads % c_.n_kv_heads == 0 &&
                    (size_t)c_.n_heads * 128 <= 4096;
    }
    bool ok = true;
    ok &= dalloc(&x_, H) == hipSuccess && dalloc(&x2_, H) == hipSuccess;
    ok &= dalloc(&qkv_, (size_t)(c_.n_heads + 2 * c_.n_kv_heads) * D) == hipSuccess;
    ok &= dalloc(&att_, (size_t)c_.n_heads * D) == hipSuccess;
    ok &= dalloc(&h_, (size_t)c_.ffn) == hipSuccess;
    {
        // QAct records + LayerNorm statistics pairs of the four vectors that travel between the step's GEMVs (zero-filled
        // once: bytes past a vector's last group are never written and must read as zero digits)
        const size_t qh = bitnet_hip_qact_bytes(H), qq = bitnet_hip_qact_bytes((size_t)c_.n_heads * D), qf = bitnet_hip_qact_bytes((size_t)c_.ffn);
        const size_t sb = bitnet_hip_qact_stats_bytes(H);
        ok = ok && hipMalloc(&qa_x_, qh) == hipSuccess && hipMalloc(&qa_x2_, qh) == hipSuccess && hipMalloc(&qa_att_, qq) == hipSuccess &&
             hipMalloc(&qa_h_, qf) == hipSuccess && hipMalloc((void **)&st_x_, sb) == hipSuccess && hipMalloc((void **)&st_x2_, sb) == hipSuccess;
        ok = ok && hipMemset(qa_x_, 0, qh) == hipSuccess && hipMemset(qa_x2_, 0, qh) == hipSuccess && hipMemset(qa_att_, 0, qq) == hipSuccess &&
             hipMemset(qa_h_, 0, qf) == hipSuccess && hipMemset(st_x_, 0, sb) == hipSuccess && hipMemset(st_x2_, 0, sb) == hipSuccess;
    }
    ok &= dalloc(&ref_n_, H > (size_t)c_.ffn ? H : (size_t)c_.ffn) == hipSuccess && dalloc(&ref_gu_, 2 * (size_t)c_.ffn) == hipSuccess &&
          dalloc(&ref_t_, H) == hipSuccess;
    ok &= dalloc(&logits_, (size_t)c_.vocab) == hipSuccess;
    ok &= hipMalloc(&scratch_, 8 * (size_t)logits_wgs_) == hipSuccess;
    {
        const size_t sb = bitnet_hip_attention_scratch_bytes((size_t)c_.n_kv_heads, (size_t)c_.max_pos);
        // zero-filled: the merging o-projection reads records of chunks past the context (and gives them zero weight)
        ok = ok && hipMalloc((void **)&attn_scratch_, sb) == hipSuccess && hipMemset(attn_scratch_, 0, sb) == hipSuccess;
    }
    ok &= dalloc(&pos_, 1) == hipSuccess && dalloc(&n_forced_, 1) == hipSuccess && dalloc(&token_, 1) == hipSuccess;
    ok &= dalloc(&history_, (size_t)c_.max_pos + 2) == hipSuccess;
    ok &= dalloc(&rope_sin_, (size_t)c_.max_pos * half) == hipSuccess;
    ok &= dalloc(&rope_cos_, (size_t)c_.max_pos * half) == hipSuccess;
    for (auto &L : layers_) {
        const size_t n = (size_t)c_.n_kv_heads * (((size_t)c_.max_pos + 63) / 64 * 64) * D;  // whole 64-position tiles
        ok &= dalloc(&L.kcache, n) == hipSuccess && dalloc(&L.vcache, n) == hipSuccess;
        // the decode attention reads whole 64-position tiles and multiplies slots past the context by exact zeros:
        // they must hold finite bit patterns (include/bitnet_hip.h), so the caches start zero-filled
        ok = ok && hipMemset(L.kcache, 0, n * sizeof(float)) == hipSuccess && hipMemset(L.vcache, 0, n * sizeof(float)) == hipSuccess;
        ok &= dalloc(&L.attn_norm, H) == hipSuccess && dalloc(&L.ffn_norm, H) == hipSuccess;
    }
    ok &= dalloc(&final_norm_, H) == hipSuccess;
    if (!ok) {
        err_ = "device allocation failed";
        return;
    }
    // RoPE tables exactly as build_tables does (crates/bitnet-rope/src/lib.rs:59-93)
    std::vector<float> sn((size_t)c_.max_pos * half), cs((size_t)c_.max_pos * half), inv(half);
    for (size_t i = 0; i < half; ++i) inv[i] = 1.0f / powf(c_.rope_theta, (2.0f * (float)i) / (float)D);
    for (int p = 0; p < c_.max_pos; ++p)
        for (size_t i = 0; i < half; ++i) {
            const float angle = (float)p * inv[i];
            sn[(size_t)p * half + i] = sinf(angle);
            cs[(size_t)p * half + i] = cosf(angle);
        }
    if (hipMemcpy(rope_sin_, sn.data(), sn.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(rope_cos_, cs.data(), cs.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        err_ = "rope table upload failed";
    if (!err_.empty()) return;
    dead_ = false;
    reset();
}

Decoder::~Decoder() {
    drop_graphs();
    for (auto &L : layers_) {
        for (bitnet_hip_weights_t h : {L.qkv, L.o, L.gateup, L.down})
            if (h) bitnet_hip_weights_free(h);
        for (float *p : {L.attn_norm, L.ffn_norm, L.kcache, L.vcache})
            if (p) hipFree(p);
    }
    for (void *p : {(void *)embed_, (void *)final_norm_, (void *)rope_sin_, (void *)rope_cos_, (void *)x_, (void *)x2_,
                    (void *)qkv_, (void *)att_, (void *)h_, (void *)ref_n_, (void *)ref_gu_, (void *)ref_t_, qa_x_, qa_x2_, qa_att_, qa_h_, (void *)st_x_, (void *)st_x2_, (void *)logits_, scratch_, (void *)attn_scratch_, (void *)pos_, (void *)n_forced_,
                    (void *)history_, (void *)token_})
        if (p) hipFree(p);
    for (void *p : {(void *)pf_x_, (void *)pf_qkv_, (void *)pf_att_, (void *)pf_h_, pf_gemm_ws_, pf_attn_ws_, sp_kv_send_, sp_kv_all_,
                    (void *)sp_block_pos_, (void *)sp_tokens_, pf_xh_, pf_atth_, pf_hh_, (void *)pf_stats_, pf_qb_})
        if (p) hipFree(p);
    for (void *e : sp_tev_)
        if (e) hipEventDestroy((hipEvent_t)e);
    if (sp_ev_pack_) hipEventDestroy((hipEvent_t)sp_ev_pack_);
    if (sp_ev_gather_) hipEventDestroy((hipEvent_t)sp_ev_gather_);
    if (comm_stream_) hipStreamDestroy((hipStream_t)comm_stream_);
    if (stream_) hipStreamDestroy((hipStream_t)stream_);
}

// A layer that already holds weights gives them back first, and every captured step graph goes (the graphs
// hold the OLD matrices' device pointers).
void Decoder::release_layer(Layer &L) {
    for (bitnet_hip_weights_t *h : {&L.qkv, &L.o, &L.gateup, &L.down}) {
        if (!*h) continue;
        size_t ab = 0;
        if (bitnet_hip_weights_info(*h, nullptr, nullptr, &ab) == 0) weight_bytes_ -= ab < weight_bytes_ ? ab : weight_bytes_;
        bitnet_hip_weights_free(*h);
        *h = 0;
    }
    drop_graphs();
}

void Decoder::drop_graphs() {
    for (int i = 0; i < kGraphs; ++i) {
        if (graph_exec_[i]) hipGraphExecDestroy((hipGraphExec_t)graph_exec_[i]);
        if (graph_[i]) hipGraphDestroy((hipGraph_t)graph_[i]);
        graph_exec_[i] = graph_[i] = nullptr;
    }
}

// the four fused handles of a layer from its seven uploaded projections; frees the seven on every path
int Decoder::adopt_projections(Layer &L, bitnet_hip_weights_t h[7]) {
    const bitnet_hip_weights_t qkv[3] = {h[0], h[1], h[2]}, gu[2] = {h[4], h[5]};
    int rc = bitnet_hip_weights_concat(qkv, 3, 0, &L.qkv);
    if (!rc) rc = bitnet_hip_weights_concat(gu, 2, 1, &L.gateup);
    if (!rc) {
        L.o = h[3];
        L.down = h[6];
        h[3] = h[6] = 0;
    }
    const std::string keep = rc ? (bitnet_hip_get_last_error() ? bitnet_hip_get_last_error() : "error") : "";
    for (int i = 0; i < 7; ++i)
        if (h[i]) bitnet_hip_weights_free(h[i]), h[i] = 0;
    if (rc) {
        release_layer(L);
        err_ = "weights_concat: " + keep;
        return BITNET_HIP_ERR_EXECUTION;
    }
    for (bitnet_hip_weights_t hh : {L.qkv, L.o, L.gateup, L.down}) {
        size_t ab = 0;
        bitnet_hip_weights_info(hh, nullptr, nullptr, &ab);
        weight_bytes_ += ab;
    }
    // LayerNorm applied after the product for the two normalised projections (bitnet_hip_weights_bind_ln)
    BCHK(bitnet_hip_weights_bind_ln(L.qkv, L.attn_norm, stream_));
    BCHK(bitnet_hip_weights_bind_ln(L.gateup, L.ffn_norm, stream_));
    L.q_ok = bitnet_hip_gemv_q_supported(L.qkv) && bitnet_hip_gemv_q_supported(L.o) && bitnet_hip_gemv_q_supported(L.gateup) &&
             bitnet_hip_gemv_q_supported(L.down) && c_.hidden <= 4096;
    return 0;
}

// The step runs on producer-quantised activations (include/bitnet_hip.h "QAct") when every layer's matrices are on that
// path and the mode asks for it; otherwise on exact f32 activations (round 1's kernels).
bool Decoder::qact_path() const {
    if (act_mode_ == 0) return false;
    for (const auto &L : layers_)
        if (!L.q_ok) return false;
    return true;
}

int Decoder::set_act_mode(int mode) {
    if (mode != 0 && mode != 1) {
        err_ = "activation mode must be 0 (exact f32) or 1 (QAct)";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (mode != act_mode_) drop_graphs();
    act_mode_ = mode;
    return 0;
}

int Decoder::set_layer_qk256(int layer, const LayerWeightsQk256 &w) {
    if (layer < 0 || (size_t)layer >= layers_.size()) return fail_arg("layer index out of range");
    Layer &L = layers_[(size_t)layer];
    release_layer(L);
    const size_t H = c_.hidden, QD = (size_t)c_.n_heads * c_.head_dim, KD = (size_t)c_.n_kv_heads * c_.head_dim, F = c_.ffn;
    auto stride = [](size_t cols) { return (cols + 255) / 256 * 64; };
    HCHK(hipMemcpy(L.attn_norm, w.attn_norm, H * 4, hipMemcpyHostToDevice));
    HCHK(hipMemcpy(L.ffn_norm, w.ffn_norm, H * 4, hipMemcpyHostToDevice));
    bitnet_hip_weights_t h[7] = {0};
    const uint8_t *src[7] = {w.q, w.k, w.v, w.o, w.gate, w.up, w.down};
    const size_t n[7] = {QD, KD, KD, H, F, F, H}, k[7] = {H, H, H, QD, H, H, F};
    for (int i = 0; i < 7; ++i) {
        if (bitnet_hip_weights_upload_qk256(src[i], n[i] * stride(k[i]), n[i], k[i], stride(k[i]), &h[i]) != 0) {
            const int rc = fail("bitnet_hip_weights_upload_qk256");
            for (int j = 0; j < i; ++j) bitnet_hip_weights_free(h[j]);
            return rc;
        }
    }
    return adopt_projections(L, h);
}

int Decoder::set_layer_i2s(int layer, const LayerWeightsI2s &w) {
    if (layer < 0 || (size_t)layer >= layers_.size()) return fail_arg("layer index out of range");
    Layer &L = layers_[(size_t)layer];
    release_layer(L);
    const size_t H = c_.hidden, QD = (size_t)c_.n_heads * c_.head_dim, KD = (size_t)c_.n_kv_heads * c_.head_dim, F = c_.ffn;
    const size_t n[7] = {QD, KD, KD, H, F, F, H}, k[7] = {H, H, H, QD, H, H, F};
    HCHK(hipMemcpy(L.attn_norm, w.attn_norm, H * 4, hipMemcpyHostToDevice));
    HCHK(hipMemcpy(L.ffn_norm, w.ffn_norm, H * 4, hipMemcpyHostToDevice));
    bitnet_hip_weights_t h[7] = {0};
    for (int i = 0; i < 7; ++i) {
        const size_t pk = (k[i] + 3) / 4, nb = (k[i] + w.block_size - 1) / w.block_size;
        if (bitnet_hip_weights_upload_i2s(w.w[i], pk * n[i], w.scales[i], nb * n[i], n[i], k[i], w.block_size, &h[i]) != 0) {
            const int rc = fail("bitnet_hip_weights_upload_i2s");
            for (int j = 0; j < i; ++j) bitnet_hip_weights_free(h[j]);
            return rc;
        }
    }
    return adopt_projections(L, h);
}

int Decoder::set_layer_specs(int layer, const float *attn_norm, const float *ffn_norm, const ProjSpec p[7]) {
    if (layer < 0 || (size_t)layer >= layers_.size()) return fail_arg("layer index out of range");
    Layer &L = layers_[(size_t)layer];
    const size_t H = c_.hidden, QD = (size_t)c_.n_heads * c_.head_dim, KD = (size_t)c_.n_kv_heads * c_.head_dim, F = c_.ffn;
    const size_t n[7] = {QD, KD, KD, H, F, F, H}, k[7] = {H, H, H, QD, H, H, F};
    if (p[0].qk256 != p[1].qk256 || p[0].qk256 != p[2].qk256 || p[4].qk256 != p[5].qk256) {
        err_ = "q|k|v (and gate|up) must share one I2_S flavour to be fused";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    release_layer(L);
    HCHK(hipMemcpy(L.attn_norm, attn_norm, H * 4, hipMemcpyHostToDevice));
    HCHK(hipMemcpy(L.ffn_norm, ffn_norm, H * 4, hipMemcpyHostToDevice));
    bitnet_hip_weights_t h[7] = {0};
    for (int i = 0; i < 7; ++i) {
        int rc;
        if (p[i].qk256) {
            const size_t stride = (k[i] + 255) / 256 * 64;
            rc = bitnet_hip_weights_upload_qk256(p[i].bytes, p[i].len, n[i], k[i], stride, &h[i]);
        } else {
            rc = bitnet_hip_weights_upload_coded(p[i].bytes, p[i].len, p[i].scales, p[i].n_scales, n[i], k[i], p[i].block, p[i].code_map, &h[i]);
        }
        if (rc != 0) {
            rc = fail("weights upload");
            for (int j = 0; j < i; ++j) bitnet_hip_weights_free(h[j]);
            return rc;
        }
    }
    return adopt_projections(L, h);
}

int Decoder::set_globals(const uint16_t *embed_f16, const float *final_norm) {
    const size_t n = (size_t)c_.vocab * c_.hidden * 2;
    if (!embed_) HCHK(hipMalloc(&embed_, n));
    HCHK(hipMemcpy(embed_, embed_f16, n, hipMemcpyHostToDevice));
    HCHK(hipMemcpy(final_norm_, final_norm, (size_t)c_.hidden * 4, hipMemcpyHostToDevice));
    return 0;
}

int Decoder::reset() {
    host_forced_ = 0;
    HCHK(hipMemset(pos_, 0, 4));
    HCHK(hipMemset(n_forced_, 0, 4));
    HCHK(hipMemset(history_, 0, ((size_t)c_.max_pos + 2) * 4));
    return 0;
}

int Decoder::feed(const int32_t *tokens, int n) {
    const int p = position();
    if (p < 0) return BITNET_HIP_ERR_GPU;
    const int base = p > host_forced_ ? p : host_forced_;
    if (n < 0 || (n > 0 && !tokens)) {
        err_ = "feed: null tokens";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    for (int i = 0; i < n; ++i)
        if (tokens[i] < 0 || tokens[i] >= c_.vocab) {  // TransformerModel::embed's index_select fails on it (T:1390-1426)
            char b[96];
            snprintf(b, sizeof(b), "token id %d out of range [0, %d)", tokens[i], c_.vocab);
            err_ = b;
            return BITNET_HIP_ERR_INVALID_ARGUMENT;
        }
    if (base + n > c_.max_pos) {
        err_ = "KV cache overflow";  // T:1190-1194
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    HCHK(hipMemcpy(history_ + base, tokens, (size_t)n * 4, hipMemcpyHostToDevice));
    host_forced_ = base + n;
    HCHK(hipMemcpy(n_forced_, &host_forced_, 4, hipMemcpyHostToDevice));
    return 0;
}

int Decoder::position() {
    int p = -1;
    if (hipMemcpy(&p, pos_, 4, hipMemcpyDeviceToHost) != hipSuccess) {
        err_ = "position readback failed";
        return -1;
    }
    return p;
}

// The decode attention in the form and cache type this decoder runs: form 0 two kernels (64-position records), 1 records only
// (the o-projection merges them), 2 two kernels with 128-position records; out / qout: f32 output and / or its QAct.
int Decoder::attn_launch(Layer &L, int form, float *out, void *qout) {
    // forms (form_at): 0 = 64-position records + combine, 1 = 64-position records merged by the o-projection (<= 256 keys),
    //                  2 = 128-position records + combine
    const int flags = (form >= 2 ? BITNET_HIP_ATTN_WIDE : 0) | ((form & 1) ? BITNET_HIP_ATTN_PARTIAL : 0) | (kv_f16_ ? BITNET_HIP_ATTN_KV_F16 : 0);
    BCHK(bitnet_hip_attention_decode_q_dev(qkv_, rope_sin_, rope_cos_, L.kcache, L.vcache, (size_t)c_.n_heads, (size_t)c_.n_kv_heads, (size_t)c_.head_dim,
                                           (size_t)c_.max_pos, pos_, attn_scratch_, flags, out, qout, stream_));
    return 0;
}

int Decoder::set_kv_f16(bool on) {
    if (position() != 0 || host_forced_ != 0) {
        err_ = "set_kv_f16: only on a fresh sequence (reset() first): a cache is f16 or f32 for its whole life";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (on != kv_f16_) {
        drop_graphs();
        for (auto &L : layers_) {  // stale bytes of the other type could be NaN / Inf patterns in this one
            const size_t n = (size_t)c_.n_kv_heads * (((size_t)c_.max_pos + 63) / 64 * 64) * c_.head_dim;
            HCHK(hipMemset(L.kcache, 0, n * sizeof(float)));
            HCHK(hipMemset(L.vcache, 0, n * sizeof(float)));
        }
    }
    kv_f16_ = on;
    return 0;
}

// Reads a device f32 vector back after the kernel that wrote it and leaves one trace record (reference format).
struct Decoder::Tracer {
    std::string dir;
    int seq = 0;
    void *stream = nullptr;
    std::string err;
    bool dump(const std::string &name, const char *stage, int layer, const float *dev, size_t n) {
        std::vector<float> h(n);
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemcpy(h.data(), dev, n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            err = "trace: read-back failed";
            return false;
        }
        double ss = 0.0;
        for (float v : h) ss += (double)v * (double)v;
        const double rms = std::sqrt(ss / (double)n);
        std::string file = name;
        for (char &c : file)
            if (c == '/' || c == '\\') c = '_';
        FILE *f = fopen((dir + "/" + file + ".trace").c_str(), "w");
        if (!f) {
            err = "trace: cannot write into " + dir;
            return false;
        }
        fprintf(f, "{\n  \"name\": \"%s\",\n  \"shape\": [\n    1,\n    %zu\n  ],\n  \"dtype\": \"F32\",\n  \"blake3\": \"%s\",\n  \"rms\": %.17g,\n"
                   "  \"num_elements\": %zu,\n  \"seq\": %d,\n  \"layer\": %d,\n  \"stage\": \"%s\"\n}",
                name.c_str(), n, Blake3::hex(h.data(), n * 4).c_str(), rms, n, seq, layer, stage);
        fclose(f);
        return true;
    }
};
#define TRACE(name, stage, layer, ptr, n)                                   \
    do {                                                                    \
        if (tr && !tr->dump(name, stage, layer, ptr, n)) {                  \
            err_ = tr->err;                                                 \
            return BITNET_HIP_ERR_EXECUTION;                                \
        }                                                                   \
    } while (0)

// One decode step = TransformerModel::forward (T:1557-1597) on one token + logits.
int Decoder::step_launches(bool with_logits, int form, Tracer *tr) {
    void *s = stream_;
    const size_t H = c_.hidden;
    const size_t tQD = (size_t)c_.n_heads * c_.head_dim, tKD = (size_t)c_.n_kv_heads * c_.head_dim;
    const std::string tp = tr ? "t" + std::to_string(tr->seq) + "/" : "";
    if (tr && (form & 1)) form -= 1;  // the merging o-projection never materialises the attention output
    if (qact_path()) {
        // Every vector that travels between two GEMVs goes as a QAct written by its producer's epilogue (x: embedding /
        // down-projection, attention output: combine kernel or merging o-projection, x2: o-projection, h: gate|up).
        const size_t NH = (size_t)c_.n_heads, NK = (size_t)c_.n_kv_heads, MP = (size_t)c_.max_pos;
        BCHK(bitnet_hip_embed_q_dev(embed_, history_, pos_, H, (size_t)c_.vocab, x_, layers_[0].attn_norm, qa_x_, st_x_, s));
        TRACE(tp + "embeddings", "embeddings", -1, x_, H);
        for (size_t l = 0; l < layers_.size(); ++l) {
            Layer &L = layers_[l];
            const std::string bp = tp + "blk" + std::to_string(l) + "/";
            BCHK(bitnet_hip_gemv_q_dev(L.qkv, qa_x_, st_x_, L.attn_norm, c_.eps, nullptr, 0, qkv_, nullptr, nullptr, nullptr, s));
            TRACE(bp + "q_proj", "q_proj", (int)l, qkv_, tQD);
            TRACE(bp + "k_proj", "k_proj", (int)l, qkv_ + tQD, tKD);
            TRACE(bp + "v_proj", "v_proj", (int)l, qkv_ + tQD + tKD, tKD);
            if (form == 1) {
                if (int rc = attn_launch(L, 1, nullptr, nullptr)) return rc;
                BCHK(bitnet_hip_gemv_attn_merge_q_dev(L.o, attn_scratch_, NH, NK, MP, pos_, x2_, x_, qa_x2_, L.ffn_norm, st_x2_, s));
            } else {
                if (int rc = attn_launch(L, form, tr ? att_ : nullptr, qa_att_)) return rc;
                TRACE(bp + "attn_out", "attn_out", (int)l, att_, tQD);
                BCHK(bitnet_hip_gemv_q_dev(L.o, qa_att_, nullptr, nullptr, 0.f, x_, 0, x2_, qa_x2_, L.ffn_norm, st_x2_, s));
            }
            TRACE(bp + "attn_residual", "attn_residual", (int)l, x2_, H);
            BCHK(bitnet_hip_gemv_q_dev(L.gateup, qa_x2_, st_x2_, L.ffn_norm, c_.eps, nullptr, BITNET_HIP_FUSE_SILU_MUL, tr ? h_ : nullptr, qa_h_, nullptr, nullptr, s));
            TRACE(bp + "ffn_hidden", "ffn_hidden", (int)l, h_, (size_t)c_.ffn);
            const bool more = l + 1 < layers_.size();
            BCHK(bitnet_hip_gemv_q_dev(L.down, qa_h_, nullptr, nullptr, 0.f, x2_, 0, x_, more ? qa_x_ : nullptr, more ? layers_[l + 1].attn_norm : nullptr,
                                       more ? st_x_ : nullptr, s));
            TRACE(bp + "ffn_out", "ffn_out", (int)l, x_, H);
        }
        if (tr) TRACE("t" + std::to_string(tr->seq) + "_all_layers_out", "all_layers_out", -1, x_, H);
        if (with_logits) {
            BCHK(bitnet_hip_logits_f16_dev(embed_, x_, final_norm_, c_.eps, H, (size_t)c_.vocab, logits_, scratch_, (size_t)logits_wgs_, token_, pos_,
                                           history_, n_forced_, s));
            TRACE(tp + "logits", "logits", -1, logits_, (size_t)c_.vocab);
        } else {
            BCHK(bitnet_hip_advance_pos_dev(pos_, s));
        }
        return 0;
    }
    BCHK(bitnet_hip_embed_f16_dev(embed_, history_, pos_, 1, H, (size_t)c_.vocab, x_, s));
    TRACE(tp + "embeddings", "embeddings", -1, x_, H);
    for (size_t l = 0; l < layers_.size(); ++l) {
        Layer &L = layers_[l];
        const std::string bp = tp + "blk" + std::to_string(l) + "/";
        // attention_norm -> q,k,v (T:1015, T:288-290), fused into one launch
        BCHK(bitnet_hip_gemv_fused_dev(L.qkv, x_, qkv_, 1, L.attn_norm, c_.eps, nullptr, 0, s));
        TRACE(bp + "q_proj", "q_proj", (int)l, qkv_, tQD);
        TRACE(bp + "k_proj", "k_proj", (int)l, qkv_ + tQD, tKD);
        TRACE(bp + "v_proj", "v_proj", (int)l, qkv_ + tQD + tKD, tKD);
        if (form == 2) {
            if (int rc = attn_launch(L, 2, att_, nullptr)) return rc;
            TRACE(bp + "attn_out", "attn_out", (int)l, att_, tQD);
            BCHK(bitnet_hip_gemv_fused_dev(L.o, att_, x2_, 1, nullptr, 0.f, x_, 0, s));
        } else if (form == 1) {
            // short contexts: one attention launch; the o-projection merges the chunk records itself
            if (int rc = attn_launch(L, 1, nullptr, nullptr)) return rc;
            BCHK(bitnet_hip_gemv_attn_merge_dev(L.o, attn_scratch_, (size_t)c_.n_heads, (size_t)c_.n_kv_heads, (size_t)c_.max_pos, pos_, x2_, x_, s));
        } else {
            if (int rc = attn_launch(L, 0, att_, nullptr)) return rc;
            TRACE(bp + "attn_out", "attn_out", (int)l, att_, tQD);
            // o_proj + residual (T:542, T:1073)
            BCHK(bitnet_hip_gemv_fused_dev(L.o, att_, x2_, 1, nullptr, 0.f, x_, 0, s));
        }
        TRACE(bp + "attn_residual", "attn_residual", (int)l, x2_, H);
        // post_attention_layernorm -> gate, up -> silu(gate)*up (T:1104, T:756-781)
        BCHK(bitnet_hip_gemv_fused_dev(L.gateup, x2_, h_, 1, L.ffn_norm, c_.eps, nullptr, BITNET_HIP_FUSE_SILU_MUL, s));
        TRACE(bp + "ffn_hidden", "ffn_hidden", (int)l, h_, (size_t)c_.ffn);
        // down_proj + residual (T:789, T:1125)
        BCHK(bitnet_hip_gemv_fused_dev(L.down, h_, x_, 1, nullptr, 0.f, x2_, 0, s));
        TRACE(bp + "ffn_out", "ffn_out", (int)l, x_, H);
    }
    if (tr) TRACE("t" + std::to_string(tr->seq) + "_all_layers_out", "all_layers_out", -1, x_, H);
    if (with_logits) {
        // final norm + tied logits + greedy token (T:1589, T:1599-1630, sampling.rs:189-202)
        BCHK(bitnet_hip_logits_f16_dev(embed_, x_, final_norm_, c_.eps, H, (size_t)c_.vocab, logits_, scratch_,
                                       (size_t)logits_wgs_, token_, pos_, history_, n_forced_, s));
        TRACE(tp + "logits", "logits", -1, logits_, (size_t)c_.vocab);
    } else {
        BCHK(bitnet_hip_advance_pos_dev(pos_, s));
    }
    return 0;
}

int Decoder::trace_step(const char *dir, bool with_logits) {
    if (!embed_) {
        err_ = "model globals not set";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (!dir || !*dir) {
        err_ = "trace_step: no directory";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    const int p = position();
    if (p < 0) return BITNET_HIP_ERR_GPU;
    if (p + 1 > c_.max_pos - 1) {
        err_ = "KV cache overflow";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (!with_logits && p + 1 > host_forced_) {
        err_ = "trace_step(with_logits = false) past the fed tokens";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    Tracer tr;
    tr.dir = dir;
    tr.seq = p;  // token position of this step (the reference counts 0 = first position)
    tr.stream = stream_;
    const int rc = step_launches(with_logits, form_at(p), &tr);
    if (rc) return rc;
    HCHK(hipStreamSynchronize((hipStream_t)stream_));
    return 0;
}

// The same step UNFUSED, in the reference's own op order (T:977-1134), every projection on the reference-order kernel
// (BITNET_HIP_KERNEL_EXACT: bit-identical to the scalar CPU loops): LayerNorm rows -> GEMV -> attention -> GEMV ->
// residual add -> LayerNorm rows -> GEMV (gate|up tiles) -> silu*mul -> GEMV -> residual add.  Slow (one thread per
// output row); bench.py and the tests hold the fast step's logits against it at the full model size.
int Decoder::step_launches_reference(bool with_logits) {
    void *s = stream_;
    const size_t H = c_.hidden, F = c_.ffn;
    const int K = BITNET_HIP_KERNEL_EXACT;
    BCHK(bitnet_hip_embed_f16_dev(embed_, history_, pos_, 1, H, (size_t)c_.vocab, x_, s));
    for (auto &L : layers_) {
        BCHK(bitnet_hip_norm_rows_dev(x_, L.attn_norm, ref_n_, 1, H, c_.eps, 0, s));
        BCHK(bitnet_hip_matmul_kernel_dev(L.qkv, ref_n_, qkv_, 1, K, s));
        if (int rc = attn_launch(L, 0, att_, nullptr)) return rc;
        BCHK(bitnet_hip_matmul_kernel_dev(L.o, att_, ref_t_, 1, K, s));
        BCHK(bitnet_hip_add_dev(x_, ref_t_, x2_, H, s));
        BCHK(bitnet_hip_norm_rows_dev(x2_, L.ffn_norm, ref_n_, 1, H, c_.eps, 0, s));
        BCHK(bitnet_hip_matmul_kernel_dev(L.gateup, ref_n_, ref_gu_, 1, K, s));  // alternating 16-row (gate, up) tiles
        BCHK(bitnet_hip_silu_mul_dev(ref_gu_, ref_gu_ + 16, h_, F, 16, s));
        BCHK(bitnet_hip_matmul_kernel_dev(L.down, h_, ref_t_, 1, K, s));
        BCHK(bitnet_hip_add_dev(x2_, ref_t_, x_, H, s));
    }
    if (with_logits) {
        BCHK(bitnet_hip_logits_f16_dev(embed_, x_, final_norm_, c_.eps, H, (size_t)c_.vocab, logits_, scratch_, (size_t)logits_wgs_, token_, pos_,
                                       history_, n_forced_, s));
    } else {
        BCHK(bitnet_hip_advance_pos_dev(pos_, s));
    }
    return 0;
}

int Decoder::run_reference(int n, bool with_logits) {
    if (!embed_) {
        err_ = "model globals not set";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (c_.ffn % 16 != 0) {
        err_ = "run_reference: ffn must be a multiple of 16";
        return BITNET_HIP_ERR_UNSUPPORTED;
    }
    const int p = position();
    if (p < 0) return BITNET_HIP_ERR_GPU;
    if (p + n > c_.max_pos - 1) {
        err_ = "KV cache overflow";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    for (int i = 0; i < n; ++i) {
        const int rc = step_launches_reference(with_logits);
        if (rc) return rc;
    }
    HCHK(hipStreamSynchronize((hipStream_t)stream_));
    // the reference-order kernels rebuilt a row-major copy of every matrix: give them back (one copy of the weights on the device)
    for (auto &L : layers_)
        for (bitnet_hip_weights_t h : {L.qkv, L.o, L.gateup, L.down})
            if (h) bitnet_hip_weights_trim(h);
    return 0;
}

// Which attention form a step at `pos` (pos + 1 keys) takes; the host knows every step's position.
int Decoder::form_at(int pos) const {
    static const bool merge_env = !(getenv("BITNET_HOST_ATTN_MERGE") && atoi(getenv("BITNET_HOST_ATTN_MERGE")) == 0);
    static const bool wide_env = !(getenv("BITNET_HOST_ATTN_WIDE") && atoi(getenv("BITNET_HOST_ATTN_WIDE")) == 0);
    const int keys = pos + 1;
    if (merge_env && merge_ok_ && keys <= (int)bitnet_hip_attention_merge_max_keys()) return 1;
    // (257..512 keys as four 128-position records merged by the o-projection measured SLOWER than 64-position records + combine --
    // c2 1201-1214 vs 1224 tok/s, round 2 -- and was removed in round 3)
    if (wide_env && c_.n_kv_heads * ((keys + 63) / 64) > 256) return 2;  // more 64-position chunks than CUs
    return 0;
}

int Decoder::prepare_graphs(bool with_logits) {
    if (!embed_) {
        err_ = "model globals not set";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    unsigned forms = 0;
    for (int pos = 0; pos + 1 < c_.max_pos; ++pos) forms |= 1u << form_at(pos);
    for (int f = 0; f < kGraphs / 2; ++f)
        if (forms & (1u << f))
            if (int rc = ensure_graph(with_logits, f)) return rc;
    return 0;
}

int Decoder::ensure_graph(bool with_logits, int form) {
    const int gi = 2 * form + (with_logits ? 1 : 0);
    if (graph_exec_[gi]) return 0;
    hipStream_t s = (hipStream_t)stream_;
    hipGraph_t g = nullptr;
    HCHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    const int rc = step_launches(with_logits, form);
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (rc != 0) {
        if (g) hipGraphDestroy(g);
        return rc;
    }
    HCHK(e);
    hipGraphExec_t ex = nullptr;
    HCHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    graph_[gi] = g;
    graph_exec_[gi] = ex;
    return 0;
}

int Decoder::run(int n, bool with_logits, bool use_graph, float *elapsed_ms) {
    if (!embed_) {
        err_ = "model globals not set";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    const int p = position();
    if (p < 0) return BITNET_HIP_ERR_GPU;
    if (p + n > c_.max_pos - 1) {
        err_ = "KV cache overflow";  // T:1190-1194
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (!with_logits && p + n > host_forced_) {
        // a step without logits samples nothing: the NEXT position's token must already be in the history
        err_ = "run(with_logits = false) past the fed tokens: nothing would choose the next token";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (!trace_dir_.empty()) {  // BITNET_TRACE_DIR: every step eagerly, with read-backs (slow by design)
        for (int i = 0; i < n; ++i) {
            const int rc = trace_step(trace_dir_.c_str(), with_logits);
            if (rc) return rc;
        }
        if (elapsed_ms) *elapsed_ms = 0.f;
        return 0;
    }
    hipStream_t s = (hipStream_t)stream_;
    if (use_graph) {
        for (int i = 0; i < n; ++i) {  // every form this run needs (at most three captures)
            int rc = ensure_graph(with_logits, form_at(p + i));
            if (rc) return rc;
        }
    }
    Event ev0, ev1;
    HCHK(hipEventCreate(&ev0.e));
    HCHK(hipEventCreate(&ev1.e));
    hipEvent_t e0 = ev0.e, e1 = ev1.e;
    HCHK(hipEventRecord(e0, s));
    for (int i = 0; i < n; ++i) {
        if (use_graph) {
            HCHK(hipGraphLaunch((hipGraphExec_t)graph_exec_[2 * form_at(p + i) + (with_logits ? 1 : 0)], s));
        } else {
            int rc = step_launches(with_logits, form_at(p + i));
            if (rc) return rc;
        }
    }
    HCHK(hipEventRecord(e1, s));
    HCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    HCHK(hipEventElapsedTime(&ms, e0, e1));
    if (elapsed_ms) *elapsed_ms = ms;
    return 0;
}

// The prompt forward as an f16 ACTIVATION CHAIN (bitnet_hip_matmul_f16_dev): every projection reads f16 rows that the kernel producing
// them wrote -- the o- / down-projection's epilogue leaves x (f32, the residual stream), f16(gamma_next * x) and the LayerNorm statistics
// partials; the attention writes f16 rows; gate|up's epilogue writes f16(silu(gate) * up) -- so no row quantiser / conversion launch sits
// between two matmuls (four per layer otherwise: 69 us of a 925 us BitNet32-F16 layer at 4096 tokens, and the f32 round trips they read).
// Taken at digits = 2 (the f16-activation precision class) when every projection of the model takes the chain; automatic for the
// block-scaled format, whose matmul runs on f16 activations anyway; BITNET_HOST_PREFILL_CHAIN=1 forces it for QK256 (whose int8 digit
// matmul is the faster kernel: the default there stays the digit planes), =0 switches it off.
bool Decoder::chain_applies(int digits) const {
    if (force_scaled_ || digits != 2 || prefill_chain_ == 0 || layers_.empty()) return false;
    bool scaled_all = true;
    for (const auto &L : layers_) {
        for (bitnet_hip_weights_t h : {L.qkv, L.o, L.gateup, L.down})
            if (!h || !bitnet_hip_matmul_f16_supported(h)) return false;
        size_t rows = 0, cols = 0, ab = 0;
        if (bitnet_hip_weights_info(L.qkv, &rows, &cols, &ab) != 0) return false;
        scaled_all = scaled_all && ab > rows * cols / 4;  // algorithmic bytes beyond the 2-bit codes: block scales
    }
    return prefill_chain_ == 1 || scaled_all;
}

// The int8 digit form (QK256's prompt matmul) keeps its row quantisers, but the two inputs no LayerNorm touches travel as f16: the
// attention writes f16 rows, gate|up's epilogue writes f16(silu(gate) * up), and the o- / down-projection's quantisers read them
// (BITNET_HIP_FUSE_X_F16 / _Y_F16): half the bytes on both sides of the two largest hand-overs.  digits = 2 only (an f16 row holds 11
// bits of each element: more than the 2-digit planes take from most elements, fewer than 3 or 4 digits).
bool Decoder::handover16_applies(int digits) const {
    if (force_scaled_ || digits != 2 || prefill_chain_ == 0 || layers_.empty()) return false;
    for (const auto &L : layers_)
        for (bitnet_hip_weights_t h : {L.o, L.gateup, L.down}) {
            size_t rows = 0, cols = 0, ab = 0;
            if (!h || bitnet_hip_weights_info(h, &rows, &cols, &ab) != 0) return false;
            if (ab != rows * (cols / 4) || cols % 256 != 0 || rows % 16 != 0) return false;  // unscaled 2-bit codes, the tiled matmul's shapes
        }
    return c_.ffn % 4 == 0;
}

// q|k|v and gate|up of the digit-plane prompt forward (unscaled matrices, digits = 2) on the block-scaled fp6 x fp4 MFMA reading RESIDENT fp4 images
// (bitnet_hip_weights_fp4_image: built here once per model, 22.6 MB per layer at the 2B-4T widths; the 2-bit tiles stay the decode copy) -- the
// same integers as the int8 planes, bit for bit, with no code expansion in the K loop.  BITNET_HOST_PREFILL_FP6=0 keeps the int8 planes.
int Decoder::fp6_flag(int digits) {
    if (digits != 2 || layers_.empty()) return 0;
    if (prefill_fp6_ < 0) {
        const char *e = getenv("BITNET_HOST_PREFILL_FP6");
        prefill_fp6_ = e ? (atoi(e) != 0 ? 1 : 0) : 1;
        for (const auto &L : layers_) {
            if (!prefill_fp6_) break;
            for (bitnet_hip_weights_t h : {L.qkv, L.gateup})
                if (!h || bitnet_hip_weights_fp4_image(h, 1, stream_) != 0) {  // scaled / other code maps: the form does not apply to this model
                    prefill_fp6_ = 0;
                    break;
                }
        }
        if (!prefill_fp6_)
            for (const auto &L : layers_)
                for (bitnet_hip_weights_t h : {L.qkv, L.gateup})
                    if (h) (void)bitnet_hip_weights_fp4_image(h, 0, stream_);
    }
    return prefill_fp6_ ? BITNET_HIP_FUSE_FP6_DIGITS : 0;
}

// (Decoder::prefill explains; BITNET_HOST_PREFILL_HYBRID=0 keeps the int8 digit planes for all four projections, =2 takes the f16 form at
// any length.)  Only for long shares -- where the hidden-row launches of 64-token tiles come to 400 workgroups or more (from 2497 tokens at
// hidden 2560: 40 token tiles x 10 row blocks): measured, QK256, same box -- 4096 tokens 21.5 -> 20.3 ms, 3072 tokens 17.3 -> 16.4, 8192 tokens 45.3 -> 43.4, but 2048
// tokens 11.8 -> 12.0 and 1024 tokens (also one rank's share of the 8-GPU prompt) 7.2 -> 7.8 ms.
bool Decoder::hybrid_applies(size_t n_rows) const {
    static const int hybrid_env = getenv("BITNET_HOST_PREFILL_HYBRID") ? atoi(getenv("BITNET_HOST_PREFILL_HYBRID")) : 1;
    if (!hybrid_env || layers_.empty()) return false;
    if (hybrid_env != 2 && ((size_t)c_.hidden / 256) * ((n_rows + 63) / 64) < 400) return false;
    for (const auto &L : layers_)
        if (bitnet_hip_matmul_f16_supported(L.o) != 1 || bitnet_hip_matmul_f16_supported(L.down) != 1) return false;
    return true;
}

int Decoder::ensure_chain_buffers(size_t N) {
    const size_t H = c_.hidden, QD = (size_t)c_.n_heads * c_.head_dim, F = c_.ffn;
    const size_t NP = (N + 63) / 64 * 64, nst = H / 64;
    hipStream_t s = (hipStream_t)stream_;
    if ((int)NP > pfc_cap_) {
        for (void *q : {pf_xh_, pf_atth_, pf_hh_, (void *)pf_stats_})
            if (q) hipFree(q);
        pf_xh_ = pf_atth_ = pf_hh_ = nullptr;
        pf_stats_ = nullptr;
        pfc_cap_ = 0;
        HCHK(hipMalloc(&pf_xh_, NP * H * 2));
        HCHK(hipMalloc(&pf_atth_, NP * QD * 2));
        HCHK(hipMalloc(&pf_hh_, NP * F * 2));
        HCHK(hipMalloc((void **)&pf_stats_, nst * NP * 2 * sizeof(float)));
        // rows past the prompt are read by the last token tile (never stored): they must hold finite values
        HCHK(hipMemsetAsync(pf_xh_, 0, NP * H * 2, s));
        HCHK(hipMemsetAsync(pf_atth_, 0, NP * QD * 2, s));
        HCHK(hipMemsetAsync(pf_hh_, 0, NP * F * 2, s));
        HCHK(hipMemsetAsync(pf_stats_, 0, nst * NP * 2 * sizeof(float), s));
        pfc_cap_ = (int)NP;
    }
    return 0;
}

int Decoder::prefill_chain_layers(size_t N) {
    const size_t H = c_.hidden, nst = H / 64;
    hipStream_t s = (hipStream_t)stream_;
    {
        const int rc = ensure_chain_buffers(N);
        if (rc) return rc;
    }
    BCHK(bitnet_hip_rows_to_f16_dev(pf_x_, layers_[0].attn_norm, N, H, pf_xh_, pf_stats_, s));
    size_t n_stats = 1;
    const int aflags = (kv_f16_ ? BITNET_HIP_ATTN_CACHE_F16 : 0) | BITNET_HIP_ATTN_OUT_F16;
    for (size_t l = 0; l < layers_.size(); ++l) {
        auto &L = layers_[l];
        BCHK(bitnet_hip_matmul_f16_dev(L.qkv, pf_xh_, N, pf_stats_, n_stats, L.attn_norm, c_.eps, pf_qkv_, nullptr, 0, nullptr, nullptr, nullptr, s));
        BCHK(bitnet_hip_attention_prefill_flags_dev(pf_qkv_, rope_sin_, rope_cos_, L.kcache, L.vcache, (size_t)c_.n_heads, (size_t)c_.n_kv_heads,
                                                    (size_t)c_.head_dim, (size_t)c_.max_pos, N, pf_attn_ws_, pf_attn_ws_bytes_, pf_atth_, aflags, s));
        BCHK(bitnet_hip_matmul_f16_dev(L.o, pf_atth_, N, nullptr, 0, nullptr, 0.f, pf_x_, pf_x_, 0, pf_xh_, L.ffn_norm, pf_stats_, s));
        n_stats = nst;
        BCHK(bitnet_hip_matmul_f16_dev(L.gateup, pf_xh_, N, pf_stats_, n_stats, L.ffn_norm, c_.eps, nullptr, nullptr, BITNET_HIP_FUSE_SILU_MUL, pf_hh_, nullptr,
                                       nullptr, s));
        const bool last = l + 1 == layers_.size();
        BCHK(bitnet_hip_matmul_f16_dev(L.down, pf_hh_, N, nullptr, 0, nullptr, 0.f, pf_x_, pf_x_, 0, last ? nullptr : pf_xh_,
                                       last ? nullptr : layers_[l + 1].attn_norm, last ? nullptr : pf_stats_, s));
    }
    return 0;
}

// The QK256 prompt forward with NO row quantiser launch (round 5): q|k|v and gate|up run the fp6 x fp4 form on QB32 rows -- block-scaled 15-bit
// fixed point, one exponent per 32 columns of a token -- that the kernel PRODUCING the residual stream wrote: the o- / down-projection (f16 matrix
// cores, 320-row workgroups: the hybrid forward's launches) leaves x (f32), QB32(gamma_next * x) and the LayerNorm statistics partials in its
// epilogue (BITNET_HIP_FUSE_YH_QB32); LayerNorm is applied after the product (as on the f16 chain and in the decode step).  Only where the hybrid
// forward applies (long prompts: the producers then run 64-token tiles) and every q|k|v / gate|up matrix takes the form.  OPT-IN
// (BITNET_HOST_PREFILL_QB32=1): measured same-box it is a tie with the quantiser launches (18.91-18.93 vs 18.97-19.00 ms at 4096 tokens: the 41.5 us of
// quantisers per layer come back as +17 / +6.5 us in gate|up / q|k|v and +9 us per producing launch -- the prompt forward runs at a power limit, and the
// memory-bound quantisers were also the matrix pipes' rest: EXPERIMENTS 8.3), and the default keeps the form that is bit-identical to the int8 planes.
bool Decoder::qb32_applies(int digits, size_t n_rows) {
    if (prefill_qb32_ < 0) {
        const char *e = getenv("BITNET_HOST_PREFILL_QB32");
        prefill_qb32_ = e ? (atoi(e) != 0 ? 1 : 0) : 0;
    }
    if (force_scaled_ || !prefill_qb32_ || digits != 2 || !handover16_applies(digits) || !hybrid_applies(n_rows) || !fp6_flag(digits)) return false;
    for (const auto &L : layers_)
        for (bitnet_hip_weights_t h : {L.qkv, L.gateup})
            if (!h || bitnet_hip_matmul_qb32_supported(h) != 1) return false;
    return true;
}

int Decoder::prefill_qb32_layers(size_t N) {
    const size_t H = c_.hidden, nst = H / 64;
    hipStream_t s = (hipStream_t)stream_;
    {
        const int rc = ensure_chain_buffers(N);
        if (rc) return rc;
    }
    const size_t NP = (N + 63) / 64 * 64;
    if ((int)NP > pf_qb_cap_) {
        if (pf_qb_) (void)hipFree(pf_qb_);
        pf_qb_ = nullptr;
        pf_qb_cap_ = 0;
        HCHK(hipMalloc(&pf_qb_, bitnet_hip_qb32_bytes(NP, H)));
        HCHK(hipMemsetAsync(pf_qb_, 0, bitnet_hip_qb32_bytes(NP, H), s));
        pf_qb_cap_ = (int)NP;
    }
    BCHK(bitnet_hip_rows_to_qb32_dev(pf_x_, layers_[0].attn_norm, N, H, pf_qb_, pf_stats_, s));
    size_t n_stats = 1;
    const int aflags = (kv_f16_ ? BITNET_HIP_ATTN_CACHE_F16 : 0) | BITNET_HIP_ATTN_OUT_F16;
    for (size_t l = 0; l < layers_.size(); ++l) {
        auto &L = layers_[l];
        BCHK(bitnet_hip_matmul_qb32_dev(L.qkv, pf_qb_, N, pf_stats_, n_stats, L.attn_norm, c_.eps, pf_qkv_, nullptr, 0, nullptr, nullptr, nullptr, s));
        BCHK(bitnet_hip_attention_prefill_flags_dev(pf_qkv_, rope_sin_, rope_cos_, L.kcache, L.vcache, (size_t)c_.n_heads, (size_t)c_.n_kv_heads,
                                                    (size_t)c_.head_dim, (size_t)c_.max_pos, N, pf_attn_ws_, pf_attn_ws_bytes_, pf_atth_, aflags, s));
        BCHK(bitnet_hip_matmul_f16_dev(L.o, pf_atth_, N, nullptr, 0, nullptr, 0.f, pf_x_, pf_x_, BITNET_HIP_FUSE_YH_QB32, pf_qb_, L.ffn_norm, pf_stats_, s));
        n_stats = nst;
        BCHK(bitnet_hip_matmul_qb32_dev(L.gateup, pf_qb_, N, pf_stats_, n_stats, L.ffn_norm, c_.eps, nullptr, nullptr, BITNET_HIP_FUSE_SILU_MUL, pf_hh_, nullptr,
                                        nullptr, s));
        const bool last = l + 1 == layers_.size();
        if (last)
            BCHK(bitnet_hip_matmul_f16_dev(L.down, pf_hh_, N, nullptr, 0, nullptr, 0.f, pf_x_, pf_x_, 0, nullptr, nullptr, nullptr, s));
        else
            BCHK(bitnet_hip_matmul_f16_dev(L.down, pf_hh_, N, nullptr, 0, nullptr, 0.f, pf_x_, pf_x_, BITNET_HIP_FUSE_YH_QB32, pf_qb_, layers_[l + 1].attn_norm,
                                           pf_stats_, s));
    }
    return 0;
}

int Decoder::prefill(int n, bool with_logits, int digits, float *elapsed_ms) {
    if (!embed_) {
        err_ = "model globals not set";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    const int p = position();
    if (p < 0) return BITNET_HIP_ERR_GPU;
    if (p != 0) {
        err_ = "prefill needs a fresh sequence (position 0): reset() first";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (n <= 0 || n > host_forced_) {
        err_ = "prefill: feed() the prompt tokens first";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (n > c_.max_pos - 1) {
        err_ = "KV cache overflow";  // T:1190-1194
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    const size_t H = c_.hidden, QD = (size_t)c_.n_heads * c_.head_dim, KD = (size_t)c_.n_kv_heads * c_.head_dim, F = c_.ffn;
    if (n > pf_cap_ || bitnet_hip_attention_prefill_workspace_bytes((size_t)c_.n_heads, (size_t)c_.n_kv_heads, (size_t)n) > pf_attn_ws_bytes_) {
        for (void *q : {(void *)pf_x_, (void *)pf_qkv_, (void *)pf_att_, (void *)pf_h_, pf_gemm_ws_, pf_attn_ws_})
            if (q) hipFree(q);
        pf_x_ = pf_qkv_ = pf_att_ = pf_h_ = nullptr;
        pf_gemm_ws_ = pf_attn_ws_ = nullptr;
        pf_cap_ = 0;
        pf_attn_ws_bytes_ = 0;
        const size_t N = (size_t)n;
        pf_gemm_ws_bytes_ = bitnet_hip_matmul_workspace_bytes(N, H > F ? H : F, 4);
        pf_attn_ws_bytes_ = bitnet_hip_attention_prefill_workspace_bytes((size_t)c_.n_heads, (size_t)c_.n_kv_heads, N);
        HCHK(dalloc(&pf_x_, N * H));
        HCHK(dalloc(&pf_qkv_, N * (QD + 2 * KD)));
        HCHK(dalloc(&pf_att_, N * QD));
        HCHK(dalloc(&pf_h_, N * F));
        HCHK(hipMalloc(&pf_gemm_ws_, pf_gemm_ws_bytes_));
        HCHK(hipMalloc(&pf_attn_ws_, pf_attn_ws_bytes_));
        pf_cap_ = n;
    }
    hipStream_t s = (hipStream_t)stream_;
    Event ev0, ev1;
    HCHK(hipEventCreate(&ev0.e));
    HCHK(hipEventCreate(&ev1.e));
    hipEvent_t e0 = ev0.e, e1 = ev1.e;
    if (!force_scaled_) (void)bitnet_hip_f16_saturations(1);  // a fresh count for this prompt
    HCHK(hipEventRecord(e0, s));
    const size_t N = (size_t)n;
    BCHK(bitnet_hip_embed_f16_dev(embed_, history_, pos_, N, H, (size_t)c_.vocab, pf_x_, s));  // *pos_ == 0
    const bool chain = chain_applies(digits);
    if (chain) {
        const int rc = prefill_chain_layers(N);
        if (rc) return rc;
    }
    const bool qb = !chain && qb32_applies(digits, N);
    if (qb) {
        const int rc = prefill_qb32_layers(N);
        if (rc) return rc;
    }
    last_prefill_path_ = chain ? 1 : qb ? 2 : 0;
    const bool h16 = !chain && !qb && handover16_applies(digits);
    if (h16) {
        const int rc = ensure_chain_buffers(N);
        if (rc) return rc;
    }
    // Hybrid (unscaled matrices): the two projections whose inputs ARE f16 rows (the attention output, silu * up) multiply them on the f16
    // matrix cores as they stand (k_gemm_f16a, 320-row workgroups: one round of the chip at 4096 tokens) -- no quantiser launch, no second
    // rounding of values that were rounded to f16 already; q|k|v and gate|up keep the faster int8 digit planes behind their LayerNorm.
    const bool hybrid = h16 && hybrid_applies(N);
    const int f6 = (chain || qb) ? 0 : fp6_flag(digits);
    static const int f6od_env = getenv("BITNET_HOST_PREFILL_FP6_OD") ? atoi(getenv("BITNET_HOST_PREFILL_FP6_OD")) : 0;  // experiment: o / down on the fp6 form too (non-hybrid)
    const int f6od = f6od_env ? f6 : 0;
    for (auto &L : layers_) {
        if (chain || qb) break;
        BCHK(bitnet_hip_matmul_fused_dev(L.qkv, pf_x_, pf_qkv_, N, L.attn_norm, c_.eps, nullptr, f6, digits, pf_gemm_ws_, pf_gemm_ws_bytes_, s));
        if (h16) {
            BCHK(bitnet_hip_attention_prefill_flags_dev(pf_qkv_, rope_sin_, rope_cos_, L.kcache, L.vcache, (size_t)c_.n_heads, (size_t)c_.n_kv_heads, (size_t)c_.head_dim,
                                                        (size_t)c_.max_pos, N, pf_attn_ws_, pf_attn_ws_bytes_, pf_atth_,
                                                        (kv_f16_ ? BITNET_HIP_ATTN_CACHE_F16 : 0) | BITNET_HIP_ATTN_OUT_F16, s));
            if (hybrid)
                BCHK(bitnet_hip_matmul_f16_dev(L.o, pf_atth_, N, nullptr, 0, nullptr, 0.f, pf_x_, pf_x_, 0, nullptr, nullptr, nullptr, s));
            else
                BCHK(bitnet_hip_matmul_fused_dev(L.o, (const float *)pf_atth_, pf_x_, N, nullptr, 0.f, pf_x_, BITNET_HIP_FUSE_X_F16 | f6od, digits, pf_gemm_ws_, pf_gemm_ws_bytes_, s));
            BCHK(bitnet_hip_matmul_fused_dev(L.gateup, pf_x_, (float *)pf_hh_, N, L.ffn_norm, c_.eps, nullptr, BITNET_HIP_FUSE_SILU_MUL | BITNET_HIP_FUSE_Y_F16 | f6, digits,
                                             pf_gemm_ws_, pf_gemm_ws_bytes_, s));
            if (hybrid)
                BCHK(bitnet_hip_matmul_f16_dev(L.down, pf_hh_, N, nullptr, 0, nullptr, 0.f, pf_x_, pf_x_, 0, nullptr, nullptr, nullptr, s));
            else
                BCHK(bitnet_hip_matmul_fused_dev(L.down, (const float *)pf_hh_, pf_x_, N, nullptr, 0.f, pf_x_, BITNET_HIP_FUSE_X_F16 | f6od, digits, pf_gemm_ws_, pf_gemm_ws_bytes_, s));
            continue;
        }
        if (kv_f16_)
            BCHK(bitnet_hip_attention_prefill_kv16_dev(pf_qkv_, rope_sin_, rope_cos_, L.kcache, L.vcache, (size_t)c_.n_heads, (size_t)c_.n_kv_heads,
                                                       (size_t)c_.head_dim, (size_t)c_.max_pos, N, pf_attn_ws_, pf_attn_ws_bytes_, pf_att_, s));
        else
            BCHK(bitnet_hip_attention_prefill_dev(pf_qkv_, rope_sin_, rope_cos_, L.kcache, L.vcache, (size_t)c_.n_heads, (size_t)c_.n_kv_heads,
                                                  (size_t)c_.head_dim, (size_t)c_.max_pos, N, pf_attn_ws_, pf_attn_ws_bytes_, pf_att_, s));
        BCHK(bitnet_hip_matmul_fused_dev(L.o, pf_att_, pf_x_, N, nullptr, 0.f, pf_x_, 0, digits, pf_gemm_ws_, pf_gemm_ws_bytes_, s));
        BCHK(bitnet_hip_matmul_fused_dev(L.gateup, pf_x_, pf_h_, N, L.ffn_norm, c_.eps, nullptr, BITNET_HIP_FUSE_SILU_MUL | f6, digits, pf_gemm_ws_,
                                         pf_gemm_ws_bytes_, s));
        BCHK(bitnet_hip_matmul_fused_dev(L.down, pf_h_, pf_x_, N, nullptr, 0.f, pf_x_, 0, digits, pf_gemm_ws_, pf_gemm_ws_bytes_, s));
    }
    if (chain || qb || h16) {
        // f16 hand-over rows carry no row scale: a value beyond +-65504 (outlier channels x gamma: none in the synthetic models, possible in a real
        // checkpoint) is clamped AND counted (bitnet_hip_f16_saturations).  A prompt that clamped anything is repeated on the digit planes --
        // 30-bit fixed point behind a per-row power-of-two scale (digits = 4: a row whose outlier is 10^5 times its typical element still keeps 13 bits of
        // those; the 2-digit and row-scaled f16 forms would flush them) -- before its logits are taken (no silent saturation).
        HCHK(hipStreamSynchronize(s));
        if (bitnet_hip_f16_saturations(1) > 0) {
            force_scaled_ = true;
            ++saturation_fallbacks_;
            const int rc = prefill(n, with_logits, 4, elapsed_ms);
            force_scaled_ = false;
            return rc;
        }
    }
    {
        const int rc = finish_prefill(n, pf_x_ + (N - 1) * H, with_logits);
        if (rc) return rc;
    }
    HCHK(hipEventRecord(e1, s));
    HCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    HCHK(hipEventElapsedTime(&ms, e0, e1));
    if (elapsed_ms) *elapsed_ms = ms;
    return 0;
}

int Decoder::prefill_sharded(int n, int rank, int world, bitnet_host_allgather_fn gather, void *gather_ctx, bool with_logits, int digits,
                             bool wire_f16, float *elapsed_ms) {
    if (!embed_) {
        err_ = "model globals not set";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !gather)) {
        err_ = "prefill_sharded: bad rank / world / gather";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (n <= 0 || n % (2 * world * 64) != 0) {
        char b[128];
        snprintf(b, sizeof(b), "prompt length %d must be a positive multiple of %d (2 * world * 64) for world %d", n, 2 * world * 64, world);
        err_ = b;
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    const int p = position();
    if (p < 0) return BITNET_HIP_ERR_GPU;
    if (p != 0) {
        err_ = "prefill needs a fresh sequence (position 0): reset() first";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (n > host_forced_) {
        err_ = "prefill: feed() the prompt tokens first";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    if (n > c_.max_pos - 1) {
        err_ = "KV cache overflow";  // T:1190-1194
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    const size_t H = c_.hidden, QD = (size_t)c_.n_heads * c_.head_dim, KD = (size_t)c_.n_kv_heads * c_.head_dim, F = c_.ffn;
    const size_t NH = (size_t)c_.n_heads, NK = (size_t)c_.n_kv_heads, D = (size_t)c_.head_dim, MP = (size_t)c_.max_pos;
    const int chunk = n / (2 * world), nq = 2 * chunk;
    const size_t N = (size_t)nq, esz = wire_f16 ? 2 : 4;
    if (nq > pf_cap_) {  // the per-row buffers of the unsharded prefill, sized for this rank's rows
        for (void *q : {(void *)pf_x_, (void *)pf_qkv_, (void *)pf_att_, (void *)pf_h_, pf_gemm_ws_})
            if (q) hipFree(q);
        pf_x_ = pf_qkv_ = pf_att_ = pf_h_ = nullptr;
        pf_gemm_ws_ = nullptr;
        pf_cap_ = 0;
        pf_gemm_ws_bytes_ = bitnet_hip_matmul_workspace_bytes(N, H > F ? H : F, 4);
        HCHK(dalloc(&pf_x_, N * H));
        HCHK(dalloc(&pf_qkv_, N * (QD + 2 * KD)));
        HCHK(dalloc(&pf_att_, N * QD));
        HCHK(dalloc(&pf_h_, N * F));
        HCHK(hipMalloc(&pf_gemm_ws_, pf_gemm_ws_bytes_));
        pf_cap_ = nq;
    }
    const size_t awb = bitnet_hip_attention_prefill_sharded_workspace_bytes(NH, NK, N, (size_t)n);
    if (awb > pf_attn_ws_bytes_) {
        if (pf_attn_ws_) hipFree(pf_attn_ws_);
        pf_attn_ws_ = nullptr;
        pf_attn_ws_bytes_ = 0;
        HCHK(hipMalloc(&pf_attn_ws_, awb));
        pf_attn_ws_bytes_ = awb;
    }
    if (nq > sp_cap_ || n > sp_ctx_) {
        for (void *q : {sp_kv_send_, sp_kv_all_, (void *)sp_block_pos_, (void *)sp_tokens_})
            if (q) hipFree(q);
        sp_kv_send_ = sp_kv_all_ = nullptr;
        sp_block_pos_ = sp_tokens_ = nullptr;
        sp_cap_ = sp_ctx_ = 0;
        HCHK(hipMalloc(&sp_kv_send_, N * 2 * KD * 4));
        HCHK(hipMalloc(&sp_kv_all_, (size_t)n * 2 * KD * 4));
        HCHK(dalloc(&sp_block_pos_, N / 64));
        HCHK(dalloc(&sp_tokens_, N));
        sp_cap_ = nq;
        sp_ctx_ = n;
    }
    hipStream_t s = (hipStream_t)stream_;
    if (world > 1 && !comm_stream_) {  // the collective's own stream + the two events that tie it to the compute stream
        // all three or none: a later call must never find the stream set and an event missing
        hipStream_t cs = nullptr;
        hipEvent_t e1 = nullptr, e2 = nullptr;
        hipError_t er = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
        if (er == hipSuccess) er = hipEventCreateWithFlags(&e1, hipEventDisableTiming);
        if (er == hipSuccess) er = hipEventCreateWithFlags(&e2, hipEventDisableTiming);
        if (er != hipSuccess) {
            if (e2) (void)hipEventDestroy(e2);
            if (e1) (void)hipEventDestroy(e1);
            if (cs) (void)hipStreamDestroy(cs);
            HCHK(er);
        }
        comm_stream_ = cs;
        sp_ev_pack_ = e1;
        sp_ev_gather_ = e2;
    }
    {
        // this rank's rows: chunk `rank`, then chunk 2 world - 1 - rank
        std::vector<int32_t> hist((size_t)n), tok(N), bp(N / 64);
        HCHK(hipMemcpy(hist.data(), history_, (size_t)n * 4, hipMemcpyDeviceToHost));
        const int starts[2] = {rank * chunk, (2 * world - 1 - rank) * chunk};
        for (int h = 0; h < 2; ++h)
            for (int i = 0; i < chunk; ++i) {
                tok[(size_t)h * chunk + i] = hist[(size_t)starts[h] + i];
                if (i % 64 == 0) bp[((size_t)h * chunk + i) / 64] = starts[h] + i;
            }
        HCHK(hipMemcpy(sp_tokens_, tok.data(), N * 4, hipMemcpyHostToDevice));
        HCHK(hipMemcpy(sp_block_pos_, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    }
    Event ev0, ev1;
    HCHK(hipEventCreate(&ev0.e));
    HCHK(hipEventCreate(&ev1.e));
    // optional per-phase timing: 8 events per layer (6 on the compute stream, 2 around the collective on its own stream)
    const bool timing = sp_timing_;
    if (timing && sp_tev_.size() < layers_.size() * 8) {
        const size_t want = layers_.size() * 8;
        while (sp_tev_.size() < want) {
            hipEvent_t e = nullptr;
            HCHK(hipEventCreate(&e));
            sp_tev_.push_back(e);
        }
    }
    size_t li = 0;
    auto mark = [&](int idx, hipStream_t st) -> hipError_t { return timing ? hipEventRecord((hipEvent_t)sp_tev_[li * 8 + (size_t)idx], st) : hipSuccess; };
    HCHK(hipEventRecord(ev0.e, s));
    BCHK(bitnet_hip_embed_f16_dev(embed_, sp_tokens_, nullptr, N, H, (size_t)c_.vocab, pf_x_, s));
    const size_t ld = QD + 2 * KD, per_rank = N * 2 * KD * esz;
    const bool h16 = handover16_applies(digits);  // f16 hand-over of the attention output and of silu * up (Decoder::prefill explains)
    if (h16) {
        const int rc = ensure_chain_buffers(N);
        if (rc) return rc;
    }
    float *att_out = h16 ? static_cast<float *>(pf_atth_) : pf_att_;
    const bool hybrid = h16 && hybrid_applies(N);  // o / down on the f16 matrix cores, as in the unsharded prefill (long shares only)
    const int aflags = (kv_f16_ ? BITNET_HIP_ATTN_CACHE_F16 : 0) | (h16 ? BITNET_HIP_ATTN_OUT_F16 : 0);
    for (auto &L : layers_) {
        HCHK(mark(0, s));
        BCHK(bitnet_hip_matmul_fused_dev(L.qkv, pf_x_, pf_qkv_, N, L.attn_norm, c_.eps, nullptr, 0, digits, pf_gemm_ws_, pf_gemm_ws_bytes_, s));
        // the raw (pre-RoPE) k|v rows this rank contributes, compact, f32 or f16
        BCHK(bitnet_hip_pack_cols_dev(pf_qkv_, ld, QD, 2 * KD, N, world > 1 ? sp_kv_send_ : sp_kv_all_, wire_f16 ? 1 : 0, s));
        HCHK(mark(1, s));
        if (world > 1) {
            // the collective runs on its own stream beside the query-side preparation (RoPE + f16 pack of this rank's q rows);
            // the k / v slabs and the attention follow once it has finished
            HCHK(hipEventRecord((hipEvent_t)sp_ev_pack_, s));
            HCHK(hipStreamWaitEvent((hipStream_t)comm_stream_, (hipEvent_t)sp_ev_pack_, 0));
            HCHK(mark(6, (hipStream_t)comm_stream_));
            if (gather(gather_ctx, sp_kv_send_, sp_kv_all_, per_rank, comm_stream_) != 0) {
                err_ = "prefill_sharded: the all-gather callback failed";
                return BITNET_HIP_ERR_EXECUTION;
            }
            HCHK(mark(7, (hipStream_t)comm_stream_));
            HCHK(hipEventRecord((hipEvent_t)sp_ev_gather_, (hipStream_t)comm_stream_));
            BCHK(bitnet_hip_attention_prefill_gathered_phase_dev(pf_qkv_, ld, sp_block_pos_, N, sp_kv_all_, (size_t)n, (size_t)world, wire_f16 ? 1 : 0, rope_sin_,
                                                                 rope_cos_, L.kcache, L.vcache, aflags, NH, NK, D, MP, pf_attn_ws_, pf_attn_ws_bytes_, att_out, 1, s));
            HCHK(mark(2, s));
            HCHK(hipStreamWaitEvent(s, (hipEvent_t)sp_ev_gather_, 0));
            HCHK(mark(3, s));
        } else {
            HCHK(mark(2, s));
            HCHK(mark(3, s));
            HCHK(mark(6, s));
            HCHK(mark(7, s));
        }
        BCHK(bitnet_hip_attention_prefill_gathered_phase_dev(pf_qkv_, ld, sp_block_pos_, N, sp_kv_all_, (size_t)n, (size_t)world, wire_f16 ? 1 : 0, rope_sin_,
                                                             rope_cos_, L.kcache, L.vcache, aflags, NH, NK, D, MP, pf_attn_ws_, pf_attn_ws_bytes_, att_out,
                                                             world > 1 ? 2 : 0, s));
        HCHK(mark(4, s));
        const int xf = h16 ? BITNET_HIP_FUSE_X_F16 : 0, yf = h16 ? BITNET_HIP_FUSE_Y_F16 : 0;
        float *h_buf = h16 ? static_cast<float *>(pf_hh_) : pf_h_;
        if (hybrid)
            BCHK(bitnet_hip_matmul_f16_dev(L.o, pf_atth_, N, nullptr, 0, nullptr, 0.f, pf_x_, pf_x_, 0, nullptr, nullptr, nullptr, s));
        else
            BCHK(bitnet_hip_matmul_fused_dev(L.o, att_out, pf_x_, N, nullptr, 0.f, pf_x_, xf, digits, pf_gemm_ws_, pf_gemm_ws_bytes_, s));
        BCHK(bitnet_hip_matmul_fused_dev(L.gateup, pf_x_, h_buf, N, L.ffn_norm, c_.eps, nullptr, BITNET_HIP_FUSE_SILU_MUL | yf, digits, pf_gemm_ws_,
                                         pf_gemm_ws_bytes_, s));
        if (hybrid)
            BCHK(bitnet_hip_matmul_f16_dev(L.down, pf_hh_, N, nullptr, 0, nullptr, 0.f, pf_x_, pf_x_, 0, nullptr, nullptr, nullptr, s));
        else
            BCHK(bitnet_hip_matmul_fused_dev(L.down, h_buf, pf_x_, N, nullptr, 0.f, pf_x_, xf, digits, pf_gemm_ws_, pf_gemm_ws_bytes_, s));
        HCHK(mark(5, s));
        ++li;
    }
    {
        // The last prompt position sits in chunk 2 world - 1 = rank 0's second chunk: its last local row.  EVERY rank gets that row
        // (one more gather of `hidden` floats per rank through the same callback and stream ties: slot 0 is rank 0's), so any rank can
        // continue decoding -- with logits and the first sampled token -- on the cache it filled.
        const float *last = pf_x_ + (N - 1) * H;
        if (world > 1) {
            HCHK(hipMemcpyAsync(sp_kv_send_, last, H * 4, hipMemcpyDeviceToDevice, s));
            HCHK(hipEventRecord((hipEvent_t)sp_ev_pack_, s));
            HCHK(hipStreamWaitEvent((hipStream_t)comm_stream_, (hipEvent_t)sp_ev_pack_, 0));
            if (gather(gather_ctx, sp_kv_send_, sp_kv_all_, H * 4, comm_stream_) != 0) {
                err_ = "prefill_sharded: the all-gather callback failed (last row)";
                return BITNET_HIP_ERR_EXECUTION;
            }
            HCHK(hipEventRecord((hipEvent_t)sp_ev_gather_, (hipStream_t)comm_stream_));
            HCHK(hipStreamWaitEvent(s, (hipEvent_t)sp_ev_gather_, 0));
            last = static_cast<const float *>(sp_kv_all_);
        }
        const int rc = finish_prefill(n, last, with_logits);
        if (rc) return rc;
    }
    HCHK(hipEventRecord(ev1.e, s));
    HCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    HCHK(hipEventElapsedTime(&ms, ev0.e, ev1.e));
    if (elapsed_ms) *elapsed_ms = ms;
    if (timing) {
        if (comm_stream_) HCHK(hipStreamSynchronize((hipStream_t)comm_stream_));
        std::vector<float> ph[4];
        for (size_t l = 0; l < layers_.size(); ++l) {
            auto dt = [&](int a, int b) {
                float t = 0.f;
                (void)hipEventElapsedTime(&t, (hipEvent_t)sp_tev_[l * 8 + (size_t)a], (hipEvent_t)sp_tev_[l * 8 + (size_t)b]);
                return t * 1e3f;
            };
            ph[0].push_back(dt(0, 1) + dt(4, 5));
            ph[1].push_back(dt(1, 2) + dt(3, 4));
            ph[2].push_back(dt(2, 3));
            ph[3].push_back(dt(6, 7));
        }
        for (int i = 0; i < 4; ++i) {
            std::sort(ph[i].begin(), ph[i].end());
            sp_phase_us_[i] = ph[i].empty() ? 0.f : ph[i][ph[i].size() / 2];
        }
    }
    return 0;
}

int Decoder::finish_prefill(int n, const float *last_row, bool with_logits) {
    // hand over to the single-token state: residual stream of the last position, position counter
    if (n <= 0 || n > c_.max_pos - 1) {
        err_ = "KV cache overflow";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream_;
    const size_t H = c_.hidden;
    if (last_row) HCHK(hipMemcpyAsync(x_, last_row, H * 4, hipMemcpyDeviceToDevice, s));
    const int32_t last = n - 1;
    HCHK(hipMemcpyAsync(pos_, &last, 4, hipMemcpyHostToDevice, s));
    HCHK(hipStreamSynchronize(s));  // `last` is a stack variable
    if (with_logits && last_row) {
        BCHK(bitnet_hip_logits_f16_dev(embed_, x_, final_norm_, c_.eps, H, (size_t)c_.vocab, logits_, scratch_, (size_t)logits_wgs_, token_,
                                       pos_, history_, n_forced_, s));
    } else {
        BCHK(bitnet_hip_advance_pos_dev(pos_, s));
    }
    HCHK(hipStreamSynchronize(s));
    return 0;
}

void Decoder::layer_objects(int layer, uint64_t handles[4], void *ptrs[4]) const {
    if (layer < 0 || (size_t)layer >= layers_.size()) {
        for (int i = 0; i < 4; ++i) handles[i] = 0, ptrs[i] = nullptr;
        return;
    }
    const Layer &L = layers_[(size_t)layer];
    handles[0] = L.qkv, handles[1] = L.o, handles[2] = L.gateup, handles[3] = L.down;
    ptrs[0] = L.attn_norm, ptrs[1] = L.ffn_norm, ptrs[2] = L.kcache, ptrs[3] = L.vcache;
}

void Decoder::global_objects(void *ptrs[7]) const {
    ptrs[0] = embed_, ptrs[1] = final_norm_, ptrs[2] = rope_sin_, ptrs[3] = rope_cos_, ptrs[4] = history_, ptrs[5] = pos_, ptrs[6] = stream_;
}

int Decoder::history(int32_t *out, int n) {
    if (!out || n < 0 || n > c_.max_pos + 2) {
        err_ = "history: n out of range";
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    }
    HCHK(hipMemcpy(out, history_, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}
int Decoder::last_logits(float *out) {
    HCHK(hipMemcpy(out, logits_, (size_t)c_.vocab * 4, hipMemcpyDeviceToHost));
    return 0;
}
int Decoder::last_hidden(float *out) {
    HCHK(hipMemcpy(out, x_, (size_t)c_.hidden * 4, hipMemcpyDeviceToHost));
    return 0;
}

// kind: 0 q|k|v, 1 attention, 2 o_proj, 3 gate|up, 4 down, 5 logits+argmax
int Decoder::probe_kernel(int kind, int reps, float *us_per_launch, double *bytes_per_launch) {
    hipStream_t s = (hipStream_t)stream_;
    const int p0 = position();
    hipGraph_t g = nullptr;
    HCHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    int rc = 0;
    const size_t H = c_.hidden;
    int launches = 0;
    const bool qp = qact_path();
    for (size_t li = 0; li < layers_.size(); ++li) {
        Layer &L = layers_[li];
        if (qp && kind <= 4) {
            const bool more = li + 1 < layers_.size();
            switch (kind) {
                case 0: rc = bitnet_hip_gemv_q_dev(L.qkv, qa_x_, st_x_, L.attn_norm, c_.eps, nullptr, 0, qkv_, nullptr, nullptr, nullptr, stream_); break;
                case 1: rc = attn_launch(L, form_at(p0 > 0 ? p0 : 0) == 2 ? 2 : 0, nullptr, qa_att_); break;
                case 2: rc = bitnet_hip_gemv_q_dev(L.o, qa_att_, nullptr, nullptr, 0.f, x_, 0, x2_, qa_x2_, L.ffn_norm, st_x2_, stream_); break;
                case 3: rc = bitnet_hip_gemv_q_dev(L.gateup, qa_x2_, st_x2_, L.ffn_norm, c_.eps, nullptr, BITNET_HIP_FUSE_SILU_MUL, nullptr, qa_h_, nullptr, nullptr, stream_); break;
                default:
                    // x_ is both the residual stream's next value and (kind 2) its old one in the real step; here it is only written
                    rc = bitnet_hip_gemv_q_dev(L.down, qa_h_, nullptr, nullptr, 0.f, x2_, 0, ref_t_, more ? qa_x_ : nullptr, more ? layers_[li + 1].attn_norm : nullptr,
                                               more ? st_x_ : nullptr, stream_);
                    break;
            }
            ++launches;
            if (rc) break;
            continue;
        }
        switch (kind) {
            case 0: rc = bitnet_hip_gemv_fused_dev(L.qkv, x_, qkv_, 1, L.attn_norm, c_.eps, nullptr, 0, stream_); break;
            case 1: rc = attn_launch(L, form_at(p0 > 0 ? p0 : 0) == 2 ? 2 : 0, att_, nullptr); break;
            case 2: rc = bitnet_hip_gemv_fused_dev(L.o, att_, x2_, 1, nullptr, 0.f, x_, 0, stream_); break;
            case 3: rc = bitnet_hip_gemv_fused_dev(L.gateup, x2_, h_, 1, L.ffn_norm, c_.eps, nullptr, BITNET_HIP_FUSE_SILU_MUL, stream_); break;
            case 4: rc = bitnet_hip_gemv_fused_dev(L.down, h_, x_, 1, nullptr, 0.f, x2_, 0, stream_); break;
            default:
                rc = bitnet_hip_logits_f16_dev(embed_, x_, final_norm_, c_.eps, H, (size_t)c_.vocab, logits_, scratch_,
                                               (size_t)logits_wgs_, token_, nullptr, nullptr, nullptr, stream_);
                break;
        }
        ++launches;
        if (rc || kind > 4) break;
    }
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (rc) {
        if (g) hipGraphDestroy(g);
        return fail("probe_kernel launch");
    }
    HCHK(e);
    hipGraphExec_t ex = nullptr;
    HCHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    HCHK(hipGraphLaunch(ex, s));  // warm
    HCHK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    HCHK(hipEventCreate(&e0));
    HCHK(hipEventCreate(&e1));
    HCHK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) HCHK(hipGraphLaunch(ex, s));
    HCHK(hipEventRecord(e1, s));
    HCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    HCHK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipGraphExecDestroy(ex);
    hipGraphDestroy(g);
    (void)p0;
    if (us_per_launch) *us_per_launch = ms * 1e3f / (float)(reps * launches);
    if (bytes_per_launch) {
        size_t ab = 0;
        const Layer &L = layers_[0];
        const double kv = 2.0 * c_.n_kv_heads * (double)(position() + 1) * c_.head_dim * (kv_f16_ ? 2 : 4);
        if (qp && kind <= 4 && kind != 1) {
            // QAct path: codes + scales + the activation records read (+ statistics pairs) + what the launch writes
            const double qH = (double)bitnet_hip_qact_bytes(H), qF = (double)bitnet_hip_qact_bytes((size_t)c_.ffn), sH = (double)bitnet_hip_qact_stats_bytes(H);
            const bitnet_hip_weights_t hh = kind == 0 ? L.qkv : kind == 2 ? L.o : kind == 3 ? L.gateup : L.down;
            bitnet_hip_weights_info(hh, nullptr, nullptr, &ab);
            const double io = kind == 0 ? qH + sH + 4.0 * H + 4.0 * (c_.n_heads + 2 * c_.n_kv_heads) * c_.head_dim   // + g_r
                            : kind == 2 ? (double)bitnet_hip_qact_bytes((size_t)c_.n_heads * c_.head_dim) + 12.0 * H + qH + sH  // residual in, f32 out, gamma, QAct + stats out
                            : kind == 3 ? qH + sH + 8.0 * c_.ffn + qF                                                 // g_r of both halves, QAct out
                                        : qF + 12.0 * H + qH + sH;
            *bytes_per_launch = (double)ab + io;
            return 0;
        }
        switch (kind) {  // algorithmic bytes of one launch: codes + scales + vector in + vector out
            case 0: bitnet_hip_weights_info(L.qkv, nullptr, nullptr, &ab); *bytes_per_launch = (double)ab + 8.0 * H + 4.0 * (c_.n_heads + 2 * c_.n_kv_heads) * c_.head_dim; break;
            case 1: *bytes_per_launch = kv + 8.0 * c_.n_heads * c_.head_dim; break;
            case 2: bitnet_hip_weights_info(L.o, nullptr, nullptr, &ab); *bytes_per_launch = (double)ab + 4.0 * c_.n_heads * c_.head_dim + 8.0 * H; break;
            case 3: bitnet_hip_weights_info(L.gateup, nullptr, nullptr, &ab); *bytes_per_launch = (double)ab + 8.0 * H + 4.0 * c_.ffn; break;
            case 4: bitnet_hip_weights_info(L.down, nullptr, nullptr, &ab); *bytes_per_launch = (double)ab + 4.0 * c_.ffn + 8.0 * H; break;
            default: *bytes_per_launch = 2.0 * (double)c_.vocab * H + 8.0 * H + 4.0 * c_.vocab; break;
        }
    }
    return 0;
}

int Decoder::probe_gateup(int reps, float *us_per_launch, double *bytes_per_launch) {
    return probe_kernel(3, reps, us_per_launch, bytes_per_launch);
}

}  // namespace bitnet_host

using bitnet_host::Decoder;

namespace {
// a null handle or a dead decoder (rejected configuration, failed allocation): nothing but error() / destroy may be called
inline Decoder *live(void *d) {
    Decoder *p = static_cast<Decoder *>(d);
    return p && !p->dead() ? p : nullptr;
}
}  // namespace
#define LIVE(rc)           \
    Decoder *D = live(d);  \
    if (!D) return rc

extern "C" {
void *bitnet_host_create(const bitnet_host_config *cfg) {
    bitnet_host::Config c;
    c.hidden = cfg->hidden;
    c.n_layers = cfg->n_layers;
    c.n_heads = cfg->n_heads;
    c.n_kv_heads = cfg->n_kv_heads;
    c.head_dim = cfg->head_dim;
    c.ffn = cfg->ffn;
    c.vocab = cfg->vocab;
    c.max_pos = cfg->max_pos;
    c.eps = cfg->eps;
    c.rope_theta = cfg->rope_theta;
    // Nothing is thrown across the C boundary: a configuration the kernels cannot take comes back as a decoder
    // whose error() says why (every later call on it fails), allocation failures as nullptr.
    try {
        return new Decoder(c);
    } catch (...) {
        return nullptr;
    }
}
void bitnet_host_destroy(void *d) { delete static_cast<Decoder *>(d); }
const char *bitnet_host_error(void *d) { return d ? static_cast<Decoder *>(d)->error().c_str() : "null decoder"; }
int bitnet_host_set_layer_qk256(void *d, int layer, const float *attn_norm, const float *ffn_norm, const uint8_t *q,
                                const uint8_t *k, const uint8_t *v, const uint8_t *o, const uint8_t *gate,
                                const uint8_t *up, const uint8_t *down) {
    bitnet_host::LayerWeightsQk256 w{attn_norm, ffn_norm, q, k, v, o, gate, up, down};
    LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT);
    return D->set_layer_qk256(layer, w);
}
int bitnet_host_set_layer_i2s(void *d, int layer, const float *attn_norm, const float *ffn_norm, const uint8_t *const *w7,
                              const float *const *scales7, size_t block_size) {
    bitnet_host::LayerWeightsI2s w;
    w.attn_norm = attn_norm;
    w.ffn_norm = ffn_norm;
    for (int i = 0; i < 7; ++i) {
        w.w[i] = w7[i];
        w.scales[i] = scales7[i];
    }
    w.block_size = block_size;
    LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT);
    return D->set_layer_i2s(layer, w);
}
int bitnet_host_set_globals(void *d, const uint16_t *embed_f16, const float *final_norm) {
    LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT);
    return D->set_globals(embed_f16, final_norm);
}
int bitnet_host_reset(void *d) { LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT); return D->reset(); }
int bitnet_host_feed(void *d, const int32_t *tokens, int n) { LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT); return D->feed(tokens, n); }
int bitnet_host_set_kv_f16(void *d, int on) { LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT); return D->set_kv_f16(on != 0); }
int bitnet_host_set_act_mode(void *d, int mode) { LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT); return D->set_act_mode(mode); }
int bitnet_host_act_mode(void *d) { LIVE(0); return D->qact_path() ? 1 : 0; }
int bitnet_host_prepare_graphs(void *d, int with_logits) { LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT); return D->prepare_graphs(with_logits != 0); }
int bitnet_host_run_reference(void *d, int n, int with_logits) { LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT); return D->run_reference(n, with_logits != 0); }
int bitnet_host_run(void *d, int n, int with_logits, int use_graph, float *elapsed_ms) {
    LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT);
    return D->run(n, with_logits != 0, use_graph != 0, elapsed_ms);
}
int bitnet_host_prefill(void *d, int n, int with_logits, int digits, float *elapsed_ms) {
    LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT);
    return D->prefill(n, with_logits != 0, digits, elapsed_ms);
}
int bitnet_host_prefill_sharded(void *d, int n, int rank, int world, bitnet_host_allgather_fn gather, void *gather_ctx, int with_logits,
                                int digits, int wire_f16, float *elapsed_ms) {
    LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT);
    return D->prefill_sharded(n, rank, world, gather, gather_ctx, with_logits != 0, digits, wire_f16 != 0, elapsed_ms);
}
int bitnet_host_rccl_allgather(void *nccl_comm, const void *send_dev, void *recv_dev, size_t bytes_per_rank, void *stream) {
    // ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
    using fn_t = int (*)(const void *, void *, size_t, int, void *, void *);
    static fn_t fn = [] {
        void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        return h ? reinterpret_cast<fn_t>(dlsym(h, "ncclAllGather")) : nullptr;
    }();
    if (!fn || !nccl_comm) return -1;
    return fn(send_dev, recv_dev, bytes_per_rank, /* ncclUint8 */ 1, nccl_comm, stream);
}
int bitnet_host_set_phase_timing(void *d, int on) {
    LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT);
    D->set_phase_timing(on != 0);
    return 0;
}
int bitnet_host_phase_times(void *d, float out[4]) {
    LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT);
    if (!out) return BITNET_HIP_ERR_INVALID_ARGUMENT;
    D->phase_times(out);
    return 0;
}
int bitnet_host_finish_prefill(void *d, int n, const float *last_row, int with_logits) {
    LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT);
    return D->finish_prefill(n, last_row, with_logits != 0);
}
void bitnet_host_layer_objects(void *d, int layer, uint64_t *handles4, void **ptrs4) {
    for (int i = 0; i < 4; ++i) handles4[i] = 0, ptrs4[i] = nullptr;
    if (Decoder *D = live(d)) D->layer_objects(layer, handles4, ptrs4);
}
void bitnet_host_global_objects(void *d, void **ptrs7) {
    for (int i = 0; i < 7; ++i) ptrs7[i] = nullptr;
    if (Decoder *D = live(d)) D->global_objects(ptrs7);
}
int bitnet_host_position(void *d) { LIVE(-1); return D->position(); }
int bitnet_host_last_prefill_path(void *d) { LIVE(-1); return D->last_prefill_path(); }
int bitnet_host_saturation_fallbacks(void *d) { LIVE(-1); return D->saturation_fallbacks(); }
int bitnet_host_history(void *d, int32_t *out, int n) { LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT); return D->history(out, n); }
int bitnet_host_last_logits(void *d, float *out) { LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT); return D->last_logits(out); }
int bitnet_host_last_hidden(void *d, float *out) { LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT); return D->last_hidden(out); }
int bitnet_host_blake3_hex(const void *data, size_t len, char *out65) {
    if (!out65 || (!data && len)) return -1;
    const std::string h = bitnet_host::Blake3::hex(data, len);
    memcpy(out65, h.c_str(), 65);
    return 0;
}
int bitnet_host_trace_step(void *d, const char *dir, int with_logits) { LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT); return D->trace_step(dir, with_logits != 0); }
int bitnet_host_probe_gateup(void *d, int reps, float *us_per_launch, double *bytes_per_launch) {
    LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT);
    return D->probe_gateup(reps, us_per_launch, bytes_per_launch);
}
int bitnet_host_probe_kernel(void *d, int kind, int reps, float *us_per_launch, double *bytes_per_launch) {
    LIVE(BITNET_HIP_ERR_INVALID_ARGUMENT);
    return D->probe_kernel(kind, reps, us_per_launch, bytes_per_launch);
}
uint64_t bitnet_host_weight_bytes(void *d) { LIVE(0); return D->weight_bytes(); }
}
