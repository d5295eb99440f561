// decoder.hpp -- host-side decode loop over the C ABI (include/bitnet_hip.h).
//
// The reference keeps the autoregressive loop in host code (Rust) and calls kernels:
//   TransformerModel::{embed, forward, logits}   crates/bitnet-transformer/src/lib.rs:1390, 1557, 1599
//   TransformerBlock::forward                    :977-1134
//   KVCache / LayerKVCache                       :1138-1256
//   greedy loop                                  crates/bitnet-cli/src/main.rs:1282-1477
//   parity entry points                          crates/bitnet-inference/src/parity.rs:30-111,157-223
// Rust is not available in this build environment, so this is the same structure in
// C++: it owns no arithmetic, only buffers, the launch order and a captured hipGraph
// per decode step.  Everything it launches goes through bitnet_hip_* entry points.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "bitnet_hip.h"

extern "C" {
// The collective of the token-parallel prefill as the host supplies it: gather `bytes_per_rank` bytes from every rank into
// recv ([world][bytes_per_rank], rank order), ordered on `stream` (a hipStream_t) like a kernel launch.  0 = success.
typedef int (*bitnet_host_allgather_fn)(void *ctx, const void *send_dev, void *recv_dev, size_t bytes_per_rank, void *stream);
}

namespace bitnet_host {

// ModelConfig fields the path needs (crates/bitnet-common/src/config.rs:25-46; GGUF keys
// crates/bitnet-models/src/gguf_simple.rs:646-777).
struct Config {
    int hidden = 2560, n_layers = 30, n_heads = 20, n_kv_heads = 5, head_dim = 128;
    int ffn = 6912, vocab = 128256, max_pos = 4096;
    float eps = 1e-5f;          // rms_norm_eps, default 1e-5 (T:957)
    float rope_theta = 10000.f;  // DEFAULT_ROPE_BASE (crates/bitnet-rope/src/lib.rs:11)
};

// One layer's host-side weights in the reference's layouts.
struct LayerWeightsQk256 {
    const float *attn_norm, *ffn_norm;                       // [hidden]
    const uint8_t *q, *k, *v, *o, *gate, *up, *down;         // QK256 bytes [out, ceil(in/256)*64]
};
// Ternary I2_S with f32 block scales (K/cpu/quantized_matmul.rs:47-56).
struct LayerWeightsI2s {
    const float *attn_norm, *ffn_norm;
    const uint8_t *w[7];      // q,k,v,o,gate,up,down: [out, in/4]
    const float *scales[7];   // [out, in/block]
    size_t block_size;
};

// One projection as a loader hands it over: QK256 bytes, or 2-bit codes + f32 block scales
// with an explicit code map (bitnet_hip_weights_upload_coded).
struct ProjSpec {
    bool qk256 = true;
    const uint8_t *bytes = nullptr;
    size_t len = 0;
    const float *scales = nullptr;  // coded only: [rows, cols / block]
    size_t n_scales = 0, block = 32;
    int8_t code_map[4] = {-2, -1, 0, 1};
};

class Decoder {
  public:
    explicit Decoder(const Config &cfg);
    ~Decoder();
    Decoder(const Decoder &) = delete;
    Decoder &operator=(const Decoder &) = delete;

    const Config &config() const { return c_; }
    const std::string &error() const { return err_; }
    bool dead() const { return dead_; }  // construction failed: only error() and the destructor may be used
    void set_error(const std::string &e) { err_ = e; }

    // weights (uploaded once; fused q|k|v and interleaved gate/up handles are built here)
    int set_layer_qk256(int layer, const LayerWeightsQk256 &w);
    int set_layer_i2s(int layer, const LayerWeightsI2s &w);
    // q,k,v,o,gate,up,down in any mix of storage forms (q|k|v and gate|up must agree)
    int set_layer_specs(int layer, const float *attn_norm, const float *ffn_norm, const ProjSpec p[7]);
    int set_globals(const uint16_t *embed_f16, const float *final_norm);

    // KVCache::clear (T:1251-1255) + token history
    int reset();
    // Put `n` forced tokens at positions [pos, pos+n) of the history (the prompt).
    int feed(const int32_t *tokens, int n);
    // Run `n` single-token steps (T:1482-1504 body each).  with_logits=false skips the
    // logits GEMV for prompt positions nobody samples from.  Uses the captured graph
    // when use_graph, else eager launches.  Synchronises the stream before returning.
    int run(int n, bool with_logits, bool use_graph, float *elapsed_ms);
    // The same n steps UNFUSED, in the reference's op order, every projection on the reference-order (bit-exact)
    // kernel: the checker bench.py and the tests hold the fast step against at the full model size.  Eager, slow.
    int run_reference(int n, bool with_logits);
    // Activations between the step's kernels: 1 (default) = quantised once by their producer ("QAct", include/bitnet_hip.h:
    // the f16-class activation north_star names), 0 = exact f32 (round 1's kernels).  Env BITNET_HOST_ACT sets the default.
    int set_act_mode(int mode);
    // Opt-in f16 KV cache (half the bytes of the long-context decode stream; the reference's cache is f32, T:1171-1202): k / v are
    // rounded once, when appended.  Only on a fresh sequence.  Env BITNET_HOST_KV16 sets the default.
    int set_kv_f16(bool on);
    bool kv_f16() const { return kv_f16_; }
    bool qact_path() const;  // mode 1 AND every layer's matrices are on that path
    // Whole-prompt forward on a fresh sequence (position() == 0): the first n fed tokens go
    // through every layer as [n, *] matrices (TransformerModel::forward with seq_len n,
    // T:1557-1597): tiled matmuls + causal attention, KV cache filled for positions 0..n-1.
    // with_logits samples the token after the prompt exactly as run() does for the last
    // prompt position.  digits: fixed-point digits per activation in the matmuls (2..4).
    int prefill(int n, bool with_logits, int digits, float *elapsed_ms);
    // Token-parallel prefill of ONE long prompt over `world` GPUs, one process per GPU (BASELINE configs[4]):
    // this rank runs the first n fed tokens' zigzag chunks rank and 2 world - 1 - rank through every layer (weights are
    // replicated: all seven projections are collective-free); per layer ONE all-gather of the raw k|v rows through
    // `gather` (RCCL: bitnet_host_rccl_allgather; tests: any callable), after which every rank fills its own KV cache
    // for all n positions.  n % (2 * world * 64) == 0.  wire_f16: k|v travel as f16.
    int prefill_sharded(int n, int rank, int world, bitnet_host_allgather_fn gather, void *gather_ctx, bool with_logits, int digits,
                        bool wire_f16, float *elapsed_ms);
    // Tail of a prefill driven from outside (token-parallel prefill, bitnet-rs_amd/prefill_parallel.py):
    // the KV cache already holds positions 0..n-1; last_row (device, [hidden]) is the residual stream
    // of position n-1 or null on ranks that do not own it (then only the position counter moves).
    int finish_prefill(int n, const float *last_row, bool with_logits);
    bool chain_applies(int digits) const;
    bool handover16_applies(int digits) const;
    bool hybrid_applies(size_t n_rows) const;
    int last_prefill_path() const { return last_prefill_path_; }
    int saturation_fallbacks() const { return saturation_fallbacks_; }
    int fp6_flag(int digits);  // BITNET_HIP_FUSE_FP6_DIGITS for the q|k|v and gate|up launches of the digit-plane prompt forward, or 0
    int ensure_chain_buffers(size_t N);
    int prefill_chain_layers(size_t N);
    bool qb32_applies(int digits, size_t n_rows);  // the QK256 prompt forward on QB32 rows (no row quantiser launch): long prompts, digits = 2
    int prefill_qb32_layers(size_t N);
    void set_phase_timing(bool on) { sp_timing_ = on; }
    void phase_times(float out[4]) const {
        for (int i = 0; i < 4; ++i) out[i] = sp_phase_us_[i];
    }
    // Device objects of one layer / of the model for such a driver: handles {qkv, o, gate|up, down},
    // pointers {attn_norm, ffn_norm, kcache, vcache}; globals {embed, final_norm, rope_sin, rope_cos,
    // history, pos, stream}.
    void layer_objects(int layer, uint64_t handles[4], void *ptrs[4]) const;
    void global_objects(void *ptrs[7]) const;
    int position();                                  // tokens consumed so far
    int history(int32_t *out, int n);                // first n tokens of the sequence
    int last_logits(float *out);                     // [vocab], of the last step run with logits
    int last_hidden(float *out);                     // residual stream after the last block (pre final norm)
    // Activation trace of ONE decode step in the reference's format (crates/bitnet-trace/src/lib.rs:46-168: one JSON file per
    // tensor -- name, shape, dtype, blake3 of the raw f32 bytes, rms, num_elements, seq, layer, stage -- in `dir`): the step
    // runs eagerly, every intermediate is materialised as f32 and read back after its kernel.  Stages: embeddings; per
    // block q_proj, k_proj, v_proj, attn_out, attn_residual, ffn_hidden, ffn_out; all_layers_out; logits.  Off the hot path.
    // run() does this for every step when BITNET_TRACE_DIR is set (the reference's switch).
    int trace_step(const char *dir, bool with_logits);
    // Dominant-kernel probe for bench.py: every layer's fused gate/up GEMV back to back in
    // one graph, `reps` replays; returns the mean time per launch and the algorithmic
    // bytes one launch reads.
    int probe_gateup(int reps, float *us_per_launch, double *bytes_per_launch);
    // Same for any kernel of the step: kind 0 q|k|v, 1 attention, 2 o_proj, 3 gate|up, 4 down, 5 logits.
    int probe_kernel(int kind, int reps, float *us_per_launch, double *bytes_per_launch);
    size_t weight_bytes() const { return weight_bytes_; }  // algorithmic bytes of all I2_S matrices
    void *stream() const { return stream_; }

  private:
    int fail(const char *what);
    int fail_arg(const char *what);
    bool dead_ = false;
    // attention form of a step: 0 = two kernels, 64-position chunks; 1 = one kernel + merging o-projection (short
    // contexts); 2 = two kernels, 128-position chunks (more chunks than CUs)
    struct Tracer;
    int step_launches(bool with_logits, int form, Tracer *tr = nullptr);
    int step_launches_reference(bool with_logits);
    int ensure_graph(bool with_logits, int form);
  public:
    // Captures and instantiates the step graph of every attention form a sequence can reach (by position: form_at) ahead of
    // time, so that no run() pays for a capture in the middle of a generation.  Nothing executes.
    int prepare_graphs(bool with_logits);
  private:
    int form_at(int pos) const;
    bool merge_ok_ = false;  // the o-projection can merge the attention chunk records itself (short contexts)

    Config c_;
    std::string err_;
    void *stream_ = nullptr;
    struct Layer {
        float *attn_norm = nullptr, *ffn_norm = nullptr;
        bitnet_hip_weights_t qkv = 0, o = 0, gateup = 0, down = 0;
        float *kcache = nullptr, *vcache = nullptr;
        bool q_ok = false;  // all four matrices take QAct inputs (bitnet_hip_gemv_q_supported)
    };
    int attn_launch(Layer &L, int form, float *out, void *qout);
    void release_layer(Layer &L);  // frees the layer's handles, subtracts their bytes, drops the captured graphs
    void drop_graphs();
    int adopt_projections(Layer &L, bitnet_hip_weights_t h[7]);
    std::vector<Layer> layers_;
    void *embed_ = nullptr;
    float *final_norm_ = nullptr;
    float *rope_sin_ = nullptr, *rope_cos_ = nullptr;
    float *x_ = nullptr, *x2_ = nullptr, *qkv_ = nullptr, *att_ = nullptr, *h_ = nullptr, *logits_ = nullptr;
    void *qa_x_ = nullptr, *qa_x2_ = nullptr, *qa_att_ = nullptr, *qa_h_ = nullptr;  // QAct records of x, x2, attention output, silu(gate)*up
    double *st_x_ = nullptr, *st_x2_ = nullptr;                                       // LayerNorm statistics pairs of x, x2
    int act_mode_ = 1;
    bool kv_f16_ = false;
    std::string trace_dir_;  // BITNET_TRACE_DIR at construction
    float *ref_n_ = nullptr, *ref_gu_ = nullptr, *ref_t_ = nullptr;  // unfused reference step: normalised row, gate|up tiles, projection out
    void *scratch_ = nullptr;
    float *attn_scratch_ = nullptr;
    int32_t *pos_ = nullptr, *n_forced_ = nullptr, *history_ = nullptr, *token_ = nullptr;
    int host_forced_ = 0;
    // sharded prefill buffers (grown on demand)
    int sp_cap_ = 0, sp_ctx_ = 0;
    void *sp_kv_send_ = nullptr, *sp_kv_all_ = nullptr;
    void *comm_stream_ = nullptr, *sp_ev_pack_ = nullptr, *sp_ev_gather_ = nullptr;  // the all-gather's own stream and its two ties to the compute stream
    int32_t *sp_block_pos_ = nullptr, *sp_tokens_ = nullptr;
    // per-phase timing of the sharded prefill (set_phase_timing): 8 timing events per layer on the compute / comm streams, and the
    // medians over the layers of the last timed call: [0] projections (q|k|v + pack, o, gate|up, down), [1] attention (both phases),
    // [2] the part of the all-gather the compute stream had to wait for, [3] the all-gather itself on its own stream; microseconds
    bool sp_timing_ = false;
    std::vector<void *> sp_tev_;
    float sp_phase_us_[4] = {0.f, 0.f, 0.f, 0.f};
    // prefill buffers (grown on demand)
    int pf_cap_ = 0;
    float *pf_x_ = nullptr, *pf_qkv_ = nullptr, *pf_att_ = nullptr, *pf_h_ = nullptr;
    void *pf_gemm_ws_ = nullptr, *pf_attn_ws_ = nullptr;
    // the f16 activation chain of the prompt forward (prefill_chain): f16 rows handed from kernel to kernel, LayerNorm statistics partials
    int pfc_cap_ = 0;
    void *pf_xh_ = nullptr, *pf_atth_ = nullptr, *pf_hh_ = nullptr;
    float *pf_stats_ = nullptr;
    void *pf_qb_ = nullptr;   // QB32 rows of gamma * x (the input of q|k|v and of gate|up), bitnet_hip_qb32_bytes(pfc_cap_, hidden)
    int pf_qb_cap_ = 0;
    bool force_scaled_ = false;   // the prompt is being repeated on the row-scaled forms after an f16 hand-over value was clamped (prefill)
    int saturation_fallbacks_ = 0;
    int last_prefill_path_ = 0;  // what the last prefill() ran: 0 digit planes (+ f16 hand-over / hybrid), 1 the f16 chain, 2 the QB32 chain
    int prefill_qb32_ = -1;   // BITNET_HOST_PREFILL_QB32: -1 not read yet; 1 = the QB32 chain where it applies (opt-in), 0 (default) = quantiser launches
    int prefill_fp6_ = -1;    // BITNET_HOST_PREFILL_FP6: -1 not yet decided, 0 int8 digit planes, 1 the fp6 x fp4 form on resident fp4 images (fp6_flag)
    int prefill_chain_ = -1;  // BITNET_HOST_PREFILL_CHAIN: -1 automatic (the block-scaled format, whose matmul runs on f16 activations anyway), 0 off, 1 on
    size_t pf_gemm_ws_bytes_ = 0, pf_attn_ws_bytes_ = 0;
    size_t weight_bytes_ = 0;
    static constexpr int kGraphs = 6;
    void *graph_exec_[kGraphs] = {};  // [2 * form + with_logits], forms 0..2 (form_at)
    void *graph_[kGraphs] = {};
    int logits_wgs_ = 512;  // two workgroups per CU: whole rounds on the 256 CUs (768 / 1280 workgroups are 15-20 % slower)
};

}  // namespace bitnet_host

// C shim for the Python tests / bench (ctypes).  Not part of the kernel boundary.
extern "C" {
typedef struct bitnet_host_config {
    int32_t hidden, n_layers, n_heads, n_kv_heads, head_dim, ffn, vocab, max_pos;
    float eps, rope_theta;
} bitnet_host_config;
void *bitnet_host_create(const bitnet_host_config *cfg);
void bitnet_host_destroy(void *d);
const char *bitnet_host_error(void *d);
int bitnet_host_set_layer_qk256(void *d, int layer, const float *attn_norm, const float *ffn_norm,
                                const uint8_t *q, const uint8_t *k, const uint8_t *v, const uint8_t *o,
                                const uint8_t *gate, const uint8_t *up, const uint8_t *down);
int bitnet_host_set_layer_i2s(void *d, int layer, const float *attn_norm, const float *ffn_norm,
                              const uint8_t *const *w7, const float *const *scales7, size_t block_size);
int bitnet_host_set_globals(void *d, const uint16_t *embed_f16, const float *final_norm);
int bitnet_host_reset(void *d);
int bitnet_host_feed(void *d, const int32_t *tokens, int n);
int bitnet_host_run(void *d, int n, int with_logits, int use_graph, float *elapsed_ms);
int bitnet_host_run_reference(void *d, int n, int with_logits);
int bitnet_host_prepare_graphs(void *d, int with_logits);
int bitnet_host_set_act_mode(void *d, int mode);
int bitnet_host_set_kv_f16(void *d, int on);
int bitnet_host_act_mode(void *d);
int bitnet_host_prefill(void *d, int n, int with_logits, int digits, float *elapsed_ms);
int bitnet_host_finish_prefill(void *d, int n, const float *last_row, int with_logits);
// per-phase medians of the NEXT bitnet_host_prefill_sharded calls (off by default: 8 event records per layer); out[4] = projections,
// attention, exposed all-gather wait, all-gather on its own stream -- microseconds per layer, medians over the layers of the last call
int bitnet_host_set_phase_timing(void *d, int on);
int bitnet_host_phase_times(void *d, float out[4]);
int bitnet_host_prefill_sharded(void *d, int n, int rank, int world, bitnet_host_allgather_fn gather, void *gather_ctx, int with_logits,
                                int digits, int wire_f16, float *elapsed_ms);
// ncclAllGather on an existing RCCL communicator (ctx = ncclComm_t), resolved from librccl.so at first use: the host library
// itself does not link RCCL.  Pass it as `gather` with the communicator as gather_ctx.
int bitnet_host_rccl_allgather(void *nccl_comm, const void *send_dev, void *recv_dev, size_t bytes_per_rank, void *stream);
void bitnet_host_layer_objects(void *d, int layer, uint64_t *handles4, void **ptrs4);
void bitnet_host_global_objects(void *d, void **ptrs7);
int bitnet_host_position(void *d);
int bitnet_host_history(void *d, int32_t *out, int n);
int bitnet_host_last_logits(void *d, float *out);
int bitnet_host_last_hidden(void *d, float *out);
int bitnet_host_trace_step(void *d, const char *dir, int with_logits);
int bitnet_host_blake3_hex(const void *data, size_t len, char *out65);  // the trace records' hash (host/blake3.hpp), 64 hex digits + NUL
int bitnet_host_probe_gateup(void *d, int reps, float *us_per_launch, double *bytes_per_launch);
int bitnet_host_probe_kernel(void *d, int kind, int reps, float *us_per_launch, double *bytes_per_launch);
uint64_t bitnet_host_weight_bytes(void *d);
}
