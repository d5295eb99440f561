// gguf.hpp -- GGUF I2_S ingestion for the decode path: file -> flavour -> device layout.
//
// Host-side mirror of the reference loader for the tensors the hot path needs
// (paths under /root/reference/crates/bitnet-models/src):
//   header, KV and tensor-info records      formats/gguf/types.rs:76-330, 522-636
//   data start + tensor sizes FROM OFFSETS  formats/gguf/reader.rs:27-57, 116-200
//   detect_i2s_flavor                       formats/gguf/types.rs:868-1066
//   loader QK256 decision / orientation     gguf_simple.rs:279-340, qk256_utils.rs:19-88
//   32-element flavours (split / inline)    gguf_simple.rs:1078-1285
//   config keys                             gguf_simple.rs:646-777
//   vendor tensor names                     weight_mapper.rs:232-250, 378-449
// Parsing is plain host code; the packed bytes go to the GPU untouched through
// bitnet_hip_weights_upload_{qk256,coded} (include/bitnet_hip.h), which re-tile them.
#pragma once

#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace bitnet_host {

enum GgufType : uint32_t { GGUF_F32 = 0, GGUF_F16 = 1, GGUF_F64 = 4, GGUF_I2_S = 36 };

struct GgufTensor {
    std::string name;
    std::vector<uint64_t> shape;  // in file order, read as a row-major shape like the reference does
    uint32_t type = 0;
    uint64_t offset = 0;  // relative to data_start
    uint64_t size = 0;    // next tensor's offset (or file end) - offset   (reader.rs:116-200)
    uint64_t nelems() const {
        uint64_t n = 1;
        for (uint64_t d : shape) n *= d;
        return n;
    }
};

struct GgufValue {
    uint32_t type = 0;  // GGUF value type id; 9 = array
    uint64_t u = 0;     // unsigned / bool payload
    int64_t i = 0;
    double f = 0.0;
    std::string s;
    uint32_t elem_type = 0;  // arrays
    uint64_t count = 0;      // arrays (elements are skipped, only counted)
};

enum I2sFlavor : int { FLAVOR_BITNET32_F16 = 0, FLAVOR_SPLIT32_WITH_SIBLING = 1, FLAVOR_GGML_QK256_NO_SCALE = 2 };

// types.rs:868-1066.  Returns the flavour, or -1 with the reference's byte-accounting text in *err.
int detect_i2s_flavor(size_t available, size_t nelems, bool has_scale_sibling, bool strict, const std::string &name,
                      std::string *err);
// crates/bitnet-quantization/src/lib.rs:86-90
size_t qk256_tolerance_bytes(size_t expected);
// gguf_simple.rs:279-306 (per-row block count, 128-byte slack, closer than the inline-f16 size)
bool loader_is_qk256(const std::vector<uint64_t> &shape, size_t available);
// qk256_utils.rs:62-88
void detect_qk256_orientation_by_bytes(uint64_t r0, uint64_t c0, size_t available, uint64_t *rows, uint64_t *cols);

struct GgufConfig {
    uint64_t vocab = 0, hidden = 0, n_layers = 0, n_heads = 0, n_kv_heads = 0, ffn = 0;
    bool has_rope_theta = false, has_eps = false;
    float rope_theta = 0.f, eps = 0.f;
};

class GgufFile {
  public:
    ~GgufFile();
    // mmap()s the file read-only; nullptr + *err on failure.
    static GgufFile *open(const char *path, std::string *err);
    // parses a caller-owned buffer (must outlive the object)
    static GgufFile *from_memory(const uint8_t *data, size_t len, std::string *err);

    uint32_t version() const { return version_; }
    uint32_t alignment() const { return alignment_; }
    size_t data_start() const { return data_start_; }
    size_t file_len() const { return len_; }
    const std::vector<GgufTensor> &tensors() const { return tensors_; }
    const GgufTensor *find(const std::string &name) const;
    const uint8_t *tensor_data(const GgufTensor &t) const { return data_ + data_start_ + t.offset; }
    const GgufValue *kv(const std::string &key) const;
    bool get_u32(const std::string &key, uint64_t *out) const;  // U32, or non-negative I32 (reader.rs:356-362)
    bool get_f32(const std::string &key, float *out) const;
    int config(GgufConfig *out, std::string *err) const;         // gguf_simple.rs:646-777
    // find_sibling_scale (gguf_simple.rs:898-979): first existing F16/F32/F64 candidate
    const GgufTensor *find_sibling_scale(const std::string &data_name) const;

  private:
    GgufFile() = default;
    int parse(std::string *err);
    const uint8_t *data_ = nullptr;
    size_t len_ = 0;
    bool mapped_ = false;
    uint32_t version_ = 0, alignment_ = 32;
    uint64_t data_offset_field_ = 0;
    size_t data_start_ = 0;
    std::vector<GgufTensor> tensors_;
    std::map<std::string, GgufValue> kv_;
    std::map<std::string, size_t> index_;
};

}  // namespace bitnet_host

extern "C" {
void *bitnet_host_gguf_open(const char *path);
void *bitnet_host_gguf_from_memory(const uint8_t *data, size_t len);
void bitnet_host_gguf_close(void *g);
const char *bitnet_host_gguf_error(void);  // thread-local text of the last failed gguf call
int64_t bitnet_host_gguf_tensor_count(void *g);
// shape[8]; returns 0 or -1
int bitnet_host_gguf_tensor_info(void *g, int64_t idx, char *name, size_t name_cap, uint64_t *shape, uint32_t *n_dims,
                                 uint32_t *type, uint64_t *offset, uint64_t *size);
uint64_t bitnet_host_gguf_data_start(void *g);
// cfg[6] = vocab, hidden, n_layers, n_heads, n_kv_heads, ffn; f[2] = rope_theta, eps (NaN when absent)
int bitnet_host_gguf_config(void *g, uint64_t *cfg, float *f);
int bitnet_host_gguf_detect_i2s_flavor(uint64_t available, uint64_t nelems, int has_scale_sibling, int strict);
int bitnet_host_gguf_loader_is_qk256(const uint64_t *shape, uint32_t n_dims, uint64_t available);
// The loader's per-projection step WITHOUT a device: tensor `idx` as an I2_S projection [rows, cols] through the flavour
// decision, size / bounds checks and the split into codes + scales (what load_gguf hands to the uploads), touching every
// byte it would upload.  0, or -1 + bitnet_host_gguf_error().  Used by the CPU tests and the ASan fuzzer.
int bitnet_host_gguf_check_projection(void *g, int64_t idx, uint64_t rows, uint64_t cols);
// Fills an existing decoder (bitnet_host_create with the file's config) from the file.
int bitnet_host_load_gguf(void *decoder, void *g);
}
