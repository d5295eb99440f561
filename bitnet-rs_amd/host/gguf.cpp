// gguf.cpp -- see gguf.hpp.
#include "gguf.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "decoder.hpp"

namespace bitnet_host {

namespace {

struct Cursor {
    const uint8_t *d;
    size_t len, o = 0;
    bool ok = true;
    template <class T>
    T take() {
        T v{};
        if (o + sizeof(T) > len) {
            ok = false;
            return v;
        }
        memcpy(&v, d + o, sizeof(T));
        o += sizeof(T);
        return v;
    }
    std::string str() {
        const uint64_t n = take<uint64_t>();
        if (!ok || n > len - o) {
            ok = false;
            return {};
        }
        std::string s(reinterpret_cast<const char *>(d + o), (size_t)n);
        o += (size_t)n;
        return s;
    }
};

size_t scalar_size(uint32_t vt) {
    switch (vt) {
        case 0: case 1: case 7: return 1;
        case 2: case 3: return 2;
        case 4: case 5: case 6: return 4;
        case 10: case 11: case 12: return 8;
        default: return 0;
    }
}

bool read_value(Cursor &c, uint32_t vt, GgufValue &v, int depth = 0) {
    v.type = vt;
    switch (vt) {
        case 0: v.u = c.take<uint8_t>(); break;
        case 1: v.i = c.take<int8_t>(); break;
        case 2: v.u = c.take<uint16_t>(); break;
        case 3: v.i = c.take<int16_t>(); break;
        case 4: v.u = c.take<uint32_t>(); break;
        case 5: v.i = c.take<int32_t>(); break;
        case 6: v.f = c.take<float>(); break;
        case 7: v.u = c.take<uint8_t>(); break;
        case 8: v.s = c.str(); break;
        case 10: v.u = c.take<uint64_t>(); break;
        case 11: v.i = c.take<int64_t>(); break;
        case 12: v.f = c.take<double>(); break;
        case 9: {
            v.elem_type = c.take<uint32_t>();
            v.count = c.take<uint64_t>();
            if (!c.ok) return false;
            if (v.elem_type == 8) {
                for (uint64_t i = 0; i < v.count && c.ok; ++i) {
                    const uint64_t n = c.take<uint64_t>();
                    if (!c.ok || n > c.len - c.o) return c.ok = false;
                    c.o += (size_t)n;
                }
            } else if (v.elem_type == 9) {
                if (depth > 4) return c.ok = false;
                for (uint64_t i = 0; i < v.count && c.ok; ++i) {
                    GgufValue inner;
                    read_value(c, 9, inner, depth + 1);
                }
            } else {
                const size_t es = scalar_size(v.elem_type);
                if (es == 0 || v.count > (c.len - c.o) / es) return c.ok = false;
                c.o += (size_t)v.count * es;
            }
            break;
        }
        default: return c.ok = false;
    }
    return c.ok;
}

bool known_tensor_type(uint32_t t) {  // types.rs:687-710
    switch (t) {
        case 0: case 1: case 2: case 3: case 4: case 6: case 7: case 8: case 9: case 10: case 11: case 12: case 13:
        case 14: case 15: case 24: case 36: return true;
        default: return false;
    }
}

// types.rs:168-215: a v3 header WITHOUT alignment/data_offset fields (what real files have) is
// recognised by the next 8 bytes being a small string length followed by key-like ASCII.
// returns 1 early variant, 0 standard-v3 fields follow
int early_v3_variant(const uint8_t *d, size_t len, size_t off) {
    uint64_t n;
    memcpy(&n, d + off, 8);
    if (!(n > 0 && n < 256)) return 0;
    if (off + 8 + n > len) return 0;
    const size_t sample = n < 20 ? (size_t)n : 20;
    for (size_t i = 0; i < sample; ++i) {
        const uint8_t b = d[off + 8 + i];
        const bool key_char = (b >= '0' && b <= '9') || (b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z') || b == '.' || b == '_' || b == '-';
        if (!key_char) return 0;
    }
    return 1;
}

size_t abs_diff(size_t a, size_t b) { return a > b ? a - b : b - a; }
size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

}  // namespace

size_t qk256_tolerance_bytes(size_t expected) {
    const double t = std::ceil((double)expected * 0.001);
    return (size_t)(t < 8.0 ? 8.0 : t);
}

int detect_i2s_flavor(size_t available, size_t nelems, bool has_sibling, bool strict, const std::string &name,
                      std::string *err) {
    const size_t b32 = ceil_div(nelems, 32), b256 = ceil_div(nelems, 256);
    const size_t split_need = b32 * 8, inline_need = b32 * 10, qk_need = b256 * 64;
    const size_t tol = strict ? 8 : qk256_tolerance_bytes(split_need < qk_need ? split_need : qk_need);
    const size_t ds = abs_diff(available, split_need), di = abs_diff(available, inline_need), dq = abs_diff(available, qk_need);
    if (dq == 0) return FLAVOR_GGML_QK256_NO_SCALE;
    if (di == 0) return FLAVOR_BITNET32_F16;
    if (ds == 0 && has_sibling) return FLAVOR_SPLIT32_WITH_SIBLING;
    if (dq <= tol) return FLAVOR_GGML_QK256_NO_SCALE;
    if (has_sibling && ds <= tol) return FLAVOR_SPLIT32_WITH_SIBLING;
    if (di <= tol) return FLAVOR_BITNET32_F16;
    if (ds <= tol) return FLAVOR_SPLIT32_WITH_SIBLING;  // data-only split, no sibling found (the reference warns)
    if (err) {
        char b[768];
        snprintf(b, sizeof(b),
                 "I2_S '%s': no valid flavor detected. Byte accounting:\n - available: %zu\n"
                 " - split_need (32-elem blocks, 8B/block): %zu (diff: %zu)\n"
                 " - inline_need (32-elem blocks, 10B/block): %zu (diff: %zu)\n"
                 " - qk256_need (256-elem blocks, 64B/block): %zu (diff: %zu)\n"
                 " - has_scale_sibling: %s\n - tolerance: +-%zu bytes (%s)",
                 name.c_str(), available, split_need, ds, inline_need, di, qk_need, dq, has_sibling ? "true" : "false", tol,
                 strict ? "strict mode" : "~0.1% size-proportional");
        *err = b;
    }
    return -1;
}

bool loader_is_qk256(const std::vector<uint64_t> &shape, size_t available) {
    size_t nelems = 1;
    for (uint64_t d : shape) nelems *= (size_t)d;
    const size_t rows = shape.size() == 2 ? (size_t)shape[0] : 1, cols = shape.size() == 2 ? (size_t)shape[1] : nelems;
    const size_t dq = abs_diff(available, rows * ceil_div(cols, 256) * 64);
    const size_t db = abs_diff(available, ceil_div(nelems, 32) * 10);
    return dq <= 128 && dq < db;
}

void detect_qk256_orientation_by_bytes(uint64_t r0, uint64_t c0, size_t available, uint64_t *rows, uint64_t *cols) {
    const size_t e_as_is = (size_t)r0 * ceil_div((size_t)c0, 256) * 64, e_tr = (size_t)c0 * ceil_div((size_t)r0, 256) * 64;
    if (abs_diff(available, e_tr) < abs_diff(available, e_as_is)) {
        *rows = c0;
        *cols = r0;
    } else {
        *rows = r0;
        *cols = c0;
    }
}

GgufFile::~GgufFile() {
    if (mapped_ && data_) munmap(const_cast<uint8_t *>(data_), len_);
}

GgufFile *GgufFile::open(const char *path, std::string *err) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) {
        if (err) *err = std::string("cannot open ") + path;
        return nullptr;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) {
        ::close(fd);
        if (err) *err = std::string("cannot stat ") + path;
        return nullptr;
    }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) {
        if (err) *err = std::string("mmap failed for ") + path;
        return nullptr;
    }
    GgufFile *g = new GgufFile();
    g->data_ = static_cast<const uint8_t *>(m);
    g->len_ = (size_t)st.st_size;
    g->mapped_ = true;
    if (g->parse(err) != 0) {
        delete g;
        return nullptr;
    }
    return g;
}

GgufFile *GgufFile::from_memory(const uint8_t *data, size_t len, std::string *err) {
    GgufFile *g = new GgufFile();
    g->data_ = data;
    g->len_ = len;
    if (g->parse(err) != 0) {
        delete g;
        return nullptr;
    }
    return g;
}

int GgufFile::parse(std::string *err) {
    auto bad = [&](const std::string &m) {
        if (err) *err = m;
        return -1;
    };
    if (len_ < 16) return bad("File too small to be a valid GGUF file");
    if (len_ < 24) return bad("Insufficient data for GGUF header");
    if (memcmp(data_, "GGUF", 4) != 0) return bad("Invalid GGUF magic number");
    Cursor c{data_, len_};
    c.o = 4;
    version_ = c.take<uint32_t>();
    const uint64_t n_tensors = c.take<uint64_t>();
    if (n_tensors > 100000) return bad("tensor_count exceeds limit 100000");
    const uint64_t n_kv = c.take<uint64_t>();
    if (n_kv > 10000) return bad("metadata_kv_count exceeds limit 10000");
    if (version_ >= 3 && len_ >= c.o + 12 && !early_v3_variant(data_, len_, c.o)) {
        uint32_t a = c.take<uint32_t>();
        if (a == 0 || (a & (a - 1)) != 0) a = 32;
        alignment_ = a;
        data_offset_field_ = c.take<uint64_t>();
    }
    if (version_ < 2 || version_ > 3) return bad("Unsupported GGUF version " + std::to_string(version_));
    for (uint64_t i = 0; i < n_kv; ++i) {
        std::string key = c.str();
        const uint32_t vt = c.take<uint32_t>();
        GgufValue v;
        if (!c.ok || !read_value(c, vt, v)) return bad("malformed metadata entry " + std::to_string(i) + " ('" + key + "')");
        kv_[key] = std::move(v);
    }
    tensors_.reserve((size_t)n_tensors);
    for (uint64_t i = 0; i < n_tensors; ++i) {
        GgufTensor t;
        t.name = c.str();
        const uint32_t nd = c.take<uint32_t>();
        if (!c.ok) return bad("truncated tensor info");
        if (nd > 8) return bad("tensor_dimensions exceeds limit 8");
        for (uint32_t d = 0; d < nd; ++d) {
            const uint64_t dim = c.take<uint64_t>();
            if (c.ok && dim == 0) return bad("Tensor dimension " + std::to_string(d) + " cannot be zero");
            if (dim > 1000000000ull) return bad("tensor_dimension exceeds limit");
            t.shape.push_back(dim);
        }
        t.type = c.take<uint32_t>();
        t.offset = c.take<uint64_t>();
        if (!c.ok) return bad("truncated tensor info");
        if (!known_tensor_type(t.type)) return bad("Unknown tensor type: " + std::to_string(t.type));
        index_[t.name] = tensors_.size();
        tensors_.push_back(std::move(t));
    }
    // reader.rs:27-57
    const size_t a = alignment_ ? alignment_ : 1, kv_end = c.o;
    if (version_ >= 3 && data_offset_field_ != 0 && data_offset_field_ >= kv_end && data_offset_field_ <= len_ &&
        data_offset_field_ % a == 0)
        data_start_ = (size_t)data_offset_field_;
    else
        data_start_ = (kv_end + a - 1) / a * a;
    // reader.rs:116-200: offsets are authoritative, sizes come from successive offsets
    for (size_t i = 0; i < tensors_.size(); ++i) {
        GgufTensor &t = tensors_[i];
        if (t.offset > std::numeric_limits<uint64_t>::max() / 2) return bad("Tensor '" + t.name + "' offset is suspiciously large");
        const size_t start = data_start_ + (size_t)t.offset;
        if (start > len_) return bad("Tensor '" + t.name + "' start position exceeds file size");
        const size_t end = i + 1 < tensors_.size() ? data_start_ + (size_t)tensors_[i + 1].offset : len_;
        if (end < start) return bad("Tensor '" + t.name + "' has decreasing offsets");
        t.size = end - start;
        if (start + t.size > len_) return bad("Tensor '" + t.name + "' extends beyond file");
    }
    return 0;
}

const GgufTensor *GgufFile::find(const std::string &name) const {
    auto it = index_.find(name);
    return it == index_.end() ? nullptr : &tensors_[it->second];
}

const GgufValue *GgufFile::kv(const std::string &key) const {
    auto it = kv_.find(key);
    return it == kv_.end() ? nullptr : &it->second;
}

bool GgufFile::get_u32(const std::string &key, uint64_t *out) const {
    const GgufValue *v = kv(key);
    if (!v) return false;
    if (v->type == 4) {
        *out = v->u;
        return true;
    }
    if (v->type == 5 && v->i >= 0) {
        *out = (uint64_t)v->i;
        return true;
    }
    return false;
}

bool GgufFile::get_f32(const std::string &key, float *out) const {
    const GgufValue *v = kv(key);
    if (!v || v->type != 6) return false;
    *out = (float)v->f;
    return true;
}

int GgufFile::config(GgufConfig *o, std::string *err) const {
    GgufConfig c;
    const GgufValue *tok = kv("tokenizer.ggml.tokens");
    if (tok && tok->type == 9 && tok->elem_type == 8)
        c.vocab = tok->count;
    else
        get_u32("llama.vocab_size", &c.vocab);
    for (const char *k : {"bitnet-b1.58.embedding_length", "llama.embedding_length", "model.embed_dim"})
        if (get_u32(k, &c.hidden)) break;
    bool have_layers = false;
    for (const char *k : {"bitnet-b1.58.block_count", "llama.block_count"})
        if ((have_layers = get_u32(k, &c.n_layers))) break;
    if (!have_layers) {  // discover_n_layers_from_tensors: blk.<i>.* / layers.<i>.*
        long best = -1;
        for (const GgufTensor &t : tensors_) {
            for (const char *pre : {"blk.", "layers."}) {
                const size_t pl = strlen(pre);
                if (t.name.compare(0, pl, pre) == 0) {
                    char *end = nullptr;
                    const long v = strtol(t.name.c_str() + pl, &end, 10);
                    if (end != t.name.c_str() + pl && *end == '.' && v > best) best = v;
                }
            }
        }
        c.n_layers = (uint64_t)(best + 1);
    }
    for (const char *k : {"bitnet-b1.58.attention.head_count", "llama.attention.head_count"})
        if (get_u32(k, &c.n_heads)) break;
    for (const char *k : {"bitnet-b1.58.attention.head_count_kv", "llama.attention.head_count_kv"})
        if (get_u32(k, &c.n_kv_heads)) break;
    for (const char *k : {"bitnet-b1.58.feed_forward_length", "llama.feed_forward_length"})
        if (get_u32(k, &c.ffn)) break;
    for (const char *k : {"bitnet-b1.58.rope.freq_base", "llama.rope.freq_base"})
        if ((c.has_rope_theta = get_f32(k, &c.rope_theta))) break;
    for (const char *k : {"bitnet-b1.58.attention.layer_norm_rms_epsilon", "llama.attention.layer_norm_rms_epsilon"})
        if ((c.has_eps = get_f32(k, &c.eps))) break;
    if (c.vocab == 0) {
        if (err) *err = "Failed to extract vocab_size from GGUF metadata (missing tokenizer.ggml.tokens)";
        return -1;
    }
    if (c.hidden == 0) {
        if (err) *err = "Failed to extract hidden_size from GGUF metadata (tried bitnet-b1.58.embedding_length, llama.embedding_length, model.embed_dim)";
        return -1;
    }
    *o = c;
    return 0;
}

static std::string replace_first(const std::string &s, const std::string &from, const std::string &to) {
    // Rust str::replace replaces every occurrence; tensor names hold the suffix once
    std::string out = s;
    size_t pos = 0;
    while ((pos = out.find(from, pos)) != std::string::npos) {
        out.replace(pos, from.size(), to);
        pos += to.size();
    }
    return out;
}

static std::string trim_end(const std::string &s, const std::string &suf) {
    std::string out = s;
    while (out.size() >= suf.size() && out.compare(out.size() - suf.size(), suf.size(), suf) == 0) out.resize(out.size() - suf.size());
    return out;
}

const GgufTensor *GgufFile::find_sibling_scale(const std::string &name) const {
    const std::string base = trim_end(trim_end(trim_end(name, ".weight"), ".data"), ".qweight");
    std::vector<std::string> cands = {replace_first(name, ".weight", ".scale"), replace_first(name, ".weight", ".scales"),
                                      replace_first(name, ".data", ".scale"),   replace_first(name, ".data", ".scales"),
                                      replace_first(name, ".qweight", ".scale"), replace_first(name, ".qweight", ".scales")};
    for (const char *s : {".scale", ".scales", "._scale", "._scales", "_scale", "_scales", ".q_scales", ".qh", ".d", ".scl", ".s"})
        cands.push_back(base + s);
    for (const char *s : {".scale", ".scales", "._scale", "._scales"}) cands.push_back(name + s);
    for (const std::string &cn : cands) {
        const GgufTensor *t = find(cn);
        if (t && cn != name && (t->type == GGUF_F32 || t->type == GGUF_F16 || t->type == GGUF_F64)) return t;
    }
    return nullptr;
}

// ---- file -> Decoder ----------------------------------------------------------------------------

namespace {

float f16_to_f32(uint16_t h) {  // exact
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {  // subnormal: value = man * 2^-24
            float f = (float)man * 5.9604644775390625e-8f;
            memcpy(&bits, &f, 4);
            bits |= sign;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float out;
    memcpy(&out, &bits, 4);
    return out;
}

uint16_t f32_to_f16(float f) {  // round to nearest even
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u);
    if (x >= 0x477ff000u) return sign | 0x7c00u;  // rounds to >= 65520 -> inf
    if (x < 0x33000001u) return sign;             // < 2^-25 (ties to even -> 0)
    const int e = (int)(x >> 23) - 127;
    uint32_t man = (x & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;
    uint32_t half = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) ++half;
    if (e < -14) return sign | (uint16_t)half;  // subnormal (carry into the normal range is encoded correctly)
    return sign | (uint16_t)(((uint32_t)(e + 15) << 10) + (half - 0x400u));
}

// F32 / F16 / F64 tensor -> f32 vector of n elements
bool dense_to_f32(const GgufFile &g, const GgufTensor &t, size_t n, std::vector<float> &out, std::string &err) {
    const size_t es = t.type == GGUF_F32 ? 4 : t.type == GGUF_F16 ? 2 : t.type == GGUF_F64 ? 8 : 0;
    if (es == 0) {
        err = "tensor '" + t.name + "' must be F32/F16/F64";
        return false;
    }
    if (t.nelems() < n || t.size < n * es) {
        err = "tensor '" + t.name + "' is smaller than expected";
        return false;
    }
    const uint8_t *p = g.tensor_data(t);
    out.resize(n);
    for (size_t i = 0; i < n; ++i) {
        if (es == 4) {
            memcpy(&out[i], p + 4 * i, 4);
        } else if (es == 2) {
            uint16_t b;
            memcpy(&b, p + 2 * i, 2);
            out[i] = f16_to_f32(b);
        } else {
            double d;
            memcpy(&d, p + 8 * i, 8);
            out[i] = (float)d;
        }
    }
    return true;
}

const GgufTensor *first_of(const GgufFile &g, std::initializer_list<std::string> names) {
    for (const std::string &n : names)
        if (const GgufTensor *t = g.find(n)) return t;
    return nullptr;
}

struct ProjStorage {  // keeps split-out codes / scales alive until the upload returns
    std::vector<uint8_t> codes;
    std::vector<float> scales;
};

// One I2_S projection -> ProjSpec, following load_gguf_enhanced's two decisions.
bool make_spec(const GgufFile &g, const GgufTensor &t, size_t rows, size_t cols, ProjSpec &sp, ProjStorage &st, std::string &err) {
    if (t.type != GGUF_I2_S) {
        err = "tensor '" + t.name + "' is not I2_S (type " + std::to_string(t.type) + "); this path takes packed 2-bit weights only";
        return false;
    }
    if (t.nelems() != rows * cols) {
        err = "tensor '" + t.name + "' has " + std::to_string(t.nelems()) + " elements, expected " + std::to_string(rows * cols);
        return false;
    }
    const size_t avail = (size_t)t.size;
    // llama.cpp writers (the Microsoft 2B file among them: the reference's own tests/gqa_shapes.rs:22 quotes k_proj as
    // [2560, 640]) label a matrix ne[0] = in first while the bytes are [out rows][in contiguous].  With in = 2560 and
    // out = 640 the per-row byte count of the LABELLED orientation (2560 rows x 3 blocks) does not match the data, the
    // reference's second pass (gguf_simple.rs:286-302) therefore does not take the tensor although its first pass skipped
    // it as QK256 -- it ends up in neither map.  The device path takes it in the orientation the model configuration
    // names (expected_qk256_shape, qk256_utils.rs:19-55) when THAT orientation's per-row byte count matches: the bytes are
    // never moved, only the label is read the other way round.
    const size_t relabel_need = rows * ceil_div(cols, 256) * 64;
    const bool relabelled = t.shape.size() == 2 && t.shape[0] == cols && t.shape[1] == rows && rows != cols &&
                            abs_diff(avail, relabel_need) <= 128 &&
                            abs_diff(avail, relabel_need) < abs_diff(avail, ceil_div(rows * cols, 32) * 10);  // the loader's own QK256-vs-inline tie-break
    if (t.shape.size() == 2 && (loader_is_qk256(t.shape, avail) || relabelled)) {
        // orientation: as-is or transposed LABEL (the bytes are never moved), gguf_simple.rs:318-362
        const bool ok = (t.shape[0] == rows && t.shape[1] == cols) || (t.shape[1] == rows && t.shape[0] == cols);
        if (!ok) {
            err = "QK256 '" + t.name + "': shape mismatch with the model config";
            return false;
        }
        const size_t need = rows * ceil_div(cols, 256) * 64;
        if (abs_diff(avail, need) > 128) {  // I2SQk256NoScale::new, Q/i2s_qk256.rs:85-110
            err = "I2SQk256NoScale: data size mismatch for '" + t.name + "': got " + std::to_string(avail) + ", expected " + std::to_string(need);
            return false;
        }
        {
            const size_t abs = g.data_start() + (size_t)t.offset;
            if (abs > g.file_len() || avail > g.file_len() - abs) {
                err = "I2_S '" + t.name + "': insufficient file data (need " + std::to_string(avail) + " at " + std::to_string(abs) + ", file " +
                      std::to_string(g.file_len()) + ")";
                return false;
            }
        }
        sp.qk256 = true;
        sp.bytes = g.tensor_data(t);
        sp.len = avail;
        return true;
    }
    // pass 1 of the reference loader (gguf_simple.rs:1085-1112) skips anything whose byte count is
    // within 0.1 % of the WHOLE-TENSOR QK256 size; with pass 2 rejecting it the tensor is in no map
    {
        const size_t ggml_need = ceil_div(rows * cols, 256) * 64;
        if (abs_diff(avail, ggml_need) <= qk256_tolerance_bytes(ggml_need)) {
            err = "I2_S '" + t.name + "': byte count matches whole-tensor QK256 but not the per-row layout; the reference loader drops such a tensor";
            return false;
        }
    }
    // 32-element flavours (gguf_simple.rs:1128-1285): split with sibling scales, or inline f16
    if (cols % 32 != 0) {
        err = "I2_S '" + t.name + "': 32-element blocks need cols % 32 == 0 on the device path";
        return false;
    }
    const size_t blocks = ceil_div(rows * cols, 32), split_need = blocks * 8, inline_need = blocks * 10;
    size_t need;
    if (abs_diff(avail, split_need) <= 128)
        need = split_need;
    else if (abs_diff(avail, inline_need) <= 128)
        need = inline_need;
    else {
        err = "I2_S '" + t.name + "': available bytes " + std::to_string(avail) + " don't match BitNet split (" +
              std::to_string(split_need) + " +- 128) or inline (" + std::to_string(inline_need) + " +- 128)";
        return false;
    }
    {
        // `need` may exceed the tensor's own byte count by the 128-byte slack: the reference then reads on into whatever
        // follows in the file and refuses only past its end (gguf_simple.rs:1196-1207) -- the same bound, the same words
        const size_t abs = g.data_start() + (size_t)t.offset;
        if (abs > g.file_len() || need > g.file_len() - abs) {
            err = "I2_S '" + t.name + "': insufficient file data (need " + std::to_string(need) + " at " + std::to_string(abs) + ", file " +
                  std::to_string(g.file_len()) + ")";
            return false;
        }
    }
    const uint8_t *raw = g.tensor_data(t);
    const GgufTensor *sib = g.find_sibling_scale(t.name);
    st.scales.resize(blocks);
    if (sib) {
        if (sib->nelems() < blocks) {
            err = "I2_S '" + t.name + "': scales=" + std::to_string(sib->nelems()) + " < blocks " + std::to_string(blocks);
            return false;
        }
        if (need != split_need) {
            err = "I2_S '" + t.name + "': expected " + std::to_string(split_need) + " bytes for GGML split, got " + std::to_string(need);
            return false;
        }
        std::vector<float> s;
        if (!dense_to_f32(g, *sib, blocks, s, err)) return false;
        st.scales = std::move(s);
        sp.bytes = raw;
        sp.len = need;
    } else {
        if (need != inline_need) {
            err = "I2_S '" + t.name + "': expected " + std::to_string(inline_need) + " bytes for inline f16, got " + std::to_string(need);
            return false;
        }
        st.codes.resize(blocks * 8);
        for (size_t b = 0; b < blocks; ++b) {
            memcpy(&st.codes[b * 8], raw + b * 10, 8);
            uint16_t bits;
            memcpy(&bits, raw + b * 10 + 8, 2);
            st.scales[b] = f16_to_f32(bits);
        }
        sp.bytes = st.codes.data();
        sp.len = st.codes.size();
    }
    sp.qk256 = false;
    sp.scales = st.scales.data();
    sp.n_scales = st.scales.size();
    sp.block = 32;
    const int8_t map[4] = {-2, -1, 0, 1};  // (code - 2) * scale: Q/utils.rs:76-91, Q/i2s.rs:181-237
    memcpy(sp.code_map, map, 4);
    return true;
}

}  // namespace

int load_gguf_into(Decoder &dec, const GgufFile &g) {
    std::string err;
    auto fail = [&](const std::string &m) {
        dec.set_error(m);
        return BITNET_HIP_ERR_INVALID_ARGUMENT;
    };
    GgufConfig gc;
    if (g.config(&gc, &err) != 0) return fail(err);
    const Config &c = dec.config();
    if ((uint64_t)c.hidden != gc.hidden || (uint64_t)c.vocab != gc.vocab || (uint64_t)c.n_layers != gc.n_layers ||
        (gc.n_heads && (uint64_t)c.n_heads != gc.n_heads) || (gc.n_kv_heads && (uint64_t)c.n_kv_heads != gc.n_kv_heads) ||
        (gc.ffn && (uint64_t)c.ffn != gc.ffn))
        return fail("decoder config does not match the GGUF metadata");
    const size_t H = c.hidden, QD = (size_t)c.n_heads * c.head_dim, KD = (size_t)c.n_kv_heads * c.head_dim, F = c.ffn;
    const size_t n[7] = {QD, KD, KD, H, F, F, H}, k[7] = {H, H, H, QD, H, H, F};
    // weight_mapper.rs:426-449 (blk.*), :232-250 (vendor styles)
    static const char *const blk[7] = {"attn_q", "attn_k", "attn_v", "attn_output", "ffn_gate", "ffn_up", "ffn_down"};
    static const char *const canon[7] = {"attention.q_proj", "attention.k_proj", "attention.v_proj", "attention.o_proj",
                                         "feed_forward.gate_proj", "feed_forward.up_proj", "feed_forward.down_proj"};
    static const char *const hf[7] = {"self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                                      "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"};
    for (int l = 0; l < c.n_layers; ++l) {
        const std::string L = std::to_string(l);
        ProjSpec sp[7];
        ProjStorage st[7];
        for (int i = 0; i < 7; ++i) {
            const GgufTensor *t = first_of(g, {"blk." + L + "." + blk[i] + ".weight", "layers." + L + "." + canon[i] + ".weight",
                                               "model.layers." + L + "." + hf[i] + ".weight"});
            if (!t && i == 3) t = g.find("blk." + L + ".attn_o.weight");
            if (!t) return fail("missing tensor for layer " + L + " projection " + blk[i]);
            if (!make_spec(g, *t, n[i], k[i], sp[i], st[i], err)) return fail(err);
        }
        const GgufTensor *an = first_of(g, {"blk." + L + ".attn_norm.weight", "layers." + L + ".attention_norm.weight",
                                            "model.layers." + L + ".input_layernorm.weight"});
        const GgufTensor *fn = first_of(g, {"blk." + L + ".ffn_norm.weight", "layers." + L + ".post_attention_layernorm.weight",
                                            "model.layers." + L + ".post_attention_layernorm.weight"});
        if (!an || !fn) return fail("missing norm weights for layer " + L);
        std::vector<float> anv, fnv;
        if (!dense_to_f32(g, *an, H, anv, err) || !dense_to_f32(g, *fn, H, fnv, err)) return fail(err);
        const int rc = dec.set_layer_specs(l, anv.data(), fnv.data(), sp);
        if (rc != 0) return rc;
    }
    // embedding (tied logits) + final norm; normalize_embed_and_lm_head gguf_simple.rs:1483-1530
    const GgufTensor *emb = first_of(g, {"token_embd.weight", "tok_embeddings.weight", "model.embed_tokens.weight"});
    const GgufTensor *fin = first_of(g, {"output_norm.weight", "norm.weight", "model.norm.weight", "final_norm.weight"});
    if (!emb || !fin) return fail("missing token embedding or final norm tensor");
    if (emb->shape.size() != 2) return fail("token embedding must be 2-D");
    const size_t V = c.vocab;
    bool as_is = emb->shape[0] == V && emb->shape[1] == H, transposed = emb->shape[0] == H && emb->shape[1] == V;
    if (!as_is && !transposed) return fail("token embedding has unexpected shape");
    // The reference builds its tensors with the file's dimension order as row-major extents (gguf_simple.rs:1033-1073) and
    // transposes a [hidden, vocab] embedding physically (:1483-1530) -- restated here.  A llama.cpp writer lists ne[0] = hidden
    // FIRST for bytes that are already [vocab][hidden]; BITNET_GGUF_GGML_DIMS=1 reads such a label that way (no data
    // movement).  Opt-in: which of the two the real Microsoft file needs cannot be checked in this build (no file).
    if (transposed && !as_is) {
        const char *e = getenv("BITNET_GGUF_GGML_DIMS");
        if (e && e[0] == '1') {
            as_is = true;
            transposed = false;
        }
    }
    std::vector<uint16_t> table(V * H);
    const uint8_t *p = g.tensor_data(*emb);
    if (emb->type == GGUF_F16) {
        if (emb->size < V * H * 2) return fail("token embedding data too small");
        if (as_is) {
            memcpy(table.data(), p, V * H * 2);
        } else {
            for (size_t h = 0; h < H; ++h)
                for (size_t v = 0; v < V; ++v) memcpy(&table[v * H + h], p + 2 * (h * V + v), 2);
        }
    } else {
        std::vector<float> f;
        if (!dense_to_f32(g, *emb, V * H, f, err)) return fail(err);
        // the device table is f16: F32/F64 embeddings are rounded once here
        for (size_t v = 0; v < V; ++v)
            for (size_t h = 0; h < H; ++h) {
                table[v * H + h] = f32_to_f16(as_is ? f[v * H + h] : f[h * V + v]);
            }
    }
    std::vector<float> finv;
    if (!dense_to_f32(g, *fin, H, finv, err)) return fail(err);
    return dec.set_globals(table.data(), finv.data());
}

}  // namespace bitnet_host

// ---- C shim ---------------------------------------------------------------------------------------
using namespace bitnet_host;
static thread_local std::string g_gguf_error;

extern "C" {

void *bitnet_host_gguf_open(const char *path) {
    g_gguf_error.clear();
    return path ? GgufFile::open(path, &g_gguf_error) : nullptr;
}
void *bitnet_host_gguf_from_memory(const uint8_t *data, size_t len) {
    g_gguf_error.clear();
    return data ? GgufFile::from_memory(data, len, &g_gguf_error) : nullptr;
}
void bitnet_host_gguf_close(void *g) { delete static_cast<GgufFile *>(g); }
const char *bitnet_host_gguf_error(void) { return g_gguf_error.c_str(); }
int64_t bitnet_host_gguf_tensor_count(void *g) { return g ? (int64_t) static_cast<GgufFile *>(g)->tensors().size() : -1; }
int bitnet_host_gguf_tensor_info(void *g, int64_t idx, char *name, size_t name_cap, uint64_t *shape, uint32_t *n_dims,
                                 uint32_t *type, uint64_t *offset, uint64_t *size) {
    if (!g) return -1;
    const auto &ts = static_cast<GgufFile *>(g)->tensors();
    if (idx < 0 || (size_t)idx >= ts.size()) {
        g_gguf_error = "Tensor index " + std::to_string(idx) + " out of bounds";
        return -1;
    }
    const GgufTensor &t = ts[(size_t)idx];
    if (name && name_cap) snprintf(name, name_cap, "%s", t.name.c_str());
    if (shape)
        for (size_t i = 0; i < t.shape.size() && i < 8; ++i) shape[i] = t.shape[i];
    if (n_dims) *n_dims = (uint32_t)t.shape.size();
    if (type) *type = t.type;
    if (offset) *offset = t.offset;
    if (size) *size = t.size;
    return 0;
}
uint64_t bitnet_host_gguf_data_start(void *g) { return g ? static_cast<GgufFile *>(g)->data_start() : 0; }
int bitnet_host_gguf_config(void *g, uint64_t *cfg, float *f) {
    if (!g || !cfg || !f) return -1;
    GgufConfig c;
    if (static_cast<GgufFile *>(g)->config(&c, &g_gguf_error) != 0) return -1;
    cfg[0] = c.vocab, cfg[1] = c.hidden, cfg[2] = c.n_layers, cfg[3] = c.n_heads, cfg[4] = c.n_kv_heads, cfg[5] = c.ffn;
    f[0] = c.has_rope_theta ? c.rope_theta : NAN;
    f[1] = c.has_eps ? c.eps : NAN;
    return 0;
}
int bitnet_host_gguf_detect_i2s_flavor(uint64_t available, uint64_t nelems, int has_scale_sibling, int strict) {
    return detect_i2s_flavor((size_t)available, (size_t)nelems, has_scale_sibling != 0, strict != 0, "", &g_gguf_error);
}
int bitnet_host_gguf_loader_is_qk256(const uint64_t *shape, uint32_t n_dims, uint64_t available) {
    return loader_is_qk256(std::vector<uint64_t>(shape, shape + n_dims), (size_t)available) ? 1 : 0;
}
int bitnet_host_gguf_check_projection(void *gp, int64_t idx, uint64_t rows, uint64_t cols) {
    if (!gp) return -1;
    const GgufFile &g = *static_cast<GgufFile *>(gp);
    if (idx < 0 || (size_t)idx >= g.tensors().size()) {
        g_gguf_error = "tensor index out of range";
        return -1;
    }
    if (rows == 0 || cols == 0 || rows > (1ull << 31) || cols > (1ull << 31)) {
        g_gguf_error = "bad projection shape";
        return -1;
    }
    ProjSpec sp;
    ProjStorage st;
    std::string err;
    if (!make_spec(g, g.tensors()[(size_t)idx], (size_t)rows, (size_t)cols, sp, st, err)) {
        g_gguf_error = err;
        return -1;
    }
    // every byte an upload would read
    volatile uint64_t sink = 0;
    for (size_t i = 0; i < sp.len; ++i) sink += sp.bytes[i];
    for (size_t i = 0; i < sp.n_scales; ++i) sink += sp.scales[i] != 0.0f;
    (void)sink;
    return 0;
}
int bitnet_host_load_gguf(void *decoder, void *g) {
    if (!decoder || !g || static_cast<Decoder *>(decoder)->dead()) return BITNET_HIP_ERR_INVALID_ARGUMENT;
    return load_gguf_into(*static_cast<Decoder *>(decoder), *static_cast<GgufFile *>(g));
}
}
