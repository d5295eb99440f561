// fuzz_gguf.cpp -- mutation fuzzer for the GGUF reader (bitnet-rs_amd/host/gguf.cpp), built with
// AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on the pool).
//
//   tools/fuzz_gguf.sh [iterations]            builds and runs over tests/golden/*.gguf
//
// Every iteration takes one of the fixture files, applies a few random mutations (bit flips, byte
// overwrites with "interesting" values, 32/64-bit length fields blown up, truncation) and pushes it
// through everything that touches file bytes without a GPU: parse, config, tensor table, sibling
// lookup, flavour detection.  A malformed file must produce an error string, never a fault.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gguf.hpp"

using namespace bitnet_host;

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {  // splitmix64
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static std::vector<uint8_t> slurp(const char *path) {
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n > 0 ? size_t(n) : 0);
    if (n > 0 && fread(v.data(), 1, v.size(), f) != v.size()) v.clear();
    fclose(f);
    return v;
}

static void exercise(const std::vector<uint8_t> &buf, uint64_t *ok, uint64_t *bad) {
    // exact-size heap copy so that ASan sees any read past the end
    uint8_t *p = static_cast<uint8_t *>(malloc(buf.size() ? buf.size() : 1));
    if (!buf.empty()) memcpy(p, buf.data(), buf.size());
    std::string err;
    GgufFile *g = GgufFile::from_memory(p, buf.size(), &err);
    if (!g) {
        ++*bad;
        free(p);
        return;
    }
    ++*ok;
    GgufConfig cfg;
    std::string cerr;
    (void)g->config(&cfg, &cerr);
    uint64_t u;
    float f;
    (void)g->get_u32("general.alignment", &u);
    (void)g->get_f32("llama.rope.freq_base", &f);
    volatile uint64_t sink = 0;
    for (const GgufTensor &t : g->tensors()) {
        const GgufTensor *again = g->find(t.name);
        if (!again) abort();
        const GgufTensor *sib = g->find_sibling_scale(t.name);
        std::string derr;
        int fl = detect_i2s_flavor(size_t(t.size), size_t(t.nelems()), sib != nullptr, (rnd() & 1) != 0, t.name, &derr);
        (void)fl;
        (void)loader_is_qk256(t.shape, size_t(t.size));
        if (t.shape.size() == 2) {
            uint64_t r, c;
            detect_qk256_orientation_by_bytes(t.shape[0], t.shape[1], size_t(t.size), &r, &c);
        }
        // every byte the table says belongs to the tensor must be inside the buffer
        const uint8_t *d = g->tensor_data(t);
        if (t.size) sink += d[0] + d[t.size - 1];
        // the loader's projection step (flavour decision, +-128-byte slack, split into codes + scales) reads what an upload
        // would: as labelled and with the transposed label
        if (t.shape.size() == 2 && t.shape[0] && t.shape[1] && t.shape[0] < (1u << 20) && t.shape[1] < (1u << 20)) {
            const int64_t idx = &t - g->tensors().data();
            (void)bitnet_host_gguf_check_projection(g, idx, t.shape[0], t.shape[1]);
            (void)bitnet_host_gguf_check_projection(g, idx, t.shape[1], t.shape[0]);
        }
    }
    delete g;
    free(p);
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s iterations file.gguf [file.gguf ...]\n", argv[0]);
        return 2;
    }
    const long iters = atol(argv[1]);
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 2; i < argc; ++i) {
        seeds.push_back(slurp(argv[i]));
        if (seeds.back().empty()) {
            fprintf(stderr, "cannot read %s\n", argv[i]);
            return 2;
        }
    }
    static const uint64_t interesting[] = {0,          1,          0x7f,       0x80,       0xff,       0x100,
                                           0x7fff,     0xffff,     0x7fffffff, 0x80000000, 0xffffffff, 0x100000000ull,
                                           0x7fffffffffffffffull, 0x8000000000000000ull, 0xffffffffffffffffull, 32, 36, 9};
    const size_t n_int = sizeof(interesting) / sizeof(interesting[0]);
    uint64_t ok = 0, bad = 0;
    for (const auto &s : seeds) exercise(s, &ok, &bad);  // the clean files parse
    if (ok != seeds.size()) {
        fprintf(stderr, "a seed file did not parse\n");
        return 1;
    }
    for (long it = 0; it < iters; ++it) {
        std::vector<uint8_t> b = seeds[rnd() % seeds.size()];
        // most of a GGUF file's structure is in the header: bias mutations to the first 2 KiB
        const size_t hot = b.size() < 2048 ? b.size() : 2048;
        const int n_mut = 1 + int(rnd() % 4);
        for (int m = 0; m < n_mut && !b.empty(); ++m) {
            const size_t pos = (rnd() % 4) ? rnd() % hot : rnd() % b.size();
            switch (rnd() % 6) {
                case 0: b[pos] ^= uint8_t(1u << (rnd() % 8)); break;
                case 1: b[pos] = uint8_t(rnd()); break;
                case 2: {
                    uint32_t v = uint32_t(interesting[rnd() % n_int]);
                    if (pos + 4 <= b.size()) memcpy(&b[pos], &v, 4);
                    break;
                }
                case 3: {
                    uint64_t v = interesting[rnd() % n_int];
                    if (pos + 8 <= b.size()) memcpy(&b[pos], &v, 8);
                    break;
                }
                case 4: b.resize(rnd() % (b.size() + 1)); break;
                case 5: {  // duplicate a span over another place (shifts record boundaries)
                    const size_t len = rnd() % 64, src = rnd() % b.size();
                    for (size_t i = 0; i < len && pos + i < b.size() && src + i < b.size(); ++i) b[pos + i] = b[src + i];
                    break;
                }
            }
        }
        exercise(b, &ok, &bad);
    }
    printf("fuzz_gguf: %ld mutated inputs, %llu parsed, %llu rejected, no faults\n", iters,
           (unsigned long long)(ok - seeds.size()), (unsigned long long)bad);
    return 0;
}
