// blake3.hpp -- BLAKE3 (default hash mode, 32-byte output) for the activation trace records: the reference's TraceRecord
// carries `blake3` of a tensor's raw little-endian f32 bytes (crates/bitnet-trace/src/lib.rs:130-136).  Written from the
// published specification (portable scalar form: chunks of 1024 B, 64-B blocks, 7-round compression, binary tree of chaining
// values); checked in tests/test_trace.py against the specification's published test vectors.  Host-side, off every hot path.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace bitnet_host {

class Blake3 {
  public:
    static std::string hex(const void *data, size_t len) {
        static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
        const uint8_t *p = static_cast<const uint8_t *>(data);
        const size_t n_chunks = len == 0 ? 1 : (len + 1023) / 1024;
        std::vector<Output> stack;  // chaining values of completed subtrees, left to right
        Output last{};
        for (size_t c = 0; c < n_chunks; ++c) {
            const size_t off = c * 1024, clen = len - off < 1024 ? len - off : 1024;
            Output o = chunk_output(IV, p + off, clen, (uint64_t)c);
            if (c + 1 == n_chunks) {
                last = o;
                break;
            }
            // a completed chunk: merge with every completed subtree of the same size (trailing one bits of the chunk count)
            uint32_t cv[8];
            chaining_value(o, cv);
            uint64_t total = c + 1;
            while ((total & 1) == 0) {
                Output parent = parent_output(IV, stack.back().cv_out, cv);
                stack.pop_back();
                chaining_value(parent, cv);
                total >>= 1;
            }
            Output keep{};
            memcpy(keep.cv_out, cv, 32);
            stack.push_back(keep);
        }
        // fold the stack onto the last chunk, right to left; the final node is the root
        Output node = last;
        for (size_t i = stack.size(); i-- > 0;) {
            uint32_t cv[8];
            chaining_value(node, cv);
            node = parent_output(IV, stack[i].cv_out, cv);
        }
        uint32_t out[16];
        compress(node.in_cv, node.block, 0, node.block_len, node.flags | ROOT, out);
        static const char *d = "0123456789abcdef";
        std::string s;
        for (int i = 0; i < 8; ++i)
            for (int b = 0; b < 4; ++b) {
                const uint8_t v = (uint8_t)(out[i] >> (8 * b));
                s.push_back(d[v >> 4]);
                s.push_back(d[v & 15]);
            }
        return s;
    }

  private:
    enum : uint32_t { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };
    struct Output {
        uint32_t in_cv[8];   // input chaining value of the node's LAST compression
        uint32_t block[16];  // its message block
        uint64_t counter;
        uint32_t block_len, flags;
        uint32_t cv_out[8];  // (stack entries only) the node's chaining value
    };
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    static void g(uint32_t *s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
        s[a] = s[a] + s[b] + mx;
        s[d] = rotr(s[d] ^ s[a], 16);
        s[c] = s[c] + s[d];
        s[b] = rotr(s[b] ^ s[c], 12);
        s[a] = s[a] + s[b] + my;
        s[d] = rotr(s[d] ^ s[a], 8);
        s[c] = s[c] + s[d];
        s[b] = rotr(s[b] ^ s[c], 7);
    }
    static void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len, uint32_t flags, uint32_t out[16]) {
        static const uint32_t IV[4] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au};
        static const int PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
        uint32_t s[16], m[16];
        memcpy(s, cv, 32);
        memcpy(s + 8, IV, 16);
        s[12] = (uint32_t)counter;
        s[13] = (uint32_t)(counter >> 32);
        s[14] = block_len;
        s[15] = flags;
        memcpy(m, block, 64);
        for (int r = 0; r < 7; ++r) {
            g(s, 0, 4, 8, 12, m[0], m[1]);
            g(s, 1, 5, 9, 13, m[2], m[3]);
            g(s, 2, 6, 10, 14, m[4], m[5]);
            g(s, 3, 7, 11, 15, m[6], m[7]);
            g(s, 0, 5, 10, 15, m[8], m[9]);
            g(s, 1, 6, 11, 12, m[10], m[11]);
            g(s, 2, 7, 8, 13, m[12], m[13]);
            g(s, 3, 4, 9, 14, m[14], m[15]);
            uint32_t t[16];
            for (int i = 0; i < 16; ++i) t[i] = m[PERM[i]];
            memcpy(m, t, 64);
        }
        for (int i = 0; i < 8; ++i) {
            out[i] = s[i] ^ s[i + 8];
            out[i + 8] = s[i + 8] ^ cv[i];
        }
    }
    static void load_block(const uint8_t *p, size_t n, uint32_t block[16]) {
        uint8_t b[64] = {0};
        memcpy(b, p, n);
        for (int i = 0; i < 16; ++i) block[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    }
    static Output chunk_output(const uint32_t key[8], const uint8_t *p, size_t len, uint64_t counter) {
        uint32_t cv[8];
        memcpy(cv, key, 32);
        const size_t n_blocks = len == 0 ? 1 : (len + 63) / 64;
        Output o{};
        for (size_t b = 0; b < n_blocks; ++b) {
            const size_t off = b * 64, blen = len - off < 64 ? len - off : 64;
            uint32_t block[16];
            load_block(p + off, blen, block);
            const uint32_t flags = (b == 0 ? CHUNK_START : 0u) | (b + 1 == n_blocks ? CHUNK_END : 0u);
            if (b + 1 == n_blocks) {
                memcpy(o.in_cv, cv, 32);
                memcpy(o.block, block, 64);
                o.counter = counter;
                o.block_len = (uint32_t)blen;
                o.flags = flags;
            } else {
                uint32_t out[16];
                compress(cv, block, counter, 64, flags, out);
                memcpy(cv, out, 32);
            }
        }
        return o;
    }
    static void chaining_value(const Output &o, uint32_t cv[8]) {
        uint32_t out[16];
        compress(o.in_cv, o.block, o.counter, o.block_len, o.flags, out);
        memcpy(cv, out, 32);
    }
    static Output parent_output(const uint32_t key[8], const uint32_t left[8], const uint32_t right[8]) {
        Output o{};
        memcpy(o.in_cv, key, 32);
        memcpy(o.block, left, 32);
        memcpy(o.block + 8, right, 32);
        o.counter = 0;
        o.block_len = 64;
        o.flags = PARENT;
        return o;
    }
};

}  // namespace bitnet_host
