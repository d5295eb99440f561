// persistent_layer_probe.hip -- VERDICT r02 item 3(ii): ONE persistent launch for the batch-1 decode chain, built the way
// MI355X_MICROARCH.md's "engine-vs-launches" row and cdna_hip_programming.md Guideline 16 prescribe, at the byte sizes of a
// bitnet-b1.58-2B-4T layer (BitNet32-F16: 1 KiB of codes + 256 B of scales per 16-row x 256-column tile):
//
//   * one 9-wave workgroup per CU (256 workgroups): waves 0..7 compute, wave 8 is the LOADER;
//   * the loader keeps a whole layer of this CU's weight share in LDS (qkv 13 KiB, o 13, gate|up 50, down 34 = 110 KiB):
//     the moment the compute waves have finished with phase p of layer l (the phase's second s_barrier) it re-fills that
//     region with phase p of layer l + 1 by LDS-DMA (global_load_lds_dwordx4, non-temporal) -- one layer of run-ahead, so the
//     weights of a phase are in LDS long before its activation vector arrives;
//   * hand-offs are data-tagged 8-byte granules {tag = phase epoch, value} (Guideline 16 R2): producers store them write-through
//     (sc1), every consuming workgroup sweeps the granules of its input vector with sc1 loads (all 8 compute waves, each its own
//     slice, every load of a slice in flight at once) until every tag matches -- no flag, no fence, no agent-scope acquire;
//   * phases per layer: q|k|v (240 row tiles) -> attention (15 workgroups: 5 KV heads x 3 chunks of 64 keys, a 64 KB K/V read
//     each) -> o-projection merging the 15 chunk records (160 tiles) -> gate|up (432 tile pairs) -> down (160 tiles, K = 6912).
//
// The arithmetic is the real kernel's per 1-KiB tile (2-bit code expansion by v_perm_b32, 4 x v_mfma_i32_16x16x64_i8 with the
// activation digits as the A operand from LDS) on synthetic data, INTEGER end to end, so the result of the whole chain is
// independent of the partition and of timing: the host recomputes it serially and compares every word (a stale or torn
// hand-off cannot pass).  Every spin is bounded by s_memrealtime (2 ms) and raises a global abort word that every other spin
// watches: the probe cannot hang the GPU.
//
// Output: per-phase medians over workgroups of  edge wait (sweep start -> vector in LDS) / compute / publish,  the layer period,
// and the same numbers with the loader switched off (mode 1: weights fetched from global memory when the phase starts).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/persistent_layer_probe tools/probes/persistent_layer_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x)                                                                   \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

// ---- model shape (bitnet-b1.58-2B-4T) -------------------------------------------------------------------------------
constexpr int H = 2560, F = 6912, NHEAD = 20, NKV = 5, QKV = (NHEAD + 2 * NKV) * 128;  // 3840
constexpr int NCH = 3, NREC = NKV * NCH, RECV = 528;                                    // 4 x 128 + 8, padded to 33 tiles
constexpr int G = 256, NCW = 7, NT = (NCW + 1) * 64;  // 7 compute waves + the loader = 2 waves per SIMD (256 registers each)
constexpr int SW = NCW * 64;                           // granules per sweep round of a workgroup
constexpr int NPH = 5;  // qkv, att, o, gu, dn
constexpr unsigned kLut = 0xff000100u;  // ternary {0, +1, 0, -1}: byte c = value of code c

// granule buffers (values per vector)
constexpr int NX = H, NQKV = QKV, NRECV = NREC * RECV, NH = F;
// weight pieces (1 KiB) per workgroup and phase, codes only; the loader fetches 1.25 x as many (the f16 block scales)
constexpr int kTilesQkv = 1, kTilesO = 1, kPairsGu = 2, kTilesDn = 1;
constexpr int kPiecesQkv = kTilesQkv * 10, kPiecesO = kTilesO * 10, kPiecesGu = kPairsGu * 2 * 10, kPiecesDn = kTilesDn * 27;
constexpr int ld125(int p) { return (p * 5 + 3) / 4; }
constexpr int kRegQkv = 0, kRegO = kRegQkv + ld125(kPiecesQkv), kRegGu = kRegO + ld125(kPiecesO), kRegDn = kRegGu + ld125(kPiecesGu),
              kRegEnd = kRegDn + ld125(kPiecesDn);  // in KiB: 13 + 13 + 50 + 34 = 110
constexpr int kVecBytes = 768 * 4;                   // u32 values: only the attention phase wants them (its 768 inputs)
constexpr int kPlane = 7936;                         // digit planes (byte 0 / byte 1 of every value) of the largest matrix input (15 records)
constexpr int kLdsBytes = kRegEnd * 1024 + kVecBytes + 2 * kPlane + 1024 /* zero area */ + NCW * 4 * 16 * 4 /* partials */ + 64;

struct Params {
    const unsigned char *w;  // [layer][phase region][workgroup][pieces][1 KiB]
    size_t layer_stride;     // bytes
    const unsigned char *kv; // [layer][NREC][64 KiB]
    gu64 *gx, *gqkv, *grec, *gx2, *gh;
    int n_layers, n_iter, mode;  // mode 0: loader + LDS ring; 1: weights straight from global memory at phase start
    unsigned *abort_word;
    u64 *stamps;  // [layer][phase][workgroup][4]  (last iteration)
    unsigned *out;           // final x [H]
};

__device__ __forceinline__ v4i decode16(unsigned w) {
    v4i a;
    a[0] = (int)__builtin_amdgcn_perm(0u, kLut, w & 0x03030303u);
    a[1] = (int)__builtin_amdgcn_perm(0u, kLut, (w >> 2) & 0x03030303u);
    a[2] = (int)__builtin_amdgcn_perm(0u, kLut, (w >> 4) & 0x03030303u);
    a[3] = (int)__builtin_amdgcn_perm(0u, kLut, (w >> 6) & 0x03030303u);
    return a;
}

// LDS-DMA: 64 lanes x 16 B from global memory to LDS bytes [lds_dst, lds_dst + 1024), non-temporal (cdna_hip_programming.md 5.7)
__device__ __forceinline__ void dma1k(const void *gsrc_lane, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_dst)
                 : "memory");
}

struct Spin {
    u64 t0;
    unsigned *abort_word;
    __device__ __forceinline__ bool expired(unsigned it) const {
        if ((it & 31u) != 31u) return false;
        if (__hip_atomic_load((gu32 *)abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000ull) {  // 2 ms at 100 MHz
            __hip_atomic_store((gu32 *)abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        return false;
    }
};

// Sweep n granules (a list of up to 3 ranges) into vec[]: wave w, lane l take granule indices w * 64 + l + 512 j.  U loads in
// flight per lane; a wave repeats ITS loads until all of its tags match.  Returns false on abort.
struct Range {
    const gu64 *g;
    int n;
};
template <int U>
__device__ __forceinline__ bool sweep(const Range *r, int nr, unsigned tag, unsigned *vec, unsigned char *plane0, unsigned char *plane1, int wave, int lane,
                                      unsigned *abort_word) {
    int total = 0;
    for (int i = 0; i < nr; ++i) total += r[i].n;
    const int idx0 = wave * 64 + lane;
    // granule index -> address: ranges are contiguous granule arrays (dead lanes poll the vector's first word and ignore it)
    auto addr = [&](int idx) -> const gu64 * {
        if (idx >= total) return r[0].g;
        if (nr == 1) return r[0].g + idx;
        return idx < r[0].n ? r[0].g + idx : idx < r[0].n + r[1].n ? r[1].g + (idx - r[0].n) : r[2].g + (idx - r[0].n - r[1].n);
    };
    Spin sp{__builtin_amdgcn_s_memrealtime(), abort_word};
    for (unsigned it = 0;; ++it) {
        u64 x[U];
#pragma unroll
        for (int j = 0; j < U; ++j) x[j] = __hip_atomic_load(addr(idx0 + SW * j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
#pragma unroll
        for (int j = 0; j < U; ++j) ok &= idx0 + SW * j >= total || (unsigned)(x[j] >> 32) == tag;
        if (__all(ok)) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int idx = idx0 + SW * j;
                if (idx < total) {
                    if (vec) vec[idx] = (unsigned)x[j];  // the attention phase's inputs: whole words
                    else plane0[idx] = (unsigned char)x[j], plane1[idx] = (unsigned char)(x[j] >> 8);  // a matrix input: its two digit planes
                }
            }
            return true;
        }
        if (sp.expired(it)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

__global__ __launch_bounds__(NT) void k_layer(Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *ring = lds;
    unsigned *vec = reinterpret_cast<unsigned *>(lds + kRegEnd * 1024);
    unsigned char *plane0 = lds + kRegEnd * 1024 + kVecBytes, *plane1 = plane0 + kPlane;
    unsigned char *zero = plane1 + kPlane;
    int *part = reinterpret_cast<int *>(zero + 1024);  // [wave][4 tiles][16]
    unsigned *flag = reinterpret_cast<unsigned *>(part + NCW * 4 * 16);
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave == NCW;
    for (int i = tid; i < 256; i += NT) reinterpret_cast<unsigned *>(zero)[i] = 0u;
    if (tid == 0) flag[0] = 0u;
    __syncthreads();
    const unsigned ring_lds = (unsigned)(uintptr_t)ring;  // LDS byte address of the ring (low 32 bits of the generic pointer)

    // this workgroup's work per phase
    const int n_qkv = b < 240 ? 1 : 0, n_o = b < 160 ? 1 : 0, n_gu = b < 216 ? 2 : 0, n_dn = b < 160 ? 1 : 0;
    const bool has_att = b < NREC;
    const int reg[NPH] = {kRegQkv, 0, kRegO, kRegGu, kRegDn};
    const int pieces[NPH] = {n_qkv * 10, 0, n_o * 10, n_gu * 2 * 10, n_dn * 27};
    const bool work[NPH] = {n_qkv > 0, has_att, n_o > 0, n_gu > 0, n_dn > 0};
    // byte offset of this workgroup's pieces inside a layer: regions in order, each [workgroup][pieces * 1.25][1 KiB]
    size_t goff[NPH];
    {
        size_t o = 0;
        const int per[NPH] = {ld125(kPiecesQkv), 0, ld125(kPiecesO), ld125(kPiecesGu), ld125(kPiecesDn)};
        for (int ph = 0; ph < NPH; ++ph) {
            goff[ph] = o + (size_t)b * per[ph] * 1024;
            o += (size_t)G * per[ph] * 1024;
        }
    }
    auto issue = [&](int layer, int ph) {  // loader: this workgroup's pieces of (layer, ph) -> its LDS region
        const int np = ld125(pieces[ph]);
        const unsigned char *src = p.w + (size_t)layer * p.layer_stride + goff[ph] + lane * 16;
        for (int i = 0; i < np; ++i) dma1k(src + (size_t)i * 1024, __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(reg[ph] + i) * 1024u));
    };
    if (loader && p.mode == 0) {
        for (int ph = 0; ph < NPH; ++ph)
            if (work[ph] && pieces[ph]) issue(0, ph);
    }

    for (int iter = 0; iter < p.n_iter; ++iter) {
        for (int layer = 0; layer < p.n_layers; ++layer) {
            const bool stamp = p.stamps && iter == p.n_iter - 1;
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                if (!work[ph]) continue;  // uniform over the workgroup: its 9 waves skip the phase's barriers together
                const unsigned tag_in = (unsigned)((iter * p.n_layers + layer) * NPH + ph);      // written by the previous phase (0: the host's x)
                const unsigned tag_out = tag_in + 1u;
                u64 t0 = 0, t1 = 0, t2 = 0;
                bool ok = true;
                if (!loader) {
                    if (stamp && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
                    // ---- edge: the input vector -> LDS ------------------------------------------------------------------
                    Range r[3];
                    int nr = 1;
                    if (ph == 0) r[0] = {p.gx, NX};
                    else if (ph == 1) {
                        const int kvh = b / NCH;
                        r[0] = {p.gqkv + kvh * 512, 512};
                        r[1] = {p.gqkv + NHEAD * 128 + kvh * 128, 128};
                        r[2] = {p.gqkv + (NHEAD + NKV) * 128 + kvh * 128, 128};
                        nr = 3;
                    } else if (ph == 2) r[0] = {p.grec, NRECV};
                    else if (ph == 3) r[0] = {p.gx2, NX};
                    else r[0] = {p.gh, NH};
                    const unsigned tin = (ph == 0 && layer == 0) ? (unsigned)(iter * p.n_layers * NPH) : tag_in;
                    if (ph == 0 || ph == 3) ok = sweep<(NX + SW - 1) / SW>(r, nr, tin, nullptr, plane0, plane1, wave, lane, p.abort_word);
                    else if (ph == 1) ok = sweep<(768 + SW - 1) / SW>(r, nr, tin, vec, plane0, plane1, wave, lane, p.abort_word);
                    else if (ph == 2) ok = sweep<(NRECV + SW - 1) / SW>(r, nr, tin, nullptr, plane0, plane1, wave, lane, p.abort_word);
                    else ok = sweep<(NH + SW - 1) / SW>(r, nr, tin, nullptr, plane0, plane1, wave, lane, p.abort_word);
                    if (!ok) flag[0] = 1u;
                } else if (p.mode == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every fill issued so far has landed (this phase's: a layer ago)
                }
                __syncthreads();  // #1: vector and weights are in LDS
                if (flag[0]) return;
                if (!loader && stamp && tid == 0) t1 = __builtin_amdgcn_s_memrealtime();
                if (!loader) {
                    if (ph == 1) {
                        // "attention": 64 KB of K/V for (kv head, chunk) + the 768 inputs -> 528 record values
                        const v4u *kvp = reinterpret_cast<const v4u *>(p.kv + ((size_t)layer * NREC + b) * 65536);
                        unsigned s = 0;
#pragma unroll
                        for (int i = 0; i < 10; ++i) {  // 4096 x 16 B over 448 threads: every load in flight at once
                            const int e = tid + SW * i;
                            const v4u q = __builtin_nontemporal_load(kvp + (e < 4096 ? e : 4095));
                            s += e < 4096 ? q[0] ^ q[1] ^ q[2] ^ q[3] : 0u;
                        }
                        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor((int)s, o);
                        if (lane == 0) part[wave] = (int)s;
                    } else {
                        const int nblk = ph == 4 ? 27 : 10;
                        const int nt = ph == 3 ? n_gu * 2 : 1;
                        // unit (tile lt, block kb) -> wave (lt * nblk + kb) % 8: K split over the waves
                        for (int lt = 0; lt < nt; ++lt) {
                            const int first = (lt * nblk) % NCW;
                            const int kb0 = (wave - first + NCW) % NCW;
                            const unsigned char *wl = ring + (size_t)(reg[ph] + lt * nblk) * 1024 + lane * 16;
                            const unsigned char *wg = p.w + (size_t)layer * p.layer_stride + goff[ph] + (size_t)lt * nblk * 1024 + lane * 16;
                            v4i acc = {0, 0, 0, 0};
                            const int g = lane >> 4, c = lane & 15;
                            for (int kb = kb0; kb < nblk; kb += NCW) {
                                v4u wv = p.mode == 0 ? *reinterpret_cast<const v4u *>(wl + (size_t)kb * 1024)
                                                     : __builtin_nontemporal_load(reinterpret_cast<const v4u *>(wg + (size_t)kb * 1024));
                                const unsigned wd[4] = {wv[0], wv[1], wv[2], wv[3]};
#pragma unroll
                                for (int m = 0; m < 4; ++m) {
                                    // A row c: c == 0 digit 0, c == 1 digit 1 of columns k0 .. k0 + 15, k0 = kb * 256 + m * 64 + g * 16
                                    const int k0 = kb * 256 + m * 64 + g * 16;
                                    const unsigned char *ap = c == 0 ? plane0 + k0 : c == 1 ? plane1 + k0 : zero;  // dead rows read zeros (as k_gemv_q does)
                                    const v4i a = *reinterpret_cast<const v4i *>(ap);
                                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, decode16(wd[m]), acc, 0, 0, 0);
                                }
                            }
                            if (g == 0) part[(wave * 4 + lt) * 16 + c] = (int)((unsigned)acc[0] + ((unsigned)acc[1] << 8));
                        }
                    }
                }
                __syncthreads();  // #2: partial sums are in LDS; nobody reads this phase's weights any more
                if (loader) {
                    if (p.mode == 0 && pieces[ph] && (layer + 1 < p.n_layers || iter + 1 < p.n_iter)) issue((layer + 1) % p.n_layers, ph);
                    continue;
                }
                if (stamp && tid == 0) t2 = __builtin_amdgcn_s_memrealtime();
                // ---- publish: 16 lanes per output tile ------------------------------------------------------------------
                if (ph == 1) {
                    unsigned s = 0;
                    for (int w = 0; w < NCW; ++w) s += (unsigned)part[w];
                    gu64 *dst = p.grec + (size_t)b * RECV;
                    for (int j = tid; j < RECV; j += SW) {
                        const unsigned v = vec[j % 768] * 3u + vec[(j * 5 + 1) % 768] + s;
                        __hip_atomic_store(dst + j, ((u64)tag_out << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else {
                    const int n_out = ph == 3 ? n_gu : 1;  // output tiles of this workgroup
                    if (tid < 16 * n_out) {
                        const int ot = tid >> 4, c = tid & 15;
                        int v;
                        if (ph == 3) {
                            int a = 0, u = 0;
                            for (int w = 0; w < NCW; ++w) a += part[(w * 4 + 2 * ot) * 16 + c], u += part[(w * 4 + 2 * ot + 1) * 16 + c];
                            v = a + 3 * u;
                        } else {
                            v = 0;
                            for (int w = 0; w < NCW; ++w) v += part[(w * 4) * 16 + c];
                        }
                        gu64 *dst;
                        int idx;
                        if (ph == 0) dst = p.gqkv, idx = b * 16 + c;
                        else if (ph == 2) dst = p.gx2, idx = b * 16 + c;
                        else if (ph == 3) dst = p.gh, idx = (b * 2 + ot) * 16 + c;
                        else dst = p.gx, idx = b * 16 + c;
                        __hip_atomic_store(dst + idx, ((u64)tag_out << 32) | (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ph == 4 && layer == p.n_layers - 1 && iter == p.n_iter - 1) p.out[idx] = (unsigned)v;
                    }
                }
                if (stamp && tid == 0) {
                    u64 *st = p.stamps + (((size_t)layer * NPH + ph) * G + b) * 4;
                    st[0] = t0, st[1] = t1, st[2] = t2, st[3] = __builtin_amdgcn_s_memrealtime();
                }
            }
        }
    }
}

// ---- host reference of the same integer chain (one iteration) --------------------------------------------------------------
static inline int lutv(unsigned c) { return c == 1 ? 1 : c == 3 ? -1 : 0; }
// piece (1 KiB) of a tile: lane l = (g = l >> 4, row c = l & 15), 16 bytes = dwords m = 0..3; dword m byte j field i (bits 2i):
// decode16: a[i] byte j = lut((w >> (2 i + 8 j)) & 3) -> B operand k index within the MFMA = g * 16 + 4 i + j ... of MFMA m
static void host_tile(const unsigned char *pieces, int nblk, const std::vector<unsigned> &vec, int out[16]) {
    long long acc0[16] = {0}, acc1[16] = {0};
    for (int kb = 0; kb < nblk; ++kb)
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, c = l & 15;
            const unsigned char *pb = pieces + (size_t)kb * 1024 + l * 16;
            for (int m = 0; m < 4; ++m) {
                unsigned w;
                memcpy(&w, pb + 4 * m, 4);
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        const int wv = lutv((w >> (2 * i + 8 * j)) & 3u);
                        const int k = kb * 256 + m * 64 + g * 16 + 4 * i + j;
                        acc0[c] += wv * (int)(signed char)(vec[k] & 0xff);
                        acc1[c] += wv * (int)(signed char)((vec[k] >> 8) & 0xff);
                    }
            }
        }
    for (int c = 0; c < 16; ++c) out[c] = (int)((unsigned)acc0[c] + ((unsigned)acc1[c] << 8));
}

int main(int argc, char **argv) {
    const int n_layers = argc > 1 ? atoi(argv[1]) : 30, n_iter = argc > 2 ? atoi(argv[2]) : 8, check_layers = argc > 3 ? atoi(argv[3]) : 2;
    const int per[NPH] = {ld125(kPiecesQkv), 0, ld125(kPiecesO), ld125(kPiecesGu), ld125(kPiecesDn)};
    size_t layer_bytes = 0;
    for (int ph = 0; ph < NPH; ++ph) layer_bytes += (size_t)G * per[ph] * 1024;
    printf("layer weight bytes (incl. 25 %% scale bytes, padded to 256 workgroups): %.2f MB; LDS per workgroup %d B\n", layer_bytes / 1e6, kLdsBytes);
    std::vector<unsigned char> hw(layer_bytes * n_layers);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < hw.size(); i += 8) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        memcpy(&hw[i], &s, 8);
    }
    std::vector<unsigned char> hkv((size_t)n_layers * NREC * 65536);
    for (size_t i = 0; i < hkv.size(); i += 8) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        memcpy(&hkv[i], &s, 8);
    }
    unsigned char *dw, *dkv;
    CHK(hipMalloc(&dw, hw.size()));
    CHK(hipMemcpy(dw, hw.data(), hw.size(), hipMemcpyHostToDevice));
    CHK(hipMalloc(&dkv, hkv.size()));
    CHK(hipMemcpy(dkv, hkv.data(), hkv.size(), hipMemcpyHostToDevice));
    u64 *gx, *gqkv, *grec, *gx2, *gh, *stamps;
    unsigned *abort_word, *out;
    CHK(hipMalloc(&gx, NX * 8));
    CHK(hipMalloc(&gqkv, NQKV * 8));
    CHK(hipMalloc(&grec, NRECV * 8));
    CHK(hipMalloc(&gx2, NX * 8));
    CHK(hipMalloc(&gh, NH * 8));
    CHK(hipMalloc(&abort_word, 256));
    CHK(hipMalloc(&out, NX * 4));
    const size_t stamp_words = (size_t)n_layers * NPH * G * 4;
    CHK(hipMalloc(&stamps, stamp_words * 8));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_layer), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    std::vector<unsigned> x0(NX);
    for (int i = 0; i < NX; ++i) x0[i] = (unsigned)(i * 2654435761u) >> 7;

    auto run = [&](int layers, int iters, int mode, bool print) -> std::vector<unsigned> {
        CHK(hipMemset(gqkv, 0, NQKV * 8));
        CHK(hipMemset(grec, 0, NRECV * 8));
        CHK(hipMemset(gx2, 0, NX * 8));
        CHK(hipMemset(gh, 0, NH * 8));
        CHK(hipMemset(abort_word, 0, 256));
        CHK(hipMemset(stamps, 0, stamp_words * 8));
        std::vector<u64> hx(NX);
        for (int i = 0; i < NX; ++i) hx[i] = x0[i];  // tag 0: the first phase of iteration 0 expects tag 0 ... see kernel (tin)
        CHK(hipMemcpy(gx, hx.data(), NX * 8, hipMemcpyHostToDevice));
        Params p;
        p.w = dw, p.layer_stride = layer_bytes, p.kv = dkv;
        p.gx = (gu64 *)gx, p.gqkv = (gu64 *)gqkv, p.grec = (gu64 *)grec, p.gx2 = (gu64 *)gx2, p.gh = (gu64 *)gh;
        p.n_layers = layers, p.n_iter = iters, p.mode = mode, p.abort_word = abort_word, p.stamps = stamps, p.out = out;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_layer, dim3(G), dim3(NT), kLdsBytes, 0, p);
        CHK(hipGetLastError());
        CHK(hipEventRecord(e1, 0));
        CHK(hipDeviceSynchronize());
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned ab = 0;
        CHK(hipMemcpy(&ab, abort_word, 4, hipMemcpyDeviceToHost));
        std::vector<unsigned> res(NX);
        CHK(hipMemcpy(res.data(), out, NX * 4, hipMemcpyDeviceToHost));
        if (print) {
            printf("mode %d: %d layers x %d iterations: %.1f us total, %.2f us per layer%s\n", mode, layers, iters, ms * 1e3, ms * 1e3 / (layers * iters),
                   ab ? "  ** ABORTED (a spin timed out) **" : "");
            std::vector<u64> hs(stamp_words);
            CHK(hipMemcpy(hs.data(), stamps, stamp_words * 8, hipMemcpyDeviceToHost));
            const char *names[NPH] = {"qkv", "att", "o", "gate|up", "down"};
            const int nwg[NPH] = {240, NREC, 160, 216, 160};
            double tot = 0;
            for (int ph = 0; ph < NPH; ++ph) {
                std::vector<double> edge, comp, pub;
                for (int l = layers / 2; l < layers; ++l)
                    for (int b = 0; b < nwg[ph]; ++b) {
                        const u64 *st = &hs[(((size_t)l * NPH + ph) * G + b) * 4];
                        if (!st[3]) continue;
                        edge.push_back((double)(st[1] - st[0]) * 10.0);
                        comp.push_back((double)(st[2] - st[1]) * 10.0);
                        pub.push_back((double)(st[3] - st[2]) * 10.0);
                    }
                auto med = [](std::vector<double> &v) {
                    if (v.empty()) return 0.0;
                    std::sort(v.begin(), v.end());
                    return v[v.size() / 2];
                };
                const double e = med(edge), c = med(comp), q = med(pub);
                tot += e + c + q;
                printf("    %-8s edge wait %6.0f ns   compute %6.0f ns   publish %5.0f ns\n", names[ph], e, c, q);
            }
            printf("    sum of medians %.2f us per layer\n", tot / 1e3);
            fflush(stdout);
        }
        return ab ? std::vector<unsigned>() : res;
    };

    // ---- correctness: a short chain against the serial host model ----------------------------------------------------------
    {
        std::vector<unsigned> x = x0, qkv(NQKV), rec(NRECV), x2(NX), h(NH);
        for (int l = 0; l < check_layers; ++l) {
            const unsigned char *lw = hw.data() + (size_t)l * layer_bytes;
            size_t off = 0;
            int o16[16];
            for (int t = 0; t < 240; ++t) {
                host_tile(lw + off + (size_t)t * per[0] * 1024, 10, x, o16);
                for (int c = 0; c < 16; ++c) qkv[t * 16 + c] = (unsigned)o16[c];
            }
            off += (size_t)G * per[0] * 1024;
            for (int b = 0; b < NREC; ++b) {
                const int kvh = b / NCH;
                std::vector<unsigned> in(768);
                for (int i = 0; i < 512; ++i) in[i] = qkv[kvh * 512 + i];
                for (int i = 0; i < 128; ++i) in[512 + i] = qkv[NHEAD * 128 + kvh * 128 + i], in[640 + i] = qkv[(NHEAD + NKV) * 128 + kvh * 128 + i];
                const unsigned *kvw = reinterpret_cast<const unsigned *>(hkv.data() + ((size_t)l * NREC + b) * 65536);
                unsigned sum = 0;
                for (int e = 0; e < 4096; ++e) {
                    const unsigned *q = kvw + (size_t)e * 4;
                    sum += q[0] ^ q[1] ^ q[2] ^ q[3];
                }
                for (int j = 0; j < RECV; ++j) rec[b * RECV + j] = in[j % 768] * 3u + in[(j * 5 + 1) % 768] + sum;
            }
            for (int t = 0; t < 160; ++t) {
                host_tile(lw + off + (size_t)t * per[2] * 1024, 10, rec, o16);
                for (int c = 0; c < 16; ++c) x2[t * 16 + c] = (unsigned)o16[c];
            }
            off += (size_t)G * per[2] * 1024;
            for (int b = 0; b < 216; ++b)
                for (int ot = 0; ot < 2; ++ot) {
                    int a16[16], u16[16];
                    host_tile(lw + off + ((size_t)b * per[3] + (size_t)(2 * ot) * 10) * 1024, 10, x2, a16);
                    host_tile(lw + off + ((size_t)b * per[3] + (size_t)(2 * ot + 1) * 10) * 1024, 10, x2, u16);
                    for (int c = 0; c < 16; ++c) h[(b * 2 + ot) * 16 + c] = (unsigned)(a16[c] + 3 * u16[c]);
                }
            off += (size_t)G * per[3] * 1024;
            for (int t = 0; t < 160; ++t) {
                host_tile(lw + off + (size_t)t * per[4] * 1024, 27, h, o16);
                for (int c = 0; c < 16; ++c) x[t * 16 + c] = (unsigned)o16[c];
            }
        }
        for (int mode = 0; mode < 2; ++mode) {
            const std::vector<unsigned> got = run(check_layers, 1, mode, false);
            size_t bad = 0;
            for (int i = 0; i < NX && !got.empty(); ++i) bad += got[i] != x[i];
            printf("check mode %d (%d layers vs serial host model): %s (%zu of %d words differ)\n", mode, check_layers,
                   got.empty() ? "ABORTED" : bad ? "MISMATCH" : "exact", bad, NX);
            if (got.empty() || bad) return 1;
        }
    }
    // ---- timing --------------------------------------------------------------------------------------------------------------
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 2; ++mode) run(n_layers, n_iter, mode, rep == 1);
    return 0;
}
