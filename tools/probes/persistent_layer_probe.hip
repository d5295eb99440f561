// persistent_layer_probe.hip -- VERDICT r02 item 3(ii): ONE persistent launch for the batch-1 decode chain, built the way
// MI355X_MICROARCH.md's "engine-vs-launches" row and cdna_hip_programming.md Guideline 16 prescribe, at the byte sizes of a
// bitnet-b1.58-2B-4T layer (BitNet32-F16: 1 KiB of codes + 256 B of scales per 16-row x 256-column tile), 3 chunks of 64 keys.
//
//   * 256 workgroups of 8 waves, one per CU.  Workgroups 0..239 are MATRIX workgroups (waves 0..6 compute, wave 7 is the
//     LOADER), 240..254 ATTENTION workgroups (one per KV head and 64-key chunk; they own no weights), 255 idles.
//   * A matrix workgroup's loader keeps a whole layer of its weight share in LDS (qkv 13 KiB, o 13, gate|up 50, down 34 =
//     110 KiB): the moment the compute waves have released region p of layer l (an LDS counter) it refills it with layer l + 1
//     by LDS-DMA (global_load_lds_dwordx4, non-temporal) -- one layer of run-ahead, so a phase's weights are in LDS long before
//     its activation vector arrives.  An attention workgroup prefetches the NEXT layer's 64 KB K/V chunk the same way.
//   * Hand-offs are data-tagged 8-byte granules {tag = stage epoch, value} (Guideline 16 R2): producers store them write-through
//     (sc1), consumers sweep them with sc1 loads until every tag matches -- no flag, no fence, no agent-scope acquire.
//     Inside a matrix workgroup there is NO s_barrier: wave w sweeps the input blocks kb = w (mod 7) into LDS digit planes and
//     raises an LDS word, a wave waits only for the slices its own units read, partial sums meet through LDS and the LAST
//     wave to arrive (an LDS counter) publishes -- so the loader is never part of a rendezvous.
//     (Version 1 of this probe had the loader in two s_barriers per stage and the o-projection sweeping all 15 chunk records:
//     23.1 us per layer -- the loader's DMA issue, 50 pieces for gate|up, sat between the barriers, and the record sweep was 63 KB
//     per workgroup.)
//   * Stages per layer: q|k|v (240 row tiles) -> attention chunk records (15 workgroups) -> merge of a KV head's records by its
//     chunk-0 workgroup -> o-projection (160 tiles) -> gate|up (432 tile pairs) -> down (160 tiles, K = 6912).
//
// The arithmetic is the real kernel's per 1-KiB tile (2-bit code expansion by v_perm_b32, 4 x v_mfma_i32_16x16x64_i8 with the
// activation digits as the A operand from LDS) on synthetic data, INTEGER end to end, so the result of the whole chain is
// independent of the partition and of timing: the host recomputes it serially and compares every word (a stale or torn
// hand-off cannot pass).  Every spin is bounded by s_memrealtime (2 ms) and raises a global abort word that every other spin
// watches: the probe cannot hang the GPU.
//
// mode 0: as above.  mode 1: no loader -- weights / K/V fetched from global memory when the stage starts (what a persistent
// kernel WITHOUT run-ahead staging would do).  Output: per-stage medians over workgroups (wave 0's view) and the layer period.
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
constexpr int NCH = 3, NREC = NKV * NCH, RECV = 528;                                    // 4 x 128 + 8 values per chunk record
constexpr int G = 256, NMAT = 240, NCW = 7, NT = 512;
constexpr int NST = 6;  // stages: 0 qkv, 1 attention chunk records, 2 record merge, 3 o, 4 gate|up, 5 down
constexpr unsigned kLut = 0xff000100u;  // ternary {0, +1, 0, -1}: byte c = value of code c

// weight pieces (1 KiB) per matrix workgroup and stage, codes only; the loader fetches 1.25 x as many (the f16 block scales)
__host__ __device__ constexpr int ld125(int p) { return (p * 5 + 3) / 4; }
__host__ __device__ constexpr int kPcOf(int r) { return r == 0 ? 10 : r == 1 ? 10 : r == 2 ? 40 : 27; }   // qkv, o, gate|up (2 pairs), down
__host__ __device__ constexpr int kRegOf(int r) { return r == 0 ? 0 : r == 1 ? 13 : r == 2 ? 26 : r == 3 ? 76 : 110; }  // region starts, KiB
constexpr int kPlane = 6912;                                                    // digits of the largest matrix input (down)
constexpr int kOffPlanes = kRegOf(4) * 1024;                                    // [parity][digit][kPlane]
constexpr int kOffZero = kOffPlanes + 4 * kPlane;
constexpr int kOffPart = kOffZero + 1024;                                       // [parity][wave][4 tiles][16] int
constexpr int kOffCtl = kOffPart + 2 * NCW * 4 * 16 * 4;                        // control words
constexpr int kLdsMat = kOffCtl + 256;
// attention workgroups: K/V chunk [2][64 KiB], inputs u32[768], own record u32[528], sibling records u32[2 * 528], reduction words
constexpr int kOffAttIn = 2 * 65536, kOffAttOwn = kOffAttIn + 768 * 4, kOffAttSib = kOffAttOwn + RECV * 4, kOffAttRed = kOffAttSib + (NCH - 1) * RECV * 4,
              kLdsAtt = kOffAttRed + 64;
constexpr int kLdsBytes = kLdsMat > kLdsAtt ? kLdsMat : kLdsAtt;
static_assert(kLdsBytes <= 160 * 1024, "LDS");

struct Ctl {  // LDS control words of a matrix workgroup
    unsigned cnt[2];         // arrivals of the compute waves at the end of a stage (by stage parity)
    unsigned swept[2][8];    // tag + 1 of the input slice wave w has swept (by stage parity)
    unsigned released[4];    // compute waves done with weight region r, running total
    unsigned landed[4];      // global layer index + 1 whose weights are in region r
};

struct Params {
    const unsigned char *w;  // [layer][region][matrix workgroup][pieces * 1.25][1 KiB]
    size_t layer_stride;
    const unsigned char *kv; // [layer][NREC][64 KiB]
    gu64 *gx, *gqkv, *grec, *gatt, *gx2, *gh;
    int n_layers, n_iter, mode;
    unsigned *abort_word;
    u64 *stamps;  // [layer][stage][workgroup][4]  (last iteration)
    unsigned *out;  // final x [H]
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
__device__ __forceinline__ unsigned lds_ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct Spin {
    u64 t0;
    unsigned *abort_word;
    unsigned it = 0;
    __device__ __forceinline__ bool expired() {
        if ((++it & 31u) != 0u) return false;
        if (__hip_atomic_load((gu32 *)abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000ull) {  // 2 ms at 100 MHz
            __hip_atomic_store((gu32 *)abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        return false;
    }
};
// wait until an LDS word reaches a value (another wave of this workgroup writes it)
__device__ __forceinline__ bool lds_wait_ge(const unsigned *p, unsigned v, unsigned *abort_word) {
    if (lds_ld(p) >= v) return true;
    Spin sp{__builtin_amdgcn_s_memrealtime(), abort_word};
    while (lds_ld(p) < v) {
        if (sp.expired()) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

// Wave w sweeps the 256-granule BLOCKS kb = w, w + 7, ... < nblk_in of a contiguous granule vector into the digit planes:
// 4 loads per lane and block, all in flight; repeated until every tag matches.
template <int MAXB>
__device__ __forceinline__ bool sweep_blocks(const gu64 *g, int nblk_in, int w, unsigned tag, unsigned char *plane0, unsigned char *plane1, int lane,
                                             unsigned *abort_word) {
    Spin sp{__builtin_amdgcn_s_memrealtime(), abort_word};
    for (;;) {
        u64 x[MAXB][4];
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            const int kb = w + NCW * i < nblk_in ? w + NCW * i : w;  // past the end: re-poll the wave's first block (ignored)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[i][j] = __hip_atomic_load(g + kb * 256 + 64 * j + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < MAXB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) ok &= w + NCW * i >= nblk_in || (unsigned)(x[i][j] >> 32) == tag;
        if (__all(ok)) {
#pragma unroll
            for (int i = 0; i < MAXB; ++i)
                if (w + NCW * i < nblk_in) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int idx = (w + NCW * i) * 256 + 64 * j + lane;
                        plane0[idx] = (unsigned char)x[i][j];
                        plane1[idx] = (unsigned char)(x[i][j] >> 8);
                    }
                }
            return true;
        }
        if (sp.expired()) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// U granules per thread of a list of up to 3 contiguous ranges, whole words into vec[] (attention workgroups: all 8 waves)
struct Range {
    const gu64 *g;
    int n;
};
template <int U>
__device__ __forceinline__ bool sweep_words(const Range *r, int nr, unsigned tag, unsigned *vec, int tid, unsigned *abort_word) {
    int total = 0;
    for (int i = 0; i < nr; ++i) total += r[i].n;
    auto addr = [&](int idx) -> const gu64 * {
        if (idx >= total) return r[0].g;
        if (nr == 1) return r[0].g + idx;
        return idx < r[0].n ? r[0].g + idx : idx < r[0].n + r[1].n ? r[1].g + (idx - r[0].n) : r[2].g + (idx - r[0].n - r[1].n);
    };
    Spin sp{__builtin_amdgcn_s_memrealtime(), abort_word};
    for (;;) {
        u64 x[U];
#pragma unroll
        for (int j = 0; j < U; ++j) x[j] = __hip_atomic_load(addr(tid + NT * j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
#pragma unroll
        for (int j = 0; j < U; ++j) ok &= tid + NT * j >= total || (unsigned)(x[j] >> 32) == tag;
        if (__all(ok)) {
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (tid + NT * j < total) vec[tid + NT * j] = (unsigned)x[j];
            return true;
        }
        if (sp.expired()) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// ---- one matrix stage of a matrix workgroup's compute wave -------------------------------------------------------------
// R: region index (0 qkv, 1 o, 2 gate|up, 3 down); NTL: row tiles of this workgroup; NBLK: 256-column blocks of K
typedef __attribute__((address_space(3))) const unsigned char lds_cbyte;
template <int MODE, int R, int NTL, int NBLK>
__device__ __forceinline__ bool matrix_stage(const Params &p, unsigned char *lds, int b, int wave, int lane, int lg, int layer, int par, const gu64 *gin,
                                             unsigned tag_in, gu64 *gout, unsigned tag_out, bool stamp, int stage) {
    Ctl *ctl = reinterpret_cast<Ctl *>(lds + kOffCtl);
    unsigned char *plane0 = lds + kOffPlanes + (2 * par) * kPlane, *plane1 = plane0 + kPlane;
    const unsigned char *zero = lds + kOffZero;
    int *part = reinterpret_cast<int *>(lds + kOffPart) + par * NCW * 4 * 16;
    u64 t0 = 0, t1 = 0, t2 = 0;
    if (stamp && wave == 0 && lane == 0) t0 = __builtin_amdgcn_s_memrealtime();
    // ---- edge: this wave's slice of the input vector -> digit planes, then its LDS word ------------------------------------
    constexpr int MAXB = (NBLK + NCW - 1) / NCW;
    if (!sweep_blocks<MAXB>(gin, NBLK, wave, tag_in, plane0, plane1, lane, p.abort_word)) return false;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the plane bytes are in LDS before the word says so
    if (lane == 0) lds_st(&ctl->swept[par][wave], tag_in + 1u);
    if (stamp && wave == 0 && lane == 0) t1 = __builtin_amdgcn_s_memrealtime();
    // ---- this wave's units, block-major: u = kb * NTL + lt ---------------------------------------------------------------------
    constexpr int NU = NTL * NBLK;
    const int u0 = wave * NU / NCW, u1 = (wave + 1) * NU / NCW;
    if (MODE == 0 && !lds_wait_ge(&ctl->landed[R], (unsigned)lg + 1u, p.abort_word)) return false;
    const int g = lane >> 4, c = lane & 15;
    v4i acc[NTL];
#pragma unroll
    for (int i = 0; i < NTL; ++i) acc[i] = (v4i){0, 0, 0, 0};
    lds_cbyte *wl = (lds_cbyte *)(lds + (size_t)kRegOf(R) * 1024 + lane * 16);
    size_t goff = 0;
    for (int r = 0; r < R; ++r) goff += (size_t)NMAT * ld125(kPcOf(r)) * 1024;
    const unsigned char *wg = p.w + (size_t)layer * p.layer_stride + goff + (size_t)b * ld125(kPcOf(R)) * 1024 + lane * 16;
    // every weight piece of this wave's units first (they are in LDS / requested before anything waits), as k_gemv_q does
    constexpr int MAXU = (NU + NCW - 1) / NCW;
    v4u wv[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = u0 + i < u1 ? u0 + i : u0;
        const int kb = u / NTL, lt = u % NTL;
        if (MODE == 0) wv[i] = *reinterpret_cast<__attribute__((address_space(3))) const v4u *>(wl + (size_t)(lt * NBLK + kb) * 1024);
        else wv[i] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(wg + (size_t)(lt * NBLK + kb) * 1024));
    }
    // A-operand lane: row c = 0 reads digit plane 0, c = 1 plane 1, every other row the zero area (no masking instructions)
    lds_cbyte *abase = (lds_cbyte *)(c == 0 ? plane0 : c == 1 ? plane1 : zero);
    const int astep = c < 2 ? 1 : 0;
    int kb_ready = -1;
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = u0 + i;
        if (u >= u1) break;
        const int kb = u / NTL, lt = u % NTL;
        if (kb != kb_ready) {  // the slice that holds block kb was swept by wave kb % 7
            if (!lds_wait_ge(&ctl->swept[par][kb % NCW], tag_in + 1u, p.abort_word)) return false;
            kb_ready = kb;
        }
        v4i av[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<__attribute__((address_space(3))) const v4i *>(abase + astep * (kb * 256 + m * 64 + g * 16));
        const unsigned wd[4] = {wv[i][0], wv[i][1], wv[i][2], wv[i][3]};
        v4i a4 = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m) a4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[m], decode16(wd[m]), a4, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NTL; ++q)
            if (q == lt) acc[q] += a4;
    }
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < NTL; ++i) part[(wave * 4 + i) * 16 + c] = (int)((unsigned)acc[i][0] + ((unsigned)acc[i][1] << 8));
    }
    unsigned old = 0;
    if (lane == 0) {
        __hip_atomic_fetch_add(&ctl->released[R], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // the loader may refill the region
        old = __hip_atomic_fetch_add(&ctl->cnt[par], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    old = __builtin_amdgcn_readfirstlane(old);
    if (stamp && wave == 0 && lane == 0) t2 = __builtin_amdgcn_s_memrealtime();
    u64 t3 = 0;
    if (old == NCW - 1) {  // the last wave to arrive publishes (every other wave's partial sums were written before its add)
        if (lane == 0) lds_st(&ctl->cnt[par], 0u);  // nobody touches the counter again before this stage's output is out
        constexpr int NOUT = R == 2 ? NTL / 2 : NTL;
        if (lane < 16 * NOUT) {
            const int ot = lane >> 4;
            int v;
            if (R == 2) {
                int a = 0, uu = 0;
                for (int w = 0; w < NCW; ++w) a += part[(w * 4 + 2 * ot) * 16 + c], uu += part[(w * 4 + 2 * ot + 1) * 16 + c];
                v = a + 3 * uu;
            } else {
                v = 0;
                for (int w = 0; w < NCW; ++w) v += part[(w * 4 + ot) * 16 + c];
            }
            const int idx = (b * NOUT + ot) * 16 + c;
            __hip_atomic_store(gout + idx, ((u64)tag_out << 32) | (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (R == 3 && lg == p.n_layers * p.n_iter - 1) p.out[idx] = (unsigned)v;
        }
        if (stamp && lane == 0) t3 = __builtin_amdgcn_s_memrealtime();
    }
    if (stamp && lane == 0) {
        u64 *st = p.stamps + (((size_t)layer * NST + stage) * G + b) * 4;
        if (wave == 0) st[0] = t0, st[1] = t1, st[2] = t2;
        if (t3) st[3] = t3;
    }
    return true;
}

template <int MODE>
__global__ __launch_bounds__(NT) void k_layer(Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total_layers = p.n_layers * p.n_iter;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;  // LDS byte address (low 32 bits of the generic pointer)
    if (b >= NMAT + NREC) return;
    if (b < NMAT) {
        // ================================================= matrix workgroup =================================================
        Ctl *ctl = reinterpret_cast<Ctl *>(lds + kOffCtl);
        for (int i = tid; i < 256; i += NT) reinterpret_cast<unsigned *>(lds + kOffZero)[i] = 0u;
        for (int i = tid; i < 64; i += NT) reinterpret_cast<unsigned *>(ctl)[i] = 0u;
        __syncthreads();
        const bool has1 = b < 160, has2 = b < 216, has3 = b < 160;
        if (wave == NCW) {
            // ---- loader: region r of global layer lg once the compute waves have released it for lg - 1 ----------------------
            if (MODE != 0) return;
            for (int lg = 0; lg < total_layers; ++lg) {
                size_t o = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t goff = o + (size_t)b * ld125(kPcOf(r)) * 1024;
                    o += (size_t)NMAT * ld125(kPcOf(r)) * 1024;
                    if ((r == 1 && !has1) || (r == 2 && !has2) || (r == 3 && !has3)) continue;
                    if (lg > 0 && !lds_wait_ge(&ctl->released[r], (unsigned)(NCW * lg), p.abort_word)) return;
                    const unsigned char *src = p.w + (size_t)(lg % p.n_layers) * p.layer_stride + goff + lane * 16;
                    const int np = ld125(kPcOf(r));
                    for (int i = 0; i < np; ++i) dma1k(src + (size_t)i * 1024, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(kRegOf(r) + i) * 1024u));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) lds_st(&ctl->landed[r], (unsigned)lg + 1u);
                }
            }
            return;
        }
        for (int lg = 0; lg < total_layers; ++lg) {
            const int layer = lg % p.n_layers;
            const bool stamp = p.stamps && lg >= total_layers - p.n_layers;
            const unsigned T0 = (unsigned)lg * NST;  // tag of this layer's input x (0: the host's)
            // stage parity: consecutive stages of THIS workgroup alternate plane / partial-sum / counter sets
            if (!matrix_stage<MODE, 0, 1, 10>(p, lds, b, wave, lane, lg, layer, 0, p.gx, T0, p.gqkv, T0 + 1, stamp, 0)) return;
            if (has1 && !matrix_stage<MODE, 1, 1, 10>(p, lds, b, wave, lane, lg, layer, 1, p.gatt, T0 + 3, p.gx2, T0 + 4, stamp, 3)) return;
            if (has2 && !matrix_stage<MODE, 2, 4, 10>(p, lds, b, wave, lane, lg, layer, 0, p.gx2, T0 + 4, p.gh, T0 + 5, stamp, 4)) return;
            if (has3 && !matrix_stage<MODE, 3, 1, 27>(p, lds, b, wave, lane, lg, layer, 1, p.gh, T0 + 5, p.gx, T0 + 6, stamp, 5)) return;
        }
        return;
    }
    // ===================================================== attention workgroup ==================================================
    const int rix = b - NMAT, kvh = rix / NCH, chunk = rix % NCH;
    unsigned *vin = reinterpret_cast<unsigned *>(lds + kOffAttIn), *own = reinterpret_cast<unsigned *>(lds + kOffAttOwn),
             *sib = reinterpret_cast<unsigned *>(lds + kOffAttSib), *red = reinterpret_cast<unsigned *>(lds + kOffAttRed);
    auto prefetch = [&](int lg) {  // the 64 KB chunk of global layer lg -> kv buffer lg & 1: 8 pieces per wave, fire and forget
        const unsigned char *src = p.kv + ((size_t)(lg % p.n_layers) * NREC + rix) * 65536 + (size_t)wave * 8192 + lane * 16;
        for (int i = 0; i < 8; ++i) dma1k(src + (size_t)i * 1024, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((lg & 1) * 65536 + wave * 8192 + i * 1024)));
    };
    if (MODE == 0) prefetch(0);
    for (int lg = 0; lg < total_layers; ++lg) {
        const int layer = lg % p.n_layers;
        const bool stamp = p.stamps && lg >= total_layers - p.n_layers;
        const unsigned T0 = (unsigned)lg * NST;
        u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (stamp && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
        Range r[3] = {{p.gqkv + kvh * 512, 512}, {p.gqkv + NHEAD * 128 + kvh * 128, 128}, {p.gqkv + (NHEAD + NKV) * 128 + kvh * 128, 128}};
        if (!sweep_words<2>(r, 3, T0 + 1, vin, tid, p.abort_word)) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the K/V chunk has landed (issued a layer ago)
        __syncthreads();
        if (stamp && tid == 0) t1 = __builtin_amdgcn_s_memrealtime();
        // "attention": the 64 KB chunk + the 768 inputs -> 528 record values
        unsigned s = 0;
        if (MODE == 0) {
            const v4u *kl = reinterpret_cast<const v4u *>(lds + (lg & 1) * 65536);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const v4u q = kl[tid + NT * i];
                s += q[0] ^ q[1] ^ q[2] ^ q[3];
            }
        } else {
            const v4u *kvp = reinterpret_cast<const v4u *>(p.kv + ((size_t)layer * NREC + rix) * 65536);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const v4u q = __builtin_nontemporal_load(kvp + tid + NT * i);
                s += q[0] ^ q[1] ^ q[2] ^ q[3];
            }
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor((int)s, o);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        s = 0;
        for (int w = 0; w < 8; ++w) s += red[w];
        for (int j = tid; j < RECV; j += NT) {
            const unsigned v = vin[j % 768] * 3u + vin[(j * 5 + 1) % 768] + s;
            if (chunk == 0) own[j] = v;
            else __hip_atomic_store(p.grec + (size_t)rix * RECV + j, ((u64)(T0 + 2) << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (stamp && tid == 0) t2 = __builtin_amdgcn_s_memrealtime();
        if (chunk != 0) {
            if (MODE == 0 && lg + 1 < total_layers) prefetch(lg + 1);
        } else {
            // ---- stage 2: merge the KV head's chunk records, publish the 4 heads' 512 outputs --------------------------------
            if (NCH > 1) {
                Range rs[1] = {{p.grec + (size_t)(rix + 1) * RECV, (NCH - 1) * RECV}};
                if (!sweep_words<((NCH - 1) * RECV + NT - 1) / NT>(rs, 1, T0 + 2, sib, tid, p.abort_word)) return;
            }
            __syncthreads();
            {
                unsigned v = own[tid] + own[512 + (tid & 15)];
                for (int cc = 0; cc < NCH - 1; ++cc) v = v * 5u + sib[cc * RECV + tid] + sib[cc * RECV + 512 + (tid & 15)];
                __hip_atomic_store(p.gatt + kvh * 512 + tid, ((u64)(T0 + 3) << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (stamp && tid == 0) t3 = __builtin_amdgcn_s_memrealtime();
            if (MODE == 0 && lg + 1 < total_layers) prefetch(lg + 1);
        }
        __syncthreads();  // own[] / sib[] / vin[] are rewritten by the next layer
        if (stamp && tid == 0) {
            u64 *st = p.stamps + (((size_t)layer * NST + 1) * G + b) * 4;
            st[0] = t0, st[1] = t1, st[2] = t2, st[3] = t3 ? t3 : t2;
        }
    }
}

// ---- host reference of the same integer chain ---------------------------------------------------------------------------------
static inline int lutv(unsigned c) { return c == 1 ? 1 : c == 3 ? -1 : 0; }
// piece (1 KiB) of a tile: lane l = (g = l >> 4, row c = l & 15), 16 bytes = dwords m = 0..3; field (i, j) of dword m (bits 2 i + 8 j)
// is the weight of column kb * 256 + m * 64 + g * 16 + 4 i + j (decode16 + the MFMA's B layout)
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
    const int per[4] = {ld125(kPcOf(0)), ld125(kPcOf(1)), ld125(kPcOf(2)), ld125(kPcOf(3))};
    size_t layer_bytes = 0;
    for (int r = 0; r < 4; ++r) layer_bytes += (size_t)NMAT * per[r] * 1024;
    printf("layer weight bytes (incl. 25 %% scale bytes, padded to %d workgroups): %.2f MB; LDS per workgroup %d B\n", NMAT, layer_bytes / 1e6, kLdsBytes);
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
    u64 *gx, *gqkv, *grec, *gatt, *gx2, *gh, *stamps;
    unsigned *abort_word, *out;
    CHK(hipMalloc(&gx, H * 8));
    CHK(hipMalloc(&gqkv, QKV * 8));
    CHK(hipMalloc(&grec, NREC * RECV * 8));
    CHK(hipMalloc(&gatt, H * 8));
    CHK(hipMalloc(&gx2, H * 8));
    CHK(hipMalloc(&gh, F * 8));
    CHK(hipMalloc(&abort_word, 256));
    CHK(hipMalloc(&out, H * 4));
    const size_t stamp_words = (size_t)n_layers * NST * G * 4;
    CHK(hipMalloc(&stamps, stamp_words * 8));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_layer<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_layer<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    std::vector<unsigned> x0(H);
    for (int i = 0; i < H; ++i) x0[i] = (unsigned)(i * 2654435761u) >> 7;

    auto run = [&](int layers, int iters, int mode, bool print) -> std::vector<unsigned> {
        CHK(hipMemset(gqkv, 0, QKV * 8));
        CHK(hipMemset(grec, 0, NREC * RECV * 8));
        CHK(hipMemset(gatt, 0, H * 8));
        CHK(hipMemset(gx2, 0, H * 8));
        CHK(hipMemset(gh, 0, F * 8));
        CHK(hipMemset(abort_word, 0, 256));
        CHK(hipMemset(stamps, 0, stamp_words * 8));
        std::vector<u64> hx(H);
        for (int i = 0; i < H; ++i) hx[i] = x0[i];  // tag 0
        CHK(hipMemcpy(gx, hx.data(), H * 8, hipMemcpyHostToDevice));
        Params p;
        p.w = dw, p.layer_stride = layer_bytes, p.kv = dkv;
        p.gx = (gu64 *)gx, p.gqkv = (gu64 *)gqkv, p.grec = (gu64 *)grec, p.gatt = (gu64 *)gatt, p.gx2 = (gu64 *)gx2, p.gh = (gu64 *)gh;
        p.n_layers = layers, p.n_iter = iters, p.mode = mode, p.abort_word = abort_word, p.stamps = stamps, p.out = out;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL(k_layer<0>, dim3(G), dim3(NT), kLdsBytes, 0, p);
        else hipLaunchKernelGGL(k_layer<1>, dim3(G), dim3(NT), kLdsBytes, 0, p);
        CHK(hipGetLastError());
        CHK(hipEventRecord(e1, 0));
        CHK(hipDeviceSynchronize());
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned ab = 0;
        CHK(hipMemcpy(&ab, abort_word, 4, hipMemcpyDeviceToHost));
        std::vector<unsigned> res(H);
        CHK(hipMemcpy(res.data(), out, H * 4, hipMemcpyDeviceToHost));
        if (print) {
            printf("mode %d: %d layers x %d iterations: %.1f us total, %.2f us per layer%s\n", mode, layers, iters, ms * 1e3, ms * 1e3 / (layers * iters),
                   ab ? "  ** ABORTED (a spin timed out) **" : "");
            std::vector<u64> hs(stamp_words);
            CHK(hipMemcpy(hs.data(), stamps, stamp_words * 8, hipMemcpyDeviceToHost));
            const char *names[NST] = {"qkv", "att", "(merge)", "o", "gate|up", "down"};
            const int b0[NST] = {0, NMAT, 0, 0, 0, 0}, b1[NST] = {240, NMAT + NREC, 0, 160, 216, 160};
            for (int st_i = 0; st_i < NST; ++st_i) {
                if (st_i == 2) continue;
                std::vector<double> edge, comp, pub;
                for (int l = layers / 2; l < layers; ++l)
                    for (int b = b0[st_i]; b < b1[st_i]; ++b) {
                        const u64 *st = &hs[(((size_t)l * NST + st_i) * G + b) * 4];
                        if (!st[2]) continue;
                        edge.push_back((double)(st[1] - st[0]) * 10.0);
                        comp.push_back((double)(st[2] - st[1]) * 10.0);
                        if (st[3]) pub.push_back((double)((long long)st[3] - (long long)st[2]) * 10.0);
                    }
                auto med = [](std::vector<double> &v) {
                    if (v.empty()) return 0.0;
                    std::sort(v.begin(), v.end());
                    return v[v.size() / 2];
                };
                printf("    %-8s wave 0: edge wait %6.0f ns   compute %6.0f ns   -> published %6.0f ns later (att: chunk-0 merge)\n", names[st_i], med(edge), med(comp),
                       med(pub));
            }
            // the layer period as workgroup 0 sees it: qkv sweep start of consecutive layers
            std::vector<double> per_l;
            for (int l = layers / 2 + 1; l < layers; ++l) {
                const u64 a = hs[(((size_t)(l - 1) * NST + 0) * G + 0) * 4], c = hs[(((size_t)l * NST + 0) * G + 0) * 4];
                if (a && c) per_l.push_back((double)(c - a) * 10.0);
            }
            if (!per_l.empty()) {
                std::sort(per_l.begin(), per_l.end());
                printf("    layer period (workgroup 0, median): %.2f us\n", per_l[per_l.size() / 2] / 1e3);
            }
            fflush(stdout);
        }
        return ab ? std::vector<unsigned>() : res;
    };

    // ---- correctness: a short chain against the serial host model ----------------------------------------------------------
    {
        std::vector<unsigned> x = x0, qkv(QKV), rec(NREC * RECV), att(H), x2(H), h(F);
        for (int l = 0; l < check_layers; ++l) {
            const unsigned char *lw = hw.data() + (size_t)l * layer_bytes;
            size_t off = 0;
            int o16[16];
            for (int t = 0; t < 240; ++t) {
                host_tile(lw + off + (size_t)t * per[0] * 1024, 10, x, o16);
                for (int c = 0; c < 16; ++c) qkv[t * 16 + c] = (unsigned)o16[c];
            }
            off += (size_t)NMAT * per[0] * 1024;
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
            for (int kvh = 0; kvh < NKV; ++kvh)
                for (int t = 0; t < 512; ++t) {
                    const unsigned *own = &rec[(kvh * NCH) * RECV];
                    unsigned v = own[t] + own[512 + (t & 15)];
                    for (int cc = 0; cc < NCH - 1; ++cc) {
                        const unsigned *sb = &rec[(kvh * NCH + 1 + cc) * RECV];
                        v = v * 5u + sb[t] + sb[512 + (t & 15)];
                    }
                    att[kvh * 512 + t] = v;
                }
            for (int t = 0; t < 160; ++t) {
                host_tile(lw + off + (size_t)t * per[1] * 1024, 10, att, o16);
                for (int c = 0; c < 16; ++c) x2[t * 16 + c] = (unsigned)o16[c];
            }
            off += (size_t)NMAT * per[1] * 1024;
            for (int b = 0; b < 216; ++b)
                for (int ot = 0; ot < 2; ++ot) {
                    int a16[16], u16[16];
                    host_tile(lw + off + ((size_t)b * per[2] + (size_t)(2 * ot) * 10) * 1024, 10, x2, a16);
                    host_tile(lw + off + ((size_t)b * per[2] + (size_t)(2 * ot + 1) * 10) * 1024, 10, x2, u16);
                    for (int c = 0; c < 16; ++c) h[(b * 2 + ot) * 16 + c] = (unsigned)(a16[c] + 3 * u16[c]);
                }
            off += (size_t)NMAT * per[2] * 1024;
            for (int t = 0; t < 160; ++t) {
                host_tile(lw + off + (size_t)t * per[3] * 1024, 27, h, o16);
                for (int c = 0; c < 16; ++c) x[t * 16 + c] = (unsigned)o16[c];
            }
        }
        for (int mode = 0; mode < 2; ++mode) {
            const std::vector<unsigned> got = run(check_layers, 1, mode, false);
            size_t bad = 0;
            for (int i = 0; i < H && !got.empty(); ++i) bad += got[i] != x[i];
            printf("check mode %d (%d layers vs serial host model): %s (%zu of %d words differ)\n", mode, check_layers,
                   got.empty() ? "ABORTED" : bad ? "MISMATCH" : "exact", bad, H);
            if (got.empty() || bad) return 1;
        }
    }
    // ---- timing --------------------------------------------------------------------------------------------------------------
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 2; ++mode) run(n_layers, n_iter, mode, rep == 1);
    return 0;
}
