// kernels_prefill_attn.hip -- causal self-attention over a whole prompt (prefill) on gfx950.
//
// The reference runs the same per-head softmax(Q K^T / sqrt(d) + causal mask) V for a
// [T]-token forward (T:410-533) with RoPE applied to q and k (T:134-163) and k, v appended
// to the cache (T:1171-1202).  Here, for a FRESH sequence (positions 0..T-1):
//
//   k_prefill_prep  grid (T/64, heads + 2*kv_heads): one 64-token x 128-dim slab of q, k or v
//                   through LDS: RoPE (q, k); q -> f16 [head][T][128], pre-multiplied by the softmax
//                   scale and log2(e); k -> f16 [kv][T][128] and the decode cache (f32 transposed
//                   [kv][128][max_pos], or the f16 cache layout); v -> f16 transposed [kv][128][T]
//                   with the keys of every 32-group in MFMA-operand order, and the decode cache.
//                   Every global access is contiguous along the fastest index.
//   k_prefill_attn  4 waves = (heads of one KV head) x (wave columns of 32 queries), two workgroups per
//                   CU: flash attention on v_mfma_f32_16x16x32_f16, f32 accumulation, f32 online
//                   softmax in base 2.  K / V^T tiles arrive by LDS-DMA into unpadded XOR-swizzled
//                   double buffers; operand reads run two chunks ahead through a register ring.
//                   It works on S^T = K Q^T and O^T = V^T P^T so that the probabilities never
//                   leave registers: an S^T accumulator (lane: query c, keys 4g..4g+3) is
//                   already in B-operand form for the second product once the V^T operand
//                   uses the same key -> k-slot map.
//   k_prefill_merge combines the key-split partials of launches too small to fill the chip.
// q, k, v and the probabilities are rounded to f16 for the matrix cores (relative 2^-11);
// the f32 KV cache the decode steps read afterwards holds the exact f32 values.
#include <cstdlib>
#include <mutex>
#include <unordered_set>

#include "common.hpp"

namespace bitnet_hip {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kPD = 128;       // head dim
constexpr int kQB = 64;        // keys per tile; query rows come in 64-row blocks (q_block_pos)
constexpr int kQPad = 128;     // query rows are padded to this (the largest workgroup query tile)
constexpr int kKTile = kQB * kPD * 2;   // one K tile in LDS: 64 keys x 256 bytes, unpadded; 16-byte unit u of key r sits at u ^ (r & 15)
constexpr int kVTile = kPD * kQB * 2;   // one V^T tile: 128 dims x 128 bytes (keys in operand order, k_prefill_prep); unit u of dim r sits at u ^ ((r >> 1) & 7)
constexpr int kKVBuf = kKTile + kVTile; // the kernel keeps two (tile t + 1 lands while tile t is multiplied)
constexpr int kTilesPerSplit = 16;  // key split (PrefillArgs::ksplit): key tiles one workgroup walks at least before a block is cut

struct PrefillArgs {
    // element (head h, row t, dim d) of q / k / v sits at base + h * hs + t * ld + d
    const float *q, *k, *v;
    int ld_q, ld_kv;        // row strides (floats)
    int hs_q, hs_kv;        // head strides (floats): 128 for [row][head][dim] rows, rows*128 for [head][row][dim]
    int out_hs, out_ld;     // output element (h, t, d) at out + h * out_hs + t * out_ld + d
    int rope, causal;       // apply RoPE to q and k here / causal mask
    float scale;            // softmax scale (1/sqrt(d) in the transformer)
    const int *q_block_pos;  // absolute position of each 64-query block (null: block b starts at 64 b)
    int nq, nq_pad;          // query rows held here (token-parallel prefill: a subset of the prompt)
    const float *rope_sin, *rope_cos;
    float *kcache, *vcache;
    int n_heads, n_kv, max_pos, T, Tpad;  // T = context tokens (keys)
    _Float16 *qh, *kh, *vt;  // workspace: q [head][nq_pad][128], k [kv][Tpad][128], v^T [kv][128][Tpad]
    float *out;              // [nq, heads * 128]
    // token-parallel prefill: the k|v rows as the all-gather left them, [rank][that rank's two zigzag chunks] -- the row
    // of absolute position t is found here instead of by a scatter pass on the host side.  zz_world == 0: absolute order.
    int zz_world, zz_chunk;  // ranks, tokens per chunk (T = 2 * world * chunk; chunk % 64 == 0)
    int kv_f16;              // the k|v rows travelled as f16 (half the bytes on the wire); k / v then point at _Float16
    int cache_f16;           // the decode caches hold f16 (kernels_attn.hip KV16 layout)
    int head_fast;           // grid = (head groups, query groups): the dispatch order walks ALL heads' longest blocks first (see launch_attn_kernel)
    int out_f16;             // `out` holds _Float16 (same element strides): the o-projection's f16 chain reads it as it is (kernels_gemm.hip k_gemm_f16a)
    // key split: the key tiles of a query block are dealt to `ksplit` workgroups (blockIdx.z), each leaving an un-normalised partial
    // (o, m, l) that k_prefill_merge combines -- for launches whose query blocks alone would not fill the chip (one rank's 1024
    // queries x 8192 keys of the 8-GPU prefill are 160 workgroups with up to 128 key tiles each; a 1024-token prompt likewise)
    int phase;               // 0: prepare q, k, v and attend; 1: the query slabs only (needs no k|v: runs beside the all-gather);
                             // 2: the k / v slabs and the attention (the query slabs were prepared by a phase-1 call on the same workspace)
    int ksplit;
    int split_tiles;         // key tiles one workgroup walks at least before a query block is cut (attn_split_tiles)
    int slot0;               // k_prefill_prep: first slot of its grid (n_heads: the query slabs are prepared by k_prefill_attn itself)
    int q_in_kernel;         // k_prefill_attn reads the f32 query rows itself (RoPE, scale, f16) instead of the f16 image k_prefill_prep left
    float *part_o;           // [ksplit][nq_pad][heads][128]
    float *part_ml;          // [ksplit][nq_pad][heads][2]  (running maximum in base-2 units, sum)
};

// row of absolute position t (a multiple of 64) in the gathered k|v buffer: chunk c = t / chunk belongs to rank c (first
// half of the chunks) or 2 world - 1 - c (second half), as that rank's first or second chunk
__device__ __forceinline__ int zz_row(const PrefillArgs &p, int t) {
    if (p.zz_world == 0) return t;
    const int c = t / p.zz_chunk, off = t - c * p.zz_chunk;
    const bool first = c < p.zz_world;
    const int r = first ? c : 2 * p.zz_world - 1 - c;
    return r * 2 * p.zz_chunk + (first ? 0 : p.zz_chunk) + off;
}

// grid (max(nq_pad, Tpad) / 64, heads + 2 kv): slot < heads: query head (rows of p.q, positions from
// q_block_pos); then k heads, then v heads (rows of p.kv, position = row).
__global__ __launch_bounds__(256) void k_prefill_prep(PrefillArgs p) {
    // one 64-token x 128-dim slab through LDS; every global access is 16 bytes wide (8 for f16 rows) and contiguous along the
    // fastest index of its tensor (the scalar version of this kernel took 48 us per layer for 115 MB of traffic)
    __shared__ __attribute__((aligned(16))) float tile[kQB][kPD + 4];  // + 4: float4 rows stay 16-byte aligned, columns spread over the banks
    const int slot = blockIdx.y + p.slot0, t0 = blockIdx.x * kQB, tid = threadIdx.x;
    const bool is_q = slot < p.n_heads, is_k = !is_q && slot < p.n_heads + p.n_kv;
    if ((p.phase == 1 && !is_q) || (p.phase == 2 && is_q)) return;
    const int n_rows = is_q ? p.nq : p.T;
    if (t0 >= (is_q ? p.nq_pad : p.Tpad)) return;
    const float *src = is_q ? p.q + (size_t)slot * p.hs_q
                     : is_k ? p.k + (size_t)(slot - p.n_heads) * p.hs_kv : p.v + (size_t)(slot - p.n_heads - p.n_kv) * p.hs_kv;
    const int ld = is_q ? p.ld_q : p.ld_kv;
    // absolute position of row t0 (padding blocks past the last real one hold zeros: any position will do)
    const int pos0 = is_q && p.q_block_pos ? ((int)blockIdx.x < (p.nq + kQB - 1) / kQB ? p.q_block_pos[blockIdx.x] : 0) : t0;
    const int r0 = is_q ? t0 : zz_row(p, t0);  // a 64-row block of positions is 64 consecutive gathered rows (chunk % 64 == 0)
    const bool vec_ok = (ld & 3) == 0 && ((uintptr_t)src & 15) == 0;  // rows 16-byte aligned (always, for the decoder's buffers)
    if (!is_q && p.kv_f16) {
        const _Float16 *sh = reinterpret_cast<const _Float16 *>(p.k) + (size_t)(slot - p.n_heads) * p.hs_kv;  // k heads then v heads
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, tok = idx >> 5, d = (idx & 31) * 4;
            float4 v = {0.f, 0.f, 0.f, 0.f};
            if (t0 + tok < n_rows) {
                const _Float16 *e = sh + (size_t)(r0 + tok) * ld + d;
                v = float4{(float)e[0], (float)e[1], (float)e[2], (float)e[3]};
            }
            *reinterpret_cast<float4 *>(&tile[tok][d]) = v;
        }
    } else {
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, tok = idx >> 5, d = (idx & 31) * 4;
            float4 v = {0.f, 0.f, 0.f, 0.f};
            if (t0 + tok < n_rows) {
                const float *e = src + (size_t)(r0 + tok) * ld + d;
                v = vec_ok ? *reinterpret_cast<const float4 *>(e) : float4{e[0], e[1], e[2], e[3]};
            }
            *reinterpret_cast<float4 *>(&tile[tok][d]) = v;
        }
    }
    __syncthreads();
    if ((is_q || is_k) && p.rope) {
        // split-half RoPE (crates/bitnet-rope/src/lib.rs:59-93, T:134-163): pairs (j, j + 64); a thread takes 4 adjacent pairs
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, tok = idx >> 4, j = (idx & 15) * 4;
            const int pos = t0 + tok < n_rows ? pos0 + tok : 0;
            const float4 s = *reinterpret_cast<const float4 *>(p.rope_sin + (size_t)pos * 64 + j), c = *reinterpret_cast<const float4 *>(p.rope_cos + (size_t)pos * 64 + j);
            const float4 x0 = *reinterpret_cast<const float4 *>(&tile[tok][j]), x1 = *reinterpret_cast<const float4 *>(&tile[tok][64 + j]);
            *reinterpret_cast<float4 *>(&tile[tok][j]) = float4{x0.x * c.x - x1.x * s.x, x0.y * c.y - x1.y * s.y, x0.z * c.z - x1.z * s.z, x0.w * c.w - x1.w * s.w};
            *reinterpret_cast<float4 *>(&tile[tok][64 + j]) = float4{x0.x * s.x + x1.x * c.x, x0.y * s.y + x1.y * c.y, x0.z * s.z + x1.z * c.z, x0.w * s.w + x1.w * c.w};
        }
        __syncthreads();
    }
    // the softmax scale (and the change to base 2) rides in the f16 image of q: one multiply per query element here instead of
    // one per score in every key tile
    const float qmul = is_q ? 1.4426950408889634f * p.scale : 1.0f;
    auto store_rows_f16 = [&](_Float16 *dst) {  // [64 tokens][128] halves, contiguous: 8 bytes per thread and iteration
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, tok = idx >> 5, d = (idx & 31) * 4;
            const float4 v = *reinterpret_cast<const float4 *>(&tile[tok][d]);
            const v4h h = {(_Float16)(v.x * qmul), (_Float16)(v.y * qmul), (_Float16)(v.z * qmul), (_Float16)(v.w * qmul)};
            *reinterpret_cast<v4h *>(dst + (size_t)tok * kPD + d) = h;
        }
    };
    if (is_q) {
        store_rows_f16(p.qh + ((size_t)slot * p.nq_pad + t0) * kPD);
    } else if (is_k) {
        const int kvh = slot - p.n_heads;
        store_rows_f16(p.kh + ((size_t)kvh * p.Tpad + t0) * kPD);
        // decode cache: K transposed in 64-position tiles [chunk][128][64] (kernels_attn.hip); t0 is a tile start
        const size_t head_floats = (size_t)((p.max_pos + 63) / 64) * 64 * kPD;
        float *kt = p.kcache + (size_t)kvh * head_floats + (size_t)(t0 >> 6) * kPD * 64;
        if (p.cache_f16) {  // [chunk][D / 2][64][2] halves
            _Float16 *k16 = reinterpret_cast<_Float16 *>(p.kcache) + (size_t)kvh * head_floats + (size_t)(t0 >> 6) * kPD * 64;
            for (int i = 0; i < 32 && p.kcache; ++i) {
                const int idx = tid + 256 * i, d = idx >> 6, tok = idx & 63;
                if (t0 + tok < p.T) k16[((size_t)(d >> 1) * 64 + tok) * 2 + (d & 1)] = (_Float16)tile[tok][d];
            }
        } else {
            for (int i = 0; i < 8 && p.kcache; ++i) {  // 4 adjacent positions of one dim: 16 contiguous bytes
                const int idx = tid + 256 * i, d = idx >> 4, tok = (idx & 15) * 4;
                if (t0 + tok + 3 < p.T) {
                    *reinterpret_cast<float4 *>(kt + (size_t)d * 64 + tok) = float4{tile[tok][d], tile[tok + 1][d], tile[tok + 2][d], tile[tok + 3][d]};
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (t0 + tok + e < p.T) kt[(size_t)d * 64 + tok + e] = tile[tok + e][d];
                }
            }
        }
    } else {
        const int kvh = slot - p.n_heads - p.n_kv;
        float *vc = p.vcache + (size_t)kvh * ((size_t)((p.max_pos + 63) / 64) * 64 * kPD);  // [max_pos][128]
        if (p.cache_f16) {
            _Float16 *v16 = reinterpret_cast<_Float16 *>(p.vcache) + (size_t)kvh * ((size_t)((p.max_pos + 63) / 64) * 64 * kPD);
            for (int i = 0; i < 32 && p.vcache; ++i) {
                const int idx = tid + 256 * i, tok = idx >> 7, d = idx & 127;
                if (t0 + tok < p.T) v16[(size_t)(t0 + tok) * kPD + d] = (_Float16)tile[tok][d];
            }
        } else {
            for (int i = 0; i < 8 && p.vcache; ++i) {
                const int idx = tid + 256 * i, tok = idx >> 5, d = (idx & 31) * 4;
                if (t0 + tok < p.T) *reinterpret_cast<float4 *>(vc + (size_t)(t0 + tok) * kPD + d) = *reinterpret_cast<const float4 *>(&tile[tok][d]);
            }
        }
        // [128][Tpad], the keys of every group of 32 in the order the second product's operand wants them: key 4 a + 16 b + j
        // (a < 4, b < 2, j < 4) at slot 8 a + 4 b + j, so that a lane's 8 k-slots (keys 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3: the
        // S^T accumulator layout) are ONE 16-byte unit.  4 adjacent positions of one dim = 8 contiguous bytes here as well.
        _Float16 *vt = p.vt + (size_t)kvh * kPD * p.Tpad;
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, d = idx >> 4, t4 = idx & 15, tok = t4 * 4;
            const int slot = 32 * (t4 >> 3) + 8 * (t4 & 3) + 4 * ((t4 >> 2) & 1);
            const v4h h = {(_Float16)tile[tok][d], (_Float16)tile[tok + 1][d], (_Float16)tile[tok + 2][d], (_Float16)tile[tok + 3][d]};
            *reinterpret_cast<v4h *>(vt + (size_t)d * p.Tpad + t0 + slot) = h;
        }
    }
}

// NW waves per workgroup = HW query heads of ONE KV head x (NW / HW) wave columns; a wave owns NQ groups of 16 queries of its
// head.  All waves multiply the same K / V^T tiles, staged into LDS once per workgroup.
// Every MFMA takes its A operand (16 keys x 32 dims = 1 KiB per wave) from LDS, and those reads are what bounds this kernel
// (ablation, 4096 tokens: 238 us; without the operand reads 137; without softmax arithmetic 218; without tile staging 205;
// MFMAs alone ~100): with ONE query group per wave the reads need 64 B/clk per SIMD of a CU's 128 B/clk.  NQ = 2 lets two
// query groups share each operand read.  That takes ~240 registers, so the GQA form runs 4-wave workgroups (4 heads x 32
// queries) with a launch bound of two waves per SIMD: TWO workgroups per CU with independent barriers (one in its softmax
// while the other multiplies).  The next key tile is requested into registers one iteration ahead and the LDS operand reads
// are pinned one chunk ahead of their MFMAs (hipcc's own schedule waited for every read right before its MFMAs).
typedef unsigned pv4u __attribute__((ext_vector_type(4)));

// max of three without the v_max(x, x) canonicalisation hipcc puts in front of every fmaxf of an MFMA result (no NaNs here)
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// max over the four lanes (c, g = 0..3) that hold one query's keys, in every lane, on the VALU: v_permlane16_swap / v_permlane32_swap
// (gfx950) instead of two dependent ds_bpermute round trips through the LDS pipe per query group and key tile.  With both operands
// the same register, permlane16_swap leaves (rows 0, 0, 2, 2) and (rows 1, 1, 3, 3) of 16 lanes, permlane32_swap (low half twice)
// and (high half twice): the maximum of each pair is the xor-16 / xor-32 butterfly step.
__device__ __forceinline__ float max_over_g(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m1 = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned w = __float_as_uint(m1);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// LDS-DMA: 64 lanes x 16 B from (scalar base + per-lane byte offset) to LDS bytes [lds_dst, lds_dst + 1024).  Invisible to hipcc's
// vmcnt bookkeeping (cdna_hip_programming.md 5.7): the kernel waits with explicit s_waitcnt vmcnt(0) and issues no other vector
// memory loads while these are in flight.
__device__ __forceinline__ void gdma1k_s(unsigned lane_off, const void *sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

template <int HW, int NW, int NQ>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_prefill_attn(PrefillArgs p) {
    constexpr int QW = NW / HW, QG = 16 * QW * NQ;  // wave columns per workgroup, queries per workgroup
    static_assert(NW == 4, "tile staging deals 32 pieces to 4 waves");
    extern __shared__ __attribute__((aligned(16))) uint8_t kv_lds[];  // [2][K tile | V^T tile]
    __shared__ int s_last;
    const int qg = p.head_fast ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)gridDim.x - 1 - (int)blockIdx.x;  // long (late) blocks first
    const int hg = p.head_fast ? blockIdx.x : blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4;
    const int h = hg * HW + wave % HW, kvh = h / (p.n_heads / p.n_kv);  // HW divides the group: one KV head per workgroup
    const int qbase = qg * QG + (wave / HW) * 16 * NQ;            // first query row of this wave (16 NQ rows inside one 64-row block)
    const int blk64 = qbase >> 6;                                 // the 64-row block the wave's rows lie in
    const int bpos = (p.q_block_pos ? p.q_block_pos[blk64 < (p.nq + kQB - 1) / kQB ? blk64 : 0] : blk64 * kQB) + (qbase & 63);
    if (tid == 0) s_last = 0;
    __syncthreads();
    {   // last key tile any query of the workgroup sees
        const int top = bpos + 16 * NQ - 1;
        const int need = p.causal ? (top < p.T ? top : p.T - 1) / kQB : (p.T - 1) / kQB;
        if (lane == 0) atomicMax(&s_last, need);
    }
    __syncthreads();
    int kt_first = 0, kt_last = s_last;
    if (p.ksplit > 1) {
        // this workgroup's share of the key tiles: a query block with n tiles is cut into ceil(n / kTilesPerSplit) parts (at most
        // ksplit), so early blocks (few tiles) stay whole and late ones spread over the chip; surplus workgroups only mark their
        // partial empty (m = -inf), which the merge skips
        const int n_t = kt_last + 1;
        int parts = (n_t + p.split_tiles - 1) / p.split_tiles;
        parts = parts < p.ksplit ? parts : p.ksplit;
        if ((int)blockIdx.z >= parts) {
            if (g == 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const size_t ro = ((size_t)blockIdx.z * p.nq_pad + qbase + 16 * q + c) * p.n_heads + h;
                    *reinterpret_cast<float2 *>(p.part_ml + ro * 2) = float2{-INFINITY, 0.0f};
                }
            }
            return;
        }
        const int per = (n_t + parts - 1) / parts;
        kt_first = (int)blockIdx.z * per;
        kt_last = kt_first + per - 1 < kt_last ? kt_first + per - 1 : kt_last;
    }
    int qlim[NQ];  // highest visible key position of this lane's query in group q
#pragma unroll
    for (int q = 0; q < NQ; ++q) qlim[q] = p.causal ? bpos + 16 * q + c : p.T - 1;
    // Q^T operands: 8 consecutive dims per k-slot group, kept in registers for the whole block (filled below, behind the first tile requests)
    v8h qreg[NQ][4];
    v4f o[NQ][8];
    float m_run[NQ], l_run[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) o[q][dt] = (v4f){0.f, 0.f, 0.f, 0.f};
        m_run[q] = -INFINITY;
        l_run[q] = 0.0f;
    }
    // softmax in base 2; the scale is already in q (k_prefill_prep)
    const int qmin = __builtin_amdgcn_readfirstlane(p.causal ? bpos : p.T - 1);  // lowest key limit of any query of this wave
    const uint8_t *kbase = reinterpret_cast<const uint8_t *>(p.kh + (size_t)kvh * p.Tpad * kPD);
    const uint8_t *vbase = reinterpret_cast<const uint8_t *>(p.vt + (size_t)kvh * kPD * p.Tpad);
    // Tile staging by LDS-DMA: a K tile is 16 pieces of 1 KiB (4 keys), a V^T tile 16 pieces (8 dim rows x 128 B); wave w moves
    // pieces 4 w .. 4 w + 3 of each.  A lane fetches the 16-byte unit that belongs at ITS slot of the piece (the XOR swizzles
    // above), so the unpadded tiles read conflict-free.  Byte offsets from the tile's first byte (scalar base):
    unsigned ksrc[4], vsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = 4 * (4 * wave + i) + (lane >> 4), dim = 8 * (4 * wave + i) + (lane >> 3);
        ksrc[i] = (unsigned)(key * 256 + (((lane & 15) ^ (key & 15)) * 16));
        vsrc[i] = (unsigned)dim * (unsigned)(p.Tpad * 2) + (unsigned)((((lane & 7) ^ ((dim >> 1) & 7))) * 16);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)kv_lds;
    auto stage_k = [&](int kt, int buf) {
        const uint8_t *src = kbase + (size_t)kt * kKTile;
        const unsigned dst = lds0 + (unsigned)buf * kKVBuf + (unsigned)(4 * wave) * 1024u;
#pragma unroll
        for (int i = 0; i < 4; ++i) gdma1k_s(ksrc[i], src, dst + 1024u * i);
    };
    auto stage_v = [&](int kt, int buf) {
        const uint8_t *src = vbase + (size_t)kt * (kQB * 2);
        const unsigned dst = lds0 + (unsigned)buf * kKVBuf + kKTile + (unsigned)(4 * wave) * 1024u;
#pragma unroll
        for (int i = 0; i < 4; ++i) gdma1k_s(vsrc[i], src, dst + 1024u * i);
    };
    // operand read offsets inside a tile (+ 4096 i per key tile of 16, + 2048 dt per dim tile of 16: immediates)
    const unsigned kro = (unsigned)(c * 256 + ((g ^ c) * 16));                               // ^ (64 ch): unit 4 ch + g of key c
    const unsigned vro = (unsigned)(c * 128 + ((g ^ (c >> 1)) * 16));                         // ^ (64 u): unit 4 u + g of dim c
    // The MFMA operands of a tile form one stream of 8 chunks (K dims 0-31 .. 96-127, then V^T (keys 0-31, dims 0-63), (0-31, 64-127),
    // (32-63, 0-63), (32-63, 64-127)), 4 operands of 1 KiB per wave each; chunk j's LDS reads are issued TWO chunks ahead of its
    // MFMAs, across the softmax and across the tile boundary (hipcc's own schedule waited for every read right before its MFMAs;
    // one chunk ahead still left the waves waiting: the reads' latency, not LDS bandwidth, was what the kernel stood on).
    v8h ring[4][4];
    auto kread = [&](int slot, int ch, const uint8_t *ks) {
        const uint8_t *kp = ks + (kro ^ (unsigned)(64 * ch));
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[slot][i] = *reinterpret_cast<const v8h *>(kp + 4096 * i);
    };
    auto vread = [&](int slot, int st, const uint8_t *vs) {
        const uint8_t *vp = vs + (vro ^ (unsigned)(64 * (st >> 1))) + 2048 * 4 * (st & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[slot][i] = *reinterpret_cast<const v8h *>(vp + 2048 * i);
    };
    // Staging: K and V^T tiles have separate double buffers and separate schedules, each requested a whole tile ahead of its first
    // read.  Two barriers per tile: B1 (before the K chunk 2 MFMAs = before the first V read of this tile is issued) waits for
    // V(kt) and requests V(kt + 1) into the buffer V(kt - 1) left (every wave is past chunk 1 of tile kt: done with tile kt - 1);
    // B2 (before the V chunk 2 MFMAs = before the first K read of the NEXT tile is issued) waits for K(kt + 1) and requests K(kt + 2)
    // into the buffer K(kt) left (its last read was issued two chunks ago and consumed by K chunk 3).  Loads return in order, so
    // "the 4 pieces requested last may still be in flight" is s_waitcnt vmcnt(4).
    stage_k(kt_first, 0);
    stage_v(kt_first, 0);
    if (kt_first < kt_last) stage_k(kt_first + 1, 1);
    // the query operands, requested BEHIND the first K / V^T tiles so that their round trip (strided f32 rows + the RoPE tables) overlaps the tiles'
    if (p.q_in_kernel) {
        // straight from the f32 rows of the q|k|v projection (round 4: two thirds of k_prefill_prep's traffic were the query slabs): a lane's
        // dims 32 ch + 8 g + i (ch < 4) hold both members of every rotation pair (d, d + 64) = (ch, ch + 2), so the split-half RoPE
        // (crates/bitnet-rope/src/lib.rs:59-93, T:134-163) needs no exchange; the same arithmetic as k_prefill_prep, the softmax scale and the
        // change to base 2 folded into the f16 image.  Rows past the last query are zeros.  Every load is unconditional (clamped row).
        const float qmul = 1.4426950408889634f * p.scale;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = qbase + 16 * q + c;
            const bool live = row < p.nq;
            const float *qp = p.q + (size_t)h * p.hs_q + (size_t)(live ? row : 0) * p.ld_q + 8 * g;
            float x[4][8];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const float4 a = *reinterpret_cast<const float4 *>(qp + 32 * ch), b = *reinterpret_cast<const float4 *>(qp + 32 * ch + 4);
                x[ch][0] = a.x, x[ch][1] = a.y, x[ch][2] = a.z, x[ch][3] = a.w, x[ch][4] = b.x, x[ch][5] = b.y, x[ch][6] = b.z, x[ch][7] = b.w;
            }
            if (p.rope) {
                const int pos = live ? bpos + 16 * q + c : 0;
                const float *sp = p.rope_sin + (size_t)pos * 64 + 8 * g, *cp = p.rope_cos + (size_t)pos * 64 + 8 * g;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const float4 s0 = *reinterpret_cast<const float4 *>(sp + 32 * ch), s1 = *reinterpret_cast<const float4 *>(sp + 32 * ch + 4);
                    const float4 c0 = *reinterpret_cast<const float4 *>(cp + 32 * ch), c1 = *reinterpret_cast<const float4 *>(cp + 32 * ch + 4);
                    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float x0 = x[ch][i], x1 = x[ch + 2][i];
                        x[ch][i] = x0 * cs[i] - x1 * sn[i];
                        x[ch + 2][i] = x0 * sn[i] + x1 * cs[i];
                    }
                }
            }
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int i = 0; i < 8; ++i) qreg[q][ch][i] = live ? (_Float16)(x[ch][i] * qmul) : (_Float16)0.0f;
        }
    } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const _Float16 *qp = p.qh + ((size_t)h * p.nq_pad + qbase + 16 * q + c) * kPD + 8 * g;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) qreg[q][ch] = *reinterpret_cast<const v8h *>(qp + 32 * ch);
        }
    }
    // the builtin form, so that hipcc's own bookkeeping sees its q loads retired here: left to itself it re-waits for them with
    // s_waitcnt vmcnt(0..7) in front of the MFMAs of EVERY iteration -- which, in hardware, waits for the tile requests it cannot see
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    asm volatile("" ::: "memory");
    __syncthreads();
    kread(0, 0, kv_lds);
    kread(1, 1, kv_lds);

    for (int kt = kt_first; kt <= kt_last; ++kt) {
        const int par = (kt - kt_first) & 1;
        const uint8_t *ks = kv_lds + par * kKVBuf, *vs = ks + kKTile, *ks_next = kv_lds + (par ^ 1) * kKVBuf;
        const bool more = kt < kt_last;  // workgroup-uniform
        // ---- S^T = K Q^T - m: 4 key tiles of 16, reduced over 4 dim chunks of 32; one operand read per NQ MFMAs.  The accumulators
        // start at minus the running reference point, so the common tile (reference unchanged) exponentiates them as they are ----
        v4f s[NQ][4];
        float m_init[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            m_init[q] = m_run[q] < -1.0e30f ? 0.0f : m_run[q];  // no reference yet (first tile, or only masked keys so far)
#pragma unroll
            for (int i = 0; i < 4; ++i) s[q][i] = (v4f){-m_init[q], -m_init[q], -m_init[q], -m_init[q]};
        }
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (ch == 2) {  // B1
                if (more)
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (more) stage_v(kt + 1, par ^ 1);
            }
            if (ch < 2)
                kread((ch + 2) & 3, ch + 2, ks);
            else
                vread((ch + 2) & 3, ch - 2, vs);
            __builtin_amdgcn_sched_barrier(0);  // the reads above stay above these MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < NQ; ++q) s[q][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[ch][i], qreg[q][ch], s[q][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- online softmax for query column c of each group (keys of this lane: 16 i + 4 g + j) ----
        // Only tiles that reach past some query's limit are masked (the diagonal ones, and the context's last): wave-uniform.
        const bool need_mask = kt * kQB + kQB - 1 > qmin;
        v8h pb[NQ][2];
        // The query groups' softmax chains are independent: each stage below handles ALL groups in one basic block (one branch per
        // stage, not per group), so hipcc interleaves their dependent chains -- maximum, exponentials, sum -- instead of running
        // them one after the other.
        float mt[NQ];
        if (need_mask) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                mt[q] = -INFINITY;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int pos = kt * kQB + 16 * i + 4 * g + j;
                        const float v = pos <= qlim[q] ? s[q][i][j] : -INFINITY;  // causal mask (T:452-470) / end of the context
                        s[q][i][j] = v;
                        mt[q] = fmaxf(mt[q], v);
                    }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {  // two half-chains per group
                const float ma = max3_raw(max3_raw(max3_raw(s[q][0][0], s[q][0][1], s[q][0][2]), s[q][0][3], s[q][1][0]), s[q][1][1], s[q][1][2]);
                const float mb = max3_raw(max3_raw(max3_raw(s[q][2][0], s[q][2][1], s[q][2][2]), s[q][2][3], s[q][3][0]), s[q][3][1], s[q][3][2]);
                mt[q] = max3_raw(max3_raw(ma, mb, s[q][1][3]), s[q][3][3], -INFINITY);
            }
        }
        // mt is relative to the running reference.  The reference only moves when the maximum outgrows it by more than 2^8 (the
        // f16 probabilities then stay <= 256, the sums are f32): most tiles skip both the shift of the scores and the rescaling
        // of the 32 output accumulators.  A tile wholly above a query's limit (a later key split's first tile: its 64-row block
        // spans several query groups) leaves mt = -inf: the clamp keeps 2^(-inf - -inf) out and the reference stays "unset".
        bool unset[NQ], grow[NQ], any_grow = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            mt[q] = max_over_g(mt[q]);
            unset[q] = m_run[q] < -1.0e30f;
            grow[q] = mt[q] > 8.0f || unset[q];
            any_grow = any_grow || grow[q];
        }
        if (__any(any_grow)) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float shift = grow[q] ? fmaxf(mt[q], -3.0e38f) : 0.0f;
                const float m_new = unset[q] ? shift : m_run[q] + shift;
                const float alpha = __builtin_amdgcn_exp2f(m_run[q] - m_new);  // (no lane of the group grows: shift 0, alpha 1)
                l_run[q] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) o[q][dt] *= alpha;
                m_run[q] = m_new;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[q][i][j] -= shift;
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float ls[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) s[q][i][j] = __builtin_amdgcn_exp2f(s[q][i][j]);
                ls[i] = (s[q][i][0] + s[q][i][1]) + (s[q][i][2] + s[q][i][3]);
            }
            l_run[q] += (ls[0] + ls[1]) + (ls[2] + ls[3]);
            // P^T operand: k-slot (g, j) = key 32u + 4g + j (j < 4), 32u + 16 + 4g + (j - 4)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pb[q][u][j] = (_Float16)s[q][2 * u][j];
                    pb[q][u][4 + j] = (_Float16)s[q][2 * u + 1][j];
                }
        }
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int st = 0; st < 4; ++st) {  // step = (u, half of the dim tiles)
            const int u = st >> 1, d0 = 4 * (st & 1);
            if (st == 2) {  // B2
                if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                __syncthreads();
                if (kt + 2 <= kt_last) stage_k(kt + 2, par);
            }
            if (st < 2)
                vread(st + 2, st + 2, vs);
            else if (more)
                kread(st - 2, st - 2, ks_next);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < NQ; ++q) o[q][d0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[st][i], pb[q][u], o[q][d0 + i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float l = l_run[q];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const int qrow = qbase + 16 * q + c;
        if (p.ksplit > 1) {  // un-normalised partial; padding rows are written too (the merge never reads them)
            const size_t ro = ((size_t)blockIdx.z * p.nq_pad + qrow) * p.n_heads + h;
            float *op = p.part_o + ro * kPD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<float4 *>(op + 16 * dt) = float4{o[q][dt][0], o[q][dt][1], o[q][dt][2], o[q][dt][3]};
            if (g == 0) *reinterpret_cast<float2 *>(p.part_ml + ro * 2) = float2{m_run[q], l};
            continue;
        }
        if (qrow < p.nq) {
            const float inv = 1.0f / l;
            const size_t oo = (size_t)qrow * p.out_ld + (size_t)h * p.out_hs + 4 * g;
            if (p.out_f16) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                _Float16 *oh = reinterpret_cast<_Float16 *>(p.out) + oo;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt)
                    *reinterpret_cast<h4 *>(oh + 16 * dt) = (h4){(_Float16)(o[q][dt][0] * inv), (_Float16)(o[q][dt][1] * inv), (_Float16)(o[q][dt][2] * inv), (_Float16)(o[q][dt][3] * inv)};
            } else {
                float *op = p.out + oo;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) {
                    const float4 v = {o[q][dt][0] * inv, o[q][dt][1] * inv, o[q][dt][2] * inv, o[q][dt][3] * inv};
                    *reinterpret_cast<float4 *>(op + 16 * dt) = v;
                }
            }
        }
    }
}

// out[row][head][:] = sum_s 2^(m_s - M) o_s / sum_s 2^(m_s - M) l_s over the key splits; 32 threads (4 dims each) per (row, head)
__global__ __launch_bounds__(256) void k_prefill_merge(PrefillArgs p) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int d4 = (int)(i & 31);
    const size_t rh = i >> 5;  // row * heads + head
    if (rh >= (size_t)p.nq * p.n_heads) return;
    const size_t row = rh / p.n_heads, h = rh - row * p.n_heads;
    const size_t stride = (size_t)p.nq_pad * p.n_heads;
    float M = -INFINITY;
    for (int s = 0; s < p.ksplit; ++s) M = fmaxf(M, p.part_ml[(s * stride + rh) * 2]);  // split 0 always holds key 0: finite
    float L = 0.0f;
    float4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.ksplit; ++s) {
        const float2 ml = *reinterpret_cast<const float2 *>(p.part_ml + (s * stride + rh) * 2);
        if (ml.x == -INFINITY) continue;  // a part this block did not need: nothing but the mark was written
        const float w = __builtin_amdgcn_exp2f(ml.x - M);
        const float4 o = *reinterpret_cast<const float4 *>(p.part_o + (s * stride + rh) * kPD + 4 * d4);
        L += w * ml.y;
        a.x += w * o.x, a.y += w * o.y, a.z += w * o.z, a.w += w * o.w;
    }
    const float inv = 1.0f / L;
    if (p.out_f16) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(p.out) + row * p.out_ld + h * p.out_hs + 4 * d4) = (h4){(_Float16)(a.x * inv), (_Float16)(a.y * inv), (_Float16)(a.z * inv), (_Float16)(a.w * inv)};
        return;
    }
    *reinterpret_cast<float4 *>(p.out + row * p.out_ld + h * p.out_hs + 4 * d4) = float4{a.x * inv, a.y * inv, a.z * inv, a.w * inv};
}

// key splits of a launch (grid.z): only when the query blocks alone leave CUs idle (fewer than two workgroups per CU), as many
// as the longest block has 16-tile parts, at most 8
static int attn_split_tiles() {
    static const int v = [] { const char *e = getenv("BITNET_HIP_ATTN_SPLIT_TILES"); const int x = e ? atoi(e) : kTilesPerSplit; return x < 1 ? 1 : x; }();
    return v;
}
static int attn_ksplit(int n_heads, int n_kv, int nq_pad, int T) {
    const int group = n_heads / n_kv, hw = group % 4 == 0 ? 4 : group % 2 == 0 ? 2 : 1, qg = 16 * (4 / hw) * 2;
    const long n_wg = (long)(nq_pad / qg) * (n_heads / hw);
    const int tiles = (T + kQB - 1) / kQB;
    if (n_wg >= 512) return 1;
    const int s = (tiles + attn_split_tiles() - 1) / attn_split_tiles();
    return s < 1 ? 1 : s > 8 ? 8 : s;
}

static hipError_t launch_attn_kernel(const PrefillArgs &p, hipStream_t stream) {
    const int group = p.n_heads / p.n_kv;
    const unsigned z = (unsigned)p.ksplit;
    void (*ak)(PrefillArgs);
    dim3 grid;
    if (group % 4 == 0) {  // 4 waves = the 4 heads of a KV head x 32 queries each; two such workgroups per CU
        ak = k_prefill_attn<4, 4, 2>;
        grid = dim3((unsigned)(p.nq_pad / 32), (unsigned)(p.n_heads / 4), z);
    } else if (group % 2 == 0) {  // 2 heads x 2 wave columns x 32 queries
        ak = k_prefill_attn<2, 4, 2>;
        grid = dim3((unsigned)(p.nq_pad / 64), (unsigned)(p.n_heads / 2), z);
    } else {
        ak = k_prefill_attn<1, 4, 2>;
        grid = dim3((unsigned)(p.nq_pad / 128), (unsigned)p.n_heads, z);
    }
    // Workgroups are dispatched in linear id order (x fastest).  With the query groups in x, head group 0's blocks all start before head
    // group 1's first: on a causal prompt (640 workgroups for 512 slots at 4096 tokens x 20 heads) the LAST head group's longest blocks only
    // start once earlier groups' short ones have finished, and stand alone at the end.  With the head groups in x the order is longest
    // blocks of every head first, shortest last (longest-processing-time order over the whole launch).
    static const bool head_fast = !(getenv("BITNET_HIP_ATTN_HEAD_FAST") && atoi(getenv("BITNET_HIP_ATTN_HEAD_FAST")) == 0);
    PrefillArgs pa = p;
    pa.head_fast = head_fast ? 1 : 0;
    if (head_fast) grid = dim3(grid.y, grid.x, grid.z);
    {
        // two tile buffers = 64 KiB of dynamic LDS (+ a static word): raised once per kernel; entry points may run concurrently
        static std::mutex raised_mu;
        static std::unordered_set<const void *> raised;
        std::lock_guard<std::mutex> lk(raised_mu);
        if (!raised.count((const void *)ak)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ak), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kKVBuf);
            if (e != hipSuccess) return e;
            raised.insert((const void *)ak);
        }
    }
    hipLaunchKernelGGL(ak, grid, dim3(256), 2 * kKVBuf, stream, pa);
    if (p.ksplit > 1)
        hipLaunchKernelGGL(k_prefill_merge, dim3((unsigned)div_ceil((size_t)p.nq * p.n_heads * 32, 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

static size_t attn_f16_bytes(int n_heads, int n_kv, size_t qpad, size_t tpad) {
    return div_ceil(((size_t)n_heads * qpad + 2 * (size_t)n_kv * tpad) * kPD * sizeof(_Float16), 256) * 256;
}

size_t attn_prefill_workspace_bytes(int n_heads, int n_kv, int nq, int T) {
    const size_t qpad = div_ceil((size_t)nq, kQPad) * kQPad, tpad = div_ceil((size_t)T, kQB) * kQB;
    const int ks = attn_ksplit(n_heads, n_kv, (int)qpad, T);
    return attn_f16_bytes(n_heads, n_kv, qpad, tpad) + (ks > 1 ? (size_t)ks * qpad * n_heads * (kPD + 2) * sizeof(float) : 0) + 256;
}

// q: nq query rows (stride ld_q floats) whose 64-row blocks sit at absolute positions q_block_pos
// (null: 64 b); kv: T context rows in absolute order (stride ld_kv: k heads then v heads).
hipError_t launch_attn_prefill(const float *q, int ld_q, const int *q_block_pos, int nq, const float *kv, int ld_kv, int T,
                               const float *rope_sin, const float *rope_cos, float *kcache, float *vcache, int n_heads, int n_kv,
                               int D, int max_pos, void *workspace, size_t workspace_bytes, float *out, hipStream_t stream,
                               int zz_world, int kv_f16, int cache_f16, int phase) {
    if (phase < 0 || phase > 2) return hipErrorInvalidValue;
    if (D != kPD || T <= 0 || nq <= 0 || T > max_pos || n_heads % n_kv != 0) return hipErrorInvalidValue;
    if (zz_world < 0 || (zz_world > 0 && (T % (2 * zz_world * kQB) != 0))) return hipErrorInvalidValue;
    if (!workspace || workspace_bytes < attn_prefill_workspace_bytes(n_heads, n_kv, nq, T)) return hipErrorInvalidValue;
    PrefillArgs p;
    p.q = q;
    p.k = kv;
    p.v = kv + (size_t)n_kv * kPD;  // (f16 rows: the prep kernel indexes k heads then v heads from p.k itself)
    p.zz_world = zz_world;
    p.zz_chunk = zz_world > 0 ? T / (2 * zz_world) : 0;
    p.kv_f16 = kv_f16;
    p.cache_f16 = cache_f16 & 1;
    p.out_f16 = (cache_f16 >> 1) & 1;  // bit 1 of the flag word: f16 output rows
    p.ld_q = ld_q;
    p.ld_kv = ld_kv;
    p.hs_q = p.hs_kv = kPD;
    p.out_hs = kPD;
    p.out_ld = n_heads * kPD;
    p.rope = 1;
    p.causal = 1;
    p.scale = 1.0f / sqrtf((float)kPD);
    p.q_block_pos = q_block_pos;
    p.nq = nq;
    p.nq_pad = (int)(div_ceil((size_t)nq, kQPad) * kQPad);
    p.rope_sin = rope_sin;
    p.rope_cos = rope_cos;
    p.kcache = kcache;
    p.vcache = vcache;
    p.n_heads = n_heads;
    p.n_kv = n_kv;
    p.max_pos = max_pos;
    p.T = T;
    p.Tpad = (int)(div_ceil((size_t)T, kQB) * kQB);
    uint8_t *ws = reinterpret_cast<uint8_t *>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    p.qh = reinterpret_cast<_Float16 *>(ws);
    p.kh = p.qh + (size_t)n_heads * p.nq_pad * kPD;
    p.vt = p.kh + (size_t)n_kv * p.Tpad * kPD;
    p.out = out;
    p.ksplit = attn_ksplit(n_heads, n_kv, p.nq_pad, T);
    p.split_tiles = attn_split_tiles();
    p.part_o = reinterpret_cast<float *>(ws + attn_f16_bytes(n_heads, n_kv, (size_t)p.nq_pad, (size_t)p.Tpad));
    p.part_ml = p.part_o + (size_t)p.ksplit * p.nq_pad * n_heads * kPD;
    p.phase = phase;
    const unsigned nbq = (unsigned)(p.nq_pad / kQB), nbk = (unsigned)(p.Tpad / kQB);
    static const bool q_in_kernel = !(getenv("BITNET_HIP_ATTN_Q_IN_KERNEL") && atoi(getenv("BITNET_HIP_ATTN_Q_IN_KERNEL")) == 0);
    // the attention kernel needs 16-byte aligned f32 query rows for its own q path; anything else keeps the f16 image of the prep kernel
    p.q_in_kernel = q_in_kernel && (ld_q & 3) == 0 && ((uintptr_t)q & 15) == 0 ? 1 : 0;
    p.slot0 = 0;
    if (phase == 1) {  // the query slabs alone (nothing to do when the attention kernel prepares them itself)
        if (!p.q_in_kernel) hipLaunchKernelGGL(k_prefill_prep, dim3(nbq, (unsigned)n_heads), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    if (p.q_in_kernel) {
        p.slot0 = n_heads;
        hipLaunchKernelGGL(k_prefill_prep, dim3(nbk, (unsigned)(2 * n_kv)), dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL(k_prefill_prep, dim3(nbq > nbk ? nbq : nbk, (unsigned)(n_heads + 2 * n_kv)), dim3(256), 0, stream, p);
    }
    return launch_attn_kernel(p, stream);
}

// rows x [col0, col0 + ncols) of a row-major f32 matrix -> a compact [rows, ncols] buffer, f32 or f16: the k|v columns of the
// q|k|v projection as one contiguous send buffer for the token-parallel prefill's all-gather
__global__ void k_pack_cols(const float *__restrict__ src, size_t ld, size_t col0, size_t ncols, size_t total, void *__restrict__ dst, int f16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t r = i / ncols, c = i - r * ncols;
    const float v = src[r * ld + col0 + c];
    if (f16)
        static_cast<_Float16 *>(dst)[i] = (_Float16)v;
    else
        static_cast<float *>(dst)[i] = v;
}
hipError_t launch_pack_cols(const float *src, size_t ld, size_t col0, size_t ncols, size_t rows, void *dst, int f16, hipStream_t stream) {
    const size_t total = rows * ncols;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_cols, dim3((unsigned)div_ceil(total, 256)), dim3(256), 0, stream, src, ld, col0, ncols, total, dst, f16);
    return hipGetLastError();
}

// Plain multi-head attention over caller tensors [heads][seq][128] (no RoPE, no cache): the shape of
// the reference's fused_attention_hip stub (K/rocm/attention.rs:54-65), one batch element.
hipError_t launch_attn_generic(const float *q, const float *k, const float *v, float *out, int n_heads, int seq, int causal,
                               float scale, void *workspace, size_t workspace_bytes, hipStream_t stream) {
    if (seq <= 0 || n_heads <= 0 || !workspace || workspace_bytes < attn_prefill_workspace_bytes(n_heads, n_heads, seq, seq))
        return hipErrorInvalidValue;
    PrefillArgs p;
    p.q = q;
    p.k = k;
    p.v = v;
    p.ld_q = p.ld_kv = kPD;
    p.hs_q = p.hs_kv = seq * kPD;
    p.out_hs = seq * kPD;
    p.out_ld = kPD;
    p.rope = 0;
    p.causal = causal;
    p.scale = scale;
    p.q_block_pos = nullptr;
    p.zz_world = p.zz_chunk = p.kv_f16 = p.cache_f16 = p.out_f16 = p.head_fast = p.phase = 0;
    p.nq = seq;
    p.nq_pad = (int)(div_ceil((size_t)seq, kQPad) * kQPad);
    p.rope_sin = p.rope_cos = nullptr;
    p.kcache = p.vcache = nullptr;
    p.n_heads = p.n_kv = n_heads;
    p.max_pos = seq;
    p.T = seq;
    p.Tpad = (int)(div_ceil((size_t)seq, kQB) * kQB);
    uint8_t *ws = reinterpret_cast<uint8_t *>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    p.qh = reinterpret_cast<_Float16 *>(ws);
    p.kh = p.qh + (size_t)n_heads * p.nq_pad * kPD;
    p.vt = p.kh + (size_t)n_heads * p.Tpad * kPD;
    p.out = out;
    p.ksplit = attn_ksplit(n_heads, n_heads, p.nq_pad, seq);
    p.split_tiles = attn_split_tiles();
    p.slot0 = 0;
    p.q_in_kernel = 0;
    p.part_o = reinterpret_cast<float *>(ws + attn_f16_bytes(n_heads, n_heads, (size_t)p.nq_pad, (size_t)p.Tpad));
    p.part_ml = p.part_o + (size_t)p.ksplit * p.nq_pad * n_heads * kPD;
    const unsigned nb = (unsigned)(p.nq_pad / kQB);
    hipLaunchKernelGGL(k_prefill_prep, dim3(nb, (unsigned)(3 * n_heads)), dim3(256), 0, stream, p);
    return launch_attn_kernel(p, stream);
}

}  // namespace bitnet_hip
