// kernels_attn.hip -- one-token attention for the decode step (gfx950): RoPE (T:134-163)
// + KV append (T:1171-1202) + GQA softmax attention (T:410-533), split over 64-position
// chunks of the context so the work spreads over many CUs and every load of a thread is
// in flight at once (a per-head serial loop over the context is pure L2/HBM latency).
//
//   k_attn_partial  grid (n_kv, ceil(max_pos/64)): chunk-local scores, max m_c, exp-sum l_c
//                   and un-normalised P.V partial o_c for the whole query group of one KV
//                   head (each K/V element is read once per group);
//   k_attn_combine  grid (n_kv): out = sum_c e^(m_c-M) o_c / sum_c e^(m_c-M) l_c.
// Same value as the reference's softmax(QK^T/sqrt(d)) V up to f32 rounding.
//
// Cache layout (private to this library): K transposed in 64-position tiles,
// kcache[n_kv][ceil(max_pos/64)][D][64] -- one chunk's keys are 32 contiguous KB (a plain
// [D][max_pos] transpose scatters them over 128 segments of 256 B, which HBM serves badly at
// long contexts); element (d, pos) at ((pos / 64) * D + d) * 64 + pos % 64 (the score
// pass reads positions contiguously: lane = position); vcache[n_kv][max_pos][D] (the P.V pass
// reads dims contiguously: lane = dim).  *pos_ptr = number of cached tokens.
#include "common.hpp"
#include "qact.hpp"

namespace bitnet_hip {

constexpr int kAttnChunk = 64;
constexpr int kMaxGroup = 4;
constexpr int kD = 128;
// per (kv head, chunk) record in the scratch buffer: (m, l)[4], o[4][128]   (kAttnRecFloats, common.hpp)
constexpr int kRec = kAttnRecFloats;
static_assert(kRec == 2 * kMaxGroup + kMaxGroup * kD, "record layout");
typedef float v2f __attribute__((ext_vector_type(2)));

// K cache element (dim d, position pos) of one KV head
__device__ __forceinline__ size_t kidx(int d, int pos) { return ((size_t)(pos >> 6) * kD + d) * 64 + (pos & 63); }
__host__ __device__ __forceinline__ size_t kv_head_floats(int max_pos) { return (size_t)((max_pos + 63) / 64) * 64 * kD; }

template <int CTRL>
__device__ __forceinline__ float adpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float arl(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float awave_max(float v) {
    v = fmaxf(v, adpp<0xB1>(v));
    v = fmaxf(v, adpp<0x4E>(v));
    v = fmaxf(v, adpp<0x141>(v));
    v = fmaxf(v, adpp<0x140>(v));
    return fmaxf(fmaxf(arl(v, 0), arl(v, 16)), fmaxf(arl(v, 32), arl(v, 48)));
}
__device__ __forceinline__ float awave_sum(float v) {
    v += adpp<0xB1>(v);
    v += adpp<0x4E>(v);
    v += adpp<0x141>(v);
    v += adpp<0x140>(v);
    return (arl(v, 0) + arl(v, 16)) + (arl(v, 32) + arl(v, 48));
}

// In-kernel time stamps (s_memrealtime, 100 MHz): diagnostic build only (-DBH_STAMPS; tools/stamp_attn.py), 8 x u64 per workgroup.
#ifdef BH_STAMPS
#define BH_ASTAMP(i)                                                                                                        \
    do {                                                                                                                    \
        if (stamps && threadIdx.x == 0) stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define BH_ASTAMP(i) \
    do {             \
    } while (0)
#endif

// NH = 64-position halves per workgroup (256 threads each).  NH = 2: a workgroup covers 128 positions, its two halves
// run the chunk algorithm side by side and meet in LDS, so there is ONE record per 128 positions: half as many
// records for whoever merges them (the combine kernel, or the o-projection at short contexts) and, at 4k keys, 160
// workgroups instead of 320 (one per CU instead of 64 CUs with two).
__device__ __forceinline__ float amix_lo(float a, uint32_t h, float c) {  // a * f16(h.lo) + c, no conversion instruction
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ float amix_hi(float a, uint32_t h, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
    const _Float16 a = (_Float16)lo, b = (_Float16)hi;
    return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}

// KV16: the cache holds f16 (opt-in: half the bytes of the long-context stream; the reference's cache is f32 Tensor::cat,
// T:1171-1202 -- values rounded ONCE, from the exact f32 k / v, when they are appended).  Layout then:
// K [kv][chunk][D / 2][64 positions][2 dims] (a lane still reads 4 bytes per load: a dim PAIR of its position),
// V [kv][pos][D] halves (a thread takes a dim pair of every 4th position).
template <int NH, bool KV16>
__global__ __launch_bounds__(256 * NH) void k_attn_partial(const float *__restrict__ qkv, const float *__restrict__ rope_sin,
                                                           const float *__restrict__ rope_cos, float *__restrict__ kcache,
                                                           float *__restrict__ vcache, int n_heads, int n_kv, int group, int max_pos,
                                                           const int *__restrict__ pos_ptr, float *__restrict__ scratch,
                                                           unsigned long long *stamps /* diagnostic builds only */) {
    // every kernel argument is requested together with pos_ptr: left alone hipcc fetches the others only behind
    // the early exit, a second dependent scalar-load round trip for the workgroups that stay
    asm volatile("" ::"s"(qkv), "s"(rope_sin), "s"(rope_cos), "s"(kcache), "s"(vcache), "s"(n_heads), "s"(n_kv), "s"(group), "s"(max_pos), "s"(scratch));
    const int pos = *pos_ptr, t_k = pos + 1;
    const int kvh = blockIdx.x, pc = blockIdx.y;
    if (pc * NH * kAttnChunk >= t_k) return;  // record beyond the context (the grid is sized for max_pos)
    BH_ASTAMP(0);  // the position has arrived (a dependent scalar load behind the kernel arguments)
    __shared__ __attribute__((aligned(16))) float qs[kMaxGroup * kD];
    __shared__ __attribute__((aligned(16))) float kn[kD];
    __shared__ float vn[kD];
    __shared__ float partial[NH][4][kAttnChunk][kMaxGroup];
    constexpr int NPQ = KV16 ? 4 : 2;  // position classes of the P.V pass (a thread takes every NPQ-th position)
    __shared__ float red[NH][NPQ * kMaxGroup * kD];  // [half][position class][head][dim] partial P.V sums
    // softmax weights, stored [head][position class][position / NPQ]: the P.V pass reads four of its positions per
    // ds_read_b128 instead of one per ds_read_b32
    __shared__ __attribute__((aligned(16))) float sc[NH][kMaxGroup][NPQ][kAttnChunk / NPQ];
    __shared__ __attribute__((aligned(16))) float enew[kMaxGroup];  // softmax weight of the new token (its half only)
    __shared__ float hm[NH][kMaxGroup], hl[NH][kMaxGroup];          // per-half (m, l) for the merge (NH == 2)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);  // in an SGPR: LDS / cache bases on the scalar unit
    const int half = NH == 2 ? wave_all >> 2 : 0, wave = wave_all & 3, t4 = tid & 255;
    const int chunk = pc * NH + half, j0 = chunk * kAttnChunk;
    const bool live = j0 < t_k;  // NH == 2: the second half of the last record may lie past the context
    const int hd = kD / 2;
    // The chunk that holds the new token is also the only partial one (pos in [j0, j0 + 64)  <=>  t_k - j0 <= 64).
    // Wave-uniform, so everything that treats the new token or the tail sits behind ONE scalar branch
    // and the other chunks run straight-line code (per-element selects around LDS reads cost an exec-mask
    // branch each: 64 of them were a third of this kernel's instructions).
    const bool last = live && t_k - j0 <= kAttnChunk;
    const float *sr = rope_sin + (size_t)pos * hd, *cr = rope_cos + (size_t)pos * hd;
    // f16 caches: the same element counts, half the bytes (the pointers are kept as float* in the f32 build only)
    float *kt = kcache + (size_t)kvh * kv_head_floats(max_pos) / (KV16 ? 2 : 1);  // [chunk][D][64]   | f16: [chunk][D/2][64][2]
    float *vc = vcache + (size_t)kvh * kv_head_floats(max_pos) / (KV16 ? 2 : 1);  // [max_pos][D]
    // ---- the few loads RoPE needs go first (vmcnt retires in order: behind the 64 cache loads they
    //      would only count as arrived once the whole K/V chunk has) ------------------------------------
    //      All unconditional (clamped indices): a load under a branch makes hipcc wait at the join.
    const int rg = wave, rj = lane;  // RoPE on q: 4 heads x 64 rotation pairs = 256 threads (every half does it; half 0 stores)
    const float *q_raw = qkv + (size_t)(kvh * group + (rg < group ? rg : group - 1)) * kD;
    const float *k_raw = qkv + (size_t)n_heads * kD + (size_t)kvh * kD;
    const float rs = sr[rj], rc = cr[rj], rq0 = q_raw[rj], rq1 = q_raw[hd + rj];        // q: pair rj of head rg
    const float rk0 = k_raw[rj], rk1 = k_raw[hd + rj];                                   // new key: pair rj (threads < 64 use it)
    const float rv = qkv[(size_t)(n_heads + n_kv) * kD + (size_t)kvh * kD + (t4 & 127)];  // new value: dim t4 & 127
    __builtin_amdgcn_sched_barrier(0);  // these seven requests first, then the cache stream
    // ---- every cache load of this thread is issued up front (none depends on q): the K slice for the
    //      score pass (lane = position) and the V column for the P.V pass (lane = dim).  Uniform base + one
    //      lane offset + immediates, no clamping: the whole 64-position tile is inside the allocation (the cache
    //      is padded to whole chunks, and a dead half re-reads its workgroup's first tile); positions at or past
    //      the new token hold stale bytes, discarded below ----
    float kv[KV16 ? 1 : 32], vv[KV16 ? 1 : kAttnChunk / 2];
    uint32_t kh[KV16 ? 16 : 1], vh[KV16 ? kAttnChunk / 4 : 1];  // f16 pairs: (dim 2i, 2i + 1) of the wave's slice / of the thread's dim pair
    {
        const int cl = live ? chunk : pc * NH, jl = cl * kAttnChunk;
        if (!KV16) {
            const float *kp = kt + ((size_t)cl * kD + 32 * wave) * 64 + lane;  // kidx(32 * wave, jl + lane)
#pragma unroll
            for (int i = 0; i < 32; ++i) kv[i] = __builtin_nontemporal_load(kp + i * 64);  // cache bytes are read once per token
            // P.V: thread (d = t4 & 127, parity hp = t4 >> 7) takes positions jl + hp, jl + hp + 2, ...:
            // element (jl + 2 i + hp) * D + d = jl * D + t4 + 2 i D
            const float *vp = vc + (size_t)jl * kD + t4;
#pragma unroll
            for (int i = 0; i < kAttnChunk / 2; ++i) vv[i] = __builtin_nontemporal_load(vp + i * 2 * kD);
        } else {
            // K: dword (dim pair 16 wave + i, position lane) of tile cl
            const uint32_t *kp = reinterpret_cast<const uint32_t *>(kt) + ((size_t)cl * (kD / 2) + 16 * wave) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) kh[i] = __builtin_nontemporal_load(kp + i * 64);
            // P.V: thread (dim pair dp = t4 & 63, class pq = t4 >> 6) takes positions jl + pq + 4 i: dword (jl + pq + 4 i) * 64 + dp
            const uint32_t *vp = reinterpret_cast<const uint32_t *>(vc) + (size_t)jl * (kD / 2) + t4;
#pragma unroll
            for (int i = 0; i < kAttnChunk / 4; ++i) vh[i] = __builtin_nontemporal_load(vp + i * 4 * (kD / 2));
        }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep hipcc from moving the RoPE arithmetic (and its wait) up between the loads
    BH_ASTAMP(1);  // every load of this thread is requested
    // ---- RoPE on the group's queries (and, in the owning half, on the new key) ----------
    if (half == 0) {
        float a = 0.0f, b = 0.0f;
        if (rg < group) {
            a = rq0 * rc - rq1 * rs;
            b = rq0 * rs + rq1 * rc;
        }
        qs[rg * kD + rj] = a;
        qs[rg * kD + hd + rj] = b;
    }
    // the new key / value go to LDS from EVERY thread of every workgroup (all waves write the same
    // values): used only under a branch, hipcc sinks their loads behind the cache stream, where the
    // in-order counter makes them as late as the last cache byte.  Only the owning half appends to the cache.
    {
        const float a = rk0 * rc - rk1 * rs, b = rk0 * rs + rk1 * rc;  // rotation pair rj of the new key
        kn[rj] = a;
        kn[hd + rj] = b;
        vn[t4 & 127] = rv;
        if (last) {
            if (!KV16) {
                if (t4 < hd) {
                    kt[kidx(t4, pos)] = a;  // append (transposed)
                    kt[kidx(hd + t4, pos)] = b;
                } else if (t4 >= 128) {
                    vc[(size_t)pos * kD + (t4 - 128)] = rv;
                }
            } else {
                _Float16 *k16 = reinterpret_cast<_Float16 *>(kt), *v16 = reinterpret_cast<_Float16 *>(vc);
                if (t4 < hd) {  // element (d, pos) at ((chunk * 64 + d / 2) * 64 + pos % 64) * 2 + d % 2
                    const size_t tb = (size_t)(pos >> 6) * (kD / 2), pp = pos & 63;
                    k16[((tb + (t4 >> 1)) * 64 + pp) * 2 + (t4 & 1)] = (_Float16)a;
                    k16[((tb + ((hd + t4) >> 1)) * 64 + pp) * 2 + (t4 & 1)] = (_Float16)b;
                } else if (t4 >= 128) {
                    v16[(size_t)pos * kD + (t4 - 128)] = (_Float16)rv;
                }
            }
        }
    }
    __syncthreads();
    BH_ASTAMP(2);  // q, the new key / value are in LDS (the seven RoPE operands have arrived)
    // ---- scores: lane = position, wave = 32-dim slice; all 32 loads of a thread in flight ---
    // Two partial sums per head (even / odd dims) so that each v_pk_fma_f32 takes an adjacent (q[d], q[d+1]) pair
    // from one LDS read and an adjacent (k[d], k[d+1]) register pair: no operand shuffling.
    if (KV16) {
        if (last) {  // the new token's key comes from LDS, rounded as the cache will hold it
            const bool isnew = j0 + lane == pos;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                float4 k4 = *reinterpret_cast<const float4 *>(kn + 32 * wave + 2 * i);
                asm volatile("" : "+v"(k4.x), "+v"(k4.y), "+v"(k4.z), "+v"(k4.w));
                const uint32_t p0 = pack_h2(k4.x, k4.y), p1 = pack_h2(k4.z, k4.w);
                kh[i] = isnew ? p0 : kh[i];
                kh[i + 1] = isnew ? p1 : kh[i + 1];
            }
        }
        float acc[kMaxGroup] = {0.0f, 0.0f, 0.0f, 0.0f}, acc2[kMaxGroup] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
#pragma unroll
            for (int g = 0; g < kMaxGroup; ++g) {
                const float4 q4 = *reinterpret_cast<const float4 *>(qs + g * kD + 32 * wave + 2 * i);
                acc[g] = amix_lo(q4.x, kh[i], acc[g]);
                acc2[g] = amix_hi(q4.y, kh[i], acc2[g]);
                acc[g] = amix_lo(q4.z, kh[i + 1], acc[g]);
                acc2[g] = amix_hi(q4.w, kh[i + 1], acc2[g]);
            }
        }
#pragma unroll
        for (int g = 0; g < kMaxGroup; ++g) partial[half][wave][lane][g] = acc[g] + acc2[g];
    } else {
        if (last) {  // the new token's key comes from LDS (its cache slot was read before it was written)
            const bool isnew = j0 + lane == pos;
#pragma unroll
            for (int i = 0; i < 32; i += 4) {
                float4 k4 = *reinterpret_cast<const float4 *>(kn + 32 * wave + i);
                asm volatile("" : "+v"(k4.x), "+v"(k4.y), "+v"(k4.z), "+v"(k4.w));  // read for all lanes, then select: no exec-mask branch
                kv[i] = isnew ? k4.x : kv[i];
                kv[i + 1] = isnew ? k4.y : kv[i + 1];
                kv[i + 2] = isnew ? k4.z : kv[i + 2];
                kv[i + 3] = isnew ? k4.w : kv[i + 3];
            }
        }
        v2f acc[kMaxGroup] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
        for (int i = 0; i < 32; i += 4) {
            const v2f k01 = {kv[i], kv[i + 1]}, k23 = {kv[i + 2], kv[i + 3]};
#pragma unroll
            for (int g = 0; g < kMaxGroup; ++g) {
                const float4 q4 = *reinterpret_cast<const float4 *>(qs + g * kD + 32 * wave + i);
                acc[g] = __builtin_elementwise_fma((v2f){q4.x, q4.y}, k01, acc[g]);
                acc[g] = __builtin_elementwise_fma((v2f){q4.z, q4.w}, k23, acc[g]);
            }
        }
#pragma unroll
        for (int g = 0; g < kMaxGroup; ++g) partial[half][wave][lane][g] = acc[g][0] + acc[g][1];
    }
    BH_ASTAMP(3);  // scores done: the K slice has arrived
    __syncthreads();
    // ---- chunk-local softmax pieces: wave g owns head g, lane = position ---------------------
    // (positions past the context carried stale keys: their scores, whatever they are, are replaced here;
    //  a dead half has t_k <= j0: every score becomes -inf, m = -inf, l = 0)
    const float scale = 1.0f / sqrtf((float)kD);
    float m_c, l_c;
    {
        const int g = wave, j = j0 + lane;
        float s = ((partial[half][0][lane][g] + partial[half][1][lane][g]) + partial[half][2][lane][g]) + partial[half][3][lane][g];
        s = j < t_k ? s * scale : -INFINITY;
        m_c = awave_max(s);
        const float e = j < t_k ? expf(s - m_c) : 0.0f;
        l_c = awave_sum(e);
        // the new token's value is not in the cache registers: its weight goes aside (enew) and its slot gets 0
        if (KV16)
            sc[half][g][lane & 3][lane >> 2] = j == pos ? 0.0f : e;
        else
            sc[half][g][lane & 1][lane >> 1] = j == pos ? 0.0f : e;
        if (j == pos) enew[g] = e;
        if (NH == 2 && lane == 0) {
            hm[half][g] = m_c;
            hl[half][g] = l_c;
        }
    }
    __syncthreads();
    BH_ASTAMP(4);  // softmax weights in LDS
    // ---- un-normalised P.V: thread = (dim d, position parity hp); V already in registers ---
    float *rec = scratch + ((size_t)kvh * gridDim.y + pc) * kRec;
    if (KV16) {
        const int dp = t4 & 63, pq = t4 >> 6;
        float a0[kMaxGroup] = {0.0f, 0.0f, 0.0f, 0.0f}, a1[kMaxGroup] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < kAttnChunk / 4; i += 4) {
#pragma unroll
            for (int g = 0; g < kMaxGroup; ++g) {
                const float4 w = *reinterpret_cast<const float4 *>(&sc[half][g][pq][i]);
                a0[g] = amix_lo(w.x, vh[i], a0[g]);
                a1[g] = amix_hi(w.x, vh[i], a1[g]);
                a0[g] = amix_lo(w.y, vh[i + 1], a0[g]);
                a1[g] = amix_hi(w.y, vh[i + 1], a1[g]);
                a0[g] = amix_lo(w.z, vh[i + 2], a0[g]);
                a1[g] = amix_hi(w.z, vh[i + 2], a1[g]);
                a0[g] = amix_lo(w.w, vh[i + 3], a0[g]);
                a1[g] = amix_hi(w.w, vh[i + 3], a1[g]);
            }
        }
        if (last && pq == 0) {  // the new token: value from LDS (rounded as the cache holds it), weight from enew
            const float v0 = (float)(_Float16)vn[2 * dp], v1 = (float)(_Float16)vn[2 * dp + 1];
            const float4 en = *reinterpret_cast<const float4 *>(enew);
            a0[0] += en.x * v0, a1[0] += en.x * v1;
            a0[1] += en.y * v0, a1[1] += en.y * v1;
            a0[2] += en.z * v0, a1[2] += en.z * v1;
            a0[3] += en.w * v0, a1[3] += en.w * v1;
        }
#pragma unroll
        for (int g = 0; g < kMaxGroup; ++g) {
            red[half][(pq * kMaxGroup + g) * kD + 2 * dp] = a0[g];
            red[half][(pq * kMaxGroup + g) * kD + 2 * dp + 1] = a1[g];
        }
        __syncthreads();
        auto rsum = [&](int hf, int g, int d) {
            return (red[hf][g * kD + d] + red[hf][(kMaxGroup + g) * kD + d]) + (red[hf][(2 * kMaxGroup + g) * kD + d] + red[hf][(3 * kMaxGroup + g) * kD + d]);
        };
        if (NH == 1) {
            const int d = t4 & 127, hp = t4 >> 7;
#pragma unroll
            for (int g = 2 * hp; g < 2 * hp + 2; ++g) rec[2 * kMaxGroup + g * kD + d] = rsum(0, g, d);
        } else {
            const int g = tid >> 7, d = tid & 127;
            const float m0 = hm[0][g], m1 = hm[1][g], M = fmaxf(m0, m1);
            const float e0 = __expf(m0 - M), e1 = __expf(m1 - M);
            rec[2 * kMaxGroup + g * kD + d] = e0 * rsum(0, g, d) + e1 * rsum(1, g, d);
            if (d == 0) {
                rec[2 * g] = M;
                rec[2 * g + 1] = e0 * hl[0][g] + e1 * hl[1][g];
            }
        }
    } else {
        const int d = t4 & 127, hp = t4 >> 7;
        // Cache slots at or past the new token hold stale bytes; their weights are exact zeros (the new token's own
        // goes through enew), and 0 * finite = 0: the caches must never hold NaN / Inf bit patterns, i.e. be
        // zero-filled before first use (include/bitnet_hip.h) -- then this loop needs no masking at all.
        v2f a[kMaxGroup] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
        for (int i = 0; i < kAttnChunk / 2; i += 4) {
            const v2f v01 = {vv[i], vv[i + 1]}, v23 = {vv[i + 2], vv[i + 3]};
#pragma unroll
            for (int g = 0; g < kMaxGroup; ++g) {
                const float4 w = *reinterpret_cast<const float4 *>(&sc[half][g][hp][i]);
                a[g] = __builtin_elementwise_fma((v2f){w.x, w.y}, v01, a[g]);
                a[g] = __builtin_elementwise_fma((v2f){w.z, w.w}, v23, a[g]);
            }
        }
        float ar[kMaxGroup];
#pragma unroll
        for (int g = 0; g < kMaxGroup; ++g) ar[g] = a[g][0] + a[g][1];
        if (last && hp == 0) {  // the new token: value from LDS, weight from enew (one parity adds it)
            const float vnd = vn[d];
            const float4 en = *reinterpret_cast<const float4 *>(enew);
            ar[0] += en.x * vnd;
            ar[1] += en.y * vnd;
            ar[2] += en.z * vnd;
            ar[3] += en.w * vnd;
        }
        // the position parities (and, NH == 2, the two halves) meet through LDS
#pragma unroll
        for (int g = 0; g < kMaxGroup; ++g) red[half][(hp * kMaxGroup + g) * kD + d] = ar[g];
        __syncthreads();
        if (NH == 1) {
#pragma unroll
            for (int g = 2 * hp; g < 2 * hp + 2; ++g) rec[2 * kMaxGroup + g * kD + d] = red[0][g * kD + d] + red[0][(kMaxGroup + g) * kD + d];
        } else {
            // 512 threads: head g = tid >> 7, dim d.  out = e^(m0 - M) o0 + e^(m1 - M) o1 with M = max(m0, m1); a dead
            // second half has m1 = -inf, l1 = 0, o1 = 0 * stale = 0
            const int g = tid >> 7;
            const float m0 = hm[0][g], m1 = hm[1][g], M = fmaxf(m0, m1);
            const float e0 = __expf(m0 - M), e1 = __expf(m1 - M);
            const float o0 = red[0][g * kD + d] + red[0][(kMaxGroup + g) * kD + d];
            const float o1 = red[1][g * kD + d] + red[1][(kMaxGroup + g) * kD + d];
            rec[2 * kMaxGroup + g * kD + d] = e0 * o0 + e1 * o1;
            if (d == 0) {
                rec[2 * g] = M;
                rec[2 * g + 1] = e0 * hl[0][g] + e1 * hl[1][g];
            }
        }
    }
    if (NH == 1 && lane == 0) {
        rec[2 * wave] = m_c;
        rec[2 * wave + 1] = l_c;
    }
    BH_ASTAMP(5);
}

// One workgroup per head: 8 thread groups walk the chunk records in parallel (chunk c -> group
// c % 8), each merging (m, l, o) online; the 8 partial states meet through LDS.  Threads of a
// group take 4 dims each (float4: one 512-byte record row per group and step).
// NR = records per thread group requested UP FRONT, before the position (hence the live record count) is known: the
// records were written by other CUs' workgroups, so every dependent trip is a cross-XCD miss (~0.7-1 us); with
// 8 * NR >= n_chunks_max the whole merge is ONE round trip.  Records past the context are valid memory (the scratch buffer
// is sized for max_pos and zero-filled once) and are dropped by their index.
template <int NR>
__global__ __launch_bounds__(256) void k_attn_combine(const float *__restrict__ scratch, int n_kv, int group,
                                                      int n_chunks_max, int chunk_log2, const int *__restrict__ pos_ptr,
                                                      float *__restrict__ out, uint8_t *__restrict__ qout) {
    asm volatile("" ::"s"(scratch), "s"(n_kv), "s"(group), "s"(n_chunks_max), "s"(chunk_log2), "s"(out), "s"(qout));  // all arguments in one scalar-load round
    const int kvh = blockIdx.x, g = blockIdx.y;
    const int tid = threadIdx.x, d4 = tid & 31, part = tid >> 5;
    __shared__ float sm[8], sl[8];
    __shared__ __attribute__((aligned(16))) float sa[8][kD];
    const float *base = scratch + (size_t)kvh * n_chunks_max * kRec;
    float mc[NR], lc[NR];
    float4 o[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int c = part + 8 * u;
        const float *rec = base + (size_t)(c < n_chunks_max ? c : n_chunks_max - 1) * kRec;
        mc[u] = rec[2 * g];
        lc[u] = rec[2 * g + 1];
        o[u] = *reinterpret_cast<const float4 *>(rec + 2 * kMaxGroup + g * kD + 4 * d4);
    }
    const int t_k = *pos_ptr + 1;
    const int n_chunks = (t_k + (1 << chunk_log2) - 1) >> chunk_log2;  // records of 64 or 128 positions
    float m = -INFINITY, l = 0.0f;
    float4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        if (part + 8 * u < n_chunks) {  // records merge in index order: the same value whatever NR is
            const float m_new = fmaxf(m, mc[u]);
            const float s_old = expf(m - m_new), s_c = expf(mc[u] - m_new);
            l = l * s_old + lc[u] * s_c;
            a.x = a.x * s_old + o[u].x * s_c;
            a.y = a.y * s_old + o[u].y * s_c;
            a.z = a.z * s_old + o[u].z * s_c;
            a.w = a.w * s_old + o[u].w * s_c;
            m = m_new;
        }
    }
    // contexts beyond 8 * NR records: further trips, four records each
    for (int c = part + 8 * NR; c < n_chunks; c += 32) {
        float mc2[4], lc2[4];
        float4 o2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cu = c + 8 * u < n_chunks ? c + 8 * u : c;  // past the end: re-read this trip's first record, dropped below
            const float *rec = base + (size_t)cu * kRec;
            mc2[u] = rec[2 * g];
            lc2[u] = rec[2 * g + 1];
            o2[u] = *reinterpret_cast<const float4 *>(rec + 2 * kMaxGroup + g * kD + 4 * d4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c + 8 * u < n_chunks) {
                const float m_new = fmaxf(m, mc2[u]);
                const float s_old = expf(m - m_new), s_c = expf(mc2[u] - m_new);
                l = l * s_old + lc2[u] * s_c;
                a.x = a.x * s_old + o2[u].x * s_c;
                a.y = a.y * s_old + o2[u].y * s_c;
                a.z = a.z * s_old + o2[u].z * s_c;
                a.w = a.w * s_old + o2[u].w * s_c;
                m = m_new;
            }
        }
    }
    if (d4 == 0) {
        sm[part] = m;
        sl[part] = l;
    }
    *reinterpret_cast<float4 *>(&sa[part][4 * d4]) = a;
    __syncthreads();
    if (tid < kD) {
        float M = sm[0];
#pragma unroll
        for (int p = 1; p < 8; ++p) M = fmaxf(M, sm[p]);  // chunk 0 always exists: M is finite
        float L = 0.0f, acc = 0.0f;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float w = expf(sm[p] - M);  // groups without a chunk: exp(-inf) = 0
            L += w * sl[p];
            acc += w * sa[p][tid];
        }
        const float v = acc / L;
        const int col = (kvh * group + g) * kD + tid;
        if (out) out[col] = v;
        // the o-projection's input as a QAct (qact.hpp): 16 consecutive dims = one group, no LayerNorm in between (T:542)
        if (qout) qact_emit(qout, nullptr, col >> 4, col & 15, v, v);
    }
}

size_t attn_scratch_floats(int n_kv, int max_pos) {
    return (size_t)n_kv * ((max_pos + kAttnChunk - 1) / kAttnChunk) * kRec;
}

hipError_t launch_attn_decode(const float *qkv, const float *rope_sin, const float *rope_cos, float *kcache,
                              float *vcache, int n_heads, int n_kv, int D, int max_pos, const int *pos_ptr,
                              float *scratch, float *out, hipStream_t stream, bool combine, int halves, void *qout, int kv_f16) {
    if (D != kD || n_heads / n_kv > kMaxGroup || (halves != 1 && halves != 2)) return hipErrorInvalidValue;
    const int rec_pos = kAttnChunk * halves;  // positions per record
    const int n_rec = (max_pos + rec_pos - 1) / rec_pos;
    auto kfn = halves == 2 ? (kv_f16 ? k_attn_partial<2, true> : k_attn_partial<2, false>) : (kv_f16 ? k_attn_partial<1, true> : k_attn_partial<1, false>);
    hipLaunchKernelGGL(kfn, dim3(n_kv, n_rec), dim3(256 * halves), 0, stream, qkv, rope_sin, rope_cos, kcache, vcache, n_heads, n_kv, n_heads / n_kv,
                       max_pos, pos_ptr, scratch, g_mfma_stamps);
    // combine == false: the chunk records stay in `scratch` for a consumer that merges them itself
    // (launch_gemv_mfma with GemvFusion::attn_rec)
    if (combine) {
        auto cfn = n_rec <= 8 ? k_attn_combine<1> : n_rec <= 16 ? k_attn_combine<2> : n_rec <= 40 ? k_attn_combine<5> : k_attn_combine<8>;
        hipLaunchKernelGGL(cfn, dim3(n_kv, n_heads / n_kv), dim3(256), 0, stream, scratch, n_kv, n_heads / n_kv, n_rec, halves == 2 ? 7 : 6, pos_ptr, out,
                           static_cast<uint8_t *>(qout));
    }
    return hipGetLastError();
}

}  // namespace bitnet_hip
