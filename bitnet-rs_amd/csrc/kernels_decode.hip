// kernels_decode.hip -- the decode-step operators around the I2_S GEMVs (gfx950):
// embedding gather, LayerNorm / RMSNorm rows, RoPE + KV append + GQA attention for
// one new token, tied-embedding logits (f16 table, f32 accumulate) with fused final
// norm and greedy argmax.  Semantics follow the reference's transformer
// (T = crates/bitnet-transformer/src/lib.rs); see oracle/transformer_oracle.c.
//
// Position and token live in device memory (pos_ptr / token_ptr) so one captured
// hipGraph replays for every decode step without host round trips.
#include <cstdlib>

#include "common.hpp"

namespace bitnet_hip {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// 256-thread block reductions; slot = 4 floats of LDS.
__device__ __forceinline__ float bsum(float v, float *slot) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    return (slot[0] + slot[1]) + (slot[2] + slot[3]);
}
__device__ __forceinline__ float bmax(float v, float *slot) {
    v = wmax(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(slot[0], slot[1]), fmaxf(slot[2], slot[3]));
}

// ---- embedding: row gather from the f16 table (T:1415-1424) -------------------------
__global__ void k_embed_f16(const _Float16 *__restrict__ table, const int *__restrict__ tokens,
                            const int *__restrict__ offset_ptr, int n, int hidden, int vocab,
                            float *__restrict__ out) {
    const int per_row = hidden >> 3;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * per_row) return;
    const int t = gid / per_row, c = gid % per_row;
    int tok = tokens[t + (offset_ptr ? *offset_ptr : 0)];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    const half8 h = *reinterpret_cast<const half8 *>(table + (size_t)tok * hidden + 8 * c);
    float *o = out + (size_t)t * hidden + 8 * c;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)h[i];
}

hipError_t launch_embed_f16(const void *table, const int *tokens, const int *offset_ptr, int n, int hidden,
                            int vocab, float *out, hipStream_t stream) {
    const int total = n * (hidden >> 3);
    hipLaunchKernelGGL(k_embed_f16, dim3((total + 255) / 256), dim3(256), 0, stream,
                       static_cast<const _Float16 *>(table), tokens, offset_ptr, n, hidden, vocab, out);
    return hipGetLastError();
}

// ---- LayerNorm (no bias, mean-subtracting; T:67-100) / RMSNorm (K/rocm/rmsnorm.rs) -----
// One 256-thread workgroup per row.
template <bool RMS>
__global__ __launch_bounds__(256) void k_norm_rows(const float *__restrict__ x, const float *__restrict__ gamma,
                                                   float *__restrict__ out, int hidden, float eps) {
    __shared__ float slot[4];
    const float *xr = x + (size_t)blockIdx.x * hidden;
    float *orow = out + (size_t)blockIdx.x * hidden;
    float mean = 0.0f;
    if (!RMS) {
        float s = 0.0f;
        for (int i = threadIdx.x; i < hidden; i += 256) s += xr[i];
        mean = bsum(s, slot) / (float)hidden;
    }
    float ss = 0.0f;
    for (int i = threadIdx.x; i < hidden; i += 256) {
        const float d = xr[i] - mean;
        ss += d * d;
    }
    const float denom = sqrtf(bsum(ss, slot) / (float)hidden + eps);
    for (int i = threadIdx.x; i < hidden; i += 256) orow[i] = (xr[i] - mean) / denom * gamma[i];
}

hipError_t launch_norm_rows(const float *x, const float *gamma, float *out, int rows, int hidden, float eps,
                            bool rms, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    if (rms)
        hipLaunchKernelGGL(k_norm_rows<true>, dim3(rows), dim3(256), 0, stream, x, gamma, out, hidden, eps);
    else
        hipLaunchKernelGGL(k_norm_rows<false>, dim3(rows), dim3(256), 0, stream, x, gamma, out, hidden, eps);
    return hipGetLastError();
}

// ---- tied-embedding logits (T:1599-1630) with fused final LayerNorm and per-workgroup
//      argmax partials (crates/bitnet-cli/src/sampling.rs:189-202) ----------------------------
// logits[v] = sum_k LN(x)[k] * (float)E[v,k].  One wave per vocabulary row, grid-strided;
// each lane keeps its slice of LN(x) in registers (hidden <= 8192).
constexpr int kLogitChunks = 16;  // 512 columns each

// NCH = hidden / 512 (compile time: the activation slice lives in NCH*8 registers), R = vocabulary rows
// per wave iteration (R * hidden * 2 bytes in flight per wave).
// GUARD: hidden / 512 may be smaller than NCH (generic instance), chunks past it are skipped.
template <int NCH, int R, bool GUARD = false>
__global__ __launch_bounds__(256) void k_logits_f16(const _Float16 *__restrict__ table, const float *__restrict__ x,
                                                    const float *__restrict__ gamma, float eps, int hidden, int vocab,
                                                    float *__restrict__ logits, float *__restrict__ best_val,
                                                    int *__restrict__ best_idx) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [hidden] normalised activations
    __shared__ float slot[4];
    __shared__ float wbv[4];
    __shared__ int wbi[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // final norm (T:1589), recomputed per workgroup: 10 KB from L2
    float s = 0.0f;
    for (int i = tid; i < hidden; i += 256) s += x[i];
    const float mean = gamma ? bsum(s, slot) / (float)hidden : 0.0f;
    float ss = 0.0f;
    for (int i = tid; i < hidden; i += 256) {
        const float d = x[i] - mean;
        ss += d * d;
    }
    const float denom = gamma ? sqrtf(bsum(ss, slot) / (float)hidden + eps) : 1.0f;
    for (int i = tid; i < hidden; i += 256) xs[i] = gamma ? (x[i] - mean) / denom * gamma[i] : x[i];
    __syncthreads();
    const int nchunks = hidden >> 9;
    float xr[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[c][i] = (!GUARD || c < nchunks) ? xs[512 * c + 8 * lane + i] : 0.0f;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    const int total_waves = gridDim.x * 4;
    for (int row = (blockIdx.x * 4 + wave) * R; row < vocab; row += total_waves * R) {
        half8 w[R][NCH];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // the table is read once per token by one wave: non-temporal (MI355X_MICROARCH.md, nt-weights);
            // rows past the end re-read the last row (never stored)
            const _Float16 *e = table + (size_t)(row + r < vocab ? row + r : vocab - 1) * hidden + 8 * lane;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const half8 *>(e + 512 * ((!GUARD || c < nchunks) ? c : 0)));
        }
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r] = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[r] += xr[c][i] * (float)w[r][c][i];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] += __shfl_xor(acc[r], off, 64);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (row + r < vocab) {
                    logits[row + r] = acc[r];
                    const float v = acc[r] != acc[r] ? -INFINITY : acc[r];  // NaN -> -inf (sampling.rs:45-49)
                    if (v > bv || (v == bv && row + r < bi)) {
                        bv = v;
                        bi = row + r;
                    }
                }
        }
    }
    if (lane == 0) {
        wbv[wave] = bv;
        wbi[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w2 = 1; w2 < 4; ++w2)
            if (wbv[w2] > bv || (wbv[w2] == bv && wbi[w2] < bi)) {
                bv = wbv[w2];
                bi = wbi[w2];
            }
        best_val[blockIdx.x] = bv;
        best_idx[blockIdx.x] = bi;
    }
}

// Reduce the per-workgroup partials, write the token, advance the position.
__global__ __launch_bounds__(256) void k_argmax_final(const float *__restrict__ best_val, const int *__restrict__ best_idx,
                                                      int n, int *__restrict__ token_out, int *__restrict__ pos_ptr,
                                                      int *__restrict__ history, const int *__restrict__ n_forced) {
    __shared__ float sv[256];
    __shared__ int si[256];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = best_val[i];
        const int idx = best_idx[i];
        if (v > bv || (v == bv && idx < bi)) {
            bv = v;
            bi = idx;
        }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float v = sv[threadIdx.x + off];
            const int idx = si[threadIdx.x + off];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && idx < si[threadIdx.x])) {
                sv[threadIdx.x] = v;
                si[threadIdx.x] = idx;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int tok = si[0] == 0x7fffffff ? 0 : si[0];
        if (token_out) *token_out = tok;
        if (pos_ptr) {
            const int p = *pos_ptr;
            // the token at position p+1 is the prediction unless the caller forces it (prompt)
            if (history && (!n_forced || p + 1 >= *n_forced)) history[p + 1] = tok;
            *pos_ptr = p + 1;
        }
    }
}

hipError_t launch_logits_f16(const void *table, const float *x, const float *gamma, float eps, int hidden, int vocab,
                             float *logits, float *best_val, int *best_idx, int n_wg, hipStream_t stream) {
    if (hidden % 512 != 0 || hidden > 512 * kLogitChunks) return hipErrorInvalidValue;
    void (*kfn)(const _Float16 *, const float *, const float *, float, int, int, float *, float *, int *) = nullptr;
    static const int rows = getenv("BITNET_HIP_LOGIT_ROWS") ? atoi(getenv("BITNET_HIP_LOGIT_ROWS")) : 3;  // tuning knob, read once
    switch (hidden / 512) {
        case 1: kfn = k_logits_f16<1, 4>; break;
        case 2: kfn = k_logits_f16<2, 4>; break;
        case 4: kfn = k_logits_f16<4, 4>; break;
        case 5: kfn = rows == 2 ? k_logits_f16<5, 2> : rows == 3 ? k_logits_f16<5, 3> : k_logits_f16<5, 4>; break;  // 2560: bitnet-b1.58-2B-4T
        case 8: kfn = k_logits_f16<8, 2>; break;
        default: kfn = hidden / 512 <= 8 ? k_logits_f16<8, 2, true> : k_logits_f16<16, 1, true>; break;
    }
    hipLaunchKernelGGL(kfn, dim3(n_wg), dim3(256), (size_t)hidden * sizeof(float), stream, static_cast<const _Float16 *>(table), x, gamma,
                       eps, hidden, vocab, logits, best_val, best_idx);
    return hipGetLastError();
}

hipError_t launch_argmax_final(const float *best_val, const int *best_idx, int n, int *token_out, int *pos_ptr,
                               int *history, const int *n_forced, hipStream_t stream) {
    hipLaunchKernelGGL(k_argmax_final, dim3(1), dim3(256), 0, stream, best_val, best_idx, n, token_out, pos_ptr, history,
                       n_forced);
    return hipGetLastError();
}

// Advance the position without sampling (prompt positions whose logits nobody reads).
// ---- elementwise steps of the UNFUSED decode step (the reference's own op order: T:1073 residual add,
//      T:765-781 silu(gate) * up); the fast path fuses both into the GEMV epilogues -----------------------
__global__ void k_add(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
hipError_t launch_add(const float *a, const float *b, float *out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, out, n);
    return hipGetLastError();
}
__global__ void k_silu_mul(const float *__restrict__ gate, const float *__restrict__ up, float *__restrict__ out, size_t n, size_t tile) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t j = tile ? (i / tile) * 2 * tile + i % tile : i;
    const float g = gate[j], u = up[j];
    out[i] = g / (1.0f + expf(-g)) * u;
}
hipError_t launch_silu_mul(const float *gate, const float *up, float *out, size_t n, size_t tile, hipStream_t stream) {
    hipLaunchKernelGGL(k_silu_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, gate, up, out, n, tile);
    return hipGetLastError();
}

__global__ void k_advance_pos(int *pos_ptr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *pos_ptr = *pos_ptr + 1;
}
hipError_t launch_advance_pos(int *pos_ptr, hipStream_t stream) {
    hipLaunchKernelGGL(k_advance_pos, dim3(1), dim3(64), 0, stream, pos_ptr);
    return hipGetLastError();
}

// Plain argmax over a logits vector (host-visible entry point).
__global__ __launch_bounds__(256) void k_argmax_partial(const float *__restrict__ v, int n, float *__restrict__ best_val,
                                                        int *__restrict__ best_idx) {
    __shared__ float sv[256];
    __shared__ int si[256];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float a = v[i] != v[i] ? -INFINITY : v[i];
        if (a > bv || (a == bv && i < bi)) {
            bv = a;
            bi = i;
        }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float a = sv[threadIdx.x + off];
            const int idx = si[threadIdx.x + off];
            if (a > sv[threadIdx.x] || (a == sv[threadIdx.x] && idx < si[threadIdx.x])) {
                sv[threadIdx.x] = a;
                si[threadIdx.x] = idx;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        best_val[blockIdx.x] = sv[0];
        best_idx[blockIdx.x] = si[0];
    }
}

hipError_t launch_argmax(const float *v, int n, float *best_val, int *best_idx, int n_wg, int *token_out,
                         hipStream_t stream) {
    hipLaunchKernelGGL(k_argmax_partial, dim3(n_wg), dim3(256), 0, stream, v, n, best_val, best_idx);
    hipLaunchKernelGGL(k_argmax_final, dim3(1), dim3(256), 0, stream, best_val, best_idx, n_wg, token_out, nullptr, nullptr, nullptr);
    return hipGetLastError();
}

// ---- measured HBM read ceiling (bitnet_hip_hbm_read_ceiling) ----------------
// Read-only stream: every thread keeps 8 non-temporal 16-byte loads in flight, a workgroup walks a
// 64 KiB span per step, the grid covers the buffer in a handful of steps.  The fold keeps the loads.
__global__ __launch_bounds__(512) void k_stream_read(const uint4 *__restrict__ buf, size_t n_vec, unsigned *sink) {
    const size_t stride = (size_t)gridDim.x * 512 * 8;
    unsigned acc = 0;
    for (size_t i = ((size_t)blockIdx.x * 8) * 512 + threadIdx.x; i < n_vec; i += stride) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t idx = i + (size_t)j * 512;
            const unsigned *p = reinterpret_cast<const unsigned *>(buf + (idx < n_vec ? idx : i));
            v[j].x = __builtin_nontemporal_load(p);
            v[j].y = __builtin_nontemporal_load(p + 1);
            v[j].z = __builtin_nontemporal_load(p + 2);
            v[j].w = __builtin_nontemporal_load(p + 3);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;  // (practically) never true
}

hipError_t launch_stream_read(const void *buf, size_t bytes, unsigned *sink, hipStream_t stream) {
    const size_t n_vec = bytes / 16;
    size_t grid = div_ceil(n_vec, (size_t)512 * 8 * 4);  // ~4 steps per workgroup
    if (grid < 256) grid = 256;
    if (grid > 65535u * 16u) grid = 65535u * 16u;
    hipLaunchKernelGGL(k_stream_read, dim3((unsigned)grid), dim3(512), 0, stream, static_cast<const uint4 *>(buf), n_vec, sink);
    return hipGetLastError();
}

}  // namespace bitnet_hip
