// l2_prefetch_probe.hip -- does pulling a GEMV's weight bytes into the consuming XCD's L2 from an
// EARLIER kernel shorten the GEMV's stream phase?
//
// Consumer C models the batch-1 GEMV stream: G workgroups x 512 threads, workgroup b reads the
// contiguous chunk b of a matrix (all loads in flight at once, non-temporal), folds it and writes 16
// floats.  Prefetcher P: workgroup i reads the chunks of consumer workgroups b == i (mod 8) -- a
// dispatch places workgroup n on XCD n % 8, so the lines land in the L2 the consumer will ask.
//   mode 0: C only (cold: M matrices in rotation, M x bytes > MALL)
//   mode 1: P(mat) then C(mat)
//   mode 2: P with the XCD mapping shifted by one (lines land in the WRONG L2) then C
// Per-kernel durations come from s_memrealtime stamps (100 MHz) written by every workgroup:
// span = max(end) - min(start) over the consumer's workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_prefetch_probe tools/probes/l2_prefetch_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                  \
    do {                                                                        \
        hipError_t e = (x);                                                     \
        if (e != hipSuccess) {                                                  \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                            \
        }                                                                       \
    } while (0)

__device__ inline uint4 load_nt(const uint4 *p) {
    uint4 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    v.z = __builtin_nontemporal_load(&p->z);
    v.w = __builtin_nontemporal_load(&p->w);
    return v;
}

// chunk_vec = uint4 per chunk, a multiple of 512
template <int NT>
__global__ __launch_bounds__(512) void k_consume(const uint4 *__restrict__ w, int chunk_vec, float *out, unsigned long long *stamps) {
    const int tid = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const uint4 *base = w + (size_t)blockIdx.x * chunk_vec;
    unsigned acc = 0;
    for (int i = tid; i < chunk_vec; i += 512 * 8) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = i + j * 512;
            const uint4 *p = base + (idx < chunk_vec ? idx : tid);
            v[j] = NT ? load_nt(p) : *p;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    __shared__ unsigned red[512];
    red[tid] = acc;
    __syncthreads();
    if (tid < 16) {
        unsigned a = 0;
        for (int i = tid; i < 512; i += 16) a ^= red[i];
        out[blockIdx.x * 16 + tid] = (float)(a & 0xff);
    }
    __syncthreads();
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

__global__ __launch_bounds__(512) void k_prefetch(const uint4 *__restrict__ w, int chunk_vec, int n_chunks, int shift, unsigned *sink) {
    const int tid = threadIdx.x;
    const int xcd = (blockIdx.x + shift) & 7, slot = blockIdx.x >> 3, n_slots = gridDim.x >> 3;
    unsigned acc = 0;
    // chunks b with b % 8 == xcd, dealt round-robin to this XCD's prefetch workgroups
    for (int b = xcd + 8 * slot; b < n_chunks; b += 8 * n_slots) {
        const uint4 *base = w + (size_t)b * chunk_vec;
        for (int i = tid; i < chunk_vec; i += 512 * 8) {
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = i + j * 512;
                v[j] = base[idx < chunk_vec ? idx : tid];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        }
    }
    if (acc == 0x9e3779b9u && tid == 77) sink[0] = acc;  // keeps the loads alive
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 216;          // consumer workgroups
    const size_t bytes = argc > 2 ? atol(argv[2]) : 11404800;  // gate|up tiles + f16 scales at c2
    const int PG = argc > 3 ? atoi(argv[3]) : 256;         // prefetch workgroups
    const int M = 30, reps = 6;
    int chunk_vec = (int)((bytes / G + 16 * 512 - 1) / (16 * 512)) * 512;
    const size_t mat_vec = (size_t)chunk_vec * G;
    printf("G=%d chunk=%d B matrix=%zu B prefetch WGs=%d\n", G, chunk_vec * 16, mat_vec * 16, PG);
    uint4 *w;
    CHK(hipMalloc(&w, mat_vec * 16 * M));
    CHK(hipMemset(w, 1, mat_vec * 16 * M));
    float *out;
    CHK(hipMalloc(&out, G * 16 * 4));
    unsigned *sink;
    CHK(hipMalloc(&sink, 64));
    unsigned long long *stamps;
    CHK(hipMalloc(&stamps, (size_t)M * reps * G * 2 * 8));
    hipStream_t s;
    CHK(hipStreamCreate(&s));
    std::vector<unsigned long long> h((size_t)M * reps * G * 2);
    for (int nt = 1; nt >= 0; --nt)
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1;
            CHK(hipEventCreate(&e0));
            CHK(hipEventCreate(&e1));
            CHK(hipStreamSynchronize(s));
            CHK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r)
                for (int m = 0; m < M; ++m) {
                    const uint4 *mat = w + (size_t)m * mat_vec;
                    if (mode) k_prefetch<<<PG, 512, 0, s>>>(mat, chunk_vec, G, mode == 2 ? 1 : 0, sink);
                    unsigned long long *st = stamps + ((size_t)r * M + m) * G * 2;
                    if (nt)
                        k_consume<1><<<G, 512, 0, s>>>(mat, chunk_vec, out, st);
                    else
                        k_consume<0><<<G, 512, 0, s>>>(mat, chunk_vec, out, st);
                }
            CHK(hipEventRecord(e1, s));
            CHK(hipStreamSynchronize(s));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            CHK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> spans;
            for (int k = M; k < M * reps; ++k) {  // first rep = warm-up
                unsigned long long lo = ~0ull, hi = 0;
                for (int b = 0; b < G; ++b) {
                    lo = std::min(lo, h[((size_t)k * G + b) * 2]);
                    hi = std::max(hi, h[((size_t)k * G + b) * 2 + 1]);
                }
                spans.push_back((hi - lo) * 0.01);
            }
            std::sort(spans.begin(), spans.end());
            printf("nt=%d mode %d: consumer span median %.2f us (p10 %.2f, p90 %.2f); stream total %.2f us per matrix\n", nt, mode,
                   spans[spans.size() / 2], spans[spans.size() / 10], spans[spans.size() * 9 / 10], ms * 1e3 / (M * reps));
        }
    return 0;
}
