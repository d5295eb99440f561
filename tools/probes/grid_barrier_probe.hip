// grid_barrier_probe.hip -- cost of a software grid barrier on MI355X (256 CUs, 8 XCDs):
// G workgroups (one per CU, cooperative launch) run NB barriers; between barriers every
// workgroup publishes a value the others read back (cross-XCD visibility check).
//   variant 0: one global counter; variant 1: 8 per-XCD counters (blockIdx % 8) + one top counter.
// Every spin is bounded (never hangs the GPU).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x)                                                                  \
    do {                                                                        \
        hipError_t e = (x);                                                     \
        if (e != hipSuccess) {                                                  \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                            \
        }                                                                       \
    } while (0)

__device__ __forceinline__ bool spin_until(unsigned *p, unsigned target, unsigned *err) {
    unsigned it = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++it > 200000u) {
            atomicAdd(err, 1u);
            return false;
        }
    }
    return true;
}

__global__ __launch_bounds__(512) void k_barriers(unsigned *ctr, float *data, int nb, int variant, unsigned *err, unsigned *bad) {
    const int tid = threadIdx.x, G = gridDim.x, b = blockIdx.x;
    unsigned *top = ctr, *xcd = ctr + 64 + 64 * (b & 7);
    const int per_xcd = G / 8;
    float seen = 0.f;
    for (int i = 0; i < nb; ++i) {
        // publish
        if (tid < 16) data[(size_t)(i & 1) * G * 16 + b * 16 + tid] = (float)(i * 1000 + b);
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (variant == 0) {
                __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                spin_until(top, (unsigned)(i + 1) * G, err);
            } else {
                const unsigned old = __hip_atomic_fetch_add(xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned)(i + 1) * per_xcd - 1) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                spin_until(top, (unsigned)(i + 1) * 8, err);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // read what another workgroup (other XCD) published
        const int o = (b + 1 + (i % 7)) % G;
        if (tid < 16) {
            const float v = __builtin_nontemporal_load(&data[(size_t)(i & 1) * G * 16 + o * 16 + tid]);
            if (v != (float)(i * 1000 + o)) atomicAdd(bad, 1u);
            seen += v;
        }
    }
    if (seen == -1.f) data[0] = seen;
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 256, nb = 2000;
    unsigned *ctr, *err, *bad;
    float *data;
    CHK(hipMalloc(&ctr, 4096 * 4));
    CHK(hipMalloc(&err, 4));
    CHK(hipMalloc(&bad, 4));
    CHK(hipMalloc(&data, (size_t)2 * G * 16 * 4));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int variant = 0; variant < 2; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipMemset(ctr, 0, 4096 * 4));
            CHK(hipMemset(err, 0, 4));
            CHK(hipMemset(bad, 0, 4));
            CHK(hipMemset(data, 0, (size_t)2 * G * 16 * 4));
            hipEvent_t e0, e1;
            CHK(hipEventCreate(&e0));
            CHK(hipEventCreate(&e1));
            int nbv = nb, var = variant;
            void *args[] = {&ctr, &data, &nbv, &var, &err, &bad};
            CHK(hipEventRecord(e0, s));
            CHK(hipLaunchCooperativeKernel((void *)k_barriers, dim3(G), dim3(512), args, 0, s));
            CHK(hipEventRecord(e1, s));
            CHK(hipStreamSynchronize(s));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned he = 0, hb = 0;
            CHK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            printf("variant %d G=%d: %.3f us per barrier (+publish/readback), spin-timeouts %u, stale reads %u\n", variant, G, ms * 1e3 / nb, he, hb);
            fflush(stdout);
        }
    }
    return 0;
}
