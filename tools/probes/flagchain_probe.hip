// flagchain_probe.hip -- two hipGraphs on two streams (even / odd kernels of a dependent chain), the data dependency
// kernel i-1 -> i carried by a device-side counter instead of a graph edge: kernel i requests its "weights" (NL x 16 B per
// thread, independent of its predecessor) BEFORE it waits, so launch, kernel-argument fetch, instruction-cache warm-up
// and the weight stream's first-byte latency overlap the predecessor.  (overlap_probe.hip showed that ONE graph with
// edges i-2 -> i is serialised by the runtime; two graphs on two streams are two hardware queues.)
//   mode 0: one graph, edges i-1 -> i (what the decode graph does today)
//   mode 1: two graphs on two streams + flags, agent-scope release / acquire fences (L2 write-back / invalidate)
//   mode 2: the same with write-through stores and cache-bypassing loads for the exchanged vector instead of fences
// Flags are never reset: kernel i waits for flags[i-1] >= G * epoch, epoch = its own chain's launch counter (bumped by
// a one-thread kernel at the head of each graph; in-order inside a chain, so every kernel of the chain sees it).
// Every spin is bounded (an error counter is bumped): the probe cannot hang the GPU.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/flagchain_probe tools/probes/flagchain_probe.hip
#include <hip/hip_runtime.h>

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

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ void k_epoch(unsigned *epoch) { *epoch += 1; }

template <int NL>
__global__ __launch_bounds__(512) void k_step(const v4u *__restrict__ w, const float *x_in, float *x_out, unsigned *flags, int idx,
                                              unsigned g_prev, const unsigned *epoch, int n, unsigned *err, unsigned long long *stamps, int bypass) {
    const int tid = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    v4u v[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) v[j] = __builtin_nontemporal_load(w + ((size_t)blockIdx.x * NL + j) * 512 + tid);  // independent of the predecessor
    unsigned long long t1 = t0;
    if (epoch && idx > 0) {
        if (tid == 0) {
            const unsigned expect = g_prev * __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned it = 0;
            while (__hip_atomic_load(&flags[idx - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
                if (++it > 5000u) {  // bounded: never hang the GPU
                    atomicAdd(err, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!bypass) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            t1 = __builtin_amdgcn_s_memrealtime();
        }
        __syncthreads();
    }
    float acc = 0.f;
    if (bypass) {  // the predecessor's vector straight from memory (cache-bypassing loads): no L2 invalidate needed
        for (int i = tid; i < n; i += 512) acc += __hip_atomic_load(x_in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        for (int i = tid; i < n; i += 512) acc += __builtin_nontemporal_load(x_in + i);
    }
    unsigned fold = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) fold ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    acc += (float)(fold & 1u);
    __shared__ float red[8];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    const float ov = (red[0] + red[1] + red[2] + red[3] + red[4] + red[5] + red[6] + red[7]) * 1e-9f + (float)tid;
    if (tid < 16) {
        if (bypass)
            __hip_atomic_store(x_out + blockIdx.x * 16 + tid, ov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // write-through
        else
            x_out[blockIdx.x * 16 + tid] = ov;
    }
    if (epoch) {
        if (bypass)
            __builtin_amdgcn_s_waitcnt(0);  // the write-through stores are acknowledged
        else
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&flags[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0 && blockIdx.x == 0 && stamps) {
        stamps[3 * idx] = t0;
        stamps[3 * idx + 1] = t1;
        stamps[3 * idx + 2] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int NL>
static void run(int G) {
    const int K = 150, n = 2560, reps = 20;
    v4u *w;
    float *xa, *xb;
    unsigned *flags, *err, *epochs;
    unsigned long long *stamps;
    const size_t wbytes = (size_t)G * 512 * 16 * NL;
    CHK(hipMalloc(&w, wbytes * K));
    CHK(hipMemset(w, 0, wbytes * K));
    CHK(hipMalloc(&xa, 65536));
    CHK(hipMalloc(&xb, 65536));
    CHK(hipMemset(xa, 0, 65536));
    CHK(hipMemset(xb, 0, 65536));
    CHK(hipMalloc(&flags, K * 4));
    CHK(hipMemset(flags, 0, K * 4));
    CHK(hipMalloc(&err, 4));
    CHK(hipMemset(err, 0, 4));
    CHK(hipMalloc(&epochs, 256));
    CHK(hipMemset(epochs, 0, 256));
    CHK(hipMalloc(&stamps, K * 24));
    hipStream_t s[2];
    CHK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    for (int mode = 0; mode < 3; ++mode) {
        const int ng = mode == 0 ? 1 : 2;
        CHK(hipMemset(epochs, 0, 256));
        CHK(hipMemset(flags, 0, K * 4));
        CHK(hipDeviceSynchronize());
        hipGraph_t g[2];
        hipGraphExec_t ex[2];
        for (int c = 0; c < ng; ++c) {
            CHK(hipGraphCreate(&g[c], 0));
            hipGraphNode_t prev;
            unsigned *ep = epochs + 32 * c;
            {
                void *args[] = {&ep};
                hipKernelNodeParams kp = {};
                kp.func = (void *)k_epoch;
                kp.gridDim = dim3(1);
                kp.blockDim = dim3(1);
                kp.kernelParams = args;
                CHK(hipGraphAddKernelNode(&prev, g[c], nullptr, 0, &kp));
            }
            for (int i = c; i < K; i += ng) {
                const v4u *wi = w + (size_t)i * G * 512 * NL;
                const float *xin = (i & 1) ? xb : xa;
                float *xout = (i & 1) ? xa : xb;
                unsigned *fl = flags;
                const unsigned *epp = mode >= 1 ? ep : nullptr;
                int bypass = mode == 2;
                int idx = i, nn = n;
                unsigned gp = (unsigned)G;
                unsigned long long *st = stamps;
                void *args[] = {&wi, &xin, &xout, &fl, &idx, &gp, &epp, &nn, &err, &st, &bypass};
                hipKernelNodeParams kp = {};
                kp.func = (void *)k_step<NL>;
                kp.gridDim = dim3(G);
                kp.blockDim = dim3(512);
                kp.kernelParams = args;
                hipGraphNode_t node;
                CHK(hipGraphAddKernelNode(&node, g[c], &prev, 1, &kp));
                prev = node;
            }
            CHK(hipGraphInstantiate(&ex[c], g[c], nullptr, nullptr, 0));
        }
        for (int c = 0; c < ng; ++c) CHK(hipGraphLaunch(ex[c], s[c]));
        for (int c = 0; c < ng; ++c) CHK(hipStreamSynchronize(s[c]));
        hipEvent_t e0, e1, ej;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        CHK(hipEventCreate(&ej));
        CHK(hipEventRecord(e0, s[0]));
        for (int r = 0; r < reps; ++r)
            for (int c = 0; c < ng; ++c) CHK(hipGraphLaunch(ex[c], s[c]));
        if (ng == 2) {
            CHK(hipEventRecord(ej, s[1]));
            CHK(hipStreamWaitEvent(s[0], ej, 0));
        }
        CHK(hipEventRecord(e1, s[0]));
        for (int c = 0; c < ng; ++c) CHK(hipStreamSynchronize(s[c]));
        float ms_t = 0;
        CHK(hipEventElapsedTime(&ms_t, e0, e1));
        unsigned herr = 0;
        CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        {
            unsigned hf[8], he[64];
            CHK(hipMemcpy(hf, flags, 32, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(he, epochs, 256, hipMemcpyDeviceToHost));
            printf("  flags[0..5] = %u %u %u %u %u %u, epochs = %u %u\n", hf[0], hf[1], hf[2], hf[3], hf[4], hf[5], he[0], he[32]);
        }
        std::vector<unsigned long long> hs(3 * K);
        CHK(hipMemcpy(hs.data(), stamps, K * 24, hipMemcpyDeviceToHost));
        double wait_ns = 0, span_ns = 0, period_ns = 0;
        for (int i = 10; i < K - 1; ++i) {
            wait_ns += (double)(hs[3 * i + 1] - hs[3 * i]) * 10.0;
            span_ns += (double)(hs[3 * i + 2] - hs[3 * i]) * 10.0;
            period_ns += (double)((long long)hs[3 * (i + 1) + 2] - (long long)hs[3 * i + 2]) * 10.0;
        }
        const int cnt = K - 11;
        printf("NL=%d (%.1f MB per kernel) mode %d (G=%d): %.3f us per kernel (events), spin-timeouts %u | workgroup 0: start->flag %.0f ns, start->end %.0f ns, end->end %.0f ns\n",
               NL, wbytes / 1e6, mode, G, ms_t * 1e3 / reps / K, herr, wait_ns / cnt, span_ns / cnt, period_ns / cnt);
        fflush(stdout);
        for (int c = 0; c < ng; ++c) {
            CHK(hipGraphExecDestroy(ex[c]));
            CHK(hipGraphDestroy(g[c]));
        }
    }
    CHK(hipFree(w));
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 216;
    run<1>(G);
    run<5>(G);
    return 0;
}
