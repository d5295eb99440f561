// overlap_probe.hip -- can consecutive dependent kernels of a hipGraph overlap their
// launch + prologue when the data dependency is a device-side flag instead of a graph edge?
//   mode 0: chain, edge i-1 -> i (what the decode graph does today)
//   mode 1: edges i-2 -> i only; kernel i spins (bounded) on a counter kernel i-1 bumps
// Each kernel: G workgroups x 512 threads, prologue = one 16-byte global load per thread
// (its "weights"), then [wait], then reads the previous kernel's 10 KB output, writes its own.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap_probe tools/probes/overlap_probe.hip
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

__global__ __launch_bounds__(512) void k_step(const uint4 *__restrict__ w, const float *x_in, float *x_out, unsigned *flags, int idx,
                                              unsigned expect, int wait, int n, unsigned *err) {
    const int tid = threadIdx.x;
    uint4 v = w[(size_t)blockIdx.x * 512 + tid];  // prologue: independent of the previous kernel
    if (wait) {
        if (tid == 0) {
            unsigned it = 0;
            while (__hip_atomic_load(&flags[idx - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expect) {
                if (++it > 20000u) {  // bounded: never hang the GPU
                    atomicAdd(err, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        __threadfence();
    }
    float acc = (float)(v.x & 1u);
    for (int i = tid; i < n; i += 512) acc += x_in[i];
    // block reduce (cheap stand-in for the MFMA body)
    __shared__ float red[512];
    red[tid] = acc;
    __syncthreads();
    for (int s = 256; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid < 16) x_out[blockIdx.x * 16 + tid] = red[0] * 1e-9f + (float)tid;
    if (flags) {
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&flags[idx], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char **argv) {
    const int K = 150, G = argc > 1 ? atoi(argv[1]) : 160, n = 2560, reps = 20;
    uint4 *w;
    float *xa, *xb;
    unsigned *flags, *err;
    CHK(hipMalloc(&w, (size_t)G * 512 * 16 * K));
    CHK(hipMemset(w, 0, (size_t)G * 512 * 16 * K));
    CHK(hipMalloc(&xa, 65536));
    CHK(hipMalloc(&xb, 65536));
    CHK(hipMemset(xa, 0, 65536));
    CHK(hipMemset(xb, 0, 65536));
    CHK(hipMalloc(&flags, K * 4));
    CHK(hipMalloc(&err, 4));
    CHK(hipMemset(err, 0, 4));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g;
        CHK(hipGraphCreate(&g, 0));
        std::vector<hipGraphNode_t> nodes(K);
        // epoch-free flags: memset node first
        hipGraphNode_t ms;
        hipMemsetParams mp = {};
        mp.dst = flags;
        mp.value = 0;
        mp.elementSize = 4;
        mp.width = K;
        mp.height = 1;
        CHK(hipGraphAddMemsetNode(&ms, g, nullptr, 0, &mp));
        for (int i = 0; i < K; ++i) {
            const uint4 *wi = w + (size_t)i * G * 512;
            const float *xin = (i & 1) ? xb : xa;
            float *xout = (i & 1) ? xa : xb;
            unsigned *fl = mode == 1 ? flags : nullptr;
            int idx = i, wait = (mode == 1 && i > 0) ? 1 : 0, nn = n;
            unsigned expect = (unsigned)G;
            void *args[] = {&wi, &xin, &xout, &fl, &idx, &expect, &wait, &nn, &err};
            hipKernelNodeParams kp = {};
            kp.func = (void *)k_step;
            kp.gridDim = dim3(G);
            kp.blockDim = dim3(512);
            kp.kernelParams = args;
            std::vector<hipGraphNode_t> deps;
            if (mode == 0) {
                deps.push_back(i == 0 ? ms : nodes[i - 1]);
            } else {
                if (i < 2) deps.push_back(ms);
                if (i >= 2) deps.push_back(nodes[i - 2]);
            }
            CHK(hipGraphAddKernelNode(&nodes[i], g, deps.data(), deps.size(), &kp));
        }
        hipGraphExec_t ex;
        CHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        CHK(hipGraphLaunch(ex, s));
        CHK(hipStreamSynchronize(s));
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) CHK(hipGraphLaunch(ex, s));
        CHK(hipEventRecord(e1, s));
        CHK(hipStreamSynchronize(s));
        float ms_t = 0;
        CHK(hipEventElapsedTime(&ms_t, e0, e1));
        unsigned herr = 0;
        CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("mode %d (G=%d): %.3f us per kernel, spin-timeouts %u\n", mode, G, ms_t * 1e3 / reps / K, herr);
        fflush(stdout);
        CHK(hipGraphExecDestroy(ex));
        CHK(hipGraphDestroy(g));
    }
    return 0;
}
