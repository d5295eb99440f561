// launch_floor.hip -- developer probe: what does one dependent kernel cost on this box?
// Back-to-back dependent launches of trivial kernels, eager vs hipGraph, a few grid sizes,
// plus a "code size" ladder (straight-line code executed once) to price the cold
// instruction cache per launch.
//   hipcc --offload-arch=gfx950 -O2 launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

__global__ void k_inc(int *p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p = *p + 1;
}

// N straight-line FMAs on a value derived from memory: ~8 B of code each, executed once.
template <int N>
__global__ void k_code(float *p) {
    float v = p[threadIdx.x & 63];
#pragma unroll
    for (int i = 0; i < N; ++i) v = fmaf(v, 1.0001f + i * 1e-7f, 0.5f + i);
    if (v == 123.456f) p[0] = v;
}

template <class F>
static float time_graph(hipStream_t s, int n, F launch) {
    hipGraph_t g;
    hipGraphExec_t ex;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    hipGraphLaunch(ex, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ex, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ex);
    hipGraphDestroy(g);
    return ms * 1e3f / (5 * n);
}

template <class F>
static float time_eager(hipStream_t s, int n, F launch) {
    launch();
    hipStreamSynchronize(s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / n;
}

int main() {
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *p;
    float *f;
    CHECK(hipMalloc(&p, 4));
    CHECK(hipMalloc(&f, 1024));
    CHECK(hipMemset(p, 0, 4));
    CHECK(hipMemset(f, 0, 1024));
    const int n = 600;
    for (int grid : {1, 256, 1024}) {
        auto l = [&] { hipLaunchKernelGGL(k_inc, dim3(grid), dim3(256), 0, s, p); };
        printf("k_inc grid %4d: eager %.2f us/launch, graph %.2f us/launch\n", grid, time_eager(s, n, l), time_graph(s, n, l));
    }
    auto l0 = [&] { hipLaunchKernelGGL(k_code<16>, dim3(256), dim3(256), 0, s, f); };
    auto l1 = [&] { hipLaunchKernelGGL(k_code<128>, dim3(256), dim3(256), 0, s, f); };
    auto l2 = [&] { hipLaunchKernelGGL(k_code<512>, dim3(256), dim3(256), 0, s, f); };
    auto l3 = [&] { hipLaunchKernelGGL(k_code<2048>, dim3(256), dim3(256), 0, s, f); };
    printf("straight-line code, 256 WGs x 256 thr (graph us/launch): 16 fma %.2f | 128 fma %.2f | 512 fma %.2f | 2048 fma %.2f\n",
           time_graph(s, n, l0), time_graph(s, n, l1), time_graph(s, n, l2), time_graph(s, n, l3));
    auto m0 = [&] { hipLaunchKernelGGL(k_code<16>, dim3(1), dim3(64), 0, s, f); };
    auto m3 = [&] { hipLaunchKernelGGL(k_code<2048>, dim3(1), dim3(64), 0, s, f); };
    printf("straight-line code, 1 WG x 64 thr (graph us/launch): 16 fma %.2f | 2048 fma %.2f\n", time_graph(s, n, m0), time_graph(s, n, m3));
    int v = 0;
    CHECK(hipMemcpy(&v, p, 4, hipMemcpyDeviceToHost));
    printf("counter = %d\n", v);
    return 0;
}
