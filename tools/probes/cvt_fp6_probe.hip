// cvt_fp6_probe.hip -- operand order and scale meaning of v_cvt_scalef32_2xpk16_fp6_f32 (32 f32 -> 32 packed fp6 e2m3), for the
// prefill quantiser's hardware pack.  Lane 0 converts a = (1 .. 16) / 8, b = -(1 .. 16) / 8 with scale 1, then the integers with
// scale 8 and scale 0.125; prints the 32 six-bit fields of the result.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/cvt_fp6_probe tools/probes/cvt_fp6_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
__global__ void k(const float *in, unsigned *out, float scale) {
    v16f a, b;
    for (int i = 0; i < 16; ++i) a[i] = in[i], b[i] = in[16 + i];
    v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    for (int i = 0; i < 6; ++i) out[i] = r[i];
}
static void run(const float *h, float scale, const char *what) {
    float *d;
    unsigned *o, ho[6];
    hipMalloc(&d, 128), hipMalloc(&o, 24);
    hipMemcpy(d, h, 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o, scale);
    hipMemcpy(ho, o, 24, hipMemcpyDeviceToHost);
    printf("%s, scale %g:", what, scale);
    for (int j = 0; j < 32; ++j) {
        unsigned v = 0;
        for (int b = 0; b < 6; ++b) v |= ((ho[(6 * j + b) >> 5] >> ((6 * j + b) & 31)) & 1u) << b;
        printf(" %u", v);
    }
    printf("\n");
    hipFree(d), hipFree(o);
}
int main() {
    float e[32], n[32];
    for (int i = 0; i < 16; ++i) e[i] = (i + 1) / 8.0f, e[16 + i] = -(i + 1) / 8.0f, n[i] = (float)(i + 1), n[16 + i] = -(float)(i + 1);
    run(e, 1.0f, "eighths");
    run(n, 8.0f, "integers");
    run(n, 0.125f, "integers");
    run(n, 1.0f, "integers");
    return 0;
}
