// record_fanout_probe.hip -- what would folding k_attn_combine<5> into the o-projection cost at c4's context?
// At 4.3 k keys the wide attention form leaves 34 records of 128 positions per KV head; a record holds, for the 4 query heads of the
// group, 128 un-normalised output values + (m, l): 4 x 130 x 4 B = 2,080 B -> 34 x 5 x 2,080 B = 353,600 B for the whole activation row.
// The o-projection is 160 workgroups of 16 output rows, each over ALL 2560 columns (K split over its 8 waves): a workgroup that merged
// the records itself would have to read every one of them.  This probe times exactly that fan-out -- `wgs` workgroups x 512 threads each
// pulling `bytes` from ONE shared buffer (16-byte loads, all requested up front, summed so nothing is optimised away) -- next to the
// combine kernel's own shape (20 workgroups, each the 34 records of ONE query head: 17,680 B), back-to-back launches, HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/record_fanout_probe tools/probes/record_fanout_probe.hip
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

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_fanout(const v4f *rec, int n16, int stride_wg, float *out) {
    const v4f *p = rec + (size_t)blockIdx.x * stride_wg;
    v4f acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < n16; i += 512) acc += __builtin_nontemporal_load(p + i);
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, s);
}

static float run(int wgs, size_t bytes, int stride16, const v4f *rec, float *out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int reps = 200;
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_fanout, dim3(wgs), dim3(512), 0, 0, rec, (int)(bytes / 16), stride16, out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_fanout, dim3(wgs), dim3(512), 0, 0, rec, (int)(bytes / 16), stride16, out);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main() {
    const size_t all = 34 * 5 * 2080, one_head = 34 * 520;
    v4f *rec;
    float *out;
    CHK(hipMalloc(&rec, all + 4096));
    CHK(hipMemset(rec, 0, all + 4096));
    CHK(hipMalloc(&out, 4096));
    CHK(hipMemset(out, 0, 4096));
    const float empty = run(160, 16, 0, rec, out);
    const float fan = run(160, all - all % 16, 0, rec, out);
    const float comb = run(20, one_head - one_head % 16, (int)(one_head / 16), rec, out);
    printf("launch of 160 workgroups reading 16 B each (the floor of a dependent launch, back to back): %.2f us\n", empty);
    printf("160 workgroups x %zu B of records each (merge folded into the o-projection):            %.2f us  (+%.2f us over the floor)\n", all, fan, fan - empty);
    printf("20 workgroups x %zu B each (k_attn_combine's own read shape):                            %.2f us  (+%.2f us)\n", one_head, comb, comb - empty);
    return 0;
}
