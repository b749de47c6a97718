// scattered-atomic throughput on MI355X: fp32 add vs u64 add vs packed f16 add vs f64 add (table = 2^23 slots x 8 B, random slots)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t rnd(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE> __global__ void k(int64_t n, void *tab, uint32_t mask, int rep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int q = 0; q < rep; q++) {
        const uint32_t s = rnd((uint32_t)i * 9781u + q * 7919u) & mask;
        if (MODE == 0) { float *t = (float *)tab; unsafeAtomicAdd(t + 2 * s, 1.0f); unsafeAtomicAdd(t + 2 * s + 1, 2.0f); }     // two fp32 atomics per slot
        if (MODE == 1) { unsigned long long *t = (unsigned long long *)tab; atomicAdd(t + s, 0x0000000200000001ull); }
        if (MODE == 2) { __half2 *t = (__half2 *)tab; unsafeAtomicAdd(t + 2 * s, __half2{(__half)1.0f, (__half)2.0f}); }
        if (MODE == 3) { double *t = (double *)tab; unsafeAtomicAdd(t + s, 1.0); }
        if (MODE == 4) { float *t = (float *)tab; unsafeAtomicAdd(t + 2 * s, 1.0f); }                                              // one fp32 atomic per slot
        if (MODE == 5) { uint32_t *t = (uint32_t *)tab; atomicAdd(t + 2 * s, 1u); }
    }
}
int main()
{
    const int64_t n = 1 << 22; const int rep = 32;
    for (uint32_t logt : {14u, 19u, 23u}) {
        void *tab; hipMalloc(&tab, (size_t)8 << 23); hipMemset(tab, 0, (size_t)8 << 23);
        const uint32_t mask = (1u << logt) - 1;
        const char *names[] = {"2 x f32 add", "1 x u64 add", "1 x pk f16 add", "1 x f64 add", "1 x f32 add", "1 x u32 add"};
        for (int mode = 0; mode < 6; mode++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int it = 0; it < 2; it++) {
                hipEventRecord(a);
                dim3 g((unsigned)(n / 256)), bl(256);
                if (mode == 0) hipLaunchKernelGGL(k<0>, g, bl, 0, 0, n, tab, mask, rep);
                if (mode == 1) hipLaunchKernelGGL(k<1>, g, bl, 0, 0, n, tab, mask, rep);
                if (mode == 2) hipLaunchKernelGGL(k<2>, g, bl, 0, 0, n, tab, mask, rep);
                if (mode == 3) hipLaunchKernelGGL(k<3>, g, bl, 0, 0, n, tab, mask, rep);
                if (mode == 4) hipLaunchKernelGGL(k<4>, g, bl, 0, 0, n, tab, mask, rep);
                if (mode == 5) hipLaunchKernelGGL(k<5>, g, bl, 0, 0, n, tab, mask, rep);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("slots 2^%u  %-16s %8.3f ms  %7.2f G slot-updates/s\n", logt, names[mode], ms, n * rep / (ms * 1e6));
        }
        hipFree(tab);
    }
    return 0;
}
