// Are XCD-local (workgroup-scope) atomics faster than device-scope ones when every address is only ever touched from one XCD?
// Each block reads its XCC id from the hardware register and only updates slots with (slot & 7) == xcc.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ inline uint32_t rnd(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
template <int MODE> __global__ void k(int64_t n, unsigned long long *tab, uint32_t mask, int rep, uint32_t *xcc_hist)
{
    const uint32_t x = xcc_id();
    if (threadIdx.x == 0) atomicAdd(xcc_hist + (x & 15), 1u);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int q = 0; q < rep; q++) {
        uint32_t s = rnd((uint32_t)i * 9781u + q * 7919u) & mask;
        if (MODE == 0) atomicAdd(tab + s, 0x0000000200000001ull);                                                       // device scope, any slot
        if (MODE == 1) { s = (s & ~7u) | x; atomicAdd(tab + s, 0x0000000200000001ull); }                                 // device scope, XCD-partitioned slots
        if (MODE == 2) { s = (s & ~7u) | x; __hip_atomic_fetch_add(tab + s, 0x0000000200000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (MODE == 3) { s = (s & ~7u) | x; __hip_atomic_fetch_add(tab + s, 0x0000000200000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
    }
}
int main()
{
    const int64_t n = 1 << 22; const int rep = 32;
    for (uint32_t logt : {19u, 23u}) {
        unsigned long long *tab; hipMalloc(&tab, (size_t)8 << 23);
        uint32_t *hist; hipMalloc(&hist, 64);
        const uint32_t mask = (1u << logt) - 1;
        const char *names[] = {"agent scope, any slot", "agent scope, xcd slots", "workgroup scope, xcd slots", "wavefront scope, xcd slots"};
        for (int mode = 0; mode < 4; mode++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            float ms = 0;
            for (int it = 0; it < 2; it++) {
                hipMemset(tab, 0, (size_t)8 << 23); hipMemset(hist, 0, 64);
                hipEventRecord(a);
                dim3 g((unsigned)(n / 256)), bl(256);
                if (mode == 0) hipLaunchKernelGGL(k<0>, g, bl, 0, 0, n, tab, mask, rep, hist);
                if (mode == 1) hipLaunchKernelGGL(k<1>, g, bl, 0, 0, n, tab, mask, rep, hist);
                if (mode == 2) hipLaunchKernelGGL(k<2>, g, bl, 0, 0, n, tab, mask, rep, hist);
                if (mode == 3) hipLaunchKernelGGL(k<3>, g, bl, 0, 0, n, tab, mask, rep, hist);
                hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            std::vector<unsigned long long> h((size_t)1 << 23);
            hipMemcpy(h.data(), tab, (size_t)8 << 23, hipMemcpyDeviceToHost);
            unsigned long long lo = 0, hi = 0;
            for (auto v : h) { lo += v & 0xffffffffull; hi += v >> 32; }
            uint32_t hh[16]; hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost);
            printf("slots 2^%u  %-28s %8.3f ms  %7.2f G updates/s   sum lo %llu (expect %lld) hi %llu   xcc hist:", logt, names[mode], ms, n * rep / (ms * 1e6), lo, (long long)n * rep, hi);
            for (int q = 0; q < 9; q++) printf(" %u", hh[q]);
            printf("\n");
        }
        hipFree(tab); hipFree(hist);
    }
    return 0;
}
