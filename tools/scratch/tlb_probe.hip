// tlb_probe.hip -- does the number of distinct PAGES a wave's gather touches cost anything on top of the number of distinct LINES?
// Every lane loads 16 bytes from a random 128-byte line of a big buffer (every load misses the caches).  mode 0: the 64 lanes of a wave-instruction stay inside ONE
// 2 MB region (region random per wave and iteration); mode 1: inside one 64 KB region; mode 2: every lane in its own random 2 MB region; mode 3: lanes of a wave in
// 8 regions.  Same lines-per-instruction (64), same hit rate (none): the difference is address translation.
// (the buffer size must be a power of two: regions are chosen by mask)
//   hipcc --offload-arch=gfx950 -O3 -o tlb_probe tlb_probe.hip && ./tlb_probe [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_probe(const u32x4 *__restrict__ buf, uint64_t bytes, int mode, int iters, uint32_t *__restrict__ out)
{
    const uint32_t gtid = blockIdx.x * 256 + threadIdx.x, wave = gtid >> 6, lane = gtid & 63;
    const uint64_t regions = bytes >> 21;          // 2 MB regions
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll 4
    for (int it = 0; it < iters; it++) {
        const uint32_t hw = mix(wave * 0x9e3779b9u + it * 0x85ebca6bu), hl = mix(hw ^ (lane * 0xc2b2ae35u + 0x27d4eb2fu));
        uint64_t region, off;
        if (mode == 0) { region = hw & (uint32_t)(regions - 1); off = (uint64_t)(hl & 0x3fff) << 7; }                                  // one 2 MB region per wave-instruction
        else if (mode == 1) { region = hw & (uint32_t)(regions - 1); off = ((uint64_t)(mix(hw) & 31) << 16) + ((uint64_t)(hl & 0x1ff) << 7); }   // one 64 KB region
        else if (mode == 2) { region = hl & (uint32_t)(regions - 1); off = (uint64_t)(mix(hl) & 0x3fff) << 7; }                        // a region per lane
        else if (mode == 4) { region = hl & (uint32_t)(regions - 1); off = (uint64_t)(mix(hl) & 7) << 7; }             // L2-resident set (8 lines per region), a region per lane
        else if (mode == 5) { region = 0; off = (uint64_t)(hl & 0x3fff) << 7; }                                           // L2-resident set of the same size inside ONE region
        else if (mode == 6) { region = hl & 63; off = (uint64_t)(mix(hl) & 0xff) << 7; }                                  // the same in 64 regions
        else { region = mix(hw + (lane >> 3)) & (uint32_t)(regions - 1); off = (uint64_t)(hl & 0x3fff) << 7; }                         // 8 regions per wave-instruction
        const u32x4 v = buf[((region << 21) + off) >> 4];
        acc ^= v;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[gtid] = 1;
}
int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) & ~((1ull << 21) - 1);
    void *buf; uint32_t *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 1 << 26) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    const int blocks = 256 * 8 * 4, iters = 64;          // 8 waves per SIMD resident, 4 rounds
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[7] = {"one 2 MB region / wave-instr", "one 64 KB region / wave-instr", "a 2 MB region per LANE", "8 regions / wave-instr", "cached lines, region per lane", "cached lines, ONE region", "cached lines, 64 regions"};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 7; mode++) {
            hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, (const u32x4 *)buf, bytes, mode, iters, out);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, (const u32x4 *)buf, bytes, mode, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double loads = (double)blocks * 256 * iters;
            printf("%.1f GiB  %-32s %.3f ms  %.1f G lane-loads/s  (%.2f per clock and CU at 2.4 GHz)\n", gib, names[mode], ms, loads / ms / 1e6, loads / (ms * 1e-3) / 256 / 2.4e9);
        }
    return 0;
}
