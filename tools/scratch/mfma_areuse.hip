// How does MFMA throughput depend on how many distinct A (or B) register quads a wave cycles through?  Pure register loop, no memory.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NA, int NB, int SHAPE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed)
{
    half8 a[NA], b[NB];
    for (int i = 0; i < NA; i++) for (int j = 0; j < 8; j++) a[i][j] = (_Float16)(seed * (threadIdx.x % 13 + i * 7 + j) * 0.01f);
    for (int i = 0; i < NB; i++) for (int j = 0; j < 8; j++) b[i][j] = (_Float16)(seed * (threadIdx.x % 11 + i * 5 + j) * 0.02f);
    float r = 0.0f;
    if constexpr (SHAPE == 16) {
        f32x4 c[8] = {};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int s = 0; s < 48; s++)       // 48 MFMAs: A index changes every 4 MFMAs, B every MFMA, 8 accumulators round-robin
                c[s & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(s >> 2) % NA], b[s % NB], c[s & 7], 0, 0, 0);
        }
        for (int i = 0; i < 8; i++) r += c[i][0] + c[i][3];
    } else {
        f32x16 c[4] = {};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int s = 0; s < 48; s++)
                c[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(s >> 2) % NA], b[s % NB], c[s & 3], 0, 0, 0);
        }
        for (int i = 0; i < 4; i++) r += c[i][0] + c[i][15];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NA, int NB, int SHAPE> void run(float *d, const char *name)
{
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NA, NB, SHAPE>), dim3(256), dim3(256), 0, 0, d, iters, 1.0f);      // one 4-wave block per CU: one wave per SIMD
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double mf = 48.0 * iters;   // MFMAs per wave
    printf("%-28s NA=%d NB=%d  %.3f ms  %.1f ns per MFMA per SIMD\n", name, NA, NB, ms, ms * 1e6 / mf);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 256 * 4);
    run<1, 4, 16>(d, "16x16x32"); run<2, 4, 16>(d, "16x16x32"); run<3, 4, 16>(d, "16x16x32"); run<4, 4, 16>(d, "16x16x32"); run<8, 4, 16>(d, "16x16x32"); run<12, 4, 16>(d, "16x16x32");
    run<2, 1, 16>(d, "16x16x32"); run<2, 2, 16>(d, "16x16x32"); run<2, 8, 16>(d, "16x16x32"); run<8, 8, 16>(d, "16x16x32");
    run<1, 4, 32>(d, "32x32x16"); run<2, 4, 32>(d, "32x32x16"); run<4, 4, 32>(d, "32x32x16"); run<8, 4, 32>(d, "32x32x16"); run<8, 8, 32>(d, "32x32x16");
    return 0;
}
