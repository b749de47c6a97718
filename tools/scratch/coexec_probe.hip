// coexec_probe.hip -- do the fp32 matrix instruction (v_mfma_f32_32x32x2_f32) and the packed fp32 FMA (v_pk_fma_f32) of one SIMD run side by side?
// One 8-wave workgroup per CU (2 waves per SIMD); waves 0-3 run kernel role A (matrix chains), waves 4-7 role B (packed FMA chains), registers only.
// Modes: 0 = all eight waves A, 1 = all eight B, 2 = A on waves 0-3 and B on 4-7, 3 = A on waves 0-3 only (4-7 exit), 4 = B on waves 4-7 only.
// Second table (kind 1): A = the fp16 matrix instruction (v_mfma_f32_32x32x16_f16), B = the split kernels' conversion mix (v_max_f32, v_cvt_pk_f16_f32,
// v_fma_mixlo/hi_f16: 2.5 vector instructions per value), the same number of SIMD cycles of each per iteration at nominal rates.
//   hipcc -O3 --offload-arch=gfx950 coexec_probe.hip -o coexec_probe && ./coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(512) k16(int mode, int iters, float *out)
{
    const int wave = threadIdx.x >> 6;
    const bool roleA = mode == 0 || ((mode == 2 || mode == 3) && wave < 4);
    const bool roleB = mode == 1 || ((mode == 2 || mode == 4) && wave >= 4);
    float r = 0.0f;
    if (roleA) {
        f32x16 acc[4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) acc[i][j] = (float)(threadIdx.x + i + j);
        half8 a, b;
        for (int j = 0; j < 8; j++) { a[j] = (_Float16)(1.0f + 0.001f * (threadIdx.x + j)); b[j] = (_Float16)(1.0f - 0.001f * (threadIdx.x + j)); }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 32; u++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);          // 128 matrix instructions x 32 pipe cycles = 4 096 cycles per iteration
        }
        for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) r += acc[i][j];
    } else if (roleB) {
        float v[32];
        for (int i = 0; i < 32; i++) v[i] = 1.0f + 0.01f * (threadIdx.x + i);
        unsigned int sink = 0;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 13; u++) {          // 13 x 16 pairs x 5 instructions = 1 040 vector instructions ~ 4 160 issue cycles per iteration
#pragma unroll
                for (int i = 0; i < 32; i += 2) {
                    const float m0 = fmaxf(v[i], 0.5f), m1 = fmaxf(v[i + 1], 0.5f);
                    unsigned int hi, lo;
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(m0), "v"(m1));
                    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(m0));
                    asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(m1));
                    sink ^= hi ^ lo;
                    asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]));
                }
            }
        }
        r = (float)sink;
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}

__global__ void __launch_bounds__(512) k(int mode, int iters, float *out)
{
    const int wave = threadIdx.x >> 6;
    const bool roleA = mode == 0 || ((mode == 2 || mode == 3) && wave < 4);
    const bool roleB = mode == 1 || ((mode == 2 || mode == 4) && wave >= 4);
    float r = 0.0f;
    if (roleA) {
        f32x16 acc[4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) acc[i][j] = (float)(threadIdx.x + i + j);
        float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 16; u++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);          // 64 matrix instructions per iteration
        }
        for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) r += acc[i][j];
    } else if (roleB) {
        f32x2 acc[32];
        for (int i = 0; i < 32; i++) acc[i] = f32x2{(float)(threadIdx.x + i), (float)i};
        f32x2 w = {1.0f + threadIdx.x * 1e-6f, 1.0f - threadIdx.x * 1e-6f}, x = {0.999f, 1.001f};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 32; u++)
#pragma unroll
                for (int i = 0; i < 32; i++) acc[i] = __builtin_elementwise_fma(w, x, acc[i]);                                // 1 024 packed FMAs per iteration = the flops of 64 matrix instructions
            asm volatile("" : "+v"(w), "+v"(x));
        }
        for (int i = 0; i < 32; i++) r += acc[i][0] + acc[i][1];
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}

int main()
{
    float *out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char *names[5] = {"all 8 waves matrix", "all 8 waves packed FMA", "4 matrix + 4 packed FMA", "4 matrix only", "4 packed FMA only"};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 5; mode++) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, 10, out);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double waves = (mode < 2 ? 8 : (mode == 2 ? 8 : 4)) * 256.0;
            const double flop = waves * iters * 64.0 * 32 * 32 * 2 * 2;          // either role: 64 x 4 096 flop per iteration and wave
            printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", names[mode], ms, flop / ms * 1e-9);
        }
    printf("-- fp16 matrix instruction vs the split kernels' conversion mix --\n");
    const char *names16[5] = {"all 8 waves fp16 matrix", "all 8 waves conversions", "4 matrix + 4 conversions", "4 matrix only", "4 conversions only"};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 5; mode++) {
            hipLaunchKernelGGL(k16, dim3(256), dim3(512), 0, 0, mode, 10, out);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k16, dim3(256), dim3(512), 0, 0, mode, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-28s %8.3f ms\n", names16[mode], ms);
        }
    return 0;
}
