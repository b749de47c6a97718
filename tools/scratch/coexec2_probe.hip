// coexec2_probe.hip -- WHICH vector instructions run beside the fp16 matrix instruction on one SIMD, and which take its time?
// coexec_probe.hip found that the split kernels' conversion MIX (v_max_f32, v_cvt_pk_f16_f32, v_fma_mixlo/hi_f16) and v_mfma_f32_32x32x16_f16 take the SUM of their times.
// Here one instruction kind at a time: an 8-wave workgroup per CU (2 waves per SIMD), waves 0-3 run matrix chains (128 instructions = 4 096 pipe cycles per iteration),
// waves 4-7 a stream of ONE vector instruction kind sized to ~4 096 issue cycles when alone.  Reported: matrix alone, vector alone, both -- "both" near the larger of the two
// means the kind co-executes, near their sum that it shares the matrix instruction's issue / data path.
//   hipcc -O3 --offload-arch=gfx950 coexec2_probe.hip -o coexec2_probe && ./coexec2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(512) k(int mode, int iters, float *out)
{
    const int wave = threadIdx.x >> 6;
    const bool roleA = (mode == 0 || mode == 2) && wave < 4;
    const bool roleB = (mode == 1 || mode == 2) && wave >= 4;
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
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) r += acc[i][j];
    } else if (roleB) {
        float v[32];
        unsigned int w[32];
        for (int i = 0; i < 32; i++) { v[i] = 1.0f + 0.01f * (threadIdx.x + i); w[i] = 0x3c003c00u + i; }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 32; u++) {
#pragma unroll
                for (int i = 0; i < 32; i++) {
                    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 31]));
                    if constexpr (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 31]));
                    if constexpr (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(v[i]), "v"(v[(i + 1) & 31]));
                    if constexpr (KIND == 3) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(w[i]) : "v"(w[(i + 1) & 31]), "v"(v[i]));
                    if constexpr (KIND == 4) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[i]) : "v"(w[i]));
                    if constexpr (KIND == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 31]));
                    if constexpr (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<f32x2 *>(&v[i & 30])) : "v"(*reinterpret_cast<f32x2 *>(&v[(i + 2) & 30])));
                    if constexpr (KIND == 7) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(w[i]), "v"(v[(i + 1) & 31]));
                    if constexpr (KIND == 8) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v[i]) : "v"(w[i]));
                    if constexpr (KIND == 9) asm volatile("v_max_i32 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 31]));
                }
            }
        }
        for (int i = 0; i < 32; i++) r += v[i] + (float)w[i];
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int KIND>
static void run(const char *name, float *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float t[3];
    for (int mode = 0; mode < 3; mode++) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, mode, 10, out);
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, mode, 1000, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        t[mode] = best;
    }
    printf("%-22s matrix alone %6.3f ms   vector alone %6.3f ms (%4.1f cycles per instruction at 2.4 GHz)   both %6.3f ms = %.2f of the sum, %.2f of the larger\n", name, t[0], t[1],
           t[1] * 1e-3 * 2.4e9 / (1000.0 * 1024.0), t[2], t[2] / (t[0] + t[1]), t[2] / (t[0] > t[1] ? t[0] : t[1]));
}

int main()
{
    float *out; (void)hipMalloc(&out, 4096);
    for (int rep = 0; rep < 2; rep++) {
        run<0>("v_fma_f32", out);
        run<1>("v_max_f32", out);
        run<9>("v_max_i32", out);
        run<5>("v_sub_f32", out);
        run<2>("v_cvt_pk_f16_f32", out);
        run<3>("v_fma_mixlo_f16", out);
        run<7>("v_fma_mix_f32", out);
        run<4>("v_cvt_f32_f16", out);
        run<8>("v_cvt_f32_f16 sdwa", out);
        run<6>("v_pk_add_f32", out);
    }
    return 0;
}
