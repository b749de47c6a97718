// valu_cost_probe.hip -- issue cost of the vector instructions the split-precision kernels' conversions are made of: one wave per SIMD (256-thread workgroups, one
// per CU), a long stream of INDEPENDENT instances of one instruction, cycles per instruction from s_memtime.
//   hipcc -O3 --offload-arch=gfx950 valu_cost_probe.hip -o valu_cost_probe && ./valu_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void __launch_bounds__(256) k(int iters, unsigned long long *cyc, float *out)
{
    float v[16]; unsigned int u[16];
    for (int i = 0; i < 16; i++) { v[i] = 1.0f + 0.01f * (threadIdx.x + i); u[i] = 0x3c003c00u + i; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
                if (KIND == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[(i + 1) & 15]));
                if (KIND == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[(i + 1) & 15]));
                if (KIND == 3) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(v[i]));
                if (KIND == 4) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(v[i]));
                if (KIND == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
                if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
                if (KIND == 7) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[i]) : "v"(u[i]));
                if (KIND == 8) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (KIND == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&v[(i & 7) * 2])) : "v"(*reinterpret_cast<double *>(&v[((i + 1) & 7) * 2])));
                if (KIND == 10) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[i]) : "v"(v[i]));
                if (KIND == 11) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(v[i]) : "v"(u[i]), "v"(v[(i + 1) & 15]));
                if (KIND == 12) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (KIND == 13) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0; for (int i = 0; i < 16; i++) r += v[i] + (float)u[i];
    if (r == 123.456f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int K> void run(const char *name, unsigned long long *cyc, float *out)
{
    const int iters = 4000;
    hipLaunchKernelGGL(k<K>, dim3(256), dim3(256), 0, 0, 10, cyc, out);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<K>, dim3(256), dim3(256), 0, 0, iters, cyc, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s %6.2f counter ticks per instruction, %7.2f ns per 64 instructions (one wave per SIMD)\n", name, (double)c / (iters * 64.0), ms * 1e6 / iters);
}
int main()
{
    unsigned long long *cyc; float *out; (void)hipMalloc(&cyc, 64); (void)hipMalloc(&out, 64);
    run<0>("v_max_f32", cyc, out); run<1>("v_cvt_pk_f16_f32", cyc, out); run<2>("v_cvt_pkrtz_f16_f32", cyc, out); run<3>("v_fma_mixlo_f16", cyc, out);
    run<4>("v_fma_mixhi_f16", cyc, out); run<5>("v_sub_f32", cyc, out); run<6>("v_fma_f32", cyc, out); run<7>("v_cvt_f32_f16", cyc, out);
    run<8>("v_pk_max_f16", cyc, out); run<9>("v_pk_add_f32", cyc, out); run<10>("v_cvt_f16_f32", cyc, out); run<11>("v_fma_mix_f32", cyc, out);
    run<12>("v_pk_fma_f16", cyc, out); run<13>("v_med3_f32", cyc, out);
    return 0;
}
