// cycles per wave64 instruction for the conversion instructions of the split-precision kernels (independent chains, one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP> __global__ void __launch_bounds__(256) k(float *out, int iters, float seed)
{
    float x[16]; uint32_t y[16];
    for (int i = 0; i < 16; i++) { x[i] = seed * (threadIdx.x + i); y[i] = threadIdx.x * 7 + i; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) asm volatile("v_max_f32 %0, 0, %0" : "+v"(x[i]));
            if (OP == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(y[i]) : "v"(x[i]));
            if (OP == 2) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(y[i]) : "v"(y[(i + 1) & 15]), "v"(x[i]));
            if (OP == 3) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(y[i]) : "v"(y[(i + 3) & 15]));
            if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(x[(i + 5) & 15]));
            if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double *)&x[2 * (i & 7)]) : "v"(*(double *)&x[2 * ((i + 3) & 7)]));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0; for (int i = 0; i < 16; i++) r += x[i] + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[256 * 256] = (float)(t1 - t0);
}
int main()
{
    float *d; hipMalloc(&d, (256 * 256 + 4) * 4);
    const char *names[] = {"v_max_f32", "v_cvt_pk_f16_f32", "v_fma_mixlo_f16", "v_pk_max_f16", "v_fma_f32", "v_pk_fma_f32"};
    for (int op = 0; op < 6; op++) {
        const int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
            if (op == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
            if (op == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
            if (op == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        float ticks; hipMemcpy(&ticks, d + 256 * 256, 4, hipMemcpyDeviceToHost);
        printf("%-20s %.3f ms  %.2f ns per instruction per SIMD (one wave)   s_memtime ticks per instr %.3f (100 MHz ticks)\n", names[op], ms, ms * 1e6 / (16.0 * iters), ticks / (16.0 * iters));
    }
    return 0;
}
