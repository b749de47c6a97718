// What does a packed fp32 instruction read from an SGPR-pair source under op_sel / op_sel_hi?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float *out, float s_lo, float s_hi, float v_lo, float v_hi)
{
    f32x2 v = {v_lo + threadIdx.x * 0.0f, v_hi};
    f32x2 r0, r1, r2, r3;
    // s pair built from two scalar kernel arguments
    asm volatile("s_mov_b32 s40, %4\n\ts_mov_b32 s41, %5\n\t"
                 "v_pk_add_f32 %0, %6, s[40:41]\n\t"                                   // plain: (v.lo + s40, v.hi + s41)
                 "v_pk_add_f32 %1, %6, s[40:41] op_sel_hi:[1,0]\n\t"                   // both lanes take the pair's LOW register?
                 "v_pk_add_f32 %2, %6, s[40:41] op_sel:[0,1]\n\t"                      // both lanes take the pair's HIGH register?
                 "v_pk_add_f32 %3, %6, s[40:41] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "s"(s_lo), "s"(s_hi), "v"(v) : "s40", "s41");
    if (threadIdx.x == 0) { out[0] = r0.x; out[1] = r0.y; out[2] = r1.x; out[3] = r1.y; out[4] = r2.x; out[5] = r2.y; out[6] = r3.x; out[7] = r3.y; }
}
int main()
{
    float *d; hipMalloc(&d, 64); float h[8];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 10.0f, 20.0f, 1.0f, 2.0f);
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("v = (1, 2), s pair = (10, 20)\n plain             : %g %g   (expect 11 22)\n op_sel_hi:[1,0]   : %g %g   (expect 11 12 if honoured, 11 22 if ignored)\n"
           " op_sel:[0,1]      : %g %g   (expect 21 22 if honoured, 11 22 if ignored)\n op_sel:[0,1] + neg: %g %g   (expect -19 -18 if honoured, -9 -18 if ignored)\n",
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    return 0;
}
