// wstream_probe.hip -- what does a weight stream L2 -> LDS cost a one-wave-per-SIMD matrix kernel, by the way it is issued?
// The weight-streaming kernels (classic NeRF, LeRF kernels A / B, the exact sigma kernels) run 4 waves per workgroup, one per SIMD, and stream one-tile weight chunks
// (32 fragments of 1 KB) through three LDS buffers two chunks ahead, the pieces dealt out between the matrix instructions.  This probe keeps that skeleton -- per chunk
// 16 k-steps x (2 ds_read_b128, 3 matrix instructions), 8 pieces per wave, counted wait + barrier at the end of a chunk -- and swaps the way a piece travels:
//   mode 0  no stream at all (the buffers hold what they hold)
//   mode 1  LDS-DMA: global_load_lds_dwordx4 (what the kernels do)
//   mode 2  global_load_dwordx4 into registers during chunk c, ds_write_b128 during chunk c + 1 (for chunk c + 2)
//   mode 3  the loads of mode 2 without the LDS writes (the load half alone)
//   mode 4  the LDS writes of mode 2 without the loads (the write half alone)
//   hipcc -O3 --offload-arch=gfx950 wstream_probe.hip -o wstream_probe && ./wstream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#ifndef KSTEPS
#define KSTEPS 16
#endif
#ifndef AHEAD
#define AHEAD 2                           // chunks the stream runs ahead (ring of AHEAD + 1 buffers)
#endif
constexpr int KST = KSTEPS;               // k-steps per chunk (kernel A's sigma0 chunks: 8, two products each)
constexpr int FR = 2 * KST;               // fragments per chunk (k-steps x (hi, lo)), 1 KB each
constexpr int NW = 4;
constexpr int PIECES = FR / NW;           // 8 per wave per chunk
constexpr int CHUNKS_IN_IMAGE = 40;       // 1.25 MB image: L2-resident, as the kernels' weight images are

template <int MODE>
__global__ void __launch_bounds__(64 * NW) k(const half8 *__restrict__ image, int chunks, float *out, long long *cycles)
{
    constexpr int RING = AHEAD + 1;
    __shared__ half8 buf[RING][FR * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RING * FR * 64; i += blockDim.x) (&buf[0][0])[i] = image[i];
    __syncthreads();
    f32x16 acc;
    for (int j = 0; j < 16; j++) acc[j] = (float)(lane + j);
    half8 bh, bl;
    for (int j = 0; j < 8; j++) { bh[j] = (_Float16)(1.0f + 0.001f * (lane + j)); bl[j] = (_Float16)(0.001f * (lane - j)); }
    half8 regs[2][PIECES];
    for (int p = 0; p < PIECES; p++) { regs[0][p] = bh; regs[1][p] = bl; }
    const long long t0 = clock64();
    // chunk c computes from buf[c % 3]; chunk c + 2's pieces are issued during chunk c
    auto chunk = [&](int c, auto parity) {
        constexpr int PAR = decltype(parity)::value;
        const half8 *w = buf[c % RING];
        half8 *dst = buf[(c + AHEAD) % RING];
        half8 *dst1 = buf[(c + 1) % RING];
        const half8 *src = image + (size_t)((c + AHEAD) % CHUNKS_IN_IMAGE) * FR * 64;
        // A fragments two k-steps ahead (as the kernels read them), so the LDS latency is not what a k-step waits for
        half8 ah[KST + 2], al[KST + 2];
        ah[0] = w[0 * 64 + lane]; al[0] = w[1 * 64 + lane];
        ah[1] = w[2 * 64 + lane]; al[1] = w[3 * 64 + lane];
#pragma unroll
        for (int ks = 0; ks < KST; ks++) {
            if (ks + 2 < KST) { ah[ks + 2] = w[(2 * ks + 4) * 64 + lane]; al[ks + 2] = w[(2 * ks + 5) * 64 + lane]; }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, acc, 0, 0, 0);
            if ((ks & 1) == 0) {
                const int q = ks >> 1;
                const int f = q * NW + wave;
                if constexpr (MODE == 1) __builtin_amdgcn_global_load_lds(src + f * 64 + lane, (__attribute__((address_space(3))) void *)(dst + f * 64), 16, 0, 0);
                if constexpr (MODE == 2 || MODE == 3) {
                    // this chunk: write what the previous chunk loaded (for chunk c + 1), then load chunk c + 2's piece into the other register set
                    if constexpr (MODE == 2) dst1[f * 64 + lane] = regs[PAR ^ 1][q];
                    regs[PAR][q] = __builtin_nontemporal_load(src + f * 64 + lane);
                }
                if constexpr (MODE == 4) dst1[f * 64 + lane] = regs[PAR ^ 1][q];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (AHEAD - 1)) : "memory");
        __syncthreads();
    };
    for (int c = 0; c + 1 < chunks; c += 2) {
        chunk(c, std::integral_constant<int, 0>{});
        chunk(c + 1, std::integral_constant<int, 1>{});
    }
    const long long t1 = clock64();
    float r = 0.0f;
    for (int j = 0; j < 16; j++) r += acc[j];
    for (int p = 0; p < PIECES; p++) r += (float)regs[0][p][0] + (float)regs[1][p][1];
    if (r == 123.456f) out[threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int MODE>
static void run(const half8 *img, float *out, long long *cyc, int chunks, const char *what)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * NW), 0, 0, img, 64, out, cyc);
    hipDeviceSynchronize();
    float best = 1e9f; long long c = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * NW), 0, 0, img, chunks, out, cyc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); }
    }
    printf("mode %d  %-58s %8.3f ms   %7.0f clock64 ticks per chunk (%d matrix instructions = %d pipe cycles; %d chunks ahead)\n", MODE, what, best, (double)c / chunks, 3 * KST, 96 * KST, AHEAD);
}

int main()
{
    const size_t n = (size_t)CHUNKS_IN_IMAGE * FR * 64;
    std::vector<half8> h(n);
    for (size_t i = 0; i < n; i++) for (int j = 0; j < 8; j++) h[i][j] = (_Float16)(0.001f * (float)((i + j) % 97));
    half8 *img; float *out; long long *cyc;
    hipMalloc(&img, n * sizeof(half8)); hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    hipMemcpy(img, h.data(), n * sizeof(half8), hipMemcpyHostToDevice);
    const int chunks = 4000;
    for (int rep = 0; rep < 2; rep++) {
        run<0>(img, out, cyc, chunks, "no stream");
        run<1>(img, out, cyc, chunks, "LDS-DMA (global_load_lds_dwordx4)");
        run<2>(img, out, cyc, chunks, "global_load_dwordx4 -> registers -> ds_write_b128");
        run<3>(img, out, cyc, chunks, "  the loads alone");
        run<4>(img, out, cyc, chunks, "  the LDS writes alone");
    }
    return 0;
}
