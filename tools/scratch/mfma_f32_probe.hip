// mfma_f32_probe.hip -- what does v_mfma_f32_32x32x2_f32 compute, bit for bit?
//   D[i][j] = C[i][j] + sum_k A[i][k] B[k][j], k = 0,1 inside one instruction, K/2 instructions chained through the accumulator.
// Candidates compared against the device result (all in fp32):
//   asc   : acc = fmaf(a[k], b[k], acc) for k = 0,1,2,...            (ascending k, one rounding per term)
//   swap  : same chain with the two k of an instruction exchanged     (k = 1,0,3,2,...)
//   pair  : acc = acc + (a0*b0 + a1*b1) with the pair summed first
// Also checks fp32 denormal products / results (flushed or kept).
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_f32_probe mfma_f32_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int K = 64;

__global__ void k_probe(const float *A, const float *B, float *D)   // A [32][K], B [K][32], D [32][32]
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < K / 2; s++) {
        const float a = A[r * K + 2 * s + h];
        const float b = B[(2 * s + h) * 32 + r];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int q = 0; q < 16; q++) {
        const int row = 8 * (q >> 2) + 4 * h + (q & 3);
        D[row * 32 + r] = acc[q];
    }
}

int main()
{
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    std::vector<float> A(32 * K), B(K * 32), D(32 * 32);
    for (int trial = 0; trial < 3; trial++) {
        for (auto &v : A) v = u(rng) * std::ldexp(1.0f, (int)(u(rng) * 8));
        for (auto &v : B) v = u(rng) * std::ldexp(1.0f, (int)(u(rng) * 8));
        if (trial == 2) {   // denormal territory: products ~ 2^-140, sums stay denormal
            for (auto &v : A) v = u(rng) * std::ldexp(1.0f, -70);
            for (auto &v : B) v = u(rng) * std::ldexp(1.0f, -70);
        }
        float *dA, *dB, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        int eq_asc = 0, eq_swap = 0, eq_pair = 0, nz = 0;
        for (int i = 0; i < 32; i++)
            for (int j = 0; j < 32; j++) {
                float asc = 0.0f, sw = 0.0f, pr = 0.0f;
                for (int k = 0; k < K; k++) asc = std::fmaf(A[i * K + k], B[k * 32 + j], asc);
                for (int s = 0; s < K / 2; s++) {
                    sw = std::fmaf(A[i * K + 2 * s + 1], B[(2 * s + 1) * 32 + j], sw);
                    sw = std::fmaf(A[i * K + 2 * s], B[(2 * s) * 32 + j], sw);
                    pr = pr + std::fmaf(A[i * K + 2 * s + 1], B[(2 * s + 1) * 32 + j], A[i * K + 2 * s] * B[(2 * s) * 32 + j]);
                }
                const float d = D[i * 32 + j];
                eq_asc += std::memcmp(&d, &asc, 4) == 0;
                eq_swap += std::memcmp(&d, &sw, 4) == 0;
                eq_pair += std::memcmp(&d, &pr, 4) == 0;
                nz += d != 0.0f;
            }
        printf("trial %d: of 1024 outputs  == ascending fmaf chain: %d   == swapped-pair chain: %d   == pair-first: %d   nonzero: %d\n", trial, eq_asc, eq_swap,
               eq_pair, nz);
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    return 0;
}
