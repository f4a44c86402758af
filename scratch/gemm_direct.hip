// prototype: LDS-free fp32 MFMA GEMM, one wave per 32x32 output tile, operands straight from global memory (L1/L2),
// K permuted so that a lane's 4 consecutive floats feed 4 MFMA steps.  Y[M,N] = X[M,K] W[N,K]^T
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f4 = __attribute__((ext_vector_type(4))) float;

template <int CH>   // groups of 8 k per chunk
__global__ __launch_bounds__(256) void gemm_direct(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y,
                                                   int M, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = (blockIdx.y * 4 + wave) * 32;
    if (n0 >= N) return;
    const f4* xp = reinterpret_cast<const f4*>(X + (size_t)(m0 + r) * K) + h;
    const f4* wp = reinterpret_cast<const f4*>(W + (size_t)(n0 + r) * K) + h;
    f32x16 acc0 = {0}, acc1 = {0};
    f4 xa[2][CH], wb[2][CH];
    const int G = K / 8;   // groups
#pragma unroll
    for (int g = 0; g < CH; ++g) { xa[0][g] = xp[2 * g]; wb[0][g] = wp[2 * g]; }
    for (int g0 = 0; g0 < G; g0 += 2 * CH) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int gb = g0 + half * CH;
            if (gb < G) {
                if (gb + CH < G) {
#pragma unroll
                    for (int g = 0; g < CH; ++g) { xa[half ^ 1][g] = xp[2 * (gb + CH + g)]; wb[half ^ 1][g] = wp[2 * (gb + CH + g)]; }
                }
#pragma unroll
                for (int g = 0; g < CH; ++g) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[half][g][0], wb[half][g][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[half][g][1], wb[half][g][1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[half][g][2], wb[half][g][2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[half][g][3], wb[half][g][3], acc1, 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        Y[(size_t)row * N + n0 + r] = acc0[i] + acc1[i];
    }
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 8192, K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 128;
    std::vector<float> hx((size_t)M * K), hw((size_t)N * K);
    for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) / 20000.f;
    float *X, *W, *Y;
    hipMalloc(&X, hx.size() * 4); hipMalloc(&W, hw.size() * 4); hipMalloc(&Y, (size_t)M * N * 4);
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    dim3 grid(M / 32, (N + 127) / 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant) {
        auto launch = [&]() {
            if (variant == 0) hipLaunchKernelGGL(gemm_direct<4>, grid, dim3(256), 0, 0, X, W, Y, M, N, K);
            else hipLaunchKernelGGL(gemm_direct<8>, grid, dim3(256), 0, 0, X, W, Y, M, N, K);
        };
        for (int i = 0; i < 5; ++i) launch();
        hipDeviceSynchronize();
        const int reps = 50;
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, fl = 2.0 * M * K * N;
        std::vector<float> hy((size_t)M * N);
        hipMemcpy(hy.data(), Y, hy.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0;
        for (int t = 0; t < 200; ++t) {
            int m = rand() % M, n = rand() % N; double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hx[(size_t)m * K + k] * hw[(size_t)n * K + k];
            maxerr = fmax(maxerr, fabs(s - hy[(size_t)m * N + n]));
        }
        printf("CH=%d [%d,%d]->%d: %.2f us  %.1f%% of 157 TF  max err %.2e\n", variant ? 8 : 4, M, K, N, us, fl / us / 1e6 / 157 * 100, maxerr);
    }
}
