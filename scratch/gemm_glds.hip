// prototype 2: fp32 MFMA GEMM, one wave per 32x32 output tile, operands staged global -> LDS by glds in full 128-B lines
// (wave-private XOR-swizzled image, no workgroup barrier), fragments by ds_read_b128.  Y[M,N] = X[M,K] W[N,K]^T
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

template <int NBUF, int KS>
__global__ __launch_bounds__(512) void gemm_glds(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y,
                                                 int M, int N, int K, int tiles_n) {
    extern __shared__ __align__(16) float lds[];   // [8 waves][NBUF][2][1024]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int TPB = 8 / KS;                    // tiles per block
    const int tile = blockIdx.x * TPB + wave % TPB;
    const int kpart = wave / TPB;
    const int mt = tile / tiles_n, nt = tile % tiles_n;
    const int m0 = mt * 32, n0 = nt * 32;
    const bool live = m0 < M;
    float* my = lds + (size_t)wave * NBUF * 2048;
    // staging: instruction i of a block brings rows 8i..8i+7; lane l -> row 8i + l/8, LDS slot l%8, source chunk slot ^ ((row>>1)&7)
    const int srow = lane >> 3, sslot = lane & 7;
    const float* xsrc[4];
    const float* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + srow;
        const int c = sslot ^ ((row >> 1) & 7);
        int xm = m0 + row; xm = xm < M ? xm : M - 1;
        int wn = n0 + row; wn = wn < N ? wn : N - 1;
        xsrc[i] = X + (size_t)xm * K + 4 * c;
        wsrc[i] = W + (size_t)wn * K + 4 * c;
    }
    auto stage = [&](int chunk, int buf) {
        float* xb = my + buf * 2048;
        float* wb = xb + 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[i] + chunk * 32), (lds_ptr_t)(xb + i * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + chunk * 32), (lds_ptr_t)(wb + i * 256), 16, 0, 0);
        }
    };
    const int r = lane & 31, h = lane >> 5;
    const int sw = (r >> 1) & 7;
    const int nchunk = live ? K / 32 / KS : 0, c0 = kpart * (K / 32 / KS);
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
        if (b < nchunk) stage(c0 + b, b);
    f32x16 acc0 = {0}, acc1 = {0};
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)my;
    unsigned ro[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ro[j] = lbase + r * 128 + (((2 * j + h) ^ sw) << 4);
    for (int t0 = 0; t0 < nchunk; t0 += NBUF) {
#pragma unroll
        for (int buf = 0; buf < NBUF; ++buf) {
            const int t = t0 + buf;
            if (t < nchunk) {
                // issue chunk t + NBUF - 1 into the buffer consumed at t - 1
                if (t + NBUF - 1 < nchunk) {
                    stage(c0 + t + NBUF - 1, (buf + NBUF - 1) % NBUF);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NBUF - 1)) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                f4 a[4], b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[j]) : "v"(ro[j]), "n"(buf * 8192));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[j]) : "v"(ro[j]), "n"(buf * 8192 + 4096));
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][0], b[j][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][1], b[j][1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][2], b[j][2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][3], b[j][3], acc1, 0, 0, 0);
                }
            }
        }
    }
    f32x16 acc = acc0 + acc1;
    if (KS > 1) {   // the K parts of a tile meet in LDS (the staging buffers are free now)
        __syncthreads();
        float* red = lds + (size_t)(wave % TPB) * 1024 * (KS - 1);
        if (kpart > 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(kpart - 1) * 1024 + i * 64 + lane] = acc[i];
        }
        __syncthreads();
        if (kpart > 0) return;
#pragma unroll
        for (int p = 0; p < KS - 1; ++p)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += red[p * 1024 + i * 64 + lane];
    }
    if (!live) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row < M && n0 + r < N) Y[(size_t)row * N + n0 + r] = acc[i];
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
    const int tiles_n = (N + 31) / 32, tiles = ((M + 31) / 32) * tiles_n;

    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 3; ++variant) {
        const int nbuf = 2, ks = variant == 0 ? 1 : (variant == 1 ? 2 : 4);
        const size_t ldsb = (size_t)8 * nbuf * 2048 * 4;
        dim3 grid((tiles + 8 / ks - 1) / (8 / ks));
        auto launch = [&]() {
            if (variant == 0) hipLaunchKernelGGL((gemm_glds<2, 1>), grid, dim3(512), ldsb, 0, X, W, Y, M, N, K, tiles_n);
            else if (variant == 1) hipLaunchKernelGGL((gemm_glds<2, 2>), grid, dim3(512), ldsb, 0, X, W, Y, M, N, K, tiles_n);
            else hipLaunchKernelGGL((gemm_glds<2, 4>), grid, dim3(512), ldsb, 0, X, W, Y, M, N, K, tiles_n);
        };
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipMemset(Y, 0, (size_t)M * N * 4);
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
        for (int t = 0; t < 400; ++t) {
            int m = t < 8 ? M - 1 - t : rand() % M, n = rand() % N; double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hx[(size_t)m * K + k] * hw[(size_t)n * K + k];
            maxerr = fmax(maxerr, fabs(s - hy[(size_t)m * N + n]));
        }
        printf("KS=%d [%d,%d]->%d: %.2f us  %.1f%% of 157 TF  max err %.2e (%s)\n", ks, M, K, N, us, fl / us / 1e6 / 157 * 100, maxerr, hipGetErrorString(hipGetLastError()));
    }
}
