// prototype 3: the LDS-DMA GEMM of gemm_glds.hip with operand chunks SHARED by the four waves of a workgroup:
// a 64 x 64 macro tile = 2 x 2 waves of 32 x 32; per 32-element reduction chunk the workgroup brings 64 x 32 of X and
// 64 x 32 of W (16 KB) for 4 tiles' worth of MFMAs -- half the L2 -> LDS bytes per flop of the wave-private form.
// One s_barrier per chunk, NBUF-deep ring.  KS: reduction parts per macro tile (KS groups of 4 waves in a workgroup).
// Y[M,N] = X[M,K] W[N,K]^T
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
__global__ __launch_bounds__(256 * KS) void gemm_glds64(const float* __restrict__ X, const float* __restrict__ W,
                                                        float* __restrict__ Y, int M, int N, int K, int tiles_n) {
    extern __shared__ __align__(16) float lds[];   // [KS][NBUF][4 sub-blocks: A0 A1 B0 B1][1024]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kpart = wave >> 2, w4 = wave & 3, wi = w4 >> 1, wj = w4 & 1;
    const int mt = blockIdx.x / tiles_n, nt = blockIdx.x % tiles_n;
    const int m0 = mt * 64, n0 = nt * 64;
    float* grp = lds + (size_t)kpart * NBUF * 4096;
    // staging: wave w4 brings sub-block w4 (0,1: rows of X; 2,3: rows of W), 4 instructions of 8 rows x 128 bytes
    const int srow = lane >> 3, sslot = lane & 7;
    const float* src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + srow;
        const int c = sslot ^ ((row >> 1) & 7);
        if (w4 < 2) {
            int xm = m0 + 32 * w4 + row; xm = xm < M ? xm : M - 1;
            src[i] = X + (size_t)xm * K + 4 * c;
        } else {
            int wn = n0 + 32 * (w4 - 2) + row; wn = wn < N ? wn : N - 1;
            src[i] = W + (size_t)wn * K + 4 * c;
        }
    }
    auto stage = [&](int chunk, int buf) {
        float* dst = grp + buf * 4096 + w4 * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + chunk * 32), (lds_ptr_t)(dst + i * 256), 16, 0, 0);
    };
    const int r = lane & 31, h = lane >> 5;
    const int sw = (r >> 1) & 7;
    const int nchunk = K / 32 / KS, c0 = kpart * nchunk;
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
        if (b < nchunk) stage(c0 + b, b);
    f32x16 acc0 = {0}, acc1 = {0};
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)grp;
    unsigned ra[4], rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned o = r * 128 + (((2 * j + h) ^ sw) << 4);
        ra[j] = lbase + wi * 4096 + o;
        rb[j] = lbase + (2 + wj) * 4096 + o;
    }
    for (int t0 = 0; t0 < nchunk; t0 += NBUF) {
#pragma unroll
        for (int buf = 0; buf < NBUF; ++buf) {
            const int t = t0 + buf;
            if (t < nchunk) {
                // own part of chunk t has landed (the newer chunks stay in flight) ...
                const int newer = nchunk - 1 - t < NBUF - 2 ? nchunk - 1 - t : NBUF - 2;
                if (newer >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // ... and everyone's: after this barrier all waves have also finished reading the buffer of chunk t - 1
                __builtin_amdgcn_s_barrier();
                if (t + NBUF - 1 < nchunk) stage(c0 + t + NBUF - 1, (buf + NBUF - 1) % NBUF);
                f4 a[4], b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[j]) : "v"(ra[j]), "n"(buf * 16384));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[j]) : "v"(rb[j]), "n"(buf * 16384));
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
    if (KS > 1) {
        __syncthreads();
        float* red = lds + (size_t)w4 * 1024 * (KS - 1);
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
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + 32 * wi + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int col = n0 + 32 * wj + r;
        if (row < M && col < N) Y[(size_t)row * N + col] = acc[i];
    }
}

template <int NBUF, int KS>
__global__ __launch_bounds__(256 * KS) void gemm_glds64p(const float* __restrict__ X, const float* __restrict__ W,
                                                        float* __restrict__ Y, int M, int N, int K, int tiles_n) {
    extern __shared__ __align__(16) float lds[];   // [KS][NBUF][4 sub-blocks: A0 A1 B0 B1][1024]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kpart = wave >> 2, w4 = wave & 3, wi = w4 >> 1, wj = w4 & 1;
    const int mt = blockIdx.x / tiles_n, nt = blockIdx.x % tiles_n;
    const int m0 = mt * 64, n0 = nt * 64;
    float* grp = lds + (size_t)kpart * NBUF * 4096;
    // staging: wave w4 brings sub-block w4 (0,1: rows of X; 2,3: rows of W), 4 instructions of 8 rows x 128 bytes
    const int srow = lane >> 3, sslot = lane & 7;
    const float* src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + srow;
        const int c = sslot ^ ((row >> 1) & 7);
        if (w4 < 2) {
            int xm = m0 + 32 * w4 + row; xm = xm < M ? xm : M - 1;
            src[i] = X + (size_t)xm * K + 4 * c;
        } else {
            int wn = n0 + 32 * (w4 - 2) + row; wn = wn < N ? wn : N - 1;
            src[i] = W + (size_t)wn * K + 4 * c;
        }
    }
    auto stage = [&](int chunk, int buf) {
        float* dst = grp + buf * 4096 + w4 * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + chunk * 32), (lds_ptr_t)(dst + i * 256), 16, 0, 0);
    };
    const int r = lane & 31, h = lane >> 5;
    const int sw = (r >> 1) & 7;
    const int nchunk = K / 32 / KS, c0 = kpart * nchunk;
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
        if (b < nchunk) stage(c0 + b, b);
    f32x16 acc0 = {0}, acc1 = {0};
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)grp;
    unsigned ra[4], rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned o = r * 128 + (((2 * j + h) ^ sw) << 4);
        ra[j] = lbase + wi * 4096 + o;
        rb[j] = lbase + (2 + wj) * 4096 + o;
    }
    // software pipeline: the fragments of chunk t + 1 are read from LDS while the MFMAs of chunk t run
    f4 fa[2][4], fb[2][4];
    auto wait_landed = [&](int t) {   // own block of chunk t has landed (chunks t+1 .. stay in flight)
        const int newer = nchunk - 1 - t < NBUF - 2 ? nchunk - 1 - t : NBUF - 2;
        if (newer >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
#define READ_FRAGS(S, T)                                                                                         \
    {                                                                                                             \
        const unsigned bo = (unsigned)((T) % NBUF) * 16384u;                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[S][j]) : "v"(ra[j] + bo));                               \
            asm volatile("ds_read_b128 %0, %1" : "=v"(fb[S][j]) : "v"(rb[j] + bo));                               \
        }                                                                                                         \
    }
#define MFMAS(S)                                                                                                  \
    {                                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                       \
                     : "+v"(fa[S][0]), "+v"(fa[S][1]), "+v"(fa[S][2]), "+v"(fa[S][3]), "+v"(fb[S][0]), "+v"(fb[S][1]), \
                       "+v"(fb[S][2]), "+v"(fb[S][3]));                                                           \
    }
#define MFMAS2(S)                                                                                                 \
    {                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S][j][0], fb[S][j][0], acc0, 0, 0, 0);                 \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S][j][1], fb[S][j][1], acc1, 0, 0, 0);                 \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S][j][2], fb[S][j][2], acc0, 0, 0, 0);                 \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S][j][3], fb[S][j][3], acc1, 0, 0, 0);                 \
        }                                                                                                         \
    }
    if (nchunk > 0) {
        wait_landed(0);
        __builtin_amdgcn_s_barrier();
        if (NBUF - 1 < nchunk) stage(c0 + NBUF - 1, NBUF - 1);
        READ_FRAGS(0, 0)
    }
    for (int t = 0; t < nchunk; t += 2) {
        // ---- even step: fragments of chunk t are in set 0 (reads in flight) ----
        MFMAS(0)                                   // set 0 has arrived; every wave passing here is done reading buffer t
        if (t + 1 < nchunk) {
            wait_landed(t + 1);
            __builtin_amdgcn_s_barrier();          // chunk t + 1 visible; buffer of chunk t free
            if (t + NBUF < nchunk) stage(c0 + t + NBUF, t % NBUF);
            READ_FRAGS(1, t + 1)
        }
        MFMAS2(0)
        if (t + 1 < nchunk) {
            MFMAS(1)
            if (t + 2 < nchunk) {
                wait_landed(t + 2);
                __builtin_amdgcn_s_barrier();
                if (t + 1 + NBUF < nchunk) stage(c0 + t + 1 + NBUF, (t + 1) % NBUF);
                READ_FRAGS(0, t + 2)
            }
            MFMAS2(1)
        }
    }
    f32x16 acc = acc0 + acc1;
    if (KS > 1) {
        __syncthreads();
        float* red = lds + (size_t)w4 * 1024 * (KS - 1);
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
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + 32 * wi + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int col = n0 + 32 * wj + r;
        if (row < M && col < N) Y[(size_t)row * N + col] = acc[i];
    }
}

template <int NBUF, int KS>
static void run(const float* X, const float* W, float* Y, int M, int N, int K, const std::vector<float>& hx,
                const std::vector<float>& hw) {
    const int tiles_n = (N + 63) / 64, tiles = ((M + 63) / 64) * tiles_n;
    const size_t ldsb = (size_t)KS * NBUF * 4096 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds64<NBUF, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto launch = [&]() { hipLaunchKernelGGL((gemm_glds64<NBUF, KS>), dim3(tiles), dim3(256 * KS), ldsb, 0, X, W, Y, M, N, K, tiles_n); };
    hipMemset(Y, 0, (size_t)M * N * 4);
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
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
    for (int t = 0; t < 600; ++t) {
        int m = t < 8 ? M - 1 - t : rand() % M, n = t < 16 ? N - 1 - (t % 8) : rand() % N; double s = 0;
        for (int k = 0; k < K; ++k) s += (double)hx[(size_t)m * K + k] * hw[(size_t)n * K + k];
        maxerr = fmax(maxerr, fabs(s - hy[(size_t)m * N + n]));
    }
    printf("shared 64x64 NBUF=%d KS=%d [%d,%d]->%d: %.2f us  %.1f%% of 157 TF  max err %.2e (%s)\n", NBUF, KS, M, K, N, us,
           fl / us / 1e6 / 157 * 100, maxerr, hipGetErrorString(hipGetLastError()));
}

template <int NBUF, int KS>
static void runp(const float* X, const float* W, float* Y, int M, int N, int K, const std::vector<float>& hx,
                const std::vector<float>& hw) {
    const int tiles_n = (N + 63) / 64, tiles = ((M + 63) / 64) * tiles_n;
    const size_t ldsb = (size_t)KS * NBUF * 4096 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds64p<NBUF, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto launch = [&]() { hipLaunchKernelGGL((gemm_glds64p<NBUF, KS>), dim3(tiles), dim3(256 * KS), ldsb, 0, X, W, Y, M, N, K, tiles_n); };
    hipMemset(Y, 0, (size_t)M * N * 4);
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
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
    for (int t = 0; t < 600; ++t) {
        int m = t < 8 ? M - 1 - t : rand() % M, n = t < 16 ? N - 1 - (t % 8) : rand() % N; double s = 0;
        for (int k = 0; k < K; ++k) s += (double)hx[(size_t)m * K + k] * hw[(size_t)n * K + k];
        maxerr = fmax(maxerr, fabs(s - hy[(size_t)m * N + n]));
    }
    printf("shared+pipelined 64x64 NBUF=%d KS=%d [%d,%d]->%d: %.2f us  %.1f%% of 157 TF  max err %.2e (%s)\n", NBUF, KS, M, K, N, us,
           fl / us / 1e6 / 157 * 100, maxerr, hipGetErrorString(hipGetLastError()));
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
    run<3, 1>(X, W, Y, M, N, K, hx, hw);
    runp<3, 1>(X, W, Y, M, N, K, hx, hw);
    runp<4, 1>(X, W, Y, M, N, K, hx, hw);
    if (K / 32 % 2 == 0) runp<3, 2>(X, W, Y, M, N, K, hx, hw);
    run<4, 1>(X, W, Y, M, N, K, hx, hw);
    if (K / 32 % 2 == 0) { run<3, 2>(X, W, Y, M, N, K, hx, hw); run<4, 2>(X, W, Y, M, N, K, hx, hw); }
    if (K / 32 % 4 == 0) run<3, 4>(X, W, Y, M, N, K, hx, hw);
}
