// Micro-benchmark: cost per replayed zero-gradient Adam step (64-element row) for several formulations.
//   hipcc -O3 --offload-arch=gfx950 scratch/replay_bench.hip -o gpurun_out/replay_bench && ./gpurun_out/replay_bench
// Every variant starts from the same (p, m', v') state reached by NINIT exact steps and replays K more steps.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(4))) * ConstFPtr;

struct Hyp {
    const float* sc;   // [cap+1][4] : step_size, ib, A, B
    float b1, b2;
};

// ---- V0: current kernel's step, one row per wave ----
__device__ __forceinline__ void step_exact(float& p, float& m, float& v, float A, float B, float b1, float b2) {
    m = fmaf(b1, m, p);
    v = fmaf(b2, v, p * p);
    const float den = fmaf(__builtin_amdgcn_sqrtf(v), A, B);
    p = fmaf(-m, __builtin_amdgcn_rcpf(den), p);
}

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ void step_exact2(v2f& p, v2f& m, v2f& v, float A, float B, float b1, float b2) {
    m = pk_fma(v2f{b1, b1}, m, p);
    v = pk_fma(v2f{b2, b2}, v, p * p);
    v2f sq = {__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)};
    v2f den = pk_fma(sq, v2f{A, A}, v2f{B, B});
    v2f r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    p = pk_fma(-m, r, p);
}

// rcp by one Newton step from the previous step's reciprocal
__device__ __forceinline__ void step_nr_rcp2(v2f& p, v2f& m, v2f& v, v2f& r, float A, float B, float b1, float b2) {
    m = pk_fma(v2f{b1, b1}, m, p);
    v = pk_fma(v2f{b2, b2}, v, p * p);
    v2f sq = {__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)};
    v2f den = pk_fma(sq, v2f{A, A}, v2f{B, B});
    v2f e = pk_fma(-den, r, v2f{1.f, 1.f});
    r = pk_fma(r, e, r);
    p = pk_fma(-m, r, p);
}

// rsq transcendental, reciprocal of (A*sqrt(v)+B) by Newton
__device__ __forceinline__ void step_rsq_nr2(v2f& p, v2f& m, v2f& v, v2f& r, float A, float B, float b1, float b2) {
    m = pk_fma(v2f{b1, b1}, m, p);
    v = pk_fma(v2f{b2, b2}, v, p * p);
    v2f y = {__builtin_amdgcn_rsqf(v.x), __builtin_amdgcn_rsqf(v.y)};
    v2f den = pk_fma(v * y, v2f{A, A}, v2f{B, B});
    v2f e = pk_fma(-den, r, v2f{1.f, 1.f});
    r = pk_fma(r, e, r);
    p = pk_fma(-m, r, p);
}

// no transcendental at all: z ~ 1/sqrt(2 v) by Newton (z <- z (1.5 - v z^2) form with the 1/sqrt2 folded into A)
__device__ __forceinline__ void step_all_nr2(v2f& p, v2f& m, v2f& v, v2f& z, v2f& r, float A2, float B, float b1, float b2) {
    m = pk_fma(v2f{b1, b1}, m, p);
    v = pk_fma(v2f{b2, b2}, v, p * p);
    v2f t = z * z;
    v2f h = pk_fma(-v, t, v2f{0.5f, 0.5f});
    z = pk_fma(z, h, z);
    v2f den = pk_fma(v * z, v2f{A2, A2}, v2f{B, B});     // sqrt(v) = sqrt2 * v * z ; A2 = A * sqrt2
    v2f e = pk_fma(-den, r, v2f{1.f, 1.f});
    r = pk_fma(r, e, r);
    p = pk_fma(-m, r, p);
}

template <int VAR>
__global__ __launch_bounds__(256) void replay_kernel(float* P, float* M, float* V, int n_rows, int j0, int K, Hyp h,
                                                     unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    ConstFPtr sc = (ConstFPtr)h.sc;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (VAR == 0) {
        const int row = wv;
        if (row >= n_rows) return;
        float p = P[(size_t)row * 64 + lane], m = M[(size_t)row * 64 + lane], v = V[(size_t)row * 64 + lane];
        for (int j = j0 + 1; j <= j0 + K; ++j) step_exact(p, m, v, sc[4 * j + 2], sc[4 * j + 3], h.b1, h.b2);
        P[(size_t)row * 64 + lane] = p; M[(size_t)row * 64 + lane] = m; V[(size_t)row * 64 + lane] = v;
    } else {
        const int row = 2 * wv;
        if (row + 1 >= n_rows) return;
        v2f p = {P[(size_t)row * 64 + lane], P[(size_t)(row + 1) * 64 + lane]};
        v2f m = {M[(size_t)row * 64 + lane], M[(size_t)(row + 1) * 64 + lane]};
        v2f v = {V[(size_t)row * 64 + lane], V[(size_t)(row + 1) * 64 + lane]};
        int j = j0 + 1;
        if (VAR == 1) {
            for (; j <= j0 + K; ++j) step_exact2(p, m, v, sc[4 * j + 2], sc[4 * j + 3], h.b1, h.b2);
        } else {
            // first step exact, which also seeds the Newton state
            float A = sc[4 * j + 2], B = sc[4 * j + 3];
            m = pk_fma(v2f{h.b1, h.b1}, m, p);
            v = pk_fma(v2f{h.b2, h.b2}, v, p * p);
            v2f y = {__builtin_amdgcn_rsqf(v.x), __builtin_amdgcn_rsqf(v.y)};
            v2f den = pk_fma(v * y, v2f{A, A}, v2f{B, B});
            v2f r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            p = pk_fma(-m, r, p);
            v2f z = y * 0.70710678118654752f;
            ++j;
            if (VAR == 2) for (; j <= j0 + K; ++j) step_nr_rcp2(p, m, v, r, sc[4 * j + 2], sc[4 * j + 3], h.b1, h.b2);
            if (VAR == 3) for (; j <= j0 + K; ++j) step_rsq_nr2(p, m, v, r, sc[4 * j + 2], sc[4 * j + 3], h.b1, h.b2);
            if (VAR == 4) for (; j <= j0 + K; ++j) step_all_nr2(p, m, v, z, r, sc[4 * j + 2] * 1.41421356237309505f, sc[4 * j + 3], h.b1, h.b2);
        }
        P[(size_t)row * 64 + lane] = p.x; P[(size_t)(row + 1) * 64 + lane] = p.y;
        M[(size_t)row * 64 + lane] = m.x; M[(size_t)(row + 1) * 64 + lane] = m.y;
        V[(size_t)row * 64 + lane] = v.x; V[(size_t)(row + 1) * 64 + lane] = v.y;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wv == 0) cyc[0] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n_rows = 256 * 32 * 2 * 4;   // 65536 rows: 4 residencies of single-row waves
    const int K = argc > 1 ? atoi(argv[1]) : 128;
    const int NINIT = argc > 2 ? atoi(argv[2]) : 3000;
    const double lr = 1e-3, b1 = 0.9, b2 = 0.999, eps = 1e-8, wd = 1e-3;
    const int cap = NINIT + K + 8;
    std::vector<float> sc(4 * (cap + 1), 0.f);
    const double k1 = (1 - b1) * wd, k2 = (1 - b2) * wd * wd;
    for (int j = 1; j <= cap; ++j) {
        double ss = lr / (1 - pow(b1, j)), ib = 1 / sqrt(1 - pow(b2, j));
        sc[4 * j] = (float)ss; sc[4 * j + 1] = (float)ib;
        sc[4 * j + 2] = (float)(sqrt(k2) * ib / (ss * k1)); sc[4 * j + 3] = (float)(eps / (ss * k1));
    }
    float* dsc; CK(hipMalloc(&dsc, sc.size() * 4)); CK(hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    Hyp h{dsc, (float)b1, (float)b2};
    const size_t n = (size_t)n_rows * 64;
    std::vector<float> p0(n), z0(n, 0.f);
    srand(1);
    for (size_t i = 0; i < n; ++i) {   // Box-Muller, sigma = 1.4e-3 (xavier of the 1M x 64 table)
        double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
        p0[i] = (float)(1.4e-3 * sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2));
    }
    float *P0, *M0, *V0, *P, *M, *V; unsigned long long* cyc;
    CK(hipMalloc(&P0, n * 4)); CK(hipMalloc(&M0, n * 4)); CK(hipMalloc(&V0, n * 4));
    CK(hipMalloc(&P, n * 4)); CK(hipMalloc(&M, n * 4)); CK(hipMalloc(&V, n * 4)); CK(hipMalloc(&cyc, 8));
    CK(hipMemcpy(P0, p0.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(M0, z0.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(V0, z0.data(), n * 4, hipMemcpyHostToDevice));
    // state after NINIT exact steps (scaled moments)
    if (NINIT > 0) hipLaunchKernelGGL(replay_kernel<0>, dim3(n_rows / 4), dim3(256), 0, 0, P0, M0, V0, n_rows, 0, NINIT, h, cyc);
    CK(hipDeviceSynchronize());
    std::vector<float> ref(n), got(n);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[5] = {"V0 exact 1row/wave", "V1 exact packed 2rows", "V2 packed sqrt + NR rcp", "V3 packed rsq + NR rcp", "V4 packed all-NR"};
    for (int var = 0; var < 5; ++var) {
        float best = 1e9f; unsigned long long c = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemcpy(P, P0, n * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(M, M0, n * 4, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(V, V0, n * 4, hipMemcpyDeviceToDevice));
            CK(hipEventRecord(a));
            const int grid = var == 0 ? n_rows / 4 : n_rows / 8;
            switch (var) {
                case 0: hipLaunchKernelGGL(replay_kernel<0>, dim3(grid), dim3(256), 0, 0, P, M, V, n_rows, NINIT, K, h, cyc); break;
                case 1: hipLaunchKernelGGL(replay_kernel<1>, dim3(grid), dim3(256), 0, 0, P, M, V, n_rows, NINIT, K, h, cyc); break;
                case 2: hipLaunchKernelGGL(replay_kernel<2>, dim3(grid), dim3(256), 0, 0, P, M, V, n_rows, NINIT, K, h, cyc); break;
                case 3: hipLaunchKernelGGL(replay_kernel<3>, dim3(grid), dim3(256), 0, 0, P, M, V, n_rows, NINIT, K, h, cyc); break;
                case 4: hipLaunchKernelGGL(replay_kernel<4>, dim3(grid), dim3(256), 0, 0, P, M, V, n_rows, NINIT, K, h, cyc); break;
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
            CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        }
        CK(hipMemcpy(got.data(), P, n * 4, hipMemcpyDeviceToHost));
        if (var == 0) ref = got;
        double maxabs = 0, maxrel = 0, maxtol = 0;
        for (size_t i = 0; i < n; ++i) {
            double d = fabs((double)got[i] - ref[i]);
            if (d > maxabs) maxabs = d;
            double rel = d / (fabs(ref[i]) + 1e-30);
            if (fabs(ref[i]) > 1e-5 && rel > maxrel) maxrel = rel;
            double tol = d / (1e-4 * fabs(ref[i]) + 1e-6);
            if (tol > maxtol) maxtol = tol;
        }
        // SIMD cycles per 64-element row-step, assuming 1024 SIMDs at 2.4 GHz fully occupied
        double cyc_per = best * 1e-3 * 2.4e9 * 1024 / ((double)n_rows * K);
        printf("%-26s %8.3f ms  %6.2f simd-cyc/row-step (at 2.4GHz)  wave0 %llu cyc (%.1f/step)  max|d| %.3g  maxrel %.3g  d/tol %.3g\n",
               names[var], best, cyc_per, c, (double)c / K, maxabs, maxrel, maxtol);
    }
    return 0;
}
