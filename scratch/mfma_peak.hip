// calibration: pure v_mfma_f32_32x32x2_f32 loop, W waves per SIMD, NACC independent accumulators
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-9f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) for (int nacc : {1, 2, 4}) {
        const int iters = 4096 / nacc;   // 4096 MFMAs per wave
        auto launch = [&]() {
            if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
            else if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
            else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0, 0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 100.0, fl = (double)blocks * 4 * 4096 * 4096;
        printf("blocks %4d (waves/SIMD %d) nacc %d: %.1f us  %.1f TF  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", blocks, blocks / 256, nacc, us, fl / us / 1e6, us * 2400.0 / (4096.0 * blocks / 256));
    }
}
