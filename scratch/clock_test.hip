#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(float* out, int n, unsigned long long* stamps) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < n; ++i) x = fmaf(x, 1.0001f, 0.5f);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    float* out; unsigned long long* st; hipMalloc(&out, 1 << 24); hipMalloc(&st, 1 << 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {1, 2, 256, 2048}) for (int threads : {64, 1024}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a); spin<<<blocks, threads>>>(out, 100000, st); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
            if (rep == 2) printf("blocks=%4d threads=%4d: %.1f us, cycles=%llu, realtime ticks=%llu -> clock %.0f MHz, cyc/iter %.2f\n", blocks, threads, ms * 1e3, h[0], h[1], (double)h[0] / h[1] * 100.0, h[0] / 100000.0);
        }
    }
    return 0;
}
