// How fast can 16 384 random rows of (p, m, v) be read (and written back) from 1 M-row fp32 tables at D = 64?
//   layout A: three separate [N, 64] arrays (today: 3 random 256-B accesses per row)
//   layout B: one [N, 3, 64] array (one random 768-B access per row)
// one wave per row (dword per lane), or one wave per row with 16-B lanes (B only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void gather_sep(const float* p, const float* m, const float* v, const int* idx, int n, float* out, int wr,
                                                  float* p2, float* m2, float* v2) {
    const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n) return;
    const int r = __builtin_amdgcn_readfirstlane(idx[w]);
    float a = p[(size_t)r * 64 + lane], b = m[(size_t)r * 64 + lane], c = v[(size_t)r * 64 + lane];
    a = a * 1.0001f + b; b = b * 0.9f + c; c = c * 0.999f + a;
    if (wr) { p2[(size_t)r * 64 + lane] = a; m2[(size_t)r * 64 + lane] = b; v2[(size_t)r * 64 + lane] = c; }
    else if (a + b + c == 12345.f) out[w] = a;
}
__global__ __launch_bounds__(256) void gather_il(const float* t, const int* idx, int n, float* out, int wr, float* t2) {
    const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n) return;
    const int r = __builtin_amdgcn_readfirstlane(idx[w]);
    const float* base = t + (size_t)r * 192;
    float a = base[lane], b = base[64 + lane], c = base[128 + lane];
    a = a * 1.0001f + b; b = b * 0.9f + c; c = c * 0.999f + a;
    if (wr) { float* o = t2 + (size_t)r * 192; o[lane] = a; o[64 + lane] = b; o[128 + lane] = c; }
    else if (a + b + c == 12345.f) out[w] = a;
}
// 48 lanes x 16 B = one 768-B row per wave instruction
__global__ __launch_bounds__(256) void gather_il16(const float4* t, const int* idx, int n, float* out, int wr, float4* t2) {
    const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n) return;
    const int r = __builtin_amdgcn_readfirstlane(idx[w]);
    if (lane < 48) {
        float4 a = t[(size_t)r * 48 + lane];
        a.x = a.x * 1.0001f + a.y; a.z = a.z * 0.9f + a.w;
        if (wr) t2[(size_t)r * 48 + lane] = a;
        else if (a.x + a.z == 12345.f) out[w] = a.x;
    }
}
int main() {
    const int N = 1000001, n = 16384, reps = 50;
    size_t bytes = (size_t)N * 64 * 4;
    float *p, *m, *v, *t, *out; int* idx;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&m, bytes)); CK(hipMalloc(&v, bytes)); CK(hipMalloc(&t, 3 * bytes)); CK(hipMalloc(&out, n * 4));
    CK(hipMemset(p, 0, bytes)); CK(hipMemset(m, 0, bytes)); CK(hipMemset(v, 0, bytes)); CK(hipMemset(t, 0, 3 * bytes));
    std::vector<int> h((size_t)n * reps);
    srand(3);
    for (auto& x : h) x = (int)(((long long)rand() * 32768 + rand()) % N);
    CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int wr = 0; wr < 2; ++wr)
        for (int var = 0; var < 3; ++var) {
            float ms;
            for (int pass = 0; pass < 2; ++pass) {
                CK(hipEventRecord(a));
                for (int k = 0; k < reps; ++k) {
                    const int* ix = idx + (size_t)k * n;
                    if (var == 0) hipLaunchKernelGGL(gather_sep, dim3(n / 4), dim3(256), 0, 0, p, m, v, ix, n, out, wr, p, m, v);
                    if (var == 1) hipLaunchKernelGGL(gather_il, dim3(n / 4), dim3(256), 0, 0, t, ix, n, out, wr, t);
                    if (var == 2) hipLaunchKernelGGL(gather_il16, dim3(n / 4), dim3(256), 0, 0, (const float4*)t, ix, n, out, wr, (float4*)t);
                }
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                CK(hipEventElapsedTime(&ms, a, b));
            }
            const double us = ms * 1e3 / reps, mb = n * 768.0 * (wr ? 2 : 1) / 1e6;
            printf("%s %-34s %7.2f us per launch of %d rows  (%.1f MB -> %.2f TB/s)\n", wr ? "read+write" : "read only ",
                   var == 0 ? "3 separate [N,64] arrays" : var == 1 ? "interleaved [N,3,64], dword lanes" : "interleaved [N,3,64], 16-B lanes", us, n, mb, mb / us);
        }
    return 0;
}
