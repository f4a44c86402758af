// Does this runtime honour hipExtAnyOrderLaunch on gfx950 (AQL barrier bit cleared: the dispatcher may start a kernel's
// workgroups while the previous kernel of the SAME stream still has waves running)?
//   hipcc --offload-arch=gfx950 -O2 scratch/anyorder.hip -o scratch/bin/anyorder && scratch/bin/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// every wave spins for `ticks[blockIdx]` 10 ns ticks; records (start, end) of the workgroup
#ifndef HEAVY
#define HEAVY 0
#endif
struct BigArgs { unsigned long long pad[100]; };
__global__
#if HEAVY & 1
__launch_bounds__(256, 6)
#endif
void spin_kernel(const int* ticks, unsigned long long* stamps, int slot
#if HEAVY & 4
                 , BigArgs big
#endif
) {
#if HEAVY & 1
    asm volatile("v_mov_b32 v79, 0" ::: "v79");      // 80 VGPRs: six waves per SIMD
#endif
#if HEAVY & 2
    __shared__ int lds_word[8];
    if (threadIdx.x == 0) lds_word[0] = slot;
    __syncthreads();
    if (lds_word[0] == 0x7fffffff) return;
#endif
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long want = (unsigned long long)ticks[blockIdx.x];
    while (__builtin_amdgcn_s_memrealtime() - t0 < want) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) {
        stamps[2 * ((size_t)slot * gridDim.x + blockIdx.x)] = t0;
        stamps[2 * ((size_t)slot * gridDim.x + blockIdx.x) + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

// as above, but first waits (bounded) until the previous launch's `done` counter has reached `need`; adds one at its end
__global__ void chain_kernel(const int* ticks, unsigned long long* stamps, int slot, unsigned* done_prev, unsigned need,
                             unsigned* done_mine, unsigned* timeouts) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (done_prev) {
        int spins = 0;
        while (__hip_atomic_load(done_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 20)) { if (threadIdx.x == 0) atomicAdd(timeouts, 1u); break; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long want = (unsigned long long)ticks[blockIdx.x];
    while (__builtin_amdgcn_s_memrealtime() - t1 < want) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[2 * ((size_t)slot * gridDim.x + blockIdx.x)] = t0;
        stamps[2 * ((size_t)slot * gridDim.x + blockIdx.x) + 1] = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_fetch_add(done_mine, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const int NB = argc > 1 ? atoi(argv[1]) : 2200, NL = argc > 2 ? atoi(argv[2]) : 64;
    std::vector<int> ticks(NB);
    srand(1);
    for (int i = 0; i < NB; ++i) ticks[i] = 300 + (rand() % 100 < 3 ? 1500 : rand() % 600);     // 3-9 us, 3 % of them 18 us
    int* d_ticks;
    unsigned long long* d_st;
    unsigned *d_done, *d_to;
    CK(hipMalloc(&d_ticks, NB * sizeof(int)));
    CK(hipMalloc(&d_st, (size_t)NL * NB * 16));
    CK(hipMalloc(&d_done, (NL + 1) * 256));
    CK(hipMalloc(&d_to, 4));
    CK(hipMemcpy(d_ticks, ticks.data(), NB * sizeof(int), hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<unsigned long long> h((size_t)NL * NB * 2);
    for (int mode = 0; mode < 6; ++mode) {      // 0: plain launches  1: any-order  2: chain, plain  3: chain, any-order  4: spin, ODD launches any-order  5: spin, all but every 4th
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(d_done, 0, (NL + 1) * 256, st));
            CK(hipMemsetAsync(d_to, 0, 4, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int l = 0; l < NL; ++l) {
                const int flags = mode == 4 ? ((l & 1) ? hipExtAnyOrderLaunch : 0) : mode == 5 ? ((l & 3) ? hipExtAnyOrderLaunch : 0) : (mode & 1) ? hipExtAnyOrderLaunch : 0;
                if (mode < 2 || mode >= 4) {
#if HEAVY & 4
                    hipExtLaunchKernelGGL(spin_kernel, dim3(NB), dim3(256), 0, st, nullptr, nullptr, flags, (const int*)d_ticks, d_st, l, BigArgs{});
#else
                    hipExtLaunchKernelGGL(spin_kernel, dim3(NB), dim3(256), 0, st, nullptr, nullptr, flags, (const int*)d_ticks, d_st, l);
#endif
                } else {
                    hipExtLaunchKernelGGL(chain_kernel, dim3(NB), dim3(256), 0, st, nullptr, nullptr, flags, (const int*)d_ticks, d_st, l,
                                          l >= 2 ? d_done + 64 * (l - 2) : (unsigned*)nullptr, (unsigned)NB, d_done + 64 * l, d_to);
                }
            }
            CK(hipGetLastError());
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned to;
            CK(hipMemcpy(&to, d_to, 4, hipMemcpyDeviceToHost));
            // per launch: first start, last end; overlap = launches whose first start precedes the previous launch's last end
            int overl = 0;
            double gap = 0, span = 0;
            unsigned long long prev_end = 0;
            for (int l = 0; l < NL; ++l) {
                unsigned long long s = ~0ull, e = 0;
                for (int b = 0; b < NB; ++b) {
                    s = std::min(s, h[2 * ((size_t)l * NB + b)]);
                    e = std::max(e, h[2 * ((size_t)l * NB + b) + 1]);
                }
                if (l) {
                    if (s < prev_end) ++overl;
                    gap += (double)((long long)s - (long long)prev_end) * 0.01;
                }
                span += (double)(e - s) * 0.01;
                prev_end = e;
            }
            if (mode == 4 && rep == 2)
                for (int l = 8; l < 14; ++l) {
                    unsigned long long s0 = ~0ull, s1 = 0, e = 0;
                    for (int b = 0; b < NB; ++b) {
                        s0 = std::min(s0, h[2 * ((size_t)l * NB + b)]);
                        s1 = std::max(s1, h[2 * ((size_t)l * NB + b)]);
                        e = std::max(e, h[2 * ((size_t)l * NB + b) + 1]);
                    }
                    static unsigned long long base = 0;
                    if (!base) base = s0;
                    printf("   launch %d (%s): first start %.2f, last workgroup start %.2f, last end %.2f us\n", l, (l & 1) ? "any-order" : "plain",
                           (s0 - base) * 0.01, (s1 - base) * 0.01, (e - base) * 0.01);
                }
            printf("mode %d rep %d: %.2f us per launch (event), span %.2f us, first-start minus prev-last-end %.2f us, "
                   "overlapping launches %d / %d, wait timeouts %u\n", mode, rep, ms * 1000.0 / NL, span / NL, gap / (NL - 1), overl,
                   NL - 1, to);
        }
    }
    return 0;
}
