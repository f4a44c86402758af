// Where do the blocks of a sub-residency grid land?  Each wave records (xcc, se, sh, cu, simd) and spins ~20 us.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void census(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = hw; out[2 * w + 1] = xcc;
    }
}
int main(int argc, char** argv) {
    int nb = argc > 1 ? atoi(argv[1]) : 1125;
    unsigned* d; hipMalloc(&d, nb * 4 * 8);
    hipLaunchKernelGGL(census, dim3(nb), dim3(256), 0, 0, d, 40000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8); hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu, per_simd;
    for (int w = 0; w < nb * 4; ++w) {
        unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
        unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        unsigned cukey = (xcc << 8) | (se << 5) | (sh << 4) | cu;
        per_cu[cukey]++; per_simd[(cukey << 2) | simd]++;
    }
    std::map<int, int> hist_cu, hist_simd;
    for (auto& kv : per_cu) hist_cu[kv.second]++;
    for (auto& kv : per_simd) hist_simd[kv.second]++;
    printf("blocks %d: distinct CUs %zu, distinct SIMDs %zu\n waves per CU histogram:", nb, per_cu.size(), per_simd.size());
    for (auto& kv : hist_cu) printf(" %d:%d", kv.first, kv.second);
    printf("\n waves per SIMD histogram:");
    for (auto& kv : hist_simd) printf(" %d:%d", kv.first, kv.second);
    printf("\n first 12 blocks (xcc,se,sh,cu,simd of wave0):");
    for (int b = 0; b < 12; ++b) { unsigned hw = h[8 * b], x = h[8 * b + 1] & 0xf; printf(" (%u,%u,%u,%u,%u)", x, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf, (hw >> 4) & 3); }
    printf("\n");
    return 0;
}
