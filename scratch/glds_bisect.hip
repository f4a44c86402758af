// bisect harness: the production kernel source compiled standalone, optional -DVARIANT edits
#include "../recbole-fairrec_amd/csrc/mlp_glds.hip"
#include <vector>
#include <stdio.h>
#include <stdlib.h>
namespace fr { void set_error(const char*, ...) {} bool prof_on() { return false; } bool prof_take(int, hipEvent_t*, hipEvent_t*) { return false; } }
int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 8192, K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 128;
    float *X, *W, *Y;
    hipMalloc(&X, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&Y, (size_t)M * N * 4);
    hipMemset(X, 0, (size_t)M * K * 4); hipMemset(W, 0, (size_t)N * K * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) fr::glds_linear_fwd(fr::GlMat{X, nullptr, K, 0, K}, W, nullptr, M, N, K, 0, Y, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) fr::glds_linear_fwd(fr::GlMat{X, nullptr, K, 0, K}, W, nullptr, M, N, K, 0, Y, nullptr);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("standalone production kernel [%d,%d]->%d: %.2f us (%s)\n", M, K, N, ms * 1e3 / 50, hipGetErrorString(hipGetLastError()));
}
