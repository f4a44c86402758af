// diagnostic: build sort_segments.hip with stamps and print phase durations
#define FR_SORT_STAMPS 1
#include "../recbole-fairrec_amd/csrc/sort_segments.hip"
#include <vector>
#include <stdlib.h>
namespace fr { void set_error(const char*, ...) {} bool prof_on() { return false; } void* prof_begin(int, hipStream_t) { return nullptr; } void prof_end(void*, hipStream_t) {} }
int main() {
    const int M = 8192; const long long N = 1000001;
    std::vector<long long> h(M); for (auto& x : h) x = rand() % N;
    long long* idx; int *perm, *ss, *sr, *so, *ns; unsigned* err;
    hipMalloc(&idx, M * 8); hipMalloc(&perm, M * 4 + 4); hipMalloc(&ss, M * 4 + 4); hipMalloc(&sr, M * 4 + 4); hipMalloc(&so, M * 4 + 4); hipMalloc(&ns, 4); hipMalloc(&err, 4);
    hipMemcpy(idx, h.data(), M * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        fr_sort_segments((const int64_t*)idx, M, N, perm, ss, sr, so, ns, err, nullptr);
        hipDeviceSynchronize();
        unsigned long long st[16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(fr::g_sort_stamps), sizeof(st));
        printf("rep %d: load %llu | ", rep, st[1] - st[0]);
        for (int p = 0; p < 3; ++p) printf("pass%d rank %llu scan %llu scatter %llu | ", p, st[2 + 3 * p] - (p ? st[4 + 3 * (p - 1)] : st[1]), st[3 + 3 * p] - st[2 + 3 * p], st[4 + 3 * p] - st[3 + 3 * p]);
        printf("heads %llu scan %llu out %llu total %llu cycles\n", st[12] - st[10], st[13] - st[12], st[14] - st[13], st[14] - st[0]);
    }
}
