// diagnostic: build sort_segments.hip with stamps and print phase durations (plain job and the fused-step job)
#define FR_SORT_STAMPS 1
#include "../recbole-fairrec_amd/csrc/sort_segments.hip"
#include <vector>
#include <stdlib.h>
namespace fr { void set_error(const char*, ...) {} bool prof_on() { return false; } void* prof_begin(int, hipStream_t) { return nullptr; } void prof_end(void*, hipStream_t) {} }
int main() {
    const int M = 8192; const long long N = 1000001, NI = 100001;
    std::vector<long long> h(M), hi(M); for (auto& x : h) x = rand() % N; for (auto& x : hi) x = rand() % NI;
    long long *idx, *idx2; int *perm, *ss, *sr, *so, *ns, *stamp; unsigned *err, *cnt; float *f0, *aux, *mm; int4* rec; int4* info;
    hipMalloc(&idx, M * 8); hipMalloc(&idx2, M * 8); hipMalloc(&perm, M * 4 + 4); hipMalloc(&ss, M * 4 + 4); hipMalloc(&sr, M * 4 + 4); hipMalloc(&so, M * 4 + 4); hipMalloc(&ns, 16); hipMalloc(&err, 4);
    hipMalloc(&stamp, N * 4); hipMalloc(&cnt, M * 4 + 4); hipMalloc(&f0, M * 4); hipMalloc(&aux, M * 4); hipMalloc(&mm, 8); hipMalloc(&rec, M * 16); hipMalloc(&info, M * 16);
    hipMemset(stamp, 0, N * 4); hipMemset(f0, 0, M * 4); hipMemset(aux, 0, M * 4);
    hipMemcpy(idx, h.data(), M * 8, hipMemcpyHostToDevice);
    hipMemcpy(idx2, hi.data(), M * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 6; ++rep) {
        fr::SortJob j{};
        j.idx = (const int64_t*)idx; j.n_rows = N; j.perm = perm; j.seg_start = ss; j.seg_row = sr; j.seg_of = rep < 3 ? so : nullptr; j.n_seg = ns;
        if (rep >= 3) {
            j.seg_first = so; j.info = (int2*)info; j.info_stride = 2; j.cnt = cnt; j.stamp = stamp; j.stamp_val = rep;
            j.rec = rec; j.rec_idx = (const int64_t*)idx2; j.rec_rows = NI; j.rec_f0 = f0; j.aux = aux; j.aux_minmax = mm;
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, nullptr);
        fr::launch_sort(j, nullptr, M, err, nullptr);
        hipEventRecord(e1, nullptr);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(fr::g_sort_stamps), sizeof(st));
        printf("rep %d (%s, %.1f us): load %llu | ", rep, rep < 3 ? "plain" : "step", ms * 1e3, st[1] - st[0]);
        for (int p = 0; p < 3; ++p) printf("pass%d rank %llu scan %llu scatter %llu | ", p, st[2 + 3 * p] - (p ? st[4 + 3 * (p - 1)] : st[1]), st[3 + 3 * p] - st[2 + 3 * p], st[4 + 3 * p] - st[3 + 3 * p]);
        printf("heads %llu scan %llu out %llu total %llu ticks\n", st[12] - st[10], st[13] - st[12], st[14] - st[13], st[14] - st[0]);
    }
}
