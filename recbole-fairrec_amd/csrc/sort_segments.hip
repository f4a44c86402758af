// Batch-index sort + segmentation: one 1024-thread workgroup per index list, keys held in LDS.
//
// Replaces torch.unique(return_inverse=True) (focf.py:77-78, nfcf.py:79-80) and the implicit
// duplicate-row accumulation of embedding_dense_backward: the backward/Adam kernels consume the
// segments so that duplicate rows are reduced in ascending batch order (the order the reference's
// CPU index_put_/embedding backward accumulates in) -- deterministic, no atomics.
//
// Keys are (row id << 32 | batch position); the sort is a stable LSD radix sort on the row-id bits.
#include "common.hpp"
#include "kernels.hpp"

namespace fr {

}  // namespace fr
#include "sort_body.hpp"
namespace fr {

template <int KPT>
__global__ __launch_bounds__(SORT_THREADS) void sort_segments_kernel(SortJobList jobs, int npass, uint32_t* err) {
    extern __shared__ __align__(16) unsigned char smem[];
    sort_segments_body<KPT>(jobs, npass, err, blockIdx.x, smem);
}

template <int KPT>
static int launch_sort_kpt(const SortJobList& jobs, int bits, uint32_t* err, hipStream_t stream) {
    const int passes = (bits + sort_digit_bits(KPT) - 1) / sort_digit_bits(KPT);
    const int npass = passes | ((bits - sort_digit_bits(KPT) * (passes - 1)) << 8);      // (+ the last pass's bit count: sort_body.hpp)
    const size_t lds = sort_lds_bytes<KPT>();
    static bool attr_set = false;
    if (!attr_set) {
        FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sort_segments_kernel<KPT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    int n_stamp = 0;   // workgroups that only stamp table rows (see the kernel)
    for (int q = 0; q < jobs.n; ++q)
        if (jobs.j[q].stamp) n_stamp += (jobs.M[q] + SORT_THREADS - 1) / SORT_THREADS;
    ProfScope prof(K_SORT, stream);
    FR_LAUNCH(prof, sort_segments_kernel<KPT>, dim3(jobs.n + n_stamp), dim3(SORT_THREADS), lds, stream, jobs, npass, err);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

int launch_sort_many(const SortJobList& jobs, int64_t n_rows_max, uint32_t* err, hipStream_t stream) {
    FR_CHECK_ARG(jobs.n >= 1 && jobs.n <= FR_SORT_JOBS, "sort: %d lists not in 1..%d", jobs.n, FR_SORT_JOBS);
    int M = 0;
    for (int q = 0; q < jobs.n; ++q) {
        FR_CHECK_ARG(jobs.M[q] >= 0 && jobs.M[q] <= FR_SORT_MAX, "sort: M=%d exceeds FR_SORT_MAX=%d", jobs.M[q], FR_SORT_MAX);
        if (jobs.M[q] > M) M = jobs.M[q];
    }
    int bits = 1;
    while (bits < 32 && (1ll << bits) < n_rows_max) ++bits;
    if (M <= 1 * SORT_THREADS) return launch_sort_kpt<1>(jobs, bits, err, stream);
    if (M <= 2 * SORT_THREADS) return launch_sort_kpt<2>(jobs, bits, err, stream);
    if (M <= 4 * SORT_THREADS) return launch_sort_kpt<4>(jobs, bits, err, stream);
    if (M <= 8 * SORT_THREADS) return launch_sort_kpt<8>(jobs, bits, err, stream);
    return launch_sort_kpt<16>(jobs, bits, err, stream);
}

int launch_sort(const SortJob& a, const SortJob* b, int64_t M, uint32_t* err, hipStream_t stream) {
    FR_CHECK_ARG(M >= 0 && M <= FR_SORT_MAX, "sort: M=%lld exceeds FR_SORT_MAX=%d", (long long)M, FR_SORT_MAX);
    SortJobList jobs{};
    jobs.j[0] = a;
    jobs.M[0] = (int)M;
    jobs.n = 1;
    long long nmax = a.n_rows;
    if (b) {
        jobs.j[1] = *b;
        jobs.M[1] = (int)M;
        jobs.n = 2;
        if (b->n_rows > nmax) nmax = b->n_rows;
    }
    return launch_sort_many(jobs, nmax, err, stream);
}

}  // namespace fr

#ifdef FR_SORT_STAMPS
extern "C" __attribute__((visibility("default"))) int fr_debug_sort_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fr::g_sort_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int fr_sort_segments(const int64_t* idx, int64_t M, int64_t n_rows, int32_t* perm, int32_t* seg_start,
                                int32_t* seg_row, int32_t* seg_of, int32_t* n_seg, uint32_t* err_flag, void* stream) {
    FR_CHECK_ARG(idx && perm && seg_start && seg_row && n_seg, "fr_sort_segments: null pointer");
    fr::SortJob j{idx, n_rows, perm, seg_start, seg_row, seg_of, n_seg, nullptr, nullptr};
    return fr::launch_sort(j, nullptr, M, err_flag, (hipStream_t)stream);
}
