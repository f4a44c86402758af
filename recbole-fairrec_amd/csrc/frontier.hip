// The row sets of a frontier-restricted graph propagation (fairrec/model/fair_recommender/fairgo_pmf.py::_frontier): which rows of
// H_l = L H_(l-1) a batch's local embeddings depend on (fairgo_pmf.py:196-216 aggregates every layer's rows of the batch's users).
// Our own index work, not the reference's (it propagates whole tables): built with torch it was ~15 launches over 11 M-entry maps
// and 5.7 M neighbour ids per filter step at BASELINE configs[3] (5 ms of a 36 ms step).  Here a set is a BITMAP over the graph rows:
//   frontier_mark     bits |= {ids}                          (the batch's users)
//   frontier_expand   bits |= columns of the rows in a list  (one wave per listed row, its nonzeros lane-strided)
//   frontier_count    count[w] = popcount(bits[w])           (then one cumulative sum, torch's)
//   frontier_scatter  rows_out = the set's row ids ascending, pos[row] = its rank (-1 for rows outside): what fr_spmm_csr_sel reads
#include "common.hpp"
#include "kernels.hpp"

namespace fr {

__global__ __launch_bounds__(256) void frontier_mark_kernel(const int64_t* __restrict__ ids, long long n, long long n_rows,
                                                            unsigned* __restrict__ bits, unsigned* err) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long r = ids[i];
    if (r < 0 || r >= n_rows) {
        if (err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
        return;
    }
    atomicOr(bits + (r >> 5), 1u << (r & 31));
}

__global__ __launch_bounds__(256) void frontier_expand_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ col,
                                                              const int32_t* __restrict__ rows, long long n_list,
                                                              unsigned* __restrict__ bits) {
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_list) return;
    const int lane = threadIdx.x & 63;
    const int r = rows[w];
    const long long lo = indptr[r], hi = indptr[r + 1];
    for (long long j = lo + lane; j < hi; j += 64) {
        const int c = col[j];
        atomicOr(bits + (c >> 5), 1u << (c & 31));
    }
}

__global__ __launch_bounds__(256) void frontier_count_kernel(const unsigned* __restrict__ bits, long long n_words,
                                                             int32_t* __restrict__ count) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n_words) count[i] = __popc(bits[i]);
}

// incl[w] = number of set bits in words 0 .. w (the cumulative sum of frontier_count's output)
__global__ __launch_bounds__(256) void frontier_scatter_kernel(const unsigned* __restrict__ bits, const int32_t* __restrict__ incl,
                                                               long long n_words, long long n_rows, int32_t* __restrict__ rows_out,
                                                               int32_t* __restrict__ pos) {
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    const unsigned b = bits[w];
    int rank = incl[w] - __popc(b);
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
        const long long row = w * 32 + k;
        if (row >= n_rows) break;
        if (b >> k & 1u) {
            rows_out[rank] = (int32_t)row;
            pos[row] = rank++;
        } else {
            pos[row] = -1;
        }
    }
}

}  // namespace fr

using namespace fr;

extern "C" int fr_frontier_mark(const int64_t* ids, int64_t n, int64_t n_rows, uint32_t* bits, uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(ids && bits && n >= 0 && n_rows >= 1, "fr_frontier_mark: bad argument");
    if (n == 0) return FR_OK;
    hipLaunchKernelGGL(frontier_mark_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, ids, (long long)n,
                       (long long)n_rows, bits, err_flag);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_frontier_expand(const int64_t* indptr, const int32_t* col, const int32_t* rows, int64_t n_list, uint32_t* bits,
                                  void* stream_) {
    FR_CHECK_ARG(indptr && col && rows && bits && n_list >= 0, "fr_frontier_expand: bad argument");
    if (n_list == 0) return FR_OK;
    hipLaunchKernelGGL(frontier_expand_kernel, dim3((unsigned)((n_list + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, indptr, col, rows,
                       (long long)n_list, bits);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_frontier_count(const uint32_t* bits, int64_t n_rows, int32_t* count, void* stream_) {
    FR_CHECK_ARG(bits && count && n_rows >= 1, "fr_frontier_count: bad argument");
    const long long nw = (n_rows + 31) / 32;
    hipLaunchKernelGGL(frontier_count_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, bits, nw, count);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_frontier_scatter(const uint32_t* bits, const int32_t* incl, int64_t n_rows, int32_t* rows_out, int32_t* pos,
                                   void* stream_) {
    FR_CHECK_ARG(bits && incl && rows_out && pos && n_rows >= 1, "fr_frontier_scatter: bad argument");
    const long long nw = (n_rows + 31) / 32;
    hipLaunchKernelGGL(frontier_scatter_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, bits, incl, nw,
                       (long long)n_rows, rows_out, pos);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
