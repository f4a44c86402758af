// FairGo graph kernels: CSR SpMM over the row-normalised rating graph, row gather / duplicate-summed row scatter on
// whole-table activations, MSE head.
//
// Replaces fairgo_pmf.py:196-200 (torch.sparse.mm(L, E), L = D^-1 A of get_norm_rating_matrix :102-129), the
// `all_embeddings[user]` indexing (:178-179, :194) with its index_put backward, and nn.MSELoss (:182).
#include "common.hpp"
#include "kernels.hpp"

namespace fr {

// Y[r,:] = sum_{j in row r} val[j] * X[col[j],:]      one wave per row, fixed (ascending j) order.
// V > 0: D == 64 * V and lane l owns columns [l*V, l*V+V) (one 64*V*4-byte gather per nonzero); the row's (col, val)
// pairs are loaded 64 at a time, one per lane, and broadcast with readlane so that the gathers of consecutive
// nonzeros are independent loads in flight.  V == 0: any D, strided columns.
template <int V>
__global__ __launch_bounds__(256) void spmm_csr_kernel(const long long* __restrict__ indptr, const int* __restrict__ col,
                                                       const float* __restrict__ val, const float* __restrict__ X,
                                                       long long n_rows, int D, float* __restrict__ Y) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const long long j0 = indptr[r], j1 = indptr[r + 1];
    if constexpr (V == 0) {
        for (int d = lane; d < D; d += 64) {
            float acc = 0.f;
            for (long long j = j0; j < j1; ++j) acc = fmaf(val[j], X[(size_t)col[j] * D + d], acc);
            Y[(size_t)r * D + d] = acc;
        }
    } else {
        typedef float vec __attribute__((ext_vector_type(V)));
        vec acc = {};
        for (long long jb = j0; jb < j1; jb += 64) {
            const int cnt = (int)min((long long)64, j1 - jb);
            int my_c = 0;
            float my_v = 0.f;
            if (lane < cnt) {
                my_c = col[jb + lane];
                my_v = val[jb + lane];
            }
#pragma unroll 4
            for (int t = 0; t < cnt; ++t) {
                const int c = __builtin_amdgcn_readlane(my_c, t);
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_v), t));
                const vec x = *reinterpret_cast<const vec*>(X + (size_t)c * D + lane * V);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] = fmaf(v, x[e], acc[e]);
            }
        }
        *reinterpret_cast<vec*>(Y + (size_t)r * D + lane * V) = acc;
    }
}

// out[j,:] = X[idx[j],:]
__global__ __launch_bounds__(256) void row_gather_kernel(const float* __restrict__ X, const long long* __restrict__ idx,
                                                         long long M, long long n_rows, int D, float* __restrict__ out,
                                                         uint32_t* err) {
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= M) return;
    long long r = idx[j];
    if (r < 0 || r >= n_rows) {
        if (lane == 0 && err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
        r = 0;
    }
    for (int d = lane; d < D; d += 64) out[(size_t)j * D + d] = X[(size_t)r * D + d];
}

// dX[seg_row[k],:] = sum over the segment's members (ascending position) of g[member,:]; dX is pre-zeroed
__global__ __launch_bounds__(256) void row_scatter_sum_kernel(const float* __restrict__ g, const int32_t* __restrict__ perm,
                                                              const int32_t* __restrict__ seg_start,
                                                              const int32_t* __restrict__ seg_row,
                                                              const int32_t* __restrict__ nseg, int M, int D,
                                                              float* __restrict__ dX) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= M || k >= nseg[0]) return;
    const int j0 = seg_start[k], j1 = seg_start[k + 1];
    const size_t row = (size_t)seg_row[k];
    for (int d = lane; d < D; d += 64) {
        float acc = 0.f;
        for (int j = j0; j < j1; ++j) acc += g[(size_t)perm[j] * D + d];
        dX[row * D + d] = acc;
    }
}

// nn.MSELoss: partial sums of (pred - target)^2 and d/dpred = 2 (pred - target) / B
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ target, int B,
                                                  float* __restrict__ dpred, float* __restrict__ part) {
    __shared__ float red[4];
    const int b = blockIdx.x * 256 + threadIdx.x;
    float l = 0.f;
    if (b < B) {
        const float e = pred[b] - target[b];
        l = e * e;
        dpred[b] = 2.f * e / (float)B;
    }
    l = wave_sum(l);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(256) void mse_finalize_kernel(const float* __restrict__ part, int n, int B,
                                                           float* __restrict__ out) {
    __shared__ float red[4];
    float a = 0.f;
    for (int q = threadIdx.x; q < n; q += 256) a += part[q];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (((red[0] + red[1]) + red[2]) + red[3]) / (float)B;
}

}  // namespace fr

using namespace fr;

extern "C" int fr_spmm_csr(const int64_t* indptr, const int32_t* col, const float* val, const float* X, int64_t n_rows,
                           int32_t dim, float* Y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(indptr && col && val && X && Y && n_rows >= 1 && dim >= 1, "fr_spmm_csr: bad argument");
    ProfScope prof(K_SPMM, stream);
    if (prof_on()) {      // bytes of SURVEY.md §8-d: (col, val) per nonzero + X read and Y written once per row; nnz from the host
        static thread_local long long nnz_cache = -1;     // copy of indptr[n_rows] (one 8-byte read per call while profiling)
        long long nnz = 0;
        if (hipMemcpyAsync(&nnz, indptr + n_rows, 8, hipMemcpyDeviceToHost, stream) == hipSuccess &&
            hipStreamSynchronize(stream) == hipSuccess)
            nnz_cache = nnz;
        prof_work(K_SPMM, 12.0 * (double)nnz_cache + 8.0 * (double)n_rows * dim);
    }
    const dim3 grid((unsigned)((n_rows + 3) / 4));
#define FR_SPMM(V)                                                                                              \
    FR_LAUNCH(prof, spmm_csr_kernel<V>, grid, dim3(256), 0, stream, (const long long*)indptr, col, val, X, \
              (long long)n_rows, (int)dim, Y)
    switch (dim) {
        case 64: FR_SPMM(1); break;
        case 128: FR_SPMM(2); break;
        case 256: FR_SPMM(4); break;
        default: FR_SPMM(0); break;
    }
#undef FR_SPMM
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_row_gather(const float* X, const int64_t* idx, int64_t M, int64_t n_rows, int32_t dim, float* out,
                             uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(X && idx && out && M >= 1 && n_rows >= 1 && dim >= 1, "fr_row_gather: bad argument");
    ProfScope prof(K_ROW_GATHER, stream);
    FR_LAUNCH(prof, row_gather_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, X, (const long long*)idx,
              (long long)M, (long long)n_rows, (int)dim, out, err_flag);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" size_t fr_row_scatter_workspace_bytes(int64_t M) {
    return M < 1 ? 0 : align_up(((size_t)M + 1) * 4, 256) * 3 + 256;
}

// dX [n_rows, dim] (zeroed here) += rows of g [M, dim] at idx, duplicates summed in ascending position
extern "C" int fr_row_scatter_sum(const float* g, const int64_t* idx, int64_t M, int64_t n_rows, int32_t dim, float* dX,
                                  void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(g && idx && dX && ws && M >= 1 && M <= FR_SORT_MAX && dim >= 1 && ws_bytes >= fr_row_scatter_workspace_bytes(M),
                 "fr_row_scatter_sum: bad argument");
    char* p = (char*)ws;
    const size_t stride = align_up(((size_t)M + 1) * 4, 256);
    int32_t* perm = (int32_t*)p;
    int32_t* seg_start = (int32_t*)(p + stride);
    int32_t* seg_row = (int32_t*)(p + 2 * stride);
    int32_t* nseg = (int32_t*)(p + 3 * stride);
    SortJob job{idx, n_rows, perm, seg_start, seg_row, nullptr, nseg, nullptr, nullptr};
    int rc = launch_sort(job, nullptr, M, err_flag, stream);
    if (rc) return rc;
    FR_CHECK_HIP(hipMemsetAsync(dX, 0, (size_t)n_rows * dim * sizeof(float), stream));
    ProfScope prof(K_ROW_GATHER, stream);
    FR_LAUNCH(prof, row_scatter_sum_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, g, (const int32_t*)perm,
              (const int32_t*)seg_start, (const int32_t*)seg_row, (const int32_t*)nseg, (int)M, (int)dim, dX);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_mse(const float* pred, const float* target, int64_t B, float* loss, float* dpred, void* ws,
                      size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int nb = (int)((B + 255) / 256);
    FR_CHECK_ARG(pred && target && loss && dpred && ws && B >= 1 && ws_bytes >= (size_t)nb * 4, "fr_mse: bad argument");
    hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, stream, pred, target, (int)B, dpred, (float*)ws);
    FR_CHECK_LAUNCH();
    hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, nb, (int)B, loss);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
