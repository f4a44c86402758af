// Lazy-Adam embedding tables: maintenance (flush / read-only gather / dense Adam) and the generic
// training pair  gather_train -> (model-specific forward/backward elsewhere) -> apply_grad  used by the
// models whose score is not a plain dot product and by the row-sharded multi-GPU path.
//
// Replaces: nn.Embedding forward (nfcf.py:70-71, pfcn_biasedmf.py:145-148), embedding_dense_backward and the
// per-tensor torch.optim.Adam state/step (trainer.py:139,196).
#include "common.hpp"
#include "kernels.hpp"
#include "table.hpp"

namespace fr {

// ------------------------------------------------------------------------------------------------
// table maintenance
// ------------------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void table_flush_kernel(TableV T, AdamC c) {
    const int lane = threadIdx.x & 63;
    const long long nw = (long long)gridDim.x * 4;
    for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < T.n_rows; row += nw)
        sweep_row<E>(T, c, row, T.step, false, lane);
}

template <int E>
__global__ __launch_bounds__(256) void table_gather_kernel(TableV T, AdamC c, const int64_t* __restrict__ idx,
                                                           long long M, float* __restrict__ out, uint32_t* err) {
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= M) return;
    long long r = idx[j];
    if (r < 0 || r >= T.n_rows) {
        if (lane == 0 && err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
        r = 0;
    }
    const int row = uniform((int)r);
    const int D = T.D;
    const int t0 = uniform(T.last[row]);
    RowFrag<E> p, m, v;
    load_row<E>(p, T.p + (size_t)row * D, D, lane);
    if (t0 < T.step) {
        load_row<E>(m, T.m + (size_t)row * D, D, lane);
        load_row<E>(v, T.v + (size_t)row * D, D, lane);
        replay<E>(p, m, v, t0, T.step, c, lane);
    }
    store_row<E>(p, out + (size_t)j * D, D, lane);
}

__global__ __launch_bounds__(256) void adam_dense_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, long long n,
                                                         AdamC c, int step) {
    const float2 s = step_scalars(c, step);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_elem(pp, mm, vv, g[i], s.x, s.y, c);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}



// ------------------------------------------------------------------------------------------------
// generic training pair
// ------------------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void table_gather_train_kernel(TableV T, AdamC c, const int64_t* __restrict__ idx,
                                                                 long long M, float* __restrict__ rows_out, TableWs w,
                                                                 uint32_t* err) {
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= M) return;
    long long r = idx[j];
    const int D = T.D;
    if (r == -1) {   // padding slot of a fixed-capacity exchange buffer: no row, zero output
        RowFrag<E> z;
#pragma unroll
        for (int e = 0; e < E; ++e) z.x[e] = 0.f;
        store_row<E>(z, rows_out + (size_t)j * D, D, lane);
        return;
    }
    if (r < 0 || r >= T.n_rows) {
        if (lane == 0 && err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
        r = 0;
    }
    const int row = uniform((int)r);
    const int t0 = uniform(T.last[row]);
    RowFrag<E> p, m, v;
    load_row<E>(p, T.p + (size_t)row * D, D, lane);
    load_row<E>(m, T.m + (size_t)row * D, D, lane);
    load_row<E>(v, T.v + (size_t)row * D, D, lane);
    replay<E>(p, m, v, t0, T.step - 1, c, lane);
    store_row<E>(p, rows_out + (size_t)j * D, D, lane);
    store_row<E>(m, w.m_side + (size_t)j * D, D, lane);
    store_row<E>(v, w.v_side + (size_t)j * D, D, lane);
    if (lane == 0) T.stamp[row] = T.step;
}

template <int E>
__global__ __launch_bounds__(256) void table_apply_grad_kernel(TableV T, AdamC c, long long M, TableWs w,
                                                               const float* __restrict__ rows,
                                                               const float* __restrict__ grad_rows, long long sw_lo,
                                                               int sw_n) {
    const int lane = threadIdx.x & 63;
    long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv < sw_n) {   // longest jobs first
        sweep_row<E>(T, c, sw_lo + wv, T.step, true, lane);
        return;
    }
    wv -= sw_n;
    if (wv < M && wv < w.nseg[0])
        segment_update<E>(T, c, (int)wv, w.seg_start, w.seg_row, w.perm, nullptr, rows, w.m_side, w.v_side, grad_rows,
                          lane);
}

}  // namespace fr

using namespace fr;

extern "C" int fr_table_flush(const fr_table* t, const fr_adam* adam, void* stream_) {
    int rc;
    if ((rc = check_table(t, "fr_table_flush")) || (rc = check_adam(adam, "fr_table_flush"))) return rc;
    if (t->step < 1) return FR_OK;
    const AdamC c = make_adamc(adam);
    const TableV Tv = view(t);
    long long blocks = (t->n_rows + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    {
        ProfScope prof(K_TABLE_FLUSH, (hipStream_t)stream_);
        FR_DISPATCH_E(t->dim, FR_LAUNCH(prof, (table_flush_kernel<E>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, Tv, c));
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_table_gather(const fr_table* t, const fr_adam* adam, const int64_t* idx, int64_t M, float* out,
                               uint32_t* err_flag, void* stream_) {
    int rc;
    if ((rc = check_table(t, "fr_table_gather")) || (rc = check_adam(adam, "fr_table_gather"))) return rc;
    FR_CHECK_ARG(idx && out && M >= 0, "fr_table_gather: bad argument");
    if (M == 0) return FR_OK;
    const AdamC c = make_adamc(adam);
    const TableV Tv = view(t);
    {
        ProfScope prof(K_TABLE_GATHER, (hipStream_t)stream_);
        FR_DISPATCH_E(t->dim, FR_LAUNCH(prof, (table_gather_kernel<E>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, Tv, c, idx, (long long)M, out, err_flag));
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_adam_dense(float* p, const float* g, float* m, float* v, int64_t n, const fr_adam* adam,
                             int32_t step, void* stream_) {
    int rc;
    if ((rc = check_adam(adam, "fr_adam_dense"))) return rc;
    FR_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "fr_adam_dense: bad argument");
    if (n == 0) return FR_OK;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    {
        ProfScope prof(K_ADAM_DENSE, (hipStream_t)stream_);
        FR_LAUNCH(prof, adam_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, p, g, m, v,
                           (long long)n, make_adamc(adam), step);
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" size_t fr_table_train_workspace_bytes(int64_t M, int32_t dim) {
    if (M < 0 || dim < 1) return 0;
    return table_layout(nullptr, M, dim).bytes;
}

extern "C" int fr_table_gather_train(const fr_table* t, const fr_adam* adam, const int64_t* idx, int64_t M,
                                     float* rows_out, void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = check_table(t, "fr_table_gather_train")) || (rc = check_adam(adam, "fr_table_gather_train"))) return rc;
    FR_CHECK_ARG(idx && rows_out && ws && M >= 1 && M <= FR_SORT_MAX && t->step >= 1,
                 "fr_table_gather_train: bad argument (M=%lld, step=%d)", (long long)M, t->step);
    TableWs w = table_layout(ws, M, t->dim);
    FR_CHECK_ARG(ws_bytes >= w.bytes, "fr_table_gather_train: workspace %zu < %zu bytes", ws_bytes, w.bytes);
    SortJob job{idx, t->n_rows, w.perm, w.seg_start, w.seg_row, nullptr, w.nseg, nullptr, nullptr};
    SideStream* ss = side_stream();
    const bool overlap = ss != nullptr && !prof_on();
    if (overlap) {
        FR_CHECK_HIP(hipEventRecord(ss->fork, stream));
        FR_CHECK_HIP(hipStreamWaitEvent(ss->stream, ss->fork, 0));
        if ((rc = launch_sort(job, nullptr, M, err_flag, ss->stream))) return rc;
        FR_CHECK_HIP(hipEventRecord(ss->join, ss->stream));
    } else if ((rc = launch_sort(job, nullptr, M, err_flag, stream))) {
        return rc;
    }
    const AdamC c = make_adamc(adam);
    const TableV Tv = view(t);
    {
        ProfScope prof(K_TABLE_GATHER_TRAIN, stream);
        FR_DISPATCH_E(t->dim, FR_LAUNCH(prof, (table_gather_train_kernel<E>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, Tv, c, idx, (long long)M, rows_out, w, err_flag));
    }
    FR_CHECK_LAUNCH();
    if (overlap) FR_CHECK_HIP(hipStreamWaitEvent(stream, ss->join, 0));
    return FR_OK;
}

extern "C" int fr_table_apply_grad(const fr_table* t, const fr_adam* adam, int64_t M, const float* rows,
                                   const float* grad_rows, int32_t sweep_period, void* ws, size_t ws_bytes,
                                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = check_table(t, "fr_table_apply_grad")) || (rc = check_adam(adam, "fr_table_apply_grad"))) return rc;
    FR_CHECK_ARG(rows && grad_rows && ws && M >= 1 && M <= FR_SORT_MAX && t->step >= 1,
                 "fr_table_apply_grad: bad argument");
    TableWs w = table_layout(ws, M, t->dim);
    FR_CHECK_ARG(ws_bytes >= w.bytes, "fr_table_apply_grad: workspace %zu < %zu bytes", ws_bytes, w.bytes);
    long long lo, hi;
    sweep_range(t->n_rows, t->step, sweep_period, lo, hi);
    const long long waves = M + (hi - lo);
    const AdamC c = make_adamc(adam);
    const TableV Tv = view(t);
    {
        ProfScope prof(K_TABLE_APPLY_GRAD, stream);
        FR_DISPATCH_E(t->dim, FR_LAUNCH(prof, (table_apply_grad_kernel<E>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, Tv, c, (long long)M, w, rows, grad_rows, lo, (int)(hi - lo)));
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}
