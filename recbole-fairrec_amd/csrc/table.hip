// Lazy-Adam embedding tables: maintenance (flush / read-only gather / dense Adam) and the generic
// training pair  gather_train -> (model-specific forward/backward elsewhere) -> apply_grad  used by the
// models whose score is not a plain dot product and by the row-sharded multi-GPU path.
//
// Replaces: nn.Embedding forward (nfcf.py:70-71, pfcn_biasedmf.py:145-148), embedding_dense_backward and the
// per-tensor torch.optim.Adam state/step (trainer.py:139,196).
#include <algorithm>

#include "common.hpp"
#include "kernels.hpp"
#include "table.hpp"
#include "sort_body.hpp"

namespace fr {

// ------------------------------------------------------------------------------------------------
// table maintenance
// ------------------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void table_flush_kernel(TableV T_, AdamC c) {
    const TableV T = resolved(T_);
    const int lane = threadIdx.x & 63;
    const long long nw = (long long)gridDim.x * 4;
    if (E == 1 && T.D == 1) {      // narrow table: 64 rows per wave (table.hpp)
        for (long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); w * 64 < T.n_rows; w += nw)
            sweep_rows_narrow(T, c, w * 64, T.n_rows, T.step, 0x7fffffff, lane);
        return;
    }
    for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < T.n_rows; row += nw)
        sweep_row<E>(T, c, row, T.step, 0x7fffffff, lane);
}

template <int E>
__device__ __forceinline__ void gather_row(const TableV& T_, const AdamC& c, const int64_t* __restrict__ idx, long long M,
                                           float* __restrict__ out, uint32_t* err, const long long j, const int lane) {
    const TableV T = resolved(T_);
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

template <int E>
__global__ __launch_bounds__(256) void table_gather_kernel(TableV T_, AdamC c, const int64_t* __restrict__ idx,
                                                           long long M, float* __restrict__ out, uint32_t* err) {
    gather_row<E>(T_, c, idx, M, out, err, (long long)blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
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



// every dense parameter tensor of an optimizer group in ONE launch (blockIdx.y = tensor): the descriptors travel as
// kernel arguments, so there is no pointer table to upload
struct DenseBatch {
    fr_dense_desc t[FR_ADAM_DENSE_MAX];
};

__global__ __launch_bounds__(256) void adam_dense_multi_kernel(DenseBatch b, AdamC c) {
    const fr_dense_desc& d = b.t[blockIdx.y];
    const float2 s = step_scalars(c, d.step + (d.step_dev ? *d.step_dev : 0));
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (long long)gridDim.x * 256) {
        float pp = d.p[i], mm = d.m[i], vv = d.v[i];
        adam_elem(pp, mm, vv, d.g[i], s.x, s.y, c);
        d.p[i] = pp; d.m[i] = mm; d.v[i] = vv;
    }
}

// ------------------------------------------------------------------------------------------------
// generic training pair (one or two tables per launch: blockIdx.y selects the table)
// ------------------------------------------------------------------------------------------------
struct GatherJob {
    TableV T;
    const int64_t* idx;
    float* rows_out;
    TableWs w;
};

template <int E>
__device__ __forceinline__ void gather_train_row(const GatherJob& J, const AdamC& c, long long M, const Lay& lay, uint32_t* err,
                                                 const long long j, const int lane) {
    const TableV T = resolved(J.T);
    if (j >= M) return;
    const long long jp = lay.at(j);          // where logical position j sits in idx / rows_out
    long long r = J.idx[jp];
    const int D = T.D;
    if (r == -1) {   // padding slot of a fixed-capacity exchange buffer: no row, zero output
        RowFrag<E> z;
#pragma unroll
        for (int e = 0; e < E; ++e) z.x[e] = 0.f;
        store_row<E>(z, J.rows_out + (size_t)jp * D, D, lane);
        return;
    }
    if (r < 0 || r >= T.n_rows) {
        if (lane == 0 && err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
        r = 0;
    }
    const int row = uniform((int)r);
    const int lt = T.last[row];        // requested together with the row: one dependent round trip, not two
    RowFrag<E> p, m, v;
    load_row<E>(p, T.p + (size_t)row * D, D, lane);
    load_row<E>(m, T.m + (size_t)row * D, D, lane);
    load_row<E>(v, T.v + (size_t)row * D, D, lane);
    const int t0 = uniform(lt);
    replay<E>(p, m, v, t0, T.step - 1, c, lane);
    store_row<E>(p, J.rows_out + (size_t)jp * D, D, lane);
    store_row<E>(m, J.w.m_side + (size_t)j * D, D, lane);
    store_row<E>(v, J.w.v_side + (size_t)j * D, D, lane);
    if (lane == 0) T.stamp[row] = T.step;
}

template <int E>
__global__ __launch_bounds__(256) void table_gather_train_kernel(GatherJob ja, GatherJob jb, AdamC c, long long M,
                                                                 Lay lay, uint32_t* err) {
    gather_train_row<E>(blockIdx.y == 0 ? ja : jb, c, M, lay, err, (long long)blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// The lookups of a step in ONE launch: the sort of the id list(s) as the first workgroup(s) (sort_body.hpp: one latency-bound
// 1024-thread workgroup per list), the training gathers of one or two tables and, optionally, the read-only gather of a frozen
// table (NFCF finetune: nfcf.py:66) behind them, one row per wave.  Inside a captured step (hipGraph) nothing overlaps across
// launches -- the sort in front of the gathers was 27 of NFCF's 171 us; here it runs beside them.
struct PlainJob {
    TableV T;
    const int64_t* idx;
    float* out;
    long long M;
};
template <int E, int KPT>
__global__ __launch_bounds__(SORT_THREADS) void table_lookup_kernel(SortJobList jobs, int npass, GatherJob ja, GatherJob jb,
                                                                    int n_train, PlainJob pj, AdamC c, AdamC cp, long long M,
                                                                    Lay lay, uint32_t* err) {
    extern __shared__ __align__(16) unsigned char smem[];
    if ((int)blockIdx.x < jobs.n) {
        sort_segments_body<KPT>(jobs, npass, err, blockIdx.x, smem);
        return;
    }
    constexpr int WPB = SORT_THREADS / 64;
    const int lane = threadIdx.x & 63;
    const long long per = (M + WPB - 1) / WPB;                  // workgroups per training table
    long long b = (long long)blockIdx.x - jobs.n;
    if (b < per * n_train) {
        const bool second = b >= per;
        gather_train_row<E>(second ? jb : ja, c, M, lay, err, (second ? b - per : b) * WPB + (threadIdx.x >> 6), lane);
        return;
    }
    b -= per * n_train;
    gather_row<E>(pj.T, cp, pj.idx, pj.M, pj.out, err, b * WPB + (threadIdx.x >> 6), lane);
}

struct ApplyJob {
    TableV T;
    TableWs w;
    const float* rows;
    const float* grad_rows;
    int sweep_period;     // the sweeper slice is derived from the effective step on the device
    int sw_n;             // waves reserved for it (= pairs of rows of the longest slice)
    long long M;          // ids of this table's batch (fr_table_apply_grad_two: each table its own)
};

template <int E>
__global__ __launch_bounds__(256) void table_apply_grad_kernel(ApplyJob ja, ApplyJob jb, AdamC c, long long M, Lay lay) {
    const ApplyJob& J = blockIdx.y == 0 ? ja : jb;
    const TableV T = resolved(J.T);
    const int lane = threadIdx.x & 63;
    long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv < J.sw_n) {   // longest jobs first
        long long lo, hi;
        sweep_range(T.n_rows, T.step, J.sweep_period, lo, hi);
        if (E == 1 && T.D == 1) {     // narrow table: 64 adjacent rows of the slice, one per lane
            const long long a = lo + 64 * wv;
            if (a < hi) sweep_rows_narrow(T, c, a, hi, T.step, T.step, lane);
        } else if (sweep_pairs(E)) {     // a wave takes two adjacent rows of the slice (see sweep_row_pair)
            const long long a = lo + 2 * wv;
            if (a < hi) sweep_row_pair<E>(T, c, a, a + 1 < hi ? a + 1 : -1, T.step, T.step, lane);
        } else if (lo + wv < hi) {
            sweep_row<E>(T, c, lo + wv, T.step, T.step, lane);
        }
        return;
    }
    wv -= J.sw_n;
    if (wv < J.M && wv < J.w.nseg[0])
        segment_update<E>(T, c, (int)wv, J.w.seg_start, J.w.seg_row, J.w.perm, nullptr, J.rows, J.w.m_side, J.w.v_side,
                          J.grad_rows, lane, lay, J.w.seg_first);
}

}  // namespace fr

using namespace fr;

extern "C" int fr_table_flush(const fr_table* t, const fr_adam* adam, void* stream_) {
    int rc;
    if ((rc = check_table(t, "fr_table_flush")) || (rc = check_adam(adam, "fr_table_flush"))) return rc;
    if (t->step < 1 && !t->step_dev) return FR_OK;
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

extern "C" int fr_adam_dense_multi(const fr_dense_desc* descs, int32_t n_tensors, const fr_adam* adam, void* stream_) {
    int rc;
    if ((rc = check_adam(adam, "fr_adam_dense_multi"))) return rc;
    FR_CHECK_ARG(descs && n_tensors >= 0, "fr_adam_dense_multi: bad argument");
    const AdamC c = make_adamc(adam);
    for (int32_t base = 0; base < n_tensors; base += FR_ADAM_DENSE_MAX) {
        const int cnt = std::min<int>(FR_ADAM_DENSE_MAX, n_tensors - base);
        DenseBatch b;
        long long nmax = 1;
        for (int k = 0; k < cnt; ++k) {
            const fr_dense_desc& d = descs[base + k];
            FR_CHECK_ARG(d.p && d.g && d.m && d.v && d.n >= 0 && (d.step >= 1 || d.step_dev), "fr_adam_dense_multi: bad descriptor %d", base + k);
            b.t[k] = d;
            nmax = std::max<long long>(nmax, d.n);
        }
        for (int k = cnt; k < FR_ADAM_DENSE_MAX; ++k) b.t[k] = fr_dense_desc{nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr};
        const long long blocks = std::min<long long>((nmax + 255) / 256, 256);
        ProfScope prof(K_ADAM_DENSE, (hipStream_t)stream_);
        FR_LAUNCH(prof, adam_dense_multi_kernel, dim3((unsigned)blocks, (unsigned)cnt), dim3(256), 0, (hipStream_t)stream_, b,
                  c);
        FR_CHECK_LAUNCH();
    }
    return FR_OK;
}

extern "C" size_t fr_table_train_workspace_bytes(int64_t M, int32_t dim) {
    if (M < 0 || dim < 1) return 0;
    return table_layout(nullptr, M, dim).bytes;
}

static inline bool lay_ok(int32_t chunk, int32_t stride) { return chunk == 0 || (chunk > 0 && stride >= chunk); }

// one (tb == nullptr) or two tables of the same dim and M per call: one sort launch (a workgroup per table) on the
// side stream, one gather launch
static int gather_train_impl(const char* who, const fr_table* ta, const fr_table* tb, const fr_adam* adam,
                             const int64_t* idx_a, const int64_t* idx_b, int64_t M, int32_t chunk, int32_t stride,
                             float* rows_a, float* rows_b, void* ws_a, void* ws_b, size_t ws_bytes, uint32_t* err_flag,
                             hipStream_t stream, bool prepared = false, const fr_table* ro = nullptr,
                             const fr_adam* ro_adam = nullptr, const int64_t* ro_idx = nullptr, int64_t ro_M = 0,
                             float* ro_out = nullptr) {
    int rc;
    if ((rc = check_table(ta, who)) || (tb && (rc = check_table(tb, who))) || (rc = check_adam(adam, who))) return rc;
    FR_CHECK_ARG(idx_a && rows_a && ws_a && M >= 1 && M <= FR_SORT_MAX && ta->step >= 1 && lay_ok(chunk, stride),
                 "%s: bad argument (M=%lld, step=%d)", who, (long long)M, ta->step);
    FR_CHECK_ARG(!tb || (idx_b && rows_b && ws_b && ws_b != ws_a && tb->dim == ta->dim && tb->step >= 1),
                 "%s: bad second table", who);
    const Lay lay{chunk, stride};
    TableWs wa = table_layout(ws_a, M, ta->dim);
    TableWs wb = tb ? table_layout(ws_b, M, tb->dim) : wa;
    FR_CHECK_ARG(ws_bytes >= wa.bytes, "%s: workspace %zu < %zu bytes", who, ws_bytes, wa.bytes);
    SortJob sa{idx_a, ta->n_rows, wa.perm, wa.seg_start, wa.seg_row, nullptr, wa.nseg, nullptr, nullptr, lay};
    SortJob sb{idx_b, tb ? tb->n_rows : 0, wb.perm, wb.seg_start, wb.seg_row, nullptr, wb.nseg, nullptr, nullptr, lay};
    sa.seg_first = wa.seg_first;
    sb.seg_first = wb.seg_first;
    SideStream* ss = side_stream();
    // The sort (one latency-bound workgroup per list) runs beside the gather on the side stream -- except inside a stream
    // capture: a hipGraph replays both branches on one hardware queue, every kernel dispatched after the sort retires behind it
    // and each fork / join costs ~11 us there; measured on captured steps, sorting in line is faster
    // (NFCF finetune 0.230 -> 0.221 ms, PFCN filter pass 0.94 -> 0.86 ms).
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) != hipSuccess) {
        (void)hipGetLastError();
        capturing = hipStreamCaptureStatusNone;
    }
    const bool overlap = ss != nullptr && !prof_on() && capturing == hipStreamCaptureStatusNone;
    // one launch for the sort and the gathers where nothing would overlap otherwise (and whenever a frozen table rides along)
    static const bool no_merge = getenv("FAIRREC_LOOKUP_SEPARATE") != nullptr;
    const int kpt = M <= 2 * SORT_THREADS ? 2 : (M <= 4 * SORT_THREADS ? 4 : (M <= 8 * SORT_THREADS ? 8 : 16));
    if (!prepared && !no_merge && kpt != 0 && (!overlap || ro) && (!ro || (ro->dim + 63) / 64 == (ta->dim + 63) / 64)) {
        if (ro && ((rc = check_table(ro, who)) || (rc = check_adam(ro_adam, who)))) return rc;
        FR_CHECK_ARG(!ro || (ro_idx && ro_out && ro_M >= 0), "%s: bad read-only lookup", who);
        SortJobList jobs{};
        jobs.j[0] = sa;
        jobs.M[0] = (int)M;
        jobs.n = 1;
        long long nmax = sa.n_rows;
        if (tb) {
            jobs.j[1] = sb;
            jobs.M[1] = (int)M;
            jobs.n = 2;
            nmax = std::max<long long>(nmax, sb.n_rows);
        }
        int bits = 1;
        while (bits < 32 && (1ll << bits) < nmax) ++bits;
        const AdamC c = make_adamc(adam);
        const AdamC cp = ro ? make_adamc(ro_adam) : c;
        GatherJob ja{view(ta), idx_a, rows_a, wa};
        GatherJob jb = tb ? GatherJob{view(tb), idx_b, rows_b, wb} : ja;
        PlainJob pj{ro ? view(ro) : view(ta), ro_idx, ro_out, ro ? (long long)ro_M : 0};
        constexpr int WPB = SORT_THREADS / 64;
        const long long blocks = jobs.n + (M + WPB - 1) / WPB * (tb ? 2 : 1) + (ro ? (ro_M + WPB - 1) / WPB : 0);
        ProfScope prof(K_TABLE_GATHER_TRAIN, stream);
#define FR_LOOKUP_LAUNCH(KPT)                                                                                              \
    {                                                                                                                      \
        static bool attr_set[5] = {false, false, false, false, false};                                                     \
        if (!attr_set[E]) {                                                                                                \
            FR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(table_lookup_kernel<E, KPT>),                   \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds_bytes<KPT>()));     \
            attr_set[E] = true;                                                                                            \
        }                                                                                                                  \
        const int passes_ = (bits + sort_digit_bits(KPT) - 1) / sort_digit_bits(KPT);                                      \
        const int npass = passes_ | ((bits - sort_digit_bits(KPT) * (passes_ - 1)) << 8);                                  \
        FR_LAUNCH(prof, (table_lookup_kernel<E, KPT>), dim3((unsigned)blocks), dim3(SORT_THREADS), sort_lds_bytes<KPT>(),  \
                  stream, jobs, npass, ja, jb, tb ? 2 : 1, pj, c, cp, (long long)M, lay, err_flag);                        \
    }
        if (kpt == 2) {
            FR_DISPATCH_E(ta->dim, FR_LOOKUP_LAUNCH(2));
        } else if (kpt == 4) {
            FR_DISPATCH_E(ta->dim, FR_LOOKUP_LAUNCH(4));
        } else if (kpt == 8) {
            FR_DISPATCH_E(ta->dim, FR_LOOKUP_LAUNCH(8));
        } else {
            FR_DISPATCH_E(ta->dim, FR_LOOKUP_LAUNCH(16));
        }
#undef FR_LOOKUP_LAUNCH
        FR_CHECK_LAUNCH();
        return FR_OK;
    }
    if (ro) {      // not mergeable: the read-only gather as its own launch
        if ((rc = fr_table_gather(ro, ro_adam, ro_idx, ro_M, ro_out, err_flag, stream))) return rc;
    }
    if (prepared) {
        // fr_table_sort2 already left the segments of these id lists in the workspaces (one step ahead)
    } else if (overlap) {
        FR_CHECK_HIP(hipEventRecord(ss->fork, stream));
        FR_CHECK_HIP(hipStreamWaitEvent(ss->stream, ss->fork, 0));
        if ((rc = launch_sort(sa, tb ? &sb : nullptr, M, err_flag, ss->stream))) return rc;
        // joined by the first consumer of the segments (apply_grad / shard_fair)
        if ((rc = side_mark(ws_a)) || (tb && (rc = side_mark(ws_b)))) return rc;
    } else if ((rc = launch_sort(sa, tb ? &sb : nullptr, M, err_flag, stream))) {
        return rc;
    }
    const AdamC c = make_adamc(adam);
    GatherJob ja{view(ta), idx_a, rows_a, wa};
    GatherJob jb = tb ? GatherJob{view(tb), idx_b, rows_b, wb} : ja;
    {
        ProfScope prof(K_TABLE_GATHER_TRAIN, stream);
        FR_DISPATCH_E(ta->dim, FR_LAUNCH(prof, (table_gather_train_kernel<E>), dim3((unsigned)((M + 3) / 4), tb ? 2 : 1), dim3(256), 0, stream, ja, jb, c, (long long)M, lay, err_flag));
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_table_gather_train(const fr_table* t, const fr_adam* adam, const int64_t* idx, int64_t M,
                                     int32_t chunk, int32_t stride, float* rows_out, void* ws, size_t ws_bytes,
                                     uint32_t* err_flag, void* stream_) {
    return gather_train_impl("fr_table_gather_train", t, nullptr, adam, idx, nullptr, M, chunk, stride, rows_out, nullptr,
                             ws, nullptr, ws_bytes, err_flag, (hipStream_t)stream_);
}

// fr_table_gather_train on `t` and fr_table_gather on a second, read-only table `ro` (its own id list and hyper-parameters)
// as one launch where the shapes allow (same fragment count per row, M <= 2048 or 4096 < M <= 8192), else as the two calls.
extern "C" int fr_table_lookup_pair(const fr_table* t, const fr_adam* adam, const int64_t* idx, int64_t M, float* rows_out,
                                    void* ws, size_t ws_bytes, const fr_table* ro, const fr_adam* ro_adam,
                                    const int64_t* ro_idx, int64_t ro_M, float* ro_out, uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(ro && ro_adam && ro_idx && ro_out && ro_M >= 1, "fr_table_lookup_pair: bad read-only lookup");
    return gather_train_impl("fr_table_lookup_pair", t, nullptr, adam, idx, nullptr, M, 0, 0, rows_out, nullptr, ws, nullptr,
                             ws_bytes, err_flag, (hipStream_t)stream_, false, ro, ro_adam, ro_idx, ro_M, ro_out);
}

// The same with the segments of `idx` already in ws (FR_TABLE_PREPARED), e.g. copied from the workspace of another table
// that was looked up with the SAME id list in this step (fr_table_segments_bytes leading bytes of a workspace hold them,
// whatever the table's width): a bias table next to its embedding table needs no second sort.
extern "C" size_t fr_table_segments_bytes(int64_t M) {
    if (M < 1) return 0;
    char* const base = reinterpret_cast<char*>(256);      // (table_layout hands out null pointers for a null base)
    TableWs w = table_layout(base, M, 1);
    return (size_t)((char*)w.m_side - base);
}

// Orders `stream` behind the index work (the sort of fr_table_gather_train, which runs on the library's side stream) still
// pending on workspace `ws`: for callers that read the segments themselves (copying them to a second table's workspace).
extern "C" int fr_table_join(const void* ws, void* stream_) {
    FR_CHECK_ARG(ws, "fr_table_join: null workspace");
    return side_join(ws, (hipStream_t)stream_);
}

extern "C" int fr_table_gather_train_prepared(const fr_table* t, const fr_adam* adam, const int64_t* idx, int64_t M,
                                              int32_t chunk, int32_t stride, float* rows_out, void* ws, size_t ws_bytes,
                                              uint32_t* err_flag, void* stream_) {
    return gather_train_impl("fr_table_gather_train_prepared", t, nullptr, adam, idx, nullptr, M, chunk, stride, rows_out,
                             nullptr, ws, nullptr, ws_bytes, err_flag, (hipStream_t)stream_, true);
}

extern "C" int fr_table_gather_train2(const fr_table* ta, const fr_table* tb, const fr_adam* adam, const int64_t* idx_a,
                                      const int64_t* idx_b, int64_t M, int32_t chunk, int32_t stride, float* rows_a,
                                      float* rows_b, int32_t flags, void* ws_a, void* ws_b, size_t ws_bytes,
                                      uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(tb, "fr_table_gather_train2: second table is null");
    return gather_train_impl("fr_table_gather_train2", ta, tb, adam, idx_a, idx_b, M, chunk, stride, rows_a, rows_b, ws_a,
                             ws_b, ws_bytes, err_flag, (hipStream_t)stream_, (flags & FR_TABLE_PREPARED) != 0);
}

// The index-only part of fr_table_gather_train2, callable one batch AHEAD on another stream (it depends on nothing
// but the id lists): sort + segmentation into the two workspaces.  The caller orders it against the users of the
// workspaces with events and passes FR_TABLE_PREPARED to fr_table_gather_train2.
extern "C" int fr_table_sort2(const int64_t* idx_a, const int64_t* idx_b, int64_t n_rows_a, int64_t n_rows_b, int64_t M,
                              int32_t chunk, int32_t stride, int32_t dim, void* ws_a, void* ws_b, size_t ws_bytes,
                              uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(idx_a && idx_b && ws_a && ws_b && ws_a != ws_b && M >= 1 && M <= FR_SORT_MAX && dim >= 1 && n_rows_a >= 1 &&
                     n_rows_b >= 1 && lay_ok(chunk, stride), "fr_table_sort2: bad argument");
    TableWs wa = table_layout(ws_a, M, dim), wb = table_layout(ws_b, M, dim);
    FR_CHECK_ARG(ws_bytes >= wa.bytes, "fr_table_sort2: workspace %zu < %zu bytes", ws_bytes, wa.bytes);
    const Lay lay{chunk, stride};
    SortJob sa{idx_a, n_rows_a, wa.perm, wa.seg_start, wa.seg_row, nullptr, wa.nseg, nullptr, nullptr, lay};
    SortJob sb{idx_b, n_rows_b, wb.perm, wb.seg_start, wb.seg_row, nullptr, wb.nseg, nullptr, nullptr, lay};
    sa.seg_first = wa.seg_first;
    sb.seg_first = wb.seg_first;
    return launch_sort(sa, &sb, M, err_flag, (hipStream_t)stream_);
}

static int apply_grad_impl(const char* who, const fr_table* ta, const fr_table* tb, const fr_adam* adam, int64_t M,
                           int32_t chunk, int32_t stride, const float* rows_a, const float* grad_a, const float* rows_b,
                           const float* grad_b, int32_t sweep_a, int32_t sweep_b, void* ws_a, void* ws_b, size_t ws_bytes,
                           hipStream_t stream, int64_t Mb = -1, size_t ws_b_bytes = 0) {
    int rc;
    if (Mb < 0) Mb = M, ws_b_bytes = ws_bytes;
    if ((rc = check_table(ta, who)) || (tb && (rc = check_table(tb, who))) || (rc = check_adam(adam, who))) return rc;
    FR_CHECK_ARG(rows_a && grad_a && ws_a && M >= 1 && M <= FR_SORT_MAX && ta->step >= 1 && lay_ok(chunk, stride),
                 "%s: bad argument", who);
    FR_CHECK_ARG(!tb || (rows_b && grad_b && ws_b && ws_b != ws_a && tb->dim == ta->dim && tb->step >= 1 && Mb >= 1 &&
                         Mb <= FR_SORT_MAX && (Mb == M || chunk == 0)),
                 "%s: bad second table", who);
    const Lay lay{chunk, stride};
    TableWs wa = table_layout(ws_a, M, ta->dim);
    TableWs wb = tb ? table_layout(ws_b, Mb, tb->dim) : wa;
    FR_CHECK_ARG(ws_bytes >= wa.bytes && (!tb || ws_b_bytes >= wb.bytes), "%s: workspace %zu < %zu bytes", who, ws_bytes, wa.bytes);
    if ((rc = side_join(ws_a, stream)) || (tb && (rc = side_join(ws_b, stream)))) return rc;
    // waves reserved for the sweeper = rows of a full slice (the slice itself depends on the effective step, which may
    // live on the device)
    const int per_wave = sweep_rows_per_wave(ta->dim);
    const long long sw_a = sweep_a > 0 ? ((ta->n_rows + sweep_a - 1) / sweep_a + per_wave - 1) / per_wave : 0;
    const long long sw_b = tb && sweep_b > 0 ? ((tb->n_rows + sweep_b - 1) / sweep_b + per_wave - 1) / per_wave : 0;
    const long long waves = std::max(M + sw_a, tb ? Mb + sw_b : 0);
    const AdamC c = make_adamc(adam);
    ApplyJob ja{view(ta), wa, rows_a, grad_a, (int)sweep_a, (int)sw_a, (long long)M};
    ApplyJob jb = tb ? ApplyJob{view(tb), wb, rows_b, grad_b, (int)sweep_b, (int)sw_b, (long long)Mb} : ja;
    {
        ProfScope prof(K_TABLE_APPLY_GRAD, stream);
        FR_DISPATCH_E(ta->dim, FR_LAUNCH(prof, (table_apply_grad_kernel<E>), dim3((unsigned)((waves + 3) / 4), tb ? 2 : 1), dim3(256), 0, stream, ja, jb, c, (long long)M, lay));
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_table_apply_grad(const fr_table* t, const fr_adam* adam, int64_t M, int32_t chunk, int32_t stride,
                                   const float* rows, const float* grad_rows, int32_t sweep_period, void* ws,
                                   size_t ws_bytes, void* stream_) {
    return apply_grad_impl("fr_table_apply_grad", t, nullptr, adam, M, chunk, stride, rows, grad_rows, nullptr, nullptr,
                           sweep_period, 0, ws, nullptr, ws_bytes, (hipStream_t)stream_);
}

extern "C" int fr_table_apply_grad2(const fr_table* ta, const fr_table* tb, const fr_adam* adam, int64_t M,
                                    int32_t chunk, int32_t stride, const float* rows_a, const float* grad_a,
                                    const float* rows_b, const float* grad_b, int32_t sweep_a, int32_t sweep_b,
                                    void* ws_a, void* ws_b, size_t ws_bytes, void* stream_) {
    FR_CHECK_ARG(tb, "fr_table_apply_grad2: second table is null");
    return apply_grad_impl("fr_table_apply_grad2", ta, tb, adam, M, chunk, stride, rows_a, grad_a, rows_b, grad_b, sweep_a,
                           sweep_b, ws_a, ws_b, ws_bytes, (hipStream_t)stream_);
}

// fr_table_apply_grad on TWO tables of one width in one launch, each with its own id count (a user table next to the item table
// that was looked up with the positive and the negative ids): the shorter table's update and sweep slice run beside the
// longer one's instead of in a launch behind it.  Same results as the two calls.
extern "C" int fr_table_apply_grad_two(const fr_table* ta, const fr_table* tb, const fr_adam* adam, int64_t Ma, int64_t Mb,
                                       const float* rows_a, const float* grad_a, const float* rows_b, const float* grad_b,
                                       int32_t sweep_a, int32_t sweep_b, void* ws_a, size_t ws_a_bytes, void* ws_b,
                                       size_t ws_b_bytes, void* stream_) {
    FR_CHECK_ARG(tb && Mb >= 1, "fr_table_apply_grad_two: second table is null");
    return apply_grad_impl("fr_table_apply_grad_two", ta, tb, adam, Ma, 0, 0, rows_a, grad_a, rows_b, grad_b, sweep_a, sweep_b,
                           ws_a, ws_b, ws_a_bytes, (hipStream_t)stream_, Mb, ws_b_bytes);
}
