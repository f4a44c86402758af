// FairGo graph kernels: CSR SpMM over the row-normalised rating graph, row gather / duplicate-summed row scatter on
// whole-table activations, MSE head.
//
// Replaces fairgo_pmf.py:196-200 (torch.sparse.mm(L, E), L = D^-1 A of get_norm_rating_matrix :102-129), the
// `all_embeddings[user]` indexing (:178-179, :194) with its index_put backward, and nn.MSELoss (:182).
#include <stdlib.h>

#include <algorithm>

#include "common.hpp"
#include "kernels.hpp"
#include "mlp_act.hpp"

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

// An output row on its way out through the derivative of an activation (fr_spmm_csr_sel_act: the product is the gradient at an
// activation's OUTPUT, what goes on is the gradient at its input -- FairGo's filter step, where the whole-table gradient of
// the filtered table otherwise takes a pass of its own through fr_act_bwd: 17 GB of traffic for a table of which the batch's
// frontier touches a few per cent).  src = that output (act_bwd's argument) at the output rows; a row whose `skip` bit is set
// is left as it is: a later pass adds to it first and scales the sum (fr_row_scatter_add_act).  A row no term reached is
// stored as the zeros it is, without reading src.
struct RowAct {
    const float* src;
    const unsigned* skip;
    int act;
};
template <typename vec, int V>
__device__ __forceinline__ vec row_act(vec acc, const RowAct& ra, long long i, int D, int lane) {
    if (ra.skip && ((ra.skip[i >> 5] >> (i & 31)) & 1u)) return acc;
    const vec y = *reinterpret_cast<const vec*>(ra.src + (size_t)i * D + lane * V);
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = acc[e] * act_bwd(y[e], ra.act);
    return acc;
}


// Y[i,:] = sum over the nonzeros j of row r = (rows ? rows[i] : i), in ascending j, of val[j] * X[xrow(col[j]),:] with
// xrow(c) = map ? map[c] : c, nonzeros whose map entry is negative SKIPPED.  The frontier-restricted graph propagation of
// FairGo (fairgo_pmf.py:196-200 evaluated only where the batch can see it): `rows` selects the output rows (compact Y),
// `map` lets X be a compact [n_x, D] block of a whole-table operand -- or, in the backward direction on the CSR of L^T, names
// the columns whose gradient rows exist at all.  Every kept term is added in the order spmm_csr_kernel adds it and a
// skipped term is one that kernel would add as an exact zero, so the results are those of the whole-table product.
// The hits of a 64-nonzero chunk are walked through a ballot mask: a row of L^T with 20 nonzeros of which none is mapped
// costs one (col, map) load pair, no row gather.
template <int V, bool ACT = false>
__global__ __launch_bounds__(256) void spmm_csr_sel_kernel(const long long* __restrict__ indptr, const int* __restrict__ col,
                                                           const float* __restrict__ val, const float* __restrict__ X,
                                                           const int* __restrict__ rows, long long n_out,
                                                           const int* __restrict__ map, const unsigned* __restrict__ bits,
                                                           int D, float* __restrict__ Y, RowAct ra) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_out) return;
    const long long r = rows ? (long long)rows[i] : i;
    const long long j0 = indptr[r], j1 = indptr[r + 1];
    if constexpr (V == 0) {
        for (int d = lane; d < D; d += 64) {
            float acc = 0.f;
            for (long long j = j0; j < j1; ++j) {
                const int c = map ? map[col[j]] : col[j];
                if (c >= 0) acc = fmaf(val[j], X[(size_t)c * D + d], acc);
            }
            if (ACT && !(ra.skip && ((ra.skip[i >> 5] >> (i & 31)) & 1u))) acc = acc * act_bwd(ra.src[(size_t)i * D + d], ra.act);
            Y[(size_t)i * D + d] = acc;
        }
    } else {
        typedef float vec __attribute__((ext_vector_type(V)));
        vec acc = {};
        bool touched = false;
        for (long long jb = j0; jb < j1; jb += 64) {
            const int cnt = (int)min((long long)64, j1 - jb);
            int my_c = -1;
            float my_v = 0.f;
            if (lane < cnt) {
                my_c = col[jb + lane];
                if (map) my_c = (!bits || ((bits[my_c >> 5] >> (my_c & 31)) & 1u)) ? map[my_c] : -1;
                if (my_c >= 0) my_v = val[jb + lane];
            }
            unsigned long long hits = __ballot(my_c >= 0);
            touched |= hits != 0ull;
            // four hits at a time: their row gathers are independent loads in flight together (one at a time the loop is a
            // chain of memory round trips: 36 per row of the first layer); the terms are still added one by one in CSR order
            while (hits) {
                int c[4];
                float v[4];
                vec x[4];
                int n = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (hits) {     // wave-uniform
                        const int t = __ffsll((long long)hits) - 1;
                        hits &= hits - 1;
                        c[q] = __builtin_amdgcn_readlane(my_c, t);
                        v[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_v), t));
                        x[q] = *reinterpret_cast<const vec*>(X + (size_t)c[q] * D + lane * V);
                        n = q + 1;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < n) {
#pragma unroll
                        for (int e = 0; e < V; ++e) acc[e] = fmaf(v[q], x[q][e], acc[e]);
                    }
                }
            }
        }
        if (ACT && touched) acc = row_act<vec, V>(acc, ra, i, D, lane);
        *reinterpret_cast<vec*>(Y + (size_t)i * D + lane * V) = acc;
    }
}

// The same for ALL rows of the matrix through a column map that keeps few of them -- the backward of the first selected
// layer walks L^T's 11 M rows (36 nonzeros each at BASELINE configs[3]) to find the 1.4 % of nonzeros that carry a gradient
// row: one wave per row is four dependent loads for half a chunk of columns, and the launch is bound by wave turnover
// (11 M waves).  Here a wave takes EIGHT consecutive rows as one contiguous run of nonzeros (their boundaries sit in lanes
// 0 .. 8) and looks the columns up 256 at a time: column ids, then their bitmap words, then map entry AND value of the set
// ones -- three round trips per 256 nonzeros.  The few hits (3.6 per trip) are compacted in CSR order into a wave-private list
// in LDS and their gradient rows requested FOUR AT A TIME (round 6; one after the other, each behind the previous one's
// accumulation, they were a memory round trip each: 6.5 -> see profiles/README.md); a row's terms are still added in CSR
// order into one accumulator that is stored when the run moves on to the next row (rows without a hit are stored as zeros):
// the same bits as the kernel above.
constexpr int SEL_BATCH = 4;
template <int V, int R, bool ACT = false>
__global__ __launch_bounds__(256) void spmm_csr_sel_runs_kernel(const long long* __restrict__ indptr, const int* __restrict__ col,
                                                                const float* __restrict__ val, const float* __restrict__ X,
                                                                long long n_out, const int* __restrict__ map,
                                                                const unsigned* __restrict__ bits, int D,
                                                                float* __restrict__ Y, RowAct ra) {
    typedef float vec __attribute__((ext_vector_type(V)));
    __shared__ int4 stage_all[4][256];      // per wave: (mapped column, value bits, nonzero index within the trip, -)
    const int lane = threadIdx.x & 63;
    int4* const stage = stage_all[threadIdx.x >> 6];
    const long long r0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (r0 >= n_out) return;
    const int nr = (int)min((long long)R, n_out - r0);
    const long long bnd = lane <= nr ? indptr[r0 + lane] : 0x7fffffffffffffffll;
    const long long j0 = __shfl(bnd, 0, 64), j1 = __shfl(bnd, nr, 64);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    vec acc = {};
    int cur = 0;
    bool touched = false;      // (wave-uniform) the row being accumulated has had a term
    vec ycur = {};             // ACT: the activation's output at that row
    bool keep = false;         // ... and whether the row is left unscaled (skip bit)
    int yrow = -1;             // ... the row of the run whose activation row was requested last (once per row with a term)
    auto flush_to = [&](int row) {
        while (cur < row) {
            if (ACT && touched && !keep) {
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] = acc[e] * act_bwd(ycur[e], ra.act);
            }
            *reinterpret_cast<vec*>(Y + (size_t)(r0 + cur) * D + lane * V) = acc;
            acc = vec{};
            touched = false;
            ++cur;
        }
    };
    for (long long jq = j0; jq < j1; jq += 256) {
        int cc4[4], mc4[4];
        unsigned w4[4];
        float mv4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long j = jq + 64 * q + lane;
            cc4[q] = j < j1 ? col[j] : -1;
        }
        // (`bits`: one bit per column, set where map >= 0 -- 1.4 MB for 11 M columns, resident in every L2, where the map
        // itself is 44 MB and each look-up of it drags a cache line out of the memory-side cache: 8.0 -> 5.1 ms per launch)
#pragma unroll
        for (int q = 0; q < 4; ++q) w4[q] = (bits && cc4[q] >= 0) ? bits[cc4[q] >> 5] : 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < 4; ++q) {      // a set bit says the map entry is a row: entry and value are requested together
            const bool set = cc4[q] >= 0 && ((w4[q] >> (cc4[q] & 31)) & 1u);
            mc4[q] = set ? map[cc4[q]] : -1;
            mv4[q] = set ? val[jq + 64 * q + lane] : 0.f;
        }
        int n = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool hit = mc4[q] >= 0;
            const unsigned long long hq = __ballot(hit);
            if (hit) stage[n + __popcll(hq & lt_mask)] = make_int4(mc4[q], __builtin_bit_cast(int, mv4[q]), 64 * q + lane, 0);
            n += __popcll(hq);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k0 = 0; k0 < n; k0 += SEL_BATCH) {
            int rowe[SEL_BATCH];
            float ve[SEL_BATCH];
            vec xe[SEL_BATCH], ye[SEL_BATCH];
            bool ke[SEL_BATCH], fresh[SEL_BATCH];
#pragma unroll
            for (int e = 0; e < SEL_BATCH; ++e) {
                if (k0 + e < n) {      // wave-uniform
                    const int4 en = stage[k0 + e];
                    const int c = __builtin_amdgcn_readfirstlane(en.x);
                    ve[e] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(en.y));
                    const long long j = jq + __builtin_amdgcn_readfirstlane(en.z);
                    xe[e] = *reinterpret_cast<const vec*>(X + (size_t)c * D + lane * V);
                    rowe[e] = __popcll(__ballot(lane >= 1 && lane <= nr && bnd <= j));      // rows of the run that end at or before j
                    fresh[e] = ACT && rowe[e] != yrow;      // wave-uniform
                    if (fresh[e]) {
                        const long long i = r0 + rowe[e];
                        ke[e] = ra.skip && ((ra.skip[i >> 5] >> (i & 31)) & 1u);
                        ye[e] = *reinterpret_cast<const vec*>(ra.src + (size_t)i * D + lane * V);
                        yrow = rowe[e];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < SEL_BATCH; ++e) {
                if (k0 + e < n) {
                    flush_to(rowe[e]);
                    if (fresh[e]) {
                        ycur = ye[e];
                        keep = ke[e];
                    }
                    touched = true;
#pragma unroll
                    for (int q = 0; q < V; ++q) acc[q] = fmaf(ve[e], xe[e][q], acc[q]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();      // (the next trip's list goes into the same words)
    }
    flush_to(nr);
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

// dX[seg_row[k],:] = sum over the segment's members (ascending position) of g[member,:]; dX is pre-zeroed -- or (ADD) holds
// another contribution to the same gradient already, to which the segment's sum is added
template <bool ADD>
__global__ __launch_bounds__(256) void row_scatter_sum_kernel(const float* __restrict__ g, const int32_t* __restrict__ perm,
                                                              const int32_t* __restrict__ seg_start,
                                                              const int32_t* __restrict__ seg_row,
                                                              const int32_t* __restrict__ nseg, int M, int D,
                                                              float* __restrict__ dX, const float* __restrict__ act_src, int act) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= M || k >= nseg[0]) return;
    const int j0 = seg_start[k], j1 = seg_start[k + 1];
    const size_t row = (size_t)seg_row[k];
    for (int d = lane; d < D; d += 64) {
        float acc = 0.f;
        for (int j = j0; j < j1; ++j) acc += g[(size_t)perm[j] * D + d];
        float out = ADD ? dX[row * D + d] + acc : acc;
        if (act_src) out = out * act_bwd(act_src[row * D + d], act);      // (fr_row_scatter_add_act: the sum, then the derivative)
        dX[row * D + d] = out;
    }
}

// p[0 .. n) = 0 as a KERNEL (float4 stores over the 16-byte aligned middle, grid-stride): see fr_row_scatter_sum for why
// this is not hipMemsetAsync
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, size_t n) {
    const size_t head = min(n, (size_t)((16 - ((uintptr_t)p & 15)) & 15) / 4);     // floats before the first aligned one
    const size_t n4 = (n - head) / 4;
    float4* p4 = reinterpret_cast<float4*>(p + head);
    const size_t stride = (size_t)gridDim.x * 256, t = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = t; i < n4; i += stride) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < head) p[t] = 0.f;
    const size_t done = head + 4 * n4;
    if (t < n - done) p[done + t] = 0.f;
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

static int spmm_csr_sel_impl(const int64_t* indptr, const int32_t* col, const float* val, const float* X, const int32_t* rows,
                             int64_t n_out, const int32_t* map, const uint32_t* map_bits, int32_t dim, float* Y, RowAct ra,
                             void* stream_);

extern "C" int fr_spmm_csr_sel(const int64_t* indptr, const int32_t* col, const float* val, const float* X, const int32_t* rows,
                               int64_t n_out, const int32_t* map, const uint32_t* map_bits, int32_t dim, float* Y,
                               void* stream_) {
    return spmm_csr_sel_impl(indptr, col, val, X, rows, n_out, map, map_bits, dim, Y, RowAct{nullptr, nullptr, 0}, stream_);
}

extern "C" int fr_spmm_csr_sel_act(const int64_t* indptr, const int32_t* col, const float* val, const float* X, const int32_t* rows,
                                   int64_t n_out, const int32_t* map, const uint32_t* map_bits, int32_t dim, float* Y,
                                   const float* act_src, int32_t act, const uint32_t* skip_bits, void* stream_) {
    FR_CHECK_ARG(act_src && act >= 1 && act <= 4 && (((uintptr_t)act_src) & 15) == 0,
                 "fr_spmm_csr_sel_act: the activation's output at the output rows (16-byte aligned) and its code 1..4");
    return spmm_csr_sel_impl(indptr, col, val, X, rows, n_out, map, map_bits, dim, Y, RowAct{act_src, skip_bits, (int)act}, stream_);
}

static int spmm_csr_sel_impl(const int64_t* indptr, const int32_t* col, const float* val, const float* X, const int32_t* rows,
                             int64_t n_out, const int32_t* map, const uint32_t* map_bits, int32_t dim, float* Y, RowAct ra,
                             void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(indptr && col && val && X && Y && n_out >= 0 && dim >= 1 && (map || !map_bits), "fr_spmm_csr_sel: bad argument");
    const bool with_act = ra.src != nullptr;
    const unsigned* bits = map_bits;
    if (n_out == 0) return FR_OK;
    ProfScope prof(K_SPMM, stream);
    if (!rows && map && n_out >= 1024 && (dim == 64 || dim == 128 || dim == 256) && !getenv("FAIRREC_SPMM_SEL_ROWWISE")) {
        // rows per wave (one contiguous run of nonzeros): 8, or FAIRREC_SEL_RUNS_ROWS = 16 / 32 (A/B)
        const char* e = getenv("FAIRREC_SEL_RUNS_ROWS");
        const int R = e ? atoi(e) : 8;
#define FR_SEL_RUNS(V, RR)                                                                                                 \
    if (with_act) {                                                                                                        \
        FR_LAUNCH(prof, (spmm_csr_sel_runs_kernel<V, RR, true>), dim3((unsigned)((n_out + 4 * RR - 1) / (4 * RR))), dim3(256), 0, \
                  stream, (const long long*)indptr, col, val, X, (long long)n_out, map, bits, (int)dim, Y, ra);            \
    } else {                                                                                                               \
        FR_LAUNCH(prof, (spmm_csr_sel_runs_kernel<V, RR, false>), dim3((unsigned)((n_out + 4 * RR - 1) / (4 * RR))), dim3(256), 0, \
                  stream, (const long long*)indptr, col, val, X, (long long)n_out, map, bits, (int)dim, Y, ra);            \
    }
#define FR_SEL_RUNS_V(V)                  \
    if (R == 32) { FR_SEL_RUNS(V, 32); }  \
    else if (R == 16) { FR_SEL_RUNS(V, 16); } \
    else { FR_SEL_RUNS(V, 8); }
        if (dim == 64) { FR_SEL_RUNS_V(1) }
        else if (dim == 128) { FR_SEL_RUNS_V(2) }
        else { FR_SEL_RUNS_V(4) }
#undef FR_SEL_RUNS_V
#undef FR_SEL_RUNS
        FR_CHECK_LAUNCH();
        return FR_OK;
    }
    const dim3 grid((unsigned)((n_out + 3) / 4));
#define FR_SPMM_SEL(V)                                                                                                  \
    if (with_act) {                                                                                                     \
        FR_LAUNCH(prof, (spmm_csr_sel_kernel<V, true>), grid, dim3(256), 0, stream, (const long long*)indptr, col, val, X, rows, \
                  (long long)n_out, map, bits, (int)dim, Y, ra);                                                        \
    } else {                                                                                                            \
        FR_LAUNCH(prof, (spmm_csr_sel_kernel<V, false>), grid, dim3(256), 0, stream, (const long long*)indptr, col, val, X, rows, \
                  (long long)n_out, map, bits, (int)dim, Y, ra);                                                        \
    }
    switch (dim) {
        case 64: FR_SPMM_SEL(1); break;
        case 128: FR_SPMM_SEL(2); break;
        case 256: FR_SPMM_SEL(4); break;
        default: FR_SPMM_SEL(0); break;
    }
#undef FR_SPMM_SEL
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
static int row_scatter_impl(const float* g, const int64_t* idx, int64_t M, int64_t n_rows, int32_t dim, float* dX, void* ws,
                            size_t ws_bytes, uint32_t* err_flag, void* stream_, bool add, const float* act_src = nullptr,
                            int act = 0);

extern "C" int fr_row_scatter_sum(const float* g, const int64_t* idx, int64_t M, int64_t n_rows, int32_t dim, float* dX,
                                  void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream_) {
    return row_scatter_impl(g, idx, M, n_rows, dim, dX, ws, ws_bytes, err_flag, stream_, false);
}

// ... the same into a dX that already holds another contribution to the gradient (no clear; the segment sums are ADDED)
extern "C" int fr_row_scatter_add(const float* g, const int64_t* idx, int64_t M, int64_t n_rows, int32_t dim, float* dX,
                                  void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream_) {
    return row_scatter_impl(g, idx, M, n_rows, dim, dX, ws, ws_bytes, err_flag, stream_, true);
}

// ... and each row the batch touches, once its sum is complete, goes on through the derivative of the activation whose output
// at that row is act_src[row] (the rows fr_spmm_csr_sel_act skipped)
extern "C" int fr_row_scatter_add_act(const float* g, const int64_t* idx, int64_t M, int64_t n_rows, int32_t dim, float* dX,
                                      void* ws, size_t ws_bytes, const float* act_src, int32_t act, uint32_t* err_flag,
                                      void* stream_) {
    FR_CHECK_ARG(act_src && act >= 1 && act <= 4, "fr_row_scatter_add_act: the activation's output [n_rows, dim] and its code 1..4");
    return row_scatter_impl(g, idx, M, n_rows, dim, dX, ws, ws_bytes, err_flag, stream_, true, act_src, (int)act);
}

static int row_scatter_impl(const float* g, const int64_t* idx, int64_t M, int64_t n_rows, int32_t dim, float* dX, void* ws,
                            size_t ws_bytes, uint32_t* err_flag, void* stream_, bool add, const float* act_src, int act) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(g && idx && dX && ws && M >= 1 && M <= FR_SORT_MAX && dim >= 1 && ws_bytes >= fr_row_scatter_workspace_bytes(M),
                 "fr_row_scatter_sum / _add: bad argument");
    char* p = (char*)ws;
    const size_t stride = align_up(((size_t)M + 1) * 4, 256);
    int32_t* perm = (int32_t*)p;
    int32_t* seg_start = (int32_t*)(p + stride);
    int32_t* seg_row = (int32_t*)(p + 2 * stride);
    int32_t* nseg = (int32_t*)(p + 3 * stride);
    SortJob job{idx, n_rows, perm, seg_start, seg_row, nullptr, nseg, nullptr, nullptr};
    int rc = launch_sort(job, nullptr, M, err_flag, stream);
    if (rc) return rc;
    // The zero fill is a kernel, not hipMemsetAsync: inside a stream capture a memset becomes a memset NODE, and on this
    // runtime (ROCm 7.0 / 7.2) the node of a replayed graph did not reliably clear the captured address -- rows no member of
    // the batch touches kept what the block's previous tenant had left there, i.e. the gradient of the whole filtered
    // table carried garbage rows into every filter weight ("the captured-step NaN"; pinned by
    // tests/test_graph_hip.py::test_captured_pieces_are_idempotent_under_allocator_churn).  FAIRREC_SCATTER_MEMSET=1 restores
    // the memset for A/B runs.
    static const bool use_memset = getenv("FAIRREC_SCATTER_MEMSET") != nullptr;
    const size_t n = (size_t)n_rows * dim;
    if (add) {
        // nothing to clear
    } else if (use_memset) {
        FR_CHECK_HIP(hipMemsetAsync(dX, 0, n * sizeof(float), stream));
    } else {
        const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>((n / 4 + 255) / 256, 1), 4096);
        hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, stream, dX, n);
        FR_CHECK_LAUNCH();
    }
    ProfScope prof(K_ROW_GATHER, stream);
    if (add) {
        FR_LAUNCH(prof, row_scatter_sum_kernel<true>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, g, (const int32_t*)perm,
                  (const int32_t*)seg_start, (const int32_t*)seg_row, (const int32_t*)nseg, (int)M, (int)dim, dX, act_src, act);
    } else {
        FR_LAUNCH(prof, row_scatter_sum_kernel<false>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, g, (const int32_t*)perm,
                  (const int32_t*)seg_start, (const int32_t*)seg_row, (const int32_t*)nseg, (int)M, (int)dim, dX, act_src, act);
    }
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
