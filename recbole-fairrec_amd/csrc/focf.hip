// FOCF training step on MI355X: fused lazy-Adam gather + dot, fairness statistics, backward + Adam.
//
// Reference being replaced (all stock PyTorch ops called from Python):
//   FOCF.forward            focf.py:136-143   two nn.Embedding gathers + mul + sum
//   FOCF.calculate_loss     focf.py:152-169   MSELoss + fair_weight * unfairness term
//   FOCF.get_item_ratings   focf.py:75-91     2x torch.unique + 3x index_put_(accumulate)
//   *_unfairness            focf.py:93-134    smooth_l1 on per-item group means
//   loss.backward()         trainer.py:193    dense embedding_dense_backward (N x D per table)
//   optimizer.step()        trainer.py:196    dense torch.optim.Adam over both tables
//
// Kernel chain per batch (one wave = one interaction / one distinct row; lane = embedding column):
//   sort      : (row,pos) sort + segmentation of user ids and item ids, min/max of the sst column
//   gather    : rows read from HBM once, missed Adam steps replayed in registers, dot product,
//               caught-up (p,m,v) parked in the workspace (L2/Infinity-Cache resident), MSE partials
//   fair      : per distinct item: group sums in batch order -> unfairness term and dLoss/dpred
//   finalize  : fixed-order reduction of the partials -> loss scalar
//   bwd_adam  : per distinct row: sum_b dLoss/dpred[b] * other_row[b] (batch order), Adam step,
//               row written back once; plus the bounded-staleness sweeper slice
#include "common.hpp"
#include "kernels.hpp"
#include "table.hpp"
#include "focf_ws.hpp"

namespace fr {


}  // namespace fr
#include "focf_gather.hpp"
namespace fr {

// SHARE: waves of a workgroup that hold the same item share its replay (FR_FOCF_ITEM_RUNS: item-complete batches)
template <int E, bool TRAIN, bool SHARE>
__global__ __launch_bounds__(GATHER_THREADS) void focf_gather_kernel(
    TableV U, TableV I, AdamC c, const int64_t* __restrict__ user, const int64_t* __restrict__ item,
    const float* __restrict__ rating, int B, int upto_u, int upto_i, FocfWs w, float max_rating,
    float* __restrict__ predict_out, uint32_t* err, DeferLoss dl) {
    __shared__ GatherLds<SHARE ? E : 0> lds;
    if (TRAIN && blockIdx.x == 0 && threadIdx.x == 0) *w.defer = dl;   // what the backward launch still has to reduce
    focf_gather_body<E, TRAIN, SHARE>(U, I, c, user, item, rating, B, upto_u, upto_i, w, max_rating, predict_out, err,
                                      (int)blockIdx.x, lds);
}



// The training gather of an item-complete batch with a share of the step's sweep slice in FRONT of it (fr_focf_step_runs):
// the gather waves spend their life waiting for rows, the sweeper waves are VALU work -- dispatched first, they run under the
// gather's latency.  (The sweeper tells the batch's rows by their stamps, which fr_focf_prepare_step wrote before this launch.)
template <int E>
__global__ __launch_bounds__(GATHER_THREADS) void focf_gather_sweep_kernel(
    TableV U, TableV I, AdamC c, const int64_t* __restrict__ user, const int64_t* __restrict__ item,
    const float* __restrict__ rating, int B, int upto_u, int upto_i, FocfWs w, uint32_t* err, DeferLoss dl, SweepSlice sw,
    long long sw_n, SortedPark sp) {
    __shared__ GatherLds<E> lds;
    const int nsb = (int)((sw_n + 3) / 4);
    if ((int)blockIdx.x < nsb) {
        const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
        if (wave < sw_n) sweep_slice_wave<E>(U, I, c, sw, wave, (int)(threadIdx.x & 63));
        return;
    }
    const int block = (int)blockIdx.x - nsb;
    if (block == 0 && threadIdx.x == 0) *w.defer = dl;
    focf_gather_body<E, true, true, true>(U, I, c, user, item, rating, B, upto_u, upto_i, w, 0.f, nullptr, err, block, lds, sp);
}

// ------------------------------------------------------------------------------------------------
// fairness term on the distinct items of the batch: 16 lanes per item
// ------------------------------------------------------------------------------------------------

// Segment arrays + member columns of the fairness term.  Single GPU: members are batch positions; sharded:
// members are exchange-buffer slots on the item's owner rank and `pred/rating/sst` are the received records.
struct FairArgs {
    const int32_t *perm, *seg_start, *nseg;
    const int32_t* seg_first;  // optional: perm[seg_start[k]] (saves the perm round trip of single-member segments)
    const float* minmax;       // (min, max) of the sst column; sharded: one pair per source rank, mm_stride floats
    int mm_count, mm_stride;   //   apart (they arrive with the id exchange), folded here over the global batch
    const float *pred, *rating, *sst;
    Lay mlay;                  // layout of the three member columns (sharded: planes of a [G, 3, cap] record buffer)
    float* coef;               // dLoss/dpred of every member
    Lay clay;                  // its layout (sharded: [G, cap + 3] reply buffer)
    float* fair_part;          // [gridDim.x] partial sums of the smooth-L1 terms
    int accumulate;            // 1: coef[b] += g, 0: coef[b] = g
    // optional in-launch finalisation by the last block to arrive (single-GPU path): loss = mse + fw * fair
    unsigned int* ticket;      // zero on entry, reset to zero by the last block; nullptr = no finalisation
    const float* mse_part;     // [n_mse_part] partial sums of squared errors (written by an EARLIER launch)
    int n_mse_part, batch;
    float* loss_out;           // [3] loss, mse, fair
    // sharded finalisation instead: tail of every destination's chunk of the reply buffer = (K_owner, fair_owner, sq_rank)
    float* tails;              // reply + cap; nullptr = single-GPU finalisation into loss_out
    int tail_count, tail_stride;
};

template <int FAIR_GROUP>
__global__ __launch_bounds__(FAIR_THREADS) void focf_fair_kernel(FairArgs w, int objective, float fair_weight,
                                                                 int defer_k, uint32_t* err) {
    const int sub = threadIdx.x & (FAIR_GROUP - 1);
    const int gib = threadIdx.x / FAIR_GROUP;
    const int k = blockIdx.x * (FAIR_THREADS / FAIR_GROUP) + gib;
    const int K = w.nseg[0];
    float smin = w.minmax[0], smax = w.minmax[1];
    for (int q = 1; q < w.mm_count; ++q) {
        smin = fminf(smin, w.minmax[q * w.mm_stride]);
        smax = fmaxf(smax, w.minmax[q * w.mm_stride + 1]);
    }
    __shared__ float red[FAIR_THREADS / FAIR_GROUP];
    float term = 0.f;
    if (k < K) {
        const int j0 = w.seg_start[k], j1 = w.seg_start[k + 1];
        const int first_b = w.seg_first ? w.seg_first[k] : -1;
        float sp0 = 0.f, sp1 = 0.f, st0 = 0.f, st1 = 0.f, n0 = 0.f, n1 = 0.f;
        bool bad = false;
        // this lane's first member: its dL/dpred so far (the MSE part) is requested together with its values, so that
        // the read-modify-write at the end is not one more dependent round trip
        float c_first = 0.f;
        int b_first = -1;
#pragma unroll 4
        for (int j = j0 + sub; j < j1; j += FAIR_GROUP) {
            const int b = (j == j0 && first_b >= 0) ? first_b : w.perm[j];
            const long long bp = w.mlay.at(b);
            if (j == j0 + sub) {
                b_first = b;
                if (w.accumulate) c_first = w.coef[w.clay.at(b)];
            }
            const float s = w.sst[bp], pr = w.pred[bp], r = w.rating[bp];
            bad |= (s != smin && s != smax);
            if (s == smin) {
                sp0 += pr; st0 += r; n0 += 1.f;
            } else {
                sp1 += pr; st1 += r; n1 += 1.f;
            }
        }
        if (bad && err) atomicOr(err, FR_DEV_ERR_SST_GROUPS);
        sp0 = group_sum<FAIR_GROUP>(sp0); sp1 = group_sum<FAIR_GROUP>(sp1);
        st0 = group_sum<FAIR_GROUP>(st0); st1 = group_sum<FAIR_GROUP>(st1);
        n0 = group_sum<FAIR_GROUP>(n0);   n1 = group_sum<FAIR_GROUP>(n1);
        // d(fw * mean_k sl1)/d delta; sharded: K is only known after an all-reduce, the requester divides later
        float g0, g1;
        focf_fair_eval(objective, fair_weight, defer_k ? 1.f : (float)K, sp0, sp1, st0, st1, n0, n1, term, g0, g1);
        for (int j = j0 + sub; j < j1; j += FAIR_GROUP) {
            const bool first = j == j0 + sub;
            const int b = first ? b_first : w.perm[j];
            const float g = (w.sst[w.mlay.at(b)] == smin) ? g0 : g1;
            const long long cp = w.clay.at(b);
            w.coef[cp] = w.accumulate ? (first ? c_first : w.coef[cp]) + g : g;
        }
    }
    if (sub == 0) red[gib] = term;
    __syncthreads();
    __shared__ int is_last;
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < FAIR_THREADS / FAIR_GROUP; ++q) s += red[q];
        is_last = 0;
        if (w.ticket) {
            // publish this block's partial with a write-through (sc1) store -- no L2 write-back fence, the L2s are
            // full of the gather kernel's dirty lines -- drain it, then draw a ticket; the block that draws the last
            // one reads every partial with sc1 loads and reduces them in index order, so the loss is
            // bit-reproducible whatever the arrival order (cdna_hip_programming.md, Guideline 16, sc1 form)
            __hip_atomic_store(&w.fair_part[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(w.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            is_last = (t == gridDim.x - 1) ? 1 : 0;
        } else {
            w.fair_part[blockIdx.x] = s;
        }
    }
    __syncthreads();
    if (!is_last) return;
    float a = 0.f, fsum = 0.f;
    for (int q = threadIdx.x; q < w.n_mse_part; q += FAIR_THREADS) a += w.mse_part[q];
    for (int q = threadIdx.x; q < (int)gridDim.x; q += FAIR_THREADS)
        fsum += __hip_atomic_load(&w.fair_part[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = wave_sum(a);
    fsum = wave_sum(fsum);
    __shared__ float red2[2][FAIR_THREADS / 64];
    if ((threadIdx.x & 63) == 0) {
        red2[0][threadIdx.x >> 6] = a;
        red2[1][threadIdx.x >> 6] = fsum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0.f;
        fsum = 0.f;
#pragma unroll
        for (int q = 0; q < FAIR_THREADS / 64; ++q) {
            a += red2[0][q];
            fsum += red2[1][q];
        }
        if (w.tails) {
            for (int g = 0; g < w.tail_count; ++g) {
                float* t = w.tails + (size_t)g * w.tail_stride;
                t[0] = (float)K;
                t[1] = fsum;
                t[2] = a;
            }
        } else {
            const float mse = a / (float)w.batch;
            const float fair = fsum / (float)K;
            w.loss_out[0] = mse + fair_weight * fair;
            w.loss_out[1] = mse;
            w.loss_out[2] = fair;
        }
        *w.ticket = 0u;
    }
}

// nonparity (focf.py:127-134): smooth_l1(mean(pred | g0), mean(pred | g1)); one workgroup, fixed order
__global__ __launch_bounds__(1024) void focf_nonparity_kernel(FocfWs w, const float* __restrict__ sst, int B,
                                                             float fair_weight, uint32_t* err) {
    __shared__ float red[4][16];
    const float smin = w.sst_minmax[0], smax = w.sst_minmax[1];
    float s0 = 0.f, s1 = 0.f, n0 = 0.f, n1 = 0.f;
    for (int b = threadIdx.x; b < B; b += 1024) {
        const float s = sst[b], pr = w.pred[b];
        if (s == smin) { s0 += pr; n0 += 1.f; }
        else if (s == smax) { s1 += pr; n1 += 1.f; }   // rows of a third group are ignored, as in the reference
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); n0 = wave_sum(n0); n1 = wave_sum(n1);
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wid] = s0; red[1][wid] = s1; red[2][wid] = n0; red[3][wid] = n1; }
    __syncthreads();
    s0 = s1 = n0 = n1 = 0.f;
    for (int q = 0; q < 16; ++q) { s0 += red[0][q]; s1 += red[1][q]; n0 += red[2][q]; n1 += red[3][q]; }
    if (smin == smax || n1 == 0.f) {  // reference: IndexError (sst_unique_value[1]) -- flag it
        if (threadIdx.x == 0) {
            if (err) atomicOr(err, FR_DEV_ERR_SST_GROUPS);
            w.fair_part[0] = 0.f;
        }
        return;
    }
    const float a = s0 / n0, bb = s1 / n1;
    const float delta = a - bb;
    const float dl = fminf(fmaxf(delta, -1.f), 1.f) * fair_weight;
    for (int b = threadIdx.x; b < B; b += 1024) {
        const float s = sst[b];
        if (s == smin) w.coef[b] += dl / n0;
        else if (s == smax) w.coef[b] -= dl / n1;
    }
    if (threadIdx.x == 0) w.fair_part[0] = smooth_l1(delta);
}

// fixed-order reduction of the partial sums of one batch -> loss (one block of 256 threads)
__device__ __forceinline__ void focf_finalize_block(const FocfWs& w, int B, int objective, float fair_weight,
                                                    float* __restrict__ loss_out, int n_fair_part) {
    __shared__ float red[2][4];
    float a = 0.f, f = 0.f;
    for (int q = threadIdx.x; q < w.n_gather_blocks; q += 256) a += w.mse_part[q];
    int nf = objective == FR_FOCF_NONE ? 0 : (objective == FR_FOCF_NONPARITY ? 1 : n_fair_part);
    for (int q = threadIdx.x; q < nf; q += 256) f += w.fair_part[q];
    a = wave_sum(a);
    f = wave_sum(f);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        f = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        const float mse = a / (float)B;
        float fair = 0.f;
        if (objective == FR_FOCF_NONPARITY) fair = f;
        else if (objective != FR_FOCF_NONE) fair = f / (float)w.nseg_i[0];
        loss_out[0] = objective == FR_FOCF_NONE ? mse : mse + fair_weight * fair;
        loss_out[1] = mse;
        loss_out[2] = fair;
    }
}

__global__ __launch_bounds__(256) void focf_finalize_kernel(FocfWs w, int B, int objective, float fair_weight,
                                                            float* __restrict__ loss_out) {
    focf_finalize_block(w, B, objective, fair_weight, loss_out, w.n_fair_blocks);
}


// ------------------------------------------------------------------------------------------------
// backward + Adam: one wave per distinct row, plus sweeper waves
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// clip_grad_norm_ (trainer.py:194-195) on gradients that are never materialised
// ------------------------------------------------------------------------------------------------
// squared 2-norm of the dense embedding gradient = sum over the distinct rows of the batch of |sum_b coef_b * other_b|^2:
// one wave per distinct row (users, then items), one partial per workgroup
template <int E>
__global__ __launch_bounds__(256) void focf_grad_sqnorm_kernel(int B, int D, FocfWs w) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    long long wv = (long long)blockIdx.x * 4 + wib;
    __shared__ float red[4];
    float sq = 0.f;
    const bool users = wv < B;
    if (!users) wv -= B;
    const int32_t* seg_start = users ? w.seg_start_u : w.seg_start_i;
    const int32_t* perm = users ? w.perm_u : w.perm_i;
    const int32_t* seg_first = users ? w.seg_first_u : w.seg_first_i;
    const int nseg = users ? w.nseg_u[0] : w.nseg_i[0];
    if (wv < B && wv < nseg) {
        const int rj0 = seg_start[wv], rj1 = seg_start[wv + 1], rb0 = seg_first[wv];
        const int j0 = uniform(rj0), j1 = uniform(rj1), b0 = uniform(rb0);
        RowFrag<E> g;
#pragma unroll
        for (int e = 0; e < E; ++e) g.x[e] = 0.f;
        segment_grad_sum<E>(g, j0, j1, b0, perm, w.coef, users ? w.side[3] : w.side[0], D, lane, Lay{0, 0});
#pragma unroll
        for (int e = 0; e < E; ++e) sq = fmaf(g.x[e], g.x[e], sq);
        sq = wave_sum(sq);
    }
    if (lane == 0) red[wib] = sq;
    __syncthreads();
    if (threadIdx.x == 0) w.clip_part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// one workgroup: total norm (fixed-order sum of the partials + the other parameters' share), the clip coefficient
// min(1, max_norm / (norm + 1e-6)) of torch.nn.utils.clip_grad_norm_, and dLoss/dpred scaled by it -- which scales every
// gradient row the backward kernel will form
__global__ __launch_bounds__(1024) void focf_clip_scale_kernel(int B, FocfWs w, int n_part, float max_norm,
                                                               const float* __restrict__ extra_sqnorm,
                                                               float* __restrict__ norm_out) {
    __shared__ float red[16];
    __shared__ float coef_s;
    float a = 0.f;
    for (int q = threadIdx.x; q < n_part; q += 1024) a += w.clip_part[q];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q];
        if (extra_sqnorm) t += extra_sqnorm[0];
        const float norm = sqrtf(t);
        float cc = max_norm / (norm + 1e-6f);
        cc = cc < 1.f ? cc : 1.f;
        coef_s = cc;
        if (norm_out) {
            norm_out[0] = norm;
            norm_out[1] = cc;
        }
    }
    __syncthreads();
    const float cc = coef_s;
    if (cc < 1.f)
        for (int b = threadIdx.x; b < B; b += 1024) w.coef[b] *= cc;
}

// waves [0, sweep waves): the sweep slice when no earlier launch of the step carried it (longest jobs first: up to S
// replayed steps per row, the segment waves exactly one); then one wave per distinct user row, per distinct item row
template <int E>
__global__ __launch_bounds__(256) void focf_backward_adam_kernel(TableV U, TableV I, AdamC c, int B, FocfWs w,
                                                                 SweepSlice sw, int n_sweep_waves) {
    if (blockIdx.x == 0) {   // the extra workgroup, first so that it overlaps with everything else
        const DeferLoss dl = *w.defer;      // left by the forward launch of this batch (workgroup-uniform)
        if (dl.loss_out) {
            focf_finalize_block(w, B, dl.objective, dl.fair_weight, dl.loss_out, dl.n_fair_part);
            if (threadIdx.x == 0) w.defer->loss_out = nullptr;
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    long long wv = (long long)(blockIdx.x - 1) * 4 + (threadIdx.x >> 6);
    if (wv < n_sweep_waves) {
        sweep_slice_wave<E>(U, I, c, sw, wv, lane);
        return;
    }
    wv -= n_sweep_waves;
    // item rows before user rows: an item has more members in the batch than a user (~100 in item-complete batches),
    // so its wave sums more rows -- longest jobs first
    if (wv < B) {
        if (wv < w.nseg_i[0])
            segment_update<E>(I, c, (int)wv, w.seg_start_i, w.seg_row_i, w.perm_i, w.coef, w.side[3], w.side[4],
                              w.side[5], w.side[0], lane, Lay{0, 0}, w.seg_first_i);
        return;
    }
    wv -= B;
    if (wv < B && wv < w.nseg_u[0])
        segment_update<E>(U, c, (int)wv, w.seg_start_u, w.seg_row_u, w.perm_u, w.coef, w.side[0], w.side[1],
                          w.side[2], w.side[3], lane, Lay{0, 0}, w.seg_first_u);
}


// ------------------------------------------------------------------------------------------------
// row-sharded FOCF (multi-GPU): requester-side scoring / gradient rows, owner-side fairness statistics
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void focf_shard_score_kernel(const float* __restrict__ rows_u,
                                                               const float* __restrict__ rows_i,
                                                               const int32_t* __restrict__ slot_u,
                                                               const int32_t* __restrict__ slot_i,
                                                               const float* __restrict__ rating,
                                                               const float* __restrict__ sst, int B, int D,
                                                               float inv_n, float* __restrict__ pred,
                                                               float* __restrict__ coef, float* __restrict__ rec,
                                                               int cap, int slot_stride, int slot_offset,
                                                               float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wib;
    __shared__ float red[4];
    float e2 = 0.f;
    if (b < B) {
        const int su = uniform(slot_u[b]), si = uniform(slot_i[b]);
        float dot = 0.f;
        if (su >= 0 && si >= 0)
            for (int d = lane; d < D; d += 64) dot = fmaf(rows_u[(size_t)su * D + d], rows_i[(size_t)si * D + d], dot);
        dot = wave_sum(dot);
        const float r = rating[b];
        const float er = dot - r;
        e2 = er * er;
        if (lane == 0) {
            pred[b] = dot;
            coef[b] = 2.f * er * inv_n;
            if (rec && si >= 0) {   // item slot (g, k) -> planes of the [G, 3, cap] record buffer
                float* rg = rec + (size_t)(si / slot_stride) * 3 * cap + (si % slot_stride - slot_offset);
                rg[0] = dot;
                rg[cap] = r;
                rg[2 * cap] = sst ? sst[b] : 0.f;
            }
        }
    }
    if (lane == 0) red[wib] = e2;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// out[0] = sum(part[0..n)) in index order (one block); out[1] = *count when given
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int n,
                                                              float* __restrict__ out,
                                                              const int32_t* __restrict__ count) {
    __shared__ float red[4];
    float a = 0.f;
    for (int q = threadIdx.x; q < n; q += 256) a += part[q];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = ((red[0] + red[1]) + red[2]) + red[3];
        if (count) out[1] = (float)count[0];
    }
}

// nonparity on the row-sharded path (focf.py:127-134 on the GLOBAL batch): this rank's share of the two group sums.
// out5 = (sum of squared errors, sum pred | g0, count g0, sum pred | g1, count g1); an all-reduce makes them global.
__global__ __launch_bounds__(1024) void focf_shard_nonparity_sums_kernel(const float* __restrict__ pred,
                                                                         const float* __restrict__ sst, int B,
                                                                         const float* __restrict__ minmax, int mm_count,
                                                                         int mm_stride, const float* __restrict__ sq_part,
                                                                         int n_sq_part, float* __restrict__ out5) {
    __shared__ float red[5][16];
    float smin = minmax[0], smax = minmax[1];
    for (int q = 1; q < mm_count; ++q) {
        smin = fminf(smin, minmax[q * mm_stride]);
        smax = fmaxf(smax, minmax[q * mm_stride + 1]);
    }
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = threadIdx.x; q < n_sq_part; q += 1024) v[0] += sq_part[q];
    for (int b = threadIdx.x; b < B; b += 1024) {
        const float s = sst[b], pr = pred[b];
        if (s == smin) { v[1] += pr; v[2] += 1.f; }
        else if (s == smax) { v[3] += pr; v[4] += 1.f; }   // rows of a third group are ignored, as in the reference
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const float w = wave_sum(v[q]);
        if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = w;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        float a = 0.f;
        for (int w = 0; w < 16; ++w) a += red[threadIdx.x][w];
        out5[threadIdx.x] = a;
    }
}

// ... and, from the all-reduced sums, the fairness part of dLoss/dpred added to coef[b] and the loss
__global__ __launch_bounds__(256) void focf_shard_nonparity_coef_kernel(float* __restrict__ coef,
                                                                        const float* __restrict__ sst, int B,
                                                                        const float* __restrict__ minmax, int mm_count,
                                                                        int mm_stride, const float* __restrict__ g5,
                                                                        float inv_n, float fair_weight,
                                                                        float* __restrict__ loss_out, uint32_t* err) {
    float smin = minmax[0], smax = minmax[1];
    for (int q = 1; q < mm_count; ++q) {
        smin = fminf(smin, minmax[q * mm_stride]);
        smax = fmaxf(smax, minmax[q * mm_stride + 1]);
    }
    const float n0 = g5[2], n1 = g5[4];
    const bool one_group = smin == smax || n1 == 0.f;     // reference: IndexError (sst_unique_value[1]) -- flag it
    const float delta = one_group ? 0.f : g5[1] / n0 - g5[3] / n1;
    const float dl = fminf(fmaxf(delta, -1.f), 1.f) * fair_weight;
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B && !one_group) {
        const float s = sst[b];
        if (s == smin) coef[b] += dl / n0;
        else if (s == smax) coef[b] -= dl / n1;
    }
    if (b == 0) {
        const float mse = g5[0] * inv_n, fair = one_group ? 0.f : smooth_l1(delta);
        loss_out[0] = mse + fair_weight * fair;
        loss_out[1] = mse;
        loss_out[2] = fair;
        if (one_group && err) atomicOr(err, FR_DEV_ERR_SST_GROUPS);
    }
}

__global__ __launch_bounds__(256) void focf_shard_grads_kernel(const float* __restrict__ rows_u,
                                                               const float* __restrict__ rows_i,
                                                               const int32_t* __restrict__ slot_u,
                                                               const int32_t* __restrict__ slot_i,
                                                               const float* __restrict__ coef,
                                                               const float* __restrict__ coef_slots,
                                                               int G, float inv_n, float fair_weight,
                                                               float* __restrict__ loss_out, int cap,
                                                               int slot_stride, int slot_offset, int B, int D,
                                                               float* __restrict__ grad_u, float* __restrict__ grad_i) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    // K, fairness sum and squared-error sum of the GLOBAL batch from the tails of the received reply chunks, owners
    // in rank order (so every rank and every wave gets the same bits); wave 0 also reports the loss
    float K = 1.f;
    if (coef_slots) {
        float fs = 0.f, sq = 0.f;
        K = 0.f;
        for (int g = 0; g < G; ++g) {
            const float* t = coef_slots + (size_t)g * (cap + FR_SHARD_TAIL) + cap;
            K += t[0];
            fs += t[1];
            sq += t[2];
        }
        if (b == 0 && lane == 0 && loss_out) {
            const float mse = sq * inv_n, fair = fs / K;
            loss_out[0] = mse + fair_weight * fair;
            loss_out[1] = mse;
            loss_out[2] = fair;
        }
    }
    if (b >= B) return;
    const int su = uniform(slot_u[b]), si = uniform(slot_i[b]);
    if (su < 0 || si < 0) return;
    float c = coef[b];
    if (coef_slots)   // the fairness part of the reply still lacks 1/K
        c += coef_slots[(size_t)(si / slot_stride) * (cap + FR_SHARD_TAIL) + (si % slot_stride - slot_offset)] / K;
    for (int d = lane; d < D; d += 64) {
        const float ue = rows_u[(size_t)su * D + d], ie = rows_i[(size_t)si * D + d];
        grad_u[(size_t)su * D + d] = c * ie;   // product rounded once: the owner adds the rows of duplicates
        grad_i[(size_t)si * D + d] = c * ue;
    }
}

}  // namespace fr

using namespace fr;

extern "C" size_t fr_focf_workspace_bytes(int64_t B, int32_t dim) {
    if (B < 0 || dim < 1) return 0;
    return focf_layout(nullptr, B, dim).bytes;
}

static int focf_launch_sort(const FocfWs& w, const int64_t* user, const int64_t* item, const float* sst, int64_t B,
                            int64_t n_users, int64_t n_items, bool want_minmax, uint32_t* err_flag, hipStream_t stream) {
    SortJob ju{user, n_users, w.perm_u, w.seg_start_u, w.seg_row_u, nullptr, w.nseg_u, nullptr, nullptr};
    SortJob ji{item, n_items, w.perm_i, w.seg_start_i, w.seg_row_i, nullptr, w.nseg_i, want_minmax ? sst : nullptr,
               w.sst_minmax};
    ju.seg_first = w.seg_first_u;
    ji.seg_first = w.seg_first_i;
    return launch_sort(ju, &ji, B, err_flag, stream);
}

extern "C" int fr_focf_prepare_many(const fr_focf_batch* batches, int32_t n, int64_t n_users, int64_t n_items,
                                    int32_t dim, uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(batches && n >= 1 && 2 * n <= FR_SORT_JOBS && dim >= 1, "fr_focf_prepare_many: 1..%d batches",
                 FR_SORT_JOBS / 2);
    SortJobList jobs{};
    for (int q = 0; q < n; ++q) {
        const fr_focf_batch& b = batches[q];
        FR_CHECK_ARG(b.user && b.item && b.ws && b.B >= 1 && b.B <= FR_SORT_MAX, "fr_focf_prepare_many: batch %d", q);
        FocfWs w = focf_layout(b.ws, b.B, dim);
        FR_CHECK_ARG(b.ws_bytes >= w.bytes, "fr_focf_prepare_many: workspace %zu < %zu bytes", b.ws_bytes, w.bytes);
        SortJob ju{b.user, n_users, w.perm_u, w.seg_start_u, w.seg_row_u, nullptr, w.nseg_u, nullptr, nullptr};
        SortJob ji{b.item, n_items, w.perm_i, w.seg_start_i, w.seg_row_i, nullptr, w.nseg_i, b.sst, w.sst_minmax};
        ju.seg_first = w.seg_first_u;
        ji.seg_first = w.seg_first_i;
        jobs.j[2 * q] = ju;
        jobs.j[2 * q + 1] = ji;
        jobs.M[2 * q] = jobs.M[2 * q + 1] = (int)b.B;
    }
    jobs.n = 2 * n;
    return launch_sort_many(jobs, n_users > n_items ? n_users : n_items, err_flag, (hipStream_t)stream_);
}

extern "C" int fr_focf_prepare(const int64_t* user, const int64_t* item, const float* sst, int64_t B, int64_t n_users,
                               int64_t n_items, int32_t dim, void* ws, size_t ws_bytes, uint32_t* err_flag,
                               void* stream_) {
    fr_focf_batch b{user, item, sst, B, ws, ws_bytes};
    return fr_focf_prepare_many(&b, 1, n_users, n_items, dim, err_flag, stream_);
}

// The training gather of an item-complete batch as a launch of its own (shared with focf_runs.hip: the first of the two
// launches of fr_focf_step_runs): rows caught up and parked at their batch positions, scores, the MSE part of dLoss/dpred.
namespace fr {
int focf_launch_gather_runs(const fr_table* U, const fr_table* I, const AdamC& c, const int64_t* user, const int64_t* item,
                            const float* rating, const float* sst, int64_t B, const FocfWs& w, uint32_t* err_flag,
                            hipStream_t stream, const SweepSlice& sw, long long sweep_waves) {
    const TableV Uv = view(U), Iv = view(I);
    const DeferLoss dl{nullptr, 0, 0.f, 0};
    const SortedPark sp{w.pos_i, w.info, sst, w.task_rec, w.task_info, w.mse_e};
    ProfScope prof(K_FOCF_GATHER, stream);
    const unsigned blocks = (unsigned)(w.n_gather_blocks + (sweep_waves + 3) / 4);
    FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_gather_sweep_kernel<E>), dim3(blocks), dim3(GATHER_THREADS), 0, stream, Uv, Iv, c,
                                    user, item, rating, (int)B, U->step - 1, I->step - 1, w, err_flag, dl, sw, sweep_waves, sp));
    FR_CHECK_LAUNCH();
    return FR_OK;
}
}  // namespace fr

extern "C" int fr_focf_forward(const fr_table* U, const fr_table* I, const fr_adam* adam, const int64_t* user,
                               const int64_t* item, const float* rating, const float* sst, int64_t B,
                               int32_t objective, float fair_weight, int32_t flags, void* ws, size_t ws_bytes,
                               float* loss_out, float* pred_out, uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = check_table(U, "fr_focf_forward(U)")) || (rc = check_table(I, "fr_focf_forward(I)")) ||
        (rc = check_adam(adam, "fr_focf_forward")))
        return rc;
    FR_CHECK_ARG(U->dim == I->dim, "fr_focf_forward: user dim %d != item dim %d", U->dim, I->dim);
    FR_CHECK_ARG(user && item && rating && loss_out && ws, "fr_focf_forward: null pointer");
    FR_CHECK_ARG(objective >= FR_FOCF_NONE && objective <= FR_FOCF_NONPARITY, "fr_focf_forward: objective %d",
                 objective);
    FR_CHECK_ARG(objective == FR_FOCF_NONE || sst, "fr_focf_forward: sst column required for a fairness objective");
    FR_CHECK_ARG(B >= 1 && B <= FR_SORT_MAX, "fr_focf_forward: batch size %lld not in 1..%d", (long long)B,
                 FR_SORT_MAX);
    FR_CHECK_ARG(U->step >= 1 && I->step >= 1, "fr_focf_forward: table.step must be the step being applied (>=1)");
    FocfWs w = focf_layout(ws, B, U->dim);
    FR_CHECK_ARG(ws_bytes >= w.bytes, "fr_focf_forward: workspace %zu < %zu bytes", ws_bytes, w.bytes);

    // The sort only reads the id columns, the gather only reads the tables: unless the caller already ran
    // fr_focf_prepare for this batch (software-pipelined one step ahead), run them side by side (fork/join through
    // events; under stream capture this becomes two parallel graph branches).
    const bool prepared = (flags & FR_FOCF_PREPARED) != 0;
    SideStream* ss = side_stream();
    const bool overlap = !prepared && ss != nullptr && !prof_on();
    if (!prepared) {
        if (overlap) {
            FR_CHECK_HIP(hipEventRecord(ss->fork, stream));
            FR_CHECK_HIP(hipStreamWaitEvent(ss->stream, ss->fork, 0));
            if ((rc = focf_launch_sort(w, user, item, sst, B, U->n_rows, I->n_rows, objective != FR_FOCF_NONE, err_flag,
                                       ss->stream)))
                return rc;
            FR_CHECK_HIP(hipEventRecord(ss->join, ss->stream));
        } else if ((rc = focf_launch_sort(w, user, item, sst, B, U->n_rows, I->n_rows, objective != FR_FOCF_NONE,
                                          err_flag, stream))) {
            return rc;
        }
    }

    const AdamC c = make_adamc(adam);
    const TableV Uv = view(U), Iv = view(I);
    const bool defer = (flags & FR_FOCF_DEFER_LOSS) != 0;
    const bool runs = (flags & FR_FOCF_ITEM_RUNS) != 0;
    // loss reduction left to the backward launch: the record travels in the workspace (no host state)
    const int n_fair_part = (objective >= FR_FOCF_VALUE && objective <= FR_FOCF_OVER && runs)
                          ? w.n_fair_blocks * (FAIR_GROUP_RUNS / FAIR_GROUP) : w.n_fair_blocks;
    const DeferLoss dl{defer ? loss_out : nullptr, (int)objective, fair_weight, n_fair_part};
    {
        ProfScope prof(K_FOCF_GATHER, stream);
        if (flags & FR_FOCF_ITEM_RUNS) {
            FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_gather_kernel<E, true, true>), dim3(w.n_gather_blocks), dim3(GATHER_THREADS), 0, stream, Uv, Iv, c, user, item, rating, (int)B, U->step - 1, I->step - 1, w, 0.f, (float*)nullptr, err_flag, dl));
        } else {
            FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_gather_kernel<E, true, false>), dim3(w.n_gather_blocks), dim3(GATHER_THREADS), 0, stream, Uv, Iv, c, user, item, rating, (int)B, U->step - 1, I->step - 1, w, 0.f, (float*)nullptr, err_flag, dl));
        }
    }
    FR_CHECK_LAUNCH();
    if (overlap) FR_CHECK_HIP(hipStreamWaitEvent(stream, ss->join, 0));
    if (objective == FR_FOCF_NONPARITY) {
        {
            ProfScope prof(K_FOCF_NONPARITY, stream);
            FR_LAUNCH(prof, focf_nonparity_kernel, dim3(1), dim3(1024), 0, stream, w, sst, (int)B, fair_weight,
                               err_flag);
        }
        FR_CHECK_LAUNCH();
    } else if (objective != FR_FOCF_NONE) {
        {
            ProfScope prof(K_FOCF_FAIR, stream);
            FairArgs fa{w.perm_i, w.seg_start_i, w.nseg_i, w.seg_first_i, w.sst_minmax, 1, 0, w.pred, rating, sst, Lay{0, 0}, w.coef, Lay{0, 0}, w.fair_part, 1,
                        defer ? nullptr : w.ticket, w.mse_part, w.n_gather_blocks, (int)B, loss_out};
            if (flags & FR_FOCF_ITEM_RUNS) {   // few items with many members each: a whole wave per item segment
                FR_LAUNCH(prof, focf_fair_kernel<FAIR_GROUP_RUNS>, dim3(n_fair_part), dim3(FAIR_THREADS), 0, stream, fa,
                                   objective, fair_weight, 0, err_flag);
            } else {
                FR_LAUNCH(prof, focf_fair_kernel<FAIR_GROUP>, dim3(w.n_fair_blocks), dim3(FAIR_THREADS), 0, stream, fa,
                                   objective, fair_weight, 0, err_flag);
            }
        }
        FR_CHECK_LAUNCH();
    }
    if (!defer && (objective == FR_FOCF_NONE || objective == FR_FOCF_NONPARITY)) {
        ProfScope prof(K_FOCF_FINALIZE, stream);
        FR_LAUNCH(prof, focf_finalize_kernel, dim3(1), dim3(256), 0, stream, w, (int)B, objective, fair_weight,
                           loss_out);
        FR_CHECK_LAUNCH();
    }
    if (pred_out) FR_CHECK_HIP(hipMemcpyAsync(pred_out, w.pred, (size_t)B * 4, hipMemcpyDeviceToDevice, stream));
    return FR_OK;
}

extern "C" int fr_focf_clip_grad_norm(const fr_table* U, const fr_table* I, int64_t B, float max_norm,
                                      const float* extra_sqnorm, float* norm_out, void* ws, size_t ws_bytes,
                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = check_table(U, "fr_focf_clip_grad_norm(U)")) || (rc = check_table(I, "fr_focf_clip_grad_norm(I)")))
        return rc;
    FR_CHECK_ARG(U->dim == I->dim && ws && B >= 1 && B <= FR_SORT_MAX && max_norm > 0.f,
                 "fr_focf_clip_grad_norm: bad argument");
    FocfWs w = focf_layout(ws, B, U->dim);
    FR_CHECK_ARG(ws_bytes >= w.bytes, "fr_focf_clip_grad_norm: workspace %zu < %zu bytes", ws_bytes, w.bytes);
    const unsigned blocks = (unsigned)((2 * B + 3) / 4);
    FR_DISPATCH_E(U->dim, hipLaunchKernelGGL((focf_grad_sqnorm_kernel<E>), dim3(blocks), dim3(256), 0, stream, (int)B, (int)U->dim, w));
    FR_CHECK_LAUNCH();
    hipLaunchKernelGGL(focf_clip_scale_kernel, dim3(1), dim3(1024), 0, stream, (int)B, w, (int)blocks, max_norm,
                       extra_sqnorm, norm_out);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_focf_backward_adam(const fr_table* U, const fr_table* I, const fr_adam* adam, int64_t B,
                                     int32_t sweep_period, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = check_table(U, "fr_focf_backward_adam(U)")) || (rc = check_table(I, "fr_focf_backward_adam(I)")) ||
        (rc = check_adam(adam, "fr_focf_backward_adam")))
        return rc;
    FR_CHECK_ARG(U->dim == I->dim && ws && B >= 1 && B <= FR_SORT_MAX, "fr_focf_backward_adam: bad argument");
    FocfWs w = focf_layout(ws, B, U->dim);
    FR_CHECK_ARG(ws_bytes >= w.bytes, "fr_focf_backward_adam: workspace %zu < %zu bytes", ws_bytes, w.bytes);
    const AdamC c = make_adamc(adam);
    const TableV Uv = view(U), Iv = view(I);
    // the sweep slice of this step rides in this launch (rows outside the batch: the gather kernel stamped the batch)
    SweepSlice sw{};
    long long sweep_waves = 0;
    if (sweep_period > 0) {
        FR_CHECK_ARG(U->step == I->step, "fr_focf_backward_adam: the tables' step counters differ");
        sw = make_sweep_slice(U, I, sweep_period);
        sweep_waves = sweep_slice_waves(sw);
    }
    {
        ProfScope prof(K_FOCF_BWD_ADAM, stream);
        FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_backward_adam_kernel<E>), dim3((unsigned)((sweep_waves + 2 * B + 3) / 4) + 1u), dim3(256), 0, stream, Uv, Iv, c, (int)B, w, sw, (int)sweep_waves));
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_focf_predict(const fr_table* U, const fr_table* I, const fr_adam* adam, const int64_t* user,
                               const int64_t* item, int64_t B, float max_rating, float* out, uint32_t* err_flag,
                               void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = check_table(U, "fr_focf_predict(U)")) || (rc = check_table(I, "fr_focf_predict(I)")) ||
        (rc = check_adam(adam, "fr_focf_predict")))
        return rc;
    FR_CHECK_ARG(U->dim == I->dim && user && item && out && B >= 0, "fr_focf_predict: bad argument");
    if (B == 0) return FR_OK;
    const AdamC c = make_adamc(adam);
    const TableV Uv = view(U), Iv = view(I);
    FocfWs w{};
    const unsigned blocks = (unsigned)((B * WAVE + GATHER_THREADS - 1) / GATHER_THREADS);
    {
        ProfScope prof(K_FOCF_GATHER, stream);
        FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_gather_kernel<E, false, false>), dim3(blocks), dim3(GATHER_THREADS), 0, stream, Uv, Iv, c, user, item, (const float*)nullptr, (int)B, U->step,
                                                  I->step, w, max_rating, out, err_flag, DeferLoss{}));
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}


extern "C" int fr_focf_shard_score(const float* rows_u, const float* rows_i, const int32_t* slot_u,
                                   const int32_t* slot_i, const float* rating, const float* sst, int64_t B,
                                   int32_t dim, int64_t n_global, float* pred, float* coef, float* rec, int32_t cap,
                                   int32_t slot_stride, int32_t slot_offset, float* sq_err_sum, float* scratch,
                                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(rows_u && rows_i && slot_u && slot_i && rating && pred && coef && scratch && B >= 1 &&
                     dim >= 1 && n_global >= B && cap >= 1 && slot_offset >= 0 && slot_stride >= slot_offset + cap,
                 "fr_focf_shard_score: bad argument");
    const int blocks = (int)((B + 3) / 4);
    {
        ProfScope prof(K_FOCF_SHARD_SCORE, stream);
        FR_LAUNCH(prof, focf_shard_score_kernel, dim3(blocks), dim3(256), 0, stream, rows_u, rows_i, slot_u, slot_i,
                           rating, sst, (int)B, (int)dim, 1.f / (float)n_global, pred, coef, rec, (int)cap,
                           (int)slot_stride, (int)slot_offset, scratch);
    }
    FR_CHECK_LAUNCH();
    if (sq_err_sum) {   // NULL: the caller hands the (B+3)/4 partials in `scratch` to fr_focf_shard_fair instead
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, stream, scratch, blocks, sq_err_sum,
                           (const int32_t*)nullptr);
        FR_CHECK_LAUNCH();
    }
    return FR_OK;
}

extern "C" int fr_focf_shard_fair(void* item_ws, size_t ws_bytes, int64_t n_slots, int32_t dim, const float* rec,
                                  int32_t cap, const float* minmax, int32_t mm_count, int32_t mm_stride,
                                  int32_t objective, float fair_weight, float* reply, const float* sq_part,
                                  int32_t n_sq_part, float* scratch, uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(item_ws && rec && minmax && reply && sq_part && n_sq_part >= 0 && scratch && n_slots >= 1 && cap >= 1 &&
                     n_slots % cap == 0 && n_slots / cap <= 64 && mm_count >= 1, "fr_focf_shard_fair: bad argument");
    FR_CHECK_ARG(objective >= FR_FOCF_VALUE && objective <= FR_FOCF_OVER,
                 "fr_focf_shard_fair: objective %d has no per-item statistics", objective);
    TableWs tw = table_layout(item_ws, n_slots, dim);
    FR_CHECK_ARG(ws_bytes >= tw.bytes, "fr_focf_shard_fair: workspace too small");
    {
        int rc = side_join(item_ws, stream);
        if (rc) return rc;
    }
    const int blocks = (int)((n_slots * FAIR_GROUP + FAIR_THREADS - 1) / FAIR_THREADS);
    {
        ProfScope prof(K_FOCF_FAIR, stream);
        // records of slot (g, k): planes rec[g][0..2][k]; padding slots are in no segment, so never read or written.
        // scratch[0] is the arrival ticket (zero between launches), partials follow; the last block to arrive writes
        // the tails (K_owner, fair_owner, sum of the sq_part partials) of all destination chunks
        FairArgs fa{tw.perm, tw.seg_start, tw.nseg, tw.seg_first, minmax, (int)mm_count, (int)mm_stride, rec, rec + cap, rec + 2 * cap,
                    Lay{cap, 3 * cap}, reply, Lay{cap, cap + FR_SHARD_TAIL}, scratch + 16, 0,
                    reinterpret_cast<unsigned int*>(scratch), sq_part, (int)n_sq_part, 0, nullptr,
                    reply + cap, (int)(n_slots / cap), cap + FR_SHARD_TAIL};
        FR_LAUNCH(prof, focf_fair_kernel<FAIR_GROUP>, dim3(blocks), dim3(FAIR_THREADS), 0, stream, fa, objective,
                           fair_weight, 1, err_flag);
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_focf_shard_nonparity_sums(const float* pred, const float* sst, int64_t B, const float* minmax,
                                            int32_t mm_count, int32_t mm_stride, const float* sq_part, int32_t n_sq_part,
                                            float* out5, void* stream_) {
    FR_CHECK_ARG(pred && sst && minmax && sq_part && out5 && B >= 1 && mm_count >= 1 && n_sq_part >= 0,
                 "fr_focf_shard_nonparity_sums: bad argument");
    hipLaunchKernelGGL(focf_shard_nonparity_sums_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream_, pred, sst, (int)B,
                       minmax, (int)mm_count, (int)mm_stride, sq_part, (int)n_sq_part, out5);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_focf_shard_nonparity_coef(float* coef, const float* sst, int64_t B, const float* minmax,
                                            int32_t mm_count, int32_t mm_stride, const float* global5, int64_t n_global,
                                            float fair_weight, float* loss_out, uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(coef && sst && minmax && global5 && loss_out && B >= 1 && mm_count >= 1 && n_global >= 1,
                 "fr_focf_shard_nonparity_coef: bad argument");
    hipLaunchKernelGGL(focf_shard_nonparity_coef_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream_, coef, sst, (int)B, minmax, (int)mm_count, (int)mm_stride, global5,
                       1.f / (float)n_global, fair_weight, loss_out, err_flag);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_focf_shard_grads(const float* rows_u, const float* rows_i, const int32_t* slot_u,
                                   const int32_t* slot_i, const float* coef, const float* coef_slots, int32_t G,
                                   int64_t n_global, float fair_weight, float* loss_out, int32_t cap,
                                   int32_t slot_stride, int32_t slot_offset, int64_t B, int32_t dim,
                                   float* grad_u_slots, float* grad_i_slots, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(rows_u && rows_i && slot_u && slot_i && coef && grad_u_slots && grad_i_slots && B >= 1 && dim >= 1 &&
                     cap >= 1 && slot_offset >= 0 && slot_stride >= slot_offset + cap && G >= 1 && n_global >= 1,
                 "fr_focf_shard_grads: bad argument");
    ProfScope prof(K_FOCF_SHARD_GRADS, stream);
    FR_LAUNCH(prof, focf_shard_grads_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, rows_u, rows_i,
                       slot_u, slot_i, coef, coef_slots, (int)G, 1.f / (float)n_global, fair_weight, loss_out, (int)cap,
                       (int)slot_stride, (int)slot_offset, (int)B, (int)dim, grad_u_slots,
                       grad_i_slots);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
