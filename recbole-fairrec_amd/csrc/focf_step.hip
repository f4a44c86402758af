// FOCF training step as ONE launch: gather + lazy-Adam replay + dot + fairness term + backward + Adam + sweeper.
//
// Reference being replaced (stock PyTorch ops called from Python, one optimizer step):
//   FOCF.calculate_loss     focf.py:152-169   (= forward :136-143, MSELoss :158, get_item_ratings :75-91, *_unfairness :93-125)
//   loss.backward()         trainer.py:193    dense embedding_dense_backward of both tables
//   optimizer.step()        trainer.py:196    dense torch.optim.Adam over both tables
//
// Why one launch: the three-launch chain of focf.hip (gather -> fair -> backward_adam) parks the caught-up (p, m, v) of
// every gathered row in a [B, D] x 6 side buffer between the launches (12.6 MB written and read back at B = 8192, D = 64)
// and pays its dependent load levels three times.  Here a wave keeps the two rows of its interaction in registers from the
// gather to the Adam write-back:
//   * an interaction whose user AND item occur once in the batch (97 % / 92 % of the rows for uniform pairs at the
//     BASELINE sizes) needs nobody else: for a one-member item the fairness statistics of focf.py:75-91 are a function of
//     its own (pred, rating, group) and of two batch-wide values known before the launch (K = number of distinct items
//     and the two sensitive values present; both come out of the look-ahead index sort);
//   * rows shared by several interactions are finished by the LAST of their waves to arrive (an arrival counter per
//     segment, no waiting, hence no residency requirement and no deadlock): the item level forms the per-item statistics,
//     dLoss/dpred of every member and the item row's update; the user level sums a user's gradient rows.  Members hand
//     over through write-through (sc1) stores drained before the counter add and are read with sc1 loads by the wave whose
//     add returned last (MI355X_MICROARCH.md, "Valid forms"; the same pattern as the ticket of focf_fair_kernel).
//     All sums run in ascending batch position, so the result does not depend on who arrives last.
// The sweeper slice of the step (bounded staleness) rides in the same launch: its rows are told from the
// batch's rows by stamps that the look-ahead sort wrote (fr_focf_prepare_step), so both kinds of wave start at once and the
// sweeper's VALU work hides the interaction waves' two dependent load levels.
#include "common.hpp"
#include "kernels.hpp"
#include "table.hpp"
#include "focf_ws.hpp"

namespace fr {

// loss of an EARLIER step still to be reduced (its per-interaction squared errors and per-item terms are complete once
// its launch has ended): one extra workgroup of the next launch, or fr_focf_step_finish, does it
struct PrevLoss {
    const float* mse_e;
    const float* term;
    const int32_t* nseg_i;
    int B, objective;
    float fair_weight;
    float* loss_out;   // [3] loss, mse, fair; nullptr = nothing to reduce
    float* acc;        // optional [3]: += the three values (a running epoch total kept on the device)
};

struct StepArgs {
    TableV U, I;
    AdamC c;
    const int64_t *user, *item;
    const float *rating, *sst;
    int B, objective;
    float fair_weight;
    FocfWs w;
    SweepSlice sw;
    int n_sweep_blocks, n_inter_blocks;
    uint32_t* err;
    PrevLoss prev;
};

__device__ __forceinline__ float ld_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every store of this wave has left for memory (the storing wave's part of a hand-off; inline asm so that no compiler pass
// drops it)
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int E>
__device__ __forceinline__ void load_row_sc1(RowFrag<E>& f, const float* base, int D, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane + 64 * e;
        f.x[e] = d < D ? ld_sc1(base + d) : 0.f;
    }
}

template <int E>
__device__ __forceinline__ void store_row_sc1(const RowFrag<E>& f, float* base, int D, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane + 64 * e;
        if (d < D) st_sc1(base + d, f.x[e]);
    }
}

// arrival at a segment's counter; true for the wave whose add came last (it then owns the segment's work)
__device__ __forceinline__ bool arrive_last(unsigned int* cnt, int n, int lane) {
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = (unsigned)uniform((int)t);
    if (t != (unsigned)(n - 1)) return false;
    if (lane == 0) *cnt = 0u;      // ready for the next use of the workspace
    return true;
}

// Adam step `T.step` with data gradient g on a caught-up row held in registers; the row is written back once
template <int E>
__device__ __forceinline__ void adam_write(const TableV& T, const AdamC& c, int row, RowFrag<E>& p, RowFrag<E>& m,
                                           RowFrag<E>& v, const RowFrag<E>& g, float2 s, int lane) {
    const int D = T.D;
#pragma unroll
    for (int e = 0; e < E; ++e) adam_elem(p.x[e], m.x[e], v.x[e], g.x[e], s.x, s.y, c);
    store_row<E>(p, T.p + (size_t)row * D, D, lane);
    store_row<E>(m, T.m + (size_t)row * D, D, lane);
    store_row<E>(v, T.v + (size_t)row * D, D, lane);
    if (lane == 0) T.last[row] = T.step;
}

// g = sum over the members [j0, j0 + n) of a segment, in ascending batch position, of coef[b] * other[b, :] -- the product
// rounded, then added (embedding_dense_backward's accumulation order), as segment_grad_sum of table.hpp, but on values
// other waves of this launch handed over: sc1 loads throughout
template <int E>
__device__ __forceinline__ void handed_grad_sum(RowFrag<E>& g, int j0, int n, const int32_t* perm, const float* coef,
                                                const float* other, int D, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) g.x[e] = 0.f;
    constexpr int UN = 4;
    for (int jb = 0; jb < n; jb += 64) {
        const int cnt = min(64, n - jb);
        int my_b = 0;
        float my_c = 0.f;
        if (lane < cnt) {
            my_b = perm[j0 + jb + lane];
            my_c = ld_sc1(coef + my_b);
        }
        for (int t0 = 0; t0 < cnt; t0 += UN) {
            RowFrag<E> o[UN];
            float cb[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const int t = t0 + q < cnt ? t0 + q : cnt - 1;     // tail: re-read the last member, weight 0
                const int b = __builtin_amdgcn_readlane(my_b, t);
                cb[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_c), t));
                load_row_sc1<E>(o[q], other + (size_t)b * D, D, lane);
            }
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int q = 0; q < UN; ++q) {
                    if (t0 + q < cnt) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            float prod = cb[q] * o[q].x[e];
                            g.x[e] = g.x[e] + prod;
                        }
                    }
                }
            }
        }
    }
}

// The last wave to arrive at a user segment: sum the members' gradient rows coef[b] * (item row of b before its update),
// one Adam step on the user's caught-up row (parked by its first member), write back.
template <int E>
__device__ __forceinline__ void user_finish(const StepArgs& a, int j0u, int nu, float2 s, int lane) {
    const int D = a.U.D;
    const int c0 = uniform(a.w.perm_u[j0u]);
    long long ul = a.user[c0];
    if (ul < 0 || ul >= a.U.n_rows) ul = 0;     // reported by the member's own wave
    const int ur = uniform((int)ul);
    RowFrag<E> p, m, v, g;
    load_row_sc1<E>(p, a.w.side[0] + (size_t)c0 * D, D, lane);
    load_row_sc1<E>(m, a.w.side[1] + (size_t)c0 * D, D, lane);
    load_row_sc1<E>(v, a.w.side[2] + (size_t)c0 * D, D, lane);
    handed_grad_sum<E>(g, j0u, nu, a.w.perm_u, a.w.coef, a.w.side[3], D, lane);
    adam_write<E>(a.U, a.c, ur, p, m, v, g, s, lane);
}

// One interaction of the batch.
template <int E>
__device__ __forceinline__ void step_interaction(const StepArgs& a, int b, int lane) {
    const TableV& U = a.U;
    const TableV& I = a.I;
    const AdamC& c = a.c;
    const int D = U.D;
    const bool fair = a.objective != FR_FOCF_NONE;
    // ---- level 1: everything addressed by the batch position
    long long ul = a.user[b], il = a.item[b];
    const float r = a.rating[b];
    const float s = fair ? a.sst[b] : 0.f;
    const int2 riu = a.w.info_u[b], rii = a.w.info_i[b];
    const int rK = a.w.nseg_i[0];
    const float smin = fair ? a.w.sst_minmax[0] : 0.f, smax = fair ? a.w.sst_minmax[1] : 0.f;
    if (ul < 0 || ul >= U.n_rows || il < 0 || il >= I.n_rows) {
        if (lane == 0 && a.err) atomicOr(a.err, FR_DEV_ERR_INDEX_RANGE);
        ul = ul < 0 || ul >= U.n_rows ? 0 : ul;      // the sort clamped the same way
        il = il < 0 || il >= I.n_rows ? 0 : il;
    }
    const int ur = uniform((int)ul), ir = uniform((int)il);
    const int iux = uniform(riu.x), iix = uniform(rii.x), seg_u = uniform(riu.y), seg_i = uniform(rii.y);
    const int nu = iux >> 16, ni = iix >> 16, j0u = iux & 0xffff, j0i = iix & 0xffff;
    const float K = (float)uniform(rK);
    // ---- level 2: the rows and their `last` stamps, requested together
    const int lu = U.last[ur], li = I.last[ir];
    RowFrag<E> pu, mu, vu, pi, mi, vi;
    load_row<E>(pu, U.p + (size_t)ur * D, D, lane);
    load_row<E>(pi, I.p + (size_t)ir * D, D, lane);
    load_row<E>(mu, U.m + (size_t)ur * D, D, lane);
    load_row<E>(vu, U.v + (size_t)ur * D, D, lane);
    load_row<E>(mi, I.m + (size_t)ir * D, D, lane);
    load_row<E>(vi, I.v + (size_t)ir * D, D, lane);
    const int t0u = uniform(lu), t0i = uniform(li);
    // replay the optimizer steps each row missed (zero data gradient, weight decay only): first the steps only the staler
    // row missed, then the common tail on both rows interleaved
    const int upto_u = U.step - 1, upto_i = I.step - 1;
    if (upto_u == upto_i) {
        if (t0u < t0i) replay<E>(pu, mu, vu, t0u, t0i, c, lane);
        else if (t0i < t0u) replay<E>(pi, mi, vi, t0i, t0u, c, lane);
        replay2<E>(pu, mu, vu, pi, mi, vi, t0u > t0i ? t0u : t0i, upto_u, c, lane);
    } else {
        replay<E>(pu, mu, vu, t0u, upto_u, c, lane);
        replay<E>(pi, mi, vi, t0i, upto_i, c, lane);
    }
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) dot = fmaf(pu.x[e], pi.x[e], dot);
    dot = wave_sum(dot);
    const float er = dot - r;
    if (lane == 0) a.w.mse_e[b] = er * er;
    const float cm = 2.f * er / (float)a.B;        // d mean((pred - r)^2) / d pred
    const float2 su = step_scalars(c, U.step), si = step_scalars(c, I.step);

    // dLoss/dpred of an interaction whose item has no other member in the batch: its per-item statistics are its own
    float coef = cm;
    if (ni == 1 && fair) {
        const bool in0 = s == smin;
        if (s != smin && s != smax && lane == 0 && a.err) atomicOr(a.err, FR_DEV_ERR_SST_GROUPS);
        float term, g0, g1;
        focf_fair_eval(a.objective, a.fair_weight, K, in0 ? dot : 0.f, in0 ? 0.f : dot, in0 ? r : 0.f, in0 ? 0.f : r,
                       in0 ? 1.f : 0.f, in0 ? 0.f : 1.f, term, g0, g1);
        coef = cm + (in0 ? g0 : g1);
        if (lane == 0) a.w.term[seg_i] = term;
    }

    if (ni == 1 && nu == 1) {      // ---- nobody else touches either row: finish here
        RowFrag<E> gu, gi;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            gu.x[e] = coef * pi.x[e];
            gi.x[e] = coef * pu.x[e];
        }
        adam_write<E>(U, c, ur, pu, mu, vu, gu, su, lane);
        adam_write<E>(I, c, ir, pi, mi, vi, gi, si, lane);
        return;
    }

    // ---- shared rows: hand over, then whoever arrives last at a segment finishes it
    const size_t so = (size_t)b * D;
    store_row_sc1<E>(pu, a.w.side[0] + so, D, lane);
    store_row_sc1<E>(mu, a.w.side[1] + so, D, lane);
    store_row_sc1<E>(vu, a.w.side[2] + so, D, lane);
    if (nu > 1) store_row_sc1<E>(pi, a.w.side[3] + so, D, lane);     // the item row BEFORE its update: users' gradients
    RowFrag<E> pi0 = pi;
    if (ni == 1) {
        // item level is this wave alone; the user has other members
        RowFrag<E> gi;
#pragma unroll
        for (int e = 0; e < E; ++e) gi.x[e] = coef * pu.x[e];
        adam_write<E>(I, c, ir, pi, mi, vi, gi, si, lane);
        if (lane == 0) st_sc1(a.w.coef + b, coef);
        drain_stores();
        if (arrive_last(a.w.cnt_u + seg_u, nu, lane)) user_finish<E>(a, j0u, nu, su, lane);
        return;
    }
    if (lane == 0) st_sc1(a.w.pred + b, dot);
    drain_stores();
    if (!arrive_last(a.w.cnt_i + seg_i, ni, lane)) return;

    // ---- item level, last arriver: statistics of the item over its members (the order of focf_fair_kernel: 16 lanes,
    // members strided over them, butterfly), dLoss/dpred of every member, the item row's gradient and update
    float sp0 = 0.f, sp1 = 0.f, st0 = 0.f, st1 = 0.f, n0 = 0.f, n1 = 0.f;
    if (fair) {
        bool bad = false;
        if (lane < FAIR_GROUP) {
            for (int j = j0i + lane; j < j0i + ni; j += FAIR_GROUP) {
                const int bq = a.w.perm_i[j];
                const float sq = a.sst[bq], pr = ld_sc1(a.w.pred + bq), rq = a.rating[bq];
                bad |= (sq != smin && sq != smax);
                if (sq == smin) {
                    sp0 += pr; st0 += rq; n0 += 1.f;
                } else {
                    sp1 += pr; st1 += rq; n1 += 1.f;
                }
            }
        }
        if (bad && a.err) atomicOr(a.err, FR_DEV_ERR_SST_GROUPS);
        sp0 = group_sum<FAIR_GROUP>(sp0); sp1 = group_sum<FAIR_GROUP>(sp1);
        st0 = group_sum<FAIR_GROUP>(st0); st1 = group_sum<FAIR_GROUP>(st1);
        n0 = group_sum<FAIR_GROUP>(n0);   n1 = group_sum<FAIR_GROUP>(n1);
        sp0 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, sp0)));
        sp1 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, sp1)));
        st0 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, st0)));
        st1 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, st1)));
        n0 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, n0)));
        n1 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, n1)));
    }
    float term = 0.f, g0 = 0.f, g1 = 0.f;
    if (fair) {
        focf_fair_eval(a.objective, a.fair_weight, K, sp0, sp1, st0, st1, n0, n1, term, g0, g1);
        if (lane == 0) a.w.term[seg_i] = term;
    }
    // dLoss/dpred of the members, 64 at a time (one per lane), written for the user level
    for (int jb = 0; jb < ni; jb += 64) {
        if (jb + lane < ni) {
            const int bq = a.w.perm_i[j0i + jb + lane];
            const float erq = ld_sc1(a.w.pred + bq) - a.rating[bq];
            float cq = 2.f * erq / (float)a.B;
            if (fair) cq = cq + (a.sst[bq] == smin ? g0 : g1);
            st_sc1(a.w.coef + bq, cq);
        }
    }
    drain_stores();      // this wave reads them back below (sc1 loads are served past the L1)
    {
        RowFrag<E> gi;
        handed_grad_sum<E>(gi, j0i, ni, a.w.perm_i, a.w.coef, a.w.side[0], D, lane);
        adam_write<E>(I, c, ir, pi, mi, vi, gi, si, lane);
    }
    // ---- user level of every member, ascending; members whose user is theirs alone are updated here, four in flight
    constexpr int UN = 4;
    for (int jb = 0; jb < ni; jb += 64) {
        const int cnt = min(64, ni - jb);
        int my_b = 0, my_u = 0, my_iux = 0, my_seg = 0;
        float my_c = 0.f;
        if (lane < cnt) {
            my_b = a.w.perm_i[j0i + jb + lane];
            my_c = ld_sc1(a.w.coef + my_b);
            long long uq = a.user[my_b];
            my_u = (uq < 0 || uq >= U.n_rows) ? 0 : (int)uq;
            const int2 q = a.w.info_u[my_b];
            my_iux = q.x;
            my_seg = q.y;
        }
        for (int t0 = 0; t0 < cnt; t0 += UN) {
            RowFrag<E> p[UN], m[UN], v[UN];
            int bq[UN], uq[UN], nq[UN];
            float cq[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const int t = t0 + q < cnt ? t0 + q : cnt - 1;
                bq[q] = __builtin_amdgcn_readlane(my_b, t);
                uq[q] = __builtin_amdgcn_readlane(my_u, t);
                nq[q] = __builtin_amdgcn_readlane(my_iux, t) >> 16;
                cq[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_c), t));
                if (t0 + q < cnt && nq[q] == 1) {
                    const size_t sq = (size_t)bq[q] * D;
                    load_row_sc1<E>(p[q], a.w.side[0] + sq, D, lane);
                    load_row_sc1<E>(m[q], a.w.side[1] + sq, D, lane);
                    load_row_sc1<E>(v[q], a.w.side[2] + sq, D, lane);
                }
            }
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                if (t0 + q >= cnt) continue;
                if (nq[q] == 1) {
                    RowFrag<E> gu;
#pragma unroll
                    for (int e = 0; e < E; ++e) gu.x[e] = cq[q] * pi0.x[e];
                    adam_write<E>(U, c, uq[q], p[q], m[q], v[q], gu, su, lane);
                } else {
                    const int t = t0 + q;
                    const int sg = __builtin_amdgcn_readlane(my_seg, t);
                    const int j0 = __builtin_amdgcn_readlane(my_iux, t) & 0xffff;
                    if (arrive_last(a.w.cnt_u + sg, nq[q], lane)) user_finish<E>(a, j0, nq[q], su, lane);
                }
            }
        }
    }
}

// fixed-order reduction of one batch's squared errors and per-item terms -> loss (one workgroup of 256 threads).  The
// association is that of the three-launch path (per 4 interactions, then strided over 256 threads, butterfly, 4 waves;
// terms per 64 items, then the same), so both paths report the same bits.
__device__ __forceinline__ void step_reduce_loss(const PrevLoss& pl) {
    __shared__ float red[2][4];
    const int B = pl.B;
    const int nb = (B + 3) / 4;
    float a = 0.f, f = 0.f;
    for (int q = threadIdx.x; q < nb; q += 256) {
        const int b0 = 4 * q;
        const float e0 = pl.mse_e[b0], e1 = b0 + 1 < B ? pl.mse_e[b0 + 1] : 0.f, e2 = b0 + 2 < B ? pl.mse_e[b0 + 2] : 0.f,
                    e3 = b0 + 3 < B ? pl.mse_e[b0 + 3] : 0.f;
        a += ((e0 + e1) + e2) + e3;
    }
    const bool per_item = pl.objective >= FR_FOCF_VALUE && pl.objective <= FR_FOCF_OVER;
    const int K = pl.nseg_i[0];
    if (per_item) {
        constexpr int PER = FAIR_THREADS / FAIR_GROUP;     // items per workgroup of the fairness launch
        const int nf = (B * FAIR_GROUP + FAIR_THREADS - 1) / FAIR_THREADS;
        for (int q = threadIdx.x; q < nf; q += 256) {
            float sblk = 0.f;
            const int k1 = min(K, (q + 1) * PER);
            for (int k = q * PER; k < k1; ++k) sblk += pl.term[k];
            f += sblk;
        }
    }
    a = wave_sum(a);
    f = wave_sum(f);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        f = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        const float mse = a / (float)B;
        const float fairv = per_item ? f / (float)K : 0.f;
        const float loss = per_item ? mse + pl.fair_weight * fairv : mse;
        pl.loss_out[0] = loss;
        pl.loss_out[1] = mse;
        pl.loss_out[2] = fairv;
        if (pl.acc) {
            pl.acc[0] += loss;
            pl.acc[1] += mse;
            pl.acc[2] += fairv;
        }
    }
}

// Block roles: block 0 reduces an earlier step's loss (if any); the sweeper blocks and the interaction blocks are dealt
// evenly through the rest of the grid, so that from the first moment the resident waves are a mix of sweeper waves (one
// round trip, then up to S replayed steps of pure VALU work) and interaction waves (two dependent round trips first).
template <int E>
__global__ __launch_bounds__(256) void focf_step_kernel(StepArgs a) {
    if (blockIdx.x == 0) {
        if (a.prev.loss_out) step_reduce_loss(a.prev);
        return;
    }
    const int x = blockIdx.x - 1;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const long long ns = a.n_sweep_blocks, nt = (long long)a.n_sweep_blocks + a.n_inter_blocks;
    // sweeper blocks before x: floor(x * ns / nt); block x is a sweeper block when that count steps at x + 1
    const long long before = (long long)x * ns / nt;
    const bool sweeper = ((long long)(x + 1) * ns / nt) != before;
    if (sweeper) {
        sweep_slice_wave<E>(a.U, a.I, a.c, a.sw, before * 4 + wib, lane);
        return;
    }
    const int b = (int)(x - before) * 4 + wib;
    if (b < a.B) step_interaction<E>(a, b, lane);
}

__global__ __launch_bounds__(256) void focf_step_finish_kernel(PrevLoss pl) { step_reduce_loss(pl); }

static PrevLoss prev_of(void* ws, int64_t B, int dim, int objective, float fair_weight, float* loss_out, float* acc) {
    PrevLoss pl{};
    if (!ws || !loss_out) return pl;
    const FocfWs w = focf_layout(ws, B, dim);
    pl.mse_e = w.mse_e;
    pl.term = w.term;
    pl.nseg_i = w.nseg_i;
    pl.B = (int)B;
    pl.objective = objective;
    pl.fair_weight = fair_weight;
    pl.loss_out = loss_out;
    pl.acc = acc;
    return pl;
}

}  // namespace fr

using namespace fr;

extern "C" int fr_focf_prepare_step(const fr_focf_batch* batches, const int32_t* stamps, int32_t n, const fr_table* U,
                                    const fr_table* I, uint32_t* err_flag, void* stream_) {
    int rc;
    if ((rc = check_table(U, "fr_focf_prepare_step(U)")) || (rc = check_table(I, "fr_focf_prepare_step(I)"))) return rc;
    FR_CHECK_ARG(batches && stamps && n >= 1 && 2 * n <= FR_SORT_JOBS && U->dim == I->dim,
                 "fr_focf_prepare_step: 1..%d batches", FR_SORT_JOBS / 2);
    SortJobList jobs{};
    for (int q = 0; q < n; ++q) {
        const fr_focf_batch& b = batches[q];
        FR_CHECK_ARG(b.user && b.item && b.ws && b.B >= 1 && b.B <= FR_SORT_MAX, "fr_focf_prepare_step: batch %d", q);
        FocfWs w = focf_layout(b.ws, b.B, U->dim);
        FR_CHECK_ARG(b.ws_bytes >= w.bytes, "fr_focf_prepare_step: workspace %zu < %zu bytes", b.ws_bytes, w.bytes);
        SortJob ju{b.user, U->n_rows, w.perm_u, w.seg_start_u, w.seg_row_u, nullptr, w.nseg_u, nullptr, nullptr};
        SortJob ji{b.item, I->n_rows, w.perm_i, w.seg_start_i, w.seg_row_i, nullptr, w.nseg_i, b.sst, w.sst_minmax};
        ju.seg_first = w.seg_first_u;
        ji.seg_first = w.seg_first_i;
        ju.info = w.info_u;
        ji.info = w.info_i;
        ju.cnt = w.cnt_u;
        ji.cnt = w.cnt_i;
        ju.stamp = U->stamp;
        ji.stamp = I->stamp;
        ju.stamp_val = ji.stamp_val = stamps[q];
        jobs.j[2 * q] = ju;
        jobs.j[2 * q + 1] = ji;
        jobs.M[2 * q] = jobs.M[2 * q + 1] = (int)b.B;
    }
    jobs.n = 2 * n;
    return launch_sort_many(jobs, U->n_rows > I->n_rows ? U->n_rows : I->n_rows, err_flag, (hipStream_t)stream_);
}

extern "C" int fr_focf_step(const fr_table* U, const fr_table* I, const fr_adam* adam, const int64_t* user,
                            const int64_t* item, const float* rating, const float* sst, int64_t B, int32_t objective,
                            float fair_weight, int32_t sweep_period, int32_t stamp, void* ws, size_t ws_bytes,
                            float* loss_out, void* prev_ws, int64_t prev_B, float* prev_loss_out, float* loss_acc,
                            uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = check_table(U, "fr_focf_step(U)")) || (rc = check_table(I, "fr_focf_step(I)")) ||
        (rc = check_adam(adam, "fr_focf_step")))
        return rc;
    FR_CHECK_ARG(U->dim == I->dim, "fr_focf_step: user dim %d != item dim %d", U->dim, I->dim);
    FR_CHECK_ARG(user && item && rating && ws, "fr_focf_step: null pointer");
    FR_CHECK_ARG(objective >= FR_FOCF_NONE && objective <= FR_FOCF_OVER,
                 "fr_focf_step: objective %d needs batch-wide statistics before the update (use fr_focf_forward)", objective);
    FR_CHECK_ARG(objective == FR_FOCF_NONE || sst, "fr_focf_step: sst column required for a fairness objective");
    FR_CHECK_ARG(B >= 1 && B <= FR_SORT_MAX, "fr_focf_step: batch size %lld not in 1..%d", (long long)B, FR_SORT_MAX);
    FR_CHECK_ARG(U->step >= 1 && U->step == I->step, "fr_focf_step: table.step must be the step being applied (>=1), "
                 "the same for both tables");
    FR_CHECK_ARG(!U->step_dev && !I->step_dev, "fr_focf_step: device step counters are not supported");
    StepArgs a{};
    a.w = focf_layout(ws, B, U->dim);
    FR_CHECK_ARG(ws_bytes >= a.w.bytes, "fr_focf_step: workspace %zu < %zu bytes", ws_bytes, a.w.bytes);
    a.U = view(U);
    a.I = view(I);
    a.c = make_adamc(adam);
    a.user = user;
    a.item = item;
    a.rating = rating;
    a.sst = sst;
    a.B = (int)B;
    a.objective = objective;
    a.fair_weight = fair_weight;
    a.err = err_flag;
    long long sweep_waves = 0;
    if (sweep_period > 0) {
        a.sw = make_sweep_slice(U, I, sweep_period);
        a.sw.skip_from = stamp;      // the rows of this batch (and of batches prepared for later steps) carry stamps >= it
        sweep_waves = sweep_slice_waves(a.sw);
    }
    a.n_sweep_blocks = (int)((sweep_waves + 3) / 4);
    a.n_inter_blocks = (int)((B + 3) / 4);
    a.prev = prev_of(prev_ws, prev_B, U->dim, objective, fair_weight, prev_loss_out, loss_acc);
    (void)loss_out;   // reduced by the NEXT fr_focf_step (prev_*) or by fr_focf_step_finish
    {
        ProfScope prof(K_FOCF_STEP, stream);
        FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_step_kernel<E>), dim3((unsigned)(1 + a.n_sweep_blocks + a.n_inter_blocks)), dim3(256), 0, stream, a));
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_focf_step_finish(void* ws, size_t ws_bytes, int64_t B, int32_t dim, int32_t objective,
                                   float fair_weight, float* loss_out, float* loss_acc, void* stream_) {
    FR_CHECK_ARG(ws && loss_out && B >= 1 && B <= FR_SORT_MAX && dim >= 1, "fr_focf_step_finish: bad argument");
    FR_CHECK_ARG(ws_bytes >= focf_layout(nullptr, B, dim).bytes, "fr_focf_step_finish: workspace too small");
    const PrevLoss pl = prev_of(ws, B, dim, objective, fair_weight, loss_out, loss_acc);
    hipLaunchKernelGGL(focf_step_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, pl);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
