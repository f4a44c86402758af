// The FOCF gather (one wave per interaction), as a device function: focf.hip's kernels and the pipelined item-run step
// (focf_runs.hip) carry it.
#pragma once
#include "common.hpp"
#include "kernels.hpp"
#include "table.hpp"
#include "focf_ws.hpp"

namespace fr {

// ------------------------------------------------------------------------------------------------
// gather: one wave per interaction
// ------------------------------------------------------------------------------------------------
// LDS of one gather workgroup: the item row of each of its waves and room for the caught-up item rows that waves of the
// same item share (item-complete batches -- the shape FOCFDataLoader produces, focf_dataloader.py:37-51 -- put the ~100
// interactions of an item side by side, and the item row carries the longest replay of the wave: replaying it once per
// workgroup instead of once per wave takes it off three waves out of four)
template <int E, int NWAVES = GATHER_THREADS / WAVE>
struct GatherLds {
    float red[NWAVES];
    float irow[NWAVES][3][64 * E + 1];   // E = 0: not used (one float each)
};

// SORTED (fr_focf_step_runs): everything a member leaves behind is parked at its position in the ITEM-SORTED order instead of
// at its batch position, and its scalars are packed into two 16-byte records -- the workgroup that finishes an item run then
// finds the run's members side by side and needs no id, permutation or record lookup first (a dependent round trip costs
// ~3 us in that launch: profiles/r04_runs_finish_wave_trace.txt).
struct SortedPark {
    const int32_t* pos_of;     // [B] sorted position of a batch position (fr_focf_prepare_step)
    const int4* info;          // [B] (user j0 | n << 16, user segment, item j0 | n << 16, item segment)
    const float* sst;          // [B] or null
    int4* recs;                // [B] by sorted position: (user row, user j0 | n << 16, user segment, batch position)
    int4* vals;                // [B] by sorted position: (rating, sst, pred, MSE part of dLoss/dpred) as float bits
    float* mse_e;              // [B] squared errors (any fixed order serves the loss reduction: by batch position)
};

// PIPE (fr_focf_step_runs_pipe): this gather runs in the SAME launch as the finisher of the previous batch.  A row that batch
// holds too (`own_prev[row]` = its step: written by that batch's own gather, one launch ago) is taken only after the finisher
// has published it -- `last[row]` reaches the step, stored device-wide after the row's write-through stores have drained --
// and then through device-scope loads.  Nobody else waits, and a waiting workgroup was dispatched after every finisher
// workgroup (they are first in the grid), so the wait ends; it is bounded anyway (FR_DEV_ERR_PIPE_WAIT).
struct PipeWait {
    const int32_t *own_prev_u, *own_prev_i;
    int32_t *own_cur_u, *own_cur_i;
    int fin_step;      // step of the batch being finished beside this gather; < 0: none
};

template <int E>
__device__ __forceinline__ int pipe_take_row(const TableV& T, int row, int need, RowFrag<E>& p, RowFrag<E>& m, RowFrag<E>& v,
                                             int lane, uint32_t* err) {
    int l = 0;
    for (int spin = 0; spin < (1 << 18); ++spin) {
        l = __hip_atomic_load(T.last + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (l >= need) break;
        __builtin_amdgcn_s_sleep(8);
    }
    l = uniform(l);
    if (l < need && lane == 0 && err) atomicOr(err, FR_DEV_ERR_PIPE_WAIT);
    const int D = T.D;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane + 64 * e;
        const size_t o = (size_t)row * D + d;
        p.x[e] = d < D ? __hip_atomic_load(T.p + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        m.x[e] = d < D ? __hip_atomic_load(T.m + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        v.x[e] = d < D ? __hip_atomic_load(T.v + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    }
    return l;
}

// NW = waves (interactions) per workgroup: 4 in the gather's own launches; 8 where it rides in a launch of 512-thread
// workgroups -- the squared errors are still summed in groups of four (the loss keeps its bits) and the item row of a run is
// still replayed by the first wave that holds it (the same bits whoever replays it).
template <int E, bool TRAIN, bool SHARE, bool SORTED = false, bool PIPE = false, int NW = GATHER_THREADS / WAVE>
__device__ __forceinline__ void focf_gather_body(
    const TableV& U, const TableV& I, const AdamC& c, const int64_t* __restrict__ user, const int64_t* __restrict__ item,
    const float* __restrict__ rating, int B, int upto_u, int upto_i, const FocfWs& w, float max_rating,
    float* __restrict__ predict_out, uint32_t* err, int block, GatherLds<SHARE ? E : 0, NW>& lds, SortedPark sp = SortedPark{},
    PipeWait pw = PipeWait{}) {
    static_assert(NW % 4 == 0, "squared errors are reduced four waves at a time");
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int b = block * NW + wib;
    const bool valid = b < B;
    const int D = U.D;
    float e2 = 0.f;
    int ur = 0, ir = -1 - wib;       // an idle wave's "item" matches nobody's
    int js = b;                      // where this member parks: its batch position, or (SORTED) its place in the item order
    int4 inf = make_int4(0, 0, 0, 0);
    float sv = 0.f;
    if (SORTED && valid) {           // (requested together with the ids: the same round trip)
        js = sp.pos_of[b];
        inf = sp.info[b];
        if (sp.sst) sv = sp.sst[b];
    }
    if (valid) {
        long long ul = user[b], il = item[b];
        if (ul < 0 || ul >= U.n_rows || il < 0 || il >= I.n_rows) {
            if (lane == 0 && err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
            ul = ul < 0 || ul >= U.n_rows ? 0 : ul;
            il = il < 0 || il >= I.n_rows ? 0 : il;
        }
        ur = uniform((int)ul);
        ir = uniform((int)il);
    }
    // the first wave of the workgroup with the same item replays its row for all of them; every wave reads the item ids
    // of the whole workgroup itself (one cache line, same round trip as its own ids): no barrier unless rows are shared
    int lead = wib;
    bool shares = false;
    if (SHARE) {
        int it[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int bq = block * NW + q;
            long long v = bq < B ? item[bq] : -1;
            it[q] = (v < 0 || v >= I.n_rows) ? (bq < B ? 0 : -1 - q) : (int)v;     // same clamping as above
        }
#pragma unroll
        for (int q = NW - 1; q >= 0; --q) {
            if (q < wib && it[q] == ir) lead = q;
#pragma unroll
            for (int r = 0; r < q; ++r) shares |= it[q] == it[r];
        }
    }
    lead = uniform(lead);
    shares = uniform((int)shares) != 0;
    RowFrag<E> pu, mu, vu, pi, mi, vi;
    if (valid) {
        // one level of dependent loads: the rows and their `last` stamps are requested together (a wave-uniform value
        // is a load + readfirstlane, i.e. a full memory round trip each time one is consumed: ~3-4 us per level here)
        int lu = U.last[ur];
        int li = lead == wib ? I.last[ir] : 0;
        int ou = -1, oi = -1;
        if (PIPE) {
            ou = pw.own_prev_u[ur];
            if (lead == wib) oi = pw.own_prev_i[ir];
        }
        load_row<E>(pu, U.p + (size_t)ur * D, D, lane);
        if (lead == wib) load_row<E>(pi, I.p + (size_t)ir * D, D, lane);
        load_row<E>(mu, U.m + (size_t)ur * D, D, lane);
        load_row<E>(vu, U.v + (size_t)ur * D, D, lane);
        if (lead == wib) {
            load_row<E>(mi, I.m + (size_t)ir * D, D, lane);
            load_row<E>(vi, I.v + (size_t)ir * D, D, lane);
        }
        if (PIPE) {
            if (pw.fin_step >= 0 && uniform(ou) == pw.fin_step) lu = pipe_take_row<E>(U, ur, pw.fin_step, pu, mu, vu, lane, err);
            if (lead == wib && pw.fin_step >= 0 && uniform(oi) == pw.fin_step)
                li = pipe_take_row<E>(I, ir, pw.fin_step, pi, mi, vi, lane, err);
            if (lane == 0) {
                pw.own_cur_u[ur] = upto_u + 1;
                if (lead == wib) pw.own_cur_i[ir] = upto_i + 1;
            }
        }
#ifdef FR_DIAG_GATHER_NO_REPLAY     // timing experiment only (wrong numbers): what the launch costs when its gathers find fresh rows
        if (PIPE) lu = upto_u, li = upto_i;
#endif
        const int t0u = uniform(lu);
        // replay the optimizer steps each row missed (zero data gradient, weight decay only): first the
        // steps only the staler row missed, then the common tail on both rows interleaved
        if (lead != wib) {
            replay<E>(pu, mu, vu, t0u, upto_u, c, lane);
        } else {
            const int t0i = uniform(li);
            if (upto_u == upto_i) {
                if (t0u < t0i) replay<E>(pu, mu, vu, t0u, t0i, c, lane);
                else if (t0i < t0u) replay<E>(pi, mi, vi, t0i, t0u, c, lane);
                replay2<E>(pu, mu, vu, pi, mi, vi, t0u > t0i ? t0u : t0i, upto_u, c, lane);
            } else {
                replay<E>(pu, mu, vu, t0u, upto_u, c, lane);
                replay<E>(pi, mi, vi, t0i, upto_i, c, lane);
            }
        }
    }
    if (SHARE && shares) {      // workgroup-uniform: hand the caught-up item rows over through LDS
        if (valid && lead == wib) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                lds.irow[wib][0][lane + 64 * e] = pi.x[e];
                lds.irow[wib][1][lane + 64 * e] = mi.x[e];
                lds.irow[wib][2][lane + 64 * e] = vi.x[e];
            }
        }
        __syncthreads();
        if (valid && lead != wib) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                pi.x[e] = lds.irow[lead][0][lane + 64 * e];
                mi.x[e] = lds.irow[lead][1][lane + 64 * e];
                vi.x[e] = lds.irow[lead][2][lane + 64 * e];
            }
        }
    }
    if (valid) {
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) dot = fmaf(pu.x[e], pi.x[e], dot);
        dot = wave_sum(dot);

        if (TRAIN) {
            const size_t so = (size_t)(SORTED ? uniform(js) : b) * D;
            store_row<E>(pu, w.side[0] + so, D, lane);
            store_row<E>(mu, w.side[1] + so, D, lane);
            store_row<E>(vu, w.side[2] + so, D, lane);
            store_row<E>(pi, w.side[3] + so, D, lane);
            store_row<E>(mi, w.side[4] + so, D, lane);
            store_row<E>(vi, w.side[5] + so, D, lane);
            const float er = dot - rating[b];
            e2 = er * er;
            if (lane == 0) {
                // never lowered: a look-ahead fr_focf_prepare_step may already have stamped the row for a later batch
                atomicMax(&U.stamp[ur], upto_u + 1);
                atomicMax(&I.stamp[ir], upto_i + 1);
                const float cm = 2.f * er / (float)B;  // d mean((pred-r)^2) / d pred
                if (SORTED) {
                    sp.recs[js] = make_int4(ur, inf.x, inf.y, b);
                    sp.vals[js] = make_int4(__float_as_int(rating[b]), __float_as_int(sv), __float_as_int(dot), __float_as_int(cm));
                    sp.mse_e[b] = e2;
                } else {
                    w.pred[b] = dot;
                    w.coef[b] = cm;
                }
            }
        } else if (lane == 0) {
            predict_out[b] = fminf(fmaxf(dot, 0.f), max_rating) / max_rating;
        }
    }
    if (TRAIN) {
        if (lane == 0) lds.red[wib] = e2;
        __syncthreads();
        if ((threadIdx.x & 255) == 0) {    // one partial per FOUR interactions, whatever the workgroup's size
            const int q = threadIdx.x >> 8;
            if (block * (NW / 4) + q < w.n_gather_blocks)
                w.mse_part[block * (NW / 4) + q] = ((lds.red[4 * q] + lds.red[4 * q + 1]) + lds.red[4 * q + 2]) + lds.red[4 * q + 3];
            if (block == 0 && q == 0) *w.ticket = 0u;   // arm the fair kernel's in-launch finalisation (next launch)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the pipelined gather with TWO interactions per wave
// ------------------------------------------------------------------------------------------------
// fr_focf_step_runs_pipe asks for a gather wave per interaction beside the item runs and the sweeper, and a CU holds 32 waves:
// the gathers that found no slot at the start ended the launch.  Here a wave takes two NEIGHBOURING
// interactions: half the waves, one round trip for the ids and one for the rows of both, and the two user rows replay as
// packed pairs over the steps both missed (replay2: 34 instead of 41 cycles per row-step).  Same values as
// focf_gather_body<E, true, true, true, true>: a replay gives the same bits alone or paired, the item row of a run is replayed
// by the wave that holds its first member in the workgroup, the squared errors are summed four interactions at a time.
template <int E>
struct PairRows {
    RowFrag<E> p0, m0, v0, p1, m1, v1;
};

// rows current as of t0 / t1 brought to `upto` (t == upto: that row is left alone); by value in and out (through a reference the
// fragments stay in scratch memory, focf_step.hip::replay_two_v)
template <int E>
__device__ __forceinline__ PairRows<E> replay_pair_v(PairRows<E> r, int t0, int t1, int upto, const AdamC& c, int lane) {
    t0 = t0 < upto ? t0 : upto;
    t1 = t1 < upto ? t1 : upto;
    if (t0 != t1) {     // wave-uniform: the steps only the staler row missed
        const bool old0 = t0 < t1;
        RowFrag<E> p, m, v;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            p.x[e] = old0 ? r.p0.x[e] : r.p1.x[e];
            m.x[e] = old0 ? r.m0.x[e] : r.m1.x[e];
            v.x[e] = old0 ? r.v0.x[e] : r.v1.x[e];
        }
        replay<E>(p, m, v, old0 ? t0 : t1, old0 ? t1 : t0, c, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (old0) {
                r.p0.x[e] = p.x[e]; r.m0.x[e] = m.x[e]; r.v0.x[e] = v.x[e];
            } else {
                r.p1.x[e] = p.x[e]; r.m1.x[e] = m.x[e]; r.v1.x[e] = v.x[e];
            }
        }
    }
    replay2<E>(r.p0, r.m0, r.v0, r.p1, r.m1, r.v1, t0 > t1 ? t0 : t1, upto, c, lane);
    return r;
}

// NW waves, 2 NW interactions per workgroup; `lds` may alias memory of the launch's other roles (a workgroup has one role)
template <int E, int NW>
__device__ __forceinline__ void focf_gather_pair_body(
    const TableV& U, const TableV& I, const AdamC& c, const int64_t* __restrict__ user, const int64_t* __restrict__ item,
    const float* __restrict__ rating, int B, int upto, const FocfWs& w, uint32_t* err, int block, GatherLds<E, 2 * NW>& lds,
    SortedPark sp, PipeWait pw) {
    constexpr int NS = 2 * NW;
    static_assert(NS % 4 == 0, "squared errors are reduced four interactions at a time");
    const int lane = threadIdx.x & 63;
    const int wib = uniform((int)(threadIdx.x >> 6));
    const int s0 = 2 * wib, s1 = s0 + 1;
    const int b0 = block * NS + s0, b1 = b0 + 1;
    const bool val0 = b0 < B, val1 = b1 < B;
    const int D = U.D;
    // -- level 1: the ids of both interactions, the item ids of the whole workgroup (two cache lines), the prepare's records
    const int c0 = val0 ? b0 : 0, c1 = val1 ? b1 : 0;       // (clamped: the loads stay unconditional)
    long long ul0 = user[c0], ul1 = user[c1], il0 = item[c0], il1 = item[c1];
    int js0 = sp.pos_of[c0], js1 = sp.pos_of[c1];
    const int4 inf0 = sp.info[c0], inf1 = sp.info[c1];
    const float sv0 = sp.sst ? sp.sst[c0] : 0.f, sv1 = sp.sst ? sp.sst[c1] : 0.f;
    const float rt0 = rating[c0], rt1 = rating[c1];
    int it[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const int bq = block * NS + q;
        const long long v = item[bq < B ? bq : 0];
        it[q] = uniform(bq < B ? ((v < 0 || v >= I.n_rows) ? 0 : (int)v) : -1 - q);     // same clamping as the ids below
    }
    if ((val0 && (ul0 < 0 || ul0 >= U.n_rows || il0 < 0 || il0 >= I.n_rows)) ||
        (val1 && (ul1 < 0 || ul1 >= U.n_rows || il1 < 0 || il1 >= I.n_rows))) {
        if (lane == 0 && err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
    }
    const int ur0 = uniform((ul0 < 0 || ul0 >= U.n_rows) ? 0 : (int)ul0), ur1 = uniform((ul1 < 0 || ul1 >= U.n_rows) ? 0 : (int)ul1);
    const int ir0 = uniform(val0 ? ((il0 < 0 || il0 >= I.n_rows) ? 0 : (int)il0) : -1 - s0);
    const int ir1 = uniform(val1 ? ((il1 < 0 || il1 >= I.n_rows) ? 0 : (int)il1) : -1 - s1);
    // the first interaction of the workgroup with an item replays its row for all of them
    int lead0 = s0, lead1 = s1;
#pragma unroll
    for (int q = NS - 1; q >= 0; --q) {
        if (q < s0 && it[q] == ir0) lead0 = q;
        if (q < s1 && it[q] == ir1) lead1 = q;
    }
    const bool own0 = val0 && lead0 == s0, own1 = val1 && lead1 == s1;
    PairRows<E> ru, ri;
    float e20 = 0.f, e21 = 0.f;
    if (val0) {
        // -- level 2: the rows, their `last` stamps and their owners of the previous batch, all requested together
        const int ro0 = own0 ? ir0 : 0, ro1 = own1 ? ir1 : 0;
        int lu0 = U.last[ur0], lu1 = U.last[ur1];
        int li0 = I.last[ro0], li1 = I.last[ro1];
        const int ou0 = pw.own_prev_u[ur0], ou1 = pw.own_prev_u[ur1];
        const int oi0 = pw.own_prev_i[ro0], oi1 = pw.own_prev_i[ro1];
        load_row<E>(ru.p0, U.p + (size_t)ur0 * D, D, lane);
        load_row<E>(ru.p1, U.p + (size_t)ur1 * D, D, lane);
        load_row<E>(ru.m0, U.m + (size_t)ur0 * D, D, lane);
        load_row<E>(ru.m1, U.m + (size_t)ur1 * D, D, lane);
        load_row<E>(ru.v0, U.v + (size_t)ur0 * D, D, lane);
        load_row<E>(ru.v1, U.v + (size_t)ur1 * D, D, lane);
        if (own0) {
            load_row<E>(ri.p0, I.p + (size_t)ir0 * D, D, lane);
            load_row<E>(ri.m0, I.m + (size_t)ir0 * D, D, lane);
            load_row<E>(ri.v0, I.v + (size_t)ir0 * D, D, lane);
        }
        if (own1) {
            load_row<E>(ri.p1, I.p + (size_t)ir1 * D, D, lane);
            load_row<E>(ri.m1, I.m + (size_t)ir1 * D, D, lane);
            load_row<E>(ri.v1, I.v + (size_t)ir1 * D, D, lane);
        }
        if (pw.fin_step >= 0) {
            if (uniform(ou0) == pw.fin_step) lu0 = pipe_take_row<E>(U, ur0, pw.fin_step, ru.p0, ru.m0, ru.v0, lane, err);
            if (val1 && uniform(ou1) == pw.fin_step) lu1 = pipe_take_row<E>(U, ur1, pw.fin_step, ru.p1, ru.m1, ru.v1, lane, err);
            if (own0 && uniform(oi0) == pw.fin_step) li0 = pipe_take_row<E>(I, ir0, pw.fin_step, ri.p0, ri.m0, ri.v0, lane, err);
            if (own1 && uniform(oi1) == pw.fin_step) li1 = pipe_take_row<E>(I, ir1, pw.fin_step, ri.p1, ri.m1, ri.v1, lane, err);
        }
        if (lane == 0) {
            pw.own_cur_u[ur0] = upto + 1;
            if (val1) pw.own_cur_u[ur1] = upto + 1;
            if (own0) pw.own_cur_i[ir0] = upto + 1;
            if (own1) pw.own_cur_i[ir1] = upto + 1;
        }
        // the steps each row missed (zero data gradient, weight decay only); a row this wave does not replay counts as current
        ru = replay_pair_v<E>(ru, uniform(lu0), val1 ? uniform(lu1) : upto, upto, c, lane);
        ri = replay_pair_v<E>(ri, own0 ? uniform(li0) : upto, own1 ? uniform(li1) : upto, upto, c, lane);
        // hand the caught-up item rows over through LDS
        if (own0) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                lds.irow[s0][0][lane + 64 * e] = ri.p0.x[e];
                lds.irow[s0][1][lane + 64 * e] = ri.m0.x[e];
                lds.irow[s0][2][lane + 64 * e] = ri.v0.x[e];
            }
        }
        if (own1) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                lds.irow[s1][0][lane + 64 * e] = ri.p1.x[e];
                lds.irow[s1][1][lane + 64 * e] = ri.m1.x[e];
                lds.irow[s1][2][lane + 64 * e] = ri.v1.x[e];
            }
        }
    }
    __syncthreads();
    if (val0) {
        if (!own0) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ri.p0.x[e] = lds.irow[lead0][0][lane + 64 * e];
                ri.m0.x[e] = lds.irow[lead0][1][lane + 64 * e];
                ri.v0.x[e] = lds.irow[lead0][2][lane + 64 * e];
            }
        }
        if (val1 && !own1) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ri.p1.x[e] = lds.irow[lead1][0][lane + 64 * e];
                ri.m1.x[e] = lds.irow[lead1][1][lane + 64 * e];
                ri.v1.x[e] = lds.irow[lead1][2][lane + 64 * e];
            }
        }
        float dot0 = 0.f, dot1 = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            dot0 = fmaf(ru.p0.x[e], ri.p0.x[e], dot0);
            dot1 = fmaf(ru.p1.x[e], ri.p1.x[e], dot1);
        }
        dot0 = wave_sum(dot0);
        dot1 = wave_sum(dot1);
        js0 = uniform(js0);
        js1 = uniform(js1);
        const size_t so0 = (size_t)js0 * D, so1 = (size_t)js1 * D;
        store_row<E>(ru.p0, w.side[0] + so0, D, lane);
        store_row<E>(ru.m0, w.side[1] + so0, D, lane);
        store_row<E>(ru.v0, w.side[2] + so0, D, lane);
        store_row<E>(ri.p0, w.side[3] + so0, D, lane);
        store_row<E>(ri.m0, w.side[4] + so0, D, lane);
        store_row<E>(ri.v0, w.side[5] + so0, D, lane);
        if (val1) {
            store_row<E>(ru.p1, w.side[0] + so1, D, lane);
            store_row<E>(ru.m1, w.side[1] + so1, D, lane);
            store_row<E>(ru.v1, w.side[2] + so1, D, lane);
            store_row<E>(ri.p1, w.side[3] + so1, D, lane);
            store_row<E>(ri.m1, w.side[4] + so1, D, lane);
            store_row<E>(ri.v1, w.side[5] + so1, D, lane);
        }
        const float er0 = dot0 - rt0, er1 = dot1 - rt1;
        e20 = er0 * er0;
        e21 = val1 ? er1 * er1 : 0.f;
        if (lane < 2 && (lane == 0 || val1)) {     // lane 0 leaves the first interaction's records, lane 1 the second's
            const bool k = lane == 1;
            const int ur = k ? ur1 : ur0, ir = k ? ir1 : ir0, js = k ? js1 : js0, b = k ? b1 : b0;
            const int4 inf = k ? inf1 : inf0;
            const float er = k ? er1 : er0;
            // never lowered: a look-ahead fr_focf_prepare_step may already have stamped the row for a later batch
            atomicMax(&U.stamp[ur], upto + 1);
            atomicMax(&I.stamp[ir], upto + 1);
            const float cm = 2.f * er / (float)B;  // d mean((pred-r)^2) / d pred
            sp.recs[js] = make_int4(ur, inf.x, inf.y, b);
            sp.vals[js] = make_int4(__float_as_int(k ? rt1 : rt0), __float_as_int(k ? sv1 : sv0), __float_as_int(k ? dot1 : dot0),
                                    __float_as_int(cm));
            sp.mse_e[b] = k ? e21 : e20;
        }
    }
    if (lane == 0) {
        lds.red[s0] = e20;
        lds.red[s1] = e21;
    }
    __syncthreads();
    if ((threadIdx.x & 127) == 0) {        // one partial per FOUR interactions
        const int q = threadIdx.x >> 7;
        if (block * (NS / 4) + q < w.n_gather_blocks)
            w.mse_part[block * (NS / 4) + q] = ((lds.red[4 * q] + lds.red[4 * q + 1]) + lds.red[4 * q + 2]) + lds.red[4 * q + 3];
        if (block == 0 && q == 0) *w.ticket = 0u;   // arm the fair kernel's in-launch finalisation (next launch)
    }
}

}  // namespace fr
