// Device-side building blocks of the lazy-Adam embedding tables (shared by focf.hip and table.hip).
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace fr {

// Gradient row of one distinct row of a batch: sum of the contributions of its members [j0, j1) of the sorted list in
// ascending batch position (the reference's accumulation order); b0 = perm[j0] is known already.
//   coef != nullptr : contribution of member b = coef[b] * other[b,:]   (rank-1 form: MF models)
//   coef == nullptr : contribution of member b = other[b,:]             (gradient rows from an MLP backward)
template <int E>
__device__ __forceinline__ void segment_grad_sum(RowFrag<E>& g, int j0, int j1, int b0, const int32_t* perm,
                                                 const float* coef, const float* other, int D, int lane, Lay lay) {
#pragma unroll
    for (int e = 0; e < E; ++e) g.x[e] = 0.f;
    // Members in ascending batch position (the reference's accumulation order).  A hot row can have hundreds of
    // members (item-complete FOCF batches: ~100 per item), so the loop must not be a chain of dependent loads: the
    // member ids and coefficients of up to 64 members are read with one coalesced load each and broadcast by
    // readlane, and the rows of SEG_UNROLL members are in flight before the first of them is added.
    constexpr int SEG_UNROLL = 8;
    if (j1 - j0 < SEG_UNROLL) {      // the common case (uniform batches: one or two members): plain uniform loads
        for (int j = j0; j < j1; ++j) {
            const int b = j == j0 ? b0 : uniform(perm[j]);
            const float c1 = coef ? coef[b] : 1.f;
            RowFrag<E> o;
            load_row<E>(o, other + (size_t)lay.at(b) * D, D, lane);
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    float prod = c1 * o.x[e];
                    g.x[e] = g.x[e] + prod;
                }
            }
        }
    } else
    for (int jb = j0; jb < j1; jb += 64) {
        const int cnt = min(64, j1 - jb);
        int my_b = 0;
        float my_c = 1.f;
        if (lane < cnt) {
            my_b = perm[jb + lane];
            if (coef) my_c = coef[my_b];
        }
        int t0 = 0;
        for (; t0 + SEG_UNROLL <= cnt; t0 += SEG_UNROLL) {
            RowFrag<E> o[SEG_UNROLL];
            float cb[SEG_UNROLL];
#pragma unroll
            for (int q = 0; q < SEG_UNROLL; ++q) {
                const int b = __builtin_amdgcn_readlane(my_b, t0 + q);
                cb[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_c), t0 + q));
                load_row<E>(o[q], other + (size_t)lay.at(b) * D, D, lane);
            }
            {
#pragma clang fp contract(off)  // product rounded, then added: grad_row += coef * other_row (autograd order)
#pragma unroll
                for (int q = 0; q < SEG_UNROLL; ++q) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        float prod = cb[q] * o[q].x[e];
                        g.x[e] = g.x[e] + prod;
                    }
                }
            }
        }
        for (; t0 < cnt; ++t0) {      // short segments (the common case: one or two members) and the tail
            const int b = __builtin_amdgcn_readlane(my_b, t0);
            const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_c), t0));
            RowFrag<E> o;
            load_row<E>(o, other + (size_t)lay.at(b) * D, D, lane);
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    float prod = c1 * o.x[e];
                    g.x[e] = g.x[e] + prod;
                }
            }
        }
    }
}

// One distinct row of a batch: sum the contributions of its members in ascending batch position, apply the
// Adam step `T.step` to the caught-up state parked in the workspace, write the row back once.
//   coef != nullptr : contribution of member b = coef[b] * other[b,:]   (rank-1 form: MF models)
//   coef == nullptr : contribution of member b = other[b,:]             (gradient rows from an MLP backward)
template <int E>
__device__ __forceinline__ void segment_update(const TableV& T, const AdamC& c, int k, const int32_t* seg_start,
                                               const int32_t* seg_row, const int32_t* perm, const float* coef,
                                               const float* sp, const float* sm, const float* sv, const float* other,
                                               int lane, Lay lay = Lay{0, 0},   // lay: layout of sp and other
                                               const int32_t* seg_first = nullptr) {
    const int D = T.D;
    // first level of loads: everything that depends on k only (a wave-uniform value costs a memory round trip when it
    // is consumed; seg_first saves the perm[j0] trip of the first -- usually the only -- member)
    const int rj0 = seg_start[k], rj1 = seg_start[k + 1], rrow = seg_row[k];
    const int rb0 = seg_first ? seg_first[k] : 0;
    const int j0 = uniform(rj0), j1 = uniform(rj1);
    const int row = uniform(rrow);
    const int b0 = seg_first ? uniform(rb0) : uniform(perm[j0]);
    RowFrag<E> p, m, v, g;
    load_row<E>(p, sp + (size_t)lay.at(b0) * D, D, lane);
    load_row<E>(m, sm + (size_t)b0 * D, D, lane);
    load_row<E>(v, sv + (size_t)b0 * D, D, lane);
    segment_grad_sum<E>(g, j0, j1, b0, perm, coef, other, D, lane, lay);
    const float2 s = step_scalars(c, T.step);
#pragma unroll
    for (int e = 0; e < E; ++e) adam_elem(p.x[e], m.x[e], v.x[e], g.x[e], s.x, s.y, c);
    store_row<E>(p, T.p + (size_t)row * D, D, lane);
    store_row<E>(m, T.m + (size_t)row * D, D, lane);
    store_row<E>(v, T.v + (size_t)row * D, D, lane);
    if (lane == 0) T.last[row] = T.step;
}

// Bring one untouched row up to `upto` (all missed steps have zero data gradient).  Rows whose stamp is >= skip_from
// belong to a batch of step skip_from or later (a segment wave owns them) and are left alone.
template <int E>
__device__ __forceinline__ void sweep_row(const TableV& T, const AdamC& c, long long row, int upto, int skip_from,
                                          int lane) {
    const int D = T.D;
    // stamp, last and the row itself in ONE round trip (the row is wasted for the few rows that are skipped)
    const int st = T.stamp[row];
    const int lt = T.last[row];
    RowFrag<E> p, m, v;
    load_row<E>(p, T.p + (size_t)row * D, D, lane);
    load_row<E>(m, T.m + (size_t)row * D, D, lane);
    load_row<E>(v, T.v + (size_t)row * D, D, lane);
    if (uniform(st) >= skip_from) return;
    const int t0 = uniform(lt);
    if (t0 >= upto) return;
    replay<E>(p, m, v, t0, upto, c, lane);
    store_row<E>(p, T.p + (size_t)row * D, D, lane);
    store_row<E>(m, T.m + (size_t)row * D, D, lane);
    store_row<E>(v, T.v + (size_t)row * D, D, lane);
    if (lane == 0) T.last[row] = upto;
}

// A NARROW table (D == 1: the user / item bias columns of PFCN_BiasedMF, [N, 1]): a wave takes 64 consecutive rows of the
// sweep slice, one row per lane, instead of one row with 63 idle lanes -- the bias tables' share of a PFCN filter step went
// from as much VALU time as the embedding tables' sweep (128 us of 1.2 ms at 10 M rows) to a few us.  Every lane replays its
// own stretch (last[row], upto] inside one wave-uniform loop over the steps (the per-step scalars are the same for all rows),
// masked off before its own first step: the same operations per element as replay<1>, so the same bits.
__device__ __forceinline__ void sweep_rows_narrow(const TableV& T, const AdamC& c, long long row0, long long hi, int upto,
                                                  int skip_from, int lane) {
    const long long row = row0 + lane;
    const bool in = row < hi;
    const int st = in ? T.stamp[row] : 0x7fffffff;
    const int lt = in ? T.last[row] : upto;
    float p = in ? T.p[row] : 0.f, m = in ? T.m[row] : 0.f, v = in ? T.v[row] : 0.f;
    const bool act = in && st < skip_from && lt < upto;
    const int t0 = act ? lt : upto;
    int jmin = t0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) jmin = min(jmin, __shfl_xor(jmin, o, 64));
    jmin = uniform(jmin);
    if (jmin >= upto) return;
    const bool scaled = !FR_ADAM_PRECISE && c.k1 != 0.f;       // as replay_n chooses
    if (scaled) {
        m *= c.inv_k1;
        v *= c.inv_k2;
    }
    int j = jmin + 1;
    if (scaled) {
        auto one = [&](int jj, float A, float Bc) {      // computed by every lane, kept by the lanes whose stretch has begun
            float pn = p, mn = m, vn = v;                 // (selects, not a branch: no exec-mask round trip per step)
            adam_zero_scaled(pn, mn, vn, A, Bc, c);
            const bool on = jj > t0;
            p = on ? pn : p;
            m = on ? mn : m;
            v = on ? vn : v;
        };
        // The per-step scalars (A_j, B_j) of 64 steps arrive with ONE vector load (lane l: step jb + l) and are handed out by
        // v_readlane; the next 64 are requested before the current ones are used.  (As scalar loads four steps ahead -- the
        // way the wide tables' loops get them -- every iteration waited for the scalar cache: with one such wave per SIMD
        // nothing hides that, 220 cycles per step.)
        auto fetch = [&](int jb) {
            const int jj = jb + lane < c.cap ? jb + lane : c.cap;
            return c.sc[2 * jj + 1];
        };
        float2 cur = fetch(j);
        for (int jb = j; jb <= upto; jb += 64) {
            const float2 nxt = fetch(jb + 64);
            const int n = upto - jb + 1 < 64 ? upto - jb + 1 : 64;
            for (int t = 0; t < n; ++t) {
                const float A = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur.x), t));
                const float Bc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur.y), t));
                one(jb + t, A, Bc);
            }
            cur = nxt;
        }
        m *= c.k1;
        v *= c.k2;
    } else {
        for (; j <= upto; ++j) {
            const float4 s = step_scalars4(c, j);
            if (j > t0) adam_zero(p, m, v, s.x, s.y, c);
        }
    }
    if (act) {
        T.p[row] = p;
        T.m[row] = m;
        T.v[row] = v;
        T.last[row] = upto;
    }
}

// rows of a sweep slice one sweeper wave takes
__host__ __device__ constexpr int sweep_rows_per_wave(int dim) { return dim == 1 ? 64 : (dim <= 64 ? 2 : 1); }

// Pairs pay off while both rows' fragments fit a small register budget: D <= 64 (one register per fragment).  Measured
// with pairs at every width: NFCF (D = 256) step 0.49 -> 0.60 ms, PFCN (D = 128) filter pass 2.09 -> 2.43 ms -- the extra
// fragments cost more occupancy than the packed VALU saves -- so wider rows go one per wave.
__host__ __device__ constexpr bool sweep_pairs(int E) { return E <= 1; }

// Two rows of a sweep slice in one wave: both are requested at once (twice the rows in flight per wave slot) and their
// common stretch of missed steps is replayed interleaved, which the compiler packs into v_pk_* instructions (26 instead
// of 36 issue cycles per row and step).  rowB < 0: only rowA.
template <int E>
__device__ __forceinline__ void sweep_row_pair(const TableV& T, const AdamC& c, long long rowA, long long rowB, int upto,
                                               int skip_from, int lane) {
    if (rowB < 0) {
        sweep_row<E>(T, c, rowA, upto, skip_from, lane);
        return;
    }
    const int D = T.D;
    const int sa = T.stamp[rowA], sb = T.stamp[rowB];
    const int la = T.last[rowA], lb = T.last[rowB];
    RowFrag<E> pa, ma, va, pb, mb, vb;
    load_row<E>(pa, T.p + (size_t)rowA * D, D, lane);
    load_row<E>(ma, T.m + (size_t)rowA * D, D, lane);
    load_row<E>(va, T.v + (size_t)rowA * D, D, lane);
    load_row<E>(pb, T.p + (size_t)rowB * D, D, lane);
    load_row<E>(mb, T.m + (size_t)rowB * D, D, lane);
    load_row<E>(vb, T.v + (size_t)rowB * D, D, lane);
    const int ta = uniform(la), tb = uniform(lb);
    const bool doA = uniform(sa) < skip_from && ta < upto;
    const bool doB = uniform(sb) < skip_from && tb < upto;
    if (doA && doB) {
        if (ta < tb) replay<E>(pa, ma, va, ta, tb, c, lane);
        else if (tb < ta) replay<E>(pb, mb, vb, tb, ta, c, lane);
        replay2<E>(pa, ma, va, pb, mb, vb, ta > tb ? ta : tb, upto, c, lane);
    } else if (doA) {
        replay<E>(pa, ma, va, ta, upto, c, lane);
    } else if (doB) {
        replay<E>(pb, mb, vb, tb, upto, c, lane);
    }
    if (doA) {
        store_row<E>(pa, T.p + (size_t)rowA * D, D, lane);
        store_row<E>(ma, T.m + (size_t)rowA * D, D, lane);
        store_row<E>(va, T.v + (size_t)rowA * D, D, lane);
        if (lane == 0) T.last[rowA] = upto;
    }
    if (doB) {
        store_row<E>(pb, T.p + (size_t)rowB * D, D, lane);
        store_row<E>(mb, T.m + (size_t)rowB * D, D, lane);
        store_row<E>(vb, T.v + (size_t)rowB * D, D, lane);
        if (lane == 0) T.last[rowB] = upto;
    }
}

// Workspace of the generic training pair (fr_table_gather_train / fr_table_apply_grad).
struct TableWs {
    int32_t *perm, *seg_start, *seg_row, *nseg;
    int32_t* seg_first;       // perm[seg_start[k]]: the first (usually the only) member of segment k
    float *m_side, *v_side;   // [M, D] caught-up moments of the gathered rows
    size_t bytes;
};

inline TableWs table_layout(void* base, int64_t M, int D) {
    TableWs w;
    size_t off = 0;
    auto take = [&](size_t nbytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off = align_up(off + nbytes, 256);
        return p;
    };
    const size_t Mp = (size_t)M + 1;
    w.perm = (int32_t*)take(Mp * 4);
    w.seg_start = (int32_t*)take(Mp * 4);
    w.seg_row = (int32_t*)take(Mp * 4);
    w.nseg = (int32_t*)take(4);
    w.seg_first = (int32_t*)take(Mp * 4);
    w.m_side = (float*)take((size_t)M * D * 4);
    w.v_side = (float*)take((size_t)M * D * 4);
    w.bytes = off;
    return w;
}

#define FR_DISPATCH_E(D, ...)                          \
    switch (((D) + 63) / 64) {                         \
        case 1: { constexpr int E = 1; __VA_ARGS__; } break;  \
        case 2: { constexpr int E = 2; __VA_ARGS__; } break;  \
        case 3: { constexpr int E = 3; __VA_ARGS__; } break;  \
        default: { constexpr int E = 4; __VA_ARGS__; } break; \
    }


}  // namespace fr
