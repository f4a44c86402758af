// FOCF training step on ITEM-COMPLETE batches -- the shape the reference's own loader feeds (focf_dataloader.py:37-51: random
// items, ALL interactions of each, until >= train_batch_size rows: K ~ 40-80 distinct items of ~50-600 rows per batch) -- in
// TWO launches instead of the chain's three.
//
// Reference being replaced (one optimizer step, stock PyTorch ops called from Python):
//   FOCF.calculate_loss  focf.py:152-169  (forward :136-143, MSELoss :158, get_item_ratings :75-91, *_unfairness :93-125)
//   loss.backward()      trainer.py:193   dense embedding_dense_backward of both tables
//   optimizer.step()     trainer.py:196   dense torch.optim.Adam over both tables
//
//   launch 1  the chain's gather (focf.hip: one wave per interaction, rows caught up, the item row of a run replayed once per
//             workgroup, scores) in its SORTED form: what a member leaves behind -- caught-up (p, m, v) of both rows, two packed
//             16-byte records -- is parked at the member's position in the item-sorted order, so a run's members sit side by side.
//   launch 2  focf_runs_finish_kernel: ONE WORKGROUP PER ITEM RUN, RUN_WAVES waves.  Wave 0 forms the per-(item, group) sums of
//             focf.py:75-91 in focf_fair_kernel<64>'s lane order (64 lanes strided over the members, butterfly: the same bits);
//             every wave then takes every RUN_WAVES-th member -- dLoss/dpred = MSE part + fairness part, the member's user row
//             updated and stored -- and leaves the member's caught-up user row in LDS; wave 0 sums the item's gradient from LDS
//             in ascending batch position (product rounded, then added: the reference's accumulation order), applies Adam and
//             stores the item row.  Long runs go through LDS in passes of RUN_CAP members.  Only a user that recurs under
//             several items of the batch goes through an arrival counter (its last arriver sums the user's gradient rows in
//             ascending batch position); nobody waits for anybody.  The sweep slice of the step (bounded staleness) and the
//             reduction of the previous step's loss ride in this launch.
// What the chain does in its second and third launch (fairness kernel: 82 waves; backward: the 82 item rows with ~100 members
// each summed by ONE wave per row, 13 dependent batches of parked rows) is done here by 82 x RUN_WAVES waves with the parked rows
// of a run fetched once.
//
// History (round 4): a ONE-launch form -- chunks of the item-sorted order as workgroups, arrival counters
// per item, the completing chunk finishing the item -- was built first and measured at 57.7-62 us per step against the chain's
// 58.7: ten dependent round trips per chunk and 75-89 VGPRs (one or two such workgroups per CU) left the sweeper's VALU work and
// the chunks' latency chains running one after the other instead of side by side.
//
// Every sum has the chain's order and nothing depends on who arrives last: a step is bit-reproducible, and it equals the
// three-launch chain's (tests/test_focf_hip.py: modes `runs*` of the goldens, test_runs_step_*).
#include <stddef.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"
#include "table.hpp"
#include "focf_ws.hpp"
#include "focf_loss.hpp"
#include "focf_gather.hpp"

namespace fr {

#ifndef FR_RUN_WAVES
#define FR_RUN_WAVES 8      // waves per item run / per sweeper workgroup.  Measured (BASELINE sizes, item-complete batches, hipGraph
#endif                    // replay): 8 -> 26.6 us for this launch, 16 -> 29.5 (56 VGPRs, SGPR-bound at 7 waves per SIMD: a CU takes
                          // one 16-wave workgroup or three 8-wave ones)
constexpr int RUN_WAVES = FR_RUN_WAVES;
constexpr int RUN_THREADS = 64 * RUN_WAVES;
#ifndef FR_RUN_DIAG
#define FR_RUN_DIAG 0        // diagnostic builds: 1 skips the members' updates, 2 the item's gradient sum, 3 the statistics,
#endif                     // 4 the whole item work (what the launch costs with its first and last workgroup only)
#ifndef FR_RUN_MB1
#define FR_RUN_MB1 8         // members of a wave whose parked rows are in flight at once, D <= 64
#endif
#ifndef FR_RUN_CAP_BYTES
#define FR_RUN_CAP_BYTES 32768        // LDS of a pass: RUN_CAP caught-up user rows (+ their coefficients)
#endif
#ifndef FR_RUN_ITEM_BLOCKS
#define FR_RUN_ITEM_BLOCKS 128        // workgroups that walk the item runs (k, k + 128, ...): K ~ 40-80, one run each
#endif

__host__ __device__ constexpr int run_cap(int E) { return FR_RUN_CAP_BYTES / (64 * E * 4); }  // members per pass

struct RunArgs {
    TableV U, I;
    AdamC c;
    int B, objective;
    float fair_weight;
    FocfWs w;
    SweepSlice sw;
    int n_item_blocks;
    long long sweep_wave0, n_sweep_waves;     // this launch's share of the sweep slice: waves [wave0, wave0 + n)
    uint32_t* err;
    PrevLoss prev;
    int publish;      // pipelined form: the next batch's gather runs beside this finisher -- rows are stored write-through and
                      // `last` only after they have drained (see PipeWait, focf_gather.hpp)
};

namespace {

// Every pointer of this kernel comes out of v_readlane as an integer, i.e. as a GENERIC pointer: left like that, every access
// is a flat_ instruction (per-lane 64-bit address, LDS aperture check, both wait counters).  The casts below say "global".
typedef __attribute__((address_space(1))) float gfloat_;
typedef __attribute__((address_space(1))) int32_t gint_;
typedef __attribute__((address_space(1))) unsigned int guint_;
typedef int v4i__ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4i__ gint4_;
__device__ __forceinline__ const gfloat_* G(const float* p) { return (const gfloat_*)p; }
__device__ __forceinline__ gfloat_* G(float* p) { return (gfloat_*)p; }
__device__ __forceinline__ const gint_* G(const int32_t* p) { return (const gint_*)p; }
__device__ __forceinline__ gint_* G(int32_t* p) { return (gint_*)p; }
__device__ __forceinline__ guint_* G(unsigned int* p) { return (guint_*)p; }
__device__ __forceinline__ int4 ld4(const int4* p, long long i) {
    const v4i__ v = ((const gint4_*)p)[i];
    return make_int4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float ld1(const float* p) { return __hip_atomic_load(G(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st1(float* p, float v) { __hip_atomic_store(G(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void st1i(int32_t* p, int v) { __hip_atomic_store(G(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int E>
__device__ __forceinline__ void load_row1(RowFrag<E>& f, const float* base, int D, int lane) {      // write-through data
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane + 64 * e;
        f.x[e] = d < D ? ld1(base + d) : 0.f;
    }
}

template <int E>
__device__ __forceinline__ void store_row1(const RowFrag<E>& f, float* base, int D, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane + 64 * e;
        float v = f.x[e];
        asm volatile("" : "+v"(v));      // (the atomic builtin reads its operand through memory otherwise: focf_step.hip)
        if (d < D) st1(base + d, v);
    }
}

template <int E>
__device__ __forceinline__ void gload_row(RowFrag<E>& f, const float* base, int D, int lane) {       // plain, global
    const gfloat_* g = G(base);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane + 64 * e;
        f.x[e] = d < D ? g[d] : 0.f;
    }
}

template <int E>
__device__ __forceinline__ void gstore_row(const RowFrag<E>& f, float* base, int D, int lane) {
    gfloat_* g = G(base);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane + 64 * e;
        if (d < D) g[d] = f.x[e];
    }
}

__device__ __forceinline__ int4 uni4(int4 v) { return make_int4(uniform(v.x), uniform(v.y), uniform(v.z), uniform(v.w)); }

// The kernel's argument block, taken ONCE through VGPRs: lane l of register i loads dword 64 i + l of the kernarg segment (one
// vector load per 64 dwords), every field is then a v_readlane away.  Left to scalar loads, the ~100 fields of RunArgs do not
// fit the SGPR file: the compiler re-fetches them where they are used, each behind its own s_waitcnt, and with thousands of
// waves starting at once those loads miss the small scalar caches (several us per dependent level: the same finding as
// focf_step.hip's KV).  Values that came out of v_readlane cannot be re-fetched, so under pressure they are spilled to VGPR
// lanes instead -- no memory round trip either way.
template <typename T>
__device__ __forceinline__ T args_through_vgprs(int lane) {
    constexpr int ND = (int)((sizeof(T) + 3) / 4);
    static_assert(ND <= 256, "argument block too large for four VGPRs of dwords");
    const unsigned* kp = reinterpret_cast<const unsigned*>(
        (const void*)(const __attribute__((address_space(4))) void*)__builtin_amdgcn_kernarg_segment_ptr());
    unsigned v[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < (ND + 63) / 64; ++i) v[i] = 64 * i + lane < ND ? kp[64 * i + lane] : 0u;
    unsigned d[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = (unsigned)__builtin_amdgcn_readlane((int)v[i >> 6], i & 63);
    T out;
    __builtin_memcpy(&out, d, sizeof(T));
    return out;
}

}  // namespace

#if FR_RUN_DIAG == 5       // wave time lines: (item k, wave 0 / last wave) x 8 stamps of the 100 MHz clock into the unused task_rec array
#define RUN_STAMP(i) do { if (wv == 0 || wv == RUN_WAVES - 1) reinterpret_cast<unsigned long long*>(a.w.task_rec)[((size_t)k * 2 + (wv != 0)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RUN_STAMP(i) do {} while (0)
#endif

// One item run [j0, j0 + n) of the item-sorted order, by all RUN_WAVES waves of a workgroup; `ir` = the item's table row.
// Launch 1 parked everything BY SORTED POSITION (SortedPark, focf.hip): member q of the run sits at j0 + q -- its packed
// records (user row, the extent of the user's segment, batch position | rating, sst, score, MSE part of dLoss/dpred) and its
// caught-up rows -- and every member's copy of the item's caught-up row holds the same bits (the run's first one is read).
// So ONE level of loads serves the whole run: a dependent round trip costs ~3 us in this launch, and the first version of this
// function (positions -> records -> rows, four members at a time) was a chain of six (profiles/r04_runs_finish_wave_trace.txt).
template <int E, int MB = (E == 1 ? FR_RUN_MB1 : 4)>
__device__ __forceinline__ void run_finish_item(const RunArgs& a, int k, int ij0, int n, int ir, float* pu_s, float* coef_s,
                                                float* sh, int lane, int wv) {
    constexpr int CAP = run_cap(E);      // MB = parked user rows in flight per wave (x 3 row fragments)
    const FocfWs& w = a.w;
    const int D = a.U.D, step = a.U.step;
    const bool per_item = a.objective >= FR_FOCF_VALUE && a.objective <= FR_FOCF_OVER;
    RUN_STAMP(0);
    // ---- the one level of loads ---------------------------------------------------------------------------------------------
    RowFrag<E> pi, mi, vi;
    gload_row<E>(pi, w.side[3] + (size_t)ij0 * D, D, lane);
    if (wv == 0) {
        gload_row<E>(mi, w.side[4] + (size_t)ij0 * D, D, lane);
        gload_row<E>(vi, w.side[5] + (size_t)ij0 * D, D, lane);
    }
    float smin = 0.f, smax = 0.f, Kf = 1.f;
    if (per_item) {
        smin = G(w.sst_minmax)[0];
        smax = G(w.sst_minmax)[1];
        Kf = (float)G(w.nseg_i)[0];
    }
    // statistics (wave 0): lane `sub` takes members sub, sub + 64, ... -- the first two now, further ones in a loop
    const bool s0 = wv == 0 && per_item && lane < n, s1 = wv == 0 && per_item && lane + 64 < n;
    int4 sv0 = make_int4(0, 0, 0, 0), sv1 = make_int4(0, 0, 0, 0);
    if (s0) sv0 = ld4(w.task_info, ij0 + lane);
    if (s1) sv1 = ld4(w.task_info, ij0 + lane + 64);
    // this wave's members of the FIRST pass: every RUN_WAVES-th one, lane t holds the records of the t-th
    const int cnt0 = min(CAP, n);
    const int mine0 = cnt0 > wv ? (cnt0 - wv + RUN_WAVES - 1) / RUN_WAVES : 0;
    int4 lrc = make_int4(0, 0, 0, 0), lvl = make_int4(0, 0, 0, 0);
    if (lane < mine0) {
        lrc = ld4(w.task_rec, ij0 + wv + RUN_WAVES * lane);
        lvl = ld4(w.task_info, ij0 + wv + RUN_WAVES * lane);
    }
    RowFrag<E> pu[MB], mu[MB], vu[MB];
#pragma unroll
    for (int u = 0; u < MB; ++u) {
        if (u < mine0) {
            const size_t so = (size_t)(ij0 + wv + RUN_WAVES * u) * D;
            gload_row<E>(pu[u], w.side[0] + so, D, lane);
            gload_row<E>(mu[u], w.side[1] + so, D, lane);
            gload_row<E>(vu[u], w.side[2] + so, D, lane);
        }
    }
    RUN_STAMP(1);
    // (a) the per-(item, group) sums of focf.py:75-91 in focf_fair_kernel<64>'s order: lane `sub` adds its members one after
    //     the other (ascending position), then the butterfly
    if (wv == 0) {
        float term = 0.f, g0 = 0.f, g1 = 0.f;
        if (per_item && FR_RUN_DIAG != 3) {
            float sp0 = 0.f, sp1 = 0.f, st0 = 0.f, st1_ = 0.f, n0 = 0.f, n1 = 0.f;
            bool bad = false;
            auto add = [&](bool on, const int4& v) {
                if (!on) return;
                const float r = __int_as_float(v.x), s = __int_as_float(v.y), pr = __int_as_float(v.z);
                bad |= (s != smin && s != smax);
                if (s == smin) {
                    sp0 += pr; st0 += r; n0 += 1.f;
                } else {
                    sp1 += pr; st1_ += r; n1 += 1.f;
                }
            };
            add(s0, sv0);
            add(s1, sv1);
            for (int j = ij0 + lane + 128; j < ij0 + n; j += 64) add(true, ld4(w.task_info, j));      // runs of > 128 members
            if (bad && a.err) atomicOr(a.err, FR_DEV_ERR_SST_GROUPS);
            sp0 = group_sum<64>(sp0); sp1 = group_sum<64>(sp1);
            st0 = group_sum<64>(st0); st1_ = group_sum<64>(st1_);
            n0 = group_sum<64>(n0);   n1 = group_sum<64>(n1);
            focf_fair_eval(a.objective, a.fair_weight, Kf, sp0, sp1, st0, st1_, n0, n1, term, g0, g1);
        }
        if (lane == 0) {
            sh[0] = g0;
            sh[1] = g1;
            G(w.term)[k] = term;
        }
    }
    RUN_STAMP(2);
    __syncthreads();
    RUN_STAMP(3);
    const float g0 = sh[0], g1 = sh[1];
    const float2 sc = step_scalars(a.c, step);
    RowFrag<E> gi;
#pragma unroll
    for (int e = 0; e < E; ++e) gi.x[e] = 0.f;
    for (int base = 0; base < n; base += CAP) {
        const int cnt = min(CAP, n - base);
        const int mine = cnt > wv ? (cnt - wv + RUN_WAVES - 1) / RUN_WAVES : 0;
        if (base > 0) {                            // (a further pass of a long run: its records now)
            lrc = make_int4(0, 0, 0, 0);
            lvl = make_int4(0, 0, 0, 0);
            if (lane < mine) {
                lrc = ld4(w.task_rec, ij0 + base + wv + RUN_WAVES * lane);
                lvl = ld4(w.task_info, ij0 + base + wv + RUN_WAVES * lane);
            }
        }
        // Users that occur under several items of the batch: ALL of this wave's such members arrive now, lane-parallel, before
        // the wave has any row store in flight -- dLoss/dpred handed over (write-through, by batch position), ONE drain, one
        // atomic per lane.
        float lco = __int_as_float(lvl.w);         // the MSE part 2 (pred - r) / B, as focf_gather_kernel formed it
        if (per_item) lco = lco + ((__int_as_float(lvl.y) == smin) ? g0 : g1);      // + the fairness part, as focf_fair_kernel adds it
        const int lnu = (lrc.y >> 16) & 0xffff;
        const bool lmulti = lane < mine && lnu > 1;
        unsigned lold = 0;
        if (__ballot(lmulti)) {
            if (lmulti) st1(w.coef + lrc.w, lco);
            drain();
            if (lmulti) lold = __hip_atomic_fetch_add(G(w.cnt_u) + lrc.z, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool llast = lmulti && (int)lold + 1 == lnu;
        unsigned long long done_mask = 0ull;      // (publish) members whose user row this wave wrote in this pass
        // (b) the members of this pass, MB at a time
#if FR_RUN_DIAG == 1
        for (int t0 = 0; t0 < 0; t0 += MB) {
#else
        for (int t0 = 0; t0 < mine; t0 += MB) {
#endif
            if (base > 0 || t0 > 0) {
#pragma unroll
                for (int u = 0; u < MB; ++u) {
                    if (t0 + u < mine) {
                        const size_t so = (size_t)(ij0 + base + wv + RUN_WAVES * (t0 + u)) * D;
                        gload_row<E>(pu[u], w.side[0] + so, D, lane);
                        gload_row<E>(mu[u], w.side[1] + so, D, lane);
                        gload_row<E>(vu[u], w.side[2] + so, D, lane);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < MB; ++u) {
                if (t0 + u >= mine) break;
                const int t = t0 + u, q = wv + RUN_WAVES * t;
                const int ur = __builtin_amdgcn_readlane(lrc.x, t);
                const int ux = __builtin_amdgcn_readlane(lrc.y, t);
                const float coef = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lco), t));
#pragma unroll
                for (int e = 0; e < E; ++e) pu_s[(size_t)q * (64 * E) + lane + 64 * e] = pu[u].x[e];
                if (lane == 0) coef_s[q] = coef;
                const int uj0 = ux & 0xffff, nu = (ux >> 16) & 0xffff;
                RowFrag<E> g;
#pragma unroll
                for (int e = 0; e < E; ++e) g.x[e] = 0.f;
                bool finish = true;
                if (nu == 1) {
                    // the user occurs once in the batch: its gradient row is 0 + coef * item row (segment_grad_sum's one term)
#pragma clang fp contract(off)
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const float prod = coef * pi.x[e];
                        g.x[e] = g.x[e] + prod;
                    }
                } else {
                    // the user occurs under several items (arrived above): its LAST arriver sums the user's gradient rows in
                    // ascending batch position (segment_grad_sum's order) from the items' parked rows
                    finish = __builtin_amdgcn_readlane((int)llast, t) != 0;
                    if (finish) {
                        // the caught-up state of the user's FIRST member, as segment_update takes it (launch 1 replayed the
                        // row once per occurrence: the same bits, but the rule keeps the result independent of who is last)
                        const int bu = uniform(G(w.perm_u)[uj0]);
                        const int ju0 = uniform(G(w.pos_i)[bu]);
                        gload_row<E>(pu[u], w.side[0] + (size_t)ju0 * D, D, lane);
                        gload_row<E>(mu[u], w.side[1] + (size_t)ju0 * D, D, lane);
                        gload_row<E>(vu[u], w.side[2] + (size_t)ju0 * D, D, lane);
                        for (int ju = uj0; ju < uj0 + nu; ++ju) {
                            const int bb = uniform(G(w.perm_u)[ju]);
                            const float cb = ld1(w.coef + bb);
                            const int jb = uniform(G(w.pos_i)[bb]);
                            RowFrag<E> o;
                            gload_row<E>(o, w.side[3] + (size_t)jb * D, D, lane);      // member bb's copy of ITS item's row
                            {
#pragma clang fp contract(off)
#pragma unroll
                                for (int e = 0; e < E; ++e) {
                                    const float prod = cb * o.x[e];
                                    g.x[e] = g.x[e] + prod;
                                }
                            }
                        }
                    }
                }
                if (finish) {
#pragma unroll
                    for (int e = 0; e < E; ++e) adam_elem(pu[u].x[e], mu[u].x[e], vu[u].x[e], g.x[e], sc.x, sc.y, a.c);
                    if (a.publish) {
                        store_row1<E>(pu[u], a.U.p + (size_t)ur * D, D, lane);
                        store_row1<E>(mu[u], a.U.m + (size_t)ur * D, D, lane);
                        store_row1<E>(vu[u], a.U.v + (size_t)ur * D, D, lane);
                        done_mask |= 1ull << t;
                    } else {
                        gstore_row<E>(pu[u], a.U.p + (size_t)ur * D, D, lane);
                        gstore_row<E>(mu[u], a.U.m + (size_t)ur * D, D, lane);
                        gstore_row<E>(vu[u], a.U.v + (size_t)ur * D, D, lane);
                        if (lane == 0) G(a.U.last)[ur] = step;
                    }
                }
            }
        }
        if (a.publish && done_mask) {      // this pass's rows have landed device-wide: now their `last`, lane t for member t
            drain();
            if ((done_mask >> lane) & 1ull) st1i(a.U.last + lrc.x, step);
        }
        RUN_STAMP(4);
        __syncthreads();
        RUN_STAMP(5);
        // (c) the item's gradient over this pass, members in ascending batch position (= ascending sorted position: the
        //     sort is stable), product rounded, then added
        if (wv == 0 && FR_RUN_DIAG != 2) {
            // (eight members' LDS reads in flight before the first product: left to itself the loop waited for every read,
            // ~100 x an LDS round trip per run)
            constexpr int GU = 8;
            int q = 0;
            for (; q + GU <= cnt; q += GU) {
                float cq[GU], rq[GU][E];
#pragma unroll
                for (int z = 0; z < GU; ++z) {
                    cq[z] = coef_s[q + z];
#pragma unroll
                    for (int e = 0; e < E; ++e) rq[z][e] = pu_s[(size_t)(q + z) * (64 * E) + lane + 64 * e];
                }
                {
#pragma clang fp contract(off)
#pragma unroll
                    for (int z = 0; z < GU; ++z) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const float prod = cq[z] * rq[z][e];
                            gi.x[e] = gi.x[e] + prod;
                        }
                    }
                }
            }
            {
#pragma clang fp contract(off)
                for (; q < cnt; ++q) {
                    const float cz = coef_s[q];
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const float prod = cz * pu_s[(size_t)q * (64 * E) + lane + 64 * e];
                        gi.x[e] = gi.x[e] + prod;
                    }
                }
            }
        }
        __syncthreads();
    }
    RUN_STAMP(6);
    if (wv == 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) adam_elem(pi.x[e], mi.x[e], vi.x[e], gi.x[e], sc.x, sc.y, a.c);
        if (a.publish) {
            store_row1<E>(pi, a.I.p + (size_t)ir * D, D, lane);
            store_row1<E>(mi, a.I.m + (size_t)ir * D, D, lane);
            store_row1<E>(vi, a.I.v + (size_t)ir * D, D, lane);
            drain();
            if (lane == 0) st1i(a.I.last + ir, step);
        } else {
            gstore_row<E>(pi, a.I.p + (size_t)ir * D, D, lane);
            gstore_row<E>(mi, a.I.m + (size_t)ir * D, D, lane);
            gstore_row<E>(vi, a.I.v + (size_t)ir * D, D, lane);
            if (lane == 0) G(a.I.last)[ir] = step;
        }
    }
    RUN_STAMP(7);
}

template <int E>
__global__ __launch_bounds__(RUN_THREADS) void focf_runs_finish_kernel(RunArgs a_kernarg) {
    const RunArgs a = args_through_vgprs<RunArgs>((int)(threadIdx.x & 63));       // (the ONLY kernel argument: offset 0)
    constexpr int CAP = run_cap(E);
    __shared__ __align__(16) float lds_rows[CAP * 64 * E];
    __shared__ float coef_s[CAP];
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int blk = (int)blockIdx.x;
    if (blk == 0) {                      // the previous step's loss (fixed order), if one is waiting
        if (a.prev.loss_out) {
            if (threadIdx.x >= 256) return;
            step_reduce_loss<256>(a.prev);
        }
        return;
    }
    blk -= 1;
    if (blk >= a.n_item_blocks) {        // the step's slice of the bounded-staleness sweep: one wave per pair of rows
        const long long wave = (long long)(blk - a.n_item_blocks) * RUN_WAVES + wv;
        if (wave < a.n_sweep_waves) sweep_slice_wave<E>(a.U, a.I, a.c, a.sw, a.sweep_wave0 + wave, lane);
        return;
    }
    for (int k = blk; k < a.B; k += a.n_item_blocks) {
        // level 1: everything about run k in ONE round trip (entries past the last run are read and not used)
        const int rK = G(a.w.nseg_i)[0], rj0 = G(a.w.seg_start_i)[k], rj1 = G(a.w.seg_start_i)[k + 1];
        const int rir = G(a.w.seg_row_i)[k];
        if (k >= uniform(rK) || FR_RUN_DIAG == 4) break;
        const int j0 = uniform(rj0);
        run_finish_item<E>(a, k, j0, uniform(rj1) - j0, uniform(rir), lds_rows, coef_s, sh, lane, wv);
        __syncthreads();
    }
}

// ---- the two launches of consecutive steps side by side ---------------------------------------------------------------------
// Launch k of the pipelined form = the finisher of batch k - 1 (first in the grid) + the gather of batch k + the sweep slice of
// step k.  The two chains that follow each other inside a step (gather ~22 us, item runs ~25 us) run beside each other here;
// what ties them -- a row that both batches hold -- goes through PipeWait (focf_gather.hpp).
#ifndef FR_PIPE_MB
#define FR_PIPE_MB 4      // (the finisher's rows in flight per wave here: fewer registers, so that three workgroups fit a CU)
#endif
struct PipeArgs {
    RunArgs fin;              // FIRST member: the finisher role takes it through VGPRs from offset 0.  fin.B == 0: nothing to finish
    TableV U, I;              // the batch being gathered (B == 0: none): tables at ITS step
    AdamC c;
    const int64_t *user, *item;
    const float* rating;
    int B, upto;
    FocfWs w;
    SortedPark sp;
    PipeWait pw;
    uint32_t* err;
    int n_fin_blocks, n_gather_blocks, n_sweep_blocks, n_sweep_front;      // item-run / gather / sweeper workgroups (of these: in front)
    SweepSlice sw;
    long long n_sweep_waves;
};

#ifndef FR_PIPE_PRIO
#define FR_PIPE_PRIO 1    // item-run waves issue ahead of gather waves ahead of sweeper waves: the item runs end at 25 us as in a
#endif                    // launch of their own instead of at 33 (role spans, -DFR_PIPE_TRACE: scratch/pipe_trace.py)
#ifndef FR_PIPE_WPE
#define FR_PIPE_WPE 8     // waves per SIMD the allocator aims at (64 registers, six spilled in the finisher's rare path): the
#endif                    // launch lives on how many gather waves wait for their rows at once -- 16 per CU at the 91 registers
                          // the compiler takes when left alone, 32 here
#ifndef FR_PIPE_PAIR_MAX_E
#define FR_PIPE_PAIR_MAX_E 0      // widths (in 64-column fragments) whose gather takes TWO interactions per wave
#endif                            // (focf_gather_pair_body: half the gather waves, packed replay of the two user rows).  0 = none:
                                  // measured SLOWER at D = 64 (44.4-48.6 us per step against 42.9-43.3 on the same box, gathers
                                  // ending at 37-39 us instead of 33.5): the launch is not short of VALU issue (the replay is ~17 us
                                  // of it) but of independent chains, and a pair wave is one chain twice as long.  Same bits as
                                  // the one-interaction gather (`-DFR_PIPE_PAIR_MAX_E=1` passes tests/test_focf_hip.py).
__host__ __device__ constexpr bool pipe_pair(int E) { return E <= FR_PIPE_PAIR_MAX_E; }
#ifdef FR_PIPE_TRACE      // diagnostic build: first start / last end of every role's waves (100 MHz clock), of the LAST launch
__device__ unsigned long long pipe_dbg[8 * 8];      // a ring of eight launches (slot = step & 7); launch n clears slot n + 4
struct PipeSpan {
    int role, slot;
    __device__ PipeSpan(int r, int step) : role(r), slot(step & 7) {
        if (r == 0 && threadIdx.x < 8) pipe_dbg[((step + 4) & 7) * 8 + threadIdx.x] = (threadIdx.x & 1) ? 0ull : ~0ull;
        if (threadIdx.x == 0 && (blockIdx.x & 15) == 0) atomicMin(&pipe_dbg[slot * 8 + 2 * role], __builtin_amdgcn_s_memrealtime());
    }
    __device__ ~PipeSpan() {
        if ((threadIdx.x & 63) == 0 && (blockIdx.x & 15) == 0) atomicMax(&pipe_dbg[slot * 8 + 2 * role + 1], __builtin_amdgcn_s_memrealtime());
    }
};
#define PIPE_SPAN(r) PipeSpan span_(r, p.upto)
#else
#define PIPE_SPAN(r) do {} while (0)
#endif

template <int E>
__global__ __launch_bounds__(RUN_THREADS) __attribute__((amdgpu_waves_per_eu(FR_PIPE_WPE, FR_PIPE_WPE))) void focf_runs_pipe_kernel(PipeArgs p) {
    constexpr int CAP = run_cap(E);
    __shared__ __align__(16) float lds_rows[CAP * 64 * E];
    __shared__ float coef_s[CAP];
    __shared__ float sh[4];
    // (a workgroup has one role: the gather's LDS lies in the item runs' row buffer)
    using GLds = GatherLds<E, pipe_pair(E) ? 2 * RUN_WAVES : RUN_WAVES>;
    static_assert(sizeof(GLds) <= sizeof(lds_rows), "the gather's LDS fits the item runs' row buffer");
    GLds& glds = *reinterpret_cast<GLds*>(lds_rows);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int blk = (int)blockIdx.x;
    // Grid order = start order: the loss, then the SWEEPER (its tasks are the longest single chains of the launch -- a pair of rows
    // replays up to two sweep periods, 13 us median, 26 us at worst -- and behind the gathers they started 13 us late and ended
    // the launch at 40 us), then the previous batch's item runs, then this batch's gathers (which wait for nobody but those).
    if (blk == 0) {
        PIPE_SPAN(0);
        const RunArgs a = args_through_vgprs<RunArgs>(lane);
        if (a.prev.loss_out && threadIdx.x < 256) step_reduce_loss<256>(a.prev);
        return;
    }
    blk -= 1;
    if (blk < p.n_sweep_front) {         // the head of the sweeper's start order: its longest tasks
        PIPE_SPAN(1);
        const long long wave = (long long)blk * RUN_WAVES + wv;
        if (wave < p.n_sweep_waves) sweep_slice_wave<E>(p.U, p.I, p.c, p.sw, wave, lane);
        return;
    }
    blk -= p.n_sweep_front;
    if (blk < p.n_fin_blocks) {          // the previous batch: focf_runs_finish_kernel's item-run workgroups
        PIPE_SPAN(2);
#if FR_PIPE_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        const RunArgs a = args_through_vgprs<RunArgs>(lane);
        for (int k = blk; k < a.B; k += a.n_item_blocks) {
            const int rK = G(a.w.nseg_i)[0], rj0 = G(a.w.seg_start_i)[k], rj1 = G(a.w.seg_start_i)[k + 1];
            const int rir = G(a.w.seg_row_i)[k];
            if (k >= uniform(rK)) break;
            const int j0 = uniform(rj0);
            run_finish_item<E, FR_PIPE_MB>(a, k, j0, uniform(rj1) - j0, uniform(rir), lds_rows, coef_s, sh, lane, wv);
            __syncthreads();
        }
        return;
    }
    blk -= p.n_fin_blocks;
    if (blk < p.n_gather_blocks) {       // this batch's gather: one or two interactions per wave
        PIPE_SPAN(3);
#if FR_PIPE_PRIO
        __builtin_amdgcn_s_setprio(2);
#endif
        if (blk == 0 && threadIdx.x == 0) *p.w.defer = DeferLoss{nullptr, 0, 0.f, 0};
        if constexpr (pipe_pair(E))
            focf_gather_pair_body<E, RUN_WAVES>(p.U, p.I, p.c, p.user, p.item, p.rating, p.B, p.upto, p.w, p.err, blk, glds, p.sp, p.pw);
        else
            focf_gather_body<E, true, true, true, true, RUN_WAVES>(p.U, p.I, p.c, p.user, p.item, p.rating, p.B, p.upto, p.upto, p.w,
                                                                   0.f, nullptr, p.err, blk, glds, p.sp, p.pw);
        return;
    }
    blk -= p.n_gather_blocks;            // the tail of the sweeper's start order: short tasks, into the slots the gathers leave
    {
        PIPE_SPAN(1);
        const long long wave = (long long)(p.n_sweep_front + blk) * RUN_WAVES + wv;
        if (wave < p.n_sweep_waves) sweep_slice_wave<E>(p.U, p.I, p.c, p.sw, wave, lane);
    }
}

}  // namespace fr

using namespace fr;

#ifdef FR_PIPE_TRACE
extern "C" __attribute__((visibility("default"))) int fr_debug_pipe_trace(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pipe_dbg), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long init[64];
        for (int i = 0; i < 64; ++i) init[i] = (i & 1) ? 0ull : ~0ull;
        if (hipMemcpyToSymbol(HIP_SYMBOL(pipe_dbg), init, sizeof(init)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// The pipelined form of fr_focf_step_runs: ONE launch = the finisher of the batch gathered by the previous call (`fin_ws`,
// `fin_B`, applied at `fin_step`; NULL: none) + the gather of this call's batch (`user` NULL / B == 0: none -- the call that
// drains the pipeline) + this step's sweep slice + the loss of the batch finished by the previous call (`prev_ws`).  `own`:
// int32 [2][n_users] and [2][n_items], zeroed by the caller when the tables' steps are rewound; parity of the step selects
// the half a gather writes.  U / I carry the step of the batch being gathered (or fin_step when there is none).
extern "C" int fr_focf_step_runs_pipe(const fr_table* U, const fr_table* I, const fr_adam* adam, const int64_t* user,
                                      const int64_t* item, const float* rating, const float* sst, int64_t B, int32_t objective,
                                      float fair_weight, int32_t sweep_period, int32_t stamp, void* ws, size_t ws_bytes,
                                      void* fin_ws, int64_t fin_B, int32_t fin_step, void* prev_ws, int64_t prev_B,
                                      float* prev_loss_out, float* loss_acc, int32_t* own_u, int32_t* own_i, uint32_t* err_flag,
                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = check_table(U, "fr_focf_step_runs_pipe(U)")) || (rc = check_table(I, "fr_focf_step_runs_pipe(I)")) ||
        (rc = check_adam(adam, "fr_focf_step_runs_pipe")))
        return rc;
    const bool gather = user != nullptr && B > 0, finish = fin_ws != nullptr && fin_B > 0;
    FR_CHECK_ARG(U->dim == I->dim && U->step >= 1 && U->step == I->step && !U->step_dev && !I->step_dev,
                 "fr_focf_step_runs_pipe: tables of one width at one host-side step");
    FR_CHECK_ARG(objective >= FR_FOCF_NONE && objective <= FR_FOCF_OVER, "fr_focf_step_runs_pipe: objective %d", objective);
    FR_CHECK_ARG(!gather || (item && rating && ws && own_u && own_i && B <= FR_SORT_MAX && (objective == FR_FOCF_NONE || sst)),
                 "fr_focf_step_runs_pipe: bad batch");
    FR_CHECK_ARG(!finish || (fin_step >= 1 && fin_B <= FR_SORT_MAX && (!gather || fin_step == U->step - 1)),
                 "fr_focf_step_runs_pipe: the batch to finish must be the previous step's");
    PipeArgs p{};
    p.c = make_adamc(adam);
    p.err = err_flag;
    // -- the finisher's side
    RunArgs& a = p.fin;
    a.c = p.c;
    a.objective = objective;
    a.fair_weight = fair_weight;
    a.err = err_flag;
    a.publish = gather ? 1 : 0;
    fr_table Uf = *U, If = *I;
    if (finish) {
        Uf.step = If.step = fin_step;
        a.w = focf_layout(fin_ws, fin_B, U->dim);
        a.B = (int)fin_B;
        a.n_item_blocks = (int)std::min<long long>(FR_RUN_ITEM_BLOCKS, fin_B);
    }
    a.U = view(&Uf);
    a.I = view(&If);
    a.prev = prev_of(prev_ws, prev_B, U->dim, objective, fair_weight, prev_loss_out, loss_acc, false);
    p.n_fin_blocks = a.n_item_blocks;
    // -- the gather's side
    if (gather) {
        p.w = focf_layout(ws, B, U->dim);
        FR_CHECK_ARG(ws_bytes >= p.w.bytes, "fr_focf_step_runs_pipe: workspace %zu < %zu bytes", ws_bytes, p.w.bytes);
        p.U = view(U);
        p.I = view(I);
        p.user = user;
        p.item = item;
        p.rating = rating;
        p.B = (int)B;
        p.upto = U->step - 1;
        p.sp = SortedPark{p.w.pos_i, p.w.info, sst, p.w.task_rec, p.w.task_info, p.w.mse_e};
        const int cur = U->step & 1, prv = cur ^ 1;
        p.pw = PipeWait{own_u + (size_t)prv * U->n_rows, own_i + (size_t)prv * I->n_rows, own_u + (size_t)cur * U->n_rows,
                        own_i + (size_t)cur * I->n_rows, finish ? fin_step : -1};
        const int per_block = pipe_pair((U->dim + 63) / 64) ? 2 * RUN_WAVES : RUN_WAVES;      // interactions per gather workgroup
        p.n_gather_blocks = (int)((B + per_block - 1) / per_block);
        if (sweep_period > 0) {
            p.sw = make_sweep_slice(U, I, sweep_period);
            p.sw.skip_from = finish ? std::min(stamp, fin_step) : stamp;      // the finisher's rows are not the sweeper's either
            p.n_sweep_waves = sweep_slice_waves(p.sw);
        }
    }
    p.n_sweep_blocks = (int)((p.n_sweep_waves + RUN_WAVES - 1) / RUN_WAVES);
    static const int front_pct = getenv("FAIRREC_PIPE_SWEEP_FRONT") ? atoi(getenv("FAIRREC_PIPE_SWEEP_FRONT")) : 100;
    p.n_sweep_front = (int)((long long)p.n_sweep_blocks * std::min(std::max(front_pct, 0), 100) / 100);
    const long long blocks = 1 + p.n_sweep_blocks + p.n_fin_blocks + p.n_gather_blocks;
    ProfScope prof(K_FOCF_STEP, stream);
    FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_runs_pipe_kernel<E>), dim3((unsigned)blocks), dim3(RUN_THREADS), 0, stream, p));
    FR_CHECK_LAUNCH();
    return FR_OK;
}

// One optimizer step of FOCF on an item-complete batch; same contract as fr_focf_step (the batch was prepared by
// fr_focf_prepare_step with `stamp`, table.step = the step being applied, the loss is reduced by the next launch or by
// fr_focf_step_finish).
extern "C" int fr_focf_step_runs(const fr_table* U, const fr_table* I, const fr_adam* adam, const int64_t* user,
                                 const int64_t* item, const float* rating, const float* sst, int64_t B, int32_t objective,
                                 float fair_weight, int32_t sweep_period, int32_t stamp, void* ws, size_t ws_bytes, void* prev_ws,
                                 int64_t prev_B, float* prev_loss_out, float* loss_acc, uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = check_table(U, "fr_focf_step_runs(U)")) || (rc = check_table(I, "fr_focf_step_runs(I)")) ||
        (rc = check_adam(adam, "fr_focf_step_runs")))
        return rc;
    FR_CHECK_ARG(U->dim == I->dim, "fr_focf_step_runs: user dim %d != item dim %d", U->dim, I->dim);
    FR_CHECK_ARG(ws && user && item && rating, "fr_focf_step_runs: null pointer");
    FR_CHECK_ARG(objective >= FR_FOCF_NONE && objective <= FR_FOCF_OVER,
                 "fr_focf_step_runs: objective %d needs batch-wide statistics before the update (use fr_focf_forward)", objective);
    FR_CHECK_ARG(objective == FR_FOCF_NONE || sst, "fr_focf_step_runs: sst column required for a fairness objective");
    FR_CHECK_ARG(B >= 1 && B <= FR_SORT_MAX, "fr_focf_step_runs: batch size %lld not in 1..%d", (long long)B, FR_SORT_MAX);
    FR_CHECK_ARG(U->step >= 1 && U->step == I->step, "fr_focf_step_runs: table.step must be the step being applied (>=1), "
                 "the same for both tables");
    FR_CHECK_ARG(!U->step_dev && !I->step_dev, "fr_focf_step_runs: device step counters are not supported");
    RunArgs a{};
    a.w = focf_layout(ws, B, U->dim);
    FR_CHECK_ARG(ws_bytes >= a.w.bytes, "fr_focf_step_runs: workspace %zu < %zu bytes", ws_bytes, a.w.bytes);
    a.U = view(U);
    a.I = view(I);
    a.c = make_adamc(adam);
    a.B = (int)B;
    a.objective = objective;
    a.fair_weight = fair_weight;
    a.err = err_flag;
    // The step's sweep slice (bounded staleness) is split between the two launches: FAIRREC_RUNS_SWEEP_SPLIT per cent of its
    // waves ride in FRONT of the gather (a latency-bound launch with idle VALUs), the rest behind the item runs of launch 2.
    static const int split_pct = getenv("FAIRREC_RUNS_SWEEP_SPLIT") ? atoi(getenv("FAIRREC_RUNS_SWEEP_SPLIT")) : 0;
    long long sweep_blocks = 0, in_gather = 0;
    if (sweep_period > 0) {
        a.sw = make_sweep_slice(U, I, sweep_period);
        a.sw.skip_from = stamp;       // the rows of this batch (and of batches prepared for later steps) carry stamps >= it
        const long long total = sweep_slice_waves(a.sw);
        in_gather = total * std::min(std::max(split_pct, 0), 100) / 100;
        a.sweep_wave0 = in_gather;
        a.n_sweep_waves = total - in_gather;
        sweep_blocks = (a.n_sweep_waves + RUN_WAVES - 1) / RUN_WAVES;
    }
    // launch 1: rows caught up and parked at their batch positions, scores, the MSE part of dLoss/dpred
    if ((rc = focf_launch_gather_runs(U, I, a.c, user, item, rating, sst, B, a.w, err_flag, stream, a.sw, in_gather))) return rc;
    a.prev = prev_of(prev_ws, prev_B, U->dim, objective, fair_weight, prev_loss_out, loss_acc, false);
    a.n_item_blocks = (int)std::min<long long>(FR_RUN_ITEM_BLOCKS, B);
    ProfScope prof(K_FOCF_STEP, stream);
    const unsigned blocks = (unsigned)(1 + a.n_item_blocks + sweep_blocks);
    FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_runs_finish_kernel<E>), dim3(blocks), dim3(RUN_THREADS), 0, stream, a));
    FR_CHECK_LAUNCH();
    return FR_OK;
}

// The step loop of trainer.py:181-196 over a run of ITEM-COMPLETE batches (what FOCFDataLoader yields) in one call: n launches
// of fr_focf_step_runs_pipe, the sorted prepare of the batches (fr_focf_prepare_step, FR_FOCF_PREPARE_MAX per launch) one group
// ahead on the library's side stream, joined once per group.  Batch k is gathered at step U->step + k with stamp
// first_stamp + k; its item runs ride in the launch of batch k + 1, its loss is reduced by the launch of batch k + 2.
extern "C" int fr_focf_runs_many(const fr_table* U, const fr_table* I, const fr_adam* adam, const fr_focf_batch* batches,
                                 int32_t n, int32_t objective, float fair_weight, int32_t sweep_period, int32_t first_stamp,
                                 void* fin_ws, int64_t fin_B, int32_t fin_step, float* fin_loss_out, void* prev_ws,
                                 int64_t prev_B, float* prev_loss_out, float* loss_ring, int32_t loss_slots, int32_t first_slot,
                                 float* loss_acc, int32_t* own_u, int32_t* own_i, uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    FR_CHECK_ARG(U && I && batches && n >= 1, "fr_focf_runs_many: null pointer / no batch");
    FR_CHECK_ARG(loss_ring && loss_slots >= 1 && first_slot >= 0 && first_slot < loss_slots,
                 "fr_focf_runs_many: loss ring (float[4 * loss_slots]) and a first slot inside it");
    FR_CHECK_ARG(first_stamp >= 1 && first_stamp <= INT32_MAX - n, "fr_focf_runs_many: first stamp");
    FR_CHECK_ARG(U->step >= 1 && U->step == I->step && U->step <= INT32_MAX - n,
                 "fr_focf_runs_many: table.step must be the step the FIRST batch is applied at (>= 1), the same for both tables");
    FR_CHECK_ARG(!fin_ws || (fin_loss_out && fin_step == U->step - 1),
                 "fr_focf_runs_many: a pending batch must be the previous step's, with its loss slot");
    constexpr int G = FR_FOCF_PREPARE_MAX;
    // a workspace is written by its batch's prepare (up to 2 G - 1 launches before its gather) and read until the launch that
    // reduces its loss (2 launches after): batches closer than 2 G + 2 need different ones, and none may be a pending one
    for (int k = 0; k < n; ++k) {
        FR_CHECK_ARG(batches[k].ws && batches[k].user && batches[k].item && batches[k].rating,
                     "fr_focf_runs_many: batch %d: null pointer", k);
        for (int j = k + 1; j < n && j < k + 2 * G + 2; ++j)
            FR_CHECK_ARG(batches[j].ws != batches[k].ws, "fr_focf_runs_many: batches %d and %d share a workspace", k, j);
        FR_CHECK_ARG(k >= 2 * G || (batches[k].ws != fin_ws && batches[k].ws != prev_ws),
                     "fr_focf_runs_many: batch %d uses a workspace that is still pending", k);
    }
    std::vector<int32_t> stamps((size_t)n);
    for (int k = 0; k < n; ++k) stamps[(size_t)k] = first_stamp + k;
    SideStream* ss = side_stream();
    // the call's own events (the library keeps no host state between calls: two host threads may drive two engines); an
    // event destroyed while work recorded behind it is still in flight is released by the runtime when that work is done
    struct Events {
        hipEvent_t fork = nullptr, done[2] = {nullptr, nullptr};
        ~Events() {
            if (fork) (void)hipEventDestroy(fork);
            if (done[0]) (void)hipEventDestroy(done[0]);
            if (done[1]) (void)hipEventDestroy(done[1]);
        }
    } ev;
    if (ss && n > G) {
        FR_CHECK_HIP(hipEventCreateWithFlags(&ev.fork, hipEventDisableTiming));
        FR_CHECK_HIP(hipEventCreateWithFlags(&ev.done[0], hipEventDisableTiming));
        FR_CHECK_HIP(hipEventCreateWithFlags(&ev.done[1], hipEventDisableTiming));
    }
    hipEvent_t ev_fork = ev.fork;
    hipEvent_t* ev_done = ev.done;
    int rc;
    auto prepare = [&](int g, hipStream_t st) {
        const int lo = g * G, cnt = std::min(G, n - lo);
        return fr_focf_prepare_step(batches + lo, stamps.data() + lo, cnt, U, I, sweep_period, err_flag, st);
    };
    if ((rc = prepare(0, stream))) return rc;       // the first group on the caller's stream: nothing to overlap it with yet
    fr_table tu = *U, ti = *I;
    for (int k = 0; k < n; ++k) {
        if (k % G == 0) {
            const int g = k / G;
            if (g > 0 && ss) FR_CHECK_HIP(hipStreamWaitEvent(stream, ev_done[g & 1], 0));
            if ((g + 1) * G < n) {
                if (ss) {      // (everything that last used the next group's workspaces was enqueued before this point)
                    FR_CHECK_HIP(hipEventRecord(ev_fork, stream));
                    FR_CHECK_HIP(hipStreamWaitEvent(ss->stream, ev_fork, 0));
                    if ((rc = prepare(g + 1, ss->stream))) return rc;
                    FR_CHECK_HIP(hipEventRecord(ev_done[(g + 1) & 1], ss->stream));
                } else if ((rc = prepare(g + 1, stream))) {
                    return rc;
                }
            }
        }
        const fr_focf_batch& b = batches[k];
        rc = fr_focf_step_runs_pipe(&tu, &ti, adam, b.user, b.item, b.rating, b.sst, b.B, objective, fair_weight, sweep_period,
                                    stamps[(size_t)k], b.ws, b.ws_bytes, fin_ws, fin_B, fin_step, prev_ws, prev_B, prev_loss_out,
                                    loss_acc, own_u, own_i, err_flag, stream_);
        if (rc) return rc;
        prev_ws = fin_ws; prev_B = fin_B; prev_loss_out = fin_loss_out;      // finished in this launch: its loss comes next
        fin_ws = b.ws; fin_B = b.B; fin_step = tu.step;
        fin_loss_out = loss_ring + 4 * (size_t)((first_slot + k) % loss_slots);
        ++tu.step; ++ti.step;
    }
    return FR_OK;
}
