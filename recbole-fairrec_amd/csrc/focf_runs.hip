// FOCF training step on ITEM-COMPLETE batches -- the shape the reference's own loader feeds (focf_dataloader.py:37-51: random
// items, ALL interactions of each, until >= train_batch_size rows: K ~ 40-80 distinct items of ~50-600 rows per batch) -- as
// ONE launch.
//
// Reference being replaced (one optimizer step, stock PyTorch ops called from Python):
//   FOCF.calculate_loss  focf.py:152-169  (forward :136-143, MSELoss :158, get_item_ratings :75-91, *_unfairness :93-125)
//   loss.backward()      trainer.py:193   dense embedding_dense_backward of both tables
//   optimizer.step()     trainer.py:196   dense torch.optim.Adam over both tables
//
// Why not focf_step_kernel: there a row shared by n interactions is finished by the ONE wave that arrives last (fine for
// n = 2..3, the uniform case); with n ~ 100 that wave walks 100 parked user rows alone (141 us per step measured).  And the
// three-launch chain (focf.hip) pays its dependent load levels three times (58.6 us).  Here:
//
//   stage 1  a workgroup = RUN_WAVES waves = a CHUNK of consecutive positions of the item-sorted order (fr_focf_prepare_step's
//            sort), so the members of an item sit side by side: user rows loaded, replayed (two per wave, packed), the item
//            row of a run replayed ONCE per chunk by the wave of its first member and handed to the others through LDS;
//            dot products; the caught-up rows and the scores are parked (write-through) for whoever finishes the item.
//            One arrival per (chunk, item) on the item's counter AFTER a workgroup barrier: nobody waits for anybody.
//   stage 2  the chunk whose arrival completes an item's count finishes it WITH ALL ITS WAVES: wave 0 forms the per-group
//            sums in focf_fair_kernel's lane order (64 lanes, butterfly: the same bits), every wave then takes every
//            RUN_WAVES-th member -- dLoss/dpred, the member's user row updated and stored (a user that occurs under several
//            items goes through a second arrival counter and is finished by its last arriver, gradients in ascending batch
//            position) -- and leaves the member's caught-up user row in LDS; wave 0 sums the item's gradient from LDS in
//            ascending batch position (product rounded, then added: the reference's accumulation order), applies Adam and
//            stores the item row.  Long runs go through LDS in passes of RUN_CAP members.
//
// Every sum has the chain's order and nothing depends on who arrives last: a step is bit-reproducible, and it equals the
// three-launch chain's except where a row's replay is cut into two stretches at another step than there (the moments are
// rescaled at a cut: a few ulp; tests/test_focf_hip.py: modes `runs*` of the goldens, test_runs_step_*).  The sweeper slice and
// the previous step's loss reduction ride in the same launch as in focf_step_kernel.
#include <stddef.h>
#include <stdlib.h>

#include <algorithm>

#include "common.hpp"
#include "kernels.hpp"
#include "table.hpp"
#include "focf_ws.hpp"
#include "focf_loss.hpp"

namespace fr {

#ifndef FR_RUN_WAVES
#define FR_RUN_WAVES 16     // 8: 62 us per step, 16: 54-56, 4: 86 (BASELINE sizes, item-complete batches; see the kernel)
#endif
constexpr int RUN_WAVES = FR_RUN_WAVES;
constexpr int RUN_THREADS = 64 * RUN_WAVES;
#ifndef FR_RUN_CAP_BYTES
#define FR_RUN_CAP_BYTES 24576        // LDS of a pass of stage 2: RUN_CAP caught-up user rows (+ their coefficients)
#endif

__host__ __device__ constexpr int run_pw(int E) { return E <= 1 ? 2 : 1; }                  // members per wave in stage 1
__host__ __device__ constexpr int run_chunk(int E) { return RUN_WAVES * run_pw(E); }        // members per workgroup
__host__ __device__ constexpr int run_cap(int E) { return FR_RUN_CAP_BYTES / (64 * E * 4); }  // members per stage-2 pass

struct RunArgs {
    TableV U, I;
    AdamC c;
    int B, objective;
    float fair_weight;
    FocfWs w;
    SweepSlice sw;
    int n_chunks;
    long long n_sweep_waves;
    uint32_t* err;
    PrevLoss prev;
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

// Stage 2 of one item segment, by all RUN_WAVES waves of the workgroup whose arrival completed it.
template <int E>
__device__ __forceinline__ void run_finish_item(const RunArgs& a, int k, int ij0, int n, float* pu_s, float* coef_s,
                                                float* sh, int lane, int wv) {
    constexpr int CAP = run_cap(E);
    const FocfWs& w = a.w;
    const int D = a.U.D, step = a.U.step, B = a.B;
    const bool per_item = a.objective >= FR_FOCF_VALUE && a.objective <= FR_FOCF_OVER;
    const float smin = G(w.sst_minmax)[0];
    // (a) the per-(item, group) sums of focf.py:75-91 in focf_fair_kernel<64>'s order: lane `sub` takes members sub, sub + 64,
    //     ... one after the other, then the butterfly
    if (wv == 0) {
        float term = 0.f, g0 = 0.f, g1 = 0.f;
        if (per_item) {
            const float smax = G(w.sst_minmax)[1];
            float sp0 = 0.f, sp1 = 0.f, st0 = 0.f, st1_ = 0.f, n0 = 0.f, n1 = 0.f;
            bool bad = false;
            for (int j = ij0 + lane; j < ij0 + n; j += 64) {
                const int b = G(w.perm_i)[j];
                const int4 rc = ld4(w.rec, b);
                const float pr = ld1(w.pred + b), r = __int_as_float(rc.z), s = __int_as_float(rc.w);
                bad |= (s != smin && s != smax);
                if (s == smin) {
                    sp0 += pr; st0 += r; n0 += 1.f;
                } else {
                    sp1 += pr; st1_ += r; n1 += 1.f;
                }
            }
            if (bad && a.err) atomicOr(a.err, FR_DEV_ERR_SST_GROUPS);
            sp0 = group_sum<64>(sp0); sp1 = group_sum<64>(sp1);
            st0 = group_sum<64>(st0); st1_ = group_sum<64>(st1_);
            n0 = group_sum<64>(n0);   n1 = group_sum<64>(n1);
            focf_fair_eval(a.objective, a.fair_weight, (float)G(w.nseg_i)[0], sp0, sp1, st0, st1_, n0, n1, term, g0, g1);
        }
        if (lane == 0) {
            sh[0] = g0;
            sh[1] = g1;
            G(w.term)[k] = term;
        }
    }
    __syncthreads();
    const float g0 = sh[0], g1 = sh[1];
    // the item's caught-up row as stage 1 parked it (every chunk of the run parked the same bits)
    RowFrag<E> pi;
    load_row1<E>(pi, w.side[3] + (size_t)k * D, D, lane);
    const float2 sc = step_scalars(a.c, step);
    RowFrag<E> gi;
#pragma unroll
    for (int e = 0; e < E; ++e) gi.x[e] = 0.f;
    for (int base = 0; base < n; base += CAP) {
        const int cnt = min(CAP, n - base);
        // (b) the members of this pass, every RUN_WAVES-th one per wave.  Their records come with ONE gather per array (lane t
        //     holds the wave's t-th member), the parked rows of four members are in flight at a time: a member costs the wave
        //     a share of two dependent round trips, not five of its own
        const int mine = cnt > wv ? (cnt - wv + RUN_WAVES - 1) / RUN_WAVES : 0;
        int lb = 0;
        int4 lrc = make_int4(0, 0, 0, 0), linf = make_int4(0, 0, 0, 0);
        float lpr = 0.f;
        if (lane < mine) {
            lb = G(w.perm_i)[ij0 + base + wv + RUN_WAVES * lane];
            lrc = ld4(w.rec, lb);
            linf = ld4(w.info, lb);
            lpr = ld1(w.pred + lb);
        }
        constexpr int MB = 4;
        for (int t0 = 0; t0 < mine; t0 += MB) {
            RowFrag<E> pu[MB], mu[MB], vu[MB];
            int bq[MB];
#pragma unroll
            for (int u = 0; u < MB; ++u) {
                bq[u] = 0;
                if (t0 + u < mine) {
                    bq[u] = __builtin_amdgcn_readlane(lb, t0 + u);
                    load_row1<E>(pu[u], w.side[0] + (size_t)bq[u] * D, D, lane);
                    load_row1<E>(mu[u], w.side[1] + (size_t)bq[u] * D, D, lane);
                    load_row1<E>(vu[u], w.side[2] + (size_t)bq[u] * D, D, lane);
                }
            }
#pragma unroll
            for (int u = 0; u < MB; ++u) {
                if (t0 + u >= mine) break;
                const int t = t0 + u, q = wv + RUN_WAVES * t, b = bq[u];
                const int ur = __builtin_amdgcn_readlane(lrc.x, t);
                const float r = __int_as_float(__builtin_amdgcn_readlane(lrc.z, t));
                const float s = __int_as_float(__builtin_amdgcn_readlane(lrc.w, t));
                const int ux = __builtin_amdgcn_readlane(linf.x, t), useg = __builtin_amdgcn_readlane(linf.y, t);
                const float pr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lpr), t));
                const float er = pr - r;
                float coef = 2.f * er / (float)B;                 // d mean((pred - r)^2) / d pred, as focf_gather_kernel forms it
                if (per_item) coef = coef + ((s == smin) ? g0 : g1);  // ... + the fairness part, as focf_fair_kernel adds it
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
                    // the user occurs under several items: hand dLoss/dpred over, arrive; the last arriver sums the user's
                    // gradient rows in ascending batch position (segment_grad_sum's order) from the items' parked rows
                    if (lane == 0) st1(w.coef + b, coef);
                    drain();
                    unsigned old = 0;
                    if (lane == 0) old = __hip_atomic_fetch_add(G(w.cnt_u) + useg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    old = (unsigned)uniform((int)old);
                    finish = (int)old + 1 == nu;
                    if (finish) {
                        // the caught-up state of the user's FIRST member, as segment_update takes it: every occurrence
                        // replayed the row beside another neighbour (another cut of the replay: a few ulp apart), and which of
                        // them arrives last must not show in the result
                        const int bf = uniform(G(w.perm_u)[uj0]);
                        if (bf != b) {
                            load_row1<E>(pu[u], w.side[0] + (size_t)bf * D, D, lane);
                            load_row1<E>(mu[u], w.side[1] + (size_t)bf * D, D, lane);
                            load_row1<E>(vu[u], w.side[2] + (size_t)bf * D, D, lane);
                        }
                        for (int ju = uj0; ju < uj0 + nu; ++ju) {
                            const int bb = uniform(G(w.perm_u)[ju]);
                            const float cb = ld1(w.coef + bb);
                            const int kk = uniform(ld4(w.info, bb).w);
                            RowFrag<E> o;
                            load_row1<E>(o, w.side[3] + (size_t)kk * D, D, lane);
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
                    gstore_row<E>(pu[u], a.U.p + (size_t)ur * D, D, lane);
                    gstore_row<E>(mu[u], a.U.m + (size_t)ur * D, D, lane);
                    gstore_row<E>(vu[u], a.U.v + (size_t)ur * D, D, lane);
                    if (lane == 0) G(a.U.last)[ur] = step;
                }
            }
        }
        __syncthreads();
        // (c) the item's gradient over this pass, members in ascending batch position (= ascending sorted position: the
        //     sort is stable), product rounded, then added
        if (wv == 0) {
#pragma clang fp contract(off)
            for (int q = 0; q < cnt; ++q) {
                const float cq = coef_s[q];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const float prod = cq * pu_s[(size_t)q * (64 * E) + lane + 64 * e];
                    gi.x[e] = gi.x[e] + prod;
                }
            }
        }
        __syncthreads();
    }
    if (wv == 0) {
        RowFrag<E> mi, vi;
        load_row1<E>(mi, w.side[4] + (size_t)k * D, D, lane);
        load_row1<E>(vi, w.side[5] + (size_t)k * D, D, lane);
        const int ir = uniform(G(w.seg_row_i)[k]);
#pragma unroll
        for (int e = 0; e < E; ++e) adam_elem(pi.x[e], mi.x[e], vi.x[e], gi.x[e], sc.x, sc.y, a.c);
        gstore_row<E>(pi, a.I.p + (size_t)ir * D, D, lane);
        gstore_row<E>(mi, a.I.m + (size_t)ir * D, D, lane);
        gstore_row<E>(vi, a.I.v + (size_t)ir * D, D, lane);
        if (lane == 0) G(a.I.last)[ir] = step;
    }
}

template <int E>
__global__ __launch_bounds__(RUN_THREADS) void focf_runs_kernel(RunArgs a_kernarg) {
    const RunArgs a = args_through_vgprs<RunArgs>((int)(threadIdx.x & 63));       // (the ONLY kernel argument: offset 0)
    constexpr int PW = run_pw(E), C = run_chunk(E), CAP = run_cap(E);
    constexpr int ROW = 64 * E;
    // LDS: stage 1 = the caught-up item rows of the chunk's leads [C][ROW]; stage 2 = a pass of user rows [CAP][ROW] + coefs
    __shared__ __align__(16) float lds_rows[(CAP > C ? CAP : C) * ROW];
    __shared__ float coef_s[CAP];
    __shared__ float sh[4];
    __shared__ int fin[C + 1];           // item segments this workgroup has to finish: (k, j0 | n << 16) pairs follow
    __shared__ int fin_arg[C];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const FocfWs& w = a.w;
    int blk = (int)blockIdx.x;
    if (blk == 0) {                      // the previous step's loss (fixed order), if one is waiting
        if (a.prev.loss_out) {
            if (threadIdx.x >= 256) return;
            step_reduce_loss<256>(a.prev);
        }
        return;
    }
    blk -= 1;
    // Chunk workgroups first, sweeper workgroups behind them (FR_RUN_INTERLEAVE=1 alternates the two kinds instead).  Measured
    // at the BASELINE sizes, item-complete batches, hipGraph replay: chunk-first 62 us per step with 8 waves per chunk and 54-56
    // with 16; alternating 84 -- at 75-89 VGPRs only one or two of these workgroups fit a CU, so every sweeper workgroup dealt
    // in early pushes a chunk workgroup (a long chain of dependent round trips) to a later round.
    {
        const int n_sw = (int)((a.n_sweep_waves + RUN_WAVES - 1) / RUN_WAVES);
        bool sweeper;
        int idx;
#if defined(FR_RUN_INTERLEAVE)
        const int both = min(a.n_chunks, n_sw);
        if (blk < 2 * both) {
            sweeper = (blk & 1) != 0;
            idx = blk >> 1;
        } else {
            sweeper = n_sw > a.n_chunks;
            idx = blk - both;
        }
#else
        (void)n_sw;
        sweeper = blk >= a.n_chunks;
        idx = sweeper ? blk - a.n_chunks : blk;
#endif
        if (sweeper) {                   // the step's slice of the bounded-staleness sweep: one wave per pair of rows
            const long long wave = (long long)idx * RUN_WAVES + wv;
            if (wave < a.n_sweep_waves) sweep_slice_wave<E>(a.U, a.I, a.c, a.sw, wave, lane);
            return;
        }
        blk = idx;
    }
    const int D = a.U.D, step = a.U.step, B = a.B;
    const int cstart = blk * C, cend = min(B, cstart + C);
    if (threadIdx.x == 0) fin[0] = 0;
    // ---- stage 1: this wave's members (sorted positions j0w .. j0w + PW) ---------------------------------------------------
    int bpos[PW], ijs[PW], ins[PW], iks[PW], irs[PW], urs[PW];
    bool val[PW], lead[PW];
    RowFrag<E> pu[PW], mu[PW], vu[PW];
    int tu[PW];
    float rat[PW];
#pragma unroll
    for (int t = 0; t < PW; ++t) {
        const int j = cstart + wv * PW + t;
        val[t] = j < cend;
        bpos[t] = 0; ijs[t] = 0; ins[t] = 0; iks[t] = 0; irs[t] = 0; urs[t] = 0; tu[t] = step - 1; rat[t] = 0.f;
        lead[t] = false;
        if (val[t]) {
            const int b = uniform(G(w.perm_i)[j]);
            const int4 rc = uni4(ld4(w.rec, b));
            const int4 inf = uni4(ld4(w.info, b));
            bpos[t] = b;
            urs[t] = rc.x;
            irs[t] = rc.y;
            rat[t] = __int_as_float(rc.z);
            ijs[t] = inf.z & 0xffff;
            ins[t] = (inf.z >> 16) & 0xffff;
            iks[t] = inf.w;
            lead[t] = j == max(ijs[t], cstart);      // the first member of its item inside this chunk
        }
    }
#pragma unroll
    for (int t = 0; t < PW; ++t) {
        if (val[t]) {
            const int lu = G(a.U.last)[urs[t]];
            gload_row<E>(pu[t], a.U.p + (size_t)urs[t] * D, D, lane);
            gload_row<E>(mu[t], a.U.m + (size_t)urs[t] * D, D, lane);
            gload_row<E>(vu[t], a.U.v + (size_t)urs[t] * D, D, lane);
            tu[t] = uniform(lu);
        }
    }
    // the item rows of the runs that begin (inside this chunk) at one of this wave's members: loaded, replayed, parked and
    // left in LDS for the other members of the chunk
#pragma unroll
    for (int t = 0; t < PW; ++t) {
        if (val[t] && lead[t]) {
            RowFrag<E> pi, mi, vi;
            const int li = G(a.I.last)[irs[t]];
            gload_row<E>(pi, a.I.p + (size_t)irs[t] * D, D, lane);
            gload_row<E>(mi, a.I.m + (size_t)irs[t] * D, D, lane);
            gload_row<E>(vi, a.I.v + (size_t)irs[t] * D, D, lane);
            replay<E>(pi, mi, vi, uniform(li), step - 1, a.c, lane);
            const int slot = wv * PW + t;
#pragma unroll
            for (int e = 0; e < E; ++e) lds_rows[slot * ROW + lane + 64 * e] = pi.x[e];
            store_row1<E>(pi, w.side[3] + (size_t)iks[t] * D, D, lane);
            store_row1<E>(mi, w.side[4] + (size_t)iks[t] * D, D, lane);
            store_row1<E>(vi, w.side[5] + (size_t)iks[t] * D, D, lane);
        }
    }
    // the user rows: the staler one alone up to the other's step, then both interleaved (packed replay)
    if (PW == 2) {
        if (val[0] && val[PW - 1]) {
            const int t0 = tu[0], t1 = tu[PW - 1];
            if (t0 < t1) replay<E>(pu[0], mu[0], vu[0], t0, t1, a.c, lane);
            else if (t1 < t0) replay<E>(pu[PW - 1], mu[PW - 1], vu[PW - 1], t1, t0, a.c, lane);
            replay2<E>(pu[0], mu[0], vu[0], pu[PW - 1], mu[PW - 1], vu[PW - 1], t0 > t1 ? t0 : t1, step - 1, a.c, lane);
        } else if (val[0]) {
            replay<E>(pu[0], mu[0], vu[0], tu[0], step - 1, a.c, lane);
        }
    } else if (val[0]) {
        replay<E>(pu[0], mu[0], vu[0], tu[0], step - 1, a.c, lane);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < PW; ++t) {
        if (val[t]) {
            const int slot = max(ijs[t], cstart) - cstart;
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) dot = fmaf(pu[t].x[e], lds_rows[slot * ROW + lane + 64 * e], dot);
            dot = wave_sum(dot);
            const int b = bpos[t];
            store_row1<E>(pu[t], w.side[0] + (size_t)b * D, D, lane);
            store_row1<E>(mu[t], w.side[1] + (size_t)b * D, D, lane);
            store_row1<E>(vu[t], w.side[2] + (size_t)b * D, D, lane);
            if (lane == 0) {
                const float er = dot - rat[t];
                st1(w.pred + b, dot);
                G(w.mse_e)[b] = er * er;
            }
        }
    }
    drain();
    __syncthreads();          // every member of the chunk is parked: now the chunk may arrive at its items' counters
#pragma unroll
    for (int t = 0; t < PW; ++t) {
        if (val[t] && lead[t] && lane == 0) {
            const int in_chunk = min(ijs[t] + ins[t], cend) - max(ijs[t], cstart);
            const unsigned old = __hip_atomic_fetch_add(G(w.cnt_i) + iks[t], (unsigned)in_chunk, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
            if ((int)old + in_chunk == ins[t]) {
                const int slot = atomicAdd(&fin[0], 1);
                fin[1 + slot] = iks[t];
                fin_arg[slot] = ijs[t] | (ins[t] << 16);
            }
        }
    }
    __syncthreads();
    // ---- stage 2: the items this chunk completed, one after the other, all waves together ----------------------------------
    const int nfin = fin[0];
    for (int f = 0; f < nfin; ++f) {
        // (ascending segment index: the order in which leads reached the LDS counter must not matter -- it does not for the
        // results, every item is independent; the loop is merely made deterministic)
        int kmin = 0x7fffffff, arg = 0;
        for (int q = 0; q < nfin; ++q) {
            const int kq = fin[1 + q];
            if (kq < kmin && kq >= 0) { kmin = kq; arg = fin_arg[q]; }
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int q = 0; q < nfin; ++q)
                if (fin[1 + q] == kmin) fin[1 + q] = -1;
#if !defined(FR_RUN_SKIP_STAGE2)      // (diagnostic builds only: what stage 1 costs alone)
        run_finish_item<E>(a, kmin, arg & 0xffff, (arg >> 16) & 0xffff, lds_rows, coef_s, sh, lane, wv);
#endif
        __syncthreads();
    }
}

}  // namespace fr

using namespace fr;

// One optimizer step of FOCF on an item-complete batch; same contract as fr_focf_step (the batch was prepared by
// fr_focf_prepare_step with `stamp`, table.step = the step being applied, the loss is reduced by the next launch or by
// fr_focf_step_finish).
extern "C" int fr_focf_step_runs(const fr_table* U, const fr_table* I, const fr_adam* adam, const float* sst, int64_t B,
                                 int32_t objective, float fair_weight, int32_t sweep_period, int32_t stamp, void* ws,
                                 size_t ws_bytes, void* prev_ws, int64_t prev_B, float* prev_loss_out, float* loss_acc,
                                 uint32_t* err_flag, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    if ((rc = check_table(U, "fr_focf_step_runs(U)")) || (rc = check_table(I, "fr_focf_step_runs(I)")) ||
        (rc = check_adam(adam, "fr_focf_step_runs")))
        return rc;
    FR_CHECK_ARG(U->dim == I->dim, "fr_focf_step_runs: user dim %d != item dim %d", U->dim, I->dim);
    FR_CHECK_ARG(ws, "fr_focf_step_runs: null pointer");
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
    long long sweep_blocks = 0;
    if (sweep_period > 0) {
        a.sw = make_sweep_slice(U, I, sweep_period);
        a.sw.skip_from = stamp;       // the rows of this batch (and of batches prepared for later steps) carry stamps >= it
        a.n_sweep_waves = sweep_slice_waves(a.sw);
        sweep_blocks = (a.n_sweep_waves + RUN_WAVES - 1) / RUN_WAVES;
    }
    a.prev = prev_of(prev_ws, prev_B, U->dim, objective, fair_weight, prev_loss_out, loss_acc, false);
    const int E = (U->dim + 63) / 64;
    a.n_chunks = (int)((B + run_chunk(E) - 1) / run_chunk(E));
    ProfScope prof(K_FOCF_STEP, stream);
    const unsigned blocks = (unsigned)(1 + a.n_chunks + sweep_blocks);
    FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_runs_kernel<E>), dim3(blocks), dim3(RUN_THREADS), 0, stream, a));
    FR_CHECK_LAUNCH();
    return FR_OK;
}
