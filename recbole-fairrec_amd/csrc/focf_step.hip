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
#include <stddef.h>
#include <stdlib.h>

#include <algorithm>

#include "common.hpp"
#include "kernels.hpp"
#include "table.hpp"
#include "focf_ws.hpp"
#include "focf_loss.hpp"

namespace fr {

#ifndef FR_LPT_SHARED_FIRST
#define FR_LPT_SHARED_FIRST 1
#endif
#ifndef FR_STEP_WPB
#define FR_STEP_WPB 4        // waves (= tasks) per workgroup: the granule the hardware dispatcher hands to a CU
#endif
constexpr int STEP_WPB = FR_STEP_WPB;

// Kernel arguments, kept to what the common path reads (every pointer is two SGPRs that stay live across the whole
// kernel; 256-thread workgroups are admitted 8 per CU only up to 80 SGPRs): the rare path derives the other workspace
// arrays from `ws` on the device.
struct StepArgs {
    float *Up, *Um, *Uv;
    int32_t *Ulast, *Ustamp;
    float *Ip, *Im, *Iv;
    int32_t *Ilast, *Istamp;
    int D, step;              // embedding size and the optimizer step being applied (the same for both tables)
    AdamC c;
    int B, objective;
    float fair_weight;
    const int4* task_rec;     // [B] (user row, item row, rating, sst) in start order, ids range-checked by the prepare
    const int4* task_info;    // [B] (user j0 | n << 16, item j0 | n << 16, user seg | item seg << 16, batch position)
    int lead;                 // interaction workgroups placed in front of the sweeper workgroups (the longest replays)
    const int32_t* hdr;       // (K = distinct items, -, smin, smax)
    float* mse_e;             // [B]
    float* term;              // [K]
    void* ws;
    long long lo_u, lo_i;     // sweep slice: rows [lo, lo + n) of each table, brought to `step`;
    int n_u, n_i, skip_from;  //   rows stamped >= skip_from belong to this or a coming batch and are left alone
    const int32_t* sw_order;  // [n_pairs + 1] start order of the slice's pairs of rows (longest replay first), then the
                              //   number of pairs it was built for; the identity order is used when that does not match
    uint32_t* err;
    unsigned long long *hcu, *hci;   // in-launch prepare: the row words (see "In-launch prepare") of the batch's generation
    PrevLoss prev;
};

// How a wave reads its kernel arguments: ONE vector load at kernel entry puts the whole argument block into two VGPRs
// (lane l holds dwords l and 64 + l of the kernarg segment); a field is then a v_readlane away wherever it is needed.
// As scalar loads from the kernarg segment the ~40 fields of the common path either stay live in SGPRs for the whole
// kernel (106 SGPRs + spills = 6 workgroups per CU) or are fetched just in time one after the other, each behind its own
// s_waitcnt: 7 us before an interaction wave had even requested its records (FR_STEP_TRACE).
struct KV {
    unsigned v0, v1;
};
template <typename T>
__device__ __forceinline__ T karg(KV kv, int off) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte argument fields");
    const int d = off >> 2;
    auto word = [&](int i) { return (unsigned)__builtin_amdgcn_readlane((int)(i < 64 ? kv.v0 : kv.v1), i & 63); };
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, word(d));
    } else {
        const unsigned long long q = (unsigned long long)word(d) | ((unsigned long long)word(d + 1) << 32);
        return __builtin_bit_cast(T, q);
    }
}
#define KA(field) karg<decltype(StepArgs::field)>(kv, (int)offsetof(StepArgs, field))
#define KAC(field) karg<decltype(AdamC::field)>(kv, (int)(offsetof(StepArgs, c) + offsetof(AdamC, field)))
static_assert(sizeof(StepArgs) <= 512, "the argument block must fit two VGPRs of dwords");

// Wave-uniform reads of per-task data (the prepared records, the rows' `last` stamps): vector loads + readfirstlane.
// (As scalar loads through the constant address space they cost 8 us per wave at launch: thousands of waves missing
// in the small scalar caches at once.)  Records needed again after the replay are re-read then, so that no register holds
// them across it.
__device__ __forceinline__ int4 uload4(const int4* p, int idx) {
    const int4 v = p[idx];
    return make_int4(uniform(v.x), uniform(v.y), uniform(v.z), uniform(v.w));
}
__device__ __forceinline__ int uload(const int32_t* p, long long idx) { return uniform(p[idx]); }

__device__ __forceinline__ float ld_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every store of this wave has left for memory (the storing wave's part of a hand-off; inline asm so that no compiler pass
// drops it)
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Row accesses of this kernel.  FULL: the embedding size is exactly 64 * E, so no lane is masked off (a masked access is
// a saveexec / branch / restore around every load and store) and D is a compile-time constant; the pointers are cast to
// the global address space, so that the compiler emits global_load / global_store with the row's base address in an SGPR
// pair (the kernel arguments come out of v_readlane as integers: as generic pointers every access is a flat_ one with a
// 64-bit per-lane address computed in VALU).
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) int32_t gint;
__device__ __forceinline__ const gfloat* gp(const float* p) { return (const gfloat*)p; }
__device__ __forceinline__ gfloat* gp(float* p) { return (gfloat*)p; }
__device__ __forceinline__ const gint* gp(const int32_t* p) { return (const gint*)p; }
__device__ __forceinline__ gint* gp(int32_t* p) { return (gint*)p; }
typedef int v4i_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4i_ gint4;
struct GInt4Ptr {       // a const int4 array read through the global address space
    const gint4* p;
    __device__ __forceinline__ int4 operator[](long long i) const {
        const v4i_ v = p[i];
        return make_int4(v.x, v.y, v.z, v.w);
    }
};
__device__ __forceinline__ GInt4Ptr gp(const int4* p) { return GInt4Ptr{(const gint4*)p}; }

__device__ __forceinline__ void st_sc1g(gfloat* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int E, bool FULL>
__device__ __forceinline__ void ldrow(RowFrag<E>& f, const float* base, int D, int lane) {
    const gfloat* g = gp(base);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane + 64 * e;
        f.x[e] = (FULL || d < D) ? g[d] : 0.f;
    }
}

template <int E, bool FULL>
__device__ __forceinline__ void load_row_sc1(RowFrag<E>& f, const float* base, int D, int lane) {
    const gfloat* g = gp(base);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane + 64 * e;
        f.x[e] = (FULL || d < D) ? __hip_atomic_load(g + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    }
}

template <int E, bool FULL>
__device__ __forceinline__ void store_row_sc1(const RowFrag<E>& f, float* base, int D, int lane) {
    gfloat* g = gp(base);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane + 64 * e;
        // (through an opaque register copy: the builtin reads its operand as an integer through memory, and that one access
        // of another type keeps the whole row structure `f` belongs to out of registers -- 16 bytes per lane of LDS or scratch)
        float val = f.x[e];
        asm volatile("" : "+v"(val));
        if (FULL || d < D) st_sc1g(g + d, val);
    }
}

// Store policy of finished table rows (FR_TROW_STORE: 1 write-through, 0 plain, 2 non-temporal) and of the tasks' one-word
// results (FR_WORD_STORE); what was measured is with the stage constants below.  (Until round 6 these defaults stood BEHIND
// their first use, where the preprocessor read the undefined names as 0: the product library stored plain, and the
// write-through numbers of round 3 were only ever seen in A/B builds that passed -D.  Moved here: 29.4 -> 27.3 us per step.)
#ifndef FR_TROW_STORE
#define FR_TROW_STORE 1
#endif
#ifndef FR_WORD_STORE
#define FR_WORD_STORE 1
#endif
// a finished row's `last` stamp, and the other one-word results of a task, with the rows' store policy
__device__ __forceinline__ void store_word(int32_t* p, int v) {
#if FR_TROW_STORE == 1 && FR_WORD_STORE
    __hip_atomic_store((gint*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *(gint*)p = v;
#endif
}
__device__ __forceinline__ void store_word(float* p, float v) {
#if FR_TROW_STORE == 1 && FR_WORD_STORE
    st_sc1g((gfloat*)p, v);
#else
    *(gfloat*)p = v;
#endif
}

// The write-back of a finished table row
template <int E, bool FULL, bool WT = true>     // WT: subject to FR_TROW_STORE (the shared-row path drains its stores
__device__ __forceinline__ void store_trow(const RowFrag<E>& f, float* base, int D, int lane) {      // between hand-offs: plain there)
    gfloat* g = gp(base);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane + 64 * e;
#if FR_TROW_STORE == 1
        if (WT) {
            if (FULL || d < D) { float val = f.x[e]; asm volatile("" : "+v"(val)); st_sc1g(g + d, val); }
        } else if (FULL || d < D) g[d] = f.x[e];
#elif FR_TROW_STORE == 2
        if (WT) {
            if (FULL || d < D) __builtin_nontemporal_store(f.x[e], g + d);
        } else if (FULL || d < D) g[d] = f.x[e];
#else
        if (FULL || d < D) g[d] = f.x[e];
#endif
    }
}

// ---- In-launch prepare -------------------------------------------------------------------------------------------
// What the step needs to know about a batch's ids before it can run -- which rows are shared by several interactions,
// by how many, who they are, K and the two sensitive values, a start order -- without a sort and without a second
// stream: every table row has a 64-bit WORD per generation (three generations: the batch being applied, the next one,
// the one after), and the stages below ride as a few extra workgroups in the step launches that precede the batch's own:
//   launch t - 2  CLAIM   one thread per interaction: max(word, tag) brings a word left by an older batch to
//                         (tag = the batch's stamp, count 0, base 0) -- stamps only grow, so no word is ever reset --
//                         then add(word, one member) returns the interaction's ARRIVAL RANK r at that row.  Rows are
//                         stamped for the sweeper.  K = interactions with item rank 0; the sensitive values by two
//                         ordered-int atomic maxima.
//   launch t - 1  PLACE   the counts are final: (n_u, n_i) of every interaction; the rank-0 member of a shared row
//                         reserves n slots of the batch's member list and adds their base to the word.  Start order:
//                         the interactions with a shared row (the longest chains of the launch) take places from the
//                         front of the task list, the others from its back, one atomic per wave and end -- a dense list
//                         without knowing how many there are of each kind, so the step maps task q to record q as it
//                         does after the sorted prepare (an order by replay length among the unshared ones, which the
//                         sorted prepare builds, measured no gain).
//                         The sweeper tasks of the stamped step ARE ordered by replay length, sixteen classes: class and
//                         rank within the class at the claim stage (one atomic per wave and class), place = class base +
//                         rank here, where the class counts are final.
//   launch t      the step: a member of a shared row writes its batch position at list[base + r], hands over as before
//                         and subtracts one member from the word: whoever brings the count to zero is the last arriver
//                         and reads the n members in one load, sorted ascending in registers (all sums of the shared path
//                         run in ascending batch position, so no result depends on the ranks or on who arrives last).
// Against the look-ahead sort on a side stream (fr_focf_prepare_step): no stream fork / join in the step loop and no
// 1024-thread sort workgroups running for 40 us beside the step launches (measured: 2.5-3 us per step at B = 8192).
constexpr int SW_NC = 16;     // cost classes of the sweeper order
constexpr int HC_TAG_SHIFT = 29;                       // word = stamp << 29 | members << 14 | base of the member list
constexpr unsigned long long HC_CNT1 = 1ull << 14;
__device__ __forceinline__ int hc_base(unsigned long long w) { return (int)(w & 0x3fffull); }
__device__ __forceinline__ int hc_cnt(unsigned long long w) { return (int)((w >> 14) & 0x7fffull); }
static_assert(FR_SORT_MAX <= (1 << 14), "a batch's member list is addressed with 14 bits");

// floats as unsigned integers of the same order (0 is below every float: the identity of an atomic maximum)
__device__ __forceinline__ unsigned ord_enc(float x) {
    const unsigned b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_dec(unsigned e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }

struct ClaimJob {        // a batch two steps ahead, and the sweeper tasks of the step it is stamped for
    const int64_t *user, *item;
    const float *rating, *sst;
    int4 *rec, *info;
    int32_t* cp;
    unsigned long long *hcu, *hci;
    int B, stamp;
    long long lo_u, lo_i;
    int n_u, n_i;
    int32_t* sw_tmp;        // nullptr: no start order for the sweeper tasks
};
struct PlaceJob {        // the batch of the next step
    int4 *rec, *info, *task_rec, *task_info;
    int32_t *cp, *hdr;
    unsigned long long *hcu, *hci;
    int B, n_pairs, stamp;
    const int32_t* sw_tmp;
    int32_t* sw_order;
};
struct StageArgs {
    ClaimJob c;
    PlaceJob p;
    int nb_claim, nb_sa, nb_place, nb_sb;      // workgroups of each stage (0 = the stage is not in this launch)
    int32_t *Ulast, *Ustamp, *Ilast, *Istamp;
    int n_rows_u, n_rows_i, cap;
    uint32_t* err;
};
constexpr int STAGE_THREADS = 64 * STEP_WPB;     // stage workgroups have the shape of the step launch's

// The wave's members of class k (k < 0: none) take consecutive places behind counters[k]: one atomic per class present,
// all of them in one round trip (lane c speaks for class c).
template <int NC>
__device__ __forceinline__ int class_rank(int k, int32_t* counters, int lane) {
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long m[NC];
    int mine = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        m[c] = __ballot(k == c);
        if (lane == c) mine = __popcll(m[c]);
    }
    int got = 0;
    if (lane < NC && mine) got = atomicAdd(counters + lane, mine);
    int rk = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int base = __shfl(got, c, 64);
        if (k == c) rk = base + __popcll(m[c] & lt);
    }
    return rk;
}

__device__ __forceinline__ int replay_class(int a, int b, int cap, int NC) {     // the cost classes of focf_lpt_kernel
    a = a < 0 ? 0 : (a > cap ? cap : a);
    b = b < 0 ? 0 : (b > cap ? cap : b);
    const int hi = a > b ? a : b, lo = a > b ? b : a;
    const int cost = 7 * hi + 2 * lo;                          // VALU instructions: alone 7, as a pair 9 per step
    return NC - 1 - min(NC - 1, cost * NC / (9 * cap + 1));
}

// Every stage thread takes STAGE_EPT elements, a block of STAGE_THREADS apart, with the loads and atomics of all of them in
// flight together: half the stage waves (each holds a wave slot for 7-19 us of dependent round trips that a sweeper task
// could have started in), chains of the same length.  Measured (uniform / zipf items, us per step): one element per thread
// everywhere 29.97 / 33.7; two, the place stage one 29.61 / 33.8 (kept); two everywhere 29.61 / 34.2; four 33-43.
// Finished table rows (and the tasks' one-word results) are stored WRITE-THROUGH (1; 0 plain, 2 non-temporal): a launch writes
// 22 MB, the eight L2s together hold 32 MB, so with write-back stores most of it is still dirty when the last wave ends and
// the launch's end-of-kernel release writes it out before the next launch may start -- 3.2 us from the last wave to the next
// launch's first against ~1.5 us for a launch that leaves nothing behind.  Measured (same box, us per step, uniform / zipf /
// --steps 20): plain 29.98 / 33.7 / 30.8, rows write-through 29.41 / 33.6 / 29.9, rows and words 29.05 / 33.4, non-temporal
// 29.8.  Not on the shared-row path: its hand-offs drain the wave's stores, and a write-through store is acknowledged by
// memory, not by the L2 (zipf items 33.6 -> 39.4 us with write-through rows there).
#ifndef FR_STAGE_EPT
#define FR_STAGE_EPT 2
#endif
constexpr int STAGE_EPT = FR_STAGE_EPT;
constexpr int STAGE_BLOCK = STAGE_THREADS * STAGE_EPT;       // elements per stage workgroup

__device__ __forceinline__ void stage_claim(const StageArgs& s, int sb) {
    const ClaimJob& J = s.c;
    const int lane = threadIdx.x & 63;
    constexpr int EPT = STAGE_EPT;
    int b[EPT];
    bool ok[EPT];
    long long u[EPT], i[EPT];
    float rt[EPT], ss[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        b[e] = (sb * EPT + e) * STAGE_THREADS + (int)threadIdx.x;
        ok[e] = b[e] < J.B;
        const int bc = ok[e] ? b[e] : 0;
        u[e] = J.user[bc];
        i[e] = J.item[bc];
        rt[e] = J.rating[bc];
        ss[e] = J.sst ? J.sst[bc] : 0.f;
    }
    bool bad = false;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        bad |= ok[e] && (u[e] < 0 || u[e] >= s.n_rows_u || i[e] < 0 || i[e] >= s.n_rows_i);
        if (u[e] < 0 || u[e] >= s.n_rows_u) u[e] = 0;
        if (i[e] < 0 || i[e] >= s.n_rows_i) i[e] = 0;
    }
    if (bad && s.err) atomicOr(s.err, FR_DEV_ERR_INDEX_RANGE);
    const unsigned long long T = (unsigned long long)J.stamp << HC_TAG_SHIFT;
    unsigned long long mu[EPT], mi[EPT], ou[EPT], oi[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        mu[e] = mi[e] = 0ull;
        if (ok[e]) {
            mu[e] = __hip_atomic_fetch_max(J.hcu + u[e], T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mi[e] = __hip_atomic_fetch_max(J.hci + i[e], T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicMax(s.Ustamp + u[e], J.stamp);
            atomicMax(s.Istamp + i[e], J.stamp);
        }
    }
    // (an add is issued only once its row's maximum has returned: its operand depends on the returned word -- on a bit that
    // is never set, a stamp has 31 bits)
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        ou[e] = oi[e] = 0ull;
        if (ok[e]) {
            ou[e] = __hip_atomic_fetch_add(J.hcu + u[e], HC_CNT1 + (mu[e] >> 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            oi[e] = __hip_atomic_fetch_add(J.hci + i[e], HC_CNT1 + (mi[e] >> 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    int first = 0;
    unsigned e1 = 0u, e2 = 0u;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int ru = hc_cnt(ou[e]), ri = hc_cnt(oi[e]);
        if (ok[e]) {
            J.rec[b[e]] = make_int4((int)u[e], (int)i[e], __float_as_int(rt[e]), __float_as_int(ss[e]));
            J.info[b[e]] = make_int4(ru, ri, 0, 0);
            e1 = max(e1, ord_enc(ss[e]));
            e2 = max(e2, ord_enc(-ss[e]));
        }
        first += __popcll(__ballot(ok[e] && ri == 0));
    }
    if (lane == 0 && first) atomicAdd(J.cp + 8, first);
    if (J.sst) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            e1 = max(e1, (unsigned)__shfl_xor((int)e1, o, 64));
            e2 = max(e2, (unsigned)__shfl_xor((int)e2, o, 64));
        }
        if (lane == 0) {
            atomicMax(reinterpret_cast<unsigned*>(J.cp + 9), e1);
            atomicMax(reinterpret_cast<unsigned*>(J.cp + 10), e2);
        }
    }
}

// class and rank of the sweeper tasks of the step the claimed batch is stamped for (sweep_order_body's cost)
__device__ __forceinline__ void stage_sweep_class(const StageArgs& s, int sb) {
    const ClaimJob& J = s.c;
    const int lane = threadIdx.x & 63;
    constexpr int EPT = STAGE_EPT;
    const int pairs_u = (J.n_u + 1) >> 1, n_pairs = pairs_u + ((J.n_i + 1) >> 1);
    const int cap = s.cap > 0 ? s.cap : 1024;
    int q[EPT], la[EPT], lb[EPT], sa[EPT], sb_[EPT], k[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        q[e] = (sb * EPT + e) * STAGE_THREADS + (int)threadIdx.x;
        const int qc = q[e] < n_pairs ? q[e] : 0;
        const bool inU = qc < pairs_u;
        const int kk = inU ? qc : qc - pairs_u;
        const long long rowA = (inU ? J.lo_u : J.lo_i) + 2 * kk;
        const long long rowB = 2 * kk + 1 < (inU ? J.n_u : J.n_i) ? rowA + 1 : rowA;
        const int32_t* Tl = inU ? s.Ulast : s.Ilast;
        const int32_t* Ts = inU ? s.Ustamp : s.Istamp;
        la[e] = Tl[rowA]; lb[e] = Tl[rowB]; sa[e] = Ts[rowA]; sb_[e] = Ts[rowB];
    }
    int mine = 0;          // lane c: the wave's tasks of class c
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        k[e] = q[e] < n_pairs ? replay_class(sa[e] >= J.stamp ? 0 : J.stamp - la[e], sb_[e] >= J.stamp ? 0 : J.stamp - lb[e],
                                            cap, SW_NC)
                              : -1;
#pragma unroll
        for (int c = 0; c < SW_NC; ++c) {
            const int n = __popcll(__ballot(k[e] == c));
            if (lane == c) mine += n;
        }
    }
    int got = 0;
    if (lane < SW_NC && mine) got = atomicAdd(J.cp + 16 + lane, mine);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int rk[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) rk[e] = 0;
#pragma unroll
    for (int c = 0; c < SW_NC; ++c) {
        int run = __shfl(got, c, 64);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const unsigned long long m = __ballot(k[e] == c);
            if (k[e] == c) rk[e] = run + __popcll(m & lt);
            run += __popcll(m);
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if (k[e] >= 0) J.sw_tmp[q[e]] = k[e] << 20 | rk[e];
}

#ifndef FR_PLACE_EPT
#define FR_PLACE_EPT 1
#endif
constexpr int PLACE_EPT = FR_PLACE_EPT;      // (the place stage ranks its wave's interactions: 64 EPT^2 compare steps)
__device__ __forceinline__ void stage_place(const StageArgs& s, int sb) {
    const PlaceJob& J = s.p;
    const int lane = threadIdx.x & 63;
    constexpr int EPT = PLACE_EPT;
    int b[EPT];
    bool ok[EPT];
    int4 rec[EPT], f[EPT];           // f = (rank at the user row, rank at the item row, -, -)
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        b[e] = (sb * EPT + e) * STAGE_THREADS + (int)threadIdx.x;
        ok[e] = b[e] < J.B;
        const int bc = ok[e] ? b[e] : 0;
        rec[e] = J.rec[bc];
        f[e] = J.info[bc];
    }
    int nu[EPT], ni[EPT], k[EPT], key[EPT];
    {
        unsigned long long wu[EPT], wi[EPT];
        int lu[EPT], li[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            wu[e] = J.hcu[rec[e].x];
            wi[e] = J.hci[rec[e].y];
            lu[e] = s.Ulast[rec[e].x];
            li[e] = s.Ilast[rec[e].y];
        }
        const int cap = s.cap > 0 ? s.cap : 1024;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            nu[e] = hc_cnt(wu[e]);
            ni[e] = hc_cnt(wi[e]);
            if (ok[e] && nu[e] > 1 && f[e].x == 0) {
                const int base = atomicAdd(J.cp + 11, nu[e]);
                __hip_atomic_fetch_add(J.hcu + rec[e].x, (unsigned long long)base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (ok[e] && ni[e] > 1 && f[e].y == 0) {
                const int base = atomicAdd(J.cp + 12, ni[e]);
                __hip_atomic_fetch_add(J.hci + rec[e].y, (unsigned long long)base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // a shared row is finished by the last of its waves to arrive, behind a hand-off through memory and three more
            // dependent load levels: the longest chain of the launch whatever its replay length, so those start first
            k[e] = !ok[e] ? -1 : ((nu[e] > 1 || ni[e] > 1) && FR_LPT_SHARED_FIRST ? 0 : 1);
            int cu = J.stamp - 1 - lu[e], ci = J.stamp - 1 - li[e];
            cu = cu < 0 ? 0 : (cu > cap ? cap : cu);
            ci = ci < 0 ? 0 : (ci > cap ? cap : ci);
            key[e] = 7 * (cu > ci ? cu : ci) + 2 * (cu > ci ? ci : cu);      // the cost of focf_lpt_kernel
        }
    }
    // Within the places its wave takes at either end, an interaction stands by estimated replay cost (longest towards the
    // front): the step gives two neighbours of the list to one wave and replays their user rows, and their item rows, as
    // packed pairs over the steps both rows of a pair missed -- rows of like staleness share most of them.  (By the exact
    // estimate: with the eight cost classes of the sorted prepare in its place a step takes 0.5 us longer.  The ranking is
    // 64 EPT^2 compare steps per wave, which is what bounds EPT: at 4 a place wave outlasts the launch.)
    int before[EPT];                     // members of the element's band in the wave that stand before it
#pragma unroll
    for (int e = 0; e < EPT; ++e) before[e] = 0;
    int n0 = 0, n1 = 0;                  // the wave's interactions of either band
#pragma unroll
    for (int e2 = 0; e2 < EPT; ++e2) {
        n0 += __popcll(__ballot(k[e2] == 0));
        n1 += __popcll(__ballot(k[e2] == 1));
        for (int t = 0; t < 64; ++t) {
            const int kt = __builtin_amdgcn_readlane(k[e2], t), ct = __builtin_amdgcn_readlane(key[e2], t);
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                before[e] += (kt == k[e] && (ct > key[e] || (ct == key[e] && (e2 < e || (e2 == e && t < lane))))) ? 1 : 0;
        }
    }
    // places taken from the front (cp[0]) and from the back (cp[1]): the wave's first place of either band
    int got = 0;
    if (lane == 0 && n0) got = atomicAdd(J.cp, n0);
    if (lane == 1 && n1) got = atomicAdd(J.cp + 1, n1);
    const int base0 = __shfl(got, 0, 64), base1 = __shfl(got, 1, 64);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        if (!ok[e]) continue;
        // front band: [base, base + n) by falling cost; back band: the wave's n places end at B - 1 - base, falling cost too
        const int pos = k[e] == 0 ? base0 + before[e] : J.B - 1 - (base1 + (n1 - 1 - before[e]));
        J.task_rec[pos] = rec[e];
        J.task_info[pos] = make_int4(f[e].x | nu[e] << 16, f[e].y | ni[e] << 16, 0, b[e]);   // the records' layout of the sorted
        J.info[b[e]] = make_int4(f[e].x, f[e].y, nu[e], ni[e]);                             //   prepare, ranks for list positions
    }
    if (sb == 0 && threadIdx.x == 0) {
        const unsigned e1 = (unsigned)J.cp[9], e2 = (unsigned)J.cp[10];
        *reinterpret_cast<int4*>(J.hdr) = make_int4(J.cp[8], 0, __float_as_int(-ord_dec(e2)), __float_as_int(ord_dec(e1)));
    }
}

__device__ __forceinline__ void stage_sweep_place(const StageArgs& s, int sb) {
    const PlaceJob& J = s.p;
    int cnt[SW_NC];
#pragma unroll
    for (int c = 0; c < SW_NC; ++c) cnt[c] = J.cp[16 + c];
    int v[STAGE_EPT];
#pragma unroll
    for (int e = 0; e < STAGE_EPT; ++e) {
        const int q = (sb * STAGE_EPT + e) * STAGE_THREADS + (int)threadIdx.x;
        v[e] = q < J.n_pairs ? J.sw_tmp[q] : -1;
    }
#pragma unroll
    for (int e = 0; e < STAGE_EPT; ++e) {
        const int q = (sb * STAGE_EPT + e) * STAGE_THREADS + (int)threadIdx.x;
        if (v[e] >= 0) {
            const int k = v[e] >> 20;
            int pos = v[e] & 0xfffff;
#pragma unroll
            for (int c = 0; c < SW_NC; ++c) pos += c < k ? cnt[c] : 0;
            J.sw_order[pos] = q;
        }
    }
    if (sb == 0 && threadIdx.x == 0) J.sw_order[J.n_pairs] = J.n_pairs;
}

// stage workgroup `sb` of a launch (the claim stages first: their atomics are the longest chains)
__device__ __forceinline__ void stage_block(const StageArgs& s, int sb) {
    if (sb < s.nb_claim) return stage_claim(s, sb);
    sb -= s.nb_claim;
    if (sb < s.nb_sa) return stage_sweep_class(s, sb);
    sb -= s.nb_sa;
    if (sb < s.nb_place) return stage_place(s, sb);
    sb -= s.nb_place;
    if (sb < s.nb_sb) stage_sweep_place(s, sb);
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long x) {
    return (unsigned long long)(unsigned)uniform((int)(unsigned)x) | ((unsigned long long)(unsigned)uniform((int)(x >> 32)) << 32);
}
__device__ __forceinline__ int ld_sc1_int(const int32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_int(int32_t* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one member leaves a row's word; returns the word as it was (count 1: this wave is the last arriver)
__device__ __forceinline__ unsigned long long hc_arrive(unsigned long long* p, int lane) {
    unsigned long long t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(p, 0ull - HC_CNT1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return uniform64(t);
}

// The members of a shared row as its last arriver sees them: batch positions in ascending order.  Up to 64 of them (every
// row of a uniform batch, all but the hottest of a skewed one) stay in registers -- one load, each member's rank by
// comparison with the others, one ds_permute; longer lists are ranked chunk against chunk and written out in order.
struct Members {
    int reg;                 // n <= 64: member `lane`
    const int32_t* sorted;   // n > 64: the sorted list
    int n;
};
__device__ __forceinline__ Members load_members(const int32_t* list, int32_t* sorted, int base, int n, int lane) {
    Members m;
    m.n = n;
    m.reg = 0;
    m.sorted = sorted + base;
    if (n <= 64) {
        const int v = lane < n ? ld_sc1_int(list + base + lane) : 0x7fffffff;
        int rank = 0;
        for (int t = 0; t < n; ++t) rank += __builtin_amdgcn_readlane(v, t) < v ? 1 : 0;
        m.reg = __builtin_amdgcn_ds_permute(rank << 2, v);       // (the lanes past n all push to lane n, which nobody reads)
        return m;
    }
    for (int jb = 0; jb < n; jb += 64) {
        const int v = jb + lane < n ? ld_sc1_int(list + base + jb + lane) : 0x7fffffff;
        int rank = 0;
        for (int jc = 0; jc < n; jc += 64) {
            const int x = jc + lane < n ? ld_sc1_int(list + base + jc + lane) : 0x7fffffff;
            const int cn = min(64, n - jc);
            for (int t = 0; t < cn; ++t) rank += __builtin_amdgcn_readlane(x, t) < v ? 1 : 0;
        }
        if (jb + lane < n) st_sc1_int(sorted + base + rank, v);
    }
    drain_stores();
    return m;
}
__device__ __forceinline__ int members_chunk(const Members& m, int jb, int lane) {     // member jb + lane (0 past the end)
    if (m.n <= 64) return lane < m.n ? m.reg : 0;
    int idx = jb + lane;
    asm volatile("" : "+v"(idx));      // (not a per-lane address kept in registers across the caller's loops)
    return idx < m.n ? ld_sc1_int(m.sorted + idx) : 0;
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

// Adam step `step` with data gradient g on a caught-up row held in registers; the row is written back once
template <int E, bool FULL>
__device__ __forceinline__ void adam_write(float* Tp, float* Tm, float* Tv, int32_t* Tlast, int D, int step,
                                           const AdamC& c, int row, RowFrag<E>& p, RowFrag<E>& m, RowFrag<E>& v,
                                           const RowFrag<E>& g, float2 s, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) adam_elem(p.x[e], m.x[e], v.x[e], g.x[e], s.x, s.y, c);
    store_trow<E, FULL, false>(p, Tp + (size_t)row * D, D, lane);
    store_trow<E, FULL, false>(m, Tm + (size_t)row * D, D, lane);
    store_trow<E, FULL, false>(v, Tv + (size_t)row * D, D, lane);
    if (lane == 0) Tlast[row] = step;
}

// g = sum over the members [j0, j0 + n) of a segment, in ascending batch position, of coef[b] * other[b, :] -- the product
// rounded, then added (embedding_dense_backward's accumulation order), as segment_grad_sum of table.hpp, but on values
// other waves of this launch handed over: sc1 loads throughout.  One member in flight (rare path: kept lean in registers).
template <int E, bool FULL, typename Chunk>     // chunk(jb) = batch position of member jb + lane (any value past the end)
__device__ __forceinline__ void handed_grad_sum(RowFrag<E>& g, int n, Chunk chunk, const float* coef, const float* other,
                                                int D, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) g.x[e] = 0.f;
    constexpr int UN = 2;
    for (int jb = 0; jb < n; jb += 64) {
        const int cnt = min(64, n - jb);
        int my_b = chunk(jb);
        float my_c = 0.f;
        if (lane >= cnt) my_b = 0;
        if (lane < cnt) my_c = ld_sc1(coef + my_b);
        for (int t0 = 0; t0 < cnt; t0 += UN) {
            RowFrag<E> o[UN];
            float cb[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const int t = t0 + q < cnt ? t0 + q : cnt - 1;     // tail: re-read the last member, not added
                const int b = __builtin_amdgcn_readlane(my_b, t);
                cb[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_c), t));
                load_row_sc1<E, FULL>(o[q], other + (size_t)b * D, D, lane);
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

// Two rows A and B, current as of steps tA and tB, brought to `upto` (t >= upto: nothing to do for that row): the steps
// only the staler row missed run on that row alone, the common tail on both rows as packed pairs.  ONE instance of this
// serves the sweeper (two neighbouring rows of the slice) and the interactions (the user row and the item row).
template <int E>
struct TwoRows {
    RowFrag<E> pA, mA, vA, pB, mB, vB;
};

template <int E>
__device__ __forceinline__ TwoRows<E> replay_two_v(TwoRows<E> r, int tA, int tB, int upto, const AdamC& c, int lane) {
    tA = tA < upto ? tA : upto;
    tB = tB < upto ? tB : upto;
    if (tA != tB) {     // wave-uniform
        const bool a_old = tA < tB;
        RowFrag<E> p, m, v;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            p.x[e] = a_old ? r.pA.x[e] : r.pB.x[e];
            m.x[e] = a_old ? r.mA.x[e] : r.mB.x[e];
            v.x[e] = a_old ? r.vA.x[e] : r.vB.x[e];
        }
        replay<E>(p, m, v, a_old ? tA : tB, a_old ? tB : tA, c, lane);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (a_old) {
                r.pA.x[e] = p.x[e]; r.mA.x[e] = m.x[e]; r.vA.x[e] = v.x[e];
            } else {
                r.pB.x[e] = p.x[e]; r.mB.x[e] = m.x[e]; r.vB.x[e] = v.x[e];
            }
        }
    }
    replay2<E>(r.pA, r.mA, r.vA, r.pB, r.mB, r.vB, tA > tB ? tA : tB, upto, c, lane);
    return r;
}
// (by value in, by value out: through a reference the fragments of E >= 2 stayed in scratch memory -- 40-88 bytes per lane)
template <int E>
__device__ __forceinline__ void replay_two(TwoRows<E>& r, int tA, int tB, int upto, const AdamC& c, int lane) {
    r = replay_two_v<E>(r, tA, tB, upto, c, lane);
}

// ---- rare path ----------------------------------------------------------------------------------------------------
// The last wave to arrive at a user segment: sum the members' gradient rows coef[b] * (item row of b before its update),
// one Adam step on the user's caught-up row (parked by its first member), write back.
template <int E, bool FULL>
__device__ __forceinline__ void user_finish(KV kv, const AdamC& c, const FocfWs& w, int j0u, int nu, float2 s, int lane) {
    const int D = FULL ? 64 * E : KA(D);
    const int c0 = uniform(w.perm_u[j0u]);
    const int ur = uniform(w.rec[c0].x);
    RowFrag<E> p, m, v, g;
    load_row_sc1<E, FULL>(p, w.side[0] + (size_t)c0 * D, D, lane);
    load_row_sc1<E, FULL>(m, w.side[1] + (size_t)c0 * D, D, lane);
    load_row_sc1<E, FULL>(v, w.side[2] + (size_t)c0 * D, D, lane);
    handed_grad_sum<E, FULL>(g, nu, [&](int jb) { return jb + lane < nu ? w.perm_u[j0u + jb + lane] : 0; }, w.coef, w.side[3],
                             D, lane);
    adam_write<E, FULL>(KA(Up), KA(Um), KA(Uv), KA(Ulast), D, KA(step), c, ur, p, m, v, g, s, lane);
}

// ... the same for a batch prepared in the launches before (row words and member lists instead of sorted segments)
template <int E, bool FULL>
__device__ __forceinline__ void user_finish_c(KV kv, const AdamC& c, const FocfWs& w, int ur, int base, int nu, float2 s,
                                              int lane) {
    const int D = FULL ? 64 * E : KA(D);
    const Members mem = load_members(w.perm_u, w.seg_start_u, base, nu, lane);
    const int c0 = __builtin_amdgcn_readlane(members_chunk(mem, 0, lane), 0);
    RowFrag<E> p, m, v, g;
    load_row_sc1<E, FULL>(p, w.side[0] + (size_t)c0 * D, D, lane);
    load_row_sc1<E, FULL>(m, w.side[1] + (size_t)c0 * D, D, lane);
    load_row_sc1<E, FULL>(v, w.side[2] + (size_t)c0 * D, D, lane);
    handed_grad_sum<E, FULL>(g, nu, [&](int jb) { return members_chunk(mem, jb, lane); }, w.coef, w.side[3], D, lane);
    adam_write<E, FULL>(KA(Up), KA(Um), KA(Uv), KA(Ulast), D, KA(step), c, ur, p, m, v, g, s, lane);
}

// The rest of an interaction whose user or item row is shared with other interactions of the batch: hand over, then
// whoever arrives last at a segment finishes it (3 % / 8 % of the interactions for uniform pairs at the BASELINE sizes).
// CLAIM: the batch was prepared in the launches before this one ("In-launch prepare"): iux / iix = the interaction's
// arrival rank at its user / item row | members << 16, the counters are the rows' words, the member lists are written
// here.  Otherwise (sorted prepare) iux / iix = first sorted position | members << 16 and seg_u / seg_i the segments.
template <int E, bool FULL, bool CLAIM>
__device__ __forceinline__ void step_shared_rows(KV kv, const AdamC& c, int b, int lane, int ur, int ir, int iux, int iix,
                                                 int seg_u, int seg_i, float dot, float coef, float smin, float smax,
                                                 float K, RowFrag<E>& pu, RowFrag<E>& mu, RowFrag<E>& vu, RowFrag<E>& pi,
                                                 RowFrag<E>& mi, RowFrag<E>& vi) {
    const int D = FULL ? 64 * E : KA(D);
    const bool fair = KA(objective) != FR_FOCF_NONE;
    const int nu = iux >> 16, ni = iix >> 16, j0u = iux & 0xffff, j0i = iix & 0xffff;
    const float2 s = step_scalars(c, KA(step));
    // the workspace arrays of this path, from the base pointer (opaque to the optimiser on purpose: hoisted out of the
    // task loop the two dozen pointers would occupy SGPRs on the common path)
    int Bq = KA(B);
    asm volatile("" : "+s"(Bq));
    const FocfWs w = focf_layout(KA(ws), Bq, D);

    if constexpr (CLAIM) {
        // this member's entry in the member list of each shared row it belongs to (base of the list: in the row's word)
        unsigned long long wu = 0, wi = 0;
        if (nu > 1) wu = __hip_atomic_load(KA(hcu) + ur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ni > 1) wi = __hip_atomic_load(KA(hci) + ir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
            if (nu > 1) st_sc1_int(w.perm_u + hc_base(wu) + j0u, b);
            if (ni > 1) st_sc1_int(w.perm_i + hc_base(wi) + j0i, b);
        }
    }
    const size_t so = (size_t)b * D;
    store_row_sc1<E, FULL>(pu, w.side[0] + so, D, lane);
    store_row_sc1<E, FULL>(mu, w.side[1] + so, D, lane);
    store_row_sc1<E, FULL>(vu, w.side[2] + so, D, lane);
    if (nu > 1) store_row_sc1<E, FULL>(pi, w.side[3] + so, D, lane);     // the item row BEFORE its update: users' gradients
    if (ni == 1) {
        // item level is this wave alone; the user has other members
        RowFrag<E> gi;
#pragma unroll
        for (int e = 0; e < E; ++e) gi.x[e] = coef * pu.x[e];
        adam_write<E, FULL>(KA(Ip), KA(Im), KA(Iv), KA(Ilast), D, KA(step), c, ir, pi, mi, vi, gi, s, lane);
        if (lane == 0) st_sc1(w.coef + b, coef);
        drain_stores();
        if constexpr (CLAIM) {
            const unsigned long long old = hc_arrive(KA(hcu) + ur, lane);
            if (hc_cnt(old) == 1) user_finish_c<E, FULL>(kv, c, w, ur, hc_base(old), nu, s, lane);
        } else {
            if (arrive_last(w.cnt_u + seg_u, nu, lane)) user_finish<E, FULL>(kv, c, w, j0u, nu, s, lane);
        }
        return;
    }
    if (lane == 0) {
        st_sc1(w.pred + b, dot);
        if (CLAIM && fair) st_sc1(w.term + b, 0.f);      // the item's term goes to its first member's place (below)
    }
    drain_stores();
    Members mem{};
    if constexpr (CLAIM) {
        const unsigned long long old = hc_arrive(KA(hci) + ir, lane);
        if (hc_cnt(old) != 1) return;
        mem = load_members(w.perm_i, w.seg_start_i, hc_base(old), ni, lane);
    } else {
        if (!arrive_last(w.cnt_i + seg_i, ni, lane)) return;
    }

    // ---- item level, last arriver.  THREE dependent load levels for the whole segment (it is the longest chain of the
    // launch): (1) the members' batch positions, one per lane; (2) their records and scores; (3) their parked user rows,
    // UN members in flight.  Statistics, dLoss/dpred, the item row's gradient and the members' user updates are formed in
    // registers in between -- nothing this wave needs again goes through memory.
    const bool big = ni > 64;       // more than one chunk of 64 members: the chunks are loaded again for each phase
    int my_b = 0, my_u = 0, my_iux = 0, my_seg = 0;     // my_iux: (sorted prepare) first position | members << 16 of the
    float my_pr = 0.f, my_rt = 0.f, my_s = 0.f;         //   member's user; (CLAIM) members << 16
    auto load_chunk = [&](int jb) {
        my_b = 0; my_u = 0; my_iux = 0; my_seg = 0;
        my_pr = 0.f; my_rt = 0.f; my_s = 0.f;
        int mb = 0;
        if constexpr (CLAIM) mb = members_chunk(mem, jb, lane);
        if (jb + lane < ni) {
            if constexpr (CLAIM) my_b = mb;
            else my_b = w.perm_i[j0i + jb + lane];
            const int4 rq = w.rec[my_b];
            const int4 fq = w.info[my_b];
            my_pr = ld_sc1(w.pred + my_b);
            my_u = rq.x; my_rt = __int_as_float(rq.z); my_s = __int_as_float(rq.w);
            if constexpr (CLAIM) {
                my_iux = fq.z << 16;          // (rank at the user, rank at the item, the user's members, the item's)
            } else {
                my_iux = fq.x; my_seg = fq.y;
            }
        }
    };
    float term = 0.f, g0 = 0.f, g1 = 0.f;
    if (fair) {
        // statistics of the item over its members in the order of focf_fair_kernel: 16 lanes, lane l takes the members
        // l, l + 16, l + 32, ... one after the other, then a butterfly
        float sp0 = 0.f, sp1 = 0.f, st0 = 0.f, st1 = 0.f, n0 = 0.f, n1 = 0.f;
        bool bad = false;
        int first_b = 0;
        for (int jb = 0; jb < ni; jb += 64) {
            load_chunk(jb);
            if (jb == 0) first_b = __builtin_amdgcn_readlane(my_b, 0);
#pragma unroll
            for (int k = 0; k < 64 / FAIR_GROUP; ++k) {
                const int src = (lane & (FAIR_GROUP - 1)) + FAIR_GROUP * k;
                const float pr = __shfl(my_pr, src, 64), rr = __shfl(my_rt, src, 64), sq = __shfl(my_s, src, 64);
                if (lane < FAIR_GROUP && jb + src < ni) {
                    bad |= (sq != smin && sq != smax);
                    if (sq == smin) {
                        sp0 += pr; st0 += rr; n0 += 1.f;
                    } else {
                        sp1 += pr; st1 += rr; n1 += 1.f;
                    }
                }
            }
        }
        if (bad && KA(err)) atomicOr(KA(err), FR_DEV_ERR_SST_GROUPS);
        sp0 = group_sum<FAIR_GROUP>(sp0); sp1 = group_sum<FAIR_GROUP>(sp1);
        st0 = group_sum<FAIR_GROUP>(st0); st1 = group_sum<FAIR_GROUP>(st1);
        n0 = group_sum<FAIR_GROUP>(n0);   n1 = group_sum<FAIR_GROUP>(n1);
        sp0 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, sp0)));
        sp1 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, sp1)));
        st0 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, st0)));
        st1 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, st1)));
        n0 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, n0)));
        n1 = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, n1)));
        focf_fair_eval(KA(objective), KA(fair_weight), K, sp0, sp1, st0, st1, n0, n1, term, g0, g1);
        // (CLAIM: at the place of the item's first member in batch order -- every member has zeroed its own place before
        // it arrived -- so that the loss sums the terms in an order that does not depend on who arrived last)
        if (lane == 0) KA(term)[CLAIM ? first_b : seg_i] = term;
    }
    // dLoss/dpred of every member (one per lane), the item row's gradient summed over the members in ascending batch
    // position -- coef * (user row), the product rounded, then added: embedding_dense_backward's order -- and the Adam step
    // of every member whose user is its alone (gradient coef * item row BEFORE its update: `pi` is updated last)
    constexpr int UN = E == 1 ? 4 : (E == 2 ? 2 : 1);
    RowFrag<E> gi;
#pragma unroll
    for (int e = 0; e < E; ++e) gi.x[e] = 0.f;
    bool shared_users = false;
    for (int jb = 0; jb < ni; jb += 64) {
        if (big || !fair) load_chunk(jb);
        const int cnt = min(64, ni - jb);
        float my_c = 0.f;
        if (lane < cnt) {
            const float erq = my_pr - my_rt;
            my_c = 2.f * erq / (float)KA(B);
            if (fair) my_c = my_c + (my_s == smin ? g0 : g1);
            if ((my_iux >> 16) > 1) st_sc1(w.coef + my_b, my_c);      // read by the user level of that member's user
        }
        shared_users |= __ballot(lane < cnt && (my_iux >> 16) > 1) != 0ull;
        for (int t0 = 0; t0 < cnt; t0 += UN) {
            RowFrag<E> p[UN], m[UN], v[UN];
            int uq[UN], nq[UN];
            float cq[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                const int t = t0 + q < cnt ? t0 + q : cnt - 1;     // tail: the last member again, not used
                const int bq = __builtin_amdgcn_readlane(my_b, t);
                uq[q] = __builtin_amdgcn_readlane(my_u, t);
                nq[q] = __builtin_amdgcn_readlane(my_iux, t) >> 16;
                cq[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_c), t));
                const size_t sq = (size_t)bq * D;
                load_row_sc1<E, FULL>(p[q], w.side[0] + sq, D, lane);
                load_row_sc1<E, FULL>(m[q], w.side[1] + sq, D, lane);
                load_row_sc1<E, FULL>(v[q], w.side[2] + sq, D, lane);
            }
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                if (t0 + q >= cnt) continue;
                {
#pragma clang fp contract(off)
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        float prod = cq[q] * p[q].x[e];
                        gi.x[e] = gi.x[e] + prod;
                    }
                }
                if (nq[q] == 1) {
                    RowFrag<E> gu;
#pragma unroll
                    for (int e = 0; e < E; ++e) gu.x[e] = cq[q] * pi.x[e];
                    adam_write<E, FULL>(KA(Up), KA(Um), KA(Uv), KA(Ulast), D, KA(step), c, uq[q], p[q], m[q], v[q], gu, s, lane);
                }
            }
        }
    }
    adam_write<E, FULL>(KA(Ip), KA(Im), KA(Iv), KA(Ilast), D, KA(step), c, ir, pi, mi, vi, gi, s, lane);
    if (!shared_users) return;
    // ---- members whose user has other interactions in the batch: arrive at the user's counter (this member's dLoss/dpred
    // and item row are in memory once the stores above have drained); whoever arrives last finishes that user
    drain_stores();
    for (int jb = 0; jb < ni; jb += 64) {
        if (big) load_chunk(jb);
        const int cnt = min(64, ni - jb);
        unsigned long long todo = __ballot(lane < cnt && (my_iux >> 16) > 1);
        while (todo) {
            const int t = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int iq = __builtin_amdgcn_readlane(my_iux, t);
            if constexpr (CLAIM) {
                const int uu = __builtin_amdgcn_readlane(my_u, t);
                const unsigned long long old = hc_arrive(KA(hcu) + uu, lane);
                if (hc_cnt(old) == 1) user_finish_c<E, FULL>(kv, c, w, uu, hc_base(old), iq >> 16, s, lane);
            } else {
                const int sg = __builtin_amdgcn_readlane(my_seg, t);
                if (arrive_last(w.cnt_u + sg, iq >> 16, lane)) user_finish<E, FULL>(kv, c, w, iq & 0xffff, iq >> 16, s, lane);
            }
        }
    }
}

// The fairness term of an item with ONE member in the batch, from that member's own values: focf_fair_eval with the other
// group's sums at zero (its P = T = 0 / 1e-5 = 0, and every objective maps 0 to d = 0), same operations on the member's
// group, so both forms give the same bits.
__device__ __forceinline__ void focf_fair_single(int objective, float fair_weight, float K, bool in0, float pred,
                                                 float rating, float& term, float& g) {
    const float cc = 1.f + 1e-5f;
    const float P = pred / cc, T = rating / cc;
    float d, q;
    if (objective == FR_FOCF_VALUE) {
        d = P - T; q = 1.f;
    } else if (objective == FR_FOCF_ABSOLUTE) {
        d = fabsf(P - T); q = (P > T) ? 1.f : (P < T ? -1.f : 0.f);
    } else if (objective == FR_FOCF_UNDER) {
        d = (T - P > 0.f) ? T - P : 0.f; q = (T - P > 0.f) ? -1.f : 0.f;
    } else {
        d = (P - T > 0.f) ? P - T : 0.f; q = (P - T > 0.f) ? 1.f : 0.f;
    }
    const float delta = in0 ? d - 0.f : 0.f - d;
    const float x = fabsf(delta);
    term = smooth_l1(x);
    const float sgn = delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f);
    const float dx = (x < 1.f ? x : 1.f) * sgn * fair_weight / K;
    g = in0 ? dx * q / cc : -dx * q / cc;
}

// One interaction with both rows caught up (x.A = its user row, x.B = its item row): score, squared error, dLoss/dpred,
// and -- when nobody else in the batch touches either row -- both gradients and both Adam steps, the two rows as packed
// pairs.  iux / iix = (j0 | n << 16) of its user / item segment, segs = user segment | item segment << 16, b = position.
template <int E, bool FULL, bool CLAIM>
__device__ __forceinline__ void step_finish(KV kv, const AdamC& c, int lane, int q, RowFrag<E>& pu, RowFrag<E>& mu,
                                            RowFrag<E>& vu, RowFrag<E>& pi, RowFrag<E>& mi, RowFrag<E>& vi) {
    const int D = FULL ? 64 * E : KA(D);
    const bool fair = KA(objective) != FR_FOCF_NONE;
    // the interaction's records again (scalar loads, cache hits): nothing of them was kept across the replay
    const int4 vrec = KA(task_rec)[q], vinf = KA(task_info)[q], vhd = *reinterpret_cast<const int4*>(KA(hdr));
    const int4 rec = make_int4(uniform(vrec.x), uniform(vrec.y), uniform(vrec.z), uniform(vrec.w));
    const int4 inf = make_int4(uniform(vinf.x), uniform(vinf.y), uniform(vinf.z), uniform(vinf.w));
    const int4 hd = make_int4(uniform(vhd.x), 0, uniform(vhd.z), uniform(vhd.w));
    const int ur = rec.x, ir = rec.y;
    const float rt = __int_as_float(rec.z), s = __int_as_float(rec.w);
    const int iux = inf.x, iix = inf.y, b = inf.w;
    const float K = (float)hd.x, smin = __int_as_float(hd.z), smax = __int_as_float(hd.w);
    const int nu = iux >> 16, ni = iix >> 16, seg_u = inf.z & 0xffff, seg_i = inf.z >> 16;
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) dot = fmaf(pu.x[e], pi.x[e], dot);
    dot = wave_sum(dot);
    const float er = dot - rt;
    if (lane == 0) store_word(KA(mse_e) + b, er * er);
    const float cm = 2.f * er / (float)KA(B);        // d mean((pred - r)^2) / d pred
    // dLoss/dpred of an interaction whose item has no other member in the batch: its per-item statistics are its own
    float coef = cm;
    if (ni == 1 && fair) {
        const bool in0 = s == smin;
        if (s != smin && s != smax && lane == 0 && KA(err)) atomicOr(KA(err), FR_DEV_ERR_SST_GROUPS);
        float term, g;
        focf_fair_single(KA(objective), KA(fair_weight), K, in0, dot, rt, term, g);
        coef = cm + g;
        if (lane == 0) KA(term)[CLAIM ? b : seg_i] = term;
    }
    if (ni == 1 && nu == 1) {      // ---- nobody else touches either row: finish here
        const float2 sc = step_scalars(c, KA(step));
#pragma unroll
        for (int e = 0; e < E; ++e) {       // adam_elem on the pair (user element, item element)
            v2f_ P = {pu.x[e], pi.x[e]}, M = {mu.x[e], mi.x[e]}, V = {vu.x[e], vi.x[e]};
            const v2f_ G = __builtin_elementwise_fma(v2f_{c.wd, c.wd}, P, v2f_{coef * P.y, coef * P.x});
            M = __builtin_elementwise_fma(v2f_{c.omb1, c.omb1}, G - M, M);
            V = __builtin_elementwise_fma(G * c.omb2, G, V * c.b2);
            const v2f_ den = __builtin_elementwise_fma(v2f_{__builtin_amdgcn_sqrtf(V.x), __builtin_amdgcn_sqrtf(V.y)},
                                                      v2f_{sc.y, sc.y}, v2f_{c.eps, c.eps});
            P = __builtin_elementwise_fma(M * -sc.x, v2f_{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)}, P);
            pu.x[e] = P.x; pi.x[e] = P.y;
            mu.x[e] = M.x; mi.x[e] = M.y;
            vu.x[e] = V.x; vi.x[e] = V.y;
        }
        store_trow<E, FULL>(pu, KA(Up) + (size_t)ur * D, D, lane);
        store_trow<E, FULL>(mu, KA(Um) + (size_t)ur * D, D, lane);
        store_trow<E, FULL>(vu, KA(Uv) + (size_t)ur * D, D, lane);
        store_trow<E, FULL>(pi, KA(Ip) + (size_t)ir * D, D, lane);
        store_trow<E, FULL>(mi, KA(Im) + (size_t)ir * D, D, lane);
        store_trow<E, FULL>(vi, KA(Iv) + (size_t)ir * D, D, lane);
        if (lane == 0) {
            store_word(KA(Ulast) + ur, KA(step));
            store_word(KA(Ilast) + ir, KA(step));
        }
        return;
    }
    step_shared_rows<E, FULL, CLAIM>(kv, c, b, lane, ur, ir, iux, iix, seg_u, seg_i, dot, coef, smin, smax, K, pu, mu, vu, pi, mi,
                                     vi);
}

// One Adam step with data gradient coef * (the other row) on an interaction's two rows -- nobody else in the batch touches
// either -- as packed pairs (user element, item element), and the write-back
template <int E, bool FULL>
__device__ __forceinline__ void adam_store_both(KV kv, const AdamC& c, int lane, int ur, int ir, float coef, float2 sc,
                                                RowFrag<E> pu, RowFrag<E> mu, RowFrag<E> vu, RowFrag<E> pi,
                                                RowFrag<E> mi, RowFrag<E> vi) {      // rows by value: the caller's stay as they are
    const int D = FULL ? 64 * E : KA(D);
#pragma unroll
    for (int e = 0; e < E; ++e) {       // adam_elem on the pair (user element, item element)
        v2f_ P = {pu.x[e], pi.x[e]}, M = {mu.x[e], mi.x[e]}, V = {vu.x[e], vi.x[e]};
        const v2f_ G = __builtin_elementwise_fma(v2f_{c.wd, c.wd}, P, v2f_{coef * P.y, coef * P.x});
        M = __builtin_elementwise_fma(v2f_{c.omb1, c.omb1}, G - M, M);
        V = __builtin_elementwise_fma(G * c.omb2, G, V * c.b2);
        const v2f_ den = __builtin_elementwise_fma(v2f_{__builtin_amdgcn_sqrtf(V.x), __builtin_amdgcn_sqrtf(V.y)},
                                                  v2f_{sc.y, sc.y}, v2f_{c.eps, c.eps});
        P = __builtin_elementwise_fma(M * -sc.x, v2f_{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)}, P);
        pu.x[e] = P.x; pi.x[e] = P.y;
        mu.x[e] = M.x; mi.x[e] = M.y;
        vu.x[e] = V.x; vi.x[e] = V.y;
    }
    store_trow<E, FULL>(pu, KA(Up) + (size_t)ur * D, D, lane);
    store_trow<E, FULL>(mu, KA(Um) + (size_t)ur * D, D, lane);
    store_trow<E, FULL>(vu, KA(Uv) + (size_t)ur * D, D, lane);
    store_trow<E, FULL>(pi, KA(Ip) + (size_t)ir * D, D, lane);
    store_trow<E, FULL>(mi, KA(Im) + (size_t)ir * D, D, lane);
    store_trow<E, FULL>(vi, KA(Iv) + (size_t)ir * D, D, lane);
    if (lane == 0) {
        store_word(KA(Ulast) + ur, KA(step));
        store_word(KA(Ilast) + ir, KA(step));
    }
}

// The two interactions of a pair wave with their rows caught up (us.A / is.A = user and item row of interaction q, us.B /
// is.B of interaction q + 1), finished TOGETHER: their records in one batch of loads, the two scores through one
// interleaved butterfly, and everything that is one value per interaction (error, dLoss/dpred, the fairness term of a
// one-member item with its IEEE divisions) computed once, lane parity choosing the interaction -- each as step_finish
// does it for one, same operations, same bits.
template <int E, bool FULL, bool CLAIM>
__device__ __forceinline__ void step_finish_pair(KV kv, const AdamC& c, int lane, int q, int q1, bool has1, TwoRows<E> us,
                                                 TwoRows<E> is) {      // q, q1: the two interactions' record slots
    const int obj = KA(objective);
    const bool fair = obj != FR_FOCF_NONE;
    const GInt4Ptr trec = gp(KA(task_rec));
    const GInt4Ptr tinf = gp(KA(task_info));
    const int4 vr0 = trec[q], vr1 = trec[q1], vi0 = tinf[q], vi1 = tinf[q1];
    const int4 vhd = gp(reinterpret_cast<const int4*>(KA(hdr)))[0];
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        d0 = fmaf(us.pA.x[e], is.pA.x[e], d0);
        d1 = fmaf(us.pB.x[e], is.pB.x[e], d1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {      // wave_sum of both
        const float y0 = __shfl_xor(d0, o, 64), y1 = __shfl_xor(d1, o, 64);
        d0 += y0;
        d1 += y1;
    }
    // ---- per interaction, in the lanes of its parity
    const bool odd = (lane & 1) != 0;
    // (component by component: a select between two int4 structs is lowered through memory)
    const int rec_z = odd ? vr1.z : vr0.z, rec_w = odd ? vr1.w : vr0.w;
    const int inf_y = odd ? vi1.y : vi0.y, inf_z = odd ? vi1.z : vi0.z, inf_w = odd ? vi1.w : vi0.w;
    const float dot = odd ? d1 : d0;
    const float rt = __int_as_float(rec_z), sst = __int_as_float(rec_w);
    const float K = (float)vhd.x, smin = __int_as_float(vhd.z), smax = __int_as_float(vhd.w);
    const int ni_l = inf_y >> 16;
    const float er = dot - rt;
    const float cm = 2.f * er / (float)KA(B);        // d mean((pred - r)^2) / d pred
    float coef = cm, term = 0.f;
    const bool single = fair && ni_l == 1;            // the item's per-item statistics are this interaction's own
    if (fair) {
        float g;
        focf_fair_single(obj, KA(fair_weight), K, sst == smin, dot, rt, term, g);
        if (single) coef = cm + g;
    }
    if (lane < (has1 ? 2 : 1)) {
        store_word(KA(mse_e) + inf_w, er * er);
        if (single) {
            store_word(KA(term) + (CLAIM ? inf_w : inf_z >> 16), term);
            if (sst != smin && sst != smax && KA(err)) atomicOr(KA(err), FR_DEV_ERR_SST_GROUPS);
        }
    }
    // ---- back to wave-uniform values
    const float coef0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, coef), 0));
    const float coef1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, coef), 1));
    const int ur0 = uniform(vr0.x), ir0 = uniform(vr0.y), ur1 = uniform(vr1.x), ir1 = uniform(vr1.y);
    const int iux0 = uniform(vi0.x), iix0 = uniform(vi0.y), iux1 = uniform(vi1.x), iix1 = uniform(vi1.y);
    const bool un0 = (iux0 >> 16) == 1 && (iix0 >> 16) == 1;
    const bool un1 = has1 && (iux1 >> 16) == 1 && (iix1 >> 16) == 1;
    const float2 sc = step_scalars(c, KA(step));
    // rows nobody else touches are finished first (two independent chains when both interactions are of that kind) ...
    if (un0 && un1) {
        // (both at once, from plain copies of the twelve fragments: two independent chains for the scheduler)
        RowFrag<E> a0 = us.pA, a1 = us.mA, a2 = us.vA, a3 = is.pA, a4 = is.mA, a5 = is.vA;
        RowFrag<E> b0 = us.pB, b1 = us.mB, b2 = us.vB, b3 = is.pB, b4 = is.mB, b5 = is.vB;
        adam_store_both<E, FULL>(kv, c, lane, ur0, ir0, coef0, sc, a0, a1, a2, a3, a4, a5);
        adam_store_both<E, FULL>(kv, c, lane, ur1, ir1, coef1, sc, b0, b1, b2, b3, b4, b5);
        return;
    }
    if (un0) adam_store_both<E, FULL>(kv, c, lane, ur0, ir0, coef0, sc, us.pA, us.mA, us.vA, is.pA, is.mA, is.vA);
    if (un1) adam_store_both<E, FULL>(kv, c, lane, ur1, ir1, coef1, sc, us.pB, us.mB, us.vB, is.pB, is.mB, is.vB);
    // ... then the hand-offs of shared rows, each a chain of dependent round trips (ONE instance of that path: the second
    // interaction's rows take the place of the first's)
    if (un0 && (un1 || !has1)) return;
    for (int k = 0; k < (has1 ? 2 : 1); ++k) {
        if (k) {                                      // second turn: the B rows take the A rows' place
#pragma unroll
            for (int e = 0; e < E; ++e) {
                us.pA.x[e] = us.pB.x[e]; us.mA.x[e] = us.mB.x[e]; us.vA.x[e] = us.vB.x[e];
                is.pA.x[e] = is.pB.x[e]; is.mA.x[e] = is.mB.x[e]; is.vA.x[e] = is.vB.x[e];
            }
        }
        if (!(k ? un1 : un0)) {
            const int sg = uniform(k ? vi1.z : vi0.z);
            step_shared_rows<E, FULL, CLAIM>(kv, c, uniform(k ? vi1.w : vi0.w), lane, k ? ur1 : ur0, k ? ir1 : ir0, k ? iux1 : iux0,
                                      k ? iix1 : iix0, sg & 0xffff, sg >> 16, k ? d1 : d0, k ? coef1 : coef0, smin, smax, K,
                                      us.pA, us.mA, us.vA, is.pA, is.mA, is.vA);
        }
    }
}

#if FR_STEP_TRACE
#define g_phase phase_stamps     // per-wave (registers): set by the interaction path, stored by the kernel epilogue
#endif
// ---- the tasks ----------------------------------------------------------------------------------------------------
// A wave's task is "two rows brought up to date, then something done with them":
//   sweeper task q  : rows 2q, 2q + 1 of the step's slice (users first, then items), written back at step `step`;
//   interaction b   : its user row and item row as of step - 1, then score, dLoss/dpred, both gradients, both updates.
template <int E, bool FULL, bool PAIR, bool CLAIM>
__device__ __forceinline__ void step_task(KV kv, bool sweeper, int q, int n_pairs, int lane, const int4* rec0,
                                          const int32_t* order0, int B0
#if FR_STEP_TRACE
                                          , unsigned long long (&phase_stamps)[4]
#endif
) {
    // `lane` made opaque per task: otherwise every per-lane address (10 table pointers + lane) is hoisted out of the
    // wave's task loop and parked in VGPR pairs for the whole kernel (101 VGPRs = 4 waves per SIMD)
    asm volatile("" : "+v"(lane));
    AdamC c;
    c.sc = KAC(sc); c.cap = KAC(cap); c.wd = KAC(wd); c.b1 = KAC(b1); c.omb1 = KAC(omb1); c.b2 = KAC(b2); c.omb2 = KAC(omb2);
    c.eps = KAC(eps); c.k1 = KAC(k1); c.k2 = KAC(k2); c.inv_k1 = KAC(inv_k1); c.inv_k2 = KAC(inv_k2);
    const int D = FULL ? 64 * E : KA(D);
    TwoRows<E> r;
    int tA, tB;
    if (sweeper) {
        // start order of the slice's pairs (fr_focf_prepare_step: longest estimated replay first, so that the four waves
        // of a workgroup -- one per SIMD of its CU -- carry alike loads and the launch ends on its shortest tasks); an
        // order built for another slice size (a batch applied at another step than it was prepared for) is not used
        if (const int32_t* so = order0) {
            const int oq = so[q], on = so[n_pairs];
            q = uniform(on) == n_pairs ? uniform(oq) : q;
        }
        const int pairs_u = (KA(n_u) + 1) >> 1;
        const bool inU = q < pairs_u;
        const int k = inU ? q : q - pairs_u;
        const long long rowA = (inU ? KA(lo_u) : KA(lo_i)) + 2 * k;
        const bool hasB = 2 * k + 1 < (inU ? KA(n_u) : KA(n_i));
        const long long rowB = hasB ? rowA + 1 : rowA;
        float* Tp = inU ? KA(Up) : KA(Ip);
        float* Tm = inU ? KA(Um) : KA(Im);
        float* Tv = inU ? KA(Uv) : KA(Iv);
        int32_t* Tl = inU ? KA(Ulast) : KA(Ilast);
        const int32_t* Ts = inU ? KA(Ustamp) : KA(Istamp);
        // stamps, `last` and the rows in ONE round trip (a row is wasted for the few that are skipped)
        const int sa = gp(Ts)[rowA], sb = gp(Ts)[rowB];
        const int la = gp((const int32_t*)Tl)[rowA], lb = gp((const int32_t*)Tl)[rowB];
        ldrow<E, FULL>(r.pA, Tp + (size_t)rowA * D, D, lane);
        ldrow<E, FULL>(r.mA, Tm + (size_t)rowA * D, D, lane);
        ldrow<E, FULL>(r.vA, Tv + (size_t)rowA * D, D, lane);
        ldrow<E, FULL>(r.pB, Tp + (size_t)rowB * D, D, lane);
        ldrow<E, FULL>(r.mB, Tm + (size_t)rowB * D, D, lane);
        ldrow<E, FULL>(r.vB, Tv + (size_t)rowB * D, D, lane);
        const int upto = KA(step);
        tA = uniform(sa) >= KA(skip_from) ? upto : uniform(la);
        tB = (!hasB || uniform(sb) >= KA(skip_from)) ? upto : uniform(lb);
        const bool doA = tA < upto, doB = tB < upto;
        replay_two<E>(r, tA, tB, upto, c, lane);
        if (doA) {
            store_trow<E, FULL>(r.pA, Tp + (size_t)rowA * D, D, lane);
            store_trow<E, FULL>(r.mA, Tm + (size_t)rowA * D, D, lane);
            store_trow<E, FULL>(r.vA, Tv + (size_t)rowA * D, D, lane);
            if (lane == 0) store_word(Tl + rowA, upto);
        }
        if (doB) {
            store_trow<E, FULL>(r.pB, Tp + (size_t)rowB * D, D, lane);
            store_trow<E, FULL>(r.mB, Tm + (size_t)rowB * D, D, lane);
            store_trow<E, FULL>(r.vB, Tv + (size_t)rowB * D, D, lane);
            if (lane == 0) store_word(Tl + rowB, upto);
        }
        return;
    }
    if (PAIR) {
        // ---- two interactions per wave, neighbours in the start order (= similar replay lengths): their two user rows go
        // through the replay as one packed pair and so do their two item rows (4.5 instead of 7 VALU instructions per row
        // and step); each interaction is then finished on its own
        const bool has1 = q + 1 < B0;
        const int q1 = has1 ? q + 1 : q;
        const int4 vrec0 = gp(rec0)[q], vrec1 = gp(rec0)[q1];
        const int u0 = uniform(vrec0.x), i0 = uniform(vrec0.y), u1 = uniform(vrec1.x), i1 = uniform(vrec1.y);
#if FR_STEP_TRACE
        g_phase[2] = __builtin_amdgcn_s_memrealtime();     // level-1 records have arrived
#endif
        const int lu0 = gp((const int32_t*)KA(Ulast))[u0], lu1 = gp((const int32_t*)KA(Ulast))[u1];
        const int li0 = gp((const int32_t*)KA(Ilast))[i0], li1 = gp((const int32_t*)KA(Ilast))[i1];
        TwoRows<E> it;       // r = the two user rows, it = the two item rows
        ldrow<E, FULL>(r.pA, KA(Up) + (size_t)u0 * D, D, lane);
        ldrow<E, FULL>(r.pB, KA(Up) + (size_t)u1 * D, D, lane);
        ldrow<E, FULL>(it.pA, KA(Ip) + (size_t)i0 * D, D, lane);
        ldrow<E, FULL>(it.pB, KA(Ip) + (size_t)i1 * D, D, lane);
        ldrow<E, FULL>(r.mA, KA(Um) + (size_t)u0 * D, D, lane);
        ldrow<E, FULL>(r.vA, KA(Uv) + (size_t)u0 * D, D, lane);
        ldrow<E, FULL>(r.mB, KA(Um) + (size_t)u1 * D, D, lane);
        ldrow<E, FULL>(r.vB, KA(Uv) + (size_t)u1 * D, D, lane);
        ldrow<E, FULL>(it.mA, KA(Im) + (size_t)i0 * D, D, lane);
        ldrow<E, FULL>(it.vA, KA(Iv) + (size_t)i0 * D, D, lane);
        ldrow<E, FULL>(it.mB, KA(Im) + (size_t)i1 * D, D, lane);
        ldrow<E, FULL>(it.vB, KA(Iv) + (size_t)i1 * D, D, lane);
        const int upto = KA(step) - 1;
        const int tu0 = uniform(lu0), tu1 = uniform(lu1), ti0 = uniform(li0), ti1 = uniform(li1);
#if FR_STEP_TRACE
        g_phase[0] = __builtin_amdgcn_s_memrealtime();     // rows have arrived
        g_phase[3] = (unsigned long long)((upto - tu0) + (upto - tu1) + (upto - ti0) + (upto - ti1));
#endif
        replay_two<E>(r, tu0, tu1, upto, c, lane);
        replay_two<E>(it, ti0, ti1, upto, c, lane);
#if FR_STEP_TRACE
        g_phase[1] = __builtin_amdgcn_s_memrealtime();     // replay done
#endif
        step_finish_pair<E, FULL, CLAIM>(kv, c, lane, q, q1, has1, r, it);
        return;
    }
    // ---- one interaction per wave (wide rows: two interactions' rows would not fit the register budget)
    const int4 vrec = rec0[q];
    const int ur = uniform(vrec.x), ir = uniform(vrec.y);
    const int lu = KA(Ulast)[ur], li = KA(Ilast)[ir];
    ldrow<E, FULL>(r.pA, KA(Up) + (size_t)ur * D, D, lane);
    ldrow<E, FULL>(r.pB, KA(Ip) + (size_t)ir * D, D, lane);
    ldrow<E, FULL>(r.mA, KA(Um) + (size_t)ur * D, D, lane);
    ldrow<E, FULL>(r.vA, KA(Uv) + (size_t)ur * D, D, lane);
    ldrow<E, FULL>(r.mB, KA(Im) + (size_t)ir * D, D, lane);
    ldrow<E, FULL>(r.vB, KA(Iv) + (size_t)ir * D, D, lane);
    tA = uniform(lu);
    tB = uniform(li);
    replay_two<E>(r, tA, tB, KA(step) - 1, c, lane);
    step_finish<E, FULL, CLAIM>(kv, c, lane, q, r.pA, r.mA, r.vA, r.pB, r.mB, r.vB);
}

// two interactions per wave while their twelve row fragments fit the register budget without scratch (D <= 64)
__host__ __device__ constexpr bool step_pairs(int E) { return E <= 1; }

// Block 0 reduces an earlier step's loss (if any); the sweeper blocks and the interaction blocks are dealt evenly through
// the rest of the grid (one task per wave), so that from the first moment the resident waves are a mix of sweeper waves
// (one round trip, then up to S replayed steps of pure VALU work) and interaction waves (two dependent round trips first).
// Measured (FR_STEP_TRACE wave timeline, cfg 2): the launch is VALU-throughput-bound (46 k VALU cycles per SIMD of the
// 73 k the kernel lasts), every SIMD time-shared by its 6-8 resident waves; residency 3..8 waves per SIMD moves the
// kernel by < 2 % (fewer waves = shorter waves but less latency hiding), a single residency of persistent waves with
// static task lists was slower (no dynamic balancing, one memory round trip per task exposed).
#ifndef FR_STEP_TRACE
#define FR_STEP_TRACE 0
#endif

#if FR_STEP_TRACE   // diagnostic build: (start, end) in 10 ns ticks, role and placement of every wave of one launch
__device__ unsigned long long g_step_trace[8 * 65536];
__device__ int g_trace_step = -1;      // >= 0: only the launch that applies this optimizer step is recorded
#endif

#ifndef FR_STEP_WAVES
#define FR_STEP_WAVES 6      // waves per SIMD the register budget is cut for (end of round 3, us per step: 5 -> 29.8, 6 -> 29.5,
#endif                       //   7 -> 30.1 (72 VGPRs, 12 bytes of scratch), 8 -> 32.5 (64 VGPRs, 60 bytes of scratch))
// The leading scalar arguments are the ones a wave needs before it can issue its first load -- its role and task from the
// grid shape, the task list, the sweeper order -- and are PRELOADED into SGPRs by the dispatcher (the file is compiled with
// -amdgpu-kernarg-preload-count, Makefile): the first records are requested in the wave's first cycles, beside the vector
// load of the argument block instead of behind it (one dependent round trip less in every wave's prologue and in the
// launch's ramp).  They repeat fields of StepArgs / StageArgs, which the rest of the kernel reads as before.  Measured:
// little -- 29.77 against 29.89 us per step in graph replay (five interleaved runs each), 30.8 against 31.4 us per launch in
// the event-timed eager pass: a wave's prologue latency is hidden by the other waves of its SIMD.
// (D = 256: the six row fragments of a task are 24 registers, their packed copies as many: at the 85 registers six waves per
// SIMD leave, 28-44 bytes per lane went to scratch memory; five waves -- 102 registers -- hold everything)
__host__ __device__ constexpr int step_waves(int E) { return E >= 4 ? (FR_STEP_WAVES < 5 ? FR_STEP_WAVES : 5) : FR_STEP_WAVES; }
constexpr int STEP_PRELOAD_DWORDS = 10;      // pre_rec, pre_order (2 each), pre_B, pre_lead, pre_nu, pre_ni, pre_stage, pad
template <int E, bool FULL, bool CLAIM>
__global__ __launch_bounds__(64 * STEP_WPB, step_waves(E)) void focf_step_kernel(const int4* pre_rec, const int32_t* pre_order,
                                                                                 int pre_B, int pre_lead, int pre_nu, int pre_ni,
                                                                                 int pre_stage, int pre_pad, StepArgs a,
                                                                                 StageArgs st) {
#if FR_STEP_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long mt0 = __builtin_amdgcn_s_memtime();      // shader clock: the clock the chip holds under this load
    unsigned long long ph[4] = {0, 0, 0, 0};
#endif
    const int lane = threadIdx.x & 63;
    const int wib = uniform((int)(threadIdx.x >> 6));      // wave-uniform, and the compiler has to know it
    int role = 0;
    const int n_stage = CLAIM ? pre_stage : 0;
    (void)pre_pad;
    // Where the stage workgroups sit in the grid: behind the `lead` interaction workgroups (2, default), first (0) or
    // last (1).  Measured at the BASELINE sizes, hipGraph replay: 29.2-29.6 / 29.9-30.2 / 31.3-31.7 us per step (last: the
    // stages' dependent atomics then end the launch); further back among the sweeper workgroups: as (2).  Stage waves at a
    // raised priority (s_setprio 3) end sooner (p90 12.8 instead of 17 us) and cost the step 1.8 us: dropped.
#ifndef FR_STAGE_POS
#define FR_STAGE_POS 2
#endif
    const int stage0 = FR_STAGE_POS == 0 ? 1 : (FR_STAGE_POS == 1 ? (int)gridDim.x - n_stage : 1 + pre_lead);
    if (blockIdx.x == 0) {
        if (a.prev.loss_out) step_reduce_loss<64 * STEP_WPB>(a.prev);
    } else if (CLAIM && (int)blockIdx.x >= stage0 && (int)blockIdx.x < stage0 + n_stage) {
        // the index work of the two coming batches ("In-launch prepare"): a few us of dependent atomics each
        if constexpr (CLAIM) stage_block(st, (int)blockIdx.x - stage0);
        role = 3;
    } else {
        const unsigned* kp = reinterpret_cast<const unsigned*>(
            (const void*)(const __attribute__((address_space(4))) void*)__builtin_amdgcn_kernarg_segment_ptr());
        KV kv;
        static_assert(STEP_PRELOAD_DWORDS * 4 % alignof(StepArgs) == 0, "StepArgs follows the preloaded scalars without padding");
        kv.v0 = kp[STEP_PRELOAD_DWORDS + lane];
        kv.v1 = lane < (int)(sizeof(StepArgs) / 4) - 64 ? kp[STEP_PRELOAD_DWORDS + 64 + lane] : 0u;
        const int x = (int)blockIdx.x - 1 - ((int)blockIdx.x >= stage0 ? n_stage : 0);
        const int n_pairs = ((pre_nu + 1) >> 1) + ((pre_ni + 1) >> 1);
        const int n_sweep = n_pairs;      // one sweeper task (wave) per pair of rows
        const int ns = (n_sweep + STEP_WPB - 1) / STEP_WPB;
        // longest jobs first: the `lead` workgroups of the interactions with the longest replays (the task list is in
        // that order), then the sweeper workgroups (a full period of replay each), then the other interactions
        constexpr bool PAIR = step_pairs(E);
        const bool sweeper = x >= pre_lead && x < pre_lead + ns;
        const int q = ((sweeper ? x - pre_lead : (x < pre_lead ? x : x - ns)) * STEP_WPB + wib) * (!sweeper && PAIR ? 2 : 1);
        role = sweeper ? 1 : 2;
#if FR_STEP_TRACE
        if (q < (sweeper ? n_sweep : pre_B)) step_task<E, FULL, PAIR, CLAIM>(kv, sweeper, q, n_pairs, lane, pre_rec, pre_order, pre_B, ph);
#else
        if (q < (sweeper ? n_sweep : pre_B)) step_task<E, FULL, PAIR, CLAIM>(kv, sweeper, q, n_pairs, lane, pre_rec, pre_order, pre_B);
#endif
    }
#if FR_STEP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long mt1 = __builtin_amdgcn_s_memtime();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // (g_trace_step >= 0: the launches that apply steps g_trace_step .. g_trace_step + 3, 16384 wave records each: consecutive
    // launches side by side, scratch/chain_trace.py)
    const int tsel = g_trace_step < 0 ? 0 : a.step - g_trace_step;
    const unsigned wq = blockIdx.x * STEP_WPB + wib + (g_trace_step < 0 ? 0u : 16384u * (unsigned)(tsel & 3));
    if (lane == 0 && wq < 65536 && (g_trace_step < 0 || (tsel >= 0 && tsel < 4 && blockIdx.x * STEP_WPB + wib < 16384))) {
        g_step_trace[4 * wq] = tr0;
        g_step_trace[4 * wq + 1] = tr1;
        g_step_trace[4 * wq + 2] = role;
        g_step_trace[4 * wq + 3] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        g_step_trace[4 * (65536 + wq)] = (ph[2] ? ((ph[2] - tr0) << 40) : 0ull) | (mt1 - mt0);   // level-1 ticks | shader cycles
        g_step_trace[4 * (65536 + wq) + 1] = ph[0];
        g_step_trace[4 * (65536 + wq) + 2] = ph[1];
        g_step_trace[4 * (65536 + wq) + 3] = ph[3];
    }
#endif
}

__global__ __launch_bounds__(256) void focf_step_finish_kernel(PrevLoss pl) { step_reduce_loss<256>(pl); }

// the stages on their own (the first steps of a loop, whose batches no earlier launch could carry)
__global__ __launch_bounds__(STAGE_THREADS) void focf_stage_kernel(StageArgs st) { stage_block(st, (int)blockIdx.x); }

// Start order of the sweeper tasks of ONE step (task = a pair of neighbouring rows of that step's sweep slice): longest
// estimated replay first.  The hardware hands the launch's workgroups to the CUs in index order, one wave of a workgroup
// per SIMD; with the tasks in slice order a SIMD's load is a sum of random replay lengths (0 .. 2 S row-steps per task)
// and the launch ends on its unluckiest SIMD (FR_STEP_TRACE: the SIMDs of one CU finish 3 us apart, the CUs 6 us).  In
// this order the four tasks of a workgroup are alike and the last tasks to start are the shortest.  Estimated from the
// rows' `last` stamps as of the prepare launch, like the interactions' order: only the order depends on it.
struct SwOrderJob {
    const int32_t *Ulast, *Ustamp, *Ilast, *Istamp;
    long long lo_u, lo_i;
    int n_u, n_i, upto, skip_from;
    int32_t* order;
};

// One workgroup of 1024 threads (an extra workgroup of the launch-order launch); cnt: [SW_NC * PT * 16] ints of LDS
template <int PT>   // rounds of 1024 tasks
__device__ __forceinline__ void sweep_order_body(const SwOrderJob& J, int cap_, int* cnt, int* wsum) {
    constexpr int NC = SW_NC, NW = 16, N = NC * PT * NW, EPT = (N + 1023) / 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int cap = cap_ > 0 ? cap_ : 1024;
    const int pairs_u = (J.n_u + 1) >> 1, n_pairs = pairs_u + ((J.n_i + 1) >> 1);
    if (!J.order) return;
    if (n_pairs > PT * 1024) {      // a slice larger than this launch ranks (an unusually short sweep period): index order
        if (tid == 0 && n_pairs <= SWEEP_ORDER_MAX) J.order[n_pairs] = -1;
        return;
    }
    int la[PT], lb[PT], sa[PT], sb[PT];
#pragma unroll
    for (int r = 0; r < PT; ++r) {      // every load of the phase in flight before the first is used
        const int q = r * 1024 + tid, qc = q < n_pairs ? q : 0;
        const bool inU = qc < pairs_u;
        const int k = inU ? qc : qc - pairs_u;
        const long long rowA = (inU ? J.lo_u : J.lo_i) + 2 * k;
        const long long rowB = 2 * k + 1 < (inU ? J.n_u : J.n_i) ? rowA + 1 : rowA;
        const int32_t* Tl = inU ? J.Ulast : J.Ilast;
        const int32_t* Ts = inU ? J.Ustamp : J.Istamp;
        la[r] = Tl[rowA]; lb[r] = Tl[rowB]; sa[r] = Ts[rowA]; sb[r] = Ts[rowB];
    }
    int cls[PT], rank[PT];
#pragma unroll
    for (int r = 0; r < PT; ++r) {
        const int q = r * 1024 + tid;
        int k = -1;
        if (q < n_pairs) {
            int ca = sa[r] >= J.skip_from ? 0 : J.upto - la[r], cb = sb[r] >= J.skip_from ? 0 : J.upto - lb[r];
            ca = ca < 0 ? 0 : (ca > cap ? cap : ca);
            cb = cb < 0 ? 0 : (cb > cap ? cap : cb);
            const int hi = ca > cb ? ca : cb, lo = ca > cb ? cb : ca;
            const int cost = 7 * hi + 2 * lo;                          // VALU instructions: alone 7, as a pair 9 per step
            k = NC - 1 - min(NC - 1, cost * NC / (9 * cap + 1));
        }
        cls[r] = k;
        rank[r] = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned long long m = __ballot(k == c);
            if (k == c) rank[r] = __popcll(m & lt);
            if (lane == 0) cnt[(c * PT + r) * NW + wave] = __popcll(m);
        }
    }
    __syncthreads();
    {   // exclusive scan of the N counts, EPT consecutive ones per thread
        int x[EPT], own = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid * EPT + e;
            x[e] = i < N ? cnt[i] : 0;
            own += x[e];
        }
        int inc = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(inc, o, 64);
            if (lane >= o) inc += y;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int ex = inc - own;
        for (int w = 0; w < wave; ++w) ex += wsum[w];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid * EPT + e;
            if (i < N) cnt[i] = ex;
            ex += x[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PT; ++r)
        if (cls[r] >= 0) J.order[cnt[(cls[r] * PT + r) * NW + wave] + rank[r]] = r * 1024 + tid;
    if (tid == 0) J.order[n_pairs] = n_pairs;
}

// Start order of the interactions of a batch: longest replay first (the launch ends one wave latency after its last wave
// starts, and a wave's latency is its replay length: 3.4 us with nothing to replay, 25 us with 2 x 123 steps).  One
// workgroup per batch, behind the index sort on the look-ahead stream: estimated VALU cost from the rows' `last` stamps as
// of NOW (a row touched again before the batch runs has less to replay than estimated -- only the order is affected,
// never a result), stable partition into 8 cost classes, records rewritten in that order.
struct LptJob {
    const int32_t *age_u, *age_i;
    const int4 *rec, *info;
    int4 *task_rec, *task_info;
    int B, upto;
};
struct LptJobs {
    LptJob j[FR_FOCF_PREPARE_MAX];
    SwOrderJob sw[FR_FOCF_PREPARE_MAX];   // the same steps' sweeper tasks: ranked by one extra workgroup per batch
    int n;        // batches
    int cap;      // replay lengths are bounded by the sweep period
};

#ifdef FR_LPT_STAMPS   // diagnostic build only: phase time stamps of block 0 / thread 0
__device__ unsigned long long g_lpt_stamps[8];
#define LPT_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_lpt_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LPT_STAMP(i) do {} while (0)
#endif

constexpr int LPT_SPLIT = 4;

template <int PT>   // rounds of 1024 positions: B <= PT * 1024
__global__ __launch_bounds__(1024) void focf_lpt_kernel(LptJobs jobs) {
    // Stable partition into NC cost classes (class 0 = the longest replays) with ballots and prefix counts -- no atomics
    // (a counting sort on LDS counters serialises the 64 lanes of every wave on the few hot cost classes).
    // LPT_SPLIT workgroups per batch: each classifies and scans the whole batch (coalesced reads, cheap) and rewrites
    // its share of the rounds (16-byte stores to scattered places, what a single CU is slow at).
    constexpr int NC = 8, NW = 16, N = NC * PT * NW, EPT = (N + 1023) / 1024;
    __shared__ int cnt[SW_NC * PT * NW];     // [class][round][wave], scanned in that order (sized for the sweeper ranking)
    __shared__ int wsum[NW];
    if ((int)blockIdx.x >= jobs.n * LPT_SPLIT) {
        sweep_order_body<PT>(jobs.sw[blockIdx.x - jobs.n * LPT_SPLIT], jobs.cap, cnt, wsum);
        return;
    }
    const LptJob& J = jobs.j[blockIdx.x / LPT_SPLIT];
    const int part = blockIdx.x % LPT_SPLIT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int cap = jobs.cap > 0 ? jobs.cap : 1024;
    int cls[PT], rank[PT];
    // every load of a phase is issued before the first one is waited for (clamped positions instead of branches: with a
    // branch per round the compiler serialises the rounds' memory round trips); the rows' `last` stamps were gathered
    // by the stamp workgroups of the sort launch (16K random misses from this one CU took 16 us)
    int lu[PT], li[PT];
    bool sh[PT];     // the interaction shares its user or its item row with another interaction of the batch
    LPT_STAMP(0);
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int b = q * 1024 + tid, bc = b < J.B ? b : 0;
        lu[q] = J.age_u[bc];
        li[q] = J.age_i[bc];
#if FR_LPT_SHARED_FIRST
        const int4 f = J.info[bc];
        sh[q] = (f.x >> 16) > 1 || (f.z >> 16) > 1;
#else
        sh[q] = false;
#endif
    }
    LPT_STAMP(1);
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int b = q * 1024 + tid;
        int k = -1;
        if (b < J.B) {
            int cu = J.upto - lu[q], ci = J.upto - li[q];
            cu = cu < 0 ? 0 : (cu > cap ? cap : cu);
            ci = ci < 0 ? 0 : (ci > cap ? cap : ci);
            const int hi = cu > ci ? cu : ci, lo = cu > ci ? ci : cu;
            const int cost = 7 * hi + 2 * lo;                          // VALU instructions: alone 7, as a pair 9 per step
            k = NC - 1 - min(NC - 1, cost * NC / (9 * cap + 1));
            // a shared row is finished by the last of its waves to arrive, behind a hand-off through memory and three more
            // dependent load levels: the longest chain of the launch whatever its replay length, so it starts first
            if (sh[q]) k = 0;
        }
        cls[q] = k;
        rank[q] = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned long long m = __ballot(k == c);
            if (k == c) rank[q] = __popcll(m & lt);
            if (lane == 0) cnt[(c * PT + q) * NW + wave] = __popcll(m);
        }
    }
    LPT_STAMP(2);
    __syncthreads();
    {   // exclusive scan of the N counts, EPT consecutive ones per thread
        int x[EPT], own = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid * EPT + e;
            x[e] = i < N ? cnt[i] : 0;
            own += x[e];
        }
        int inc = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(inc, o, 64);
            if (lane >= o) inc += y;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int ex = inc - own;
        for (int w = 0; w < wave; ++w) ex += wsum[w];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid * EPT + e;
            if (i < N) cnt[i] = ex;
            ex += x[e];
        }
    }
    __syncthreads();
    LPT_STAMP(3);
    constexpr int CH = (PT + LPT_SPLIT - 1) / LPT_SPLIT;   // rounds of this workgroup: part, part + LPT_SPLIT, ...
    int4 f[CH], r[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        const int b = (q * LPT_SPLIT + part) * 1024 + tid, bc = b < J.B ? b : 0;
        f[q] = J.info[bc];
        r[q] = J.rec[bc];
    }
#pragma unroll
    for (int q = 0; q < CH; ++q) {
#pragma unroll
        for (int pp = 0; pp < LPT_SPLIT; ++pp) {   // (static register indexing: the round is q * LPT_SPLIT + part)
            const int qq = q * LPT_SPLIT + pp;
            if (pp == part && qq < PT && cls[qq] >= 0) {
                const int pos = cnt[(cls[qq] * PT + qq) * NW + wave] + rank[qq];
                J.task_rec[pos] = r[q];
                J.task_info[pos] = make_int4(f[q].x, f[q].z, f[q].y | (f[q].w << 16), qq * 1024 + tid);
            }
        }
    }
    LPT_STAMP(4);
}

}  // namespace fr

using namespace fr;

extern "C" int fr_focf_prepare_step(const fr_focf_batch* batches, const int32_t* stamps, int32_t n, const fr_table* U,
                                    const fr_table* I, int32_t replay_cap, uint32_t* err_flag, void* stream_) {
    int rc;
    if ((rc = check_table(U, "fr_focf_prepare_step(U)")) || (rc = check_table(I, "fr_focf_prepare_step(I)"))) return rc;
    FR_CHECK_ARG(batches && stamps && n >= 1 && 2 * n <= FR_SORT_JOBS && U->dim == I->dim,
                 "fr_focf_prepare_step: 1..%d batches", FR_SORT_JOBS / 2);
    SortJobList jobs{};
    for (int q = 0; q < n; ++q) {
        const fr_focf_batch& b = batches[q];
        FR_CHECK_ARG(b.user && b.item && b.rating && b.ws && b.B >= 1 && b.B <= FR_SORT_MAX,
                     "fr_focf_prepare_step: batch %d", q);
        FocfWs w = focf_layout(b.ws, b.B, U->dim);
        FR_CHECK_ARG(b.ws_bytes >= w.bytes, "fr_focf_prepare_step: workspace %zu < %zu bytes", b.ws_bytes, w.bytes);
        SortJob ju{b.user, U->n_rows, w.perm_u, w.seg_start_u, w.seg_row_u, nullptr, w.nseg_u, nullptr, nullptr};
        SortJob ji{b.item, I->n_rows, w.perm_i, w.seg_start_i, w.seg_row_i, nullptr, w.nseg_i, b.sst, w.sst_minmax};
        ju.seg_first = w.seg_first_u;
        ji.seg_first = w.seg_first_i;
        ju.info = reinterpret_cast<int2*>(w.info);          // one 16-byte record per position: user half, item half
        ji.info = reinterpret_cast<int2*>(w.info) + 1;
        ju.info_stride = ji.info_stride = 2;
        ji.rec = w.rec;
        ji.rec_idx = b.user;
        ji.rec_rows = U->n_rows;
        ji.rec_f0 = b.rating;
        ju.cnt = w.cnt_u;
        ji.cnt = w.cnt_i;
        ju.stamp = U->stamp;
        ji.stamp = I->stamp;
        ju.stamp_val = ji.stamp_val = stamps[q];
        ju.last = U->last;
        ji.last = I->last;
        ju.last_out = w.age_u;
        ji.last_out = w.age_i;
        ji.pos_of = w.pos_i;
        jobs.j[2 * q] = ju;
        jobs.j[2 * q + 1] = ji;
        jobs.M[2 * q] = jobs.M[2 * q + 1] = (int)b.B;
    }
    jobs.n = 2 * n;
    if ((rc = launch_sort_many(jobs, U->n_rows > I->n_rows ? U->n_rows : I->n_rows, err_flag, (hipStream_t)stream_)))
        return rc;
    LptJobs lj{};
    for (int q = 0; q < n; ++q) {
        const FocfWs w = focf_layout(batches[q].ws, batches[q].B, U->dim);
        lj.j[q] = LptJob{w.age_u, w.age_i, w.rec, w.info, w.task_rec, w.task_info, (int)batches[q].B, stamps[q] - 1};
    }
    lj.cap = replay_cap;
    lj.n = n;
    for (int q = 0; q < n; ++q) {      // start order of each step's sweeper tasks (the slice is a function of the step the batch is
        const FocfWs w = focf_layout(batches[q].ws, batches[q].B, U->dim);      // stamped for); no sweeper, no order
        fr_table tu = *U, ti = *I;
        tu.step = ti.step = stamps[q];
        const SweepSlice sw = make_sweep_slice(&tu, &ti, replay_cap);
        lj.sw[q] = SwOrderJob{U->last, U->stamp, I->last, I->stamp, sw.lo_u, sw.lo_i, sw.n_u, sw.n_i, stamps[q], stamps[q],
                              replay_cap > 0 ? w.sw_order : nullptr};
    }
    int Bmax = 1;
    for (int q = 0; q < n; ++q) Bmax = batches[q].B > Bmax ? (int)batches[q].B : Bmax;
    hipStream_t st = (hipStream_t)stream_;
    ProfScope prof(K_FOCF_LPT, st);
    if (Bmax <= 1024) FR_LAUNCH(prof, focf_lpt_kernel<1>, dim3(n * (LPT_SPLIT + 1)), dim3(1024), 0, st, lj);
    else if (Bmax <= 2048) FR_LAUNCH(prof, focf_lpt_kernel<2>, dim3(n * (LPT_SPLIT + 1)), dim3(1024), 0, st, lj);
    else if (Bmax <= 4096) FR_LAUNCH(prof, focf_lpt_kernel<4>, dim3(n * (LPT_SPLIT + 1)), dim3(1024), 0, st, lj);
    else if (Bmax <= 8192) FR_LAUNCH(prof, focf_lpt_kernel<8>, dim3(n * (LPT_SPLIT + 1)), dim3(1024), 0, st, lj);
    else FR_LAUNCH(prof, focf_lpt_kernel<16>, dim3(n * (LPT_SPLIT + 1)), dim3(1024), 0, st, lj);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

// generation `gen` (0..2) of the row words: [3][n_users + n_items] words
static unsigned long long* words_of(uint64_t* row_words, int32_t gen, const fr_table* U, const fr_table* I, bool item) {
    const size_t per = (size_t)U->n_rows + (size_t)I->n_rows;
    return reinterpret_cast<unsigned long long*>(row_words) + (size_t)gen * per + (item ? (size_t)U->n_rows : 0);
}

// the stage descriptors of a launch: `claim` = the batch two steps ahead (or null), `place` = the next one (or null)
// claim_step / place_step: the optimizer steps the two batches will be APPLIED at -- what the sweep slice their sweeper orders
// are built for hangs on.  A stamp only has to grow from batch to batch: it runs ahead of the steps once a claimed batch was
// never applied (a loop cut short), and an order built for the stamp's slice is then an order for the wrong rows (the step
// falls back to slice order: +3 us per step, which is what bench.py's eager and library-loop figures carried until round 6).
static int make_stages(StageArgs& st, const fr_table* U, const fr_table* I, const fr_focf_batch* claim, int32_t claim_stamp,
                       int32_t claim_gen, const fr_focf_batch* place, int32_t place_stamp, int32_t place_gen,
                       int32_t sweep_period, uint64_t* row_words, uint32_t* err_flag, int32_t claim_step, int32_t place_step) {
    if (claim_step < 1) claim_step = claim_stamp;
    if (place_step < 1) place_step = place_stamp;
    st = StageArgs{};
    st.Ulast = U->last; st.Ustamp = U->stamp; st.Ilast = I->last; st.Istamp = I->stamp;
    st.n_rows_u = (int)U->n_rows; st.n_rows_i = (int)I->n_rows;
    st.cap = sweep_period;
    st.err = err_flag;
    FR_CHECK_ARG(row_words && U->n_rows <= INT32_MAX && I->n_rows <= INT32_MAX, "fr_focf_stage: row words missing");
    FR_CHECK_ARG(U->stamp && I->stamp && U->last && I->last, "fr_focf_stage: the tables need their last / stamp arrays");
    auto slice_of = [&](int32_t step) {
        fr_table tu = *U, ti = *I;
        tu.step = ti.step = step;
        return make_sweep_slice(&tu, &ti, sweep_period);
    };
    if (claim) {
        const fr_focf_batch& b = *claim;
        FR_CHECK_ARG(b.user && b.item && b.rating && b.ws && b.B >= 1 && b.B <= FR_SORT_MAX && claim_stamp >= 1 &&
                         claim_gen >= 0 && claim_gen < 3, "fr_focf_stage: batch to claim");
        const FocfWs w = focf_layout(b.ws, b.B, U->dim);
        FR_CHECK_ARG(b.ws_bytes >= w.bytes, "fr_focf_stage: workspace %zu < %zu bytes", b.ws_bytes, w.bytes);
        ClaimJob& J = st.c;
        J.user = b.user; J.item = b.item; J.rating = b.rating; J.sst = b.sst;
        J.rec = w.rec; J.info = w.info; J.cp = w.cp;
        J.hcu = words_of(row_words, claim_gen, U, I, false);
        J.hci = words_of(row_words, claim_gen, U, I, true);
        J.B = (int)b.B; J.stamp = claim_stamp;
        st.nb_claim = (int)((b.B + STAGE_BLOCK - 1) / STAGE_BLOCK);
        if (sweep_period > 0) {
            const SweepSlice sw = slice_of(claim_step);
            const long long n_pairs = ((long long)sw.n_u + 1) / 2 + ((long long)sw.n_i + 1) / 2;
            if (n_pairs <= SWEEP_ORDER_MAX) {
                J.lo_u = sw.lo_u; J.lo_i = sw.lo_i; J.n_u = sw.n_u; J.n_i = sw.n_i;
                J.sw_tmp = w.sw_tmp;
                st.nb_sa = (int)((n_pairs + STAGE_BLOCK - 1) / STAGE_BLOCK);
            }
        }
    }
    if (place) {
        const fr_focf_batch& b = *place;
        FR_CHECK_ARG(b.ws && b.B >= 1 && b.B <= FR_SORT_MAX && place_stamp >= 1 && place_gen >= 0 && place_gen < 3,
                     "fr_focf_stage: batch to place");
        const FocfWs w = focf_layout(b.ws, b.B, U->dim);
        FR_CHECK_ARG(b.ws_bytes >= w.bytes, "fr_focf_stage: workspace %zu < %zu bytes", b.ws_bytes, w.bytes);
        PlaceJob& J = st.p;
        J.rec = w.rec; J.info = w.info; J.task_rec = w.task_rec; J.task_info = w.task_info;
        J.cp = w.cp; J.hdr = w.nseg_i;
        J.hcu = words_of(row_words, place_gen, U, I, false);
        J.hci = words_of(row_words, place_gen, U, I, true);
        J.B = (int)b.B; J.stamp = place_stamp;
        st.nb_place = (int)((b.B + STAGE_THREADS * PLACE_EPT - 1) / (STAGE_THREADS * PLACE_EPT));
        if (sweep_period > 0) {
            const SweepSlice sw = slice_of(place_step);
            const long long n_pairs = ((long long)sw.n_u + 1) / 2 + ((long long)sw.n_i + 1) / 2;
            if (n_pairs <= SWEEP_ORDER_MAX) {
                J.n_pairs = (int)n_pairs;
                J.sw_tmp = w.sw_tmp; J.sw_order = w.sw_order;
                st.nb_sb = (int)((n_pairs + STAGE_BLOCK - 1) / STAGE_BLOCK);
            }
        }
    }
    return FR_OK;
}

extern "C" size_t fr_focf_row_words(int64_t n_users, int64_t n_items) { return 3 * ((size_t)n_users + (size_t)n_items); }

extern "C" int fr_focf_stage(const fr_table* U, const fr_table* I, const fr_focf_batch* claim, int32_t claim_stamp,
                             int32_t claim_gen, const fr_focf_batch* place, int32_t place_stamp, int32_t place_gen,
                             int32_t sweep_period, uint64_t* row_words, uint32_t* err_flag, void* stream_) {
    int rc;
    if ((rc = check_table(U, "fr_focf_stage(U)")) || (rc = check_table(I, "fr_focf_stage(I)"))) return rc;
    FR_CHECK_ARG(U->dim == I->dim && (claim || place), "fr_focf_stage: nothing to do");
    StageArgs st;
    // (U->step: the step the batch to place -- without one, the batch to claim -- will be applied at; with both, the claimed batch
    // comes one step later)
    if ((rc = make_stages(st, U, I, claim, claim_stamp, claim_gen, place, place_stamp, place_gen, sweep_period, row_words,
                          err_flag, place ? U->step + 1 : U->step, U->step)))
        return rc;
    const int nb = st.nb_claim + st.nb_sa + st.nb_place + st.nb_sb;
    ProfScope prof(K_FOCF_STAGE, (hipStream_t)stream_);
    FR_LAUNCH(prof, focf_stage_kernel, dim3(nb), dim3(STAGE_THREADS), 0, (hipStream_t)stream_, st);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

static int focf_step_impl(const fr_table* U, const fr_table* I, const fr_adam* adam, const float* sst, int64_t B,
                          int32_t objective, float fair_weight, int32_t sweep_period, int32_t stamp, void* ws,
                          size_t ws_bytes, void* prev_ws, int64_t prev_B, float* prev_loss_out, float* loss_acc,
                          uint32_t* err_flag, void* stream_, const StageArgs* stages, uint64_t* row_words, int32_t gen) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool staged = row_words != nullptr;
    int rc;
    if ((rc = check_table(U, "fr_focf_step(U)")) || (rc = check_table(I, "fr_focf_step(I)")) ||
        (rc = check_adam(adam, "fr_focf_step")))
        return rc;
    FR_CHECK_ARG(U->dim == I->dim, "fr_focf_step: user dim %d != item dim %d", U->dim, I->dim);
    FR_CHECK_ARG(ws, "fr_focf_step: null pointer");
    FR_CHECK_ARG(objective >= FR_FOCF_NONE && objective <= FR_FOCF_OVER,
                 "fr_focf_step: objective %d needs batch-wide statistics before the update (use fr_focf_forward)", objective);
    FR_CHECK_ARG(objective == FR_FOCF_NONE || sst, "fr_focf_step: sst column required for a fairness objective");
    FR_CHECK_ARG(B >= 1 && B <= FR_SORT_MAX, "fr_focf_step: batch size %lld not in 1..%d", (long long)B, FR_SORT_MAX);
    FR_CHECK_ARG(U->step >= 1 && U->step == I->step, "fr_focf_step: table.step must be the step being applied (>=1), "
                 "the same for both tables");
    FR_CHECK_ARG(!U->step_dev && !I->step_dev, "fr_focf_step: device step counters are not supported");
    StepArgs a{};
    const FocfWs w = focf_layout(ws, B, U->dim);
    FR_CHECK_ARG(ws_bytes >= w.bytes, "fr_focf_step: workspace %zu < %zu bytes", ws_bytes, w.bytes);
    a.Up = U->p; a.Um = U->m; a.Uv = U->v; a.Ulast = U->last; a.Ustamp = U->stamp;
    a.Ip = I->p; a.Im = I->m; a.Iv = I->v; a.Ilast = I->last; a.Istamp = I->stamp;
    a.D = U->dim;
    a.step = U->step;
    a.c = make_adamc(adam);
    a.B = (int)B;
    a.objective = objective;
    a.fair_weight = fair_weight;
    a.task_rec = w.task_rec;
    a.task_info = w.task_info;
    static const int lead_pct = getenv("FAIRREC_STEP_LEAD") ? atoi(getenv("FAIRREC_STEP_LEAD")) : 100;

    a.hdr = w.nseg_i;
    a.mse_e = w.mse_e;
    a.term = w.term;
    a.ws = ws;
    a.err = err_flag;
    long long sweep_waves = 0;
    if (sweep_period > 0) {
        const SweepSlice sw = make_sweep_slice(U, I, sweep_period);
        a.lo_u = sw.lo_u; a.lo_i = sw.lo_i; a.n_u = sw.n_u; a.n_i = sw.n_i;
        a.skip_from = stamp;      // the rows of this batch (and of batches prepared for later steps) carry stamps >= it
        sweep_waves = ((long long)a.n_u + 1) / 2 + ((long long)a.n_i + 1) / 2;
        a.sw_order = sweep_waves <= SWEEP_ORDER_MAX ? w.sw_order : nullptr;
    }
    a.prev = prev_of(prev_ws, prev_B, U->dim, objective, fair_weight, prev_loss_out, loss_acc, staged);
    StageArgs st{};
    if (staged) {
        a.hcu = words_of(row_words, gen, U, I, false);
        a.hci = words_of(row_words, gen, U, I, true);
        if (stages) st = *stages;
    }
    {
        ProfScope prof(K_FOCF_STEP, stream);
        const long long per_wave = step_pairs((U->dim + 63) / 64) ? 2 : 1;
        const long long inter_blocks = ((B + per_wave - 1) / per_wave + STEP_WPB - 1) / STEP_WPB;
        const long long sweep_wg = (sweep_waves + STEP_WPB - 1) / STEP_WPB;
        const unsigned blocks = (unsigned)(1 + st.nb_claim + st.nb_sa + st.nb_place + st.nb_sb + sweep_wg + inter_blocks);
        a.lead = (int)(inter_blocks * lead_pct / 100);
        const dim3 block(64 * STEP_WPB);
        const int n_stage = st.nb_claim + st.nb_sa + st.nb_place + st.nb_sb;
#define STEP_PRE a.task_rec, a.sw_order, a.B, a.lead, a.n_u, a.n_i, n_stage, 0
        if (staged) {
            if (U->dim % 64 == 0) {
                FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_step_kernel<E, true, true>), dim3(blocks), block, 0, stream, STEP_PRE, a, st));
            } else {
                FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_step_kernel<E, false, true>), dim3(blocks), block, 0, stream, STEP_PRE, a, st));
            }
        } else if (U->dim % 64 == 0) {
            FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_step_kernel<E, true, false>), dim3(blocks), block, 0, stream, STEP_PRE, a, st));
        } else {
            FR_DISPATCH_E(U->dim, FR_LAUNCH(prof, (focf_step_kernel<E, false, false>), dim3(blocks), block, 0, stream, STEP_PRE, a, st));
        }
#undef STEP_PRE
    }
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_focf_step(const fr_table* U, const fr_table* I, const fr_adam* adam, const int64_t* user,
                            const int64_t* item, const float* rating, const float* sst, int64_t B, int32_t objective,
                            float fair_weight, int32_t sweep_period, int32_t stamp, void* ws, size_t ws_bytes,
                            float* loss_out, void* prev_ws, int64_t prev_B, float* prev_loss_out, float* loss_acc,
                            uint32_t* err_flag, void* stream_) {
    (void)user; (void)item; (void)rating;    // fr_focf_prepare_step packed them into the workspace
    (void)loss_out;                          // reduced by the NEXT fr_focf_step (prev_*) or by fr_focf_step_finish
    return focf_step_impl(U, I, adam, sst, B, objective, fair_weight, sweep_period, stamp, ws, ws_bytes, prev_ws, prev_B,
                          prev_loss_out, loss_acc, err_flag, stream_, nullptr, nullptr, 0);
}

extern "C" int fr_focf_step_staged(const fr_table* U, const fr_table* I, const fr_adam* adam, const float* sst, int64_t B,
                                   int32_t objective, float fair_weight, int32_t sweep_period, int32_t stamp, int32_t gen,
                                   void* ws, size_t ws_bytes, void* prev_ws, int64_t prev_B, float* prev_loss_out,
                                   float* loss_acc, uint64_t* row_words, const fr_focf_batch* claim, int32_t claim_stamp,
                                   int32_t claim_gen, const fr_focf_batch* place, int32_t place_stamp, int32_t place_gen,
                                   uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(row_words && gen >= 0 && gen < 3, "fr_focf_step_staged: row words / generation");
    FR_CHECK_ARG((!claim || claim_gen != gen) && (!place || place_gen != gen) && (!claim || !place || claim_gen != place_gen),
                 "fr_focf_step_staged: the three batches in flight need three different generations of row words");
    StageArgs st{};
    if (claim || place) {
        int rc;
        if ((rc = check_table(U, "fr_focf_step_staged(U)")) || (rc = check_table(I, "fr_focf_step_staged(I)"))) return rc;
        // (the batch to place is applied at the next step, the batch to claim at the one after)
        if ((rc = make_stages(st, U, I, claim, claim_stamp, claim_gen, place, place_stamp, place_gen, sweep_period, row_words,
                              err_flag, U->step + 2, U->step + 1)))
            return rc;
    }
    return focf_step_impl(U, I, adam, sst, B, objective, fair_weight, sweep_period, stamp, ws, ws_bytes, prev_ws, prev_B,
                          prev_loss_out, loss_acc, err_flag, stream_, &st, row_words, gen);
}

// n consecutive steps of fr_focf_step_staged in one call: the step loop of trainer.py:181-196 over a run of batches, issued
// by the library instead of by one interpreter round trip per step.  Batch k is applied at step U->step + k with stamp
// first_stamp + k and generation (first_gen + k) % 3, carries the place stage of batch k + 1 and the claim stage of batch
// k + 2, and reduces the loss of batch k - 1 (of `prev_*` for k = 0).  The first two batches' stages take two launches of
// their own (fr_focf_stage).  The LAST batch's loss is left to the caller (the next call's prev_*, or
// fr_focf_step_finish_staged), exactly as after a single fr_focf_step_staged.
extern "C" int fr_focf_steps_many(const fr_table* U, const fr_table* I, const fr_adam* adam, const fr_focf_batch* batches,
                                  int32_t n, int32_t objective, float fair_weight, int32_t sweep_period, int32_t first_stamp,
                                  int32_t first_gen, void* prev_ws, int64_t prev_B, float* prev_loss_out, float* loss_ring,
                                  int32_t loss_slots, int32_t first_slot, float* loss_acc, uint64_t* row_words,
                                  uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(U && I && batches && n >= 1, "fr_focf_steps_many: null pointer / no batch");
    FR_CHECK_ARG(loss_ring && loss_slots >= 1 && first_slot >= 0 && first_slot < loss_slots,
                 "fr_focf_steps_many: loss ring (float[4 * loss_slots]) and a first slot inside it");
    FR_CHECK_ARG(first_stamp >= 1 && first_stamp <= INT32_MAX - n && first_gen >= 0 && first_gen < 3,
                 "fr_focf_steps_many: first stamp / generation");
    FR_CHECK_ARG(U->step >= 1 && U->step == I->step && U->step <= INT32_MAX - n,
                 "fr_focf_steps_many: table.step must be the step the FIRST batch is applied at (>= 1), the same for both tables");
    // a workspace is busy from its batch's claim (two launches before its step) to the launch after its step (loss
    // reduction, counters back to zero): any four consecutive batches need four different ones
    for (int k = 0; k < n; ++k) {
        FR_CHECK_ARG(batches[k].ws && batches[k].user && batches[k].item && batches[k].rating,
                     "fr_focf_steps_many: batch %d: null pointer", k);
        for (int j = k + 1; j < n && j <= k + 3; ++j)
            FR_CHECK_ARG(batches[j].ws != batches[k].ws, "fr_focf_steps_many: batches %d and %d share a workspace", k, j);
        FR_CHECK_ARG(k > 2 || batches[k].ws != prev_ws,
                     "fr_focf_steps_many: batch %d uses the workspace whose loss is still to be reduced", k);
    }
    auto gen_of = [&](int k) { return (first_gen + k) % 3; };
    int rc;
    if ((rc = fr_focf_stage(U, I, &batches[0], first_stamp, gen_of(0), nullptr, 0, 0, sweep_period, row_words, err_flag, stream_)))
        return rc;
    if ((rc = fr_focf_stage(U, I, n > 1 ? &batches[1] : nullptr, first_stamp + 1, gen_of(1), &batches[0], first_stamp, gen_of(0),
                            sweep_period, row_words, err_flag, stream_)))
        return rc;
    fr_table tu = *U, ti = *I;
    for (int k = 0; k < n; ++k) {
        const fr_focf_batch& b = batches[k];
        const fr_focf_batch* place = k + 1 < n ? &batches[k + 1] : nullptr;
        const fr_focf_batch* claim = k + 2 < n ? &batches[k + 2] : nullptr;
        rc = fr_focf_step_staged(&tu, &ti, adam, b.sst, b.B, objective, fair_weight, sweep_period, first_stamp + k, gen_of(k),
                                 b.ws, b.ws_bytes, prev_ws, prev_B, prev_loss_out, loss_acc, row_words, claim,
                                 first_stamp + k + 2, gen_of(k + 2), place, first_stamp + k + 1, gen_of(k + 1), err_flag, stream_);
        if (rc) return rc;
        prev_ws = b.ws; prev_B = b.B;
        prev_loss_out = loss_ring + 4 * (size_t)((first_slot + k) % loss_slots);
        ++tu.step; ++ti.step;
    }
    return FR_OK;
}

#ifdef FR_LPT_STAMPS
extern "C" __attribute__((visibility("default"))) int fr_debug_lpt_stamps(unsigned long long* host_out) {
    FR_CHECK_HIP(hipDeviceSynchronize());
    FR_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_lpt_stamps), sizeof(unsigned long long) * 8));
    return FR_OK;
}
#endif

#if FR_STEP_TRACE
extern "C" __attribute__((visibility("default"))) int fr_debug_set_trace_step(int step) {
    FR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_step), &step, sizeof(int)));
    return FR_OK;
}
extern "C" __attribute__((visibility("default"))) int fr_debug_step_trace(unsigned long long* host_out, int n_waves) {
    FR_CHECK_HIP(hipDeviceSynchronize());
    FR_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_step_trace), (size_t)n_waves * 32));
    FR_CHECK_HIP(hipMemcpyFromSymbol(host_out + 4 * (size_t)n_waves, HIP_SYMBOL(g_step_trace), (size_t)n_waves * 32,
                                     (size_t)65536 * 32));
    return FR_OK;
}
#endif

extern "C" int fr_focf_step_finish(void* ws, size_t ws_bytes, int64_t B, int32_t dim, int32_t objective,
                                   float fair_weight, float* loss_out, float* loss_acc, void* stream_) {
    FR_CHECK_ARG(ws && loss_out && B >= 1 && B <= FR_SORT_MAX && dim >= 1, "fr_focf_step_finish: bad argument");
    FR_CHECK_ARG(ws_bytes >= focf_layout(nullptr, B, dim).bytes, "fr_focf_step_finish: workspace too small");
    const PrevLoss pl = prev_of(ws, B, dim, objective, fair_weight, loss_out, loss_acc);
    hipLaunchKernelGGL(focf_step_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, pl);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_focf_step_finish_staged(void* ws, size_t ws_bytes, int64_t B, int32_t dim, int32_t objective,
                                          float fair_weight, float* loss_out, float* loss_acc, void* stream_) {
    FR_CHECK_ARG(ws && loss_out && B >= 1 && B <= FR_SORT_MAX && dim >= 1, "fr_focf_step_finish_staged: bad argument");
    FR_CHECK_ARG(ws_bytes >= focf_layout(nullptr, B, dim).bytes, "fr_focf_step_finish_staged: workspace too small");
    const PrevLoss pl = prev_of(ws, B, dim, objective, fair_weight, loss_out, loss_acc, true);
    hipLaunchKernelGGL(focf_step_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, pl);
    FR_CHECK_LAUNCH();
    return FR_OK;
}
