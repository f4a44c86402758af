// Shared pieces of the FOCF kernels (focf.hip: forward / fairness / backward chain; focf_step.hip: fused step).
#pragma once
#include "common.hpp"
#include "kernels.hpp"
#include "table.hpp"

namespace fr {

// FR_FOCF_DEFER_LOSS: the loss of the batch is reduced by one extra workgroup of the backward launch instead of inside
// the forward (where it costs a ticket round trip at the end of the fairness kernel, on the step's critical path).  The
// record lives in the workspace (device memory, written by the forward launch): the entry points keep no host state.
struct DeferLoss {
    float* loss_out;   // nullptr = nothing deferred
    int objective;
    float fair_weight;
    int n_fair_part;   // workgroups of the fairness launch that wrote a partial sum
};

struct FocfWs {
    // sort outputs
    int32_t *perm_u, *seg_start_u, *seg_row_u, *seg_first_u, *nseg_u;   // seg_first[k] = perm[seg_start[k]]
    int32_t *perm_i, *seg_start_i, *seg_row_i, *seg_first_i, *nseg_i;
    float* sst_minmax;   // [2]
    float* pred;         // [B]
    float* coef;         // [B] dLoss/dpred
    float* mse_part;     // [gather blocks]
    float* fair_part;    // [fair blocks]
    unsigned int* ticket;  // in-launch finalisation counter of the fair kernel (kept zero between launches)
    float* clip_part;      // [(2B + 3) / 4] squared-norm partials of fr_focf_clip_grad_norm
    float* side[6];      // ue, mu, vu, ie, mi, vi : [B, D] each
    // fused step (focf_step.hip): per batch position ONE record (user row, item row, rating, sst) and ONE record
    // (first sorted position j0 | members n << 16, segment index) x (user segment, item segment); arrival counters per
    // segment, per-interaction squared errors, per-item smooth-L1 terms
    int4* rec;
    int4* info;
    // ... and the same two records in the order the launch should START the interactions (longest replay first), the
    // second one repacked as (user j0 | n << 16, item j0 | n << 16, user segment | item segment << 16, batch position)
    int4* task_rec;
    int4* task_info;
    unsigned int *cnt_u, *cnt_i;
    int32_t *age_u, *age_i;   // [B] `last` stamp of each position's user / item row as of the prepare launch
    int32_t* pos_i;           // [B] sorted position of each batch position in the item order (inverse of perm_i; fr_focf_prepare_step)
    float* mse_e;        // [B]
    float* term;         // [B] indexed by item segment
    int32_t* sw_order;   // [SWEEP_ORDER_MAX + 1] start order of the step's sweeper tasks (pairs of rows of the sweep slice),
                         //   followed by the number of pairs it was built for (focf_sweep_order_kernel)
    // in-launch prepare (focf_step.hip, stages riding in the step launches): counters of the batch -- [0] / [1] places taken
    // from the front / the back of the task list, [8] distinct items, [9] / [10] ordered-int maxima of sst / -sst, [11] / [12] fill of the member
    // lists of shared user / item rows, [16..31] sweeper tasks per start class.  Zero whenever no batch owns the workspace.
    int32_t* cp;         // [FOCF_CP_INTS]
    int32_t* sw_tmp;     // [SWEEP_ORDER_MAX] class << 20 | rank within the class of each sweeper task
    DeferLoss* defer;    // [1] written by the forward launch, consumed (and cleared) by the backward launch
    int n_gather_blocks, n_fair_blocks;
    size_t bytes;
};

constexpr int FOCF_CP_INTS = 32;
constexpr int SWEEP_ORDER_MAX = 16384;   // pairs of rows per sweep slice that get a start order (beyond: index order)
constexpr int GATHER_THREADS = 256;  // 4 waves = 4 interactions per block
constexpr int FAIR_THREADS = 1024;   // 16 lanes per item segment -> 64 segments per block (few blocks: cheap ticket)
constexpr int FAIR_GROUP = 16;        // lanes per item segment ...
constexpr int FAIR_GROUP_RUNS = 64;   // ... and in item-complete batches (few items, ~100 members each)

__host__ __device__ inline FocfWs focf_layout(void* base, int64_t B, int D) {
    FocfWs w;
    size_t off = 0;
    auto take = [&](size_t nbytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off = (off + nbytes + 255) / 256 * 256;
        return p;
    };
    const size_t Bp = (size_t)B + 1;
    w.cp = (int32_t*)take(FOCF_CP_INTS * 4);     // first: at the same place whatever B the buffer is used for next
    w.perm_u = (int32_t*)take(Bp * 4);
    w.seg_start_u = (int32_t*)take(Bp * 4);
    w.seg_row_u = (int32_t*)take(Bp * 4);
    w.seg_first_u = (int32_t*)take(Bp * 4);
    w.nseg_u = (int32_t*)take(4);
    w.perm_i = (int32_t*)take(Bp * 4);
    w.seg_start_i = (int32_t*)take(Bp * 4);
    w.seg_row_i = (int32_t*)take(Bp * 4);
    w.seg_first_i = (int32_t*)take(Bp * 4);
    w.nseg_i = (int32_t*)take(16);                  // header of the fused step: (K, -, smin, smax) in one 16-byte read
    w.sst_minmax = (float*)(base ? w.nseg_i + 2 : nullptr);
    w.pred = (float*)take(Bp * 4);
    w.coef = (float*)take(Bp * 4);
    w.n_gather_blocks = (int)((B * WAVE + GATHER_THREADS - 1) / GATHER_THREADS);
    w.n_fair_blocks = (int)((B * FAIR_GROUP + FAIR_THREADS - 1) / FAIR_THREADS);
    if (w.n_gather_blocks < 1) w.n_gather_blocks = 1;
    if (w.n_fair_blocks < 1) w.n_fair_blocks = 1;
    w.mse_part = (float*)take((size_t)w.n_gather_blocks * 4);
    w.fair_part = (float*)take((size_t)w.n_fair_blocks * (FAIR_GROUP_RUNS / FAIR_GROUP) * 4);   // room for either group size
    w.ticket = (unsigned int*)take(4);
    w.clip_part = (float*)take(((2 * (size_t)B + 3) / 4) * 4);
    w.rec = (int4*)take(Bp * 16);
    w.info = (int4*)take(Bp * 16);
    w.task_rec = (int4*)take(Bp * 16);
    w.task_info = (int4*)take(Bp * 16);
    w.cnt_u = (unsigned int*)take(Bp * 4);
    w.cnt_i = (unsigned int*)take(Bp * 4);
    w.age_u = (int32_t*)take(Bp * 4);
    w.age_i = (int32_t*)take(Bp * 4);
    w.pos_i = (int32_t*)take(Bp * 4);
    w.mse_e = (float*)take(Bp * 4);
    w.term = (float*)take(Bp * 4);
    w.sw_order = (int32_t*)take(((size_t)SWEEP_ORDER_MAX + 1) * 4);
    w.sw_tmp = (int32_t*)take((size_t)SWEEP_ORDER_MAX * 4);
    w.defer = (DeferLoss*)take(sizeof(DeferLoss));
    for (int k = 0; k < 6; ++k) w.side[k] = (float*)take((size_t)B * D * 4);
    w.bytes = off;
    return w;
}

// The step's slice of the bounded-staleness sweeper, as extra workgroups of the backward launch: one wave per PAIR of
// rows.  Pure VALU work (up to S replayed steps per row) on rows nothing else in the step touches -- the gather kernel
// has stamped the batch's rows by then.  Measured alternatives (profiles/README.md): its own launch on a second stream,
// riding in the look-ahead sort launch, in the gather or the fairness launch, split between launches -- every one of them
// was slower than this (a sweeper wave is a ~10 us dependency chain wherever it runs; hipGraph serialises a third branch
// and pays ~10 us per cross-stream join).
struct SweepSlice {
    long long lo_u, lo_i;
    int n_u, n_i;          // rows of the slice in each table
    int upto, skip_from;   // rows stamped >= skip_from are left to their batch; the others are brought to step `upto`
    int per_wave;          // rows a wave takes: 2 at D <= 64 (sweep_row_pair), 1 beyond (see sweep_pairs)
};

inline long long sweep_slice_waves(const SweepSlice& sw) {
    return ((long long)sw.n_u + sw.per_wave - 1) / sw.per_wave + ((long long)sw.n_i + sw.per_wave - 1) / sw.per_wave;
}

template <int E>
__device__ __forceinline__ void sweep_slice_wave(const TableV& U, const TableV& I, const AdamC& c, const SweepSlice& sw,
                                                 long long wv, int lane) {
    constexpr int PW = sweep_pairs(E) ? 2 : 1;
    const long long pu = (sw.n_u + PW - 1) / PW, pi = (sw.n_i + PW - 1) / PW;
    if (wv < pu) {
        if (PW == 2) {
            const long long a = 2 * wv;
            sweep_row_pair<E>(U, c, sw.lo_u + a, a + 1 < sw.n_u ? sw.lo_u + a + 1 : -1, sw.upto, sw.skip_from, lane);
        } else {
            sweep_row<E>(U, c, sw.lo_u + wv, sw.upto, sw.skip_from, lane);
        }
        return;
    }
    wv -= pu;
    if (wv < pi) {
        if (PW == 2) {
            const long long a = 2 * wv;
            sweep_row_pair<E>(I, c, sw.lo_i + a, a + 1 < sw.n_i ? sw.lo_i + a + 1 : -1, sw.upto, sw.skip_from, lane);
        } else {
            sweep_row<E>(I, c, sw.lo_i + wv, sw.upto, sw.skip_from, lane);
        }
    }
}

__device__ __forceinline__ float smooth_l1(float x) {  // F.smooth_l1_loss(|x|, 0), beta = 1
    float a = fabsf(x);
    return a < 1.f ? 0.5f * a * a : a - 0.5f;
}

// d_g = f(P_g - T_g) of the four per-item objectives (focf.py:93-125) and its derivative q_g = d d_g / d P_g
__device__ __forceinline__ void focf_objective(int objective, float P0, float T0, float P1, float T1, float& d0,
                                               float& d1, float& q0, float& q1) {
    if (objective == FR_FOCF_VALUE) {
        d0 = P0 - T0; d1 = P1 - T1; q0 = 1.f; q1 = 1.f;
    } else if (objective == FR_FOCF_ABSOLUTE) {
        d0 = fabsf(P0 - T0); d1 = fabsf(P1 - T1);
        q0 = (P0 > T0) ? 1.f : (P0 < T0 ? -1.f : 0.f);
        q1 = (P1 > T1) ? 1.f : (P1 < T1 ? -1.f : 0.f);
    } else if (objective == FR_FOCF_UNDER) {
        d0 = (T0 - P0 > 0.f) ? T0 - P0 : 0.f; d1 = (T1 - P1 > 0.f) ? T1 - P1 : 0.f;
        q0 = (T0 - P0 > 0.f) ? -1.f : 0.f;    q1 = (T1 - P1 > 0.f) ? -1.f : 0.f;
    } else {  // over
        d0 = (P0 - T0 > 0.f) ? P0 - T0 : 0.f; d1 = (P1 - T1 > 0.f) ? P1 - T1 : 0.f;
        q0 = (P0 - T0 > 0.f) ? 1.f : 0.f;     q1 = (P1 - T1 > 0.f) ? 1.f : 0.f;
    }
}

// One distinct item of the batch: from its per-group sums (pred, rating, count) to the smooth-L1 term of the item and
// the fairness part of dLoss/dpred of a member of group 0 / group 1 (focf.py:75-125; `sst_num += 1e-5` at :89).
// kdiv = number of distinct items K (the mean over items), or 1 when the caller divides later (row-sharded path).
__device__ __forceinline__ void focf_fair_eval(int objective, float fair_weight, float kdiv, float sp0, float sp1,
                                               float st0, float st1, float n0, float n1, float& term, float& g0,
                                               float& g1) {
    const float c0 = n0 + 1e-5f, c1 = n1 + 1e-5f;
    const float P0 = sp0 / c0, P1 = sp1 / c1, T0 = st0 / c0, T1 = st1 / c1;
    float d0, d1, q0, q1;  // d_g and d d_g / d P_g
    focf_objective(objective, P0, T0, P1, T1, d0, d1, q0, q1);
    const float delta = d0 - d1;
    const float x = fabsf(delta);
    term = smooth_l1(x);
    const float sgn = delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f);
    const float dx = (x < 1.f ? x : 1.f) * sgn * fair_weight / kdiv;   // d(fw * mean_k sl1) / d delta
    g0 = dx * q0 / c0;
    g1 = -dx * q1 / c1;
}

// focf.hip: focf_gather_kernel<E, TRAIN, SHARE> on an item-complete batch, as a launch of its own (fr_focf_step_runs)
// (the first `sweep_waves` waves of the step's sweep slice `sw` ride in front of the gather)
int focf_launch_gather_runs(const fr_table* U, const fr_table* I, const AdamC& c, const int64_t* user, const int64_t* item,
                            const float* rating, const float* sst, int64_t B, const FocfWs& w, uint32_t* err_flag,
                            hipStream_t stream, const SweepSlice& sw, long long sweep_waves);

inline SweepSlice make_sweep_slice(const fr_table* U, const fr_table* I, int32_t sweep_period) {
    SweepSlice sw{};
    long long hi_u, hi_i;
    sweep_range(U->n_rows, U->step, sweep_period, sw.lo_u, hi_u);
    sweep_range(I->n_rows, I->step, sweep_period, sw.lo_i, hi_i);
    sw.n_u = (int)(hi_u - sw.lo_u);
    sw.n_i = (int)(hi_i - sw.lo_i);
    sw.upto = U->step;
    sw.skip_from = U->step;
    sw.per_wave = sweep_pairs((U->dim + 63) / 64) ? 2 : 1;
    return sw;
}

}  // namespace fr
