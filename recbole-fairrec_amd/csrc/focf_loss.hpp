// Loss bookkeeping shared by the one-launch FOCF steps (focf_step.hip: one wave per interaction; focf_runs.hip: one
// workgroup per run of an item-complete batch): a step leaves per-interaction squared errors and per-item smooth-L1 terms in
// its workspace, and ONE workgroup of the next launch (or fr_focf_step_finish) reduces them in a fixed order.
#pragma once
#include "common.hpp"
#include "kernels.hpp"
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
    int by_pos;        // in-launch prepare: `term` is indexed by batch position (an item's term at its first member, 0 elsewhere)
    int32_t* cp;       // ... and the batch's counters are zeroed once its loss is reduced: the workspace is free again
};

// fixed-order reduction of one batch's squared errors and per-item terms -> loss (one workgroup of 256 threads).  The
// association is that of the three-launch path (per 4 interactions, then strided over 256 threads, butterfly, 4 waves;
// terms per 64 items, then the same), so both paths report the same bits.
template <int NT>     // threads of the calling workgroup: 64, 128 or 256 (they stand in for 256 "virtual" threads)
__device__ __forceinline__ void step_reduce_loss(const PrevLoss& pl) {
    __shared__ float red[2][4];
    constexpr int VT = 256 / NT;      // virtual threads per thread: virtual thread j * NT + threadIdx.x, its wave = that / 64
    const int B = pl.B;
    const int nb = (B + 3) / 4;
    const bool per_item = pl.objective >= FR_FOCF_VALUE && pl.objective <= FR_FOCF_OVER;
    const int K = pl.nseg_i[0];
#pragma unroll
    for (int j = 0; j < VT; ++j) {
        const int vt = j * NT + (int)threadIdx.x;
        float a = 0.f, f = 0.f;
        for (int q = vt; q < nb; q += 256) {
            const int b0 = 4 * q;
            const float e0 = pl.mse_e[b0], e1 = b0 + 1 < B ? pl.mse_e[b0 + 1] : 0.f, e2 = b0 + 2 < B ? pl.mse_e[b0 + 2] : 0.f,
                        e3 = b0 + 3 < B ? pl.mse_e[b0 + 3] : 0.f;
            a += ((e0 + e1) + e2) + e3;
        }
        if (per_item && pl.by_pos) {      // one term per batch position (zero where no item has its first member): as `a`
            for (int q = vt; q < nb; q += 256) {
                const int b0 = 4 * q;
                const float e0 = pl.term[b0], e1 = b0 + 1 < B ? pl.term[b0 + 1] : 0.f, e2 = b0 + 2 < B ? pl.term[b0 + 2] : 0.f,
                            e3 = b0 + 3 < B ? pl.term[b0 + 3] : 0.f;
                f += ((e0 + e1) + e2) + e3;
            }
        } else if (per_item) {
            constexpr int PER = FAIR_THREADS / FAIR_GROUP;     // items per workgroup of the fairness launch
            const int nf = (B * FAIR_GROUP + FAIR_THREADS - 1) / FAIR_THREADS;
            for (int q = vt; q < nf; q += 256) {
                float sblk = 0.f;
                const int k1 = min(K, (q + 1) * PER);
                for (int k = q * PER; k < k1; ++k) sblk += pl.term[k];
                f += sblk;
            }
        }
        a = wave_sum(a);
        f = wave_sum(f);
        if ((threadIdx.x & 63) == 0) { red[0][vt >> 6] = a; red[1][vt >> 6] = f; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float a = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        const float f = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
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
            // a sticky record of the FIRST step whose loss was not a number (trainer.py:192 raises at that step; here the
            // epoch's sum is read once, and this says where it went wrong): [3] = steps reduced into the total so far,
            // [4] = 1-based index of the first NaN step among them, 0 = none (fr_loss_accumulate keeps the same record)
            pl.acc[3] += 1.f;
            if (loss != loss && pl.acc[4] == 0.f) pl.acc[4] = pl.acc[3];
        }
    }
    if (pl.cp && threadIdx.x < FOCF_CP_INTS) pl.cp[threadIdx.x] = 0;      // nothing of the batch is needed any more
}

inline PrevLoss prev_of(void* ws, int64_t B, int dim, int objective, float fair_weight, float* loss_out, float* acc,
                        bool staged = false) {
    PrevLoss pl{};
    if (!ws || !loss_out) return pl;
    const FocfWs w = focf_layout(ws, B, dim);
    pl.by_pos = staged ? 1 : 0;
    pl.cp = staged ? w.cp : nullptr;
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
