// Internal (non-ABI) declarations shared between the .hip translation units.
#pragma once
#include "common.hpp"
#include <hip/hip_ext.h>

namespace fr {

// kernel kinds known to the built-in HIP-event profiler (fr_prof_*)
enum KernelKind {
    K_SORT = 0, K_FOCF_GATHER, K_FOCF_FAIR, K_FOCF_NONPARITY, K_FOCF_FINALIZE, K_FOCF_BWD_ADAM, K_TABLE_FLUSH,
    K_TABLE_GATHER, K_ADAM_DENSE, K_TABLE_GATHER_TRAIN, K_TABLE_APPLY_GRAD, K_BUCKET, K_UNBUCKET,
    K_BUCKET_ROWS, K_FOCF_SHARD_SCORE, K_FOCF_SHARD_GRADS, K_LINEAR_FWD,
    K_LINEAR_BWD_INPUT, K_LINEAR_BWD_WEIGHT, K_NFCF_LOSS, K_BN_FWD, K_BN_BWD, K_ROWDOT, K_BPR, K_SPMM, K_ROW_GATHER, K_SAMPLE_NEG, K_FOCF_STEP, K_FOCF_LPT, K_FOCF_STAGE, K_COUNT
};
bool prof_on();
// algorithmic work of a launch of `kind` (FLOP of a dense product, bytes of an SpMM), summed while the profiler is on
void prof_work(int kind, double amount);
// Takes an event pair from the profiler's pool and registers it for kernel `kind` (not recorded here: the pair
// is handed to hipExtLaunchKernelGGL, which stamps it at the kernel's own start and end on the GPU).
bool prof_take(int kind, hipEvent_t* start, hipEvent_t* stop);

struct ProfScope {
    int kind;
    hipStream_t s;
    ProfScope(int k, hipStream_t st) : kind(k), s(st) {}
};

// Launch `kernel`; when the profiler is on, with start/stop events that time exactly this kernel.
template <typename K, typename... Args>
inline void launch_kernel(const ProfScope& prof, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                          Args... args) {
    hipEvent_t a, b;
    if (prof_on() && prof_take(prof.kind, &a, &b))
        hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, a, b, 0, args...);
    else
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
}
#define FR_LAUNCH(prof, kernel, grid, block, lds, stream, ...) \
    ::fr::launch_kernel(prof, kernel, grid, block, lds, stream, __VA_ARGS__)

// A helper stream + fork/join events owned by the library (created on first use, one per process = one per GPU).
// nullptr when disabled with FAIRREC_NO_OVERLAP=1.
struct SideStream {
    hipStream_t stream;
    hipEvent_t fork, join;
};
SideStream* side_stream();

// Deferred join for work launched on the side stream on behalf of a workspace (the index sort of
// fr_table_gather_train): side_mark(ws) records its completion, side_join(ws, stream) makes `stream` wait for it
// (no-op when nothing is pending).  The consumers of the sort results (fr_table_apply_grad, fr_focf_shard_fair)
// join; everything launched in between overlaps with the sort.
int side_mark(const void* ws);
int side_join(const void* ws, hipStream_t stream);

struct SortJob {
    const int64_t* idx;
    int64_t n_rows;
    int32_t* perm;
    int32_t* seg_start;
    int32_t* seg_row;
    int32_t* seg_of;
    int32_t* n_seg;
    const float* aux;   // optional float column to min/max-reduce alongside (may be null)
    float* aux_minmax;  // [2]
    Lay lay;            // layout of idx (zero-initialised = dense)
    int32_t* seg_first; // optional [M]: perm[seg_start[k]] of segment k, so that its consumers need not chase perm for
                        //   the first (usually the only) member: one dependent memory round trip less per wave
    // optional outputs for the fused FOCF step (focf_step.hip):
    int2* info;         // per position b: (first sorted position j0 of its segment | members n << 16, segment index), at
    int info_stride;    //   info[b * info_stride] (two lists can interleave their halves of one 16-byte record per b)
    unsigned int* cnt;  // [n_seg] arrival counters of the segments, zeroed here
    int32_t* stamp;     // table stamps: stamp[row] = max(stamp[row], stamp_val) for every row of the list, so that the
    int stamp_val;      //   sweeper waves of the step's launch leave the rows of the batch alone ...
    const int32_t* last;   // ... and, alongside, last_out[j] = last[row of position j]: how stale each row is now (the
    int32_t* last_out;     //   launch order of the step is derived from it); both by extra workgroups of the sort launch
    // ... and one 16-byte record per position for its consumer: (row id of `rec_idx`, row id of this list, rec_f0, aux),
    // both ids range-checked and clamped like the sort keys; a -1 id counts as out of range here (no padding holes)
    int4* rec;
    const int64_t* rec_idx;
    int64_t rec_rows;
    const float* rec_f0;
    int32_t* pos_of;       // optional [M]: sorted position j of batch position b (the inverse of perm), for consumers that park
                           //   per-member data in SORTED order so that a segment's members sit side by side (focf_runs.hip)
};

// Sort index lists in one launch, one workgroup each: (a) or (a, b) of the same length M ...
int launch_sort(const SortJob& a, const SortJob* b, int64_t M, uint32_t* err, hipStream_t stream);

// ... or up to FR_SORT_JOBS lists of their own lengths (the id columns of several coming batches at once: the sort is
// one latency-bound workgroup per list, so n lists cost the time of one)
static constexpr int FR_SORT_JOBS = 16;
struct SortJobList {
    SortJob j[FR_SORT_JOBS];
    int M[FR_SORT_JOBS];
    int n;
};
int launch_sort_many(const SortJobList& jobs, int64_t n_rows_max, uint32_t* err, hipStream_t stream);

// Device view of a lazy-Adam table.
struct TableV {
    float* p;
    float* m;
    float* v;
    int32_t* last;
    int32_t* stamp;
    long long n_rows;
    int D;
    int step;
    const int32_t* step_dev;   // optional device-resident counter: effective step = *step_dev + step (see fr_table)
};

inline TableV view(const fr_table* t) {
    return TableV{t->p, t->m, t->v, t->last, t->stamp, (long long)t->n_rows, t->dim, t->step, t->step_dev};
}

// the table view with its effective step (one scalar load when a device counter is attached)
__device__ __forceinline__ TableV resolved(TableV T) {
    if (T.step_dev) T.step += *T.step_dev;
    return T;
}

inline int check_table(const fr_table* t, const char* who) {
    if (!t || !t->p || !t->m || !t->v || !t->last || !t->stamp) {
        set_error("%s: table has null pointers", who);
        return FR_EINVAL;
    }
    if (t->dim < 1 || t->dim > 256) {
        set_error("%s: embedding dim %d not in 1..256", who, t->dim);
        return FR_EUNSUPPORTED;
    }
    if (t->n_rows < 1 || t->n_rows > 0x7fffffffLL) {
        set_error("%s: n_rows %lld not in 1..2^31-1", who, (long long)t->n_rows);
        return FR_EUNSUPPORTED;
    }
    return FR_OK;
}

inline int check_adam(const fr_adam* a, const char* who) {
    if (!a || !a->scalars || a->cap < 1) {
        set_error("%s: bad fr_adam", who);
        return FR_EINVAL;
    }
    return FR_OK;
}

// Sweep slice of a table for optimizer step `step` with period S: rows [lo, hi).
__host__ __device__ inline void sweep_range(long long n_rows, int step, int S, long long& lo, long long& hi) {
    if (S <= 0) {
        lo = hi = 0;
        return;
    }
    long long chunk = (n_rows + S - 1) / S;
    lo = (long long)(step % S) * chunk;
    hi = lo + chunk;
    if (lo > n_rows) lo = n_rows;
    if (hi > n_rows) hi = n_rows;
}

}  // namespace fr
