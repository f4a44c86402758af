/*
 * fairrec_hip.h -- C ABI of libfairrec_hip.so: the MI355X (gfx950) kernels behind the
 * RecBole-FairRec training hot path (embedding gather -> score -> fairness-regularised loss ->
 * backward -> Adam).
 *
 * The reference has no native boundary at all: its "operators" on this
 * path are stock PyTorch calls made from Python.  Every entry point below therefore cites the
 * reference Python call sites (file:line under the reference checkout) whose work it replaces.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes, scalars and a hipStream_t (passed as void*); no torch types.
 *   - every call is asynchronous on `stream`, never synchronises, never allocates or frees: the caller
 *     owns all buffers and passes a workspace (size from the matching *_workspace_bytes query).
 *   - return value: 0 = ok, <0 = FR_E* (invalid argument / HIP error); message via fr_last_error()
 *     (thread-local).  Data-dependent faults that can only be seen on the device (row id out of range,
 *     more than two sensitive groups in a batch) set bits in the caller-supplied `err_flag` word.
 *   - ids are int64 at the boundary (torch.LongTensor, dataset.py:1790-1791), values fp32, tables
 *     fp32 [n_rows, dim] row-major.
 */
#ifndef FAIRREC_HIP_H
#define FAIRREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__) || defined(__clang__)
#define FR_API __attribute__((visibility("default")))
#else
#define FR_API
#endif

#define FR_OK 0
#define FR_EINVAL (-1)      /* bad argument (null pointer, unsupported dim, workspace too small ...) */
#define FR_EHIP (-2)        /* a HIP runtime call failed; see fr_last_error() */
#define FR_EUNSUPPORTED (-3)

/* bits of the device-side error word */
#define FR_DEV_ERR_INDEX_RANGE 1u   /* a row id was <0 or >= n_rows (reference: IndexError in F.embedding) */
#define FR_DEV_ERR_SST_GROUPS 2u    /* >2 distinct sensitive values in one batch (reference: IndexError at focf.py:86) */

#define FR_DEV_ERR_BUCKET_OVERFLOW 4u /* a per-owner exchange bucket exceeded its fixed capacity (raise shard_capacity) */
#define FR_DEV_ERR_PIPE_WAIT 8u     /* fr_focf_step_runs_pipe: a row of the previous batch was not published within the wait bound */

/* FOCF fairness objectives -- FOCF.get_loss_fun, focf.py:50-68 */
enum fr_focf_objective {
    FR_FOCF_NONE = 0, FR_FOCF_VALUE = 1, FR_FOCF_ABSOLUTE = 2, FR_FOCF_UNDER = 3, FR_FOCF_OVER = 4,
    FR_FOCF_NONPARITY = 5
};

/*
 * An embedding table with lazily-applied dense Adam (coupled L2).
 *
 * Replaces: nn.Embedding weight + the per-parameter state of torch.optim.Adam built at
 * trainer.py:139 (exp_avg, exp_avg_sq, step).  The reference sweeps the whole table every step
 * (dense gradient, dense Adam).  Here a row's state is (p,m,v) "as of step last[row]"; the steps a
 * row missed have gradient wd*p only and are replayed in registers, in the reference's op order,
 * when the row is next read (or by the bounded-staleness sweeper / fr_table_flush).  Results are
 * those of the dense update up to fp32 rounding.
 */
typedef struct fr_table {
    float* p;          /* [n_rows, dim] parameters (the nn.Embedding weight storage) */
    float* m;          /* [n_rows, dim] exp_avg */
    float* v;          /* [n_rows, dim] exp_avg_sq */
    int32_t* last;     /* [n_rows] optimizer step the row's (p,m,v) are current for */
    int32_t* stamp;    /* [n_rows] last step at which the row was gathered for training */
    int64_t n_rows;
    int32_t dim;       /* embedding_size; 1..256 */
    int32_t step;      /* optimizer step count of this tensor so far (torch: state['step']) */
    /* Optional device-resident step counter (NULL = none).  When set, the effective step of a call is
     * *step_dev + step, read on the device: a training step captured once in a hipGraph can then be replayed every
     * iteration (the host passes the constant offset, e.g. 1 for "the step being applied", and advances the counter
     * with a kernel inside the graph).  Honoured by fr_table_gather / _gather_train / _apply_grad / _flush. */
    const int32_t* step_dev;
} fr_table;

/*
 * Adam hyper-parameters (torch.optim.Adam defaults + lr/weight_decay from trainer.py:131-139) and the
 * per-step scalars torch computes in double on the host (torch/optim/adam.py, _single_tensor_adam):
 *   scalars[4*j]   = (float)(lr / (1 - beta1^j))                      step_size_j
 *   scalars[4*j+1] = (float)(1 / sqrt(1 - beta2^j))                   1/sqrt(bias_correction2_j)
 *   scalars[4*j+2] = (float)(sqrt(k2) * scalars[4j+1] / (step_size_j * k1))   with k1 = (1-beta1)*wd,
 *   scalars[4*j+3] = (float)(eps / (step_size_j * k1))                         k2 = (1-beta2)*wd^2 (0 when wd = 0)
 * for j = 1..cap ; entry 0 unused.  Entries 2,3 let a replayed step (gradient = wd*p only) run on scaled moments.
 * Steps beyond `cap` use entry `cap` (the host guarantees the scalars have saturated there).
 */
typedef struct fr_adam {
    const float* scalars;  /* device, float[4*(cap+1)] */
    int32_t cap;
    int32_t reserved_;
    double weight_decay, beta1, beta2, eps; /* doubles: torch derives (1 - beta) in double before the fp32 cast */
} fr_adam;

FR_API int fr_version(void);
FR_API const char* fr_last_error(void);

/*
 * Sort the M row ids of a batch and cut them into segments of equal id.
 * Replaces the dedup the reference gets implicitly from dense embedding_dense_backward (autograd of
 * nn.Embedding, focf.py:138-139) and from torch.unique(return_inverse) at focf.py:77-78 / nfcf.py:79-80.
 *   perm[j]      : batch position of the j-th smallest (id, position) pair        [M]
 *   seg_start[k] : first j of segment k; seg_start[n_seg] = M                      [M+1]
 *   seg_row[k]   : the row id of segment k                                         [M]
 *   seg_of[b]    : segment index of batch position b  (= torch.unique inverse)     [M]
 *   n_seg        : number of distinct ids                                          [1]
 * M <= FR_SORT_MAX.
 */
#define FR_SORT_MAX 16384
FR_API int fr_sort_segments(const int64_t* idx, int64_t M, int64_t n_rows, int32_t* perm, int32_t* seg_start,
                     int32_t* seg_row, int32_t* seg_of, int32_t* n_seg, uint32_t* err_flag, void* stream);

/* ---- FOCF (focf.py) ---------------------------------------------------------------------------- */

FR_API size_t fr_focf_workspace_bytes(int64_t B, int32_t dim);

/*
 * Forward of one training batch: FOCF.calculate_loss, focf.py:152-169 (= forward :136-143, MSELoss
 * :158, get_item_ratings :75-91 and the unfairness terms :93-134).  Reads the tables (never writes
 * p/m/v/last), leaves the caught-up rows and dLoss/dpred in the workspace for fr_focf_backward_adam.
 *   loss_out[0] = loss, loss_out[1] = mse part, loss_out[2] = fairness part (unweighted)   (device)
 *   pred_out    = pred_scores [B] (device, may be NULL)
 */
#define FR_FOCF_PREPARED 1   /* flags: fr_focf_prepare(_many) already ran for this batch on this workspace */
/* FR_FOCF_DEFER_LOSS: loss_out is written by the fr_focf_backward_adam that follows on the same workspace and stream
 * (one extra workgroup of its launch) instead of by this call -- for step loops that read the loss after optimizer.step().
 * Takes the reduction's ticket round trip off the end of the fairness kernel, i.e. off the step's critical path.
 * The record of what is still to be reduced travels in the workspace (device memory): the library keeps no host state. */
#define FR_FOCF_DEFER_LOSS 2
/* FR_FOCF_ITEM_RUNS: a hint -- the interactions of an item sit next to each other in the batch (item-complete batches,
 * focf_dataloader.py:37-51).  The gather kernel then replays an item row once per workgroup instead of once per wave (the
 * item row carries the longest replay there: 25.6 -> 21.1 us at K = 82 items per batch).  Same results either way. */
#define FR_FOCF_ITEM_RUNS 4
FR_API int fr_focf_forward(const fr_table* U, const fr_table* I, const fr_adam* adam, const int64_t* user,
                    const int64_t* item, const float* rating, const float* sst, int64_t B, int32_t objective,
                    float fair_weight, int32_t flags, void* ws, size_t ws_bytes, float* loss_out, float* pred_out,
                    uint32_t* err_flag, void* stream);

/*
 * The index-only part of fr_focf_forward (sort + segmentation of the id columns, min/max of sst), callable one
 * batch AHEAD on another stream while the previous batch's kernels run: it depends on nothing but the ids, the
 * way a dataloader prefetches the next batch (trainer.py:181 iterates `train_data`).  Pass `sst` = NULL for
 * fair_objective none.  The caller orders it against the consumers of `ws` with events.
 */
FR_API int fr_focf_prepare(const int64_t* user, const int64_t* item, const float* sst, int64_t B, int64_t n_users,
                    int64_t n_items, int32_t dim, void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream);

/*
 * The same for up to FR_FOCF_PREPARE_MAX COMING batches in one launch (one workgroup per id column: the sort is
 * latency-bound, so n batches cost the time of one, and the per-step critical path holds no sort and no stream join).
 */
#define FR_FOCF_PREPARE_MAX 8
typedef struct fr_focf_batch {
    const int64_t* user;
    const int64_t* item;
    const float* sst;      /* NULL for fair_objective none */
    int64_t B;
    void* ws;
    size_t ws_bytes;
    const float* rating;   /* fr_focf_prepare_step only (fr_focf_prepare_many ignores it) */
} fr_focf_batch;
FR_API int fr_focf_prepare_many(const fr_focf_batch* batches, int32_t n, int64_t n_users, int64_t n_items, int32_t dim,
                         uint32_t* err_flag, void* stream);

/*
 * torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm, norm_type=2) of the reference's step loop
 * (trainer.py:194-195, config `clip_grad_norm`), for the batch of the preceding fr_focf_forward on this workspace and
 * before its fr_focf_backward_adam.  The dense embedding gradients are never materialised: their squared 2-norm is
 * the sum over the distinct rows of the batch of |sum_b dLoss/dpred[b] * other_row[b]|^2 (one wave per row, fixed
 * order), `extra_sqnorm` (device float, may be NULL) adds other parameters' share, clip_coef = min(1, max_norm /
 * (norm + 1e-6)) scales dLoss/dpred in the workspace, i.e. every gradient row the backward launch forms.
 * `norm_out` (device float[2], may be NULL) receives (total_norm, clip_coef).
 */
FR_API int fr_focf_clip_grad_norm(const fr_table* U, const fr_table* I, int64_t B, float max_norm,
                           const float* extra_sqnorm, float* norm_out, void* ws, size_t ws_bytes, void* stream);

/*
 * loss.backward() + optimizer.step() for the batch of the preceding fr_focf_forward on the same
 * workspace: autograd of focf.py:152-169 (dense embedding gradient) followed by Adam.step()
 * (trainer.py:193-196).  Duplicate rows are summed in batch order before the update, rows not in the
 * batch are left to the lazy replay; `sweep_period` S>0 additionally brings rows
 * [s*ceil(n/S), (s+1)*ceil(n/S)), s = step mod S, up to date each call (bounds the replay length).
 * Increments U->step / I->step semantics are the CALLER's: pass tables whose .step is the step being
 * applied (old step + 1) in both calls of a batch.
 */
FR_API int fr_focf_backward_adam(const fr_table* U, const fr_table* I, const fr_adam* adam, int64_t B,
                          int32_t sweep_period, void* ws, size_t ws_bytes, void* stream);

/*
 * The whole training step of one batch as ONE launch: FOCF.calculate_loss (focf.py:152-169) + loss.backward()
 * (trainer.py:193) + optimizer.step() (trainer.py:196), for step loops that read the loss after the optimizer step
 * (what FR_FOCF_DEFER_LOSS serves on the three-launch path).  Same results as fr_focf_forward + fr_focf_backward_adam;
 * objectives none / value / absolute / under / over (nonparity and clip_grad_norm need batch-wide values between the
 * forward and the update: use the three-launch path).  A wave keeps both rows of its interaction in registers from
 * the gather to the Adam write-back; rows shared by several interactions of the batch are finished by the last of
 * their waves to arrive (csrc/focf_step.hip).
 *   fr_focf_prepare_step : fr_focf_prepare_many plus what the fused launch needs from the index side: per batch
 *                          position one packed record (user row, item row, rating, sst) -- ids range-checked here, so
 *                          the step itself reads no id column -- and the extent of its user / item segment, zeroed
 *                          arrival counters, and
 *                          stamp[row] = max(stamp[row], stamps[q]) on every row of batch q in both tables -- the
 *                          sweeper waves of fr_focf_step leave rows stamped >= its `stamp` argument alone.  Callable
 *                          ahead on another stream (depends on the id columns only; the stamp raise is an atomic max).
 *                          stamps[q]: any value >= every stamp handed out before (the engine uses the step at which
 *                          the batch is expected to be applied).  It also leaves a START ORDER of the interactions,
 *                          longest estimated replay first (from the rows' `last` as of now; replay_cap = the sweep
 *                          period that bounds a replay, 0 = unbounded): the step's launch ends one wave latency after
 *                          its last wave starts, so long replays must not start last.  Order only, never a result.
 *   fr_focf_step         : the step (U->step == I->step = the step being applied).  `stamp` = the value given to
 *                          fr_focf_prepare_step for this batch.  The loss needs every wave of the launch, so it is
 *                          reduced LATER: by the next fr_focf_step (prev_ws / prev_B / prev_loss_out name the earlier
 *                          batch; one extra workgroup) or by fr_focf_step_finish.  loss_acc (device float[8], may be
 *                          NULL): [0..2] += (loss, mse, fair) of every reduced batch, [3] += 1, [4] = 1-based index of the
 *                          first reduced batch whose loss was NaN (sticky; 0 = none): a running total for epoch loops.
 *                          loss_out is where THIS batch's loss will be written by that later reduction (recorded by the
 *                          caller; not written here).
 *   fr_focf_step_finish  : the reduction for a batch no later step will reduce (end of an epoch, before reading).
 */
FR_API int fr_focf_prepare_step(const fr_focf_batch* batches, const int32_t* stamps, int32_t n, const fr_table* U,
                                const fr_table* I, int32_t replay_cap, uint32_t* err_flag, void* stream);
FR_API int fr_focf_step(const fr_table* U, const fr_table* I, const fr_adam* adam, const int64_t* user,
                        const int64_t* item, const float* rating, const float* sst, int64_t B, int32_t objective,
                        float fair_weight, int32_t sweep_period, int32_t stamp, void* ws, size_t ws_bytes,
                        float* loss_out, void* prev_ws, int64_t prev_B, float* prev_loss_out, float* loss_acc,
                        uint32_t* err_flag, void* stream);
FR_API int fr_focf_step_finish(void* ws, size_t ws_bytes, int64_t B, int32_t dim, int32_t objective, float fair_weight,
                               float* loss_out, float* loss_acc, void* stream);
/*
 * fr_focf_step for ITEM-COMPLETE batches -- what the reference's own loader feeds FOCF with (focf_dataloader.py:37-51:
 * random items, all interactions of each, until >= train_batch_size rows: tens of distinct items of ~100 rows) -- same
 * contract (batch prepared by fr_focf_prepare_step with `stamp`, loss reduced by the next step or fr_focf_step_finish), same
 * results as fr_focf_forward(FR_FOCF_ITEM_RUNS) + fr_focf_backward_adam (every sum in the same order), bit-reproducible.
 * Two launches instead of three: the chain's gather, then ONE WORKGROUP PER ITEM RUN that forms the per-group sums by wave
 * shuffles (the fairness kernel's lane order), updates the members' user rows with all its waves, sums the item's gradient
 * from LDS in ascending batch position and updates the item row; the sweep slice rides in the second launch.  Only users that
 * recur under several items of the batch go through an arrival counter (csrc/focf_runs.hip).
 */
FR_API int fr_focf_step_runs(const fr_table* U, const fr_table* I, const fr_adam* adam, const int64_t* user,
                             const int64_t* item, const float* rating, const float* sst, int64_t B, int32_t objective,
                             float fair_weight, int32_t sweep_period, int32_t stamp, void* ws, size_t ws_bytes, void* prev_ws,
                             int64_t prev_B, float* prev_loss_out, float* loss_acc, uint32_t* err_flag, void* stream);
/* The same step with the two launches of CONSECUTIVE steps side by side (csrc/focf_runs.hip): one launch = the item runs of
 * the batch the previous call gathered (`fin_ws`, `fin_B`, applied at `fin_step`; NULL: none) + the gather of this call's batch
 * (`user` == NULL: none -- the call that drains the pipeline) + this step's sweep slice + the loss reduction of the batch
 * the previous call finished (`prev_ws`).  A row that both batches hold is taken by the gather only after the finisher has
 * published it.  `own_u` / `own_i`: int32 [2 * n_rows] each, zero-initialised (and zeroed again when the tables' steps are
 * rewound).  Same results as fr_focf_step_runs; U / I carry the step of the batch being gathered (fin_step if there is none). */
FR_API int fr_focf_step_runs_pipe(const fr_table* U, const fr_table* I, const fr_adam* adam, const int64_t* user,
                                  const int64_t* item, const float* rating, const float* sst, int64_t B, int32_t objective,
                                  float fair_weight, int32_t sweep_period, int32_t stamp, void* ws, size_t ws_bytes, void* fin_ws,
                                  int64_t fin_B, int32_t fin_step, void* prev_ws, int64_t prev_B, float* prev_loss_out,
                                  float* loss_acc, int32_t* own_u, int32_t* own_i, uint32_t* err_flag, void* stream);

/*
 * The same step with the index work of the COMING batches riding in the step launches themselves, instead of a look-ahead
 * sort on a second stream (fr_focf_prepare_step): no sort, no stream fork / join in the step loop.  Every table row has
 * one 64-bit word per generation (`row_words`: fr_focf_row_words() words, zeroed once by the caller; three generations =
 * the batch being applied and the two behind it), and a batch goes through two stages of a few workgroups each:
 *   claim (two launches before its step): one thread per interaction counts itself into its rows' words (stamp << 29 |
 *         members << 14 | base; a word left by an older batch is first raised to the new stamp by an atomic maximum, so
 *         nothing is ever reset), stamps the rows for the sweeper, takes a place in its start class by replay length; K and
 *         the sensitive values by atomics; the sweeper tasks of the stamped step are classed the same way;
 *   place (one launch before): counts are final -- (members of its user, of its item) per interaction, a slice of the
 *         member list per shared row, the records written in start order.
 * In the step, the members of a shared row enter their batch position in the row's list and leave the word one by one;
 * whoever brings the count to zero reads the list, orders it (ascending batch position, in registers) and finishes the
 * row exactly as the sorted path does: same sums in the same order, bit-identical parameters.  The reported fairness
 * value sums the items' terms in batch order of their first members instead of item-id order (last-bit differences in
 * the loss value only).
 *   fr_focf_stage        : stages on their own launch (the first batches of a loop; either batch may be NULL).  U->step /
 *                          I->step = the optimizer step the batch to place -- without one, the batch to claim -- will be
 *                          applied at (with both, the claimed batch is applied one step later): the start order of that
 *                          step's sweeper tasks is built for the step's slice of the tables, whatever the stamps are.
 *   fr_focf_step_staged  : fr_focf_step for a batch that went through both stages with stamp `stamp`, carrying the
 *                          claim of one coming batch and the place of another (either may be NULL).  Stamps must be
 *                          handed out in strictly increasing order.  `gen` (0..2) names the generation of row words a
 *                          batch uses from its claim to its step: the (up to) three batches in flight at any time --
 *                          claimed, placed, being applied -- must hold three different ones.
 *   fr_focf_step_finish_staged : fr_focf_step_finish for such a batch (also returns its workspace's counters to zero,
 *                          as the next fr_focf_step_staged would have).
 * A workspace must be zero-filled before its first claim.
 */
FR_API size_t fr_focf_row_words(int64_t n_users, int64_t n_items);
FR_API int fr_focf_stage(const fr_table* U, const fr_table* I, const fr_focf_batch* claim, int32_t claim_stamp,
                         int32_t claim_gen, const fr_focf_batch* place, int32_t place_stamp, int32_t place_gen,
                         int32_t sweep_period, uint64_t* row_words, uint32_t* err_flag, void* stream);
FR_API int fr_focf_step_staged(const fr_table* U, const fr_table* I, const fr_adam* adam, const float* sst, int64_t B,
                               int32_t objective, float fair_weight, int32_t sweep_period, int32_t stamp, int32_t gen,
                               void* ws, size_t ws_bytes, void* prev_ws, int64_t prev_B, float* prev_loss_out,
                               float* loss_acc, uint64_t* row_words, const fr_focf_batch* claim, int32_t claim_stamp,
                               int32_t claim_gen, const fr_focf_batch* place, int32_t place_stamp, int32_t place_gen,
                               uint32_t* err_flag, void* stream);
FR_API int fr_focf_step_finish_staged(void* ws, size_t ws_bytes, int64_t B, int32_t dim, int32_t objective,
                                      float fair_weight, float* loss_out, float* loss_acc, void* stream);
/*
 * The step loop itself (trainer.py:181-196: `for batch in train_data: zero_grad; calculate_loss; backward; step`) for a run
 * of `n` batches in ONE call: n launches of fr_focf_step_staged issued by the library, so the host pays one foreign call
 * per run instead of an interpreter round trip per step.  Batch k is applied at step U->step + k (U->step = the step of
 * the FIRST batch, as everywhere), with stamp first_stamp + k and generation (first_gen + k) % 3; its launch carries the
 * place stage of batch k + 1 and the claim stage of batch k + 2 and reduces the loss of batch k - 1 into
 * loss_ring[4 * ((first_slot + k - 1) % loss_slots)] and loss_acc (batch 0's launch reduces `prev_*`, an earlier staged
 * batch still unreduced, or nothing when prev_ws is NULL).  The first two batches' stages take two launches of their own.
 * Any four consecutive batches (and prev_ws with the first three) need four different workspaces, each zero-filled before
 * its first use.  Nothing may be in flight through the stages when the call starts; the LAST batch's loss stays unreduced
 * (hand it to the next call as prev_*, or to fr_focf_step_finish_staged).  Same launches, same bits as n calls of
 * fr_focf_step_staged.
 */
/*
 * The same loop for ITEM-COMPLETE batches (the only shape the reference's FOCFDataLoader yields, focf_dataloader.py:37-51):
 * n launches of fr_focf_step_runs_pipe issued by the library, the batches' sorted prepare (fr_focf_prepare_step,
 * FR_FOCF_PREPARE_MAX batches per launch) running one group ahead on the library's side stream and joined once per group.
 * Batch k is gathered at step U->step + k with stamp first_stamp + k; its item runs ride in the launch of batch k + 1, and the
 * launch of batch k + 2 reduces its loss into loss_ring[4 * ((first_slot + k) % loss_slots)] and loss_acc.  `fin_*`: a batch
 * the previous call gathered and nobody finished yet (fin_step = U->step - 1, fin_loss_out = its loss slot; NULL: none);
 * `prev_*`: a finished batch whose loss is still unreduced (NULL: none).  When the call returns, batch n - 1 is in the `fin`
 * position and batch n - 2 (or the incoming `fin`) in the `prev` position: hand them to the next call, or drain them with
 * fr_focf_step_runs_pipe(user = NULL) + fr_focf_step_finish.  Batches closer than 2 * FR_FOCF_PREPARE_MAX + 2 in the run need
 * different workspaces, none of them a pending one.  Same launches, same bits as n calls of fr_focf_step_runs_pipe.
 */
FR_API int fr_focf_runs_many(const fr_table* U, const fr_table* I, const fr_adam* adam, const fr_focf_batch* batches,
                             int32_t n, int32_t objective, float fair_weight, int32_t sweep_period, int32_t first_stamp,
                             void* fin_ws, int64_t fin_B, int32_t fin_step, float* fin_loss_out, void* prev_ws, int64_t prev_B,
                             float* prev_loss_out, float* loss_ring, int32_t loss_slots, int32_t first_slot, float* loss_acc,
                             int32_t* own_u, int32_t* own_i, uint32_t* err_flag, void* stream);
FR_API int fr_focf_steps_many(const fr_table* U, const fr_table* I, const fr_adam* adam, const fr_focf_batch* batches,
                              int32_t n, int32_t objective, float fair_weight, int32_t sweep_period, int32_t first_stamp,
                              int32_t first_gen, void* prev_ws, int64_t prev_B, float* prev_loss_out, float* loss_ring,
                              int32_t loss_slots, int32_t first_slot, float* loss_acc, uint64_t* row_words,
                              uint32_t* err_flag, void* stream);

/* FOCF.predict, focf.py:145-150: clamp(pred, 0, max_rating) / max_rating on up-to-date rows (read only). */
FR_API int fr_focf_predict(const fr_table* U, const fr_table* I, const fr_adam* adam, const int64_t* user,
                    const int64_t* item, int64_t B, float max_rating, float* out, uint32_t* err_flag, void* stream);

/* ---- lazy Adam table maintenance ---------------------------------------------------------------- */

/* Bring every row of the table up to step `t->step` (needed before anything reads the whole table:
 * state_dict / checkpoint trainer.py:221-240, full_sort_predict focf.py:171-178, evaluation). */
FR_API int fr_table_flush(const fr_table* t, const fr_adam* adam, void* stream);

/* Gather rows as of step t->step (read only; replay in registers): out[j,:] = p[idx[j],:]. */
FR_API int fr_table_gather(const fr_table* t, const fr_adam* adam, const int64_t* idx, int64_t M, float* out,
                    uint32_t* err_flag, void* stream);

/*
 * Generic training pair for models whose score is not a plain dot product (NFCF nfcf.py:69-74, PFCN
 * pfcn_biasedmf.py:144-166) and for the owner side of the row-sharded multi-GPU path:
 *   fr_table_gather_train : sort idx into segments + gather the rows as of step t->step-1 (lazy replay) into
 *                           rows_out[M,dim]; caught-up moments stay in the workspace.  Replaces nn.Embedding.forward.
 *   ... the caller runs its forward/backward on rows_out and produces grad_rows[M,dim] = dLoss/d rows_out ...
 *   fr_table_apply_grad   : per distinct row, sum grad_rows over duplicates in ascending position (what
 *                           embedding_dense_backward does), apply Adam step t->step, write the row back; plus the
 *                           sweeper slice.  Replaces loss.backward() on the embedding + optimizer.step().
 * `t->step` is the step being applied in both calls; M <= FR_SORT_MAX.
 *
 * Slot layout (chunk, stride) of idx / rows_out / rows / grad_rows: logical position j sits at
 * (j / chunk) * stride + j % chunk  rows (ids) from the pointer; chunk = 0 is the dense [M, ...] layout.  With
 * chunk = cap, stride = T*cap and pointers advanced by t*cap rows, table t of T works in place on a shared
 * [G, T, cap, ...] all-to-all buffer (multi-GPU path), so the exchange needs no packing copies.
 */
FR_API size_t fr_table_train_workspace_bytes(int64_t M, int32_t dim);
/* Orders `stream` behind the index work (the sort fr_table_gather_train launches on the library's side stream) still pending
 * on workspace `ws`.  fr_table_apply_grad, fr_focf_shard_fair and fr_nfcf_loss do this themselves; a caller that reads the
 * segments of a workspace on its own (copying them for fr_table_gather_train_prepared) calls this first. */
FR_API int fr_table_join(const void* ws, void* stream);

/*
 * The stream fr_table_gather_train runs its id sort on, beside the caller's stream (NULL when it sorts in line: profiler on,
 * FAIRREC_NO_OVERLAP=1).  The sort reads the id list and writes the workspace until the consumer of the segments
 * (fr_table_apply_grad, fr_table_join) has ordered the caller's stream behind it; a caller whose allocator recycles memory in
 * stream order must tell it that both buffers are in use on this stream as well (torch: Tensor.record_stream), or a buffer
 * freed after a forward pass that no backward pass followed can be handed out again while the sort still writes to it.
 */
FR_API void* fr_side_stream_handle(void);
/* fr_table_gather_train(t, idx) and fr_table_gather(ro, ro_idx) -- a frozen table next to a training one: NFCF's finetune
 * stage, nfcf.py:66-71 -- in ONE launch that also carries the sort of `idx` (same results as the two calls; falls back to
 * them when the two tables' rows have different fragment counts). */
FR_API int fr_table_lookup_pair(const fr_table* t, const fr_adam* adam, const int64_t* idx, int64_t M, float* rows_out,
                                void* ws, size_t ws_bytes, const fr_table* ro, const fr_adam* ro_adam, const int64_t* ro_idx,
                                int64_t ro_M, float* ro_out, uint32_t* err_flag, void* stream);
/* The first fr_table_segments_bytes(M) bytes of a workspace hold the sorted segments of the id list (independent of the
 * table's width).  fr_table_gather_train_prepared skips the sort: the caller has put the segments of `idx` there, e.g. by
 * copying them from the workspace of another table with the same number of rows looked up with the same ids in this step
 * (PFCN_BiasedMF's bias tables next to the embedding tables, pfcn_biasedmf.py:186-190). */
FR_API size_t fr_table_segments_bytes(int64_t M);
FR_API int fr_table_gather_train_prepared(const fr_table* t, const fr_adam* adam, const int64_t* idx, int64_t M,
                                          int32_t chunk, int32_t stride, float* rows_out, void* ws, size_t ws_bytes,
                                          uint32_t* err_flag, void* stream);
FR_API int fr_table_gather_train(const fr_table* t, const fr_adam* adam, const int64_t* idx, int64_t M, int32_t chunk,
                                 int32_t stride, float* rows_out, void* ws, size_t ws_bytes, uint32_t* err_flag,
                                 void* stream);
FR_API int fr_table_apply_grad(const fr_table* t, const fr_adam* adam, int64_t M, int32_t chunk, int32_t stride,
                               const float* rows, const float* grad_rows, int32_t sweep_period, void* ws,
                               size_t ws_bytes, void* stream);
/* The same for TWO tables of equal dim and M in one launch each (one sort workgroup / one grid slice per table):
 * the user and item table of a step.  ws_a != ws_b, each of fr_table_train_workspace_bytes(M, dim). */
#define FR_TABLE_PREPARED 1   /* flags: fr_table_sort2 already ran for these id lists on these workspaces */
FR_API int fr_table_gather_train2(const fr_table* ta, const fr_table* tb, const fr_adam* adam, const int64_t* idx_a,
                                  const int64_t* idx_b, int64_t M, int32_t chunk, int32_t stride, float* rows_a,
                                  float* rows_b, int32_t flags, void* ws_a, void* ws_b, size_t ws_bytes,
                                  uint32_t* err_flag, void* stream);
/* The index-only part of fr_table_gather_train2 (sort + segmentation of both id lists into the workspaces), callable
 * one batch AHEAD on another stream: it depends on nothing but the ids.  The caller orders it against the users of the
 * workspaces with events. */
FR_API int fr_table_sort2(const int64_t* idx_a, const int64_t* idx_b, int64_t n_rows_a, int64_t n_rows_b, int64_t M,
                          int32_t chunk, int32_t stride, int32_t dim, void* ws_a, void* ws_b, size_t ws_bytes,
                          uint32_t* err_flag, void* stream);
FR_API int fr_table_apply_grad2(const fr_table* ta, const fr_table* tb, const fr_adam* adam, int64_t M, int32_t chunk,
                                int32_t stride, const float* rows_a, const float* grad_a, const float* rows_b,
                                const float* grad_b, int32_t sweep_a, int32_t sweep_b, void* ws_a, void* ws_b,
                                size_t ws_bytes, void* stream);
/* fr_table_apply_grad on two tables of one width in ONE launch, each with its own id count (PFCN: the user rows of a batch and
 * the item rows of its positive + negative ids, pfcn_pmf.py:169-176; their two bias columns likewise): the shorter table's
 * update and sweep slice run beside the longer one's.  Same results as the two calls; both tables step under `adam`. */
FR_API int fr_table_apply_grad_two(const fr_table* ta, const fr_table* tb, const fr_adam* adam, int64_t Ma, int64_t Mb,
                                   const float* rows_a, const float* grad_a, const float* rows_b, const float* grad_b,
                                   int32_t sweep_a, int32_t sweep_b, void* ws_a, size_t ws_a_bytes, void* ws_b,
                                   size_t ws_b_bytes, void* stream);

/* ---- row-sharded tables (multi-GPU) -----------------------------------------------------
 * owner(row) = row mod G, local row = row div G.  All exchange buffers have the fixed shape [G, cap] slots so
 * that a step needs no host sync; -1 ids are padding (ignored by fr_sort_segments / fr_table_gather_train).
 * fr_bucket_by_owner: stable partition of idx by owner; slot (o, k) = o*stride + offset + k is where the k-th id owned
 * by rank o sits in send_ids and where its answer will sit in every reply buffer (slot_of[pos], -1 on overflow).
 * stride = cap, offset = 0: a [G, cap] buffer of its own; stride = T*cap, offset = t*cap: list t of a shared
 * [G, T, cap] buffer (one all-to-all for T id lists).
 * aux (may be NULL): (min, max) of that float column over the M positions is written, as two floats, into the int64
 * slot o*stride + aux_slot of every owner's chunk (a slot outside the id lists), so batch-wide extrema ride along
 * with the id exchange (FOCF: the two sensitive-attribute values present in the global batch, focf.py:77-79). */
FR_API int fr_bucket_by_owner(const int64_t* idx, int64_t M, int32_t G, int32_t cap, int32_t stride, int32_t offset,
                              int64_t* send_ids, int32_t* slot_of, int32_t* counts, const float* aux, int32_t aux_slot,
                              uint32_t* err_flag, void* stream);
/* Two id lists of the same length in ONE launch (one workgroup each): list a at offset_a, list b at offset_b of the
 * same [G, stride] buffer; counts = [2, G]; aux / aux_slot as above (computed by list b's workgroup). */
/* Item-owner-computes schedule of the row-sharded FOCF step (fairrec/sharded.py, ShardedFocfEngineV2): every interaction is
 * sent to the rank that owns its item row.  A record chunk is [4 * cap + 1] int64: item local rows (fr_bucket_by_owner with
 * stride 4 * cap + 1, offset 0; -1 = empty) | user ids | rating bits | sst bits | the sender's (min, max) of sst (the bucket
 * kernel's aux pair).  fr_shard_pack_records fills planes 1-3 at the slots the bucket kernel assigned; the receiver's
 * fr_shard_unpack_records turns its G chunks into per-slot arrays (item row, user id, own slot index or -1, rating or 0,
 * sst) and the G (min, max) pairs.  fr_shard_post_fair / fr_shard_loss_finish: see csrc/shard.hip. */
FR_API int fr_shard_pack_records(const int32_t* slot, const int64_t* user, const float* rating, const float* sst, int64_t B,
                                 int32_t cap, int64_t* send, void* stream);
FR_API int fr_shard_unpack_records(const int64_t* recv, int32_t G, int32_t cap, int64_t* iid, int64_t* uid, int32_t* islot,
                                   float* rating, float* sst, int64_t* mm, void* stream);
/* out[0] = number of distinct ids >= 0 of the list; bitmap = (n_rows + 31) / 32 words, all zero before and after; count = one
 * zero int32, zero again afterwards. */
FR_API int fr_shard_count_distinct(const int64_t* ids, int64_t n, int64_t n_rows, uint32_t* bitmap, int32_t* count, float* out,
                                   void* stream);
FR_API int fr_shard_post_fair(float* reply, const float* k_all, int32_t G, int32_t cap, float* sums, void* stream);
FR_API int fr_shard_loss_finish(const float* sums, const float* k_all, int32_t G, int64_t n_global, float fair_weight,
                                int32_t fair, float* loss, void* stream);
/* fr_bucket_by_owner for a list with EMPTY positions (id -1, the padding of a received exchange buffer): they are
 * given to no owner (slot -1) and raise no error. */
FR_API int fr_bucket_by_owner_sparse(const int64_t* idx, int64_t M, int32_t G, int32_t cap, int32_t stride, int32_t offset,
                                     int64_t* send_ids, int32_t* slot_of, int32_t* counts, uint32_t* err_flag, void* stream);
FR_API int fr_bucket_pair_by_owner(const int64_t* idx_a, const int64_t* idx_b, int64_t M, int32_t G, int32_t cap,
                                   int32_t stride, int32_t offset_a, int32_t offset_b, int64_t* send_ids,
                                   int32_t* slot_a, int32_t* slot_b, int32_t* counts, const float* aux,
                                   int32_t aux_slot, uint32_t* err_flag, void* stream);
/* out[j,:] = src[slot_of[j],:]  (reply buffer in slot order -> batch order) */
FR_API int fr_unbucket_rows(const float* src, const int32_t* slot_of, int64_t M, int32_t dim, float* out, void* stream);
/* dst[slot_of[j],:] = scale[j] * src[j,:]  (batch order -> slot order; scale may be NULL; dst pre-zeroed) */
FR_API int fr_bucket_rows(const float* src, const float* scale, const int32_t* slot_of, int64_t M, int32_t dim,
                          float* dst, void* stream);

/*
 * Row-sharded FOCF step (one rank's share of focf.py:152-169 on the GLOBAL batch = concatenation of all ranks'
 * batches).  Requester side works on the rows returned by the owners ([G*cap, dim] slot order):
 *   fr_focf_shard_score : pred, MSE part of dLoss/dpred (2(pred-r)/n_global), sum of squared errors (sq_err_sum; NULL:
 *                         only the (B+3)/4 per-workgroup partials in `scratch`, for fr_focf_shard_fair), and the
 *                         (pred, rating, sst) records for the item owners: rec = [G, 3, cap], record of item slot
 *                         s = o*slot_stride + slot_offset + k (as produced by fr_bucket_by_owner) at rec[o][0..2][k]
 *   fr_focf_shard_fair  : on the item OWNER, over the segments fr_table_gather_train left in the item table's
 *                         workspace: per-item group statistics of the received records ([G(src), 3, cap]) ->
 *                         fairness part of dLoss/dpred per slot, NOT yet divided by K, into the reply buffer
 *                         [G, cap + FR_SHARD_TAIL]; the tail of every chunk = (this owner's distinct items, its sum
 *                         of smooth-L1 terms, this rank's sum of squared errors = sum of sq_part[0..n_sq_part), the
 *                         partials fr_focf_shard_score left in its scratch), written by the last workgroup to
 *                         arrive, so the scalars of the loss ride along with the reply exchange (no all-reduce).
 *                         scratch: >= n_slots/64 + 32 floats, scratch[0] zero on first use (arrival ticket).
 *                         minmax: mm_count (min, max) pairs, mm_stride floats apart (one per source rank, as
 *                         delivered by fr_bucket_by_owner's aux slot), folded in the kernel
 *   fr_focf_shard_grads : gradient rows c*ie / c*ue written at the user / item slots for the owners.  coef_slots =
 *                         the RECEIVED reply buffer (NULL for fair_objective none): K, fairness sum and squared-error
 *                         sum are folded over its G tails in rank order (identical on every rank) and
 *                         loss_out[0..2] = loss, mse, fair is written (may be NULL)
 *   fr_focf_shard_nonparity_sums / _coef : fair_objective nonparity (focf.py:127-134) needs no item statistics, only
 *                         the two group means of pred over the GLOBAL batch: _sums leaves this rank's (sum err^2,
 *                         sum pred|g0, n0, sum pred|g1, n1) in out5 (one 5-float all-reduce makes them global); _coef
 *                         adds the fairness part of dLoss/dpred to coef[b] in place and writes loss_out[0..2]
 * rows_u / rows_i / grad_*_slots are addressed by slot, so both may point at one shared [G, 2, cap, dim] buffer.
 */
#define FR_SHARD_TAIL 3
FR_API int fr_focf_shard_score(const float* rows_u, const float* rows_i, const int32_t* slot_u, const int32_t* slot_i,
                               const float* rating, const float* sst, int64_t B, int32_t dim, int64_t n_global,
                               float* pred, float* coef, float* rec, int32_t cap, int32_t slot_stride,
                               int32_t slot_offset, float* sq_err_sum, float* scratch, void* stream);
FR_API int fr_focf_shard_fair(void* item_ws, size_t ws_bytes, int64_t n_slots, int32_t dim, const float* rec,
                              int32_t cap, const float* minmax, int32_t mm_count, int32_t mm_stride, int32_t objective,
                              float fair_weight, float* reply, const float* sq_part, int32_t n_sq_part, float* scratch,
                              uint32_t* err_flag, void* stream);
FR_API int fr_focf_shard_nonparity_sums(const float* pred, const float* sst, int64_t B, const float* minmax,
                                        int32_t mm_count, int32_t mm_stride, const float* sq_part, int32_t n_sq_part,
                                        float* out5, void* stream);
FR_API int fr_focf_shard_nonparity_coef(float* coef, const float* sst, int64_t B, const float* minmax, int32_t mm_count,
                                        int32_t mm_stride, const float* global5, int64_t n_global, float fair_weight,
                                        float* loss_out, uint32_t* err_flag, void* stream);
FR_API int fr_focf_shard_grads(const float* rows_u, const float* rows_i, const int32_t* slot_u, const int32_t* slot_i,
                               const float* coef, const float* coef_slots, int32_t G, int64_t n_global,
                               float fair_weight, float* loss_out, int32_t cap, int32_t slot_stride,
                               int32_t slot_offset, int64_t B, int32_t dim, float* grad_u_slots, float* grad_i_slots,
                               void* stream);

/* ---- dense layers (fp32 MFMA) ---------------------------------------------------------------------------
 * One layer of recbole/model/layers.py MLPLayers (:62-72): Dropout -> Linear -> activation, forward and backward.
 *   X = [x0 | x1] : [M, k0 + k1] (x1 may be NULL; the split serves cat(U[u], I[i]) of nfcf.py:72 without a copy)
 *   mask          : [M, k0 + k1] bytes, 0 = dropped (the Bernoulli draw stays with the host RNG),
 *                   NULL = no dropout; kept elements are multiplied by `scale` = 1/(1-p)
 *   W [N, K], bias [N] : nn.Linear layout;  act: 0 none, 1 relu, 2 leakyrelu(0.01), 3 sigmoid, 4 tanh
 *   Y [M, N]      : POST-activation output; the backward kernels rebuild act'(.) from it
 * fr_linear_bwd_weight reduces over the batch in a fixed order (per-split slabs summed in split order). */
FR_API int fr_linear_fwd(const float* x0, int32_t k0, const float* x1, int32_t k1, const uint8_t* mask, float scale,
                         const float* W, const float* bias, int64_t M, int32_t N, int32_t act, float* Y, void* stream);
FR_API int fr_linear_bwd_input(const float* dY, const float* Y, int32_t act, const float* W, const uint8_t* mask,
                               float scale, int64_t M, int32_t N, float* dx0, int32_t k0, float* dx1, int32_t k1,
                               void* stream);
FR_API size_t fr_linear_bwd_weight_workspace_bytes(int64_t M, int32_t N, int32_t K);
FR_API int fr_linear_bwd_weight(const float* dY, const float* Y, int32_t act, const float* x0, int32_t k0,
                                const float* x1, int32_t k1, const uint8_t* mask, float scale, int64_t M, int32_t N,
                                float* dW, float* db, void* ws, size_t ws_bytes, void* stream);
/* out[n] = dY o act'(Y), elementwise (n % 4 == 0, 16-byte aligned): the shared pre-pass of a layer's two backward
 * products (the autograd of the activation in layers.py:68-72), after which both are called with act = 0 and take
 * their fast form (LDS-DMA kernels: no dropout mask, act = 0, widths multiples of 32). */
/* Both backward products of a layer with ONE output (N == 1, one input block, no mask) in one pass over X: dW [K], db [1]
 * (may be NULL) and dX [M, K] (may be NULL) -- what fr_linear_bwd_weight + fr_linear_bwd_input give for that shape.
 * FR_EUNSUPPORTED unless K % 64 == 0, K <= 512 and X, W, dX are 16-byte aligned (take the two general calls then).
 * ws: fr_linear_bwd_weight_workspace_bytes(M, 1, K).
 * relu_scale > 0: X is the previous layer's ReLU output dropped in place (keep = 0 or relu_scale) and dX comes out as the
 * gradient at that layer's pre-activation (fr_act_bwd_dropped folded in); 0: plain dX. */
FR_API int fr_linear_n1_bwd(const float* dY, const float* Y, int32_t act, const float* X, int32_t K, const float* W, int64_t M,
                            float relu_scale, float* dX, float* dW, float* db, void* ws, size_t ws_bytes, void* stream);
/* dA = (dY W) o scale o [Xd > 0]: fr_linear_bwd_input followed by fr_act_bwd_dropped, in one launch, for a layer whose
 * input Xd [M, K] is the previous layer's ReLU output dropped in place.  Fast form only (FR_EUNSUPPORTED unless N % 32 == 0,
 * K % 32 == 0, 16-byte aligned operands). */
FR_API int fr_linear_bwd_input_relu(const float* dY, const float* W, int64_t M, int32_t N, int32_t K, const float* Xd,
                                    float scale, float* dA, void* stream);
/* dA = (dY W) o act'(Yin): fr_linear_bwd_input followed by the fr_act_bwd of the layer BELOW, in one launch: Yin [M, K] is this
 * layer's input = that layer's activation output (activation code `act`, 1..4), dA the gradient at its pre-activation --
 * the autograd of `activation(Linear(...))` between two Linear layers (recbole/model/layers.py:62-72) without a pass of its
 * own.  Same bits as the two calls.  Fast form only (FR_EUNSUPPORTED unless N % 32 == 0, K % 32 == 0, aligned operands). */
FR_API int fr_linear_bwd_input_act(const float* dY, const float* W, int64_t M, int32_t N, int32_t K, const float* Yin,
                                   int32_t act, float* dA, void* stream);
FR_API int fr_act_bwd(const float* dY, const float* Y, int32_t act, int64_t n, float* out, void* stream);
/* The same pre-pass through a ReLU whose OUTPUT went through dropout in place: Yd = relu(z) o keep (keep = 0 or scale),
 * dY = gradient with respect to Yd; out = dY o scale o [Yd > 0] = dY o keep o relu'(z).  Replaces the reference's
 * Dropout.backward + ReLU.backward pair between two Linear layers (recbole/model/layers.py:62-72). */
FR_API int fr_act_bwd_dropped(const float* dY, const float* Yd, float scale, int64_t n, float* out, void* stream);

/* ---- dropout without a stored mask (csrc/dropout.hip) ----------------------------------------------------------------
 * out[i] = x[i] * keep_i for i < n (in place allowed), keep_i = 0 with probability p, else 1/(1-p); keep_i is a pure
 * function (Philox4x32-10) of (seed, *counter, offset + i).  offset % 4 == 0; x, out 16-byte aligned.
 *   counter     device int64: the call counter to use.  A forward pass hands its state word; the backward pass the
 *               value that forward recorded, so that it regenerates the same pattern.
 *   used_out    optional device int64[1]: receives the counter value this launch used.
 *   tick_state  optional device int64[2] {counter, ticket}, ticket == 0 between launches: this launch advances the
 *               counter by one AFTER all of its workgroups have read it (pass it with the LAST launch of a forward pass,
 *               whose launches then all see the same value; `counter` must then be tick_state).
 * nn.Dropout of the reference's MLPLayers (recbole/model/layers.py:62-63); see DESIGN.md §8a. */
FR_API int fr_dropout_apply(const float* x, int64_t n, float p, uint64_t seed, uint64_t offset, const int64_t* counter,
                            int64_t* used_out, int64_t* tick_state, float* out, void* stream);
/* The same for two tensors in ONE launch (the two blocks [x0 | x1] of a first layer's input); x1 may be NULL. */
FR_API int fr_dropout_apply2(const float* x0, int64_t n0, uint64_t offset0, float* out0, const float* x1, int64_t n1,
                             uint64_t offset1, float* out1, float p, uint64_t seed, const int64_t* counter,
                             int64_t* used_out, int64_t* tick_state, void* stream);

/* The running loss total of a step loop that reads its losses once per epoch (trainer.py:184-193 reads `.item()` and checks
 * `isnan` every step: two host syncs per step).  acc = device float[8]: acc[0..n-1] += part[0..n-1] (n <= 3 loss values of this
 * step), acc[3] += 1 (steps so far), and the FIRST step whose loss was not a number is remembered in acc[4] (1-based, 0 =
 * none; sticky): the ValueError('Training loss is nan') raised at the epoch's end names the step the reference would have
 * stopped at.  The one-launch FOCF steps keep the same record in their `loss_acc` (csrc/focf_loss.hpp). */
FR_API int fr_loss_accumulate(const float* part, int32_t n, float* acc, void* stream);

/* n <= FR_COPY_MAX device-to-device copies of bytes[j] bytes in ONE launch (jobs must not overlap each other). */
#define FR_COPY_MAX 16
FR_API int fr_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream);

/* BatchNorm1d on batch statistics between Linear and activation (MLPLayers(bn=True), layers.py:66-67; the PFCN filters
 * and discriminators, which the reference never puts in eval mode).  Z [M,N] -> Y = act(gamma * xhat + beta);
 * xhat [M,N] and invstd [N] are kept for fr_bn_bwd; running_mean/var (may be NULL) follow torch (momentum, unbiased).
 * ws (fr_bn_workspace_bytes) holds the per-row-chunk partial statistics of the two-launch reduction. */
FR_API size_t fr_bn_workspace_bytes(int64_t M, int32_t N);
FR_API int fr_bn_fwd(const float* Z, const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, int64_t M, int32_t N, int32_t act, float* Y, float* xhat,
                     float* invstd, void* ws, size_t ws_bytes, void* stream);
/* fr_bn_fwd that also writes Yd = dropout(Y), the next layer's input, from its last launch: what
 * fr_dropout_apply(Y, M*N, p, seed, offset, counter, used_out, tick_state, Yd) would give (same pattern, same counter
 * protocol), without the extra pass.  N % 4 == 0; all tensors 16-byte aligned. */
FR_API int fr_bn_fwd_drop(const float* Z, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, int64_t M, int32_t N, int32_t act, float* Y, float* xhat,
                          float* invstd, void* ws, size_t ws_bytes, float* Yd, float p, uint64_t seed, uint64_t offset,
                          const int64_t* counter, int64_t* used_out, int64_t* tick_state, void* stream);
/* The product of a layer that has BatchNorm behind it with the BatchNorm's per-chunk statistics formed in the product's epilogue
 * (into `bn_ws`, fr_bn_workspace_bytes(M, N)), and fr_bn_fwd / fr_bn_fwd_drop as one entry that can skip its statistics launch
 * (`have_stats`): layers.py:64-67 in two BatchNorm launches instead of three.  fr_linear_fwd_bnstats: fast form only
 * (FR_EUNSUPPORTED otherwise: fr_linear_fwd + fr_bn_fwd then).  `num_batches_tracked` (may be NULL): the layer's int64 batch
 * counter, moved by `passes` in the same launch that updates the running statistics (nn.BatchNorm1d's bookkeeping). */
FR_API int fr_linear_fwd_bnstats(const float* x0, int32_t k0, const float* x1, int32_t k1, const float* W, const float* bias,
                                 int64_t M, int32_t N, float* Z, void* bn_ws, size_t bn_ws_bytes, void* stream);
FR_API int fr_bn_fwd_ex(const float* Z, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                        float* running_var, int64_t M, int32_t N, int32_t act, float* Y, float* xhat, float* invstd, void* ws,
                        size_t ws_bytes, int32_t have_stats, float* Yd, float p, uint64_t seed, uint64_t offset,
                        const int64_t* counter, int64_t* used_out, int64_t* tick_state, int64_t* num_batches_tracked,
                        int32_t passes, void* stream);
FR_API int fr_bn_bwd(const float* dY, const float* Y, int32_t act, const float* xhat, const float* invstd,
                     const float* gamma, int64_t M, int32_t N, float* dZ, float* dgamma, float* dbeta, void* ws,
                     size_t ws_bytes, void* stream);
/* A BatchNorm layer's backward fed by the product above it (layers.py:62-70 in reverse: Linear_l's input gradient is the
 * gradient at the dropped output of BatchNorm layer l - 1): fr_linear_bwd_input_bnstats computes dX = (dY W) o keep -- the
 * dropout's pattern regenerated from (p, seed, element offset, the call counter value in `used`), p = 0: none -- stores it, and
 * leaves that layer's backward statistics per 32-row tile in bn_ws (fr_bn_workspace_bytes(M, K)); fr_bn_bwd_ex(have_stats = 1)
 * on dY = dX and the same bn_ws is then the apply launch alone: one launch where there were three (product, dropout,
 * statistics).  The tile sums are added in another order than fr_bn_bwd's statistics launch adds them (rounding-level
 * differences).  Fast form only: FR_EUNSUPPORTED unless N % 32 == 0, K % 32 == 0, 16-byte aligned operands, M <= 32768. */
FR_API int fr_linear_bwd_input_bnstats(const float* dY, const float* W, int64_t M, int32_t N, int32_t K, float* dX,
                                       const float* Yb, const float* xhat_b, int32_t act_b, void* bn_ws, size_t bn_ws_bytes,
                                       float p, uint64_t seed, uint64_t offset, const int64_t* used, void* stream);
FR_API int fr_bn_bwd_ex(const float* dY, const float* Y, int32_t act, const float* xhat, const float* invstd,
                        const float* gamma, int64_t M, int32_t N, float* dZ, float* dgamma, float* dbeta, void* ws,
                        size_t ws_bytes, int32_t have_stats, void* stream);

/* ---- PFCN scoring / losses (pfcn_pmf.py, pfcn_biasedmf.py, loss.py) -------------------------------------------------
 * fr_rowdot_*   : torch.mul(a, b).sum(-1) on gathered rows and its backward (da = g*b, db = g*a; either may be NULL)
 * fr_bpr        : BPRLoss, loss.py:45-47
 * fr_bpr_outer  : the same loss under PFCN_BiasedMF's [B] + [B,1] -> [B,B] broadcast (pfcn_biasedmf.py:192-195,
 *                 SURVEY.md App. B-1): mean over all (i,j) of -log(1e-10 + sigmoid(a_j + c_i)); nothing of size B^2 is stored
 * fr_softmax_ce : nn.CrossEntropyLoss (multi-class discriminators, pfcn_biasedmf.py:216) */
FR_API int fr_rowdot_fwd(const float* a, const float* b, int64_t B, int32_t dim, float* out, void* stream);
FR_API int fr_rowdot_bwd(const float* g, const float* a, const float* b, int64_t B, int32_t dim, float* da, float* db,
                         void* stream);
FR_API size_t fr_bpr_workspace_bytes(int64_t B, int32_t outer);
FR_API int fr_bpr(const float* pos, const float* neg, int64_t B, float* loss, float* dpos, float* dneg, void* ws,
                  size_t ws_bytes, void* stream);
FR_API int fr_bpr_outer(const float* a, const float* c, int64_t B, float* loss, float* da, float* dc, void* ws,
                        size_t ws_bytes, void* stream);
/* The same on the four columns themselves: a = pos - neg and c = pos_bias - neg_bias are formed in the kernel and the
 * gradients of all four come out (d_neg = -d_pos, d_neg_bias = -d_pos_bias): no elementwise launches around the loss. */
/* fr_bpr_outer_rect: the same term matrix for Nc rows (c) and Na columns (a) of possibly different batches, scaled by
 * `inv`: loss[0] = inv * sum_ij -log(1e-10 + sigmoid(a_j + c_i)); da [Na] / dc [Nc] (either may be NULL) = the column / row
 * sums of its derivative.  A row-sharded step evaluates its part of the GLOBAL batch's [G B, G B] matrix with it: own
 * columns against all rows (da), own rows against all columns (dc), inv = 1 / (G B)^2 (fairrec/functional.py). */
FR_API size_t fr_bpr_outer_rect_workspace_bytes(int64_t Na, int64_t Nc);
FR_API int fr_bpr_outer_rect(const float* a, int64_t Na, const float* c, int64_t Nc, float inv, float* loss, float* da,
                             float* dc, void* ws, size_t ws_bytes, void* stream);
FR_API int fr_bpr_outer2(const float* pos, const float* neg, const float* pos_bias, const float* neg_bias, int64_t B,
                         float* loss, float* d_pos, float* d_neg, float* d_pos_bias, float* d_neg_bias, void* ws,
                         size_t ws_bytes, void* stream);
/* Row dots with the rows of a [A, dim] reused by `reps` row blocks of b [reps*A, dim] (a user row against its positive and
 * its negative item row of one [2B, dim] lookup): out[r*A + i] = a[i] . b[r*A + i];
 * da[i] = sum_r g[r*A + i] b[r*A + i] (in r order), db[r*A + i] = g[r*A + i] a[i]; da or db may be NULL. */
FR_API int fr_rowdot_rep_fwd(const float* a, const float* b, int64_t A, int32_t reps, int32_t dim, float* out, void* stream);
FR_API int fr_rowdot_rep_bwd(const float* g, const float* a, const float* b, int64_t A, int32_t reps, int32_t dim, float* da,
                             float* db, void* stream);
/* fr_rowdot_rep_bwd with the gradient of `a` left unsummed: da_sep[r*A + i] = g[r*A + i] b[r*A + i] ([reps*A, dim]) -- what
 * `reps` fr_rowdot_bwd calls write, in one launch, for a caller whose autograd graph adds the parts up in its own order (the
 * reference's two torch.mul(u, i).sum(-1) nodes, pfcn_pmf.py:182-183). */
FR_API int fr_rowdot_rep_bwd_sep(const float* g, const float* a, const float* b, int64_t A, int32_t reps, int32_t dim,
                                 float* da_sep, float* db, void* stream);
FR_API int fr_softmax_ce(const float* logits, const int64_t* label, int64_t M, int32_t C, float* loss, float* dlogits,
                         void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream);

/* ---- FairGo graph ops (fairgo_pmf.py / fairgo_gcn.py) -----------------------------------------------------------
 * fr_spmm_csr        : Y = L X for a CSR matrix (torch.sparse.mm at fairgo_pmf.py:198 with L = D^-1 A, :102-129); the
 *                      backward is the same call on the CSR of L^T
 * fr_row_gather      : out[j,:] = X[idx[j],:]           (`all_embeddings[user]`, fairgo_pmf.py:178-179, :194)
 * fr_row_scatter_sum : its backward as a dense [n_rows, dim] gradient, duplicates summed in ascending position
 * fr_mse             : nn.MSELoss (fairgo_pmf.py:182): loss[0], dpred = 2 (pred - target) / B */
FR_API int fr_spmm_csr(const int64_t* indptr, const int32_t* col, const float* val, const float* X, int64_t n_rows,
                       int32_t dim, float* Y, void* stream);
/* fr_spmm_csr_sel : the same product where the batch can see it (the frontier-restricted propagation of
 *                   fairgo_pmf.py:196-200, :204-216): Y[i,:] = sum_j val[j] * X[xrow(col[j]),:] over the nonzeros of row
 *                   rows[i] (rows == NULL: row i) in ascending j, xrow(c) = map ? map[c] : c, nonzeros with map[c] < 0
 *                   skipped.  `rows` (int32 [n_out]) makes Y compact; `map` (int32 [number of columns]) lets X be a compact
 *                   block of a whole-table operand or, on the CSR of L^T, names the columns that carry a gradient row.
 *                   `map_bits` (optional, uint32 [ceil(columns / 32)]): bit c set exactly where map[c] >= 0 -- the kernel
 *                   then reads the map only at set bits (the bitmap of 11 M columns is 1.4 MB and lives in every L2; the
 *                   map is 44 MB).  Kept terms are added in fr_spmm_csr's order: the rows equal the whole-table product's. */
FR_API int fr_spmm_csr_sel(const int64_t* indptr, const int32_t* col, const float* val, const float* X, const int32_t* rows,
                           int64_t n_out, const int32_t* map, const uint32_t* map_bits, int32_t dim, float* Y, void* stream);
FR_API int fr_row_gather(const float* X, const int64_t* idx, int64_t M, int64_t n_rows, int32_t dim, float* out,
                         uint32_t* err_flag, void* stream);
FR_API size_t fr_row_scatter_workspace_bytes(int64_t M);
FR_API int fr_row_scatter_sum(const float* g, const int64_t* idx, int64_t M, int64_t n_rows, int32_t dim, float* dX,
                              void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream);
/* ... ADDED into a dX that already holds another contribution to the same gradient (no clear): the backward of a table that
 * is both row-gathered and propagated (fairgo_pmf.py:178-201) then costs one dense [n_rows, dim] pass instead of three. */
FR_API int fr_row_scatter_add(const float* g, const int64_t* idx, int64_t M, int64_t n_rows, int32_t dim, float* dX,
                              void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream);
/* The same two calls when the table they differentiate is the OUTPUT of an activation (FairGo's filtered table,
 * fairgo_pmf.py:173-185 through layers.py:62-72) and what is wanted is the gradient at the activation's INPUT: every output
 * row goes on through act'(act_src[row]) (act = 1 relu, 2 leaky relu, 3 sigmoid, 4 tanh; the derivative through the output,
 * as fr_act_bwd takes it) inside the launches -- a row no term reaches is stored as zeros without reading act_src, a row whose
 * bit in `skip_bits` (uint32 [ceil(n_out / 32)], optional) is set is left unscaled for fr_row_scatter_add_act, which scales
 * the rows of `idx` after it has added to them.  Together: (L^T dY + scatter(g)) o act'(act_src), bit for bit what
 * fr_spmm_csr_sel, fr_row_scatter_add and fr_act_bwd give in three whole-table passes. */
FR_API int fr_spmm_csr_sel_act(const int64_t* indptr, const int32_t* col, const float* val, const float* X, const int32_t* rows,
                               int64_t n_out, const int32_t* map, const uint32_t* map_bits, int32_t dim, float* Y,
                               const float* act_src, int32_t act, const uint32_t* skip_bits, void* stream);
FR_API int fr_row_scatter_add_act(const float* g, const int64_t* idx, int64_t M, int64_t n_rows, int32_t dim, float* dX,
                                  void* ws, size_t ws_bytes, const float* act_src, int32_t act, uint32_t* err_flag, void* stream);
FR_API int fr_mse(const float* pred, const float* target, int64_t B, float* loss, float* dpred, void* ws, size_t ws_bytes,
                  void* stream);

/* ---- NFCF (nfcf.py) ---------------------------------------------------------------------------------------
 * Loss head of NFCF.calculate_loss, nfcf.py:99-110: y [B] is the scorer MLP's output AFTER its last ReLU
 * (layers.py:63-70); out = sigmoid(y) (:73); loss = BCELoss(out, label) (:105) [+ fair_weight * differential
 * fairness over the label==1 rows (:76-97) when `item_ws` != NULL, i.e. when fine-tuning].  `item_ws` is the
 * item table's workspace after fr_table_gather_train on the batch's item ids (its segments = torch.unique(item)).
 * Writes out [B], dy [B] = dLoss/dy (the scorer's backward starts from it), loss[3] = (loss, bce, df). */
FR_API size_t fr_nfcf_loss_workspace_bytes(int64_t B);
FR_API int fr_nfcf_loss(const float* y, const float* label, const float* sst, int64_t B, float fair_weight, void* item_ws,
                        size_t item_ws_bytes, int32_t dim, float* out, float* dy, float* loss, void* ws, size_t ws_bytes,
                        uint32_t* err_flag, void* stream);

/* The same loss head behind the fused scorer (fr_scorer_fwd below), which has already written out, the BCE share of dy and the
 * per-workgroup partials: bce_part [n_part] (sums of the rows' BCE terms) and, with `sst`, mm_part [n_part, 2] ((min, max)
 * of the attribute over each workgroup's positive rows).  Adds the differential-fairness share to dy (when item_ws != NULL)
 * and writes loss[3]; ws as for fr_nfcf_loss. */
FR_API int fr_nfcf_loss_tail(const float* label, const float* sst, int64_t B, float fair_weight, void* item_ws,
                             size_t item_ws_bytes, int32_t dim, const float* out, float* dy, float* loss,
                             const float* bce_part, const float* mm_part, int32_t n_part, void* ws, size_t ws_bytes,
                             uint32_t* err_flag, void* stream);

/* ---- the NFCF scorer in one forward and one backward launch (csrc/scorer.hip) ----------------------------------------
 * MLPLayers([k0 + k1, n1, n2, 1], dropout p) on cat(x0, x1) (nfcf.py:40, :68-73; layers.py:56-85: per layer Dropout ->
 * Linear -> ReLU, the last layer included).  Shapes: k0 == k1, multiples of 32, k0 + k1 <= 512; n1 in 32..128, n2 in 32..64,
 * multiples of 32 (fr_scorer_supported; the layer-by-layer entry points above cover everything else).  Dropout is
 * fr_dropout_apply's generator: `off_*` are the element offsets of the four dropped tensors inside the call's pattern
 * (the layered form's: x0 at 0, x1 behind it, then the two hidden activations), `counter` / `used_out` / `tick_state` as
 * there (the forward is the pattern's first and last launch).
 *   fr_scorer_fwd  x0d / x1d [B, k] = the dropped inputs (p > 0 only; what the weight gradient multiplies), h1 [B, n1],
 *                  h2 [B, n2] = the dropped hidden activations, y [B] = the output after its ReLU.  With `label`: out =
 *                  sigmoid(y), dy = d mean(BCE) / dy, bce_part / mm_part [fr_scorer_blocks(B)] (see fr_nfcf_loss_tail);
 *                  `loss` != NULL (no fairness term follows: the pre-training stage): loss[3] = (mean BCE, mean BCE, 0),
 *                  written by the launch's last workgroup -- no fr_nfcf_loss_tail call then.
 *   fr_scorer_bwd  from dy (times gscale[0] when given): dz3 [B], dz2 [B, n2], dz1 [B, n1] = the gradients at the three
 *                  pre-activations, dx0 / dx1 = the input blocks' gradients (either may be NULL: a frozen table), w3part
 *                  [fr_scorer_blocks(B), n2 + 1] = per-workgroup shares of (dW3 | db3), summed by fr_parts_sum.
 *                  dW1 / db1 = fr_linear_bwd_weight(dz1, x0d | x1d), dW2 / db2 = fr_linear_bwd_weight(dz2, h1). */
typedef struct fr_scorer {
    int32_t k0, k1, n1, n2;
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    float p;
    uint64_t seed;
    uint64_t off_x0, off_x1, off_h1, off_h2;
} fr_scorer;
FR_API int fr_scorer_supported(const fr_scorer* s);
FR_API int64_t fr_scorer_blocks(int64_t B);
FR_API int fr_scorer_fwd(const fr_scorer* s, const float* x0, const float* x1, int64_t B, const int64_t* counter,
                         int64_t* used_out, int64_t* tick_state, float* x0d, float* x1d, float* h1, float* h2, float* y,
                         const float* label, const float* sst, float* out, float* dy, float* bce_part, float* mm_part,
                         float* loss, void* stream);
FR_API int fr_scorer_bwd(const fr_scorer* s, const float* dy, const float* gscale, const float* y, const float* h1,
                         const float* h2, int64_t B, const int64_t* used, float* dz1, float* dz2, float* dz3, float* dx0,
                         float* dx1, float* w3part, void* stream);
/* The weight gradients of several layers in two launches (every product, then every slab sum): job = fr_linear_bwd_weight's
 * (dY at the pre-activation, x0 | x1, N) -> dW [N, k0 + k1], db [N] (may be NULL), fast form only (N, k0, k0 + k1 multiples
 * of 32, 16-byte aligned; FR_EUNSUPPORTED otherwise).  A job with dY == NULL sums `n_parts` partial results
 * parts[n_parts][N * (k0 + k1)] into dW (the scorer's last layer: fr_scorer_bwd's w3part).  At most FR_WGRAD_MAX jobs. */
#define FR_WGRAD_MAX 8
typedef struct fr_wgrad_job {
    const float* dY;
    const float* x0;
    int32_t k0;
    const float* x1;
    int32_t k1;
    int32_t N;
    float* dW;
    float* db;
    const float* parts;
    int32_t n_parts;
} fr_wgrad_job;
FR_API size_t fr_linear_bwd_weight_multi_workspace_bytes(const fr_wgrad_job* jobs, int32_t n, int64_t M);
FR_API int fr_linear_bwd_weight_multi(const fr_wgrad_job* jobs, int32_t n, int64_t M, void* ws, size_t ws_bytes, void* stream);
/* out[i] = sum over the parts p of part[p * n + i], in a fixed order (64 interleaved ascending chains, then a butterfly) */
FR_API int fr_parts_sum(const float* part, int32_t parts, int64_t n, float* out, void* stream);

/* The differential-fairness term (nfcf.py:76-97) of a ROW-SHARDED step on the GLOBAL batch (one process per GPU, item
 * table row r on rank r mod G): M[k, g], K and the mean of eps are statistics of the whole batch, so every positive row's
 * (score, attribute) travels to the owner of its item and the per-(item, group) sums come back -- two all-to-alls that
 * the caller issues between the three launches below.  Both buffers have one chunk of `cap` slots + 1 tail per peer:
 *   rec   [G, cap + 1, 2] floats  slot = (sigmoid score, or -1 where label != 1; attribute), tail = the sender's (min, max)
 *                                 of the attribute over its positive rows
 *   reply [G, cap + 1, 4] floats  slot = (S0 with the sign bit set on the ONE member that reports the item's eps, S1, n0, n1),
 *                                 tail = (K_owner = owned items with a positive row, smin, smax, 0)
 * A row's slot is the one its item id took in the lookup's id exchange: slot[b] = o * slot_stride + slot_off + k
 * (fr_bucket_by_owner), i.e. chunk o, position k.
 *   fr_nfcf_df_pack  (requester) writes the records of this rank's B rows and the tails.
 *   fr_nfcf_df_owner (owner) reduces over the segments of the ids it received (`item_ws` = the item table's workspace after
 *                    fr_table_gather_train on the G * cap received slots; members in rank order, then batch position = the
 *                    order of the concatenated batch) and writes the reply.
 *   fr_nfcf_df_apply (requester) K = sum of the tails' K_owner; M = (S + 1/K) / (n + 1); eps = |log M0 - log M1|;
 *                    dy[b] += scale * fair_weight * d(mean eps)/d score_b * out_b (1 - out_b); loss[0] += fair_weight *
 *                    scale * (sum of the eps this rank reports) / K, loss[2] = that DF share.  scale = G where the caller
 *                    averages the ranks' gradients and losses (the DF term is not a per-rank mean).
 * `ws` (fr_nfcf_df_workspace_bytes(B, G * cap), ZERO-initialised once, reusable) holds partial sums and arrival tickets. */
FR_API size_t fr_nfcf_df_workspace_bytes(int64_t B, int64_t n_slots);
FR_API int fr_nfcf_df_pack(const float* out, const float* label, const float* sst, const int32_t* slot, int32_t slot_stride,
                           int32_t slot_off, int32_t cap, int64_t B, int32_t G, float* rec, void* ws, size_t ws_bytes,
                           void* stream);
FR_API int fr_nfcf_df_owner(void* item_ws, size_t item_ws_bytes, int32_t dim, const float* rec, int32_t G, int32_t cap,
                            float* reply, void* ws, size_t ws_bytes, int64_t B, uint32_t* err_flag, void* stream);
FR_API int fr_nfcf_df_apply(const float* reply, const int32_t* slot, int32_t slot_stride, int32_t slot_off, int32_t cap,
                            int32_t G, const float* out, const float* label, const float* sst, int64_t B, float fair_weight,
                            float scale, float* dy, float* loss, void* ws, size_t ws_bytes, void* stream);

/* Dense fused Adam step for small dense parameters (MLP weights, biases): one step of
 * torch.optim.Adam on a flat fp32 tensor, `step` = the step being applied. */
FR_API int fr_adam_dense(float* p, const float* g, float* m, float* v, int64_t n, const fr_adam* adam, int32_t step,
                  void* stream);
/* The same for many tensors at once (all dense parameters of an optimizer group: one launch per FR_ADAM_DENSE_MAX tensors
 * instead of one per tensor); every tensor keeps its own step counter, as torch.optim.Adam's per-parameter state does. */
#define FR_ADAM_DENSE_MAX 48
typedef struct fr_dense_desc {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    int32_t step;
    const int32_t* step_dev;   /* optional device counter, as fr_table.step_dev: effective step = *step_dev + step */
} fr_dense_desc;
FR_API int fr_adam_dense_multi(const fr_dense_desc* descs, int32_t n_tensors, const fr_adam* adam, void* stream);

/* ---- negative sampler (next-row f-1: the batch feed), bit-exact with the reference's host sampler ---------------
 * state: numpy's legacy RandomState layout in device memory, uint32 key[624] followed by uint32 pos (625 words), so the
 * stream can be exchanged with np.random.get_state() / set_state() at any point.
 *   fr_mt19937_seed     : np.random.seed(seed)                          (numpy mt19937_seed)
 *   fr_sample_negatives : Sampler.sample_by_user_ids, sampler.py:283-303 -> sample_by_key_ids :145-197 with
 *                         _uni_sampling = np.random.randint(low, high, .) :240-241:
 *                         out[j] for j in [0, n_keys*num) is drawn for key key_ids[j % n_keys]; draw all, then re-draw
 *                         -- from the continuing stream, in ascending position order -- exactly the positions whose
 *                         value is in the key's used-set, until none is left (rounds_out[0] = number of rounds).
 *                         used-set of key u = used_items[used_indptr[u] .. used_indptr[u+1]) sorted ascending.
 *                         used_indptr = NULL: plain np.random.randint(low, high, n_keys*num) (key_ids unused).
 *                         high - 1 - low < 2^32 - 1 (numpy then draws single 32-bit words, masked rejection).
 * One workgroup per call; the state after the call is exactly numpy's, so calls chain without host syncs.
 *   fr_sample_negatives_calls : a SEQUENCE of single-key calls on one stream in one launch: call c fills
 *                         out[call_offsets[c] .. call_offsets[c+1]) for key call_keys[c] and completes its re-draw rounds
 *                         before call c+1 draws -- the evaluation loader's user-by-user sampling
 *                         (general_dataloader.py:141-146).  max_call >= the longest call (sizes the workspace). */
FR_API int fr_mt19937_seed(uint32_t* state, uint32_t seed, void* stream);
FR_API size_t fr_sample_negatives_workspace_bytes(int64_t total);
/* ... for fr_sample_negatives_calls: with this much workspace (`total` = values of the whole sequence) a sequence of 16 or more
 * calls is resolved SPECULATIVELY -- the stream's accepted values generated once, every call laid out as if none before it
 * had collided with its used-set, the rare colliding call resolved round by round and the rest shifted (csrc/sampler.hip) --
 * instead of call by call: the same values and the same generator state, 30 ms -> ~1.5 ms for the ~2 800 calls of an
 * evaluation batch.  With fr_sample_negatives_workspace_bytes(max_call) bytes the calls run one after the other. */
FR_API size_t fr_sample_negatives_calls_workspace_bytes(int64_t total, int64_t max_call);
FR_API int fr_sample_negatives(uint32_t* state, int64_t low, int64_t high, const int64_t* key_ids, int64_t n_keys,
                               int32_t num, const int64_t* used_indptr, const int32_t* used_items, int64_t n_users,
                               int64_t* out, int32_t* rounds_out, void* ws, size_t ws_bytes, uint32_t* err_flag,
                               void* stream);
FR_API int fr_sample_negatives_calls(uint32_t* state, int64_t low, int64_t high, const int64_t* call_keys,
                                     const int64_t* call_offsets, int64_t n_calls, int64_t max_call,
                                     const int64_t* used_indptr, const int32_t* used_items, int64_t n_users, int64_t* out,
                                     void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream);

/* ---- the row sets of a frontier-restricted propagation (csrc/frontier.hip) -----------------------------------------------------
 * Which rows of H_l = L H_(l-1) a batch can see (fairgo_pmf.py:196-216 aggregates every layer's rows of the batch's users): the
 * sets fr_spmm_csr_sel takes.  A set is a bitmap over the graph rows (bits: (n_rows + 31) / 32 words, zeroed by the caller):
 * fr_frontier_mark sets the bits of an id list (ids outside [0, n_rows) raise FR_DEV_ERR_INDEX_RANGE), fr_frontier_expand the
 * bits of the columns of the listed rows of a CSR matrix, fr_frontier_count writes popcount(bits[w]) per word, and -- given the
 * inclusive cumulative sum of those counts -- fr_frontier_scatter writes the set's row ids in ascending order (rows_out) and
 * pos[row] = rank of the row in the set, -1 outside it (pos: n_rows entries, every one written). */
FR_API int fr_frontier_mark(const int64_t* ids, int64_t n, int64_t n_rows, uint32_t* bits, uint32_t* err_flag, void* stream);
FR_API int fr_frontier_expand(const int64_t* indptr, const int32_t* col, const int32_t* rows, int64_t n_list, uint32_t* bits,
                              void* stream);
FR_API int fr_frontier_count(const uint32_t* bits, int64_t n_rows, int32_t* count, void* stream);
FR_API int fr_frontier_scatter(const uint32_t* bits, const int32_t* incl, int64_t n_rows, int32_t* rows_out, int32_t* pos,
                               void* stream);

/* ---- FOCF's item-complete batcher (next-row f-3): the picks of a whole epoch in one HOST call ---------------------------------
 * focf_dataloader.py:37-51 composes a batch by `np.random.choice(select_item[is_select], 1, False)` per picked item: numpy's
 * legacy choice permutes the whole candidate list per pick and reads one element.  fr_focf_compose_epoch makes the same draws
 * from the same stream (state: numpy's legacy layout key[624] + pos in HOST memory, advanced in place) and returns the same
 * picks without building the permutations (csrc/focf_compose.hip).  item_uniques [n_uniq] ascending, indptr [n_items + 1] the
 * CSR of the item-sorted interaction table, `pr` / `pr_end` / `step` the loader's counters; picks_out [cap] the picked items,
 * batch_end_out[b] the number of picks up to and including batch b.  No device work, no stream. */
FR_API int fr_focf_compose_epoch(uint32_t* state, const int64_t* item_uniques, int64_t n_uniq, const int64_t* indptr,
                                 int64_t step, int64_t pr, int64_t pr_end, int64_t* picks_out, int64_t cap,
                                 int64_t* batch_end_out, int64_t max_batches, int64_t* n_batches_out);

/* ---- epoch shuffle (next-row f-1: the batch feed): torch.randperm(n) on the device, bit-exact ---------------------------
 * Replaces the `torch.randperm(self.length)` behind Interaction.shuffle (interaction.py:293-297), the training loader's
 * per-epoch shuffle (abstract_dataloader.py:81-84).  Underneath: ATen's randperm_cpu for n < 2^32 / 20 -- r[i] = i, then for
 * i = 0 .. n-2: swap(r[i], r[i + mt19937() % (n - i)]) on single 32-bit outputs of torch's CPU generator (the standard
 * MT19937).  state: uint32 key[624] followed by uint32 pos (625 words; pos = index of the next unused word, 624 = regenerate
 * first), taken from torch.get_rng_state() by the caller and left where the n - 1 draws put it, so the host generator can be
 * set to exactly the state torch.randperm(n) would have left.  out: int64 [n].  The swap chain is resolved in parallel
 * (csrc/randperm.hip); only the generator is walked sequentially, by one workgroup. */
FR_API size_t fr_randperm_workspace_bytes(int64_t n);
FR_API int fr_randperm(uint32_t* state, int64_t n, int64_t* out, void* ws, size_t ws_bytes, void* stream);

/* ---- evaluation metrics (next-row f-2), recbole/evaluator/metrics.py ------------------------------------------------
 *   fr_topk_metrics           : rec_topk int32 [n_users, k+1] = hit flags of the ranked list | number of positives
 *                               (collector.py:146-154) -> out[6][k] doubles = Hit, MRR, NDCG, Recall, Precision, MAP
 *                               @ 1..k, means over users (metrics.py:40-232)
 *   fr_group_sums             : stats[s][g][0..2] = sum value, count, sum wtrue over the members
 *                               perm[seg_start[s] .. seg_start[s+1]) of segment s whose group index is g (perm = NULL:
 *                               identity).  The (item, group) tables of the fairness metrics (:948-970, :1322-1335)
 *   fr_fair_metrics_from_stats: out[0..4] = Value, Absolute, Under, Over unfairness (n_groups == 2; `+ 1e-5` counts as
 *                               the reference) and DifferentialFairness (float32 table, alpha = 1/n_segments) as means
 *                               over the segments (:972-979, :1068-1075, :1164-1171, :1260-1267, :1337-1342)
 * Double-precision sums in a fixed order. */
/* fr_eval_hits: the `rec.topk` matrix of a batch of users (collector.py:146-154): rec_topk int32 [n_rows, k+1], entry
 * [u][j] = 1 if topk_idx[u][j] (the j-th ranked item id of row u) is a positive of row u, [u][k] = its number of
 * positives.  pos_keys = the sorted keys row * n_items + item of the batch's positives (no dense 0/1 matrix). */
FR_API int fr_eval_hits(const int64_t* topk_idx, int64_t n_rows, int32_t k, int64_t n_items, const int64_t* pos_keys,
                        int64_t n_pos, int32_t* rec_topk, void* stream);
/* fr_eval_topk_segments: the ranked lists of an evaluation batch under the uniN protocol (trainer.py:441-456 + collector.py:149).
 * User row u's candidates are rows [seg_start[u], seg_start[u+1]) of the batch (items int64, scores float); topk_idx int64
 * [n_users, k] = its k best DISTINCT items, best first, padded with 0; flags int32 [n_users]: bit 0 = two of its first k + 1
 * entries score equal (rank that row with fr_topk_like_torch_cpu), bit 1 = fewer than k + 1 distinct candidates.  1 <= k <= 62.
 * fr_eval_lookup_segments: out[q] = the score of item q_items[q] among user row q_rows[q]'s candidates, -inf if absent. */
FR_API int fr_eval_topk_segments(const int64_t* seg_start, int64_t n_users, const int64_t* items, const float* scores,
                                 int32_t k, int64_t* topk_idx, int32_t* flags, void* stream);
FR_API int fr_eval_lookup_segments(const int64_t* seg_start, int64_t n_users, const int64_t* items, const float* scores,
                                   const int64_t* q_rows, const int64_t* q_items, int64_t n_q, float* out, void* stream);
/* fr_topk_like_torch_cpu (HOST, no stream): torch.topk(rows, k, dim=-1) of the CPU backend with ITS order among equal values --
 * what the reference's evaluation ranks with (collector.py:149 on the dense -inf matrix of trainer.py:441-456).  rows: float
 * [n_rows, n] in host memory; idx_out int64 [n_rows, k] (val_out float [n_rows, k], optional).  For the user rows whose list
 * hangs on an exact score tie; every other row is ranked on the device (csrc/topk_host.hip says why the order can be had). */
FR_API int fr_topk_like_torch_cpu(const float* rows, int64_t n_rows, int64_t n, int32_t k, int64_t* idx_out, float* val_out);
FR_API size_t fr_topk_metrics_workspace_bytes(int64_t n_users, int32_t k);
FR_API int fr_topk_metrics(const int32_t* rec_topk, int64_t n_users, int32_t k, double* out, void* ws, size_t ws_bytes,
                           void* stream);
FR_API int fr_group_sums(const int64_t* perm, const int64_t* seg_start, int64_t n_segments, const int32_t* group,
                         const float* value, const float* wtrue, int32_t n_groups, double* stats, void* stream);
FR_API size_t fr_fair_metrics_workspace_bytes(int64_t n_segments);
FR_API int fr_fair_metrics_from_stats(const double* stats, int64_t n_segments, int32_t n_groups, double* out, void* ws,
                                      size_t ws_bytes, void* stream);

/* ---- built-in profiler -------------------------------------------------------------------------------
 * When enabled every kernel launch of this library is bracketed by a hipEvent pair recorded on the
 * launch stream; fr_prof_read synchronises the outstanding events and returns the accumulated device
 * time and launch count of one kernel kind (bench.py's `roofline.achieved` comes from here).
 * Process-global diagnostic state (one table of kernel kinds per process = per GPU), guarded by a mutex: any thread may
 * launch while it is on, the counts of all threads are summed; enable / reset / read from one thread at a time.  Apart
 * from this table, the library's helper stream and the last-error string (thread-local), the entry points keep no state
 * between calls: everything lives in the caller's tables and workspaces. */
FR_API int fr_prof_enable(int on);
FR_API int fr_prof_reset(void);
FR_API int fr_prof_kernel_count(void);
FR_API const char* fr_prof_kernel_name(int kind);
FR_API int fr_prof_read(int kind, double* total_ms, int64_t* count);
/* Algorithmic work the launches of `kind` stood for since the last reset: FLOP (2 M N K per product) for linear_fwd /
 * linear_bwd_input / linear_bwd_weight, bytes (12 nnz + 8 n_rows dim) for spmm_csr, 0 for the kinds that do not account.
 * work / total_ms is the rate bench.py holds against the MFMA / HBM peak. */
FR_API int fr_prof_read_work(int kind, double* work);

#ifdef __cplusplus
}
#endif
#endif /* FAIRREC_HIP_H */
