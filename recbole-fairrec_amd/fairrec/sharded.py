"""Row-sharded FOCF training step for G GPUs of one node (one process per GPU, RCCL over xGMI through
torch.distributed).

Semantics: ONE optimizer step on the global batch = concatenation of every rank's local batch (rank order), i.e.
exactly what the single-device reference computes on that batch: loss = MSE over G*B interactions +
fair_weight * mean over the distinct items of the GLOBAL batch.  Tables are split by `row mod G`; Adam state
lives with its rows.

Per step, 5 all-to-alls on fixed-capacity buffers, no all-reduce, no host sync, no packing copies (the kernels read
and write the exchange layouts in place, fairrec_hip.h "slot layout"):

    ids    [G, 2*cap+1]      user ids | item ids | (min, max) of the local sensitive column
    rows   [G, 2*cap+1, D]   owners: lazy gather of the requested rows, written in place
    rec    [G, 3, cap]       (pred, rating, sst) of every interaction, to the item's owner
    reply  [G, cap+3]        owners: fairness part of dLoss/dpred per record | (K_owner, fair_owner, sq_rank)
    grads  [G, 2*cap+1, D]   gradient rows back to the owners -> duplicate-sum + Adam

Launches per step: bucket (both id lists) | sort (both tables) | gather (both tables) | score | fair | grads (+ loss) |
apply (both tables) = 7 kernels around the 5 collectives.  The first two kernels and the id exchange depend on
nothing but the id columns: when the caller names the next batch (`next_batch`, the trainer's one-batch dataloader
look-ahead) they run one step AHEAD on a side stream, on the second copy of the index-side buffers, so the dependent
chain of a step starts at the gather.  The look-ahead id exchange uses the same communicator and is issued while
the fairness kernel runs, where the main chain leaves it idle.

The three scalars at the tail of every reply chunk give each rank K, the fairness sum and the squared-error sum of
the global batch (summed in rank order, so identical everywhere).  fair_objective none needs no rec/reply; its
loss takes one 1-float all-reduce.

The kernels come from an `ops` object (default: the HIP library through fairrec._C).  Tests inject a CPU double
to exercise this exchange schedule over gloo without a GPU; the product path is HIP only.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.distributed as dist

from . import _C
from .optim import AdamHyper, LazyTable

SORT_MAX = 16384      # FR_SORT_MAX: an owner sorts the G*cap ids it receives for one table in one workgroup
TAIL = 3              # FR_SHARD_TAIL


def shard_rows(n_rows: int, rank: int, world: int) -> int:
    """Number of rows r < n_rows with r mod world == rank."""
    return (n_rows - rank + world - 1) // world if n_rows > rank else 0


def shard_of(full: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """The rows of a full table this rank owns, in local-row order (row = local * world + rank)."""
    return full[rank::world].contiguous()


def exchange_capacity(B: int, G: int, capacity_factor: float) -> int:
    """Slots per (requester, owner) pair: the mean fill B/G times `capacity_factor`, bounded by what one owner can
    sort in one launch (G*cap <= FR_SORT_MAX)."""
    return max(1, min(B, int(math.ceil(capacity_factor * B / G)) + 64, SORT_MAX // G))


def capacity_is_sort_bound(B: int, G: int, capacity_factor: float) -> bool:
    """True when the owner-side sort (FR_SORT_MAX ids per table and launch), not `capacity_factor`, sets the capacity:
    raising the factor can then not help a skewed batch, only a smaller per-rank batch (or more ranks) can."""
    return SORT_MAX // G < min(B, int(math.ceil(capacity_factor * B / G)) + 64)


class HipOps:
    """The kernels of the sharded step, bound to the HIP library.  Buffers are flat tensors; `*_off` are element
    (row) offsets into them."""

    def __init__(self, device):
        self.device = torch.device(device)
        _C.lib()

    def make_table(self, weight):
        return LazyTable(weight)

    def bucket_by_owner(self, idx, G, cap, stride, offset, send, slot, counts, aux, aux_slot, err):
        _C.check(_C.lib().fr_bucket_by_owner(idx.data_ptr(), idx.numel(), G, cap, stride, offset, send.data_ptr(),
                                             slot.data_ptr(), counts.data_ptr(), _C.ptr(aux), aux_slot, err.data_ptr(),
                                             _C.current_stream()), "fr_bucket_by_owner")

    def bucket_pair(self, idx_a, idx_b, G, cap, stride, off_a, off_b, send, slot_a, slot_b, counts, aux, aux_slot, err):
        _C.check(_C.lib().fr_bucket_pair_by_owner(idx_a.data_ptr(), idx_b.data_ptr(), idx_a.numel(), G, cap, stride, off_a,
                                                  off_b, send.data_ptr(), slot_a.data_ptr(), slot_b.data_ptr(),
                                                  counts.data_ptr(), _C.ptr(aux), aux_slot, err.data_ptr(),
                                                  _C.current_stream()), "fr_bucket_pair_by_owner")

    # --- look-ahead: the index-only part of the NEXT step runs on a side stream -------------------------
    def side(self, fork=True):
        """Context: launches go to the side stream; fork: it first waits for everything enqueued on the current one."""
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream(device=self.device)
        if fork:
            self._side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self._side)

    def join_side(self):
        torch.cuda.current_stream().wait_stream(self._side)

    def used_on_side(self, *things):
        """Tell torch's allocator that these tensors (or every tensor held by these objects) are used on the side stream as
        well: it recycles a freed tensor's memory in the order of the streams it knows the tensor was used on, and an engine
        may be dropped with look-ahead work still queued."""
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream(device=self.device)
        if torch.cuda.is_current_stream_capturing():
            return

        def walk(x, depth=0):
            if isinstance(x, torch.Tensor):
                if x.is_cuda:
                    x.record_stream(self._side)
            elif isinstance(x, (list, tuple)) and depth < 3:
                for y in x:
                    walk(y, depth + 1)
            elif hasattr(x, "__dict__") and depth < 1:
                for y in vars(x).values():
                    walk(y, depth + 1)
        for t in things:
            walk(t)

    def alloc_ws(self, table, M):
        return torch.empty(_C.lib().fr_table_train_workspace_bytes(M, table.dim), dtype=torch.uint8, device=self.device)

    def sort_pair(self, ta, tb, ids, off_a, off_b, M, chunk, stride, ws_a, ws_b, err):
        LazyTable.sort_pair(ta, tb, ids.data_ptr() + 8 * off_a, ids.data_ptr() + 8 * off_b, M, ws_a, ws_b, chunk, stride,
                            err)

    def gather_train_pair(self, ta, tb, hyper, ids, off_a, off_b, M, chunk, stride, rows, ws_a, ws_b, err):
        """Segments of both id lists are already in ws_a / ws_b (sort_pair)."""
        D = ta.dim
        ta._ws, tb._ws = ws_a, ws_b
        LazyTable.gather_train_pair(ta, tb, hyper, ids.data_ptr() + 8 * off_a, ids.data_ptr() + 8 * off_b, M,
                                    rows.data_ptr() + 4 * off_a * D, rows.data_ptr() + 4 * off_b * D, chunk, stride, err,
                                    prepared=True)

    def apply_grad_pair(self, ta, tb, hyper, M, chunk, stride, rows, grads, off_a, off_b, sweep_a, sweep_b):
        D = ta.dim
        LazyTable.apply_grad_pair(ta, tb, hyper, M, rows.data_ptr() + 4 * off_a * D, grads.data_ptr() + 4 * off_a * D,
                                  rows.data_ptr() + 4 * off_b * D, grads.data_ptr() + 4 * off_b * D, sweep_a, sweep_b,
                                  chunk, stride)

    def flush(self, table, hyper):
        table.flush(hyper)

    def shard_score(self, rows, slot_u, slot_i, rating, sst, n_global, pred, coef, rec, cap, slot_stride, slot_offset,
                    sq, sq_part):
        """sq: [1] sum of squared errors, or None to leave only the per-workgroup partials in sq_part."""
        B, D = slot_u.numel(), rows.shape[1]
        _C.check(_C.lib().fr_focf_shard_score(rows.data_ptr(), rows.data_ptr(), slot_u.data_ptr(), slot_i.data_ptr(),
                                              rating.data_ptr(), _C.ptr(sst), B, D, n_global, pred.data_ptr(),
                                              coef.data_ptr(), _C.ptr(rec), cap, slot_stride, slot_offset, _C.ptr(sq),
                                              sq_part.data_ptr(), _C.current_stream()), "fr_focf_shard_score")

    def shard_fair(self, item_table, n_slots, rec, cap, ids_recv, mm_slot, stride, objective, fair_weight, reply,
                   sq_part, n_sq_part, scratch, err):
        ws = item_table._ws
        G = n_slots // cap
        _C.check(_C.lib().fr_focf_shard_fair(ws.data_ptr(), ws.numel(), n_slots, item_table.dim, rec.data_ptr(), cap,
                                             ids_recv.data_ptr() + 8 * mm_slot, G, 2 * stride,
                                             _C.FOCF_OBJECTIVES[objective], fair_weight, reply.data_ptr(),
                                             sq_part.data_ptr(), n_sq_part, scratch.data_ptr(), err.data_ptr(),
                                             _C.current_stream()), "fr_focf_shard_fair")

    def nonparity_sums(self, pred, sst, ids_recv, mm_slot, stride, G, sq_part, n_sq_part, out5):
        _C.check(_C.lib().fr_focf_shard_nonparity_sums(pred.data_ptr(), sst.data_ptr(), pred.numel(),
                                                       ids_recv.data_ptr() + 8 * mm_slot, G, 2 * stride,
                                                       sq_part.data_ptr(), n_sq_part, out5.data_ptr(),
                                                       _C.current_stream()), "fr_focf_shard_nonparity_sums")

    def nonparity_coef(self, coef, sst, ids_recv, mm_slot, stride, G, global5, n_global, fair_weight, loss_out, err):
        _C.check(_C.lib().fr_focf_shard_nonparity_coef(coef.data_ptr(), sst.data_ptr(), coef.numel(),
                                                       ids_recv.data_ptr() + 8 * mm_slot, G, 2 * stride,
                                                       global5.data_ptr(), n_global, fair_weight, loss_out.data_ptr(),
                                                       err.data_ptr(), _C.current_stream()), "fr_focf_shard_nonparity_coef")

    def shard_grads(self, rows, slot_u, slot_i, coef, reply, G, n_global, fair_weight, loss_out, cap, slot_stride,
                    slot_offset, grads):
        B, D = slot_u.numel(), rows.shape[1]
        _C.check(_C.lib().fr_focf_shard_grads(rows.data_ptr(), rows.data_ptr(), slot_u.data_ptr(), slot_i.data_ptr(),
                                              coef.data_ptr(), _C.ptr(reply), G, n_global, fair_weight,
                                              _C.ptr(loss_out), cap, slot_stride, slot_offset, B, D, grads.data_ptr(),
                                              grads.data_ptr(), _C.current_stream()), "fr_focf_shard_grads")


    # --- item-owner-computes schedule (ShardedFocfEngineV2) ---------------------------------------------------------
    def bucket_sparse(self, idx, G, cap, stride, offset, send, slot, counts, err):
        """bucket_by_owner of a list with empty positions (id -1: they go nowhere)"""
        _C.check(_C.lib().fr_bucket_by_owner_sparse(idx.data_ptr(), idx.numel(), G, cap, stride, offset, send.data_ptr(),
                                                    slot.data_ptr(), counts.data_ptr(), err.data_ptr(),
                                                    _C.current_stream()), "fr_bucket_by_owner_sparse")

    def pack_records(self, slot, user, rating, sst, cap, send):
        _C.check(_C.lib().fr_shard_pack_records(slot.data_ptr(), user.data_ptr(), rating.data_ptr(), _C.ptr(sst), slot.numel(),
                                                cap, send.data_ptr(), _C.current_stream()), "fr_shard_pack_records")

    def unpack_records(self, recv, G, cap, iid, uid, islot, rating, sst, mm):
        _C.check(_C.lib().fr_shard_unpack_records(recv.data_ptr(), G, cap, iid.data_ptr(), uid.data_ptr(), islot.data_ptr(),
                                                  rating.data_ptr(), sst.data_ptr(), mm.data_ptr(), _C.current_stream()),
                 "fr_shard_unpack_records")

    def post_fair(self, reply, k_all, G, cap, sums):
        _C.check(_C.lib().fr_shard_post_fair(reply.data_ptr(), k_all.data_ptr(), G, cap, sums.data_ptr(), _C.current_stream()),
                 "fr_shard_post_fair")

    def loss_finish(self, sums, k_all, G, n_global, fair_weight, fair, loss):
        _C.check(_C.lib().fr_shard_loss_finish(sums.data_ptr(), k_all.data_ptr(), G, n_global, fair_weight, int(fair),
                                               loss.data_ptr(), _C.current_stream()), "fr_shard_loss_finish")

    def count_distinct(self, ids, n_rows, bitmap, count, out):
        """out[0] = number of distinct real ids of the list (bitmap over the owner's rows, left all-zero)"""
        _C.check(_C.lib().fr_shard_count_distinct(ids.data_ptr(), ids.numel(), n_rows, bitmap.data_ptr(), count.data_ptr(),
                                                  out.data_ptr(), _C.current_stream()), "fr_shard_count_distinct")

    def shard_score2(self, rows_u, rows_i, slot_u, slot_i, rating, sst, n_global, pred, coef, rec, cap, slot_stride,
                     slot_offset, sq, sq_part):
        """shard_score with the user rows and the item rows in two buffers"""
        B, D = slot_u.numel(), rows_u.shape[1]
        _C.check(_C.lib().fr_focf_shard_score(rows_u.data_ptr(), rows_i.data_ptr(), slot_u.data_ptr(), slot_i.data_ptr(),
                                              rating.data_ptr(), _C.ptr(sst), B, D, n_global, pred.data_ptr(),
                                              coef.data_ptr(), _C.ptr(rec), cap, slot_stride, slot_offset, _C.ptr(sq),
                                              sq_part.data_ptr(), _C.current_stream()), "fr_focf_shard_score")

    def shard_grads2(self, rows_u, rows_i, slot_u, slot_i, coef, reply, G, n_global, fair_weight, loss_out, cap,
                     slot_stride, slot_offset, grad_u, grad_i):
        B, D = slot_u.numel(), rows_u.shape[1]
        _C.check(_C.lib().fr_focf_shard_grads(rows_u.data_ptr(), rows_i.data_ptr(), slot_u.data_ptr(), slot_i.data_ptr(),
                                              coef.data_ptr(), _C.ptr(reply), G, n_global, fair_weight,
                                              _C.ptr(loss_out), cap, slot_stride, slot_offset, B, D, grad_u.data_ptr(),
                                              grad_i.data_ptr(), _C.current_stream()), "fr_focf_shard_grads")


class _Buffers:
    """Exchange and scratch buffers of one (B, cap, D) shape, allocated once (a captured step replays on them)."""

    def __init__(self, G, B, cap, D, dev, ops, U, I):
        S = 2 * cap + 1
        self.B, self.cap, self.S = B, cap, S
        f32, i64, i32 = torch.float32, torch.int64, torch.int32
        # index-side state exists twice: the NEXT step's bucket / id exchange / owner sort runs while this step computes
        self.ids_send = [torch.full((G * S,), -1, dtype=i64, device=dev) for _ in range(2)]
        self.ids_recv = [torch.full((G * S,), -1, dtype=i64, device=dev) for _ in range(2)]
        self.slot_u = [torch.empty(B, dtype=i32, device=dev) for _ in range(2)]
        self.slot_i = [torch.empty(B, dtype=i32, device=dev) for _ in range(2)]
        self.counts = [torch.empty(2 * G, dtype=i32, device=dev) for _ in range(2)]
        self.ws_u = [ops.alloc_ws(U, G * cap) for _ in range(2)]
        self.ws_i = [ops.alloc_ws(I, G * cap) for _ in range(2)]
        self.rows_send = torch.zeros((G * S, D), dtype=f32, device=dev)
        self.rows_recv = torch.empty((G * S, D), dtype=f32, device=dev)
        self.g_send = torch.zeros((G * S, D), dtype=f32, device=dev)
        self.g_recv = torch.empty((G * S, D), dtype=f32, device=dev)
        self.rec_send = torch.zeros(G * 3 * cap, dtype=f32, device=dev)
        self.rec_recv = torch.empty(G * 3 * cap, dtype=f32, device=dev)
        self.reply_send = torch.zeros(G * (cap + TAIL), dtype=f32, device=dev)
        self.reply_recv = torch.empty(G * (cap + TAIL), dtype=f32, device=dev)
        self.pred = torch.empty(B, dtype=f32, device=dev)
        self.coef = torch.empty(B, dtype=f32, device=dev)
        self.sq = torch.zeros(1, dtype=f32, device=dev)
        self.np5 = torch.zeros(5, dtype=f32, device=dev)
        self.loss = torch.zeros(3, dtype=f32, device=dev)
        self.n_sq_part = (B + 3) // 4
        self.sq_part = torch.zeros(self.n_sq_part + 1, dtype=f32, device=dev)
        self.scratch = torch.zeros(G * cap // 64 + 32, dtype=f32, device=dev)   # [0] = arrival ticket, kept zero


class ShardedFocfEngine:
    def __init__(self, user_shard: torch.Tensor, item_shard: torch.Tensor, objective: str, fair_weight: float,
                 lr: float, weight_decay: float, group=None, capacity_factor: float = 2.0, ops=None,
                 sweep_period: Optional[int] = None, strict_overflow: bool = True):
        # strict_overflow: never apply a step in which an exchange bucket overflowed (`_settle_capacity`)
        self.strict_overflow = strict_overflow
        self.group = group
        self.G = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = user_shard.device
        self.ops = ops or HipOps(self.device)
        self.objective, self.fair_weight = objective, float(fair_weight)
        self.U = self.ops.make_table(user_shard)
        self.I = self.ops.make_table(item_shard)
        self.hyper = AdamHyper(lr, weight_decay, device=self.device)
        self.capacity_factor = capacity_factor
        self.sweep_period = sweep_period
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._buf: Optional[_Buffers] = None
        self._armed = False
        self._sel = 0                 # which copy of the index-side state the current step uses
        self._prep_key = None         # identity of the batch whose index work is in flight on the side stream
        self.step_count = 0

    # --- collectives ------------------------------------------------------------------------------------
    def _a2a(self, out: torch.Tensor, t: torch.Tensor, group=None):
        """Chunk g of `t` goes to rank g; chunk g of `out` is what rank g sent to me."""
        dist.all_to_all_single(out, t, group=group if group is not None else self.group)

    def capacity(self, B: int) -> int:
        return exchange_capacity(B, self.G, self.capacity_factor)

    def _buffers(self, B: int) -> _Buffers:
        if self._buf is None or self._buf.B != B:
            self._buf = _Buffers(self.G, B, self.capacity(B), self.U.dim, self.device, self.ops, self.U, self.I)
            self.ops.used_on_side(self._buf)
            self._prep_key = None
        return self._buf

    # --- step -------------------------------------------------------------------------------------------
    @staticmethod
    def _key(user, item):
        return (user.data_ptr(), item.data_ptr(), user.numel())

    # The index-only part of a step (a pure function of the id columns), in two halves so that the look-ahead can
    # place its collective where the main chain leaves the communicator idle:
    def _prepare_bucket(self, b: _Buffers, sel: int, user, item, sst):
        """bucket both id lists by owner (+ the local extrema of the sensitive column)"""
        G, cap, S, fair = self.G, b.cap, b.S, self.objective != "none"
        self.ops.bucket_pair(user, item, G, cap, S, 0, cap, b.ids_send[sel], b.slot_u[sel], b.slot_i[sel], b.counts[sel],
                             sst if fair else None, 2 * cap, self.err)

    def _prepare_exchange(self, b: _Buffers, sel: int):
        """exchange the id lists and sort what this rank received into segments of equal rows"""
        G, cap, S = self.G, b.cap, b.S
        self._a2a(b.ids_recv[sel], b.ids_send[sel])
        self.ops.sort_pair(self.U, self.I, b.ids_recv[sel], 0, cap, G * cap, cap, S, b.ws_u[sel], b.ws_i[sel], self.err)

    def _settle_capacity(self, b: _Buffers, sel: int, user, item, sst, bucketed: bool):
        """NEVER DROP AN INTERACTION.  fr_bucket_by_owner gives an interaction whose owner's bucket is full slot -1 and sets
        a device error bit; left at that, the step would apply the rest of the batch and the error would surface at the
        epoch's check.  Here, before anything of the step is exchanged for good: did ANY rank overflow ANY bucket (one MAX
        all-reduce of a flag + one host read per step -- the price of the guarantee; skipped inside a stream capture, where
        a host read has no place and the caller vouches for its batches)?  Then the exchange capacity is doubled and the
        batch bucketed again (item-complete batches put whole item histories on few owners: the default factor 2 is for
        uniform ids); when the capacity is already what one owner can sort per launch (FR_SORT_MAX // G), the step is
        REFUSED with nothing applied on any rank.  Returns the buffers to use and whether the id exchange is still to do."""
        B = user.numel()
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if not self.strict_overflow or capturing:
            return b, not bucketed
        redo = not bucketed
        for _ in range(8):
            if redo:
                self._prepare_bucket(b, sel, user, item, sst)
            flag = ((b.slot_u[sel] < 0).any() | (b.slot_i[sel] < 0).any()).to(torch.int32).reshape(1)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            if not bool(flag.item()):
                return b, redo
            self.ops.join_side()                    # whatever ran ahead for this batch used the too-small buffers
            self._prep_key = None
            self.err.bitwise_and_(~_C.DEV_ERR_BUCKET_OVERFLOW)
            old = b.cap
            if capacity_is_sort_bound(B, self.G, self.capacity_factor) or old >= B:
                raise _C.FairrecError(
                    f"an exchange bucket overflowed (skewed ids) and its capacity {old} is already the most one owner can sort "
                    f"per launch (FR_SORT_MAX // G = {SORT_MAX // self.G}): lower the per-rank batch size.  Nothing of this "
                    "step was applied on any rank")
            self.capacity_factor *= 2.0
            self._buf = None
            b = self._buffers(B)
            if b.cap <= old:
                raise _C.FairrecError(f"an exchange bucket overflowed and the capacity cannot grow beyond {old}: lower the "
                                      "per-rank batch size.  Nothing of this step was applied on any rank")
            redo = True
        raise _C.FairrecError("exchange capacity did not settle")

    def forward(self, user, item, rating, sst, next_batch=None):
        """Everything up to the loss of the global batch and the gradient rows; returns (loss as a 0-dim device
        tensor, pred [B]).  next_batch = (user, item, sst) of the FOLLOWING step, when known (dataloader look-ahead):
        its index work is started on a side stream now and overlaps with this step."""
        G, ops = self.G, self.ops
        B = user.numel()
        self._last_B = B
        b = self._buffers(B)
        sel = self._sel
        fair = self.objective not in ("none", "nonparity")     # per-item statistics on the owners (rec / reply exchange)
        nonparity = self.objective == "nonparity"
        if self._prep_key == self._key(user, item):
            ops.join_side()
            b2, todo = self._settle_capacity(b, sel, user, item, sst, bucketed=True)
            if b2 is not b or todo:             # the look-ahead's exchange used buckets that overflowed: again, larger
                b = b2
                self._prepare_exchange(b, sel)
        else:
            if self._prep_key is not None:      # a look-ahead for some other batch is in flight: let it finish first
                ops.join_side()
            b, _ = self._settle_capacity(b, sel, user, item, sst, bucketed=False)
            if not self.strict_overflow or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
                self._prepare_bucket(b, sel, user, item, sst)
            self._prepare_exchange(b, sel)
        self._prep_key = None
        cap, S = b.cap, b.S
        n_slots = G * cap
        ahead = next_batch is not None and next_batch[0].numel() == B
        if ahead:   # the next step's bucket kernel starts now, beside this step's gather
            ops.used_on_side(next_batch)
            with ops.side():
                self._prepare_bucket(b, sel ^ 1, next_batch[0], next_batch[1], next_batch[2])
            self._prep_key = self._key(next_batch[0], next_batch[1])

        def exchange_ahead():
            # One communicator serves both streams, so collectives run in issue order: the next step's id exchange is
            # issued where the main chain leaves the communicator idle (while the fairness / score kernel runs).
            if ahead:
                with ops.side(fork=False):
                    self._prepare_exchange(b, sel ^ 1)
        slot_u, slot_i, ids_recv = b.slot_u[sel], b.slot_i[sel], b.ids_recv[sel]
        ops.gather_train_pair(self.U, self.I, self.hyper, ids_recv, 0, cap, n_slots, cap, S, b.rows_send, b.ws_u[sel],
                              b.ws_i[sel], self.err)
        self._a2a(b.rows_recv, b.rows_send)
        ops.shard_score(b.rows_recv, slot_u, slot_i, rating, sst if fair else None, G * B, b.pred, b.coef,
                        b.rec_send if fair else None, cap, S, cap, None if (fair or nonparity) else b.sq, b.sq_part)
        if fair:
            self._a2a(b.rec_recv, b.rec_send)
            exchange_ahead()
            ops.shard_fair(self.I, n_slots, b.rec_recv, cap, ids_recv, 2 * cap, S, self.objective, self.fair_weight,
                           b.reply_send, b.sq_part, b.n_sq_part, b.scratch, self.err)
            self._a2a(b.reply_recv, b.reply_send)
        elif nonparity:
            # two group means of pred over the global batch: one 5-float all-reduce, no item exchange
            ops.nonparity_sums(b.pred, sst, ids_recv, 2 * cap, S, G, b.sq_part, b.n_sq_part, b.np5)
            dist.all_reduce(b.np5, op=dist.ReduceOp.SUM, group=self.group)
            exchange_ahead()
            ops.nonparity_coef(b.coef, sst, ids_recv, 2 * cap, S, G, b.np5, G * B, self.fair_weight, b.loss, self.err)
        else:
            dist.all_reduce(b.sq, op=dist.ReduceOp.SUM, group=self.group)
            exchange_ahead()
        # gradient rows for the owners; with a fairness term this kernel also folds the reply tails into the loss
        ops.shard_grads(b.rows_recv, slot_u, slot_i, b.coef, b.reply_recv if fair else None, G, G * B,
                        self.fair_weight, b.loss if fair else None, cap, S, cap, b.g_send)
        loss = b.loss[0] if (fair or nonparity) else b.sq[0] / float(G * B)
        self._armed = True
        return loss, b.pred

    def backward_adam(self):
        G, ops, b = self.G, self.ops, self._buf
        if not self._armed:
            raise _C.FairrecError("backward_adam without forward")
        cap, S = b.cap, b.S
        n_slots = G * cap
        self._a2a(b.g_recv, b.g_send)
        su = self.sweep_period if self.sweep_period is not None else max(8, math.ceil(self.U.n_rows / max(n_slots // 2, 1)))
        si = self.sweep_period if self.sweep_period is not None else max(8, math.ceil(self.I.n_rows / max(n_slots // 2, 1)))
        ops.apply_grad_pair(self.U, self.I, self.hyper, n_slots, cap, S, b.rows_send, b.g_recv, 0, cap, su, si)
        self._armed = False
        self._sel ^= 1
        self.step_count += 1

    def flush(self):
        self.ops.flush(self.U, self.hyper)
        self.ops.flush(self.I, self.hyper)

    def check_device_errors(self):
        e = int(self.err.item())
        if e:
            self.err.zero_()
            if e & _C.DEV_ERR_BUCKET_OVERFLOW:
                B = self._last_B if hasattr(self, "_last_B") else 0
                if B and capacity_is_sort_bound(B, self.G, self.capacity_factor):
                    raise _C.FairrecError(
                        f"an exchange bucket overflowed (skewed ids) and its capacity {self.capacity(B)} is already the most "
                        f"one owner can sort per launch (FR_SORT_MAX // G = {SORT_MAX // self.G}): lower the per-rank "
                        "batch size; the overflowing interactions of that step were NOT applied")
                raise _C.FairrecError("an exchange bucket overflowed (skewed ids): raise capacity_factor; the "
                                      "overflowing interactions of that step were NOT applied")
            raise IndexError(f"device error word {e}")


# ======================================================================================================================
# Item-owner-computes schedule (round 2): every interaction is routed to the rank that owns its ITEM row.  That rank has
# all interactions of its items, so the per-item fairness statistics, the scores and both gradients are formed where the
# item rows live; only the USER rows travel (owner -> item owner) and only the user gradients travel back.  The step's
# dependent chain holds 2 all-to-alls (user rows, user gradient rows) instead of 4; the index-only exchanges (records to
# the item owners, user-id requests to the user owners, the owners' distinct-item counts and sst extrema) depend on nothing
# but the id columns and run one step ahead on the side stream, like the id exchange of the schedule above.
#
#     recs   [G, 4*cap]       int64 planes: item local row | user id | rating bits | sst bits      -> item owners
#     ureq   [G, cap]         user local rows an item owner asks each user owner for                -> user owners
#     meta   [G, 3] floats    all-gather: (distinct items held, min sst, max sst) of every rank
#     urows  [G, cap, D]      user rows (lazy gather at the owner)                                  -> item owners
#     ugrad  [G, cap, D]      user gradient rows                                                    -> user owners
#     2 floats all-reduce     (fairness sum, squared-error sum) for the reported loss, off the chain
#
# Objectives none / value / absolute / under / over; non-parity (two global means before any gradient) stays on the
# schedule above.  Same kernels (`ops`), other wiring: an item owner treats the G*cap record slots it received as a batch
# whose item rows are local (slot = record position) and whose user rows sit in the row buffer that came back.
class ShardedFocfEngineV2(ShardedFocfEngine):
    class _Buf:
        def __init__(self, G, B, cap, D, dev, ops, U, I):
            f32, i64, i32 = torch.float32, torch.int64, torch.int32
            n = G * cap
            self.B, self.cap, self.n, self.RS = B, cap, n, 4 * cap + 1
            two = range(2)
            # index-side state exists twice: the NEXT step's exchanges run while this step computes
            self.rec_send = [torch.full((G * self.RS,), -1, dtype=i64, device=dev) for _ in two]
            self.rec_recv = [torch.full((G * self.RS,), -1, dtype=i64, device=dev) for _ in two]
            self.slot_i = [torch.empty(B, dtype=i32, device=dev) for _ in two]
            self.cnt_i = [torch.empty(G, dtype=i32, device=dev) for _ in two]
            self.ids2 = [torch.full((2 * n,), -1, dtype=i64, device=dev) for _ in two]   # [user-row requests received | item rows held]
            self.uid = [torch.empty(n, dtype=i64, device=dev) for _ in two]      # user ids of the records held (-1: empty slot)
            self.ureq_send = [torch.full((n,), -1, dtype=i64, device=dev) for _ in two]
            self.uslot = [torch.empty(n, dtype=i32, device=dev) for _ in two]
            self.cnt_u = [torch.empty(G, dtype=i32, device=dev) for _ in two]
            self.islot = [torch.empty(n, dtype=i32, device=dev) for _ in two]
            self.rating = [torch.zeros(n, dtype=f32, device=dev) for _ in two]
            self.sst = [torch.zeros(n, dtype=f32, device=dev) for _ in two]
            self.k_send = [torch.zeros(1, dtype=f32, device=dev) for _ in two]
            self.k_all = [torch.zeros(G, dtype=f32, device=dev) for _ in two]    # distinct items held by every owner
            self.mm = [torch.zeros(G, dtype=i64, device=dev) for _ in two]       # (min, max) float pairs of every rank's sst
            self.ws_u = [ops.alloc_ws(U, n) for _ in two]
            self.ws_i = [ops.alloc_ws(I, n) for _ in two]
            self.rows = torch.zeros((2 * n, D), dtype=f32, device=dev)           # [user rows gathered for others | item rows held]
            self.urows = torch.empty((n, D), dtype=f32, device=dev)
            self.grads = torch.zeros((2 * n, D), dtype=f32, device=dev)          # [user gradients received | item gradients]
            self.g_send = torch.zeros((n, D), dtype=f32, device=dev)
            self.rec = torch.zeros(G * 3 * cap, dtype=f32, device=dev)
            self.reply = torch.zeros(G * (cap + TAIL), dtype=f32, device=dev)
            self.pred = torch.empty(n, dtype=f32, device=dev)
            self.coef = torch.empty(n, dtype=f32, device=dev)
            self.sums = torch.zeros(2, dtype=f32, device=dev)
            self.loss = torch.zeros(3, dtype=f32, device=dev)
            self.n_sq_part = (n + 3) // 4
            self.sq_part = torch.zeros(self.n_sq_part + 1, dtype=f32, device=dev)
            self.scratch = torch.zeros(n // 64 + 32, dtype=f32, device=dev)
            self.bitmap = torch.zeros((I.n_rows + 31) // 32, dtype=i32, device=dev)     # all zero between uses
            self.count = torch.zeros(1, dtype=i32, device=dev)

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        if self.objective == "nonparity":
            raise NotImplementedError("ShardedFocfEngineV2: fair_objective nonparity runs on ShardedFocfEngine")
        self.defer_loss = False      # True: the loss is all-reduced at the end of backward_adam instead of inside forward

    def _buffers(self, B):
        if self._buf is None or self._buf.B != B:
            self._buf = self._Buf(self.G, B, self.capacity(B), self.U.dim, self.device, self.ops, self.U, self.I)
            self.ops.used_on_side(self._buf)
            self._prep_key = None
        return self._buf

    # --- index-only part, a pure function of the id columns (+ rating / sst payload): 5 launches, 3 collectives ---------
    def _index_a(self, b, sel, user, item, rating, sst):
        """bucket the interactions by item owner and pack the records"""
        G, cap, fair = self.G, b.cap, self.objective != "none"
        self.ops.bucket_by_owner(item, G, cap, b.RS, 0, b.rec_send[sel], b.slot_i[sel], b.cnt_i[sel], sst if fair else None,
                                 4 * cap, self.err)
        self.ops.pack_records(b.slot_i[sel], user, rating, sst if fair else None, cap, b.rec_send[sel])

    def _index_b(self, b, sel):
        """records to the item owners; their user-row requests to the user owners; the owners' sorts and item counts"""
        G, cap, n, ops = self.G, b.cap, b.n, self.ops
        self._a2a(b.rec_recv[sel], b.rec_send[sel])
        iid = b.ids2[sel][n:]
        ops.unpack_records(b.rec_recv[sel], G, cap, iid, b.uid[sel], b.islot[sel], b.rating[sel], b.sst[sel], b.mm[sel])
        if self.objective != "none":
            # how many distinct items this owner holds, to every rank -- before the slow part (the owner sorts): every later
            # collective of either stream queues behind this one
            ops.count_distinct(iid, self.I.n_rows, b.bitmap, b.count, b.k_send[sel])
            dist.all_gather_into_tensor(b.k_all[sel], b.k_send[sel], group=self.group)
        ops.bucket_sparse(b.uid[sel], G, cap, cap, 0, b.ureq_send[sel], b.uslot[sel], b.cnt_u[sel], self.err)
        self._a2a(b.ids2[sel][:n], b.ureq_send[sel])
        ops.sort_pair(self.U, self.I, b.ids2[sel], 0, n, n, 0, 0, b.ws_u[sel], b.ws_i[sel], self.err)

    def forward(self, user, item, rating, sst, next_batch=None):
        """loss of the global batch (0-dim device tensor; with `defer_loss` complete after backward_adam) and None (the
        scores live with the item owners).  next_batch = (user, item, sst, rating) of the FOLLOWING step: its index work
        and exchanges start on the side stream now."""
        G, ops = self.G, self.ops
        B = user.numel()
        self._last_B = B
        b = self._buffers(B)
        cap, n, sel = b.cap, b.n, self._sel
        fair = self.objective != "none"
        if self._prep_key == self._key(user, item):
            ops.join_side()
        else:
            if self._prep_key is not None:
                ops.join_side()
            self._index_a(b, sel, user, item, rating, sst)
            self._index_b(b, sel)
        self._prep_key = None
        ahead = next_batch is not None and len(next_batch) >= 4 and next_batch[0].numel() == B
        if ahead:
            ops.used_on_side(next_batch)
            with ops.side():
                self._index_a(b, sel ^ 1, next_batch[0], next_batch[1], next_batch[3], next_batch[2])
            self._prep_key = self._key(next_batch[0], next_batch[1])
        # user owners gather the requested rows, item owners their own rows: one launch
        ops.gather_train_pair(self.U, self.I, self.hyper, b.ids2[sel], 0, n, n, 0, 0, b.rows, b.ws_u[sel], b.ws_i[sel], self.err)
        self._a2a(b.urows, b.rows[:n])
        if ahead:       # the next step's exchanges where the chain leaves the communicator idle
            with ops.side(fork=False):
                self._index_b(b, sel ^ 1)
        ops.shard_score2(b.urows, b.rows[n:], b.uslot[sel], b.islot[sel], b.rating[sel], b.sst[sel] if fair else None,
                         G * B, b.pred, b.coef, b.rec if fair else None, cap, cap, 0, None if fair else b.sums[1:2], b.sq_part)
        if fair:
            ops.shard_fair(self.I, n, b.rec, cap, b.mm[sel], 0, 1, self.objective, self.fair_weight, b.reply, b.sq_part,
                           b.n_sq_part, b.scratch, self.err)
            ops.post_fair(b.reply, b.k_all[sel], G, cap, b.sums)      # sums <- (fair, sq) of this owner; tails <- every owner's K
        ops.shard_grads2(b.urows, b.rows[n:], b.uslot[sel], b.islot[sel], b.coef, b.reply if fair else None, G, G * B,
                         self.fair_weight, None, cap, cap, 0, b.g_send, b.grads[n:])
        self._loss_sel = sel
        if not self.defer_loss:
            self._finish_loss(b, B)
        self._armed = True
        return b.loss[0], None

    def _finish_loss(self, b, B):
        """the reported loss: two sums over the ranks (off the dependent chain when deferred)"""
        dist.all_reduce(b.sums, op=dist.ReduceOp.SUM, group=self.group)
        self.ops.loss_finish(b.sums, b.k_all[self._loss_sel], self.G, self.G * B, self.fair_weight, self.objective != "none",
                             b.loss)

    def backward_adam(self):
        G, ops, b = self.G, self.ops, self._buf
        if not self._armed:
            raise _C.FairrecError("backward_adam without forward")
        n = b.n
        self._a2a(b.grads[:n], b.g_send)
        su = self.sweep_period if self.sweep_period is not None else max(8, math.ceil(self.U.n_rows / max(n // 2, 1)))
        si = self.sweep_period if self.sweep_period is not None else max(8, math.ceil(self.I.n_rows / max(n // 2, 1)))
        ops.apply_grad_pair(self.U, self.I, self.hyper, n, 0, 0, b.rows, b.grads, 0, n, su, si)
        if self.defer_loss:
            self._finish_loss(b, self._last_B)
        self._armed = False
        self._sel ^= 1
        self.step_count += 1
