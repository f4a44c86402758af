"""Row-sharded FOCF training step for G GPUs of one node (one process per GPU, RCCL over xGMI through
torch.distributed).

Semantics: ONE optimizer step on the global batch = concatenation of every rank's local batch (rank order), i.e.
exactly what the single-device reference computes on that batch: loss = MSE over G*B interactions +
fair_weight * mean over the distinct items of the GLOBAL batch.  Tables are split by `row mod G`; Adam state
lives with its rows.  Per step (no host sync, fixed-capacity [G, cap] buffers):

    bucket ids by owner -> all-to-all(ids) -> owners: lazy gather -> all-to-all(rows) -> score
    -> all-to-all(records) -> owners: per-item fairness statistics -> all-reduce(3 scalars)
    -> all-to-all(fairness coefficients) -> gradient rows -> all-to-all(grads) -> owners: duplicate-sum + Adam

The kernels come from an `ops` object (default: the HIP library through fairrec._C).  Tests inject a CPU double
to exercise this exchange schedule over gloo without a GPU; the product path is HIP only.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch
import torch.distributed as dist

from . import _C
from .optim import AdamHyper, LazyTable


def shard_rows(n_rows: int, rank: int, world: int) -> int:
    """Number of rows r < n_rows with r mod world == rank."""
    return (n_rows - rank + world - 1) // world if n_rows > rank else 0


def shard_of(full: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """The rows of a full table this rank owns, in local-row order (row = local * world + rank)."""
    return full[rank::world].contiguous()


class HipOps:
    """The kernels of the sharded step, bound to the HIP library."""

    def __init__(self, device):
        self.device = torch.device(device)
        _C.lib()

    def make_table(self, weight):
        return LazyTable(weight)

    def bucket_by_owner(self, idx, G, cap, err):
        M = idx.numel()
        send = torch.empty(G * cap, dtype=torch.int64, device=idx.device)
        slot = torch.empty(M, dtype=torch.int32, device=idx.device)
        counts = torch.empty(G, dtype=torch.int32, device=idx.device)
        _C.check(_C.lib().fr_bucket_by_owner(idx.data_ptr(), M, G, cap, send.data_ptr(), slot.data_ptr(),
                                             counts.data_ptr(), err.data_ptr(), _C.current_stream()), "fr_bucket_by_owner")
        return send, slot, counts

    def gather_train(self, table, hyper, ids, err):
        return table.gather_train(hyper, ids, err)

    def apply_grad(self, table, hyper, grads, sweep):
        table.apply_grad(hyper, grads, sweep)

    def flush(self, table, hyper):
        table.flush(hyper)

    def shard_score(self, rows_u, rows_i, slot_u, slot_i, rating, sst, n_global, want_rec):
        B, D = slot_u.numel(), rows_u.shape[1]
        n_slots = rows_i.shape[0]
        dev = rows_u.device
        pred = torch.empty(B, dtype=torch.float32, device=dev)
        coef = torch.empty(B, dtype=torch.float32, device=dev)
        rec = torch.zeros((3, n_slots), dtype=torch.float32, device=dev) if want_rec else None
        sq = torch.empty(1, dtype=torch.float32, device=dev)
        scratch = torch.empty((B + 3) // 4 + 1, dtype=torch.float32, device=dev)
        _C.check(_C.lib().fr_focf_shard_score(rows_u.data_ptr(), rows_i.data_ptr(), slot_u.data_ptr(), slot_i.data_ptr(),
                                              rating.data_ptr(), _C.ptr(sst), B, D, n_global, pred.data_ptr(),
                                              coef.data_ptr(), _C.ptr(rec), n_slots, sq.data_ptr(), scratch.data_ptr(),
                                              _C.current_stream()), "fr_focf_shard_score")
        return pred, coef, rec, sq

    def shard_fair(self, item_table, rec, minmax, objective, fair_weight, err):
        n_slots = rec.shape[1]
        dev = rec.device
        coef_slots = torch.empty(n_slots, dtype=torch.float32, device=dev)
        sums = torch.zeros(2, dtype=torch.float32, device=dev)
        scratch = torch.empty(n_slots // 16 + 2, dtype=torch.float32, device=dev)
        ws = item_table._ws
        _C.check(_C.lib().fr_focf_shard_fair(ws.data_ptr(), ws.numel(), n_slots, item_table.dim, rec.data_ptr(),
                                             minmax.data_ptr(), _C.FOCF_OBJECTIVES[objective], fair_weight,
                                             coef_slots.data_ptr(), sums.data_ptr(), scratch.data_ptr(),
                                             err.data_ptr(), _C.current_stream()), "fr_focf_shard_fair")
        return coef_slots, sums

    def shard_grads(self, rows_u, rows_i, slot_u, slot_i, coef, coef_slots, inv_k):
        B, D = slot_u.numel(), rows_u.shape[1]
        gu = torch.empty_like(rows_u)
        gi = torch.empty_like(rows_i)
        _C.check(_C.lib().fr_focf_shard_grads(rows_u.data_ptr(), rows_i.data_ptr(), slot_u.data_ptr(), slot_i.data_ptr(),
                                              coef.data_ptr(), _C.ptr(coef_slots), _C.ptr(inv_k), B, D, gu.data_ptr(),
                                              gi.data_ptr(), _C.current_stream()), "fr_focf_shard_grads")
        return gu, gi


class ShardedFocfEngine:
    def __init__(self, user_shard: torch.Tensor, item_shard: torch.Tensor, objective: str, fair_weight: float,
                 lr: float, weight_decay: float, group=None, capacity_factor: float = 2.0, ops=None,
                 sweep_period: Optional[int] = None):
        self.group = group
        self.G = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = user_shard.device
        self.ops = ops or HipOps(self.device)
        if objective == "nonparity":
            raise NotImplementedError("nonparity needs two global group means; not on the sharded path yet")
        self.objective, self.fair_weight = objective, float(fair_weight)
        self.U = self.ops.make_table(user_shard)
        self.I = self.ops.make_table(item_shard)
        self.hyper = AdamHyper(lr, weight_decay, device=self.device)
        self.capacity_factor = capacity_factor
        self.sweep_period = sweep_period
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._ctx = None
        self.step_count = 0

    # --- collectives ------------------------------------------------------------------------------------
    def _a2a(self, t: torch.Tensor) -> torch.Tensor:
        """t[g] goes to rank g; returns r with r[g] = what rank g sent to me.  t is [G, ...] contiguous."""
        out = torch.empty_like(t)
        dist.all_to_all_single(out, t, group=self.group)
        return out

    def capacity(self, B: int) -> int:
        return min(B, int(math.ceil(self.capacity_factor * B / self.G)) + 64)

    # --- step -------------------------------------------------------------------------------------------
    def forward(self, user, item, rating, sst):
        """Everything up to the loss of the global batch; returns the loss as a 0-dim device tensor."""
        G, ops = self.G, self.ops
        B = user.numel()
        cap = self.capacity(B)
        n_slots = G * cap
        fair = self.objective != "none"
        send_u, slot_u, _ = ops.bucket_by_owner(user, G, cap, self.err)
        send_i, slot_i, _ = ops.bucket_by_owner(item, G, cap, self.err)
        ids = self._a2a(torch.cat([send_u.view(G, cap), send_i.view(G, cap)], dim=1).contiguous())
        req_u = ids[:, :cap].reshape(-1).contiguous()
        req_i = ids[:, cap:].reshape(-1).contiguous()
        minmax = None
        if fair:
            mm = torch.stack([sst.min(), -sst.max()])
            dist.all_reduce(mm, op=dist.ReduceOp.MIN, group=self.group)
            minmax = torch.stack([mm[0], -mm[1]])
        own_u = ops.gather_train(self.U, self.hyper, req_u, self.err)      # [G*cap, D], rows I own, per requester
        own_i = ops.gather_train(self.I, self.hyper, req_i, self.err)
        D = own_u.shape[1]
        rows = self._a2a(torch.cat([own_u.view(G, cap, D), own_i.view(G, cap, D)], dim=1).contiguous())
        rows_u = rows[:, :cap].reshape(n_slots, D).contiguous()            # my requests, slot order
        rows_i = rows[:, cap:].reshape(n_slots, D).contiguous()
        pred, coef, rec, sq = ops.shard_score(rows_u, rows_i, slot_u, slot_i, rating, sst, G * B, fair)
        scal = torch.zeros(3, dtype=torch.float32, device=self.device)
        scal[0:1] = sq
        coef_own = None
        if fair:
            rec_in = self._a2a(rec.view(3, G, cap).permute(1, 0, 2).contiguous())        # [G(src), 3, cap]
            rec_own = rec_in.permute(1, 0, 2).reshape(3, n_slots).contiguous()
            coef_own, sums = ops.shard_fair(self.I, rec_own, minmax, self.objective, self.fair_weight, self.err)
            scal[1:3] = sums
        dist.all_reduce(scal, op=dist.ReduceOp.SUM, group=self.group)
        mse = scal[0] / float(G * B)
        loss = mse + self.fair_weight * scal[1] / scal[2] if fair else mse
        self._ctx = dict(B=B, cap=cap, slot_u=slot_u, slot_i=slot_i, rows_u=rows_u, rows_i=rows_i, coef=coef,
                         coef_own=coef_own, inv_k=(1.0 / scal[2]).reshape(1) if fair else None, own_u=own_u, own_i=own_i)
        return loss, pred

    def backward_adam(self):
        c, G, ops = self._ctx, self.G, self.ops
        if c is None:
            raise _C.FairrecError("backward_adam without forward")
        cap, n_slots = c["cap"], G * c["cap"]
        coef_reply = None
        if c["coef_own"] is not None:
            coef_reply = self._a2a(c["coef_own"].view(G, cap).contiguous()).reshape(-1).contiguous()
        gu, gi = ops.shard_grads(c["rows_u"], c["rows_i"], c["slot_u"], c["slot_i"], c["coef"], coef_reply, c["inv_k"])
        D = gu.shape[1]
        g = self._a2a(torch.cat([gu.view(G, cap, D), gi.view(G, cap, D)], dim=1).contiguous())
        gu_own = g[:, :cap].reshape(n_slots, D).contiguous()
        gi_own = g[:, cap:].reshape(n_slots, D).contiguous()
        su = self.sweep_period if self.sweep_period is not None else max(8, math.ceil(self.U.n_rows / max(n_slots // 2, 1)))
        si = self.sweep_period if self.sweep_period is not None else max(8, math.ceil(self.I.n_rows / max(n_slots // 2, 1)))
        ops.apply_grad(self.U, self.hyper, gu_own, su)
        ops.apply_grad(self.I, self.hyper, gi_own, si)
        self._ctx = None
        self.step_count += 1

    def flush(self):
        self.ops.flush(self.U, self.hyper)
        self.ops.flush(self.I, self.hyper)

    def check_device_errors(self):
        e = int(self.err.item())
        if e:
            self.err.zero_()
            if e & _C.DEV_ERR_BUCKET_OVERFLOW:
                raise _C.FairrecError("an exchange bucket overflowed: raise capacity_factor (skewed ids)")
            raise IndexError(f"device error word {e}")
