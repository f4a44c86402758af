"""Row-sharded embedding tables for every model that trains through `GenericEngine` (NFCF, PFCN_*, FairGo_* pretrain):
SURVEY.md §8-e items 1-3 and 5.  One process per GPU; table row r lives on rank `r mod G` as local row `r div G`, Adam
state with it; the small dense parameters (MLPs, biases) are replicated and their gradients all-reduced.

    lookup(name, idx):   bucket ids by owner -> all-to-all(ids) -> owners: lazy gather (fr_table_gather_train) ->
                         all-to-all(rows) -> back to batch order              [differentiable: torch.autograd.Function]
    backward:            dLoss/drows -> slot order (scaled 1/G) -> all-to-all -> parked on the owner
    optimizer.step():    owners: duplicate-sum + Adam + sweeper (fr_table_apply_grad); dense: one flat all-reduce, then
                         fr_adam_dense on every replica (identical inputs => replicas stay bit-identical)

Semantics: every rank feeds its own batch; a step is ONE optimizer step whose embedding and dense gradients are those
of the mean loss over the G local losses, i.e. the single-device step on the concatenated batch for every loss that is
a mean over interactions (BCE, MSE, BPR).  NFCF's differential-fairness regulariser is a statistic OF the batch (per-item,
per-group score sums, the number of items K): `global_item_df` evaluates it on the GLOBAL batch -- every positive row's
(score, group) goes to the owner of its item in the slot its id took in the lookup, the owner's per-(item, group) sums
come back, K and the groups present ride in the tails: two small all-to-alls, and a G-rank step equals the single-device
step on the concatenated batch (nfcf.py:76-97).  PFCN_BiasedMF's [B] + [B, 1] -> [B, B] broadcast loss runs over the global
batch as well (`global_bpr_broadcast`: the term matrix factors into row and column sums, one all-gather of two [B] columns).
Only BatchNorm inside the PFCN MLPs is evaluated per rank on the local batch ("parity is defined
per-GPU-batch"); FOCF, whose fairness term needs the global per-item statistics, has its own exact engine
(fairrec/sharded.py).

Kernels come from an `ops` object (default: HIP through fairrec._C); tests inject a CPU double to run the schedule
over gloo.  The product path is HIP only.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import _C
from .engine import GenericEngine
from .sharded import exchange_capacity


class HipTableOps:
    """Per-table kernels of the sharded lookup (dense [G, cap] exchange buffers)."""

    def bucket(self, idx, G, cap, send, slot, counts, err):
        _C.check(_C.lib().fr_bucket_by_owner(idx.data_ptr(), idx.numel(), G, cap, cap, 0, send.data_ptr(), slot.data_ptr(),
                                             counts.data_ptr(), None, 0, err.data_ptr(), _C.current_stream()),
                 "fr_bucket_by_owner")

    def gather_train(self, table, hyper, ids, M, rows, err):
        table.gather_train_into(hyper, ids.data_ptr(), M, rows.data_ptr(), 0, 0, err)

    # the same on one table's part of a PACKED exchange buffer (slot layout of fairrec_hip.h: chunk g of the buffer holds
    # `chunk` slots of this table at element offset `off`, chunks are `stride` slots apart)
    def bucket_at(self, idx, G, cap, stride, off, send, slot, counts, err):
        _C.check(_C.lib().fr_bucket_by_owner(idx.data_ptr(), idx.numel(), G, cap, stride, off, send.data_ptr(),
                                             slot.data_ptr(), counts.data_ptr(), None, 0, err.data_ptr(),
                                             _C.current_stream()), "fr_bucket_by_owner")

    def gather_train_at(self, table, hyper, ids, off, M, chunk, stride, rows, err):
        table.gather_train_into(hyper, ids.data_ptr() + 8 * off, M, rows.data_ptr() + 4 * off * table.dim, chunk, stride, err)

    def apply_grad_at(self, table, hyper, M, rows, grads, off, chunk, stride, sweep):
        D = table.dim
        table.apply_grad_from(hyper, M, rows.data_ptr() + 4 * off * D, grads.data_ptr() + 4 * off * D, sweep, chunk, stride)

    def gather(self, table, hyper, ids, M, rows, err):
        t = table.c(table.step)
        _C.check(_C.lib().fr_table_gather(ctypes.byref(t), ctypes.byref(hyper.c()), ids.data_ptr(), M, rows.data_ptr(),
                                          err.data_ptr(), _C.current_stream()), "fr_table_gather")

    def unbucket_rows(self, src, slot, M, D, out):
        _C.check(_C.lib().fr_unbucket_rows(src.data_ptr(), slot.data_ptr(), M, D, out.data_ptr(), _C.current_stream()),
                 "fr_unbucket_rows")

    def bucket_rows(self, src, scale, slot, M, D, dst):
        _C.check(_C.lib().fr_bucket_rows(src.data_ptr(), _C.ptr(scale), slot.data_ptr(), M, D, dst.data_ptr(),
                                         _C.current_stream()), "fr_bucket_rows")

    def apply_grad(self, table, hyper, M, rows, grads, sweep):
        table.apply_grad_from(hyper, M, rows.data_ptr(), grads.data_ptr(), sweep)

    def adam_dense(self, p, g, m, v, hyper, step):
        _C.check(_C.lib().fr_adam_dense(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                        ctypes.byref(hyper.c()), step, _C.current_stream()), "fr_adam_dense")

    # --- differential fairness on the global batch (fr_nfcf_df_*: fairrec_hip.h) ---------------------------------------
    def df_pack(self, out, label, sst, slot, S, off, cap, G, rec, ws):
        _C.check(_C.lib().fr_nfcf_df_pack(out.data_ptr(), label.data_ptr(), sst.data_ptr(), slot.data_ptr(), S, off, cap,
                                          out.numel(), G, rec.data_ptr(), ws.data_ptr(), ws.numel(), _C.current_stream()),
                 "fr_nfcf_df_pack")

    def df_owner(self, table, rec, G, cap, reply, ws, B, err):
        _C.check(_C.lib().fr_nfcf_df_owner(table._ws.data_ptr(), table._ws.numel(), table.dim, rec.data_ptr(), G, cap,
                                           reply.data_ptr(), ws.data_ptr(), ws.numel(), B, err.data_ptr(),
                                           _C.current_stream()), "fr_nfcf_df_owner")

    def df_apply(self, reply, slot, S, off, cap, G, out, label, sst, fair_weight, scale, dy, loss, ws):
        _C.check(_C.lib().fr_nfcf_df_apply(reply.data_ptr(), slot.data_ptr(), S, off, cap, G, out.data_ptr(), label.data_ptr(),
                                           sst.data_ptr(), out.numel(), fair_weight, scale, dy.data_ptr(), loss.data_ptr(),
                                           ws.data_ptr(), ws.numel(), _C.current_stream()), "fr_nfcf_df_apply")

    def bpr_outer_rect(self, a, c, inv, loss, da, dc, ws_holder):
        """fr_bpr_outer_rect: rows c x columns a of PFCN_BiasedMF's broadcast loss matrix, scaled by inv."""
        lib = _C.lib()
        need = lib.fr_bpr_outer_rect_workspace_bytes(a.numel(), c.numel())
        ws = ws_holder.get("ws")
        if ws is None or ws.numel() < need:
            ws = ws_holder["ws"] = torch.empty(need, dtype=torch.uint8, device=a.device)
        _C.check(lib.fr_bpr_outer_rect(a.data_ptr(), a.numel(), c.data_ptr(), c.numel(), inv, loss.data_ptr(), _C.ptr(da),
                                       _C.ptr(dc), ws.data_ptr(), ws.numel(), _C.current_stream()), "fr_bpr_outer_rect")

    def df_workspace(self, B, n_slots, device):
        return torch.zeros(_C.lib().fr_nfcf_df_workspace_bytes(B, n_slots), dtype=torch.uint8, device=device)

    def sort_local(self, idx, n_rows, dim, holder):
        """Segments of a LOCAL id list in table-workspace layout (per-rank batch statistics, e.g. NFCF's DF term)."""
        M = idx.numel()
        need = _C.lib().fr_table_train_workspace_bytes(M, dim)
        for k in ("_ws", "_ws_b"):
            if getattr(holder, k, None) is None or getattr(holder, k).numel() < need:
                setattr(holder, k, torch.empty(need, dtype=torch.uint8, device=idx.device))
        _C.check(_C.lib().fr_table_sort2(idx.data_ptr(), idx.data_ptr(), n_rows, n_rows, M, 0, 0, dim, holder._ws.data_ptr(),
                                         holder._ws_b.data_ptr(), need, None, _C.current_stream()), "fr_table_sort2")


class _Exchange:
    """Buffers of one table's lookup for one batch size."""

    def __init__(self, G, M, cap, D, dev):
        self.M, self.cap = M, cap
        f32, i64, i32 = torch.float32, torch.int64, torch.int32
        self.ids_send = torch.empty(G * cap, dtype=i64, device=dev)
        self.ids_recv = torch.empty(G * cap, dtype=i64, device=dev)
        self.slot = torch.empty(M, dtype=i32, device=dev)
        self.counts = torch.empty(G, dtype=i32, device=dev)
        self.rows_send = torch.zeros((G * cap, D), dtype=f32, device=dev)
        self.rows_recv = torch.empty((G * cap, D), dtype=f32, device=dev)
        self.g_send = torch.zeros((G * cap, D), dtype=f32, device=dev)
        self.g_recv = torch.empty((G * cap, D), dtype=f32, device=dev)
        self.scale = torch.full((M,), 1.0 / G, dtype=f32, device=dev)


class _ExchangePair:
    """Buffers of a PACKED lookup in two tables of the same width for one batch size: one id / row / gradient buffer per
    direction, chunk g = [cap slots of table a | cap slots of table b] -> 3 all-to-alls per step instead of 6."""

    def __init__(self, G, M, cap, D, dev):
        self.M, self.cap, self.S = M, cap, 2 * cap
        f32, i64, i32 = torch.float32, torch.int64, torch.int32
        n = G * self.S
        self.ids_send = torch.full((n,), -1, dtype=i64, device=dev)
        self.ids_recv = torch.full((n,), -1, dtype=i64, device=dev)
        self.slot = [torch.empty(M, dtype=i32, device=dev) for _ in range(2)]
        self.counts = [torch.empty(G, dtype=i32, device=dev) for _ in range(2)]
        self.rows_send = torch.zeros((n, D), dtype=f32, device=dev)
        self.rows_recv = torch.empty((n, D), dtype=f32, device=dev)
        self.g_send = torch.zeros((n, D), dtype=f32, device=dev)
        self.g_recv = torch.empty((n, D), dtype=f32, device=dev)
        self.scale = torch.full((M,), 1.0 / G, dtype=f32, device=dev)


class _ShardedLookupPair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wa, wb, engine, name_a, name_b, idx_a, idx_b):
        ctx.engine, ctx.names = engine, (name_a, name_b)
        ra, rb = engine._pair_forward(name_a, idx_a, name_b, idx_b, train=True)
        return ra, rb

    @staticmethod
    def backward(ctx, ga, gb):
        ctx.engine._pair_backward(ctx.names, (ga, gb))
        return None, None, None, None, None, None, None


class _Segments:
    """What a loss kernel needs from "the table the batch ids were looked up in": sorted segments + dim."""

    def __init__(self, dim):
        self.dim = dim
        self._ws = None
        self._ws_b = None


class _ShardedLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, engine, name, idx):
        ctx.engine, ctx.name = engine, name
        return engine._exchange_forward(name, idx, train=True)

    @staticmethod
    def backward(ctx, grad_rows):
        ctx.engine._exchange_backward(ctx.name, grad_rows)
        return None, None, None, None


class ShardedGenericEngine(GenericEngine):
    def clip_grad_norm(self, max_norm: float, group=None):
        """GenericEngine.clip_grad_norm on row-sharded tables: the gradient of a table is held by the owners of its rows
        (the rows every requester sent back, duplicates summed per owned row in slot order: requester rank, then batch
        position), so the squared norms add over the ranks -- ONE all-reduce of a scalar; the dense gradients are measured
        after their flat all-reduce (done here instead of in backward_adam, which then finds them averaged).  Every rank
        computes the same total and the same coefficient."""
        G = self.G
        self._sync_dense(group)
        sq_tables = torch.zeros((), dtype=torch.float32, device=self.device)
        held = []
        for name, t in self._tables.items():
            if name.startswith(self.NOT_MODEL_PARAMETERS) or not t.trainable or t._grad_rows is None:
                continue
            lay = getattr(self, "_lay", {}).get(name)
            ex = lay[0] if lay is not None else self._ex[name]
            if lay is not None:      # packed exchange: chunk g = [cap slots of table a | cap slots of table b]
                off, cap = lay[1], ex.cap
                ids = ex.ids_recv.view(G, ex.S)[:, off:off + cap].reshape(-1)
                g = t._grad_rows.view(G, ex.S, t.dim)[:, off:off + cap].reshape(-1, t.dim)
            else:
                ids, g = ex.ids_recv, t._grad_rows
            valid = ids >= 0
            ids_v, g_v = ids[valid], g[valid]
            if ids_v.numel():
                order = torch.argsort(ids_v, stable=True)
                _, counts = torch.unique_consecutive(ids_v[order], return_counts=True)
                rows = torch.segment_reduce(g_v[order].contiguous(), "sum", lengths=counts, axis=0)
                sq_tables = sq_tables + (rows * rows).sum()
            held.append(t)
        dist.all_reduce(sq_tables, op=dist.ReduceOp.SUM, group=self.group)
        sq = sq_tables
        dense = []
        synced = self._synced[1] if self._synced is not None else {}
        for name, d in self._dense.items():
            if name.startswith(self.NOT_MODEL_PARAMETERS) or d.p.grad is None:
                continue
            gview = synced.get(name, d.p.grad)          # averaged over the ranks when this step's group owns it
            sq = sq + (gview * gview).sum()
            dense.append(gview)
        if not held and not dense:
            return torch.zeros((), device=self.device)
        total = torch.sqrt(sq)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        for t in held:
            t._grad_rows = t._grad_rows * coef
        for gview in dense:
            gview.mul_(coef)
        return total

    def _sync_dense(self, group=None):
        """One flat all-reduce (mean over the ranks) of the replicated dense gradients the stepping group owns
        (SURVEY.md §8-e item 5); kept until backward_adam consumes it."""
        live = [(name, d) for name, d in self._dense.items() if d.p.grad is not None and self._owned(name, group)]
        if not live:
            self._synced = None
            return
        n = sum(d.p.numel() for _, d in live)
        if self._flat is None or self._flat.numel() < n:
            self._flat = torch.empty(n, dtype=torch.float32, device=self.device)
        flat = self._flat[:n]
        torch.cat([d.p.grad.reshape(-1) for _, d in live], out=flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.G)
        views, off = {}, 0
        for name, d in live:
            k = d.p.numel()
            views[name] = flat[off:off + k]
            off += k
        self._synced = (live, views)

    def __init__(self, device, group=None, capacity_factor: float = 2.0, ops=None):
        self.group = group
        self.G = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.capacity_factor = capacity_factor
        self.ops = ops
        if ops is None:
            super().__init__(device)
            self.ops = HipTableOps()
        else:                                   # CPU double (tests): no device library behind it
            self._init_host_only(device)
        self._ex: Dict[str, _Exchange] = {}
        self._seg: Dict[str, _Segments] = {}
        self._n_rows_global: Dict[str, int] = {}
        self._flat: Optional[torch.Tensor] = None
        self._synced = None          # (live dense entries, their averaged-gradient views of _flat) between clip and step

    def _init_host_only(self, device):
        from .optim import AdamHyper
        self.device = torch.device(device)
        self._tables, self._weights, self._dense, self._group, self._hyper_of = {}, {}, {}, {}, {}
        self.hyper = AdamHyper(device=self.device, cap=1)
        self.optimizer, self.sweep_period = None, None
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._group_version = {}
        self._seg_src = {}

    # --- registration: `weight` is this rank's SHARD (rows rank, rank + G, ...) --------------------------------
    def add_table(self, name, weight, trainable=True, group=None, n_rows_global: Optional[int] = None, table=None):
        if table is not None:                   # CPU double supplies its own table object
            self._tables[name], self._weights[name], self._group[name] = table, weight, group
            t = table
        else:
            t = super().add_table(name, weight, trainable, group)
        self._n_rows_global[name] = int(n_rows_global if n_rows_global is not None else t.n_rows * self.G)
        return t

    # --- exchange ------------------------------------------------------------------------------------------
    def _a2a(self, out, t):
        dist.all_to_all_single(out, t, group=self.group)

    def _buffers(self, name, M) -> _Exchange:
        ex = self._ex.get(name)
        if ex is None or ex.M != M:
            cap = exchange_capacity(M, self.G, self.capacity_factor)
            ex = self._ex[name] = _Exchange(self.G, M, cap, self._tables[name].dim, self.device)
        return ex

    def _exchange_forward(self, name, idx, train):
        t, ops, G = self._tables[name], self.ops, self.G
        M = idx.numel()
        ex = self._buffers(name, M)
        n_slots = G * ex.cap
        ops.bucket(idx, G, ex.cap, ex.ids_send, ex.slot, ex.counts, self.err_flag)
        self._a2a(ex.ids_recv, ex.ids_send)
        if train:
            ops.gather_train(t, self._hyper(name), ex.ids_recv, n_slots, ex.rows_send, self.err_flag)
        else:
            ops.gather(t, self._hyper(name), ex.ids_recv, n_slots, ex.rows_send, self.err_flag)
        self._a2a(ex.rows_recv, ex.rows_send)
        out = torch.empty((M, t.dim), dtype=torch.float32, device=self.device)
        ops.unbucket_rows(ex.rows_recv, ex.slot, M, t.dim, out)
        self._last_idx = getattr(self, "_last_idx", {})
        self._last_idx[name] = idx
        return out

    def _exchange_backward(self, name, grad_rows):
        t, ex = self._tables[name], self._ex[name]
        grad_rows = grad_rows.contiguous()
        # the local loss is a mean over the local batch: 1/G makes the step that of the mean over the global batch
        self.ops.bucket_rows(grad_rows, ex.scale, ex.slot, ex.M, t.dim, ex.g_send)
        self._a2a(ex.g_recv, ex.g_send)
        t._grad_rows = ex.g_recv

    def lookup(self, name, idx):
        t = self._tables[name]
        idx = idx.to(self.device, torch.int64).contiguous()
        if t.trainable and torch.is_grad_enabled():
            return _ShardedLookup.apply(self._weights[name], self, name, idx)
        return self._exchange_forward(name, idx, train=False)

    # --- two tables of one step in ONE exchange -------------------------------------------------------------
    def lookup_pair(self, name_a, idx_a, name_b, idx_b):
        """rows of `name_a` at idx_a and of `name_b` at idx_b with the ids, the rows and (backward) the gradient rows of both
        tables packed into one buffer per direction: 3 all-to-alls per step instead of 3 per table."""
        ta, tb = self._tables[name_a], self._tables[name_b]
        idx_a = idx_a.to(self.device, torch.int64).contiguous()
        idx_b = idx_b.to(self.device, torch.int64).contiguous()
        if ta.dim != tb.dim or idx_a.numel() != idx_b.numel():
            return self.lookup(name_a, idx_a), self.lookup(name_b, idx_b)
        if torch.is_grad_enabled() and (ta.trainable or tb.trainable):
            return _ShardedLookupPair.apply(self._weights[name_a], self._weights[name_b], self, name_a, name_b, idx_a, idx_b)
        return self._pair_forward(name_a, idx_a, name_b, idx_b, train=False)

    def _pair_forward(self, name_a, idx_a, name_b, idx_b, train):
        ops, G = self.ops, self.G
        M = idx_a.numel()
        key = (name_a, name_b)
        ex = self._ex_pair.get(key) if hasattr(self, "_ex_pair") else None
        if ex is None or ex.M != M:
            if not hasattr(self, "_ex_pair"):
                self._ex_pair, self._lay = {}, {}
            cap = exchange_capacity(M, G, self.capacity_factor)
            ex = self._ex_pair[key] = _ExchangePair(G, M, cap, self._tables[name_a].dim, self.device)
        cap, S = ex.cap, ex.S
        n_slots = G * cap
        for k, (name, idx) in enumerate(((name_a, idx_a), (name_b, idx_b))):
            ops.bucket_at(idx, G, cap, S, k * cap, ex.ids_send, ex.slot[k], ex.counts[k], self.err_flag)
        self._a2a(ex.ids_recv, ex.ids_send)
        outs = []
        for k, (name, idx) in enumerate(((name_a, idx_a), (name_b, idx_b))):
            t = self._tables[name]
            if train and t.trainable:
                ops.gather_train_at(t, self._hyper(name), ex.ids_recv, k * cap, n_slots, cap, S, ex.rows_send, self.err_flag)
                self._lay[name] = (ex, k * cap)
            else:   # read-only table: its ids out of the packed buffer, its rows into it (two strided copies)
                ids = ex.ids_recv.view(G, S)[:, k * cap:(k + 1) * cap].contiguous().view(-1)
                rows = torch.empty((n_slots, t.dim), dtype=torch.float32, device=self.device)
                ops.gather(t, self._hyper(name), ids, n_slots, rows, self.err_flag)
                ex.rows_send.view(G, S, t.dim)[:, k * cap:(k + 1) * cap].copy_(rows.view(G, cap, t.dim))
        self._a2a(ex.rows_recv, ex.rows_send)
        self._last_idx = getattr(self, "_last_idx", {})
        for k, (name, idx) in enumerate(((name_a, idx_a), (name_b, idx_b))):
            out = torch.empty((M, self._tables[name].dim), dtype=torch.float32, device=self.device)
            ops.unbucket_rows(ex.rows_recv, ex.slot[k], M, out.shape[1], out)
            outs.append(out)
            self._last_idx[name] = idx
        return outs[0], outs[1]

    def _pair_backward(self, names, grads):
        ex = self._ex_pair[names]
        sent = False
        for k, (name, g) in enumerate(zip(names, grads)):
            t = self._tables[name]
            if g is None or not t.trainable or name not in self._lay:
                continue
            # the local loss is a mean over the local batch: 1/G makes the step that of the mean over the global batch
            self.ops.bucket_rows(g.contiguous(), ex.scale, ex.slot[k], ex.M, t.dim, ex.g_send)
            sent = True
        if not sent:
            return
        self._a2a(ex.g_recv, ex.g_send)
        for name in names:
            if name in self._lay and self._tables[name].trainable:
                self._tables[name]._grad_rows = ex.g_recv

    def global_item_df(self, name, out, label, sst, fair_weight, dy, loss):
        """NFCF's differential-fairness term (nfcf.py:76-97) on the GLOBAL batch, for the rows this rank looked up in table
        `name` (the item table) in the current step: adds this rank's share of d(fair_weight * DF)/dy to `dy` (scaled by G:
        the engine averages the ranks' gradients) and of fair_weight * DF to loss[0] (the mean of the ranks' losses is the
        global loss).  Two all-to-alls of [G, cap + 1] records; K and the groups present ride in their tails."""
        t, G, ops = self._tables[name], self.G, self.ops
        lay = getattr(self, "_lay", {}).get(name)
        if lay is not None:                 # the lookup went through a packed exchange: chunk g = [.. | cap slots of `name` | ..]
            ex, off = lay
            S, cap, slot = ex.S, ex.cap, ex.slot[off // ex.cap]
        else:
            ex = self._ex[name]
            S, cap, off, slot = ex.cap, ex.cap, 0, ex.slot
        B = out.numel()
        key = (name, B, cap)
        buf = getattr(self, "_df_buf", {}).get(key)
        if buf is None:
            if not hasattr(self, "_df_buf"):
                self._df_buf = {}
            n = G * (cap + 1)
            f32 = torch.float32
            buf = self._df_buf[key] = dict(
                rec_send=torch.zeros(n * 2, dtype=f32, device=self.device), rec_recv=torch.zeros(n * 2, dtype=f32, device=self.device),
                rep_send=torch.zeros(n * 4, dtype=f32, device=self.device), rep_recv=torch.zeros(n * 4, dtype=f32, device=self.device),
                ws=ops.df_workspace(B, G * cap, self.device))
        ops.df_pack(out, label, sst, slot, S, off, cap, G, buf["rec_send"], buf["ws"])
        self._a2a(buf["rec_recv"], buf["rec_send"])
        ops.df_owner(t, buf["rec_recv"], G, cap, buf["rep_send"], buf["ws"], B, self.err_flag)
        self._a2a(buf["rep_recv"], buf["rep_send"])
        ops.df_apply(buf["rep_recv"], slot, S, off, cap, G, out, label, sst, float(fair_weight), float(G), dy, loss, buf["ws"])

    def global_bpr_broadcast(self, a, c):
        """PFCN_BiasedMF's [B] + [B, 1] -> [B, B] training loss (pfcn_biasedmf.py:192-195) on the GLOBAL batch: the term matrix
        of G B rows x G B columns factors into row and column sums, so one all-gather of the ranks' (a, c) columns and two
        rectangular passes (this rank's columns against every row, its rows against every column) give this rank's exact
        gradients.  Returns (loss share, da, dc) scaled so that the engine's mean over the ranks is the global loss / the
        global gradient (x G)."""
        G, B = self.G, a.numel()
        mine = torch.cat([a, c]).contiguous()
        allv = torch.empty(G * 2 * B, dtype=torch.float32, device=self.device)
        dist.all_gather_into_tensor(allv, mine, group=self.group)
        allv = allv.view(G, 2, B)
        a_all, c_all = allv[:, 0].reshape(-1).contiguous(), allv[:, 1].reshape(-1).contiguous()
        inv = 1.0 / (float(G * B) ** 2)
        loss, scratch = torch.empty(1, dtype=torch.float32, device=self.device), torch.empty(1, dtype=torch.float32, device=self.device)
        da, dc = torch.empty_like(a), torch.empty_like(c)
        hold = self.__dict__.setdefault("_bpr_ws", [{}, {}])
        self.ops.bpr_outer_rect(a, c_all, inv, loss, da, None, hold[0])          # own columns x all rows
        self.ops.bpr_outer_rect(a_all, c, inv, scratch, None, dc, hold[1])      # all columns x own rows
        return loss * float(G), da * float(G), dc * float(G)

    def batch_segments(self, name):
        """Sorted segments of THIS rank's ids of the last lookup in `name` (per-rank batch statistics)."""
        seg = self._seg.setdefault(name, _Segments(self._tables[name].dim))
        self.ops.sort_local(self._last_idx[name], self._n_rows_global[name], seg.dim, seg)
        return seg

    # --- optimizer.step() ------------------------------------------------------------------------------------
    def backward_adam(self, group=None):
        self.note_stepped(group)
        G = self.G
        for name, t in self._tables.items():
            if not (t.trainable and t._pending is not None):
                continue
            if not self._owned(name, group) or t._grad_rows is None:
                t._pending = None
                t._grad_rows = None
                continue
            lay = getattr(self, "_lay", {}).pop(name, None)
            ex = lay[0] if lay is not None else self._ex[name]
            n_slots = G * ex.cap
            s = self.sweep_period if self.sweep_period is not None else max(8, math.ceil(t.n_rows / max(n_slots // 2, 1)))
            if lay is not None:     # this step's lookup went through a packed exchange (lookup_pair)
                self.ops.apply_grad_at(t, self._hyper(name), n_slots, ex.rows_send, t._grad_rows, lay[1], ex.cap, ex.S, s)
            else:
                self.ops.apply_grad(t, self._hyper(name), n_slots, ex.rows_send, t._grad_rows, s)
            t._grad_rows = None
        if self._synced is None:      # (clip_grad_norm has done it already when the optimizer clips)
            self._sync_dense(group)
        if self._synced is None:
            return
        live, views = self._synced
        self._synced = None
        for name, d in live:
            d.step += 1
            h = self._hyper(name)
            h.check_step(d.step)
            self.ops.adam_dense(d.p.data, views[name], d.m, d.v, h, d.step)
            d.p.grad = None
