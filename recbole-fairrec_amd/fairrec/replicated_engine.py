"""Data-parallel training with REPLICATED tables, for the models whose hot stage reads whole tables: FairGo_PMF / FairGo_GCN
(BASELINE.json configs[3]).

The finetune stage of FairGo filters and propagates the whole frozen [n_users + n_items, D] table every step
(fairgo_pmf.py:175-199) and `trainer.py:857-862` freezes the tables there, so replicas are EXACT: one process per GPU holds
the full tables, filters, discriminators and the normalised graph; the batch is sharded (every rank feeds its own B
interactions) and ONE flat all-reduce per optimizer step averages the gradients of the replicated dense parameters
(filters / discriminators / aggr_layer, a few hundred KB) before the same fused Adam step runs on every replica -- identical
inputs, so the replicas stay bit-identical.  A step is the single-device step on the concatenated batch for every loss that
is a mean over interactions (FairGo's MSE and BCE / CE terms; its MLPs have no BatchNorm).

The pretrain stage does train the tables: there a lookup all-gathers the ranks' ids, every replica gathers the rows of the
GLOBAL batch (lazy replay) and hands its own slice to the model; the slice's gradient rows are all-gathered back (scaled
1/G) and every replica applies the same duplicate-summed Adam update -- G-fold redundant work on a stage that is a small
part of the run, no exchange buffers, no capacity limits.

Kernels come from an `ops` object (default: HIP through fairrec._C); tests inject a CPU double to run the schedule over gloo.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.distributed as dist

from .engine import GenericEngine
from .sharded_engine import HipTableOps


class _ReplicatedLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, engine, name, idx):
        ctx.engine, ctx.name = engine, name
        return engine._gather_global(name, idx, train=True)

    @staticmethod
    def backward(ctx, grad_rows):
        ctx.engine._park_global_grad(ctx.name, grad_rows)
        return None, None, None, None


class ReplicatedGenericEngine(GenericEngine):
    def __init__(self, device, group=None, ops=None):
        self.pg = group
        self.G = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.ops = ops
        if ops is None:
            super().__init__(device)
            self.ops = HipTableOps()
        else:                                   # CPU double (tests): no device library behind it
            from .optim import AdamHyper
            self.device = torch.device(device)
            self._tables, self._weights, self._dense, self._group, self._hyper_of = {}, {}, {}, {}, {}
            self.hyper = AdamHyper(device=self.device, cap=1)
            self.optimizer, self.sweep_period = None, None
            self.err_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._counters, self._group_version = None, {}
            self._seg_src = {}
        self._buf: Dict[str, dict] = {}
        self._flat: Optional[torch.Tensor] = None
        self._synced = None          # (live dense entries, their averaged-gradient views of _flat) between clip and step

    def enable_graph_mode(self):
        raise NotImplementedError("collectives inside a captured step are not enabled for the replicated engine: "
                                  "set graph_train_step: False")

    def add_table(self, name, weight, trainable=True, group=None, table=None):
        if table is not None:                   # CPU double supplies its own table object
            self._tables[name], self._weights[name], self._group[name] = table, weight, group
            return table
        return super().add_table(name, weight, trainable, group)

    # --- lookups on a trainable table: the GLOBAL batch on every replica ---------------------------------------------
    def _gather_global(self, name, idx, train):
        t, G = self._tables[name], self.G
        M = idx.numel()
        b = self._buf.get(name)
        if b is None or b["M"] != M:
            b = self._buf[name] = {"M": M, "ids": torch.empty(G * M, dtype=torch.int64, device=self.device),
                                   "rows": torch.empty((G * M, t.dim), dtype=torch.float32, device=self.device),
                                   "grads": torch.empty((G * M, t.dim), dtype=torch.float32, device=self.device)}
        dist.all_gather_into_tensor(b["ids"], idx, group=self.pg)
        if train:
            self.ops.gather_train(t, self._hyper(name), b["ids"], G * M, b["rows"], self.err_flag)
        else:
            self.ops.gather(t, self._hyper(name), b["ids"], G * M, b["rows"], self.err_flag)
        return b["rows"][self.rank * M:(self.rank + 1) * M].clone()

    def _park_global_grad(self, name, grad_rows):
        t, b = self._tables[name], self._buf[name]
        # the local loss is a mean over the local batch: 1/G makes the step that of the mean over the global batch
        dist.all_gather_into_tensor(b["grads"], (grad_rows * (1.0 / self.G)).contiguous(), group=self.pg)
        t._grad_rows = b["grads"]

    def lookup(self, name, idx):
        t = self._tables[name]
        idx = idx.to(self.device, torch.int64).contiguous()
        if t.trainable and torch.is_grad_enabled():
            return _ReplicatedLookup.apply(self._weights[name], self, name, idx)
        return super().lookup(name, idx) if self.ops.__class__ is HipTableOps else self._tables[name].weight[idx]

    # --- optimizer.step() --------------------------------------------------------------------------------------------
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
            b = self._buf[name]
            n = G * b["M"]
            s = self.sweep_period if self.sweep_period is not None else max(8, math.ceil(t.n_rows / max(n, 1)))
            self.ops.apply_grad(t, self._hyper(name), n, b["rows"], t._grad_rows, s)
            t._grad_rows = None
        if self._synced is None:      # (clip_grad_norm has done it already when the optimizer clips)
            self._sync_dense(group)
        if self._synced is None:
            return
        live, views = self._synced
        self._synced = None
        for name, d in self._dense.items():          # an other group's gradient that the clip averaged and scaled: its
            if name in views and (name, d) not in live and d.p.grad is not None:     # .grad holds that result, as torch's would
                d.p.grad.copy_(views[name].view_as(d.p.grad))
        for name, d in live:
            d.step += 1
            h = self._hyper(name)
            h.check_step(d.step)
            self.ops.adam_dense(d.p.data, views[name], d.m, d.v, h, d.step)
            d.p.grad = None

    def _sync_dense(self, group=None, measured=False):
        """One flat all-reduce (mean over the replicas) of the dense gradients the stepping group owns (SURVEY.md §8-e
        item 5); kept until backward_adam consumes it.  `measured` (clip_grad_norm): the gradients of the OTHER groups'
        model parameters ride along -- they enter the global norm, so every replica must see their average too or each rank
        would clip by its own coefficient and the replicas would drift apart."""
        live = [(name, d) for name, d in self._dense.items() if d.p.grad is not None and self._owned(name, group)]
        extra = [(name, d) for name, d in self._dense.items()
                 if measured and d.p.grad is not None and not self._owned(name, group)
                 and not name.startswith(self.NOT_MODEL_PARAMETERS)]
        if not live and not extra:
            self._synced = None
            return
        n = sum(d.p.numel() for _, d in live + extra)
        if self._flat is None or self._flat.numel() < n:
            self._flat = torch.empty(n, dtype=torch.float32, device=self.device)
        flat = self._flat[:n]
        torch.cat([d.p.grad.reshape(-1) for _, d in live + extra], out=flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg)
        flat.mul_(1.0 / self.G)
        views, off = {}, 0
        for name, d in live + extra:
            k = d.p.numel()
            views[name] = flat[off:off + k]
            off += k
        self._synced = (live, views)

    def clip_grad_norm(self, max_norm: float, group=None):
        """GenericEngine.clip_grad_norm on replicas: the norm of the GLOBAL batch's gradient, one coefficient for every
        replica.  A trainable table's gradient is the all-gathered [G * M, D] rows (identical on every rank), duplicates
        summed per global id before squaring; a dense parameter's is measured AFTER the flat all-reduce (done here instead
        of in backward_adam, which then finds it averaged) -- clipping each replica's local gradient first would give every
        rank its own coefficient and the step would not be the single-device clip of the averaged gradient."""
        self._sync_dense(group, measured=True)
        sq = torch.zeros((), dtype=torch.float32, device=self.device)
        held = []
        for name, t in self._tables.items():
            if name.startswith(self.NOT_MODEL_PARAMETERS) or not t.trainable or t._grad_rows is None:
                continue
            ids, g = self._buf[name]["ids"], t._grad_rows
            order = torch.argsort(ids, stable=True)
            _, counts = torch.unique_consecutive(ids[order], return_counts=True)
            rows = torch.segment_reduce(g[order].contiguous(), "sum", lengths=counts, axis=0)
            sq = sq + (rows * rows).sum()
            held.append(t)
        dense = []
        synced = self._synced[1] if self._synced is not None else {}
        for name, d in self._dense.items():
            if name.startswith(self.NOT_MODEL_PARAMETERS) or d.p.grad is None:
                continue
            gview = synced[name]                        # averaged over the replicas (own group's and measured others')
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
