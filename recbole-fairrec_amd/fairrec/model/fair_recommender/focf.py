"""FOCF (Yao & Huang, "Beyond parity", NIPS 2017) on the MI355X hot path.

Same plugin surface as the reference's recbole/model/fair_recommender/focf.py:24-178 (class name,
constructor signature, attribute names `user_embedding_layer` / `item_embedding_layer`, config keys
`embedding_size`, `RATING_FIELD`, `sst_attr_list`, `fair_weight`, `fair_objective`), but
calculate_loss / predict run as hand-written HIP kernels (csrc/focf.hip) and the backward + Adam
update is one fused launch driven by fairrec.optim.FusedLazyAdam.  No CPU fallback.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from ... import _C
from ...optim import AdamHyper, FusedLazyAdam, LazyTable
from ...utils.enum_type import InputType
from ..abstract_recommender import FairRecommender
from ..init import xavier_normal_initialization


class _LossHandle(torch.autograd.Function):
    """Gives the device loss scalar an autograd edge so that `loss.backward()` (trainer.py:193) is legal;
    the real backward runs fused with the Adam update in FocfEngine.backward_adam()."""

    @staticmethod
    def forward(ctx, loss, engine, *weights):
        ctx.engine = engine
        return loss.view(())

    @staticmethod
    def backward(ctx, grad):
        ctx.engine.backward_seen = True
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class FocfEngine:
    """Owns the two lazy-Adam tables, the per-batch workspace and the kernel launches of FOCF."""

    LOSS_SLOTS = 256

    def __init__(self, user_weight: torch.Tensor, item_weight: torch.Tensor, objective: str, fair_weight: float,
                 max_rating: float):
        if user_weight.device.type != "cuda":
            raise _C.FairrecError("FOCF runs only on a ROCm device (model.to('cuda')); there is no CPU fallback")
        _C.lib()  # fail loudly now if the HIP extension is missing
        self.device = user_weight.device
        self.U = LazyTable(user_weight)
        self.I = LazyTable(item_weight)
        self.objective = _C.FOCF_OBJECTIVES[objective]
        self.fair_weight = float(fair_weight)
        self.max_rating = float(max_rating)
        self.hyper = AdamHyper(device=self.device, cap=1)  # placeholder until an optimizer binds (step 0: no replay)
        self.optimizer: Optional[FusedLazyAdam] = None
        self.sweep_period: Optional[int] = None
        self.ws = [None, None]          # double-buffered per-batch workspaces (the next batch's sort runs one step ahead)
        self.ws_cur = 0
        self._side = None               # stream of the look-ahead sort
        self._prep = None               # (key, ws index, done-event) of the batch prepared ahead
        self.loss_ring = torch.zeros((self.LOSS_SLOTS, 4), dtype=torch.float32, device=self.device)
        self.loss_slot = 0
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.pending_B = 0
        self.backward_seen = False

    # --- optimizer plumbing ---------------------------------------------------------------------------
    def tables(self) -> Dict[str, LazyTable]:
        return {"user_embedding_layer.weight": self.U, "item_embedding_layer.weight": self.I}

    def bind_optimizer(self, opt: FusedLazyAdam, sweep_period: Optional[int]):
        self.optimizer = opt
        self.hyper = opt.hyper
        self.sweep_period = sweep_period
        for t in (self.U, self.I):
            t.ensure_state()

    def _sweep(self, B: int) -> int:
        if self.sweep_period is not None:
            return int(self.sweep_period)
        # default: sweep about B rows of the larger table per step => a row is never more than S ~ N/B steps stale
        return max(8, math.ceil(max(self.U.n_rows, self.I.n_rows) / max(B, 1)))

    def _workspace(self, B: int, k: int):
        need = _C.lib().fr_focf_workspace_bytes(B, self.U.dim)
        if self.ws[k] is None or self.ws[k].numel() < need:
            self.ws[k] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.ws[k]

    @staticmethod
    def _key(user, item):
        return (user.data_ptr(), item.data_ptr(), user.numel())

    def prepare(self, user, item, sst, ws_index: int, sweep_step: int = -1):
        """Index-only part of the NEXT batch (fr_focf_prepare: sort + segmentation + sst min/max) on a side stream,
        overlapping the kernels of the current batch.  With an optimizer bound it also stamps the batch's rows with the
        step it will be applied as, and -- `sweep_step` >= 1 -- carries the sweep slice of that step in the same
        launch (allowed when that step's own batch was prepared with stamps)."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._ev_start = [torch.cuda.Event() for _ in range(2)]     # reused every step: one pair per workspace
            self._ev_done = [torch.cuda.Event() for _ in range(2)]
        main = torch.cuda.current_stream()
        B = user.numel()
        ws = self._workspace(B, ws_index)
        start = self._ev_start[ws_index]
        start.record(main)                  # everything that last used ws[ws_index] was enqueued before this point
        self._side.wait_event(start)
        sst_arg = sst if self.objective != 0 else None
        # the batch will be applied as optimizer step `step + 2` (the forward in flight is `step + 1`)
        stamped = self.optimizer is not None
        for_step = self.U.step + 2
        null_t, null_a = ctypes.POINTER(_C.FrTable)(), ctypes.POINTER(_C.FrAdam)()
        if stamped:
            if sweep_step >= 1:
                self.hyper.check_step(sweep_step)
            tu, ti = self.U.c(for_step), self.I.c(for_step)
            targs = (ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()))
        else:
            targs = (null_t, null_t, null_a)
            sweep_step = -1
        rc = _C.lib().fr_focf_prepare(user.data_ptr(), item.data_ptr(), _C.ptr(sst_arg), B, self.U.n_rows,
                                      self.I.n_rows, self.U.dim, ws.data_ptr(), ws.numel(), *targs, for_step,
                                      self._sweep(B) if sweep_step >= 1 else 0, sweep_step,
                                      self.err_flag.data_ptr(), self._side.cuda_stream)
        _C.check(rc, "fr_focf_prepare")
        done = self._ev_done[ws_index]
        done.record(self._side)
        self._prep = (self._key(user, item), ws_index, done, stamped, for_step)

    def _join_prepare(self):
        """A prepare launch may carry sweeper work on the tables: anything else that touches them waits for it."""
        if self._prep is not None:
            torch.cuda.current_stream().wait_event(self._prep[2])

    # --- launches -------------------------------------------------------------------------------------
    def forward(self, user, item, rating, sst, want_pred: bool = False, next_batch=None):
        """fr_focf_forward for the step `step+1`; returns (loss[4] device view, pred or None).
        `next_batch` = (user, item, sst) of the following step, if known: its sort is launched now, one step ahead."""
        B = user.numel()
        flags = 0
        stamped_now = False
        if self._prep is not None:
            key, ws_index, done, stamped, for_step = self._prep
            torch.cuda.current_stream().wait_event(done)         # also when the batch differs: sweeper work may ride there
            if key == self._key(user, item):
                self.ws_cur = ws_index
                flags = 1                                        # FR_FOCF_PREPARED
                stamped_now = stamped and for_step == self.U.step + 1
        self._prep = None
        ws = self._workspace(B, self.ws_cur)
        if next_batch is not None:
            # this batch's rows carry their stamps already => the sweep slice of this step may ride with the next sort
            self.prepare(next_batch[0], next_batch[1], next_batch[2], 1 - self.ws_cur,
                         sweep_step=self.U.step + 1 if stamped_now else -1)
        self.loss_slot = (self.loss_slot + 1) % self.LOSS_SLOTS
        loss = self.loss_ring[self.loss_slot]
        pred = torch.empty(B, dtype=torch.float32, device=self.device) if want_pred else None
        tu, ti = self.U.c(self.U.step + 1), self.I.c(self.I.step + 1)
        self.hyper.check_step(self.U.step + 1)
        rc = _C.lib().fr_focf_forward(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()),
                                      user.data_ptr(), item.data_ptr(), rating.data_ptr(), _C.ptr(sst), B,
                                      self.objective, self.fair_weight, flags, ws.data_ptr(), ws.numel(),
                                      loss.data_ptr(), _C.ptr(pred), self.err_flag.data_ptr(), _C.current_stream())
        _C.check(rc, "fr_focf_forward")
        self.pending_B = B
        return loss, pred

    def backward_adam(self):
        """loss.backward() + optimizer.step() of the pending batch (fr_focf_backward_adam)."""
        if self.pending_B == 0:
            raise _C.FairrecError("optimizer.step() without a preceding calculate_loss()")
        if self.optimizer is None:
            raise _C.FairrecError("no optimizer bound: build fairrec.optim.FusedLazyAdam(model.hip_engine(), ...)")
        B = self.pending_B
        tu, ti = self.U.c(self.U.step + 1), self.I.c(self.I.step + 1)
        rc = _C.lib().fr_focf_backward_adam(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()), B,
                                            self._sweep(B), self.ws[self.ws_cur].data_ptr(),
                                            self.ws[self.ws_cur].numel(), _C.current_stream())
        _C.check(rc, "fr_focf_backward_adam")
        self.U.step += 1
        self.I.step += 1
        self.U._dirty = self.I._dirty = True
        self.pending_B = 0
        self.backward_seen = False

    def predict(self, user, item):
        self._join_prepare()
        B = user.numel()
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        tu, ti = self.U.c(), self.I.c()
        rc = _C.lib().fr_focf_predict(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()),
                                      user.data_ptr(), item.data_ptr(), B, self.max_rating, out.data_ptr(),
                                      self.err_flag.data_ptr(), _C.current_stream())
        _C.check(rc, "fr_focf_predict")
        return out

    def flush(self):
        self._join_prepare()
        self.U.flush(self.hyper)
        self.I.flush(self.hyper)

    def check_device_errors(self):
        """Host sync: raise what the reference would have raised eagerly (IndexError)."""
        e = int(self.err_flag.item())
        if e:
            self.err_flag.zero_()
            msgs = []
            if e & _C.DEV_ERR_INDEX_RANGE:
                msgs.append("index out of range in embedding gather")
            if e & _C.DEV_ERR_SST_GROUPS:
                msgs.append("a batch must hold exactly the 1..2 sensitive groups the objective expects")
            raise IndexError("; ".join(msgs))


class FOCF(FairRecommender):
    """MF + fairness regulariser; drop-in for recbole.model.fair_recommender.focf.FOCF."""

    input_type = InputType.POINTWISE
    # the gradient never exists as a tensor: optimizer.step() runs backward + Adam in one launch.  `loss.backward()` stays
    # legal (autograd edge in calculate_loss) for loops written against the reference; fairrec's own Trainer skips it.
    fused_backward = True

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.embedding_size = config['embedding_size']
        self.RATING = config['RATING_FIELD']
        self.SST_FIELD = config['sst_attr_list'][0]
        self.fair_weight = config['fair_weight']
        self.max_rating = dataset.inter_feat[self.RATING].max()
        self.fair_objective = self._parse_objective(config['fair_objective'])

        self.user_embedding_layer = nn.Embedding(self.n_users, self.embedding_size)
        self.item_embedding_layer = nn.Embedding(self.n_items, self.embedding_size)
        self.apply(xavier_normal_initialization)   # reference focf.py:48
        self._engine: Optional[FocfEngine] = None

    @staticmethod
    def _parse_objective(name):
        name = str(name).strip().lower()   # reference focf.py:50-68
        if name not in _C.FOCF_OBJECTIVES:
            raise ValueError("you must set config['fair_objective'] be one of (none,"
                             "value,absolute,under,over,nonparity)")
        return name

    # --- engine ---------------------------------------------------------------------------------------
    def hip_engine(self) -> FocfEngine:
        uw, iw = self.user_embedding_layer.weight, self.item_embedding_layer.weight
        if self._engine is None or self._engine.U.weight.data_ptr() != uw.data_ptr():
            self._engine = FocfEngine(uw.data, iw.data, self.fair_objective, float(self.fair_weight or 0.0),
                                      float(self.max_rating))
        return self._engine

    def _cols(self, interaction, need_targets=True):
        dev = self.user_embedding_layer.weight.device
        u = interaction[self.USER_ID].to(dev, torch.int64).contiguous()
        i = interaction[self.ITEM_ID].to(dev, torch.int64).contiguous()
        if not need_targets:
            return u, i, None, None
        r = interaction[self.RATING].to(dev, torch.float32).contiguous()
        s = None
        if self.fair_objective != 'none':
            s = interaction[self.SST_FIELD].to(dev, torch.float32).contiguous()
        return u, i, r, s

    # --- plugin surface -------------------------------------------------------------------------------
    def forward(self, user, item):
        """pred_scores and the two gathered embedding blocks, as reference focf.py:136-143 returns them."""
        eng = self.hip_engine()
        ue = eng.U.gather(eng.hyper, user, eng.err_flag)
        ie = eng.I.gather(eng.hyper, item, eng.err_flag)
        return (ue * ie).sum(-1), ue, ie

    def hint_next_batch(self, interaction):
        """Optional trainer hook: the batch that will follow the next `calculate_loss` (None at the epoch end)."""
        self._next_cols = self._cols(interaction) if interaction is not None else None

    def calculate_loss(self, interaction):
        eng = self.hip_engine()
        u, i, r, s = self._cols(interaction)
        nxt = getattr(self, '_next_cols', None)
        self._next_cols = None
        loss, _ = eng.forward(u, i, r, s, next_batch=(nxt[0], nxt[1], nxt[3]) if nxt is not None else None)
        if torch.is_grad_enabled():
            return _LossHandle.apply(loss[0], eng, self.user_embedding_layer.weight, self.item_embedding_layer.weight)
        return loss[0]

    def predict(self, interaction):
        u, i, _, _ = self._cols(interaction, need_targets=False)
        return self.hip_engine().predict(u, i)

    def full_sort_predict(self, interaction):
        eng = self.hip_engine()
        eng.flush()
        user = interaction[self.USER_ID].to(eng.device)
        scores = torch.mm(self.user_embedding_layer.weight.data[user], self.item_embedding_layer.weight.data.t())
        return (torch.clamp(scores, min=0., max=eng.max_rating) / eng.max_rating).view(-1)

    def state_dict(self, *args, **kwargs):
        if self._engine is not None:
            self._engine.flush()   # checkpoints must see every row at the current optimizer step
        return super().state_dict(*args, **kwargs)
