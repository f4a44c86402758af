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
import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from ... import _C
from ...optim import AdamHyper, FusedLazyAdam, LazyTable
from ...utils.enum_type import InputType
from ..abstract_recommender import FairRecommender
from ..init import xavier_normal_initialization


# fr_focf_batch (include/fairrec_hip.h) as a numpy record: a run of batches is described without a Python loop
_BATCH_DTYPE = [("user", "<u8"), ("item", "<u8"), ("sst", "<u8"), ("B", "<i8"), ("ws", "<u8"), ("ws_bytes", "<u8"),
                ("rating", "<u8")]


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
    GROUP = int(os.environ.get("FAIRREC_FOCF_GROUP", 16))       # coming batches prepared per fork of the side stream
                      # (FR_FOCF_PREPARE_MAX of them per launch; 32 and 48 measured slower: rows stamped further ahead are
                      # left to their batch by more sweeps)
    PER_LAUNCH = 8    # FR_FOCF_PREPARE_MAX
    LOW_WATER = int(os.environ.get("FAIRREC_FOCF_LOW_WATER", 4))  # ... launched when this few prepared batches are left, so
                      # that its join is steps old when reached
    N_WS = GROUP + LOW_WATER + 4   # workspaces: the batch in flight + the last two (item runs pending / loss) + the prepared ones + a spare
    # fr_focf_step_staged: the index work of the two coming batches rides in the step launches themselves (no sort, no side
    # stream); FAIRREC_FOCF_STAGED=0 goes back to the look-ahead sort (fr_focf_prepare_step) for the one-launch step
    STAGED = os.environ.get("FAIRREC_FOCF_STAGED", "1") != "0"
    RUNS = os.environ.get("FAIRREC_FOCF_RUNS", "1") != "0"
    # fr_focf_step_runs_pipe: the item runs of batch k - 1 and the gather of batch k in ONE launch (the two ~22-25 us chains of
    # an item-complete step side by side instead of one after the other).  The tables then lag one finisher behind `step`
    # between calls: everything that reads them goes through finish() first.
    PIPE = os.environ.get("FAIRREC_FOCF_PIPE", "1") == "1"

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
        self.ws = [None] * self.N_WS    # ring of per-batch workspaces (the sorts of the next batches run ahead)
        self.ws_cur = 0
        self._side = None               # stream of the look-ahead sorts
        self._prep = {}                 # batch key -> (ws index, launch group) of the batches prepared ahead
        self.loss_ring = torch.zeros((self.LOSS_SLOTS, 4), dtype=torch.float32, device=self.device)
        self._loss_views = [self.loss_ring[k] for k in range(self.LOSS_SLOTS)]
        self._ws_need = {}
        self.loss_slot = 0
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.pending_B = 0
        self.backward_seen = False
        # FR_FOCF_DEFER_LOSS: the loss slot is filled by backward_adam() instead of forward().  Only for step loops that
        # read the loss after optimizer.step() (fairrec's Trainer fast path, bench.py).
        self.defer_loss = False
        # FR_FOCF_ITEM_RUNS (a hint, same results either way): the batches are item-complete (FOCFDataLoader), the
        # interactions of an item sit side by side -- the gather kernel then replays an item row once per workgroup
        self.item_runs = False
        # fr_focf_step: the whole step as ONE launch (rows stay in registers from the gather to the Adam write-back).
        # Taken when the loss is read after optimizer.step() (defer_loss), the objective needs no batch-wide value between
        # forward and update (all but nonparity), no clip_grad_norm is asked for and the batches are not item-complete
        # (rows shared by ~100 interactions are finished by ONE wave there; the three-launch chain spreads them).
        self.fused_step = True
        self._stash = None              # batch of a fused step: launched by backward_adam()
        self._prev = None               # (workspace, B, loss view) of the last fused step, its loss not reduced yet
        self._pipe = None               # pipelined item-run steps: (workspace, B, step, loss view, columns) gathered, item runs pending
        self._own = None                # ... their record of which batch gathered a row last (int32 [2][n_users], [2][n_items])
        self._stamp_last = 0            # stamps handed to fr_focf_prepare_step never decrease
        self._stamp_gen = (self.U.stamp_gen, self.I.stamp_gen)
        # running (loss, mse, fair) total, [3] = steps in it, [4] = 1-based index of the first NaN step (0 = none; sticky)
        self.loss_acc = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.staged = self.STAGED
        self._st = {}                   # batch key -> entry of a batch on its way through the claim / place stages
        self._row_words = None          # [3][n_users + n_items] 64-bit words (fr_focf_row_words), zero to begin with
        self._ws_dirty = set()          # workspaces whose stage counters may be left over from a batch that never ran
        self._gen_cur = None            # generation of row words of the batch being applied

    # --- optimizer plumbing ---------------------------------------------------------------------------
    def tables(self) -> Dict[str, LazyTable]:
        if self._pipe is not None:      # whoever asks for the tables reads them: the pending item runs first
            self.finish()
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
        need = self._ws_need.get(B)
        if need is None:
            need = self._ws_need[B] = _C.lib().fr_focf_workspace_bytes(B, self.U.dim)
        if self.ws[k] is None or self.ws[k].numel() < need:
            self.ws[k] = torch.zeros(need, dtype=torch.uint8, device=self.device)     # (the stage counters start at zero)
        return self.ws[k]

    @staticmethod
    def _key(user, item):
        return (user.data_ptr(), item.data_ptr(), user.numel())

    def prepare_many(self, batches, ahead: int = 0):
        """Index-only part of COMING batches (fr_focf_prepare_step: sort + segmentation + sst min/max of each, the packed
        per-interaction records and the row stamps of the one-launch step; one launch for all of them) on a side stream,
        overlapping the kernels of the batches in flight.  `batches` = [(user, item, sst, rating)] in the order they
        will be applied ((user, item, sst) triples are accepted: such a batch is prepared for the three-launch chain
        only); `ahead` = optimizer steps that will be applied before the first of them (1 when called from inside
        forward(): the current batch comes first)."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._ws_next = 0
        main = torch.cuda.current_stream()
        busy = {self.ws_cur} | {v[0] for v in self._prep.values()}
        if self._prev is not None:   # the last fused step's loss is reduced by the NEXT launch, from its workspace
            busy |= {k for k, w in enumerate(self.ws) if w is self._prev[0]}
        if self._pipe is not None:   # ... and a pipelined step's item runs ride in the next launch
            busy |= {k for k, w in enumerate(self.ws) if w is self._pipe[0]}
        arr = (_C.FrFocfBatch * len(batches))()
        stamps = (ctypes.c_int32 * len(batches))()
        group = {"done": torch.cuda.Event(), "joined": False}
        entries = []
        self._check_stamp_gen()
        full = all(len(bt) >= 4 and bt[3] is not None for bt in batches)
        for q, bt in enumerate(batches):
            user, item, sst = bt[:3]
            if len(busy) >= self.N_WS:
                raise _C.FairrecError("too many batches prepared ahead")
            while self._ws_next in busy:
                self._ws_next = (self._ws_next + 1) % self.N_WS
            k = self._ws_next
            busy.add(k)
            B = user.numel()
            ws = self._workspace(B, k)
            if not torch.cuda.is_current_stream_capturing():
                # (the allocator recycles memory in the order of the streams it knows a tensor was used on: these are read
                # and written on the side stream, and an engine may go out of scope with such work still queued)
                for tns in (ws, user, item, sst, bt[3] if len(bt) >= 4 else None):
                    if tns is not None:
                        tns.record_stream(self._side)
            arr[q] = _C.FrFocfBatch(user.data_ptr(), item.data_ptr(), _C.ptr(sst if self.objective != 0 else None), B,
                                    ws.data_ptr(), ws.numel(), _C.ptr(bt[3]) if full else None)
            stamps[q] = self._next_stamp(q + ahead) if full else 0
            entries.append((self._key(user, item), (k, group, int(stamps[q]) if full else None)))
        # everything that last used these workspaces was enqueued before this point
        start = torch.cuda.Event()
        start.record(main)
        self._side.wait_event(start)
        PL = self.PER_LAUNCH
        if full:
            self.U.ensure_state()
            self.I.ensure_state()
            tu, ti = self.U.c(), self.I.c()
            for a in range(0, len(batches), PL):
                n = min(PL, len(batches) - a)
                rc = _C.lib().fr_focf_prepare_step(ctypes.cast(ctypes.addressof(arr) + a * ctypes.sizeof(_C.FrFocfBatch), ctypes.POINTER(_C.FrFocfBatch)),
                                                   ctypes.cast(ctypes.addressof(stamps) + a * 4, ctypes.POINTER(ctypes.c_int32)), n, ctypes.byref(tu), ctypes.byref(ti),
                                                   self._sweep(batches[0][0].numel()), self.err_flag.data_ptr(),
                                                   self._side.cuda_stream)
                _C.check(rc, "fr_focf_prepare_step")
        else:
            for a in range(0, len(batches), PL):
                n = min(PL, len(batches) - a)
                rc = _C.lib().fr_focf_prepare_many(ctypes.cast(ctypes.addressof(arr) + a * ctypes.sizeof(_C.FrFocfBatch), ctypes.POINTER(_C.FrFocfBatch)), n, self.U.n_rows,
                                                   self.I.n_rows, self.U.dim, self.err_flag.data_ptr(),
                                                   self._side.cuda_stream)
                _C.check(rc, "fr_focf_prepare_many")
        group["done"].record(self._side)
        self._prep.update(entries)

    def prepared_is_complete(self):
        """The caller has synchronised with the device: every prepare launched so far has finished, so the steps that use
        those batches need not wait for the side stream (e.g. before capturing them into a hipGraph, where a wait on an event
        recorded outside the capture has no place)."""
        for _, group, _ in self._prep.values():
            group["joined"] = True

    def join_prepared(self):
        """Make the current stream wait for every prepare in flight (the end of a captured region must leave no side-stream
        work unjoined)."""
        for _, group, _ in self._prep.values():
            if not group["joined"]:
                torch.cuda.current_stream().wait_event(group["done"])
                group["joined"] = True

    def _next_stamp(self, ahead: int = 0) -> int:
        """Stamp of a batch about to be prepared: the step at which it is expected to be applied (`ahead` steps after the
        next one), strictly increasing from one call to the next."""
        s = max(self._stamp_last + 1, self.U.step + 1 + ahead)
        self._stamp_last = s
        return s

    def _check_stamp_gen(self):
        """The tables' stamps were reset (optimizer state reloaded): what was prepared ahead lost its stamps."""
        gen = (self.U.stamp_gen, self.I.stamp_gen)
        if gen != self._stamp_gen:
            self._stamp_gen = gen
            self._join_prepare()
            self._forget_staged()
            self._pipe = None               # (its tables are gone with the old state)
            if self._own is not None:
                self._own[0].zero_()
                self._own[1].zero_()

    def _join_prepare(self):
        """Order the current stream behind every sort launch still in flight, and forget what they prepared."""
        for v in self._prep.values():
            torch.cuda.current_stream().wait_event(v[1]["done"])
        self._prep.clear()

    # --- in-launch prepare (fr_focf_stage / fr_focf_step_staged) -----------------------------------------
    def _st_entry(self, user, item, sst, rating, ahead: int):
        """A batch enters the pipeline: a workspace of its own and the stamp of the step it is expected at."""
        busy = {self.ws_cur} | {e["k"] for e in self._st.values()}
        if self._prev is not None:
            busy |= {k for k, w in enumerate(self.ws) if w is self._prev[0]}
        if self._stash is not None:
            busy |= {k for k, w in enumerate(self.ws) if w is self._stash[5]}
        k = next(j for j in range(self.N_WS) if j not in busy)
        B = user.numel()
        ws = self._workspace(B, k)
        if k in self._ws_dirty:
            ws.zero_()
            self._ws_dirty.discard(k)
        batch = _C.FrFocfBatch(user.data_ptr(), item.data_ptr(), _C.ptr(sst if self.objective != 0 else None), B,
                               ws.data_ptr(), ws.numel(), rating.data_ptr())
        # a generation of row words nobody in flight holds (the batch being applied, the placed one, the claimed one)
        held = {e["gen"] for e in self._st.values()} | ({self._gen_cur} if self._gen_cur is not None else set())
        gen = next(j for j in range(3) if j not in held)
        # (`step`: the optimizer step the batch is expected at -- the stamp may run ahead of it once a claimed batch was dropped)
        return {"k": k, "ws": ws, "B": B, "stamp": self._next_stamp(ahead), "step": self.U.step + 1 + ahead, "gen": gen,
                "stage": 0, "batch": batch, "cols": (user, item, sst, rating)}

    def _stage_now(self, claim=None, place=None):
        """Stages on a launch of their own (the first batches of a loop: no earlier step launch could carry them)."""
        at = (place or claim)["step"]       # fr_focf_stage: table.step = the step the batch will be applied at
        tu, ti = self.U.c(at), self.I.c(at)
        rc = _C.lib().fr_focf_stage(ctypes.byref(tu), ctypes.byref(ti),
                                    ctypes.byref(claim["batch"]) if claim else None, claim["stamp"] if claim else 0,
                                    claim["gen"] if claim else 0,
                                    ctypes.byref(place["batch"]) if place else None, place["stamp"] if place else 0,
                                    place["gen"] if place else 0, self._sweep((claim or place)["B"]), self._words().data_ptr(), self.err_flag.data_ptr(),
                                    _C.current_stream())
        _C.check(rc, "fr_focf_stage")
        if claim:
            claim["stage"] = 1
        if place:
            place["stage"] = 2

    def _words(self):
        if self._row_words is None:
            n = _C.lib().fr_focf_row_words(self.U.n_rows, self.I.n_rows)
            self._row_words = torch.zeros(n, dtype=torch.int64, device=self.device)
        return self._row_words

    def _forget_staged(self):
        """Batches that were claimed but will not be applied (a loop cut short, a batch nobody announced): their words are
        overwritten by the first later batch that touches the row (stamps only grow); their workspaces' counters are not."""
        for e in self._st.values():
            self._ws_dirty.add(e["k"])
        self._st.clear()

    def _staged_forward(self, user, item, rating, sst, B, coming, loss):
        self.U.ensure_state()
        self.I.ensure_state()
        ent = self._st.pop(self._key(user, item), None)
        alive = {self._key(nb[0], nb[1]) for nb in coming[:2]}
        if ent is None or any(k not in alive for k in self._st):
            self._forget_staged()
        if ent is None:
            self._gen_cur = None
            ent = self._st_entry(user, item, sst, rating, 0)
        self._gen_cur = ent["gen"]
        if ent["stage"] < 1:
            self._stage_now(claim=ent)
        if ent["stage"] < 2:
            self._stage_now(place=ent)
        self.ws_cur = ent["k"]
        place = claim = None
        if coming:
            nb = coming[0]
            e1 = self._st.get(self._key(nb[0], nb[1]))
            if e1 is None:
                e1 = self._st[self._key(nb[0], nb[1])] = self._st_entry(nb[0], nb[1], nb[2], nb[3], 1)
            if e1["stage"] < 1:
                self._stage_now(claim=e1)
            if e1["stage"] < 2:
                place = e1
        if len(coming) > 1:
            nb = coming[1]
            e2 = self._st.get(self._key(nb[0], nb[1]))
            if e2 is None:
                e2 = self._st[self._key(nb[0], nb[1])] = self._st_entry(nb[0], nb[1], nb[2], nb[3], 2)
            if e2["stage"] < 1:
                claim = e2
        self._stash = (user, item, rating, sst, B, ent["ws"], (ent["stamp"], ent["gen"]), loss, claim, place)
        self.pending_B = B
        return loss, None

    # --- launches -------------------------------------------------------------------------------------
    def forward(self, user, item, rating, sst, want_pred: bool = False, next_batch=None):
        """fr_focf_forward for the step `step+1`; returns (loss[4] device view, pred or None).
        `next_batch` = (user, item, sst, rating) of the following step, if known, or a list of such tuples for the next
        steps in order (a dataloader's prefetch queue): their index sorts are launched ahead, GROUP batches per launch."""
        B = user.numel()
        flags = 0
        self._check_stamp_gen()
        if self._stash is not None:
            raise _C.FairrecError("calculate_loss twice without optimizer.step() in between (fused step)")
        if (self.staged and self.fused_step and self.defer_loss and self.optimizer is not None and self.objective != 5
                and not want_pred and not self.item_runs and self.U.step == self.I.step and rating is not None):
            coming = []
            if next_batch is not None:
                coming = [next_batch] if isinstance(next_batch[0], torch.Tensor) else [nb for nb in next_batch if nb is not None]
            coming = [nb for nb in coming[:2] if len(nb) >= 4 and nb[3] is not None]
            self.loss_slot = (self.loss_slot + 1) % self.LOSS_SLOTS
            self.hyper.check_step(self.U.step + 1)
            return self._staged_forward(user, item, rating, sst, B, coming, self._loss_views[self.loss_slot])
        if self._st:
            self._forget_staged()
        hit = self._prep.pop(self._key(user, item), None)
        stamp = None
        if hit is not None:
            self.ws_cur, group, stamp = hit
            if not group["joined"]:                              # one join per GROUP of batches, not per step
                torch.cuda.current_stream().wait_event(group["done"])
                group["joined"] = True
            flags = 1                                            # FR_FOCF_PREPARED
        else:
            held = [t[0] for t in (self._prev, self._pipe) if t is not None]       # an unreduced loss / pending item runs
            taken = {v[0] for v in self._prep.values()}
            while any(w is self.ws[self.ws_cur] for w in held) or (held and self.ws_cur in taken):
                self.ws_cur = (self.ws_cur + 1) % self.N_WS
        if self.defer_loss and self.optimizer is not None:
            flags |= 2                                           # FR_FOCF_DEFER_LOSS
        if self.item_runs:
            flags |= 4                                           # FR_FOCF_ITEM_RUNS
        ws = self._workspace(B, self.ws_cur)
        coming = []
        if next_batch is not None:
            coming = [next_batch] if isinstance(next_batch[0], torch.Tensor) else [nb for nb in next_batch if nb is not None]
        if self._prep and (hit is None or len(self._prep) > len(coming)):
            # prepared batches that are not announced any more (a loop that was cut short): forget them, their buffers
            # may be reused.  Checked only when something looks off -- a miss, or more prepared than announced.
            alive = {self._key(nb[0], nb[1]) for nb in coming}
            for k in [k for k in self._prep if k not in alive]:
                torch.cuda.current_stream().wait_event(self._prep.pop(k)[1]["done"])
        # item-complete batches (FOCFDataLoader) take the two-launch step of csrc/focf_runs.hip (the gather, then a workgroup per
        # item run); FAIRREC_FOCF_RUNS=0 sends them through the three-launch chain as before round 4
        fused = (self.fused_step and self.defer_loss and self.optimizer is not None and self.objective != 5 and not want_pred
                 and (not self.item_runs or (self.RUNS and rating is not None)) and self.U.step == self.I.step)
        if self._pipe is not None and not (fused and self.item_runs and self.PIPE):
            self.finish()                   # this step takes another path: the pending item runs first
        # A stamp is the optimizer step at which its batch is applied: the current batch (if nobody prepared it) takes its
        # stamp BEFORE the coming ones take theirs, so that stamps rise in application order -- the sweeper leaves a row to
        # its batch by comparing stamps, and the start order of a step's sweeper tasks is built for the stamped step's slice
        stamp_now = self._next_stamp() if (fused and stamp is None) else None
        if coming and len(self._prep) <= self.LOW_WATER:
            todo = [nb for nb in coming if self._key(nb[0], nb[1]) not in self._prep]
            if todo:
                self.prepare_many(todo[:self.GROUP], ahead=1)
        self.loss_slot = (self.loss_slot + 1) % self.LOSS_SLOTS
        loss = self._loss_views[self.loss_slot]
        self.hyper.check_step(self.U.step + 1)
        if fused:
            # one launch for the whole step, issued by backward_adam(); the index side must be there first
            if stamp is None:
                arr = (_C.FrFocfBatch * 1)(_C.FrFocfBatch(user.data_ptr(), item.data_ptr(),
                                                          _C.ptr(sst if self.objective != 0 else None), B, ws.data_ptr(),
                                                          ws.numel(), rating.data_ptr()))
                stamp = stamp_now
                tu, ti = self.U.c(), self.I.c()
                rc = _C.lib().fr_focf_prepare_step(arr, (ctypes.c_int32 * 1)(stamp), 1, ctypes.byref(tu), ctypes.byref(ti),
                                                   self._sweep(B), self.err_flag.data_ptr(), _C.current_stream())
                _C.check(rc, "fr_focf_prepare_step")
            self._stash = (user, item, rating, sst, B, ws, stamp, loss)
            self._stash_runs = bool(self.item_runs)
            self.pending_B = B
            return loss, None
        pred = torch.empty(B, dtype=torch.float32, device=self.device) if want_pred else None
        return self._forward_chain(user, item, rating, sst, B, flags, ws, loss, pred)

    def _forward_chain(self, user, item, rating, sst, B, flags, ws, loss, pred):
        """fr_focf_forward: the three-launch chain (gather, fairness, [loss]); backward_adam() finishes the step."""
        tu, ti = self.U.c(self.U.step + 1), self.I.c(self.I.step + 1)
        rc = _C.lib().fr_focf_forward(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()),
                                      user.data_ptr(), item.data_ptr(), rating.data_ptr(), _C.ptr(sst), B,
                                      self.objective, self.fair_weight, flags, ws.data_ptr(), ws.numel(),
                                      loss.data_ptr(), _C.ptr(pred), self.err_flag.data_ptr(), _C.current_stream())
        _C.check(rc, "fr_focf_forward")
        self.pending_B = B
        return loss, pred

    def clip_grad_norm(self, max_norm: float, group=None):
        """torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) on the pending batch's (never materialised)
        embedding gradients (fr_focf_clip_grad_norm); returns the device pair (total_norm, clip_coef)."""
        if self.pending_B == 0:
            raise _C.FairrecError("clip_grad_norm without a preceding calculate_loss()")
        if self._pipe is not None:
            self.finish()
        if self._stash is not None:       # the norm needs every gradient row before any update: three-launch chain
            user, item, rating, sst, B, ws, _, loss = self._stash[:8]
            staged = len(self._stash) > 8
            self._stash = None
            if staged:                    # no sorted segments in that workspace, and its stage counters stay behind
                self._ws_dirty.add(self.ws_cur)
            self._forward_chain(user, item, rating, sst, B, 2 if staged else 1 | 2, ws, loss, None)   # [PREPARED |] DEFER_LOSS
        if not hasattr(self, "_clip_out"):
            self._clip_out = torch.zeros(2, dtype=torch.float32, device=self.device)
        tu, ti = self.U.c(self.U.step + 1), self.I.c(self.I.step + 1)
        ws = self.ws[self.ws_cur]
        rc = _C.lib().fr_focf_clip_grad_norm(ctypes.byref(tu), ctypes.byref(ti), self.pending_B, float(max_norm), None,
                                             self._clip_out.data_ptr(), ws.data_ptr(), ws.numel(), _C.current_stream())
        _C.check(rc, "fr_focf_clip_grad_norm")
        return self._clip_out

    def backward_adam(self):
        """loss.backward() + optimizer.step() of the pending batch (fr_focf_backward_adam)."""
        if self.pending_B == 0:
            raise _C.FairrecError("optimizer.step() without a preceding calculate_loss()")
        if self.optimizer is None:
            raise _C.FairrecError("no optimizer bound: build fairrec.optim.FusedLazyAdam(model.hip_engine(), ...)")
        B = self.pending_B
        tu, ti = self.U.c(self.U.step + 1), self.I.c(self.I.step + 1)
        if self._stash is not None and len(self._stash) > 8:
            user, item, rating, sst, B, ws, stamp, loss, claim, place = self._stash
            self._stash = None
            if self._prev is not None and not self._prev[3]:
                self.finish()
            pw, pB, ploss, _ = self._prev if self._prev is not None else (None, 0, None, True)
            rc = _C.lib().fr_focf_step_staged(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()), _C.ptr(sst), B,
                                              self.objective, self.fair_weight, self._sweep(B), stamp[0], stamp[1],
                                              ws.data_ptr(), ws.numel(), _C.ptr(pw), pB, _C.ptr(ploss),
                                              self.loss_acc.data_ptr(), self._words().data_ptr(),
                                              ctypes.byref(claim["batch"]) if claim else None, claim["stamp"] if claim else 0,
                                              claim["gen"] if claim else 0,
                                              ctypes.byref(place["batch"]) if place else None, place["stamp"] if place else 0,
                                              place["gen"] if place else 0, self.err_flag.data_ptr(), _C.current_stream())
            _C.check(rc, "fr_focf_step_staged")
            if claim:
                claim["stage"] = 1
            if place:
                place["stage"] = 2
            self._prev = (ws, B, loss, True)
            self._keep = (user, item, rating, sst)
        elif self._stash is not None:
            user, item, rating, sst, B, ws, stamp, loss = self._stash
            self._stash = None
            if self._prev is not None and self._prev[3]:
                self.finish()
            pw, pB, ploss, _ = self._prev if self._prev is not None else (None, 0, None, False)
            if getattr(self, "_stash_runs", False) and self.PIPE:
                fin = self._pipe
                own_u, own_i = self._own_arrays()
                rc = _C.lib().fr_focf_step_runs_pipe(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()),
                                                     user.data_ptr(), item.data_ptr(), rating.data_ptr(), _C.ptr(sst), B,
                                                     self.objective, self.fair_weight, self._sweep(B), stamp, ws.data_ptr(),
                                                     ws.numel(), _C.ptr(fin[0]) if fin else None, fin[1] if fin else 0,
                                                     fin[2] if fin else 0, _C.ptr(pw), pB, _C.ptr(ploss), self.loss_acc.data_ptr(),
                                                     own_u.data_ptr(), own_i.data_ptr(), self.err_flag.data_ptr(),
                                                     _C.current_stream())
                _C.check(rc, "fr_focf_step_runs_pipe")
                self._pipe = (ws, B, self.U.step + 1, loss, (user, item, rating, sst))
                self._prev = (fin[0], fin[1], fin[3], False) if fin else None     # finished in this launch: its loss comes next
                self._keep = (user, item, rating, sst)
                self.U.step += 1
                self.I.step += 1
                self.U._dirty = self.I._dirty = True
                self.pending_B = 0
                self.backward_seen = False
                return
            if self._pipe is not None:
                self.finish()
                pw, pB, ploss, _ = self._prev if self._prev is not None else (None, 0, None, False)
            if getattr(self, "_stash_runs", False):
                rc = _C.lib().fr_focf_step_runs(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()),
                                                user.data_ptr(), item.data_ptr(), rating.data_ptr(), _C.ptr(sst), B,
                                                self.objective, self.fair_weight, self._sweep(B), stamp, ws.data_ptr(),
                                                ws.numel(), _C.ptr(pw), pB, _C.ptr(ploss), self.loss_acc.data_ptr(),
                                                self.err_flag.data_ptr(), _C.current_stream())
                _C.check(rc, "fr_focf_step_runs")
            else:
                rc = _C.lib().fr_focf_step(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()),
                                           user.data_ptr(), item.data_ptr(), rating.data_ptr(), _C.ptr(sst), B,
                                           self.objective, self.fair_weight, self._sweep(B), stamp, ws.data_ptr(), ws.numel(),
                                           loss.data_ptr(), _C.ptr(pw), pB, _C.ptr(ploss), self.loss_acc.data_ptr(),
                                           self.err_flag.data_ptr(), _C.current_stream())
                _C.check(rc, "fr_focf_step")
            self._prev = (ws, B, loss, False)
            self._keep = (user, item, rating, sst)
        else:
            rc = _C.lib().fr_focf_backward_adam(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()), B,
                                                self._sweep(B), self.ws[self.ws_cur].data_ptr(),
                                                self.ws[self.ws_cur].numel(), _C.current_stream())
            _C.check(rc, "fr_focf_backward_adam")
            if self.defer_loss:         # keep the running total the one-launch step keeps on the device (and its NaN record)
                _C.check(_C.lib().fr_loss_accumulate(self._loss_views[self.loss_slot].data_ptr(), 3, self.loss_acc.data_ptr(),
                                                     _C.current_stream()), "fr_loss_accumulate")
        self.U.step += 1
        self.I.step += 1
        self.U._dirty = self.I._dirty = True
        self.pending_B = 0
        self.backward_seen = False

    # --- the step loop in the library (fr_focf_steps_many) ----------------------------------------------
    RING = 8          # workspaces a run of steps cycles through (any four consecutive batches need four different ones)

    def can_step_many(self) -> bool:
        """A run of batches can go through the library's own step loop: the staged one-launch step (fr_focf_steps_many) or,
        for item-complete batches, the pipelined item-run step (fr_focf_runs_many) applies -- what forward() checks per batch
        -- and nothing has to happen between a batch's loss and its update (clip_grad_norm)."""
        if not (self.fused_step and self.defer_loss and self.optimizer is not None and self.objective != 5
                and self.U.step == self.I.step and not getattr(self.optimizer, "clip", None)):
            return False
        return bool(self.RUNS and self.PIPE) if self.item_runs else bool(self.staged)

    def _run_sizes(self, total, sizes):
        import numpy as np
        if isinstance(sizes, int):
            n = (total + sizes - 1) // sizes
            size = np.full(n, sizes, dtype=np.int64)
            if n:
                size[-1] = total - sizes * (n - 1)
        else:
            size = np.asarray(sizes, dtype=np.int64)
        if size.shape[0] and (int(size.sum()) != total or int(size.min()) < 1 or int(size.max()) > _C.FR_SORT_MAX):
            raise _C.FairrecError("steps_many: batch sizes do not add up to the columns (or a batch outside 1..FR_SORT_MAX)")
        return size

    def _run_batches(self, user, item, rating, sst, size, ring):
        """fr_focf_batch records of a run whose columns sit back to back, workspaces taken round-robin from `ring`."""
        import numpy as np
        n = size.shape[0]
        start = np.concatenate(([0], np.cumsum(size)[:-1]))
        arr = np.zeros(n, dtype=_BATCH_DTYPE)
        arr["user"] = user.data_ptr() + 8 * start
        arr["item"] = item.data_ptr() + 8 * start
        arr["rating"] = rating.data_ptr() + 4 * start
        if self.objective != 0:
            arr["sst"] = sst.data_ptr() + 4 * start
        arr["B"] = size
        slot = np.arange(n) % len(ring)
        arr["ws"] = np.array([w.data_ptr() for _, w in ring], dtype=np.uint64)[slot]
        arr["ws_bytes"] = np.array([w.numel() for _, w in ring], dtype=np.uint64)[slot]
        return arr

    def _ring(self, Bmax, count, held):
        ring = []
        for k in range(self.N_WS):
            if len(ring) == count:
                break
            ws = self._workspace(Bmax, k)
            if any(ws is h for h in held):
                continue
            if k in self._ws_dirty:
                ws.zero_()
                self._ws_dirty.discard(k)
            ring.append((k, ws))
        if len(ring) < count:
            raise _C.FairrecError("steps_many: not enough free workspaces")
        return ring

    def _runs_many(self, user, item, rating, sst, size):
        """fr_focf_runs_many: the pipelined item-run steps of a run of item-complete batches, their sorted prepare one group
        ahead on the library's side stream."""
        from ...optim import _used_on_side_stream
        n = size.shape[0]
        if self._prev is not None and self._prev[3]:
            self.finish()
        self._forget_staged()
        self._gen_cur = None
        if self._prep:
            self._join_prepare()
        fin, prev = self._pipe, self._prev
        held = [t[0] for t in (fin, prev) if t is not None]
        ring = self._ring(int(size.max()), 2 * self.PER_LAUNCH + 2, held)
        arr = self._run_batches(user, item, rating, sst, size, ring)
        for t in (user, item, rating, sst) + tuple(w for _, w in ring):   # the prepare launches read / write them on the
            _used_on_side_stream(t)                                       # library's side stream
        step0 = self.U.step + 1
        self.hyper.check_step(step0 + n - 1)
        stamp0 = self._next_stamp()
        self._stamp_last = stamp0 + n - 1
        slot0 = (self.loss_slot + 1) % self.LOSS_SLOTS
        own_u, own_i = self._own_arrays()
        tu, ti = self.U.c(step0), self.I.c(step0)
        rc = _C.lib().fr_focf_runs_many(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()), arr.ctypes.data, n,
                                        self.objective, self.fair_weight, self._sweep(int(size.max())), stamp0,
                                        _C.ptr(fin[0]) if fin else None, fin[1] if fin else 0, fin[2] if fin else 0,
                                        _C.ptr(fin[3]) if fin else None, _C.ptr(prev[0]) if prev else None,
                                        prev[1] if prev else 0, _C.ptr(prev[2]) if prev else None, self.loss_ring.data_ptr(),
                                        self.LOSS_SLOTS, slot0, self.loss_acc.data_ptr(), own_u.data_ptr(), own_i.data_ptr(),
                                        self.err_flag.data_ptr(), _C.current_stream())
        if rc:
            self._pipe = self._prev = None
        _C.check(rc, "fr_focf_runs_many")
        view = lambda k: self._loss_views[(slot0 + k) % self.LOSS_SLOTS]
        last = ring[(n - 1) % len(ring)]
        self._pipe = (last[1], int(size[-1]), step0 + n - 1, view(n - 1), (user, item, rating, sst))
        if n >= 2:
            self._prev = (ring[(n - 2) % len(ring)][1], int(size[-2]), view(n - 2), False)
        else:
            self._prev = (fin[0], fin[1], fin[3], False) if fin else None
        self.loss_slot = (slot0 + n - 1) % self.LOSS_SLOTS
        self.ws_cur = last[0]
        self._keep = (user, item, rating, sst)
        self.U.step += n
        self.I.step += n
        self.U._dirty = self.I._dirty = True
        self.pending_B = 0
        return n

    def steps_many(self, user, item, rating, sst, sizes):
        """`calculate_loss` + `optimizer.step()` for a run of batches in ONE library call (the step loop of
        trainer.py:181-196 issued by the library): fr_focf_steps_many -- one staged launch per batch, the coming two batches'
        index work riding in each -- or, with `item_runs` (item-complete batches), fr_focf_runs_many.  The columns hold the
        batches back to back: `sizes` = rows per batch, an int (every batch that size, the last one whatever is left) or a
        sequence.  Losses go to the engine's running total (`loss_acc`), as with `defer_loss`; same launches and same bits
        as the per-batch calls."""
        if not self.can_step_many():
            raise _C.FairrecError("steps_many: neither the staged nor the pipelined item-run step applies to this engine state")
        if self._stash is not None:
            raise _C.FairrecError("steps_many with a calculate_loss() pending")
        size = self._run_sizes(user.numel(), sizes)
        n = size.shape[0]
        if n == 0:
            return 0
        self._check_stamp_gen()
        self.U.ensure_state()
        self.I.ensure_state()
        if self.item_runs:
            return self._runs_many(user, item, rating, sst, size)
        if self._pipe is not None or (self._prev is not None and not self._prev[3]):
            self.finish()
        self._forget_staged()                       # nothing may be on its way through the stages
        self._gen_cur = None
        if self._prep:
            self._join_prepare()
        # the ring of workspaces, all as large as the largest batch of the run; the one a pending loss sits in is skipped
        Bmax = int(size.max())
        ring = self._ring(Bmax, self.RING, [self._prev[0]] if self._prev is not None else [])
        arr = self._run_batches(user, item, rating, sst, size, ring)
        step0 = self.U.step + 1
        self.hyper.check_step(step0 + n - 1)
        stamp0 = self._next_stamp()
        self._stamp_last = stamp0 + n - 1
        pw, pB, ploss, _ = self._prev if self._prev is not None else (None, 0, None, True)
        slot0 = (self.loss_slot + 1) % self.LOSS_SLOTS
        tu, ti = self.U.c(step0), self.I.c(step0)
        rc = _C.lib().fr_focf_steps_many(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()),
                                         arr.ctypes.data, n, self.objective, self.fair_weight, self._sweep(Bmax), stamp0, 0,
                                         _C.ptr(pw), pB, _C.ptr(ploss), self.loss_ring.data_ptr(), self.LOSS_SLOTS, slot0,
                                         self.loss_acc.data_ptr(), self._words().data_ptr(), self.err_flag.data_ptr(),
                                         _C.current_stream())
        if rc:                                      # launches may have been issued: every workspace of the ring is suspect
            self._ws_dirty.update(k for k, _ in ring)
            self._prev = None
        _C.check(rc, "fr_focf_steps_many")
        self.loss_slot = (slot0 + n - 1) % self.LOSS_SLOTS
        k_last, ws_last = ring[(n - 1) % len(ring)]
        self.ws_cur = k_last
        self._prev = (ws_last, int(size[-1]), self._loss_views[self.loss_slot], True)
        self._keep = (user, item, rating, sst)      # the launches read the columns: keep them alive until the next call
        self.U.step += n
        self.I.step += n
        self.U._dirty = self.I._dirty = True
        self.pending_B = 0
        return n

    def reset_loss_acc(self):
        """Start a new running loss total (`loss_acc`: sums of loss, mse, fair over the steps taken with defer_loss on)."""
        self.finish()
        self.loss_acc.zero_()

    def finish(self):
        """Reduce the loss of the last fused step (fr_focf_step_finish): its loss slot and `loss_acc` are complete on the
        stream after this.  A later fused step would have done it in passing."""
        if self._pipe is not None:           # the item runs of the last pipelined step (and the loss of the one before it)
            fin, self._pipe = self._pipe, None
            pw, pB, ploss, _ = self._prev if self._prev is not None else (None, 0, None, False)
            own_u, own_i = self._own_arrays()
            tu, ti = self.U.c(fin[2]), self.I.c(fin[2])
            rc = _C.lib().fr_focf_step_runs_pipe(ctypes.byref(tu), ctypes.byref(ti), ctypes.byref(self.hyper.c()), None, None,
                                                 None, None, 0, self.objective, self.fair_weight, 0, 0, None, 0,
                                                 fin[0].data_ptr(), fin[1], fin[2], _C.ptr(pw), pB, _C.ptr(ploss),
                                                 self.loss_acc.data_ptr(), own_u.data_ptr(), own_i.data_ptr(),
                                                 self.err_flag.data_ptr(), _C.current_stream())
            _C.check(rc, "fr_focf_step_runs_pipe")
            self._prev = (fin[0], fin[1], fin[3], False)
        if self._prev is not None:
            ws, B, loss, staged = self._prev
            self._prev = None
            fn = _C.lib().fr_focf_step_finish_staged if staged else _C.lib().fr_focf_step_finish
            rc = fn(ws.data_ptr(), ws.numel(), B, self.U.dim, self.objective, self.fair_weight, loss.data_ptr(),
                    self.loss_acc.data_ptr(), _C.current_stream())
            _C.check(rc, "fr_focf_step_finish")

    def _own_arrays(self):
        if self._own is None:
            self._own = (torch.zeros(2 * self.U.n_rows, dtype=torch.int32, device=self.device),
                         torch.zeros(2 * self.I.n_rows, dtype=torch.int32, device=self.device))
        return self._own

    def predict(self, user, item):
        self._join_prepare()
        self.finish()
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
        self.finish()
        self.U.flush(self.hyper)
        self.I.flush(self.hyper)

    def check_device_errors(self, word=None):
        """Host sync (none when the caller read the error word itself and passes it): raise what the reference would have
        raised eagerly (IndexError)."""
        self.finish()
        e = int(self.err_flag.item()) if word is None else int(word)
        if e:
            self.err_flag.zero_()
            msgs = []
            if e & _C.DEV_ERR_INDEX_RANGE:
                msgs.append("index out of range in embedding gather")
            if e & _C.DEV_ERR_SST_GROUPS:
                msgs.append("a batch must hold exactly the 1..2 sensitive groups the objective expects")
            if e & _C.DEV_ERR_PIPE_WAIT:
                raise _C.FairrecError("pipelined item-run step: a row of the previous batch was never published "
                                      "(FAIRREC_FOCF_PIPE=0 selects the two-launch step)")
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

    PREFETCH = FocfEngine.LOW_WATER + FocfEngine.GROUP     # batches a trainer may announce ahead

    def hint_next_batch(self, *interactions):
        """Optional trainer hook: the batches that will follow the next `calculate_loss`, in order (none at the epoch
        end).  Their index sorts are launched ahead, several batches per launch."""
        old = getattr(self, '_cols_cache', {})
        cache = {}
        for x in interactions:
            if x is not None:
                cache[id(x)] = old.get(id(x)) or (x, self._cols(x))      # the queue moves by one batch per step
        self._cols_cache = cache
        self._next_cols = [c for _, c in cache.values()] or None

    def calculate_loss(self, interaction):
        eng = self.hip_engine()
        hit = getattr(self, '_cols_cache', {}).get(id(interaction))      # announced earlier: converted already
        u, i, r, s = hit[1] if hit is not None and hit[0] is interaction else self._cols(interaction)
        nxt = getattr(self, '_next_cols', None)
        self._next_cols = None
        loss, _ = eng.forward(u, i, r, s, next_batch=[(c[0], c[1], c[3], c[2]) for c in nxt] if nxt else None)
        if torch.is_grad_enabled():
            return _LossHandle.apply(loss[0], eng, self.user_embedding_layer.weight, self.item_embedding_layer.weight)
        return loss[0]

    def train_steps_ready(self):
        """Whether `train_steps` would take a run of batches now (asked once per epoch, before the loader shuffles)."""
        return self.hip_engine().can_step_many()

    def train_steps(self, interaction, sizes):
        """Optional trainer hook: `calculate_loss` + `optimizer.step()` for a RUN of batches held back to back in one
        Interaction (`sizes`: rows per batch -- an int, or a sequence for ragged batches), the per-batch loop issued by the
        library (FocfEngine.steps_many).  Returns the number of steps taken, or None when the engine's state calls for
        the per-batch path (nonparity, clip_grad_norm, item-complete batches, a graph capture ...): the caller then
        iterates the batches itself."""
        eng = self.hip_engine()
        if not eng.can_step_many():
            return None
        if not isinstance(sizes, int) and len(sizes) and int(max(sizes)) > _C.FR_SORT_MAX:
            return None       # a batch no step takes: the per-batch path reports it (FairrecError), not this hook
        u, i, r, s = self._cols(interaction)
        return eng.steps_many(u, i, r, s, sizes)

    def predict(self, interaction):
        u, i, _, _ = self._cols(interaction, need_targets=False)
        return self.hip_engine().predict(u, i)

    def full_sort_predict(self, interaction):
        """focf.py:171-178: clamp(U[user] @ I^T, 0, max_rating) / max_rating over ALL items, flattened [b * n_items].  The
        users' rows come caught-up from the lazy table (fr_table_gather), the item table is flushed once (every row is read),
        the product runs on the library's fp32-MFMA kernel (fr_linear_fwd: Y = X W^T with W = the [n_items, D] item table)."""
        eng = self.hip_engine()
        eng.finish()
        eng.I.flush(eng.hyper)
        user = interaction[self.USER_ID].to(eng.device, torch.int64).contiguous()
        ue = eng.U.gather(eng.hyper, user, eng.err_flag)
        W = self.item_embedding_layer.weight.data
        scores = torch.empty((user.numel(), W.shape[0]), dtype=torch.float32, device=eng.device)
        _C.check(_C.lib().fr_linear_fwd(ue.data_ptr(), ue.shape[1], None, 0, None, 1.0, W.data_ptr(), None, user.numel(),
                                        W.shape[0], 0, scores.data_ptr(), _C.current_stream()), "fr_linear_fwd")
        return (torch.clamp(scores, min=0., max=eng.max_rating) / eng.max_rating).view(-1)

    def state_dict(self, *args, **kwargs):
        if self._engine is not None:
            self._engine.flush()   # checkpoints must see every row at the current optimizer step
        return super().state_dict(*args, **kwargs)
