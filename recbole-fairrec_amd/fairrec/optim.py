"""Host side of the lazily-applied dense Adam that backs every trainable embedding table.

Replaces `torch.optim.Adam(params, lr=..., weight_decay=...)` as built by the reference at
recbole/trainer/trainer.py:114-153 (learner 'adam').  The arithmetic is torch's `_single_tensor_adam`
(coupled L2, bias corrections computed in double on the host); what changes is WHEN a row is updated:
the HIP kernels replay the steps a row missed when it is next read, so the per-step HBM traffic is
proportional to the batch, not to the table.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _C


def adam_step_scalars(lr: float, beta1: float, beta2: float, cap: int, weight_decay: float = 0.0,
                      eps: float = 1e-8) -> np.ndarray:
    """float32[4*(cap+1)]: entry j = (lr / (1 - beta1**j), 1 / sqrt(1 - beta2**j), A_j, B_j), computed in Python
    double exactly like torch/optim/adam.py (`bias_correction1 = 1 - beta1 ** step`, `step_size = lr /
    bias_correction1`, `bias_correction2_sqrt = sqrt(bias_correction2)`) and rounded to fp32 once.
    A_j, B_j (include/fairrec_hip.h) serve the scaled replay of zero-data-gradient steps; 0 when weight_decay = 0."""
    out = np.zeros(4 * (cap + 1), dtype=np.float32)
    k1 = (1.0 - beta1) * weight_decay
    k2 = (1.0 - beta2) * weight_decay * weight_decay
    for j in range(1, cap + 1):
        bc1 = 1.0 - beta1 ** j
        bc2 = 1.0 - beta2 ** j
        ss = lr / bc1
        ib = 1.0 / math.sqrt(bc2)
        out[4 * j] = ss
        out[4 * j + 1] = ib
        if weight_decay != 0.0:
            out[4 * j + 2] = math.sqrt(k2) * ib / (ss * k1)
            out[4 * j + 3] = eps / (ss * k1)
    return out


def _used_on_side_stream(t: torch.Tensor):
    """fr_table_gather_train sorts the id list on the library's OWN stream, beside the gather: the sort reads `idx` and writes
    the table's workspace until fr_table_apply_grad (or fr_table_join) orders the caller's stream behind it.  torch's
    allocator recycles a freed tensor's memory in the order of the streams it knows the tensor was used on, so it is told:
    without this, a forward pass that no backward pass follows (a loss computed and dropped, a table that goes out of
    scope) can leave a sort writing into memory the allocator has already handed to somebody else."""
    if t is None or not t.is_cuda or torch.cuda.is_current_stream_capturing():
        return          # (inside a capture the library sorts in line)
    side = _C.side_stream(t.device)
    if side is not None:
        t.record_stream(side)


class AdamHyper:
    """lr / weight_decay / betas / eps + the device table of per-step scalars (fr_adam in the C ABI)."""

    def __init__(self, lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, device="cuda", cap=32768):
        self.lr, self.weight_decay, self.betas, self.eps = float(lr), float(weight_decay), tuple(betas), float(eps)
        self.device = torch.device(device)
        while True:
            tab = adam_step_scalars(self.lr, self.betas[0], self.betas[1], cap, self.weight_decay, self.eps)
            # steps beyond `cap` reuse entry `cap`: only valid once both scalars stopped changing in fp32
            if (tab[4 * cap] == np.float32(self.lr) and tab[4 * cap + 1] == np.float32(1.0)) or cap >= (1 << 22):
                break
            cap *= 2
        self.cap = cap
        self.saturated = bool(tab[4 * cap] == np.float32(self.lr) and tab[4 * cap + 1] == np.float32(1.0))
        self.host_scalars = tab
        self.scalars = torch.from_numpy(tab).to(self.device) if self.device.type == "cuda" else None
        self._c = _C.FrAdam(_C.ptr(self.scalars), cap, 0, self.weight_decay, self.betas[0], self.betas[1], self.eps)

    def c(self) -> "_C.FrAdam":
        return self._c

    def check_step(self, step: int):
        if step > self.cap and not self.saturated:
            raise _C.FairrecError(f"Adam step {step} exceeds the scalar table ({self.cap}) and the bias "
                                  "corrections have not saturated; raise `cap`")


class LazyTable:
    """An nn.Embedding weight with lazily-applied Adam state (fr_table in the C ABI).

    `weight` stays the model's nn.Parameter (same storage), so `state_dict()` keys and shapes are the
    reference's; `flush()` must run before anything reads the whole table.
    """

    def __init__(self, weight: torch.Tensor, trainable: bool = True):
        assert weight.dim() == 2 and weight.dtype == torch.float32 and weight.is_contiguous()
        self.weight = weight
        self.n_rows, self.dim = weight.shape
        self.trainable = trainable
        self.step = 0            # optimizer steps applied so far (torch: state['step'])
        self.m = self.v = self.last = self.stamp = None
        self.stamp_gen = 0       # bumped whenever the stamps are reset: look-ahead stamping done before is void
        self._ws = None
        self._pending = None
        self._keep = None
        self._grad_rows = None
        self._dirty = False      # some row may be behind `step` (set by an update, cleared by flush)
        # optional device-resident step counter (int32 [1] view): once attached the device value is authoritative and
        # `step` is a host mirror that replays of a captured hipGraph do not advance (see `sync_step`)
        self.step_dev: Optional[torch.Tensor] = None
        self._cstruct, self._cstruct_key, self._cstruct_turn = None, None, 0

    def ensure_state(self):
        dev = self.weight.device
        if self.last is None:
            self.last = torch.zeros(self.n_rows, dtype=torch.int32, device=dev)
            self.stamp = torch.zeros(self.n_rows, dtype=torch.int32, device=dev)
        if self.m is None:
            if self.trainable:
                self.m = torch.zeros_like(self.weight)
                self.v = torch.zeros_like(self.weight)
            else:  # frozen table: never replayed (last == step == 0), m/v never touched
                self.m = self.v = self.weight

    def c(self, step: Optional[int] = None) -> "_C.FrTable":
        self.ensure_state()
        w = self.weight.data if isinstance(self.weight, torch.nn.Parameter) else self.weight
        step = self.step if step is None else step
        # the struct is rebuilt only when a buffer moved; per call just the step field changes (callers pass it to the
        # library with byref() before asking for another one)
        key = (w.data_ptr(), self.m.data_ptr(), self.step_dev.data_ptr() if self.step_dev is not None else 0)
        if self._cstruct_key != key:
            self._cstruct = [_C.FrTable(w.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.last.data_ptr(),
                                        self.stamp.data_ptr(), self.n_rows, self.dim, 0,
                                        self.step_dev.data_ptr() if self.step_dev is not None else None) for _ in range(2)]
            self._cstruct_key, self._cstruct_turn = key, 0
        self._cstruct_turn ^= 1                   # two structs alternate, so two consecutive c() results stay distinct
        t = self._cstruct[self._cstruct_turn]
        # with a device counter attached the kernels add it to this constant offset
        t.step = step - self.step if self.step_dev is not None else step
        return t

    def attach_step_counter(self, counter: torch.Tensor):
        """`counter`: int32 [1] device view that from now on holds this table's step count."""
        counter.fill_(self.step)
        self.step_dev = counter

    def sync_step(self):
        if self.step_dev is not None:
            self.step = int(self.step_dev.item())

    def flush(self, hyper: AdamHyper):
        """Bring every row up to `self.step` (fr_table_flush)."""
        if (self.step == 0 and self.step_dev is None) or not self.trainable or not self._dirty:
            return
        self._dirty = False
        t = self.c()
        _C.check(_C.lib().fr_table_flush(ctypes.byref(t), ctypes.byref(hyper.c()), _C.current_stream()),
                 "fr_table_flush")

    def gather(self, hyper: AdamHyper, idx: torch.Tensor, err_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Rows as of `self.step` without modifying the table (fr_table_gather)."""
        idx = idx.contiguous()
        out = torch.empty((idx.numel(), self.dim), dtype=torch.float32, device=self.weight.device)
        t = self.c()
        _C.check(_C.lib().fr_table_gather(ctypes.byref(t), ctypes.byref(hyper.c()), idx.data_ptr(), idx.numel(),
                                          out.data_ptr(), _C.ptr(err_flag), _C.current_stream()), "fr_table_gather")
        return out

    # --- generic training pair (models with an MLP between the embeddings and the loss) -----------------
    def gather_train(self, hyper: AdamHyper, idx: torch.Tensor, err_flag: Optional[torch.Tensor] = None,
                     segments_of: Optional["LazyTable"] = None) -> torch.Tensor:
        """fr_table_gather_train for step `self.step + 1`: caught-up rows [M, dim]; remembers the batch so that
        `apply_grad` can finish the step.  `segments_of` = a table with as many rows that was looked up with the SAME id
        tensor in this step: its sorted segments are copied instead of sorting again (a bias table next to its embedding
        table)."""
        idx = idx.contiguous()
        M = idx.numel()
        rows = torch.empty((M, self.dim), dtype=torch.float32, device=self.weight.device)
        self.gather_train_into(hyper, idx.data_ptr(), M, rows.data_ptr(), err_flag=err_flag, segments_of=segments_of)
        self._pending = (M, rows)
        self._keep = idx
        _used_on_side_stream(idx)
        return rows

    def gather_train_with(self, hyper: AdamHyper, idx: torch.Tensor, ro: "LazyTable", ro_hyper: AdamHyper, ro_idx: torch.Tensor,
                          err_flag: Optional[torch.Tensor] = None):
        """`gather_train(idx)` on this table and `ro.gather(ro_idx)` on a read-only one in ONE launch (fr_table_lookup_pair: the
        id sort rides in it as well).  Returns (rows, ro_rows)."""
        idx, ro_idx = idx.contiguous(), ro_idx.contiguous()
        M = idx.numel()
        dev = self.weight.device
        rows = torch.empty((M, self.dim), dtype=torch.float32, device=dev)
        ro_rows = torch.empty((ro_idx.numel(), ro.dim), dtype=torch.float32, device=dev)
        need = _C.lib().fr_table_train_workspace_bytes(M, self.dim)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        t, r = self.c(self.step + 1), ro.c()
        hyper.check_step(self.step + 1)
        _C.check(_C.lib().fr_table_lookup_pair(ctypes.byref(t), ctypes.byref(hyper.c()), idx.data_ptr(), M, rows.data_ptr(),
                                               self._ws.data_ptr(), self._ws.numel(), ctypes.byref(r), ctypes.byref(ro_hyper.c()),
                                               ro_idx.data_ptr(), ro_idx.numel(), ro_rows.data_ptr(), _C.ptr(err_flag),
                                               _C.current_stream()), "fr_table_lookup_pair")
        _used_on_side_stream(self._ws)
        _used_on_side_stream(idx)
        self._pending = (M, rows)
        self._keep = idx
        self._grad_rows = None
        return rows, ro_rows

    def gather_train_into(self, hyper: AdamHyper, idx_ptr: int, M: int, rows_ptr: int, chunk: int = 0, stride: int = 0,
                          err_flag: Optional[torch.Tensor] = None, segments_of: Optional["LazyTable"] = None):
        """The same on raw device pointers with a slot layout (fairrec_hip.h): M ids read from / M rows written into
        exchange buffers in place.  The caller keeps the buffers alive until `apply_grad_from`."""
        need = _C.lib().fr_table_train_workspace_bytes(M, self.dim)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.weight.device)
        t = self.c(self.step + 1)
        hyper.check_step(self.step + 1)
        if segments_of is not None and segments_of._ws is not None and segments_of.n_rows == self.n_rows:
            n = _C.lib().fr_table_segments_bytes(M)
            # (the other table's sort runs on the library's side stream: wait for it before reading its result)
            _C.check(_C.lib().fr_table_join(segments_of._ws.data_ptr(), _C.current_stream()), "fr_table_join")
            self._ws[:n].copy_(segments_of._ws[:n])
            _C.check(_C.lib().fr_table_gather_train_prepared(ctypes.byref(t), ctypes.byref(hyper.c()), idx_ptr, M, chunk,
                                                             stride, rows_ptr, self._ws.data_ptr(), self._ws.numel(),
                                                             _C.ptr(err_flag), _C.current_stream()),
                     "fr_table_gather_train_prepared")
        else:
            _C.check(_C.lib().fr_table_gather_train(ctypes.byref(t), ctypes.byref(hyper.c()), idx_ptr, M, chunk, stride,
                                                    rows_ptr, self._ws.data_ptr(), self._ws.numel(), _C.ptr(err_flag),
                                                    _C.current_stream()), "fr_table_gather_train")
            _used_on_side_stream(self._ws)
        self._pending = (M, None)
        self._grad_rows = None

    def apply_grad(self, hyper: AdamHyper, grad_rows: Optional[torch.Tensor] = None, sweep_period: int = 0):
        """fr_table_apply_grad: duplicate-summed gradient + Adam step `self.step + 1` + sweeper slice."""
        if self._pending is None:
            raise _C.FairrecError("apply_grad without a preceding gather_train")
        M, rows = self._pending
        g = grad_rows if grad_rows is not None else self._grad_rows
        if g is None:
            raise _C.FairrecError("no gradient reached the gathered rows (loss.backward() not called?)")
        g = g.contiguous()
        assert g.shape == rows.shape and g.dtype == torch.float32
        self.apply_grad_from(hyper, M, rows.data_ptr(), g.data_ptr(), sweep_period)

    def apply_grad_from(self, hyper: AdamHyper, M: int, rows_ptr: int, grad_ptr: int, sweep_period: int = 0,
                        chunk: int = 0, stride: int = 0):
        """fr_table_apply_grad on raw device pointers with a slot layout (the rows of the preceding
        `gather_train_into` and the gradient rows that came back for them)."""
        if self._pending is None or self._pending[0] != M:
            raise _C.FairrecError("apply_grad without a matching gather_train")
        t = self.c(self.step + 1)
        _C.check(_C.lib().fr_table_apply_grad(ctypes.byref(t), ctypes.byref(hyper.c()), M, chunk, stride, rows_ptr,
                                              grad_ptr, int(sweep_period), self._ws.data_ptr(), self._ws.numel(),
                                              _C.current_stream()), "fr_table_apply_grad")
        self.step += 1
        self._dirty = True
        self._pending = None
        self._grad_rows = None
        self._keep = None

    # --- two tables of a step in one launch each (the user and item table of the row-sharded step) ------
    @staticmethod
    def sort_pair(ta: "LazyTable", tb: "LazyTable", idx_a: int, idx_b: int, M: int, ws_a: torch.Tensor, ws_b: torch.Tensor,
                  chunk: int = 0, stride: int = 0, err_flag: Optional[torch.Tensor] = None):
        """fr_table_sort2 into the given workspaces (one step ahead, on the current -- side -- stream)."""
        _C.check(_C.lib().fr_table_sort2(idx_a, idx_b, ta.n_rows, tb.n_rows, M, chunk, stride, ta.dim, ws_a.data_ptr(),
                                         ws_b.data_ptr(), min(ws_a.numel(), ws_b.numel()), _C.ptr(err_flag),
                                         _C.current_stream()), "fr_table_sort2")

    @staticmethod
    def gather_train_pair(ta: "LazyTable", tb: "LazyTable", hyper: AdamHyper, idx_a: int, idx_b: int, M: int, rows_a: int,
                          rows_b: int, chunk: int = 0, stride: int = 0, err_flag: Optional[torch.Tensor] = None,
                          prepared: bool = False):
        assert ta.dim == tb.dim and ta.step == tb.step
        need = _C.lib().fr_table_train_workspace_bytes(M, ta.dim)
        for t in (ta, tb):
            if t._ws is None or t._ws.numel() < need:
                assert not prepared, "prepared workspaces must be installed as table._ws by the caller"
                t._ws = torch.empty(need, dtype=torch.uint8, device=t.weight.device)
        hyper.check_step(ta.step + 1)
        ca, cb = ta.c(ta.step + 1), tb.c(tb.step + 1)
        if not prepared:
            _used_on_side_stream(ta._ws)
            _used_on_side_stream(tb._ws)
        _C.check(_C.lib().fr_table_gather_train2(ctypes.byref(ca), ctypes.byref(cb), ctypes.byref(hyper.c()), idx_a, idx_b,
                                                 M, chunk, stride, rows_a, rows_b, 1 if prepared else 0,
                                                 ta._ws.data_ptr(), tb._ws.data_ptr(),
                                                 min(ta._ws.numel(), tb._ws.numel()), _C.ptr(err_flag),
                                                 _C.current_stream()), "fr_table_gather_train2")
        for t in (ta, tb):
            t._pending = (M, None)
            t._grad_rows = None

    @staticmethod
    def apply_grad_pair(ta: "LazyTable", tb: "LazyTable", hyper: AdamHyper, M: int, rows_a: int, grad_a: int, rows_b: int,
                        grad_b: int, sweep_a: int, sweep_b: int, chunk: int = 0, stride: int = 0):
        for t in (ta, tb):
            if t._pending is None or t._pending[0] != M:
                raise _C.FairrecError("apply_grad without a matching gather_train")
        ca, cb = ta.c(ta.step + 1), tb.c(tb.step + 1)
        _C.check(_C.lib().fr_table_apply_grad2(ctypes.byref(ca), ctypes.byref(cb), ctypes.byref(hyper.c()), M, chunk,
                                               stride, rows_a, grad_a, rows_b, grad_b, int(sweep_a), int(sweep_b),
                                               ta._ws.data_ptr(), tb._ws.data_ptr(),
                                               min(ta._ws.numel(), tb._ws.numel()), _C.current_stream()),
                 "fr_table_apply_grad2")
        for t in (ta, tb):
            t.step += 1
            t._dirty = True
            t._pending = None
            t._grad_rows = None
            t._keep = None

    NARROW_SWEEP = int(os.environ.get("FAIRREC_NARROW_SWEEP", 64))

    @staticmethod
    def apply_grad_two(ta: "LazyTable", tb: "LazyTable", hyper: AdamHyper, sweep_a: int, sweep_b: int):
        """`ta.apply_grad(...)` and `tb.apply_grad(...)` -- two tables of one width, each with the rows its own gather_train
        parked and the gradient rows autograd left -- in ONE launch (fr_table_apply_grad_two)."""
        args = []
        for t in (ta, tb):
            if t._pending is None or t._pending[1] is None:
                raise _C.FairrecError("apply_grad without a preceding gather_train")
            if t._grad_rows is None:
                raise _C.FairrecError("no gradient reached the gathered rows (loss.backward() not called?)")
            M, rows = t._pending
            g = t._grad_rows.contiguous()
            assert g.shape == rows.shape and g.dtype == torch.float32
            args.append((M, rows, g))
        ca, cb = ta.c(ta.step + 1), tb.c(tb.step + 1)
        (Ma, ra, ga), (Mb, rb, gb) = args
        _C.check(_C.lib().fr_table_apply_grad_two(ctypes.byref(ca), ctypes.byref(cb), ctypes.byref(hyper.c()), Ma, Mb, ra.data_ptr(),
                                                  ga.data_ptr(), rb.data_ptr(), gb.data_ptr(), int(sweep_a), int(sweep_b),
                                                  ta._ws.data_ptr(), ta._ws.numel(), tb._ws.data_ptr(), tb._ws.numel(),
                                                  _C.current_stream()), "fr_table_apply_grad_two")
        for t in (ta, tb):
            t.step += 1
            t._dirty = True
            t._pending = None
            t._grad_rows = None
            t._keep = None

    def default_sweep(self, M: int) -> int:
        """Sweep about M rows per step, so that no row is more than n_rows / M steps stale.  A one-column table (a bias) is
        swept 64 rows per wave: at that period a 10 M-row bias is 128 waves, each a serial chain of 1221 replayed steps --
        60 us of exposed latency per step for 2 us of arithmetic (profiles/README.md, round 4) -- so its period is capped:
        more, shorter chains, the same row-steps in all (any period gives the same values: the replay is exact)."""
        s = max(8, math.ceil(self.n_rows / max(M, 1)))
        return min(s, max(8, self.NARROW_SWEEP)) if self.dim == 1 else s

    # --- interchange with torch.optim.Adam.state_dict() (trainer.py:221-240 checkpoints) ---------------
    def adam_state(self, hyper: AdamHyper) -> Dict[str, torch.Tensor]:
        self.sync_step()
        self.flush(hyper)
        return {"step": torch.tensor(float(self.step)), "exp_avg": self.m, "exp_avg_sq": self.v}

    def load_adam_state(self, state: Dict[str, torch.Tensor]):
        self.ensure_state()
        self.step = int(state["step"])
        self.m.copy_(state["exp_avg"])
        self.v.copy_(state["exp_avg_sq"])
        self.last.fill_(self.step)
        self.stamp.zero_()          # stamps name the step of the batch that owns a row: none after a reload
        self.stamp_gen += 1
        if self.step_dev is not None:
            self.step_dev.fill_(self.step)
        self._dirty = False


class LazyLookup(torch.autograd.Function):
    """rows = table[idx] as a differentiable torch tensor: forward = fr_table_gather_train, backward parks
    dLoss/drows in the table; the optimizer's step() then runs fr_table_apply_grad.  The dense [N, D] gradient
    of nn.Embedding is never built."""

    @staticmethod
    def forward(ctx, weight, table, hyper, idx, err_flag, segments_of=None, ro=None):
        """`ro` = (read-only table, its hyper-parameters, its ids, a list that receives its rows): a frozen table's gather in
        the same launch (LazyTable.gather_train_with)."""
        ctx.table = table
        if ro is not None:
            rows, ro_rows = table.gather_train_with(hyper, idx, ro[0], ro[1], ro[2], err_flag)
            ro[3].append(ro_rows)
            return rows
        return table.gather_train(hyper, idx, err_flag, segments_of)

    @staticmethod
    def backward(ctx, grad_rows):
        t = ctx.table
        t._grad_rows = grad_rows if t._grad_rows is None else t._grad_rows + grad_rows
        return None, None, None, None, None, None, None


class FusedLazyAdam:
    """Drop-in for the `optimizer` object the reference Trainer drives (zero_grad / step / state_dict).

    The embedding gradient is never materialised: `engine.backward_adam()` does loss.backward()'s embedding part
    + optimizer.step() for the batch of the preceding `calculate_loss`.  Dense parameters (MLPs, biases)
    registered with the engine go through fr_adam_dense.  `group` restricts the optimizer to a subset of the
    engine's tensors (PFCN's optimizer_filter / optimizer_dis, trainer.py:1201-1212).
    """

    def __init__(self, engine, lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, sweep_period=None, group=None,
                 clip_grad_norm=None):
        self.engine = engine
        self.group = group
        # config `clip_grad_norm` = kwargs of torch.nn.utils.clip_grad_norm_, which the reference's loop calls between
        # backward() and step() (trainer.py:194-195); the gradient only exists inside step() here, so it is clipped there
        self.clip = dict(clip_grad_norm) if clip_grad_norm else None
        if self.clip:
            if not hasattr(engine, "clip_grad_norm"):
                raise NotImplementedError("clip_grad_norm: this model's engine has no global-gradient-norm pass yet")
            if float(self.clip.get("norm_type", 2)) != 2.0:
                raise NotImplementedError("clip_grad_norm: only norm_type 2 is on the device path")
        self.hyper = AdamHyper(lr, weight_decay, betas, eps, device=engine.device)
        self.defaults = dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)
        if group is None:
            engine.bind_optimizer(self, sweep_period)
        else:
            engine.bind_optimizer(self, sweep_period, group)

    def zero_grad(self, set_to_none: bool = True):
        if hasattr(self.engine, "zero_grad"):
            self.engine.zero_grad(self.group) if self.group is not None else self.engine.zero_grad()

    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure-based step is not supported by the fused path")
        if self.clip:
            self.last_grad_norm = self.engine.clip_grad_norm(float(self.clip["max_norm"]), self.group)
        if self.group is not None:
            self.engine.backward_adam(self.group)
        else:
            self.engine.backward_adam()

    def _tables(self):
        return self.engine.tables(self.group) if self.group is not None else self.engine.tables()

    def state_dict(self, param_names=None):
        """torch.optim.Adam's state_dict content (`exp_avg`, `exp_avg_sq`, `step` per parameter, flushed to the current
        step).  Keys: parameter NAMES by default; with `param_names` (the names of the parameters in the order the
        reference hands them to its optimizer, i.e. `[n for n, _ in model.named_parameters()]`) torch's own layout --
        integer indices and `param_groups[0]['params'] = [0, 1, ...]` -- which a reference `optimizer.load_state_dict`
        accepts as it is (trainer.py:221-240, :258-284)."""
        state = {name: t.adam_state(self.hyper) for name, t in self._tables().items()}
        if hasattr(self.engine, "dense_state"):
            state.update(self.engine.dense_state(self.group) if self.group is not None else self.engine.dense_state())
        if param_names is not None:
            idx = {n: k for k, n in enumerate(param_names)}
            missing = [n for n in state if n not in idx]
            if missing:
                raise KeyError(f"state_dict(param_names=...): no index for {missing}")
            state = {idx[n]: st for n, st in state.items()}
            # torch lists EVERY parameter the optimizer was built on, in order, and keeps state only for those that were
            # stepped (a frozen table -- NFCF's user table after reset_params, FairGo's tables in the finetune stage -- has
            # an index and no state); load_state_dict maps saved ids to parameters by POSITION and checks the group length
            groups = dict(self.defaults, params=list(range(len(param_names))), amsgrad=False, maximize=False, foreach=None,
                          capturable=False, differentiable=False, fused=None)
            return {"state": state, "param_groups": [groups]}
        return {"state": state, "param_groups": [dict(self.defaults, params=list(state.keys()))]}

    def load_state_dict(self, sd, param_names=None):
        """Accepts this class's name-keyed layout and torch.optim.Adam's integer-keyed one (a checkpoint written by the
        reference); the latter needs `param_names`, the parameter names in the reference optimizer's order."""
        tables = self._tables()
        dense = {}
        state = sd["state"]
        if state and all(isinstance(k, int) for k in state):
            if param_names is None:
                raise KeyError("optimizer state with torch's integer parameter indices: pass param_names "
                               "([n for n, _ in model.named_parameters()]) to name them")
            state = {param_names[k]: st for k, st in state.items()}
        for name, st in state.items():
            if name in tables:
                tables[name].load_adam_state(st)
            else:
                dense[name] = st
        if dense:
            self.engine.load_dense_state(dense)
