"""Generic training engine for models assembled from lazy embedding tables + dense layers.

It is the `engine` object fairrec.optim.FusedLazyAdam drives: `backward_adam()` applies, for the batch whose
`loss.backward()` just ran, fr_table_apply_grad on every table that was looked up (duplicate-summed gradient +
Adam + sweeper slice) and fr_adam_dense on every dense parameter that received a gradient -- the work of
`optimizer.step()` at trainer.py:196.  Per-tensor step counters follow torch.optim.Adam: a tensor without a
gradient in a step is skipped entirely (SURVEY.md §7 hard part 1).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import _C
from .optim import AdamHyper, LazyLookup, LazyTable


class DenseState:
    def __init__(self, p: torch.nn.Parameter):
        self.p = p
        self.m = torch.zeros_like(p.data)
        self.v = torch.zeros_like(p.data)
        self.step = 0


class GenericEngine:
    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _C.FairrecError("the training hot path runs only on a ROCm device; there is no CPU fallback")
        _C.lib()
        self._tables: Dict[str, LazyTable] = {}
        self._weights: Dict[str, torch.nn.Parameter] = {}
        self._dense: Dict[str, DenseState] = {}
        self.hyper = AdamHyper(device=self.device, cap=1)
        self.optimizer = None
        self.sweep_period: Optional[int] = None
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=self.device)

    # --- registration ---------------------------------------------------------------------------------
    def add_table(self, name: str, weight: torch.nn.Parameter, trainable: bool = True) -> LazyTable:
        t = LazyTable(weight.data, trainable=trainable)
        self._tables[name] = t
        self._weights[name] = weight
        return t

    def add_dense(self, name: str, p: torch.nn.Parameter):
        self._dense[name] = DenseState(p)

    def tables(self) -> Dict[str, LazyTable]:
        return {k: t for k, t in self._tables.items() if t.trainable}

    def bind_optimizer(self, opt, sweep_period):
        self.optimizer, self.hyper, self.sweep_period = opt, opt.hyper, sweep_period
        for t in self._tables.values():
            t.ensure_state()

    # --- forward helpers --------------------------------------------------------------------------------
    def lookup(self, name: str, idx: torch.Tensor) -> torch.Tensor:
        """Differentiable rows = table[idx] (training) or a read-only gather (frozen table / no grad)."""
        t = self._tables[name]
        idx = idx.to(self.device, torch.int64).contiguous()
        if t.trainable and torch.is_grad_enabled():
            return LazyLookup.apply(self._weights[name], t, self.hyper, idx, self.err_flag)
        return t.gather(self.hyper, idx, self.err_flag)

    # --- optimizer.step() -------------------------------------------------------------------------------
    def backward_adam(self):
        for t in self._tables.values():
            if t.trainable and t._pending is not None:
                if t._grad_rows is None:      # looked up but no gradient reached it: torch would skip the tensor
                    t._pending = None
                    continue
                M = t._pending[0]
                s = self.sweep_period if self.sweep_period is not None else t.default_sweep(M)
                t.apply_grad(self.hyper, None, s)
        st = _C.current_stream()
        for d in self._dense.values():
            g = d.p.grad
            if g is None:
                continue
            d.step += 1
            self.hyper.check_step(d.step)
            g = g.contiguous()
            _C.check(_C.lib().fr_adam_dense(d.p.data.data_ptr(), g.data_ptr(), d.m.data_ptr(), d.v.data_ptr(),
                                            d.p.numel(), ctypes.byref(self.hyper.c()), d.step, st), "fr_adam_dense")
            d.p.grad = None

    def flush(self):
        for t in self._tables.values():
            t.flush(self.hyper)

    def check_device_errors(self):
        e = int(self.err_flag.item())
        if e:
            self.err_flag.zero_()
            raise IndexError(f"device error word {e} (1: row id out of range, 2: unexpected sensitive groups)")

    # --- torch.optim.Adam-shaped state for checkpoints ----------------------------------------------------
    def dense_state(self):
        return {k: {"step": torch.tensor(float(d.step)), "exp_avg": d.m, "exp_avg_sq": d.v} for k, d in self._dense.items()}

    def load_dense_state(self, sd):
        for k, st in sd.items():
            d = self._dense[k]
            d.step = int(st["step"])
            d.m.copy_(st["exp_avg"])
            d.v.copy_(st["exp_avg_sq"])
