"""Generic training engine for models assembled from lazy embedding tables + dense layers.

It is the `engine` object fairrec.optim.FusedLazyAdam drives: `backward_adam()` applies, for the batch whose
`loss.backward()` just ran, fr_table_apply_grad on every table that was looked up (duplicate-summed gradient +
Adam + sweeper slice) and fr_adam_dense on every dense parameter that received a gradient -- the work of
`optimizer.step()` at trainer.py:196.  Per-tensor step counters follow torch.optim.Adam: a tensor without a
gradient in a step is skipped entirely (SURVEY.md §7 hard part 1).
"""
from __future__ import annotations

import ctypes
import os
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
        self.step_dev = None     # optional device-resident counter (graph mode), as LazyTable.step_dev


class GenericEngine:
    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _C.FairrecError("the training hot path runs only on a ROCm device; there is no CPU fallback")
        _C.lib()
        self._tables: Dict[str, LazyTable] = {}
        self._weights: Dict[str, torch.nn.Parameter] = {}
        self._dense: Dict[str, DenseState] = {}
        self._group: Dict[str, Optional[str]] = {}      # entry -> optimizer group ('filter' / 'dis' / None)
        self._hyper_of: Dict[str, AdamHyper] = {}       # entry -> hyper-parameters of the optimizer that owns it
        self.hyper = AdamHyper(device=self.device, cap=1)
        self.optimizer = None
        self.sweep_period: Optional[int] = None
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._group_version: Dict[Optional[str], int] = {}   # optimizer group -> steps taken (eager or graph replays)
        self._counters = None           # graph mode: one device int32 per entry (tables, then dense tensors)
        self._counter_slot: Dict[str, int] = {}
        self._advance_idx: Dict[tuple, torch.Tensor] = {}
        self._seg_src: dict = {}        # (id tensor, rows) -> the table whose workspace holds that id list's segments, this step only

    # --- graph mode: step counters on the device ----------------------------------------------------------

    def enable_graph_mode(self):
        """Move every step counter to device memory, so that a training step captured in a hipGraph advances them on
        replay (fairrec/graph.py).  From here on the device values are authoritative; `sync_steps()` refreshes the
        host mirrors."""
        if self._counters is not None:
            return
        names = list(self._tables) + list(self._dense)
        self._counters = torch.zeros(len(names), dtype=torch.int32, device=self.device)
        for k, name in enumerate(names):
            self._counter_slot[name] = k
            view = self._counters[k:k + 1]
            if name in self._tables:
                self._tables[name].ensure_state()
                self._tables[name].attach_step_counter(view)
            else:
                view.fill_(self._dense[name].step)
                self._dense[name].step_dev = view

    def sync_steps(self):
        if self._counters is None:
            return
        host = self._counters.cpu().tolist()
        for name, k in self._counter_slot.items():
            if name in self._tables:
                self._tables[name].step = host[k]
            else:
                self._dense[name].step = host[k]

    def _advance(self, names):
        """counter += 1 for the entries stepped in this optimizer step (one launch; the index tensor is cached per set
        so that nothing is uploaded inside a captured step)."""
        if self._counters is None or not names:
            return
        key = tuple(sorted(self._counter_slot[n] for n in names))
        inc = self._advance_idx.get(key)
        if inc is None:     # a 0/1 vector over all counters: one add, no fill and no index launch
            inc = torch.zeros(self._counters.numel(), dtype=torch.int32)
            inc[list(key)] = 1
            inc = self._advance_idx[key] = inc.to(self.device)
        self._counters += inc

    # --- registration ---------------------------------------------------------------------------------
    def add_table(self, name: str, weight: torch.nn.Parameter, trainable: bool = True, group=None) -> LazyTable:
        t = LazyTable(weight.data, trainable=trainable)
        self._tables[name] = t
        self._weights[name] = weight
        self._group[name] = group
        return t

    def add_dense(self, name: str, p: torch.nn.Parameter, group=None):
        self._dense[name] = DenseState(p)
        self._group[name] = group

    def _owned(self, name, group):
        return group is None or self._group.get(name) == group

    def tables(self, group=None) -> Dict[str, LazyTable]:
        return {k: t for k, t in self._tables.items() if t.trainable and self._owned(k, group)}

    def bind_optimizer(self, opt, sweep_period, group=None):
        """`group` = the subset of entries this optimizer owns (PFCN: optimizer_filter / optimizer_dis,
        trainer.py:1201-1212); None = everything (the single default optimizer)."""
        self.optimizer, self.hyper, self.sweep_period = opt, opt.hyper, sweep_period
        for name in list(self._tables) + list(self._dense):
            if self._owned(name, group):
                self._hyper_of[name] = opt.hyper
        for t in self._tables.values():
            t.ensure_state()

    def _hyper(self, name) -> AdamHyper:
        return self._hyper_of.get(name, self.hyper)

    # --- forward helpers --------------------------------------------------------------------------------
    def lookup(self, name: str, idx: torch.Tensor) -> torch.Tensor:
        """Differentiable rows = table[idx] (training) or a read-only gather (frozen table / no grad)."""
        t = self._tables[name]
        idx = idx.to(self.device, torch.int64).contiguous()
        if t.trainable and torch.is_grad_enabled():
            # a table with as many rows already looked up with this very id tensor in this step has the sorted segments
            key = (idx.data_ptr(), idx.numel(), t.n_rows)
            src = self._seg_src.get(key)
            if src is None or src is t:
                self._seg_src[key] = t
                src = None
            return LazyLookup.apply(self._weights[name], t, self._hyper(name), idx, self.err_flag, src)
        return t.gather(self._hyper(name), idx, self.err_flag)

    def lookup_pair(self, name_a: str, idx_a: torch.Tensor, name_b: str, idx_b: torch.Tensor):
        """Two lookups of one step (the row-sharded engine packs their exchanges into one buffer per direction; here a frozen
        table's read-only gather rides in the training gather's launch: fr_table_lookup_pair)."""
        ta, tb = self._tables[name_a], self._tables[name_b]
        if torch.is_grad_enabled() and tb.trainable and not ta.trainable and ta.dim == tb.dim:
            idx_a = idx_a.to(self.device, torch.int64).contiguous()
            idx_b = idx_b.to(self.device, torch.int64).contiguous()
            self._seg_src[(idx_b.data_ptr(), idx_b.numel(), tb.n_rows)] = tb
            got = []
            rows_b = LazyLookup.apply(self._weights[name_b], tb, self._hyper(name_b), idx_b, self.err_flag, None,
                                      (ta, self._hyper(name_a), idx_a, got))
            return got[0], rows_b
        return self.lookup(name_a, idx_a), self.lookup(name_b, idx_b)

    def batch_segments(self, name: str):
        """The object a loss kernel reads the sorted segments of the batch ids of table `name` from (`._ws`, `.dim`):
        here the table itself, whose workspace the preceding lookup filled."""
        return self._tables[name]

    # --- optimizer.step() -------------------------------------------------------------------------------
    def zero_grad(self, group=None):
        """optimizer.zero_grad() of the optimizer that owns `group`.  The gradients of the OTHER groups' dense parameters
        are dropped as well: the reference lets them pile up untouched between their own phases (a
        discriminator collects gradients all through the filter pass and nobody reads them), but here a stale `.grad`
        tensor is a hazard -- a step captured as a hipGraph accumulates into it IN PLACE, i.e. the graph keeps the address
        of a tensor that the owning optimizer's next zero_grad() frees, and every later replay writes through it."""
        for name, d in self._dense.items():
            d.p.grad = None
        self._seg_src = {}
        for name, t in self._tables.items():
            if self._owned(name, group):
                t._grad_rows = None

    # Parameters the reference model holds in plain dicts (pfcn_biasedmf.py:110-142 filter_layer / dis_layer_dict,
    # fairgo_pmf.py:141-157) are not in `model.parameters()`: torch.nn.utils.clip_grad_norm_(model.parameters(), ...) of the
    # reference loop (trainer.py:731-732, :925-926) neither measures nor scales their gradients.
    NOT_MODEL_PARAMETERS = ("filter.", "dis.")

    def clip_grad_norm(self, max_norm: float, group=None):
        """clip_grad_norm_(model.parameters(), max_norm) between backward() and step(), on the gradients as they are held
        here: a table's gradient is the parked [M, D] rows of the batch (duplicates summed in ascending batch position
        before squaring -- the dense gradient's rows), a dense parameter's its `.grad`.  Norm of the per-tensor norms,
        coefficient max_norm / (total + 1e-6) clamped to 1 and applied to every measured gradient, as torch does.
        Consequences of the reference's call that are kept: a discriminator step of PFCN / FairGo is not affected at all
        (its parameters are not in model.parameters()), a filter step only through the embedding / bias / registered-MLP
        gradients.  Returns the device scalar total_norm.  Eager only (data-dependent shapes): a GraphedStep whose
        optimizer clips runs its steps eagerly."""
        norms, held = [], []
        for name, t in self._tables.items():
            if name.startswith(self.NOT_MODEL_PARAMETERS) or not t.trainable or t._grad_rows is None:
                continue
            idx, g = t._keep.reshape(-1), t._grad_rows
            order = torch.argsort(idx, stable=True)
            _, counts = torch.unique_consecutive(idx[order], return_counts=True)
            dense_rows = torch.segment_reduce(g[order].contiguous(), "sum", lengths=counts, axis=0)
            norms.append(torch.linalg.vector_norm(dense_rows))
            held.append(("table", t))
        for name, d in self._dense.items():
            if name.startswith(self.NOT_MODEL_PARAMETERS) or d.p.grad is None:
                continue
            norms.append(torch.linalg.vector_norm(d.p.grad))
            held.append(("dense", d))
        if not norms:
            return torch.zeros((), device=self.device)
        total = torch.linalg.vector_norm(torch.stack(norms))
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        for kind, x in held:
            if kind == "table":
                x._grad_rows = x._grad_rows * coef
            else:
                x.p.grad.mul_(coef)
        return total

    def note_stepped(self, group=None):
        """An optimizer step of `group` was issued (here, or as a replay of a captured step: fairrec/graph.py)."""
        self._group_version[group] = self._group_version.get(group, 0) + 1
        # a REPLAYED step runs none of the host code that marks a table's rows as behind the optimizer step: without this a
        # flush that follows an earlier flush with only graph replays in between was skipped, and a whole-table reader (a
        # checkpoint per epoch, FairGo's finetune stage reading the pretrained tables) got rows short of their last
        # zero-gradient steps (found in round 5 when the FairGo trainer began to checkpoint after EVERY pretrain epoch, as the
        # reference does: tests/test_fairgo_hip.py::test_fairgo_trainer_pretrain_then_finetune, graph against eager twin)
        for name, t in self._tables.items():
            if t.trainable and self._owned(name, group):
                t._dirty = True

    def group_version(self, group=None) -> int:
        return self._group_version.get(group, 0)

    def backward_adam(self, group=None):
        self.note_stepped(group)
        self._seg_src = {}
        stepped = []
        ready = []
        for name, t in self._tables.items():
            if not (t.trainable and t._pending is not None):
                continue
            if not self._owned(name, group) or t._grad_rows is None:
                # not this optimizer's tensor, or no gradient reached it: torch would skip it (no step, no decay)
                t._pending = None
                t._grad_rows = None
                continue
            M = t._pending[0]
            s = self.sweep_period if self.sweep_period is not None else t.default_sweep(M)
            ready.append((name, t, s))
            stepped.append(name)
        # two tables of one width and one optimizer share a launch (a user table next to its item table, their two bias
        # columns): the shorter one's update and sweep slice run beside the longer one's (fr_table_apply_grad_two)
        pair_ok = os.environ.get("FAIRREC_APPLY_SEPARATE") is None
        while ready:
            name, t, s = ready.pop(0)
            mate = next((k for k, (nb, tb, sb) in enumerate(ready)
                         if pair_ok and tb.dim == t.dim and self._hyper(nb) is self._hyper(name) and t._pending[1] is not None
                         and tb._pending[1] is not None and (t.step_dev is None) == (tb.step_dev is None)), None)
            if mate is None:
                t.apply_grad(self._hyper(name), None, s)
            else:
                nb, tb, sb = ready.pop(mate)
                LazyTable.apply_grad_two(t, tb, self._hyper(name), s, sb)
        st = _C.current_stream()
        by_hyper = {}                     # all dense tensors of one optimizer in one launch (fr_adam_dense_multi)
        for name, d in self._dense.items():
            g = d.p.grad
            if g is None or not self._owned(name, group):
                continue
            d.step += 1
            h = self._hyper(name)
            h.check_step(d.step)
            by_hyper.setdefault(id(h), (h, []))[1].append((d, g.contiguous()))
            stepped.append(name)
        for h, items in by_hyper.values():
            descs = (_C.FrDenseDesc * len(items))()
            for k, (d, g) in enumerate(items):
                if d.step_dev is not None:      # effective step = device counter + 1
                    descs[k] = _C.FrDenseDesc(d.p.data.data_ptr(), g.data_ptr(), d.m.data_ptr(), d.v.data_ptr(),
                                              d.p.numel(), 1, d.step_dev.data_ptr())
                else:
                    descs[k] = _C.FrDenseDesc(d.p.data.data_ptr(), g.data_ptr(), d.m.data_ptr(), d.v.data_ptr(),
                                              d.p.numel(), d.step, None)
            _C.check(_C.lib().fr_adam_dense_multi(descs, len(items), ctypes.byref(h.c()), st), "fr_adam_dense_multi")
            for d, _ in items:
                d.p.grad = None
        self._advance(stepped)

    def flush(self):
        for name, t in self._tables.items():
            t.flush(self._hyper(name))

    def check_device_errors(self, word=None):
        e = int(self.err_flag.item()) if word is None else int(word)
        if e:
            self.err_flag.zero_()
            raise IndexError(f"device error word {e} (1: row id out of range, 2: unexpected sensitive groups)")

    # --- torch.optim.Adam-shaped state for checkpoints ----------------------------------------------------
    def dense_state(self, group=None):
        self.sync_steps()
        return {k: {"step": torch.tensor(float(d.step)), "exp_avg": d.m, "exp_avg_sq": d.v}
                for k, d in self._dense.items() if self._owned(k, group)}

    def load_dense_state(self, sd):
        for k, st in sd.items():
            d = self._dense[k]
            d.step = int(st["step"])
            if d.step_dev is not None:
                d.step_dev.fill_(d.step)
            d.m.copy_(st["exp_avg"])
            d.v.copy_(st["exp_avg_sq"])
