"""One training step as a hipGraph: capture once, replay per batch.

The generic models (NFCF, PFCN_*, FairGo_*) run tens to ~150 small launches per optimizer step; launched eagerly from
Python a step is bound by launch overhead, not by the kernels.  `GraphedStep` captures

    optimizer.zero_grad(); loss = loss_fn(batch, *args); loss.backward(); optimizer.step()

-- HIP kernels called through the C ABI, torch's autograd glue and the dropout-mask draws alike -- into one
torch.cuda.CUDAGraph on static input tensors and replays it for every later batch of the same shape.  What makes a
captured optimizer step replayable is that nothing the kernels need changes on the host between steps: the step counters
of the lazy tables and dense tensors live in device memory (`GenericEngine.enable_graph_mode`, fr_table.step_dev) and are
advanced by a kernel inside the graph; the per-step Adam scalars are a device table indexed by that counter.

Usage (what fairrec.trainer does when `graph_train_step: True`):
    gs = GraphedStep(engine, optimizer, loss_fn)
    for batch in loader: loss = gs(batch, *args)        # first calls run eagerly, then capture, then replay
The returned loss is a device tensor that the next call overwrites; read it (`.item()`) or accumulate it before that.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import os

import torch

from .data.interaction import Interaction


def quiesce_rccl(timeout_s: float = 10.0):
    """Wait until a stream capture may begin in a process that holds an RCCL process group.  ProcessGroupNCCL's watchdog
    thread polls the events of earlier EAGER collectives from another thread until it has seen each of them complete and
    RETIRED it; a capture that begins while it still holds one makes that poll fail (hipErrorCapturedEvent) and the
    watchdog aborts the process.  Every rank first makes its earlier collectives complete on the device
    (barrier + synchronize); then the wait is EVENT-BASED where torch's flight recorder runs (`TORCH_FR_BUFFER_SIZE` > 0,
    set before the process group is created -- bench.py does): poll its records until every collective of this process is
    marked `retired`, which is exactly "the watchdog has dropped it" (measured: one to two polling passes, 0.1-0.4 s).
    Without the recorder there is nothing to observe: a fixed wait of FAIRREC_RCCL_QUIESCE_S seconds (default 2.0, twenty
    watchdog passes), as before round 5."""
    import os
    import time
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"):
        return "no process group"
    dist.barrier()
    torch.cuda.synchronize()
    try:
        import pickle
        from torch._C._distributed_c10d import _dump_nccl_trace
        t0 = time.perf_counter()
        seen = False
        while time.perf_counter() - t0 < timeout_s:
            entries = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False)).get("entries", [])
            if not entries:
                break                                   # the recorder is off (or was reset): nothing to observe
            seen = True
            if all(e.get("retired", False) for e in entries):
                return "retired"
            time.sleep(0.01)
        if seen:
            return "timeout"
    except Exception:                                   # a torch build without the recorder's Python entry
        pass
    time.sleep(float(os.environ.get("FAIRREC_RCCL_QUIESCE_S", "2.0")))
    return "timed wait"


class GraphedStep:
    def __init__(self, engine, optimizer, loss_fn: Callable, eager_steps: int = 2):
        self.engine, self.optimizer, self.loss_fn = engine, optimizer, loss_fn
        self.eager_left = eager_steps          # real training steps before capture (lazy initialisation happens here)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static: Optional[Dict[str, torch.Tensor]] = None
        self.static_args: Tuple = ()
        self.loss: Optional[torch.Tensor] = None
        engine.enable_graph_mode()

    def _eager(self, inter: Interaction, args):
        self.optimizer.zero_grad()
        loss = self.loss_fn(inter, *args)
        from . import _C
        loss.backward(_C.one(loss.device) if loss.dim() == 0 and loss.dtype == torch.float32 else None)
        self.optimizer.step()
        return loss.detach()

    @staticmethod
    def _quiesce_rccl():
        quiesce_rccl()

    def _refresh(self, inter: Interaction):
        """The batch into the static tensors the graph reads: ONE launch for all columns (a copy launch per column is ~5 us
        of device time each, as much as a small kernel of the step)."""
        import ctypes
        cols = [(v, self.static[k]) for k, v in inter.interaction.items()]
        if len(cols) > 16 or any(not v.is_contiguous() or v.device != s.device for v, s in cols):
            for v, s in cols:
                s.copy_(v, non_blocking=True)
            return
        n = len(cols)
        src = (ctypes.c_void_p * n)(*[v.data_ptr() for v, _ in cols])
        dst = (ctypes.c_void_p * n)(*[s.data_ptr() for _, s in cols])
        nb = (ctypes.c_int64 * n)(*[v.numel() * v.element_size() for v, _ in cols])
        from . import _C
        _C.check(_C.lib().fr_copy_many(src, dst, nb, n, _C.current_stream()), "fr_copy_many")

    def _signature(self, inter: Interaction):
        return tuple((k, tuple(v.shape), v.dtype) for k, v in inter.interaction.items())

    def __call__(self, inter: Interaction, *args):
        dev = self.engine.device
        inter = inter.to(dev)
        if self.eager_left > 0 or getattr(self.optimizer, "clip", None):
            # (clip_grad_norm measures the batch's distinct rows: data-dependent shapes, not capturable)
            self.eager_left = max(0, self.eager_left - 1)
            return self._eager(inter, args)
        if self.graph is None:
            # capture on this batch: copy it into static tensors, record the step, then replay it once (capture does not
            # execute anything), so every batch is trained on exactly once
            self.static = {k: v.clone() for k, v in inter.interaction.items()}
            self.sig, self.static_args = self._signature(inter), args
            static_inter = Interaction(self.static)
            torch.cuda.synchronize()
            self._quiesce_rccl()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self.loss = self._eager(static_inter, args)
            except RuntimeError as e:      # an operation that refuses capture: this step object stays eager from here on
                import traceback
                import warnings
                msg = str(e).lower()
                # only what the runtime says about the CAPTURE itself is survivable; an argument error of the library, a
                # device error word or an out-of-memory inside the step is a bug or a limit of its own and stays loud
                if not any(w in msg for w in ("captur", "hiperrorstream", "cudaerrorstream")):
                    raise
                # the failed capture pass ran the host side of a step without the device doing it: drop what it parked
                self.optimizer.zero_grad()
                for t in getattr(self.engine, "_tables", {}).values():
                    t._pending = None
                where = "".join(traceback.format_tb(e.__traceback__)[-4:]) if os.environ.get("FAIRREC_GRAPH_DEBUG") else ""
                warnings.warn(f"hipGraph capture of the training step failed ({e}); running it eagerly\n{where}")
                torch.cuda.synchronize()
                self.engine.sync_steps()
                self.eager_left = 1 << 60
                return self._eager(inter, args)
            self.graph = g
            # the host ran the optimizer bookkeeping once during capture without the device doing the step: take the
            # host mirrors back from the device counters
            self.engine.sync_steps()
        elif self._signature(inter) != self.sig or args != self.static_args:
            return self._eager(inter, args)        # odd-sized last batch, other attribute subset: eager
        else:
            self._refresh(inter)
        self.graph.replay()
        if hasattr(self.engine, "note_stepped"):      # a replay is an optimizer step the host code did not run
            self.engine.note_stepped(getattr(self.optimizer, "group", None))
        return self.loss
