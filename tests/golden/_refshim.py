"""Environment shims that let the reference (RecBole-FairRec, /root/reference) be imported in the
build container so that golden fixtures can be generated from it.

This file is OUR code (test infrastructure). It is only ever used by the `gen_*.py` fixture
generators in this directory, which run in the build container; nothing on the GPU box imports it
(/root/reference does not exist there).  Recipe: SURVEY.md Appendix A.
"""
import functools
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FAIRREC_REFERENCE_ROOT", "/root/reference")


def install():
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    import numpy as np
    import torch

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # numpy aliases removed in numpy>=1.24 but used by the reference
    for name, typ in (("float", float), ("int", int), ("bool", bool), ("object", object)):
        if not hasattr(np, name):
            setattr(np, name, typ)

    # logging / tensorboard stubs (not installed in the image)
    if "colorlog" not in sys.modules:
        import logging
        m = types.ModuleType("colorlog")

        class ColoredFormatter(logging.Formatter):
            def __init__(self, fmt=None, datefmt=None, log_colors=None, **kw):
                super().__init__((fmt or "").replace("%(log_color)s", ""), datefmt)
        m.ColoredFormatter = ColoredFormatter
        sys.modules["colorlog"] = m
    if "colorama" not in sys.modules:
        m = types.ModuleType("colorama")
        m.init = lambda **kw: None
        sys.modules["colorama"] = m
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        tb = types.ModuleType("tensorboard")
        tub = types.ModuleType("torch.utils.tensorboard")

        class SummaryWriter:
            def __init__(self, *a, **kw):
                pass

            def add_scalar(self, *a, **kw):
                pass

            def add_hparams(self, *a, **kw):
                pass
        tub.SummaryWriter = SummaryWriter
        sys.modules.setdefault("tensorboard", tb)
        sys.modules["torch.utils.tensorboard"] = tub
        torch.utils.tensorboard = tub

    if not getattr(torch.load, "_fairrec_patched", False):
        patched = functools.partial(torch.load, weights_only=False)
        patched._fairrec_patched = True
        torch.load = patched
    try:
        import scipy.sparse as sp
        if not hasattr(sp.dok_matrix, "_update"):
            sp.dok_matrix._update = lambda self, d: self._dict.update(d)
    except Exception:
        pass
    del sys.argv[1:]


class float64_reference:
    """Context in which the reference's own model code runs in float64: torch's default dtype is float64 (parameters,
    buffers and every tensor the reference creates without a dtype) and `Tensor.float()` -- which the reference calls on
    label columns right before a loss (pfcn_biasedmf.py, nfcf.py) -- widens instead of narrowing.  Used by the golden
    generators for the <case>_f64.npz companions: the SAME reference code, the same recorded batches / masks / initial
    state, executed in near-exact arithmetic."""

    def __enter__(self):
        import torch
        self._torch = torch
        self._dtype = torch.get_default_dtype()
        self._float = torch.Tensor.float
        torch.set_default_dtype(torch.float64)
        torch.Tensor.float = lambda t, *a, **k: t.double()
        return self

    def __exit__(self, *exc):
        self._torch.Tensor.float = self._float
        self._torch.set_default_dtype(self._dtype)
        return False
