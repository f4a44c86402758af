"""Run a GPU test with NaN guard bands around every float tensor the Python layer allocates on the device (torch.empty /
empty_like / zeros / zeros_like): a kernel that reads past the end of (or before) a tensor and lets that value reach a result
-- even multiplied by zero -- then fails every time.  usage: python scratch/dbg_guard.py <pytest args>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import torch
GUARD = int(os.environ.get("GUARD", "2048"))
_empty, _zeros = torch.empty, torch.zeros
_empty_like, _zeros_like = torch.empty_like, torch.zeros_like

def _size(args, kw):
    if "size" in kw:
        return tuple(kw["size"])
    if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
        return tuple(args[0])
    return tuple(int(a) for a in args)

def _guarded(shape, dtype, device, fill):
    n = 1
    for s in shape:
        n *= int(s)
    buf = _empty(n + 2 * GUARD, dtype=dtype, device=device)
    buf.fill_(float("nan"))
    mid = buf[GUARD:GUARD + n]
    if fill is not None:
        mid.fill_(fill)
    return mid.view(shape)

def _want(dtype, device):
    dev = torch.device(device) if device is not None else torch.device("cpu")
    dt = dtype or torch.get_default_dtype()
    return dev.type == "cuda" and dt in (torch.float32,)

def empty(*a, **k):
    if _want(k.get("dtype"), k.get("device")) and not k.get("pin_memory") and "out" not in k:
        return _guarded(_size(a, k), k.get("dtype") or torch.float32, k["device"], None)
    return _empty(*a, **k)

def zeros(*a, **k):
    if _want(k.get("dtype"), k.get("device")) and "out" not in k:
        return _guarded(_size(a, k), k.get("dtype") or torch.float32, k["device"], 0.0)
    return _zeros(*a, **k)

def empty_like(t, *a, **k):
    dt, dev = k.get("dtype", t.dtype), k.get("device", t.device)
    if _want(dt, dev) and t.is_contiguous():
        return _guarded(tuple(t.shape), dt, dev, None)
    return _empty_like(t, *a, **k)

def zeros_like(t, *a, **k):
    dt, dev = k.get("dtype", t.dtype), k.get("device", t.device)
    if _want(dt, dev) and t.is_contiguous():
        return _guarded(tuple(t.shape), dt, dev, 0.0)
    return _zeros_like(t, *a, **k)

torch.empty, torch.zeros, torch.empty_like, torch.zeros_like = empty, zeros, empty_like, zeros_like
import pytest
sys.exit(pytest.main(sys.argv[1:]))
