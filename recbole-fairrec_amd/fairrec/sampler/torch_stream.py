"""torch's CPU generator as the source of DEVICE-side draws: `randperm(n, device)` returns the permutation
`torch.randperm(n)` would have returned -- same bits -- and leaves `torch.get_rng_state()` exactly where that call would
have left it, but the Fisher-Yates chain (30 ns per element on a host core: 270 ms for the 8.4 M interactions of a
1024-step epoch of BASELINE.json configs[1]) runs as three launches on the GPU (fr_randperm, csrc/randperm.hip).

Replaces interaction.py:293-297 for device-resident interaction tables (the training loader's per-epoch shuffle).
Layout of the state blob: ATen's CPUGeneratorImplState (aten/src/ATen/CPUGeneratorImpl.cpp) -- uint64 seed, int32 left,
int32 seeded, uint64 next, uint64 state[624], ... ; `left == 1` means "regenerate before the next draw"."""
from __future__ import annotations

import os
import struct

import numpy as np
import torch

from .. import _C

_OFF_LEFT, _OFF_NEXT, _OFF_STATE, _BLOB = 8, 16, 24, 5056
MAX_N = 0xFFFFFFFF // 20        # ATen switches to another algorithm from here on


def _read(blob: torch.Tensor):
    b = blob.numpy().tobytes()
    if len(b) != _BLOB:
        raise _C.FairrecError(f"torch CPU generator state of {len(b)} bytes: layout not known to fairrec")
    left, _seeded = struct.unpack_from("<ii", b, _OFF_LEFT)
    nxt, = struct.unpack_from("<Q", b, _OFF_NEXT)
    key = np.frombuffer(b, dtype="<u8", count=624, offset=_OFF_STATE)
    if int(key.max()) > 0xFFFFFFFF or not 1 <= left <= 624 or (left != 1 and nxt != 625 - left):
        raise _C.FairrecError("torch CPU generator state does not look like at::mt19937's")
    return key.astype(np.uint32), (624 if left == 1 else int(nxt))


def _write(blob: torch.Tensor, key: np.ndarray, pos: int) -> torch.Tensor:
    b = bytearray(blob.numpy().tobytes())
    struct.pack_into("<i", b, _OFF_LEFT, 625 - pos)
    struct.pack_into("<Q", b, _OFF_NEXT, pos)
    b[_OFF_STATE:_OFF_STATE + 624 * 8] = key.astype("<u8").tobytes()
    return torch.frombuffer(b, dtype=torch.uint8).clone()


def _launch(blob: torch.Tensor, n: int, device, stream):
    """fr_randperm for the generator state `blob` on `stream`: (permutation, device state after the draws, pinned host copy of
    that state being filled, event recorded behind everything, what must stay alive until then)."""
    key, pos = _read(blob)
    pinned_in = torch.empty(625, dtype=torch.int32).pin_memory()
    pinned_in.numpy()[:] = np.concatenate([key, np.array([pos], dtype=np.uint32)]).view(np.int32)
    pinned_out = torch.empty(625, dtype=torch.int32).pin_memory()
    with torch.cuda.stream(stream):
        state = pinned_in.to(device, non_blocking=True)
        out = torch.empty(n, dtype=torch.int64, device=device)
        ws = torch.empty(_C.lib().fr_randperm_workspace_bytes(n), dtype=torch.uint8, device=device)
        _C.check(_C.lib().fr_randperm(state.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel(), stream.cuda_stream),
                 "fr_randperm")
        pinned_out.copy_(state, non_blocking=True)
        done = torch.cuda.Event()
        done.record(stream)
    return out, pinned_out, done, (pinned_in, state, ws)


# The NEXT call's permutation, computed ahead: (n, device) -> (generator state it starts from, permutation, pinned state after,
# event, keep-alive).  Used only if the generator is found in exactly that state when the call comes -- see randperm().
_AHEAD = {}
_SIDE = {}
LOOKAHEAD = os.environ.get("FAIRREC_RANDPERM_AHEAD", "1") != "0"
_WARNED = False


def randperm(n: int, device) -> torch.Tensor:
    """`torch.randperm(n)` (default CPU generator), computed on `device`; int64 [n] there.

    The one sequential piece -- the generator itself, ~0.8 ns per element on one workgroup -- is taken off the caller's
    critical path by SPECULATION: after a call returns, the permutation the NEXT call of the same size would return (if
    nobody draws from torch's CPU generator in between) is computed on a side stream, beside whatever the caller does next
    (a training epoch).  The next call compares the generator's state with the one the speculation started from: equal --
    the epoch loop of a trainer, where the shuffle is the generator's only consumer -- and the finished permutation is
    handed out and the generator set to the state the draws leave; different -- somebody drew in between -- and the
    speculation is dropped and the permutation computed now.  Either way the caller sees torch.randperm's bits and
    torch.randperm's generator state.  FAIRREC_RANDPERM_AHEAD=0 turns the speculation off."""
    n = int(n)
    device = torch.device(device)
    if device.type != "cuda" or n >= MAX_N or n < 2:
        return torch.randperm(n).to(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    blob = torch.get_rng_state()
    cur = torch.cuda.current_stream(device)
    hit = _AHEAD.pop((n, device), None)
    try:
        if hit is not None and torch.equal(hit[0], blob):
            _, out, pinned_out, done, keep = hit
            done.synchronize()                               # (long past, when an epoch of training lies in between)
            cur.wait_event(done)
            out.record_stream(cur)
        else:
            out, pinned_out, done, keep = _launch(blob, n, device, cur)
            done.synchronize()                               # the shuffle's one host sync: 2.5 KB of generator state back
    except _C.FairrecError as e:
        # A generator whose state blob is not at::mt19937's 5056 bytes (another torch build), or a refused launch: the same
        # permutation is a host call away -- slower (30 ns per element), never wrong.  Said once.
        global _WARNED
        if not _WARNED:
            import warnings
            warnings.warn(f"fairrec: device randperm not available ({e}); epoch shuffles fall back to torch.randperm on the host")
            _WARNED = True
        _AHEAD.clear()
        torch.set_rng_state(blob)
        return torch.randperm(n).to(device)
    w = pinned_out.numpy().view(np.uint32)
    after = _write(blob, w[:624].copy(), int(w[624]))
    torch.set_rng_state(after)
    if LOOKAHEAD and not torch.cuda.is_current_stream_capturing():
        side = _SIDE.get(device)
        if side is None:
            side = _SIDE[device] = torch.cuda.Stream(device=device, priority=0)
        while len(_AHEAD) >= 2:                          # (at most two sizes kept ahead: 24 bytes per element each)
            _AHEAD.pop(next(iter(_AHEAD)))
        _AHEAD[(n, device)] = (after,) + _launch(after, n, device, side)
    return out
