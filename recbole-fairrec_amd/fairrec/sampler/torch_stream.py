"""torch's CPU generator as the source of DEVICE-side draws: `randperm(n, device)` returns the permutation
`torch.randperm(n)` would have returned -- same bits -- and leaves `torch.get_rng_state()` exactly where that call would
have left it, but the Fisher-Yates chain (30 ns per element on a host core: 270 ms for the 8.4 M interactions of a
1024-step epoch of BASELINE.json configs[1]) runs as three launches on the GPU (fr_randperm, csrc/randperm.hip).

Replaces interaction.py:293-297 for device-resident interaction tables (the training loader's per-epoch shuffle).
Layout of the state blob: ATen's CPUGeneratorImplState (aten/src/ATen/CPUGeneratorImpl.cpp) -- uint64 seed, int32 left,
int32 seeded, uint64 next, uint64 state[624], ... ; `left == 1` means "regenerate before the next draw"."""
from __future__ import annotations

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


_PINNED = {}


def _staging():
    """One pinned 625-word buffer per process: the state crosses to the device without a synchronising pageable copy."""
    if "buf" not in _PINNED:
        _PINNED["buf"] = torch.empty(625, dtype=torch.int32).pin_memory()
    return _PINNED["buf"]


def randperm(n: int, device) -> torch.Tensor:
    """`torch.randperm(n)` (default CPU generator), computed on `device`; int64 [n] there."""
    n = int(n)
    device = torch.device(device)
    if device.type != "cuda" or n >= MAX_N or n < 2:
        return torch.randperm(n).to(device)
    blob = torch.get_rng_state()
    key, pos = _read(blob)
    pinned = _staging()
    pinned.numpy()[:] = np.concatenate([key, np.array([pos], dtype=np.uint32)]).view(np.int32)
    state = pinned.to(device, non_blocking=True)
    out = torch.empty(n, dtype=torch.int64, device=device)
    ws = torch.empty(_C.lib().fr_randperm_workspace_bytes(n), dtype=torch.uint8, device=device)
    _C.check(_C.lib().fr_randperm(state.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel(), _C.current_stream()),
             "fr_randperm")
    w = state.cpu().numpy().view(np.uint32)          # (the shuffle's one host sync: 2.5 KB back; it also retires the pinned copy)
    torch.set_rng_state(_write(blob, w[:624], int(w[624])))
    return out
