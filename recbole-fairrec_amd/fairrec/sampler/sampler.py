"""Negative sampling on the device with the interface of recbole/sampler/sampler.py (`Sampler(phases, datasets,
distribution)`, `set_phase`, `sample_by_user_ids`) and the SAME numbers: the sampled ids are bit-identical to what
the reference's numpy-based sampler draws from the same generator state (fr_sample_negatives).

The numpy global generator the reference draws from (`np.random.randint`, sampler.py:240-241) is mirrored by a
`DeviceRandomState` that keeps numpy's legacy MT19937 state in device memory.  `get_state()` / `set_state()` use
numpy's own tuple format, so host code that must draw from the same stream between batches (FOCFDataLoader's
`np.random.choice`, the trainers' per-epoch attribute masks, trainer.py:879-882) can take the stream over and hand it
back:   np.random.set_state(rs.get_state()); ...host draws...; rs.set_state(np.random.get_state())
"""
from __future__ import annotations

import copy
from typing import Dict, Optional

import numpy as np
import torch

from .. import _C

STATE_WORDS = 625     # key[624] + pos


class DeviceRandomState:
    """numpy's legacy RandomState (MT19937), resident on the GPU."""

    def __init__(self, device, seed: Optional[int] = None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _C.FairrecError("DeviceRandomState needs a GPU (no CPU fallback by design)")
        _C.lib()
        self.state = torch.zeros(STATE_WORDS, dtype=torch.int32, device=self.device)
        self._ws: Optional[torch.Tensor] = None
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.seed(0 if seed is None else seed)

    def seed(self, seed: int):
        """np.random.seed(seed) for an int seed."""
        seed = int(seed)
        if not 0 <= seed <= 0xFFFFFFFF:
            raise ValueError("Seed must be between 0 and 2**32 - 1")
        _C.check(_C.lib().fr_mt19937_seed(self.state.data_ptr(), seed, _C.current_stream()), "fr_mt19937_seed")

    def get_state(self):
        """np.random.get_state() tuple of the device stream (host sync)."""
        w = self.state.cpu().numpy().view(np.uint32)
        return ("MT19937", w[:624].copy(), int(w[624]), 0, 0.0)

    def set_state(self, state):
        key = np.asarray(state[1], dtype=np.uint32)
        w = np.concatenate([key, np.array([int(state[2])], dtype=np.uint32)]).view(np.int32)
        self.state.copy_(torch.from_numpy(w.copy()))

    def _workspace(self, total: int) -> torch.Tensor:
        need = _C.lib().fr_sample_negatives_workspace_bytes(total)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def randint(self, low: int, high: int, n: int) -> torch.Tensor:
        """np.random.randint(low, high, n) as an int64 device tensor (range below 2**32 - 1)."""
        out = torch.empty(n, dtype=torch.int64, device=self.device)
        if n:
            _C.check(_C.lib().fr_sample_negatives(self.state.data_ptr(), low, high, None, n, 1, None, None, 0,
                                                  out.data_ptr(), None, None, 0, self.err_flag.data_ptr(),
                                                  _C.current_stream()), "fr_sample_negatives")
        return out

    def sample_excluding(self, low: int, high: int, key_ids: torch.Tensor, num: int, used_indptr: torch.Tensor,
                         used_items: torch.Tensor, rounds_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        key_ids = key_ids.to(self.device, torch.int64).contiguous()
        n = key_ids.numel()
        out = torch.empty(n * num, dtype=torch.int64, device=self.device)
        if n:
            ws = self._workspace(n * num)
            _C.check(_C.lib().fr_sample_negatives(self.state.data_ptr(), low, high, key_ids.data_ptr(), n, num,
                                                  used_indptr.data_ptr(), used_items.data_ptr(), used_indptr.numel() - 1,
                                                  out.data_ptr(), _C.ptr(rounds_out), ws.data_ptr(), ws.numel(),
                                                  self.err_flag.data_ptr(), _C.current_stream()), "fr_sample_negatives")
        return out


    def sample_calls(self, low: int, high: int, call_keys: torch.Tensor, call_counts: torch.Tensor,
                     used_indptr: torch.Tensor, used_items: torch.Tensor) -> torch.Tensor:
        """Consecutive single-key calls on this stream in one launch: call c draws call_counts[c] values for key
        call_keys[c] (each call finishes its re-draw rounds before the next one draws).  Returns the concatenation."""
        call_keys = call_keys.to(self.device, torch.int64).contiguous()
        counts = call_counts.to(self.device, torch.int64)
        offsets = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(counts, 0, out=offsets[1:])
        total, max_call = int(offsets[-1].item()), int(counts.max().item()) if counts.numel() else 0
        out = torch.empty(total, dtype=torch.int64, device=self.device)
        if total:
            # (room for the speculative form of a long call sequence: csrc/sampler.hip, sample_calls_fast_kernel)
            need = max(_C.lib().fr_sample_negatives_calls_workspace_bytes(total, max_call),
                       _C.lib().fr_sample_negatives_workspace_bytes(max_call))
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            ws = self._ws
            _C.check(_C.lib().fr_sample_negatives_calls(self.state.data_ptr(), low, high, call_keys.data_ptr(),
                                                        offsets.data_ptr(), call_keys.numel(), max_call,
                                                        used_indptr.data_ptr(), used_items.data_ptr(),
                                                        used_indptr.numel() - 1, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                        self.err_flag.data_ptr(), _C.current_stream()),
                     "fr_sample_negatives_calls")
        return out


_GLOBAL: Dict[str, DeviceRandomState] = {}


def global_random_state(device) -> DeviceRandomState:
    """The device mirror of numpy's global generator (one per device); `fairrec.utils.init_seed` seeds it together
    with numpy, as the reference seeds numpy (utils.py:172-189)."""
    key = str(torch.device(device))
    if key not in _GLOBAL:
        _GLOBAL[key] = DeviceRandomState(device)
        _GLOBAL[key].set_state(np.random.get_state())
    return _GLOBAL[key]


class host_numpy_stream:
    """`with host_numpy_stream():` -- host code inside draws from np.random exactly where the reference would: the
    device mirror(s) of numpy's global generator hand the stream to numpy on entry and take it back on exit (a 2.5 KB
    copy each way; meant for the rare host draws between batches -- per-epoch attribute masks, FOCF's item picks).
    Without a device mirror in use it does nothing."""

    def __enter__(self):
        self._rs = next(iter(_GLOBAL.values()), None)
        if self._rs is not None:
            np.random.set_state(self._rs.get_state())
        return self

    def __exit__(self, *exc):
        if self._rs is not None:
            st = np.random.get_state()
            for rs in _GLOBAL.values():
                rs.set_state(st)
        return False


def seed_all(seed: int):
    for rs in _GLOBAL.values():
        rs.seed(seed)


class Sampler:
    """recbole.sampler.Sampler (sampler.py:200-303): negative items per user, never one of the user's positive items
    of the current or an earlier phase.  Distribution 'uniform' (the reference default, overall.yaml) on the device,
    'popularity' with numpy on the shared stream."""

    def __init__(self, phases, datasets, distribution='uniform', device=None, random_state: Optional[DeviceRandomState] = None):
        if not isinstance(phases, list):
            phases = [phases]
        if not isinstance(datasets, list):
            datasets = [datasets]
        if len(phases) != len(datasets):
            raise ValueError(f'Phases {phases} and datasets {datasets} should have the same length.')
        if distribution not in ('uniform', 'popularity'):
            raise NotImplementedError(f'The sampling distribution [{distribution}] is not implemented.')
        self.phases, self.datasets, self.distribution = phases, datasets, distribution
        self.uid_field, self.iid_field = datasets[0].uid_field, datasets[0].iid_field
        self.user_num, self.item_num = datasets[0].user_num, datasets[0].item_num
        self.device = torch.device(device if device is not None else "cuda")
        self.rs = random_state if random_state is not None else global_random_state(self.device)
        self.used_ids = self.get_used_ids()
        self.phase = None
        if distribution == 'popularity':
            self._build_alias_table()

    def get_used_ids(self):
        """Per phase a CSR (indptr int64 [user_num+1], items int32 sorted per user) of the items used up to and
        including that phase (sampler.py:243-265: each phase's sets start from the previous phase's)."""
        out, u_all, i_all = {}, [], []
        for phase, ds in zip(self.phases, self.datasets):
            u_all.append(ds.inter_feat[self.uid_field].cpu().numpy().astype(np.int64))
            i_all.append(ds.inter_feat[self.iid_field].cpu().numpy().astype(np.int64))
            pairs = np.unique(np.concatenate(u_all) * self.item_num + np.concatenate(i_all))   # sorted by (user, item)
            users, items = pairs // self.item_num, pairs % self.item_num
            counts = np.bincount(users, minlength=self.user_num)
            indptr = np.zeros(self.user_num + 1, dtype=np.int64)
            np.cumsum(counts, out=indptr[1:])
            out[phase] = (torch.from_numpy(indptr).to(self.device), torch.from_numpy(items.astype(np.int32)).to(self.device),
                          int(counts.max()) if len(counts) else 0)
        if out and out[self.phases[-1]][2] + 1 >= self.item_num:      # [pad] is an item (sampler.py:258-264)
            raise ValueError('Some users have interacted with all items, which we can not sample negative items for '
                             'them. Please set `user_inter_num_interval` to filter those users.')
        return out

    # --- popularity-biased sampling (sampler.py:72-118): alias method over the items of all phases' interactions.  Not a
    # device kernel: the draws are numpy's (`randint` for the slot, `random` for the coin), made inside
    # `host_numpy_stream()` so that they sit at the reference's position of the ONE generator stream the device sampler
    # shares with numpy -- same ids as the reference, at host speed (the uniform default runs on the device).
    def _build_alias_table(self):
        from collections import Counter
        cand = []
        for ds in self.datasets:                                   # _get_candidates_list, sampler.py:229-233
            cand.extend(ds.inter_feat[self.iid_field].cpu().numpy())
        prob = dict(Counter(cand))                                 # insertion order = first occurrence, as in the reference
        alias = prob.copy()
        large_q, small_q = [], []
        for i in prob:
            alias[i] = -1
            prob[i] = prob[i] / len(cand) * len(prob)
            if prob[i] > 1:
                large_q.append(i)
            elif prob[i] < 1:
                small_q.append(i)
        while len(large_q) != 0 and len(small_q) != 0:
            l, s = large_q.pop(0), small_q.pop(0)
            alias[s] = l
            prob[l] = prob[l] - (1 - prob[s])
            if prob[l] < 1:
                small_q.append(l)
            elif prob[l] > 1:
                large_q.append(l)
        keys = np.array(list(prob.keys()), dtype=np.int64)
        self._pop = (keys, np.array([prob[k] for k in keys], dtype=np.float64),
                     np.array([alias[k] for k in keys], dtype=np.int64))

    def _pop_sampling(self, n):
        keys, prob, alias = self._pop
        idx = np.random.randint(0, len(keys), n)
        coin = np.random.random(n)
        return np.where(prob[idx] > coin, keys[idx], alias[idx])

    def _pop_sample_by_user_ids(self, user_ids, num):
        indptr, items, _ = self.used_ids
        if not hasattr(self, "_used_host") or self._used_host[0] is not indptr:
            self._used_host = (indptr, indptr.cpu().numpy(), items.cpu().numpy().astype(np.int64))
        _, ip, it = self._used_host
        keys = np.tile(np.asarray(user_ids.cpu() if torch.is_tensor(user_ids) else user_ids, dtype=np.int64), num)
        used_key = keys * self.item_num                            # membership in the user's used-set = a sorted (user, item) key
        allk = np.repeat(np.arange(self.user_num, dtype=np.int64), np.diff(ip)) * self.item_num + it
        value = np.zeros(len(keys), dtype=np.int64)
        check = np.arange(len(keys))
        np.random.set_state(self.rs.get_state())                   # numpy continues the stream where the device mirror stands ...
        while len(check) > 0:                                      # sample_by_key_ids, sampler.py:178-195
            value[check] = self._pop_sampling(len(check))
            k = used_key[check] + value[check]
            pos = np.searchsorted(allk, k)
            hit = (pos < len(allk)) & (allk[np.minimum(pos, len(allk) - 1)] == k)
            check = check[hit]
        st = np.random.get_state()                                 # ... and hands it back
        for rs in ([self.rs] if self.rs not in _GLOBAL.values() else list(_GLOBAL.values())):
            rs.set_state(st)
        return torch.from_numpy(value).to(self.device)

    def set_phase(self, phase):
        if phase not in self.phases:
            raise ValueError(f'Phase [{phase}] not exist.')
        new = copy.copy(self)
        new.phase = phase
        new.used_ids = self.used_ids[phase]
        return new

    def sample_by_user_ids(self, user_ids, item_ids, num):
        """[len(user_ids) * num] negative item ids on the device: entry j belongs to user_ids[j % len(user_ids)]
        (sampler.py:283-303); `item_ids` is unused, as in the reference."""
        if self.phase is None:
            raise ValueError('call set_phase() first')
        if not torch.is_tensor(user_ids):
            user_ids = torch.as_tensor(np.asarray(user_ids), dtype=torch.int64)
        if self.distribution == 'popularity':
            return self._pop_sample_by_user_ids(user_ids, int(num))
        indptr, items, _ = self.used_ids
        return self.rs.sample_excluding(1, self.item_num, user_ids, int(num), indptr, items)
