"""`Interaction`: the batch type handed to `calculate_loss` / `predict` — a dict of equal-length tensors.

Clean-room container with the contract of recbole/data/interaction.py:43-368: field access
by name, row indexing/slicing, `.to(device)`, `repeat`, `repeat_interleave`, `update`, `drop`, `shuffle`,
`sort`, plus `cat_interactions`.  Host-side plumbing only; the kernels see raw device pointers.
"""
from __future__ import annotations

from typing import Dict, Iterable, List

import numpy as np
import torch


def _as_tensor(v) -> torch.Tensor:
    if isinstance(v, torch.Tensor):
        return v
    if isinstance(v, np.ndarray):
        return torch.from_numpy(v)
    if isinstance(v, (list, tuple)):
        return torch.as_tensor(np.asarray(v))
    raise ValueError(f"cannot turn {type(v)} into a tensor column")


class Interaction:
    def __init__(self, interaction):
        if hasattr(interaction, "to_dict") and hasattr(interaction, "columns"):   # pandas.DataFrame
            interaction = {c: interaction[c].values for c in interaction.columns}
        if not isinstance(interaction, dict):
            raise ValueError("Interaction takes a dict (or DataFrame) of columns")
        self.interaction: Dict[str, torch.Tensor] = {k: _as_tensor(v) for k, v in interaction.items()}
        self.length = -1
        for v in self.interaction.values():
            self.length = max(self.length, v.shape[0])

    # --- access -----------------------------------------------------------------------------------------
    def __iter__(self):
        return iter(self.interaction)

    def __getattr__(self, item):
        if "interaction" not in self.__dict__:
            raise AttributeError("'Interaction' object has no attribute 'interaction'")
        if item in self.interaction:
            return self.interaction[item]
        raise AttributeError(f"'Interaction' object has no attribute '{item}'")

    def __getitem__(self, index):
        if isinstance(index, str):
            return self.interaction[index]
        if isinstance(index, (list, np.ndarray)):
            index = torch.as_tensor(np.asarray(index))
        if torch.is_tensor(index) and index.dtype != torch.bool:
            on = {}                 # an index list crosses to a device once, not once per column
            out = {}
            for k, v in self.interaction.items():
                if v.device not in on:
                    on[v.device] = index.to(v.device)
                out[k] = v[on[v.device]]
            return Interaction(out)
        return Interaction({k: v[index] for k, v in self.interaction.items()})

    def __setitem__(self, key, value):
        if not isinstance(key, str):
            raise KeyError(f"{type(key)} object does not support item assignment")
        self.interaction[key] = value

    def __delitem__(self, key):
        del self.interaction[key]

    def __contains__(self, item):
        return item in self.interaction

    def __len__(self):
        return self.length

    def __str__(self):
        rows = [f"The batch_size of interaction: {self.length}"]
        rows += [f"    {k}, {tuple(v.shape)}, {v.device.type}, {v.dtype}" for k, v in self.interaction.items()]
        return "\n".join(rows) + "\n"

    __repr__ = __str__

    @property
    def columns(self) -> List[str]:
        return list(self.interaction.keys())

    # --- movement ---------------------------------------------------------------------------------------
    def to(self, device, selected_field=None):
        if isinstance(selected_field, str):
            selected_field = [selected_field]
        sel = set(selected_field) if selected_field is not None else None
        return Interaction({k: (v.to(device) if sel is None or k in sel else v) for k, v in self.interaction.items()})

    def cpu(self):
        return Interaction({k: v.cpu() for k, v in self.interaction.items()})

    def numpy(self):
        return {k: v.numpy() for k, v in self.interaction.items()}

    # --- reshaping --------------------------------------------------------------------------------------
    def repeat(self, sizes):
        out = {}
        for k, v in self.interaction.items():
            out[k] = v.repeat(sizes) if v.dim() == 1 else v.repeat([sizes, 1])
        return Interaction(out)

    def repeat_interleave(self, repeats, dim=0):
        return Interaction({k: v.repeat_interleave(repeats, dim=dim) for k, v in self.interaction.items()})

    def update(self, new_inter: "Interaction"):
        for k in new_inter.interaction:
            self.interaction[k] = new_inter.interaction[k]

    def drop(self, column):
        if column not in self.interaction:
            raise ValueError(f"Column [{column}] is not in [{self}].")
        del self.interaction[column]

    def _reindex(self, index):
        on = {}                     # the index crosses to a device once, not once per column
        for k in self.interaction:
            v = self.interaction[k]
            if v.device not in on:
                on[v.device] = index.to(v.device)
            self.interaction[k] = v[on[v.device]]

    def shuffle(self):
        """interaction.py:293-297: reindex by torch.randperm(length) -- the same permutation and the same generator state
        afterwards, but for columns that live on a GPU the permutation is computed THERE (fairrec/sampler/torch_stream.py:
        the host's Fisher-Yates chain costs 30 ns per interaction, several times a training epoch at device step rates)."""
        dev = next((v.device for v in self.interaction.values() if v.is_cuda), None)
        if dev is None:
            self._reindex(torch.randperm(self.length))
        else:
            from ..sampler.torch_stream import randperm
            self._reindex(randperm(self.length, dev))

    def sort(self, by, ascending=True):
        if isinstance(by, str):
            by = [by]
        if isinstance(ascending, bool):
            ascending = [ascending] * len(by)
        if len(by) != len(ascending):
            raise ValueError(f"by [{by}] and ascending [{ascending}] should have same length.")
        for b, a in list(zip(by, ascending))[::-1]:
            if b not in self.interaction:
                raise ValueError(f"[{b}] is not exist in interaction [{self}].")
            key = self.interaction[b]
            if key.dim() != 1:
                raise ValueError("sort key must be one-dimensional")
            idx = np.argsort(key.cpu().numpy(), kind="stable")
            if not a:
                idx = idx[::-1].copy()
            self._reindex(torch.from_numpy(idx))

    def add_prefix(self, prefix):
        self.interaction = {prefix + k: v for k, v in self.interaction.items()}


def cat_interactions(interactions: Iterable[Interaction]) -> Interaction:
    interactions = list(interactions)
    if not interactions:
        raise ValueError("interactions is empty")
    cols = set(interactions[0].columns)
    for it in interactions:
        if set(it.columns) != cols:
            raise ValueError("interactions do not share the same columns")
    return Interaction({c: torch.cat([it[c] for it in interactions]) for c in interactions[0].columns})
