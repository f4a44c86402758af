"""Training batch feeds with the iteration contract of recbole/data/dataloader/*.py (len(), iteration yields
`Interaction`s, `.dataset`): the plain shuffled loader (general_dataloader.py:25-65 without negative sampling)
and the item-complete loader FOCF trains with (focf_dataloader.py:5-51).

FOCFDataLoader keeps the reference's draw order (`np.random.choice(candidates, 1, False)` per picked item)
but finds an item's interactions through a CSR built once instead of an O(#inter) `np.where` per pick
(SURVEY.md §8-f row f-3).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .interaction import Interaction


class AbstractDataLoader:
    def __init__(self, config, dataset, shuffle=False):
        self.config, self.dataset, self.shuffle = config, dataset, shuffle
        self.step = int(config['train_batch_size'])
        self.pr = 0

    @property
    def pr_end(self):
        return len(self.dataset)

    def __len__(self):
        return math.ceil(self.pr_end / self.step)

    def __iter__(self):
        if self.shuffle:
            self.dataset.shuffle()
        return self

    def __next__(self):
        if self.pr >= self.pr_end:
            self.pr = 0
            raise StopIteration()
        return self._next_batch_data()


class TrainDataLoader(AbstractDataLoader):
    """Fixed-size batches in (shuffled) dataset order, user features joined in."""

    def _next_batch_data(self):
        cur = self.dataset[self.pr:self.pr + self.step]
        self.pr += self.step
        return self.dataset.join(cur)


class FOCFDataLoader(AbstractDataLoader):
    """Item-complete batches: keep picking a random not-yet-picked item and append ALL its interactions until
    the batch holds >= train_batch_size rows (focf_dataloader.py:37-51)."""

    def __init__(self, config, dataset, shuffle=False):
        super().__init__(config, dataset, shuffle=False)
        self.ITEM_ID = config['ITEM_ID_FIELD']
        self.dataset.sort(by=self.ITEM_ID)                      # focf_dataloader.py:11
        items = self.dataset.inter_feat[self.ITEM_ID].numpy()
        self.item_num = self.dataset.item_num
        self.item_uniques = np.unique(items)
        # CSR by item over the item-sorted interaction array
        self.indptr = np.searchsorted(items, np.arange(self.item_num + 1), side="left")

    def _next_batch_data(self):
        cnt = 0
        select_item = np.arange(0, self.item_num)
        is_select = np.zeros(self.item_num, dtype=bool)
        is_select[self.item_uniques] = True
        chunks = []
        while cnt < self.step and is_select.any():
            iid = np.random.choice(select_item[is_select], 1, False)[0]   # same RNG consumption as the reference
            lo, hi = self.indptr[iid], self.indptr[iid + 1]
            cnt += hi - lo
            is_select[iid] = False
            chunks.append(np.arange(lo, hi))
        self.pr += self.step
        return self.dataset.join(self.dataset[np.concatenate(chunks)])
