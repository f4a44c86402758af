#!/usr/bin/env python3
"""Golden vectors for the FOCF item-complete batcher: RUNS the reference's FOCFDataLoader._next_batch_data
(focf_dataloader.py:37-51) on a small item-sorted interaction table and records the index lists of 4 batches
(fixture kind 4 of SURVEY.md §8-c).  Build container only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402
from recbole.data.dataloader.focf_dataloader import FOCFDataLoader  # noqa: E402
from recbole.data.interaction import Interaction  # noqa: E402


class _DS:
    def __init__(self, inter):
        self.inter_feat = inter

    def __getitem__(self, idx):
        return self.inter_feat[idx]

    def __len__(self):
        return len(self.inter_feat)


def main():
    rng = np.random.default_rng(7)
    n_items, n_users, n = 60, 80, 1500
    item = np.sort((n_items * rng.random(n) ** 1.7).astype(np.int64) + 1)     # skewed degrees, item-sorted
    user = rng.integers(1, n_users, n)
    rating = rng.integers(1, 6, n).astype(np.float32)
    inter = Interaction({"user_id": torch.from_numpy(user), "item_id": torch.from_numpy(item),
                         "rating": torch.from_numpy(rating)})
    dl = object.__new__(FOCFDataLoader)                    # bypass the Config/Sampler plumbing of __init__
    dl.dataset, dl.ITEM_ID, dl.step, dl.pr = _DS(inter), "item_id", 200, 0
    dl.item_num = n_items + 1
    dl.item_uniques = np.unique(item)
    np.random.seed(2020)
    out = {"user_id": user, "item_id": item, "rating": rating, "step": np.array(200), "item_num": np.array(n_items + 1),
           "np_seed": np.array(2020)}
    for b in range(4):
        batch = dl._next_batch_data()
        out[f"batch{b}_user"] = batch["user_id"].numpy()
        out[f"batch{b}_item"] = batch["item_id"].numpy()
        out[f"batch{b}_rating"] = batch["rating"].numpy()
    path = os.path.join(HERE, "dataloader_focf.npz")
    np.savez_compressed(path, **out)
    print(path, [len(out[f"batch{b}_user"]) for b in range(4)])


if __name__ == "__main__":
    main()
