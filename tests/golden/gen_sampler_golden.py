#!/usr/bin/env python3
"""Golden vectors for the negative sampler (SURVEY.md §8-c fixture 3, next-row f-1) by RUNNING THE REFERENCE's
`recbole.sampler.Sampler` (build container only).

Per case: (numpy seed, item_num, user_num, the training interactions that define the used-sets, a sequence of
`sample_by_user_ids(user_ids, item_ids, num)` calls on ONE continuing numpy stream) -> the sampled ids of every call
and the numpy generator state after the last one (so that a device stream can be checked to be handed back in sync).
Both branches of sample_by_key_ids are exercised (all keys equal / mixed keys), as are heavy users whose used-set
covers most of the catalogue (many rejection rounds) and the pointwise `num > 1` tiling.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402
from recbole.sampler import Sampler  # noqa: E402


class _DS:
    uid_field, iid_field = "user_id", "item_id"

    def __init__(self, user_num, item_num, u, i):
        self.user_num, self.item_num = user_num, item_num
        self.inter_feat = {"user_id": torch.from_numpy(u), "item_id": torch.from_numpy(i)}


def run_case(name, seed, user_num, item_num, n_inter, calls, heavy=(), distribution="uniform"):
    rng = np.random.default_rng(seed)
    u = rng.integers(1, user_num, size=n_inter).astype(np.int64)
    i = rng.integers(1, item_num, size=n_inter).astype(np.int64)
    for hu, frac in heavy:                      # users that have interacted with `frac` of the catalogue
        items = rng.choice(np.arange(1, item_num), size=int(frac * (item_num - 1)), replace=False)
        u = np.concatenate([u, np.full(len(items), hu, dtype=np.int64)])
        i = np.concatenate([i, items.astype(np.int64)])
    sampler = Sampler("train", _DS(user_num, item_num, u, i), distribution).set_phase("train")
    out = {"distribution": np.array(distribution), "seed": np.array(seed), "user_num": np.array(user_num), "item_num": np.array(item_num),
           "train_user": u, "train_item": i, "n_calls": np.array(len(calls))}
    np.random.seed(seed)
    for c, (kind, n, num) in enumerate(calls):
        if kind == "same":                      # all keys equal: the first branch of sample_by_key_ids
            users = np.full(n, heavy[0][0] if heavy else 1, dtype=np.int64)
        elif kind == "heavy":                   # mostly heavy users: long rejection chains
            pool = np.array([h for h, _ in heavy] + [1, 2], dtype=np.int64)
            users = pool[rng.integers(0, len(pool), size=n)]
        else:
            users = rng.integers(1, user_num, size=n).astype(np.int64)
        items = rng.integers(1, item_num, size=n).astype(np.int64)   # ignored by the sampler, part of the call
        neg = sampler.sample_by_user_ids(users, items, num)
        out[f"users{c}"], out[f"num{c}"], out[f"neg{c}"] = users, np.array(num), neg.numpy().astype(np.int64)
    st = np.random.get_state()
    out["final_key"], out["final_pos"] = st[1].astype(np.uint32), np.array(st[2])
    # a plain randint stream on the same seed (no used-sets), for the raw generator
    np.random.seed(seed)
    out["randint_stream"] = np.random.randint(1, item_num, 3000).astype(np.int64)
    path = os.path.join(HERE, f"sampler_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {len(calls)} calls, {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    run_case("ml100k_like", 2020, 944, 1683, 20000, [("mixed", 2048, 1), ("mixed", 2048, 1), ("mixed", 777, 1), ("same", 300, 1)],
             heavy=((7, 0.44),))
    run_case("heavy", 7, 50, 130, 400, [("heavy", 512, 1), ("same", 200, 1), ("heavy", 100, 3), ("mixed", 64, 2)],
             heavy=((3, 0.95), (4, 0.80), (5, 0.50)))
    run_case("pow2_range", 11, 300, 1026, 3000, [("mixed", 1500, 1), ("mixed", 1, 1), ("mixed", 625, 2)])   # rng = 1024-... mask edge
    run_case("tiny_catalogue", 5, 40, 5, 12, [("mixed", 100, 1)])        # item_num - 2 = 3: two-bit mask, no rejection by range
    # popularity-biased sampling (alias method, sampler.py:72-118): randint for the slot + random() for the coin
    run_case("popularity", 3, 200, 400, 5000, [("mixed", 1000, 1), ("same", 200, 1), ("mixed", 300, 2)], heavy=((7, 0.5),),
             distribution="popularity")
    run_case("large", 2020, 20000, 100001, 200000, [("mixed", 8192, 1), ("mixed", 8192, 1), ("mixed", 4096, 2)])


if __name__ == "__main__":
    main()
