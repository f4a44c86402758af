#!/usr/bin/env python3
"""Golden vectors for the NEGATIVE-SAMPLING (`uni100`-style) branch of the reference's Collector
(recbole/evaluator/collector.py:131-205 with `full == False`) and its Evaluator, fed the way
Trainer._neg_sample_batch_eval feeds them (trainer.py:440-456): per batch a dense [batch_users, n_items] matrix that is
-inf except at the candidate (user, item) pairs.  Batches are laid out as NegSampleEvalDataLoader builds them
(general_dataloader.py:132-158): per user its positives followed by its sampled negatives.

The quirks of that branch are part of the vectors: `rec.negative_score` / `data.negative_i` are read from ROWS
[P, 2P) of the batch (P = positives in the batch), whatever lies there, looked up in the row of the j-th positive's
user (-inf when that item is not one of that user's candidates), and `data.<sst>` is the attribute of the first P rows.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402
from recbole.data.interaction import Interaction  # noqa: E402
from recbole.evaluator import Collector, Evaluator  # noqa: E402


class _Cfg(dict):
    def __getitem__(self, k):
        return self.get(k, None)


def run_case(name, seed, n_items, batches, n_neg, topk, metrics):
    rng = np.random.default_rng(seed)
    cfg = _Cfg(metrics=metrics, topk=list(topk), metric_decimal_place=10, sst_attr_list=["gender"],
               eval_args={"mode": f"uni{n_neg}"}, device=torch.device("cpu"), ITEM_ID_FIELD="item_id",
               USER_ID_FIELD="user_id")
    col, ev = Collector(cfg), Evaluator(cfg)
    out = {"n_items": np.array(n_items), "topk": np.array(topk), "metrics": np.array(metrics),
           "n_batches": np.array(len(batches)), "n_neg": np.array(n_neg)}
    for b, Ub in enumerate(batches):
        table = rng.random((Ub, n_items)).astype(np.float32)            # predict() of every (user, item) pair
        gender = rng.integers(0, 2, Ub).astype(np.float32)
        row_idx, items, sst, pos_u, pos_i = [], [], [], [], []
        for r in range(Ub):
            perm = rng.permutation(np.arange(1, n_items))
            npos = int(rng.integers(1, 4))
            pos = perm[:npos]
            neg = rng.choice(perm[npos:], size=npos * n_neg, replace=True)     # duplicates among negatives allowed
            block = np.concatenate([pos, neg])
            row_idx += [r] * len(block)
            items += list(block)
            sst += [gender[r]] * len(block)
            pos_u += [r] * npos
            pos_i += list(pos)
        row_idx, items = np.array(row_idx, dtype=np.int64), np.array(items, dtype=np.int64)
        pos_u, pos_i = np.array(pos_u, dtype=np.int64), np.array(pos_i, dtype=np.int64)
        origin = table[row_idx, items]
        dense = torch.full((Ub, n_items), -np.inf)
        dense[torch.from_numpy(row_idx), torch.from_numpy(items)] = torch.from_numpy(origin)
        inter = Interaction({"item_id": torch.from_numpy(items), "gender": torch.from_numpy(np.array(sst, dtype=np.float32))})
        col.eval_batch_collect(dense, inter, torch.from_numpy(pos_u), torch.from_numpy(pos_i))
        out[f"row_idx{b}"], out[f"items{b}"], out[f"origin{b}"] = row_idx, items, origin
        out[f"sst{b}"], out[f"pos_u{b}"], out[f"pos_i{b}"] = np.array(sst, dtype=np.float32), pos_u, pos_i
        out[f"n_users{b}"] = np.array(Ub)
    struct = col.get_data_struct()
    for key in ("rec.topk", "rec.positive_score", "data.positive_i", "rec.negative_score", "data.negative_i", "data.gender"):
        if key in struct:
            out["collected." + key] = struct.get(key).numpy()
    with np.errstate(all="ignore"):
        res = {k: float(v) for k, v in ev.evaluate(struct).items()}
    out["result_json"] = np.array(json.dumps(res))
    path = os.path.join(HERE, f"collector_uni_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {os.path.getsize(path) / 1024:.1f} KiB", res)


def main():
    ranking = ["Recall", "MRR", "NDCG", "Hit", "Precision"]
    fair = ["NonParityUnfairness", "ValueUnfairness", "AbsoluteUnfairness", "UnderUnfairness", "OverUnfairness",
            "DifferentialFairness"]
    run_case("one_user_batches", 1, 60, [1, 1, 1, 1, 1, 1], 10, (5, 10), ranking + fair)   # negatives slice is meaningful
    run_case("multi_user_batches", 2, 200, [8, 8, 5], 20, (10,), ranking + fair)             # -inf negatives, nan metrics
    run_case("ranking_only", 3, 500, [32, 32], 100, (1, 10, 20), ranking)


if __name__ == "__main__":
    main()
