#!/usr/bin/env python3
"""Golden vectors for the full-sort evaluation pipeline (SURVEY.md §8-c fixture 6, next-row f-2) by RUNNING THE
REFERENCE's Collector and Evaluator (recbole/evaluator/collector.py:131-205 `eval_batch_collect` in `full` mode,
evaluator.py:37-52) on synthetic score matrices, batch by batch, exactly as Trainer.evaluate feeds them
(trainer.py:420-438 masks column 0 and the history before collecting).

Per case: for every user batch the score matrix [users, n_items] BEFORE masking, the history (row, item) pairs, the
positives (positive_u, positive_i) and the users' sensitive attribute; the collected arrays after all batches and the
final metric dictionary.
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


def run_case(name, seed, n_items, batches, topk, metrics, popularity_ratio=None, tail_ratio=None):
    rng = np.random.default_rng(seed)
    cfg = _Cfg(metrics=metrics, topk=list(topk), metric_decimal_place=10, sst_attr_list=["gender"],
               eval_args={"mode": "full"}, device=torch.device("cpu"), ITEM_ID_FIELD="item_id", USER_ID_FIELD="user_id",
               popularity_ratio=popularity_ratio, tail_ratio=tail_ratio)
    col, ev = Collector(cfg), Evaluator(cfg)
    out = {"n_items": np.array(n_items), "topk": np.array(topk), "metrics": np.array(metrics), "n_batches": np.array(len(batches)),
           "popularity_ratio": np.array(-1.0 if popularity_ratio is None else popularity_ratio),
           "tail_ratio": np.array(-1.0 if tail_ratio is None else tail_ratio)}
    # what Collector.data_collect(train_data) would provide (collector.py:80-97): the catalogue size and the training
    # popularity of the items that occur in training (Counter: items absent from training are absent from it)
    train_items = np.floor((n_items - 1) * rng.random(20 * n_items) ** 2).astype(np.int64) + 1
    out["train_items"] = train_items
    from collections import Counter
    col.data_struct.set("data.num_items", n_items)
    col.data_struct.set("data.count_items", Counter(train_items.tolist()))
    uid0 = 1
    for b, Ub in enumerate(batches):
        scores = rng.random((Ub, n_items)).astype(np.float32)              # predict() outputs are in [0, 1]
        gender = rng.integers(0, 2, Ub).astype(np.float32)
        hist_u, hist_i, pos_u, pos_i = [], [], [], []
        for r in range(Ub):
            items = rng.permutation(np.arange(1, n_items))
            nh, npos = rng.integers(0, n_items // 3), rng.integers(1, 6)
            hist_u += [r] * nh
            hist_i += list(items[:nh])
            pos_u += [r] * npos
            pos_i += list(items[nh:nh + npos])
        hist_u, hist_i = np.array(hist_u, dtype=np.int64), np.array(hist_i, dtype=np.int64)
        pos_u, pos_i = np.array(pos_u, dtype=np.int64), np.array(pos_i, dtype=np.int64)
        out[f"scores{b}"], out[f"gender{b}"] = scores, gender
        out[f"hist_u{b}"], out[f"hist_i{b}"], out[f"pos_u{b}"], out[f"pos_i{b}"] = hist_u, hist_i, pos_u, pos_i
        out[f"users{b}"] = np.arange(uid0, uid0 + Ub, dtype=np.int64)
        uid0 += Ub
        s = torch.from_numpy(scores.copy())
        s[:, 0] = -np.inf                                         # trainer.py:435-437
        s[torch.from_numpy(hist_u), torch.from_numpy(hist_i)] = -np.inf
        inter = Interaction({"user_id": torch.from_numpy(out[f"users{b}"]), "gender": torch.from_numpy(gender)})
        col.eval_batch_collect(s, inter, torch.from_numpy(pos_u), torch.from_numpy(pos_i))
    struct = col.get_data_struct()
    for key in ("rec.topk", "rec.items", "rec.positive_score", "data.positive_i", "data.gender"):
        if key in struct:
            out["collected." + key] = struct.get(key).numpy()
    out["result_json"] = np.array(json.dumps({k: float(v) for k, v in ev.evaluate(struct).items()}))
    path = os.path.join(HERE, f"collector_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {os.path.getsize(path) / 1024:.1f} KiB", json.loads(str(out["result_json"])))


def main():
    ranking = ["Recall", "MRR", "NDCG", "Hit", "Precision"]
    fair = ["NonParityUnfairness", "ValueUnfairness", "AbsoluteUnfairness", "UnderUnfairness", "OverUnfairness",
            "DifferentialFairness"]
    run_case("full_small", 1, 40, [7, 7, 3], (5, 10), ranking + fair)
    run_case("full_medium", 2, 600, [64, 64, 64, 17], (1, 10, 20), ranking + fair)
    run_case("full_ranking_only", 3, 90, [30, 11], (10,), ranking)
    exposure = ["GiniIndex", "PopularityPercentage", "ItemCoverage", "AveragePopularity", "ShannonEntropy", "TailPercentage"]
    run_case("full_popularity", 4, 300, [50, 50, 21], (5, 20), ranking + exposure + fair)
    run_case("full_popularity_threshold", 5, 120, [40, 9], (10,), exposure, popularity_ratio=30, tail_ratio=12)


if __name__ == "__main__":
    main()
