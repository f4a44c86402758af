#!/usr/bin/env python3
"""Golden vectors for the evaluation metrics (SURVEY.md §8-c fixture 6, next-row f-2) by RUNNING THE REFERENCE's metric
classes (recbole/evaluator/metrics.py) on synthetic collected data (build container only).

Per case the arrays a Collector would have gathered (`rec.topk`, `rec.positive_score`, `data.positive_i`,
`rec.negative_score`, `data.negative_i`, `data.<sst>`) and the metric dictionaries the reference computes from them:
Hit / MRR / NDCG / Recall / Precision @ topk and the six fairness metrics (NonParity, Value, Absolute, Under, Over
Unfairness, DifferentialFairness) in `uni100`-style (with negatives) and `full` mode.
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
from recbole.evaluator import metrics as M  # noqa: E402
from recbole.evaluator.collector import DataStruct  # noqa: E402


class _Cfg(dict):
    def __getitem__(self, k):
        return self.get(k, None)


TOPK_METRICS = ["Hit", "MRR", "NDCG", "Recall", "Precision", "MAP"]
FAIR_METRICS = ["NonParityUnfairness", "ValueUnfairness", "AbsoluteUnfairness", "UnderUnfairness", "OverUnfairness",
                "DifferentialFairness"]


def run_case(name, seed, n_users, n_items, n_pos, topk, mode, three_groups=False):
    rng = np.random.default_rng(seed)
    cfg = _Cfg(metric_decimal_place=10, topk=list(topk), sst_attr_list=["gender"] + (["age"] if three_groups else []),
               eval_args={"mode": mode})
    kmax = max(topk)
    pos_idx = (rng.random((n_users, kmax)) < 0.15).astype(np.int64)
    pos_len = np.maximum(pos_idx.sum(1) + rng.integers(0, 4, n_users), 1)
    pos_idx[rng.integers(0, n_users, 3)] = 0                         # users without a hit
    ds = DataStruct()
    ds.set("rec.topk", torch.from_numpy(np.concatenate([pos_idx, pos_len[:, None]], axis=1)))
    pos_i = rng.integers(1, n_items, n_pos).astype(np.int64)
    neg_i = rng.integers(1, n_items, n_pos).astype(np.int64)
    pos_s = rng.random(n_pos).astype(np.float32)
    neg_s = (rng.random(n_pos) * 0.8).astype(np.float32)
    gender = rng.integers(0, 2, n_pos).astype(np.float32)
    ds.set("rec.positive_score", torch.from_numpy(pos_s))
    ds.set("data.positive_i", torch.from_numpy(pos_i))
    ds.set("rec.negative_score", torch.from_numpy(neg_s))
    ds.set("data.negative_i", torch.from_numpy(neg_i))
    ds.set("data.gender", torch.from_numpy(gender))
    out = {"topk": np.array(topk), "mode": np.array(mode), "rec_topk": ds.get("rec.topk").numpy(), "pos_score": pos_s,
           "pos_i": pos_i, "neg_score": neg_s, "neg_i": neg_i, "gender": gender}
    if three_groups:
        age = rng.integers(0, 3, n_pos).astype(np.int64)
        ds.set("data.age", torch.from_numpy(age))
        out["age"] = age
    result = {}
    for m in TOPK_METRICS:
        result.update(getattr(M, m)(cfg).calculate_metric(ds))
    for m in FAIR_METRICS:
        if three_groups and m not in ("NonParityUnfairness", "DifferentialFairness"):
            continue                                                  # the value-type metrics read sst_attr_list[0] only
        result.update(getattr(M, m)(cfg).calculate_metric(ds))
    out["result_json"] = np.array(json.dumps({k: float(v) for k, v in result.items()}))
    path = os.path.join(HERE, f"metrics_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {len(result)} metric values, {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    run_case("uni100_small", 1, 40, 30, 300, (5, 10), "uni100")
    run_case("uni100_large", 2, 600, 900, 20000, (1, 5, 10, 20), "uni100")
    run_case("full_small", 3, 40, 30, 300, (5, 10), "full")
    run_case("full_three_groups", 4, 100, 60, 2000, (10,), "full", three_groups=True)
    run_case("uni100_hot_items", 5, 50, 6, 5000, (3,), "uni100")        # few items: long (item, group) segments


if __name__ == "__main__":
    main()
