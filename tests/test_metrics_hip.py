"""GPU: the device evaluation metrics (fairrec.evaluator, csrc/metrics.hip) against golden vectors produced by the
reference's own metric classes, and against the oracle at a size the reference's Python loops would take minutes for."""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN, "metrics_*.npz")))


def _collected(z, sst):
    d = lambda k: torch.from_numpy(z[k]).cuda()
    c = {"rec.topk": d("rec_topk"), "rec.positive_score": d("pos_score"), "data.positive_i": d("pos_i"),
         "rec.negative_score": d("neg_score"), "data.negative_i": d("neg_i")}
    for s in sst:
        c["data." + s] = d(s)
    return c


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_device_metrics_match_reference_golden(path):
    from fairrec.config import Config
    from fairrec.evaluator import Evaluator
    z = np.load(path)
    ref = json.loads(str(z["result_json"]))
    sst = ["gender"] + (["age"] if "age" in z.files else [])
    metrics = ["Hit", "MRR", "NDCG", "Recall", "Precision", "MAP", "NonParityUnfairness", "DifferentialFairness"]
    if len(sst) == 1:
        metrics += ["ValueUnfairness", "AbsoluteUnfairness", "UnderUnfairness", "OverUnfairness"]
    cfg = Config(config_dict={"metrics": metrics, "topk": [int(k) for k in z["topk"]], "metric_decimal_place": 10,
                              "sst_attr_list": sst, "eval_args": {"mode": str(z["mode"])}, "device": "cuda"})
    got = Evaluator(cfg).evaluate(_collected(z, sst))
    assert set(got) == set(ref)
    for k, v in ref.items():
        tol = 5e-6 if "Differential" in k or "NonParity" in k else 1e-8     # float32 tables in the reference
        assert abs(got[k] - v) <= tol * max(1.0, abs(v)), (k, got[k], v)


def test_device_metrics_match_oracle_at_scale():
    """1 M positive pairs over 50 k items (the reference loops over interactions in Python and over items x groups with
    boolean masks, metrics.py:950-960 / :1331-1335; the oracle is the vectorised restatement)."""
    from fairrec.evaluator import fairness_metrics, topk_metrics
    from oracle import metrics as OM
    rng = np.random.default_rng(0)
    n, n_items, U, K = 1_000_000, 50_000, 200_000, 20
    z = {"pos_score": rng.random(n).astype(np.float32), "pos_i": rng.integers(1, n_items, n).astype(np.int64),
         "neg_score": rng.random(n).astype(np.float32), "neg_i": rng.integers(1, n_items, n).astype(np.int64),
         "gender": rng.integers(0, 2, n).astype(np.float32)}
    pos = (rng.random((U, K)) < 0.1).astype(np.int32)
    z["rec_topk"] = np.concatenate([pos, np.maximum(pos.sum(1, keepdims=True), 1)], axis=1).astype(np.int32)
    ref = OM.all_metrics(z, [1, 10, 20], "uni100", ["gender"])
    d = lambda k: torch.from_numpy(z[k]).cuda()
    got = topk_metrics(d("rec_topk"), [1, 10, 20])
    got.update(fairness_metrics(d("pos_score"), d("pos_i"), {"gender": d("gender")}, d("neg_score"), d("neg_i"), "uni100"))
    assert set(got) == set(ref)
    for k, v in ref.items():
        tol = 2e-5 if "Differential" in k or "NonParity" in k else 1e-9
        assert abs(got[k] - v) <= tol * max(1.0, abs(v)), (k, got[k], v)
