"""CPU: the oracle's restatement of the reference's evaluation metrics (oracle/metrics.py) against golden vectors
produced by the reference's own metric classes (tests/golden/gen_metrics_golden.py)."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import metrics as OM

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN, "metrics_*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_metrics_match_reference_golden(path):
    z = np.load(path)
    ref = json.loads(str(z["result_json"]))
    sst = ["gender"] + (["age"] if "age" in z.files else [])
    got = OM.all_metrics(z, [int(k) for k in z["topk"]], str(z["mode"]), sst)
    assert set(ref) <= set(got)
    for k, v in ref.items():
        tol = 2e-6 if "Differential" in k or "NonParity" in k else 1e-9     # float32 tables in the reference
        assert abs(got[k] - v) <= tol * max(1.0, abs(v)), (k, got[k], v)


COL = sorted(glob.glob(os.path.join(GOLDEN, "collector_full_popularity*.npz")))


@pytest.mark.parametrize("path", COL, ids=[os.path.basename(p)[15:-4] for p in COL])
def test_exposure_metrics_match_reference_golden(path):
    """GiniIndex / PopularityPercentage (metrics.py:608-662, :749-821) on the `rec.items` the reference collected."""
    z = np.load(path)
    ref = json.loads(str(z["result_json"]))
    topk = [int(k) for k in z["topk"]]
    ratio = float(z["popularity_ratio"])
    got = OM.gini_index(z["collected.rec.items"], int(z["n_items"]), topk)
    got.update(OM.popularity_percentage(z["collected.rec.items"], z["train_items"], topk, None if ratio < 0 else ratio))
    for k, v in got.items():
        assert abs(v - ref[k]) <= 1e-9, (k, v, ref[k])
