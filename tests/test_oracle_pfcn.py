"""Pins oracle/pfcn.py to golden vectors produced by the reference (tests/golden/gen_pfcn_golden.py). CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import pfcn as O

CASES = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pfcn_*.npz"))
               if not p.endswith("_f64.npz"))   # <case>_f64.npz: the case's float64 companion (the reference in float64: gen_pfcn_golden.py::_run_f64)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference_golden(path):
    z = np.load(path)
    out = O.train(z)
    np.testing.assert_allclose(out["loss"], z["loss"], rtol=2e-6, atol=1e-7)
    for k, v in out.items():
        if k.startswith("final."):
            np.testing.assert_allclose(v, z[k], rtol=1e-5, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(out["predict_last"], z["predict_last"], rtol=1e-5, atol=1e-7)
