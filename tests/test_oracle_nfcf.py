"""Pins oracle/nfcf.py to golden vectors produced by the reference (tests/golden/gen_nfcf_golden.py). CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import nfcf as O

CASES = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "nfcf_*.npz"))
               if not p.endswith("_f64.npz"))   # <case>_f64.npz: the case's float64 companion (the reference in float64: gen_nfcf_golden.py::_run_f64)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference_golden(path):
    z = np.load(path)
    snaps = tuple(int(s) for s in z["snaps"])
    out = O.train(z, snaps=snaps)
    np.testing.assert_allclose(out["loss"], z["loss"], rtol=1e-6)
    for key in ("grad_step1.mlp0", "grad_step1.item"):
        np.testing.assert_allclose(out[key], z[key], rtol=1e-5, atol=1e-8, err_msg=key)
    for k, v in out.items():
        if k.startswith("after"):
            np.testing.assert_allclose(v, z[k], rtol=1e-5, atol=1e-7, err_msg=k)


def test_reset_params_projection_matches_reference():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "nfcf_finetune.npz"))
    got = O.reset_user_embedding(torch.tensor(z["pretrain_user_embedding"]), torch.tensor(z["gender"][1:]))
    np.testing.assert_allclose(got.numpy(), z["init.user_embedding.weight"], rtol=1e-6, atol=1e-7)
