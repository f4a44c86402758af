"""Pins oracle/fairgo.py to golden vectors produced by the reference (tests/golden/gen_fairgo_golden.py). CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import fairgo as O

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fairgo_*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference_golden(path):
    z = np.load(path)
    out = O.train(z)
    np.testing.assert_allclose(out["loss"], z["loss"], rtol=2e-6, atol=1e-7)
    for k, v in out.items():
        if k.startswith("final."):
            np.testing.assert_allclose(v, z[k], rtol=1e-5, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(out["predict_last"], z["predict_last"], rtol=1e-5, atol=1e-7)


def test_norm_rating_matrix_matches_reference():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "fairgo_wap.npz"))
    n_users = z["init.model.user_embedding_layer.weight"].shape[0]
    n_items = z["init.model.item_embedding_layer.weight"].shape[0]
    L = O.norm_rating_matrix(n_users, n_items, z["train_user"], z["train_item"], z["train_rating"])
    np.testing.assert_array_equal(L.indices()[0].numpy(), z["L_row"])
    np.testing.assert_array_equal(L.indices()[1].numpy(), z["L_col"])
    np.testing.assert_allclose(L.values().numpy(), z["L_val"], rtol=1e-6)
