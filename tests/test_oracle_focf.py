"""Pins the oracle (oracle/focf.py) to golden vectors produced by the reference itself
(tests/golden/gen_focf_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import focf as O

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "focf_*.npz")))


def _load(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
@pytest.mark.parametrize("torch_adam", [True, False], ids=["torchadam", "restated_adam"])
def test_oracle_matches_reference_golden(path, torch_adam):
    z = _load(path)
    lr, wd, fw = z["hyper"][:3]
    snaps = tuple(int(s) for s in z["snaps"])
    out = O.train(str(z["objective"]), z["U0"], z["I0"], z["user_id"], z["item_id"], z["rating"], z["sst"],
                  float(lr), float(wd), float(fw), snaps=snaps, use_torch_adam=torch_adam,
                  clip_max_norm=float(z["clip_max_norm"]) if "clip_max_norm" in z else None)
    if "clip_max_norm" in z:
        np.testing.assert_allclose(out["grad_norm"], z["grad_norm"], rtol=1e-6)
    # same torch kernels as the reference -> the torch-Adam oracle is expected to be (near) bit-identical
    tol = dict(rtol=0, atol=0) if torch_adam else dict(rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(out["loss"], z["loss"], rtol=1e-7 if torch_adam else 1e-5)
    for key in ("pred_step1", "gradU_step1", "gradI_step1"):
        np.testing.assert_allclose(out[key], z[key], **tol, err_msg=key)
    for s in snaps:
        for tag in ("U", "I", "mU", "mI", "vU", "vI"):
            key = f"{tag}_after{s}"
            np.testing.assert_allclose(out[key], z[key], **tol, err_msg=key)


def test_predict_matches_golden():
    z = _load(os.path.join(os.path.dirname(__file__), "golden", "focf_value.npz"))
    import torch
    s = int(z["snaps"][-1])
    p = O.predict(torch.from_numpy(z[f"U_after{s}"]), torch.from_numpy(z[f"I_after{s}"]),
                  torch.from_numpy(z["user_id"][-1]), torch.from_numpy(z["item_id"][-1]), 5.0)
    np.testing.assert_array_equal(p.numpy(), z["predict_last"])
