"""GPU parity: NFCF (lazy tables + fp32-MFMA scorer + BCE / differential-fairness head) vs the reference's golden
vectors, through the plugin surface (model.calculate_loss -> loss.backward() -> optimizer.step())."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CASES = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "nfcf_*.npz"))
               if not p.endswith("_f64.npz"))   # <case>_f64.npz: the case's float64 companion (gen_nfcf_exact64.py)


class _DS:
    def __init__(self, n_users, n_items, gender):
        from fairrec.data.interaction import Interaction
        self._n = {"user_id": n_users, "item_id": n_items}
        self._uf = Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(gender)})

    def num(self, f):
        return self._n[f]

    def get_user_feature(self):
        return self._uf


@pytest.mark.parametrize("sharded", [False, True], ids=["single", "row_sharded"])
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_nfcf_training_matches_reference_golden(path, tmp_path, sharded, request):
    if sharded:     # the row-sharded engine as a 1-rank RCCL world: same goldens (fairrec/sharded_engine.py)
        request.getfixturevalue("rccl_world1")
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.model.fair_recommender.nfcf import NFCF
    f64 = path[:-4] + "_f64.npz"
    _run_case(np.load(path), sharded, exact=np.load(f64) if os.path.exists(f64) else None)


def _run_case(z, sharded=False, exact=None):
    """One recorded NFCF run (a golden .npz or a dict of the same layout) through the plugin surface on the GPU."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.model.fair_recommender.nfcf import NFCF
    from fairrec.optim import FusedLazyAdam
    stage = str(z["stage"])
    lr, wd, fw, p = (float(x) for x in z["hyper"])
    n_users, D = z["init.user_embedding.weight"].shape
    n_items = z["init.item_embedding.weight"].shape[0]
    cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": [int(h) for h in z["hidden"]],
                                            "dropout": p, "fair_weight": fw, "device": "cuda", "load_pretrain_path": None,
                                            "row_sharded": sharded})
    model = NFCF(cfg, _DS(n_users, n_items, z["gender"]))
    init = {k[5:]: torch.tensor(z[k]) for k in (z.files if hasattr(z, "files") else z) if k.startswith("init.")}
    if stage == "finetune":     # the state reset_params produced in the reference (its projection is pinned in the oracle test)
        model.load_pretrain_path = "reference-checkpoint"
        model.user_embedding.weight.requires_grad = False
    model.load_state_dict(init)
    model = model.to("cuda")
    model.train()
    # config clip_grad_norm (trainer.py:194-195): clipped inside step(), where the embedding gradient exists
    clip = {"max_norm": float(z["clip_max_norm"])} if "clip_max_norm" in z else None
    opt = FusedLazyAdam(model.hip_engine(), lr=lr, weight_decay=wd, sweep_period=3, clip_grad_norm=clip)
    n_layers = len(z["hidden"]) + 1
    snaps = set(int(s) for s in z["snaps"])
    losses = []
    for t in range(len(z["user_id"])):
        inter = Interaction({"user_id": torch.tensor(z["user_id"][t]), "item_id": torch.tensor(z["item_id"][t]),
                             "label": torch.tensor(z["label"][t]), "gender": torch.tensor(z["sst"][t])}).to("cuda")
        if p > 0:
            model.mlp_layers.forced_masks = [torch.tensor(z[f"mask{l}"][t]) for l in range(n_layers)]
        opt.zero_grad()
        loss = model.calculate_loss(inter)
        losses.append(loss.detach().reshape(1).clone())
        loss.backward()
        opt.step()
        if clip:
            np.testing.assert_allclose(float(opt.last_grad_norm), z["grad_norm"][t], rtol=1e-4)
        if (t + 1) in snaps:
            sd = model.state_dict()
            for k, v in sd.items():
                ref = z[f"after{t + 1}." + k]
                a = v.cpu().numpy()
                # between the reference's fp32 execution (the golden) and its float64 execution of the same steps
                # (tests/golden/gen_nfcf_exact64.py), give or take the tolerance -- see tests/test_pfcn_hip.py
                key = f"after{t + 1}." + k
                r64 = exact[key] if exact is not None and key in exact.files else ref
                lo, hi = np.minimum(ref, r64), np.maximum(ref, r64)
                dist = np.maximum(np.maximum(lo - a, a - hi), 0.0)
                out = dist > 1e-4 * np.abs(ref) + 2e-6
                # Adam divides a gradient by its own magnitude: where a weight gradient nearly cancels, ITS rounding noise
                # (different in every correct implementation) moves the element by a visible fraction of lr.  One element
                # per 10 000 (at least two) may therefore leave the band, by at most 0.25 % of what Adam can move anything
                # (steps * lr); measured: 1-2 of the 65 536 elements of the [128, 512] first-layer weight at D = 256, by
                # 6e-6 .. 9e-6 (which ones depends on the GEMM's summation order).
                assert out.sum() <= max(2, out.size // 10000) and (not out.any() or dist[out].max() <= 2.5e-3 * (t + 1) * lr), \
                    (k, t + 1, np.abs(a - ref).max(), np.abs(a - r64).max(), int(out.sum()))
    np.testing.assert_allclose(torch.cat(losses).cpu().numpy(), z["loss"], rtol=1e-4)
    model.hip_engine().check_device_errors()
    model.eval()
    model.mlp_layers.forced_masks = None
    with torch.no_grad():
        pr = model.predict(inter).cpu().numpy()
    np.testing.assert_allclose(pr, z["predict_last"], rtol=1e-4, atol=1e-6)


def test_nfcf_full_batch_at_the_baseline_width():
    """BASELINE.json configs[4]'s step shape -- embedding_size 256, mlp_hidden_size [128, 64], B = 8192, finetune (user table
    frozen, differential-fairness term) and pretrain -- on tables scaled down to what the CPU oracle's dense Adam sweeps in
    seconds (200 001 x 50 001): three steps of the HIP path against oracle/nfcf.py (pinned to the reference's goldens),
    every row of both tables compared, also the rows no batch touched."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import nfcf as O
    from fairrec.config import Config
    from fairrec.model.fair_recommender.nfcf import NFCF
    n_users, n_items, D, B, T, hidden = 200_001, 50_001, 256, 8192, 3, (128, 64)
    g = torch.Generator().manual_seed(12)
    gender = (torch.rand(n_users, generator=g) < 0.5).float().numpy()
    for stage, p in (("finetune", 0.0), ("pretrain", 0.2)):
        torch.manual_seed(5)
        cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": list(hidden), "dropout": p,
                                                "fair_weight": 0.1, "device": "cpu", "load_pretrain_path": None})
        m0 = NFCF(cfg, _DS(n_users, n_items, gender))
        z = {"stage": np.array(stage), "hyper": np.array([1e-3, 1e-6, 0.1, p]), "hidden": np.array(hidden), "gender": gender,
             "snaps": np.array([T])}
        for k, v in m0.state_dict().items():
            z["init." + k] = (v * 0.1 if "embedding" in k else v).detach().numpy().copy()     # N(0,1) tables saturate the sigmoid
        u = torch.randint(1, n_users, (T, B), generator=g)
        i = torch.randint(1, n_items // 8, (T, B), generator=g)                  # items repeat: segments of several members
        r = torch.randint(1, 6, (T, B), generator=g).float()
        z.update(user_id=u.numpy(), item_id=i.numpy(), label=(r >= 3).float().numpy(), sst=gender[u.numpy()])
        if p > 0:
            sizes = [2 * D] + list(hidden)
            for l, w in enumerate(sizes):
                z[f"mask{l}"] = (torch.rand(T, B, w, generator=g) >= p).to(torch.uint8).numpy()
        ref = O.train(z, snaps=(T,))
        z.update({k: v for k, v in ref.items() if k.startswith("after") or k == "loss"})
        U, I = torch.tensor(ref[f"after{T}.user_embedding.weight"]), torch.tensor(ref[f"after{T}.item_embedding.weight"])
        n_l = len(hidden) + 1
        Ws = [torch.tensor(ref[f"after{T}.mlp_layers.mlp_layers.{3 * l + 1}.weight"]) for l in range(n_l)]
        bs = [torch.tensor(ref[f"after{T}.mlp_layers.mlp_layers.{3 * l + 1}.bias"]) for l in range(n_l)]
        z["predict_last"] = O.forward(U, I, Ws, bs, u[-1], i[-1]).numpy()
        _run_case(z, sharded=False)


@pytest.mark.parametrize("sharded", [False, True], ids=["single", "row_sharded"])
def test_reset_params_on_the_device_then_finetune(tmp_path, sharded, request):
    """NFCF.reset_params (nfcf.py:49-67) with the model on the GPU: the pre-trained checkpoint is loaded, the gender direction
    projected out of the user table on the device (row-sharded: the two group means through one all-reduce, here as a 1-rank
    RCCL world), the table frozen, the item table re-initialised -- against the state the reference's reset_params left
    (golden `init.*`), and the finetune steps then run from THAT state (not from a state loaded out of the golden) to the
    reference's first snapshot."""
    if sharded:
        request.getfixturevalue("rccl_world1")
    from fairrec.config import Config
    from fairrec.model.fair_recommender.nfcf import NFCF
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "nfcf_finetune.npz"))
    n_users, D = z["pretrain_user_embedding"].shape
    n_items = z["init.item_embedding.weight"].shape[0]
    ck = tmp_path / "pre.pth"
    torch.save({"state_dict": {"user_embedding.weight": torch.tensor(z["pretrain_user_embedding"])}}, ck)
    lr, wd, fw, p = (float(x) for x in z["hyper"])
    cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": [int(h) for h in z["hidden"]], "dropout": p,
                                            "fair_weight": fw, "device": "cuda", "load_pretrain_path": str(ck),
                                            "row_sharded": sharded})
    with torch.device("cuda"):
        model = NFCF(cfg, _DS(n_users, n_items, z["gender"]))
    model = model.to("cuda")
    assert model.user_embedding.weight.is_cuda and not model.user_embedding.weight.requires_grad
    assert model.item_embedding.weight.requires_grad
    np.testing.assert_allclose(model.user_embedding.weight.detach().cpu().numpy(), z["init.user_embedding.weight"],
                               rtol=1e-5, atol=1e-6)
    # the rest of the state the reference started its finetune stage from (fresh item table, scorer), then its steps
    init = {k[5:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.") and "user_embedding" not in k}
    model.load_state_dict(init, strict=False)
    from fairrec.data.interaction import Interaction
    from fairrec.optim import FusedLazyAdam
    model.train()
    opt = FusedLazyAdam(model.hip_engine(), lr=lr, weight_decay=wd, sweep_period=3)
    n_layers = len(z["hidden"]) + 1
    first = min(int(s) for s in z["snaps"])
    losses = []
    for t in range(first):
        inter = Interaction({"user_id": torch.tensor(z["user_id"][t]), "item_id": torch.tensor(z["item_id"][t]),
                             "label": torch.tensor(z["label"][t]), "gender": torch.tensor(z["sst"][t])}).to("cuda")
        if p > 0:
            model.mlp_layers.forced_masks = [torch.tensor(z[f"mask{l}"][t]) for l in range(n_layers)]
        opt.zero_grad()
        loss = model.calculate_loss(inter)
        losses.append(float(loss))
        loss.backward()
        opt.step()
    np.testing.assert_allclose(losses, z["loss"][:first], rtol=1e-4)
    for k, v in model.state_dict().items():
        ref = z[f"after{first}." + k]
        np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=2e-4, atol=1e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)
