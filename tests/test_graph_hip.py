"""GPU: a whole training step captured in a hipGraph (fairrec.graph.GraphedStep: device-resident step counters,
fr_table.step_dev) reproduces the reference's golden vectors exactly like the eager path -- first steps eager, one capture,
the remaining steps are replays of the same graph on new batches."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


class _DS:
    def __init__(self, n_users, n_items, gender):
        from fairrec.data.interaction import Interaction
        self._n = {"user_id": n_users, "item_id": n_items}
        self._uf = Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(gender)})

    def num(self, f):
        return self._n[f]

    def get_user_feature(self):
        return self._uf


@pytest.mark.parametrize("case", ["nfcf_pretrain", "nfcf_pretrain_wd", "nfcf_finetune", "nfcf_finetune_d64"])
def test_nfcf_graphed_steps_match_reference_golden(case):
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.graph import GraphedStep
    from fairrec.model.fair_recommender.nfcf import NFCF
    from fairrec.optim import FusedLazyAdam
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    lr, wd, fw, p = (float(x) for x in z["hyper"])
    assert p == 0.0            # recorded dropout masks are host inputs; the graphed path draws its own (test below)
    n_users, D = z["init.user_embedding.weight"].shape
    n_items = z["init.item_embedding.weight"].shape[0]
    cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": [int(h) for h in z["hidden"]],
                                            "dropout": p, "fair_weight": fw, "device": "cuda", "load_pretrain_path": None})
    model = NFCF(cfg, _DS(n_users, n_items, z["gender"]))
    if str(z["stage"]) == "finetune":
        model.load_pretrain_path = "reference-checkpoint"
        model.user_embedding.weight.requires_grad = False
    model.load_state_dict({k[5:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.")})
    model = model.to("cuda").train()
    eng = model.hip_engine()
    opt = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=3)
    gs = GraphedStep(eng, opt, model.calculate_loss, eager_steps=2)
    snaps = set(int(s) for s in z["snaps"])
    losses = []
    T = len(z["user_id"])
    for t in range(T):
        inter = Interaction({"user_id": torch.tensor(z["user_id"][t]), "item_id": torch.tensor(z["item_id"][t]),
                             "label": torch.tensor(z["label"][t]), "gender": torch.tensor(z["sst"][t])})
        losses.append(gs(inter).reshape(1).clone())
        if (t + 1) in snaps:
            for k, v in model.state_dict().items():
                ref = z[f"after{t + 1}." + k]
                a = v.cpu().numpy()
                assert (np.abs(a - ref) <= 1e-4 * np.abs(ref) + 2e-6).all(), (k, t + 1, np.abs(a - ref).max())
    assert gs.graph is not None and T > 3
    np.testing.assert_allclose(torch.cat(losses).cpu().numpy(), z["loss"], rtol=1e-4)
    eng.sync_steps()
    assert eng._tables["item_embedding.weight"].step == T
    assert all(d.step == T for d in eng._dense.values())
    eng.check_device_errors()


@pytest.mark.parametrize("case", ["pfcn_pmf_none", "pfcn_bmf_none", "pfcn_mlp_none", "pfcn_dmf_none"])
def test_pfcn_graphed_steps_match_reference_golden(case):
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.graph import GraphedStep
    from fairrec.optim import FusedLazyAdam
    from fairrec.utils import get_model
    from test_pfcn_hip import _DS as PDS
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    name = str(z["model"])
    lr, wd, dis_weight, p = (float(x) for x in z["hyper"])
    utab = "user_embedding" if name == "PFCN_MLP" else "user_embedding_layer"
    itab = "item_embedding" if name == "PFCN_MLP" else "item_embedding_layer"
    n_users, D = z[f"init.model.{utab}.weight"].shape
    n_items = z[f"init.model.{itab}.weight"].shape[0]
    cfg = Config(model=name, config_dict={"embedding_size": D, "sst_attr_list": [str(a) for a in z["attrs"]],
                                          "filter_mode": "none", "dis_hidden_size_list": [int(h) for h in z["dis_hidden"]],
                                          "dis_dropout": p, "dis_weight": dis_weight, "device": "cuda", "dropout": 0.0,
                                          "mlp_hidden_size_list": [8, 4], "num_layers": 2, "mlp_dropout": 0.0,
                                          "mlp_activation": "relu", "dis_activation": "leakyrelu", "activation": "leakyrelu"})
    model = get_model(name)(cfg, PDS(n_users, n_items, z))
    model.load_state_dict({k[11:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.model.")})
    model = model.to("cuda").train()
    eng = model.hip_engine()
    opt = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=2)
    gs = GraphedStep(eng, opt, lambda inter: model.calculate_loss(inter, None), eager_steps=1)
    losses = []
    for t in range(len(z["user_id"])):
        inter = Interaction({k: torch.tensor(z[k][t]) for k in ("user_id", "item_id", "neg_item_id")})
        inter["gender"] = torch.tensor(z["gender"][z["user_id"][t]])
        losses.append(gs(inter).reshape(1).clone())
    np.testing.assert_allclose(torch.cat(losses).cpu().numpy(), z["loss"], rtol=2e-4)
    sd = model.state_dict()
    for k, v in sd.items():
        ref = z["final.model." + k]
        a = v.cpu().numpy()
        assert (np.abs(a - ref) <= 1e-4 * np.abs(ref) + 2e-6).all(), (k, np.abs(a - ref).max())


def test_graphed_step_draws_fresh_dropout_masks_and_handles_odd_batches():
    """Dropout inside the captured step uses torch's graph-safe generator: two replays on the SAME batch give different
    losses; a batch of another size falls back to an eager step."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.graph import GraphedStep
    from fairrec.model.fair_recommender.nfcf import NFCF
    from fairrec.optim import FusedLazyAdam
    g = torch.Generator().manual_seed(0)
    n_users, n_items, B = 300, 200, 512
    cfg = Config(model="NFCF", config_dict={"embedding_size": 32, "mlp_hidden_size": [64, 32], "dropout": 0.5,
                                            "device": "cuda", "load_pretrain_path": None})
    model = NFCF(cfg, _DS(n_users, n_items, np.zeros(n_users, dtype=np.float32))).to("cuda").train()
    eng = model.hip_engine()
    opt = FusedLazyAdam(eng, lr=0.0, weight_decay=0.0)       # lr 0: parameters frozen, only the masks change
    gs = GraphedStep(eng, opt, model.calculate_loss, eager_steps=1)

    def batch(n):
        return Interaction({"user_id": torch.randint(1, n_users, (n,), generator=g),
                            "item_id": torch.randint(1, n_items, (n,), generator=g),
                            "label": (torch.rand(n, generator=g) < 0.5).float()})
    b0 = batch(B)
    vals = [float(gs(b0)) for _ in range(5)]
    assert gs.graph is not None
    assert len(set(round(v, 7) for v in vals[1:])) > 1 and all(np.isfinite(vals))
    assert np.isfinite(float(gs(batch(B // 2 + 3))))          # odd size: eager fallback
    eng.sync_steps()
    assert eng._tables["item_embedding.weight"].step == 6
