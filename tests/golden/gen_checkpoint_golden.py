#!/usr/bin/env python3
"""Golden fixture for checkpoint interchange (next-row f-4): RUNS the reference's Trainer._save_checkpoint
(trainer.py:221-240) on a small FOCF model after three reference training steps and records, as data only,
  * the key names of the checkpoint dict, of its state_dict and of its torch.optim.Adam state_dict,
  * every tensor in it (weights, exp_avg, exp_avg_sq, step), epoch / cur_step / best_valid_score,
  * the batches, and the weights the REFERENCE reaches one more step after the checkpoint (what a resumed run must reach).
Build container only (needs /root/reference)."""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402
from recbole.data.interaction import Interaction  # noqa: E402
from recbole.model.fair_recommender.focf import FOCF  # noqa: E402
from recbole.trainer.trainer import Trainer  # noqa: E402
from gen_focf_golden import _Cfg, _FakeDataset, make_batches  # noqa: E402


def main():
    seed, n_users, n_items, D, B, T = 77, 50, 40, 8, 64, 4
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    cfg = _Cfg(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device=torch.device("cpu"),
               embedding_size=D, RATING_FIELD="rating", sst_attr_list=["gender"], fair_weight=0.5, fair_objective="value",
               model="FOCF", learner="adam", learning_rate=1e-3, weight_decay=1e-3)
    model = FOCF(cfg, _FakeDataset(n_users, n_items, 5.0))
    out = {"U0": model.user_embedding_layer.weight.detach().numpy().copy(),
           "I0": model.item_embedding_layer.weight.detach().numpy().copy()}
    u, i, r, s = make_batches(rng, T, B, n_users, n_items, "uniform")
    out.update(user_id=u, item_id=i, rating=r, sst=s)
    tr = object.__new__(Trainer)              # the reference's class; its __init__ wants the whole config / logging plumbing
    tr.config, tr.model = cfg, model
    tr.learner, tr.learning_rate, tr.weight_decay = "adam", 1e-3, 1e-3
    tr.optimizer = tr._build_optimizer()      # trainer.py:114-153
    tr.cur_step, tr.best_valid_score = 2, 0.125
    tmp = tempfile.mkdtemp()
    tr.saved_model_file = os.path.join(tmp, "FOCF-ref.pth")

    def step(t):
        inter = Interaction({"user_id": torch.from_numpy(u[t]), "item_id": torch.from_numpy(i[t]),
                             "rating": torch.from_numpy(r[t]), "gender": torch.from_numpy(s[t])})
        tr.optimizer.zero_grad()
        loss = model.calculate_loss(inter)
        loss.backward()
        tr.optimizer.step()
        return float(loss.item())

    for t in range(T - 1):
        step(t)
    tr._save_checkpoint(5, verbose=False)                       # trainer.py:221-240
    ck = torch.load(tr.saved_model_file)
    out["ck_keys"] = np.array(json.dumps(sorted(k for k in ck if k != "config")))
    out["epoch"], out["cur_step"], out["best_valid_score"] = np.array(ck["epoch"]), np.array(ck["cur_step"]), np.array(ck["best_valid_score"])
    out["state_dict_keys"] = np.array(json.dumps(list(ck["state_dict"].keys())))
    for k, v in ck["state_dict"].items():
        out["sd::" + k] = v.numpy().copy()
    opt = ck["optimizer"]
    out["opt_state_keys"] = np.array(json.dumps({str(k): sorted(v.keys()) for k, v in opt["state"].items()}))
    pg = [{k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in g.items()} for g in opt["param_groups"]]
    out["opt_param_groups"] = np.array(json.dumps(pg, default=str))
    for k, st in opt["state"].items():
        for name, v in st.items():
            out[f"opt::{k}::{name}"] = (v.numpy().copy() if torch.is_tensor(v) else np.array(v))
    out["other_parameter"] = np.array(json.dumps(ck["other_parameter"]))
    out["loss_next"] = np.array(step(T - 1))                    # the reference continues one step past the checkpoint
    out["U_next"] = model.user_embedding_layer.weight.detach().numpy().copy()
    out["I_next"] = model.item_embedding_layer.weight.detach().numpy().copy()
    out["hyper"] = np.array([1e-3, 1e-3, 0.5])
    path = os.path.join(HERE, "checkpoint_focf.npz")
    np.savez_compressed(path, **out)
    print(path, json.loads(str(out["ck_keys"])), json.loads(str(out["opt_state_keys"])), json.loads(str(out["opt_param_groups"])))


if __name__ == "__main__":
    main()
