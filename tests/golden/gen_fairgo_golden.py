#!/usr/bin/env python3
"""Golden vectors for the FairGo_PMF hot path by RUNNING THE REFERENCE (build container
only).  FairGo_GCN's finetune stage is line-identical to FairGo_PMF's, so these vectors pin it too;
its torch_geometric pretrain stage stays unpinned.

Step kinds, in the order FairGoTrainer drives them (trainer.py:606-736, optimizers :837-862):
  "P": pretrain  -- calculate_loss(inter, None), optimizer_pretrain = Adam([U, I])
  "F": finetune filter pass -- calculate_loss(inter, sst_list), optimizer_filter = Adam(filters)
  "D": finetune discriminator pass -- calculate_dis_loss(inter, sst_list), optimizer_dis = Adam(discriminators [+ aggr_layer])
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402
from recbole.data.interaction import Interaction  # noqa: E402
from recbole.model.fair_recommender.fairgo_pmf import FairGo_PMF  # noqa: E402


def _fairgo_gcn_class():
    """The reference's FairGo_GCN class itself, for its FINETUNE stage (BASELINE.json configs[3]).  Its module imports
    torch_geometric.nn.GCN (fairgo_gcn.py:20), which this image does not have; the finetune stage never calls it
    (fairgo_gcn.py:175-176 applies self.gcn in the pretrain stage only), so a parameter-free placeholder module whose
    forward raises stands in for the import.  The pretrain stage stays unpinned."""
    import types
    if "torch_geometric.nn" not in sys.modules:
        class GCN(torch.nn.Module):
            def __init__(self, **kw):
                super().__init__()

            def forward(self, *a, **kw):
                raise RuntimeError("torch_geometric is not installed: the GCN pretrain stage can not be run here")
        tg, tgn = types.ModuleType("torch_geometric"), types.ModuleType("torch_geometric.nn")
        tgn.GCN = GCN
        tg.nn = tgn
        sys.modules["torch_geometric"], sys.modules["torch_geometric.nn"] = tg, tgn
    from recbole.model.fair_recommender.fairgo_gcn import FairGo_GCN
    return FairGo_GCN


class _Cfg(dict):
    def __getitem__(self, k):
        return self.get(k, None)


class _FakeDataset:
    def __init__(self, n_users, n_items, feats, tu, ti, tr):
        self._n = {"user_id": n_users, "item_id": n_items}
        cols = {"user_id": torch.arange(n_users)}
        cols.update({k: torch.from_numpy(v) for k, v in feats.items()})
        self._uf = Interaction(cols)
        self.inter_feat = {"rating": torch.from_numpy(tr)}
        self._coo = sp.coo_matrix((tr, (tu, ti)), shape=(n_users, n_items))

    def num(self, field):
        return self._n[field]

    def get_user_feature(self):
        return self._uf

    def inter_matrix(self, form="coo", value_field=None):
        return self._coo


def dump(model, prefix, out):
    for k, v in model.state_dict().items():
        out[f"{prefix}.model.{k}"] = v.detach().numpy().copy()
    for s, m in model.filter_layer_dict.items():
        for k, v in m.state_dict().items():
            out[f"{prefix}.filter.{s}.{k}"] = v.detach().numpy().copy()
    for s, m in model.dis_layer_dict.items():
        for k, v in m.state_dict().items():
            out[f"{prefix}.dis.{s}.{k}"] = v.detach().numpy().copy()


def run_case(name, aggr, attrs, phases, sst_lists, seed, n_layers=2, D=8, B=32, n_users=30, n_items=25, lr=1e-3, wd=1e-4,
             fair_weight=0.1, filter_hidden=(16, 8), dis_hidden=(8, 4), vs_weights=(4, 1), n_train=200, gcn=False):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    feats = {"gender": rng.integers(0, 2, size=n_users).astype(np.float32), "age": rng.integers(0, 3, size=n_users).astype(np.int64)}
    feats["age"][1:4] = [0, 1, 2]
    feats["gender"][1:3] = [0.0, 1.0]
    pairs = rng.choice((n_users - 1) * (n_items - 1), size=n_train, replace=False)   # distinct (user, item) pairs
    tu, ti = pairs // (n_items - 1) + 1, pairs % (n_items - 1) + 1
    tr = rng.integers(1, 6, size=n_train).astype(np.float32)
    cfg = _Cfg(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device=torch.device("cpu"),
               RATING_FIELD="rating", n_layers=n_layers, activation="leakyrelu", embedding_size=D,
               dis_hidden_size_list=list(dis_hidden), filter_hidden_size_list=list(filter_hidden), sst_attr_list=list(attrs),
               fair_weight=fair_weight, load_pretrain_weight=False, aggr_method=aggr, vs_weights=list(vs_weights))
    if gcn:
        cfg.update(hidden_channels=32, gcn_n_layers=2, gcn_dropout=0.2, gcn_act="relu")
    model = (_fairgo_gcn_class() if gcn else FairGo_PMF)(cfg, _FakeDataset(n_users, n_items, feats, tu, ti, tr))
    out = {"model": np.array("FairGo_GCN" if gcn else "FairGo_PMF"), "aggr": np.array(aggr), "attrs": np.array(list(attrs)), "phases": np.array(list(phases)),
           "sst_lists": np.array([",".join(s) for s in sst_lists]), "hyper": np.array([lr, wd, fair_weight]),
           "n_layers": np.array(n_layers), "filter_hidden": np.array(filter_hidden), "dis_hidden": np.array(dis_hidden),
           "vs_weights": np.array(vs_weights, dtype=np.float32), "gender": feats["gender"], "age": feats["age"],
           "train_user": tu.astype(np.int64), "train_item": ti.astype(np.int64), "train_rating": tr}
    L = model.norm_rating_matrix.coalesce()
    out["L_row"], out["L_col"] = L.indices()[0].numpy(), L.indices()[1].numpy()
    out["L_val"] = L.values().numpy()
    dump(model, "init", out)
    opt_p = torch.optim.Adam([model.user_embedding_layer.weight, model.item_embedding_layer.weight], lr=lr, weight_decay=wd)
    dis_groups = [{"params": m.parameters()} for m in model.dis_layer_dict.values()]
    if aggr == "LBA":
        dis_groups += [{"params": list(model.aggr_layer.parameters())}]
    opt_d = torch.optim.Adam(dis_groups, lr=lr, weight_decay=wd)
    opt_f = torch.optim.Adam([{"params": m.parameters()} for m in model.filter_layer_dict.values()], lr=lr, weight_decay=wd)
    cols = {k: [] for k in ("user_id", "item_id", "rating")}
    losses = []
    for t, ph in enumerate(phases):
        sel = rng.integers(0, n_train, size=B)
        u, i, r = tu[sel], ti[sel], tr[sel]
        inter = Interaction({"user_id": torch.from_numpy(u.astype(np.int64)), "item_id": torch.from_numpy(i.astype(np.int64)),
                             "rating": torch.from_numpy(r), "gender": torch.from_numpy(feats["gender"][u]),
                             "age": torch.from_numpy(feats["age"][u])})
        sl = list(sst_lists[t])
        if ph == "P":
            model.train_stage = "pretrain"
            opt_p.zero_grad()
            loss = model.calculate_loss(inter, None)
            loss.backward()
            opt_p.step()
        else:
            model.train_stage = "finetune"
            opt = opt_f if ph == "F" else opt_d
            opt.zero_grad()
            loss = model.calculate_loss(inter, sl) if ph == "F" else model.calculate_dis_loss(inter, sl)
            loss.backward()
            opt.step()
        losses.append(float(loss.item()))
        for k, v in (("user_id", u), ("item_id", i), ("rating", r)):
            cols[k].append(v)
    dump(model, "final", out)
    out["user_id"] = np.stack(cols["user_id"]).astype(np.int64)
    out["item_id"] = np.stack(cols["item_id"]).astype(np.int64)
    out["rating"] = np.stack(cols["rating"]).astype(np.float32)
    out["loss"] = np.array(losses)
    with torch.no_grad():
        out["predict_last"] = model.predict(inter).numpy().copy()     # finetune: filters of ALL attributes (fairgo_pmf.py:252-256)
    path = os.path.join(HERE, f"fairgo_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: loss {losses[0]:.5f} -> {losses[-1]:.5f}  {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    g, ga = ("gender",), ("gender", "age")
    run_case("wap", "WAP", g, "PPPFDFD", [g] * 7, seed=1)
    run_case("lba", "LBA", g, "PPFDFDD", [g] * 7, seed=2)
    run_case("lva", "LVA", g, "PFDFD", [g] * 5, seed=3)
    run_case("wap2", "WAP", ga, "PFDFDFD", [ga, ga, ga, g, g, ("age",), ("age",)], seed=4)
    run_case("lva2", "LVA", ga, "PFDFD", [ga, ga, ga, ("age",), ("age",)], seed=5)
    run_case("one_layer", "WAP", g, "FDFD", [g] * 4, seed=6, n_layers=1)
    # BASELINE.json configs[3]'s widths through the reference's FairGo_GCN class (finetune stage, fairgo_gcn.py:173-250):
    # embedding 128, filters [128, 64], discriminators [16, 8, 4], WAP, two graph layers
    run_case("gcn_wap_d128", "WAP", g, "FDFDFD", [g] * 6, seed=7, D=128, B=256, n_users=300, n_items=250, n_train=4000,
             filter_hidden=(128, 64), dis_hidden=(16, 8, 4), gcn=True)


if __name__ == "__main__":
    main()
