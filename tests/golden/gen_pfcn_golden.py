#!/usr/bin/env python3
"""Golden vectors for the PFCN hot path by RUNNING THE REFERENCE (build container only).

Models: PFCN_PMF and PFCN_BiasedMF (pfcn_pmf.py / pfcn_biasedmf.py) in filter modes none / sm / cm.
The step sequence is the one PFCNTrainer drives (trainer.py:875-930, optimizers of :1201-1235): "F" steps
`optimizer_filter.zero_grad(); calculate_loss(inter, sst_list).backward(); optimizer_filter.step()` and "D" steps with
`calculate_dis_loss` and `optimizer_dis`; with filter_mode none a single Adam over model.parameters().
Discriminator dropout masks are RECORDED (the reference's nn.Dropout modules are swapped for a module applying a
given Bernoulli mask with the same 1/(1-p) scaling) so inputs and outputs are both known.  BatchNorm runs on batch
statistics exactly as in the reference (the dict-held MLPs are never put in eval mode).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from recbole.data.interaction import Interaction  # noqa: E402
from recbole.model.fair_recommender.pfcn_biasedmf import PFCN_BiasedMF  # noqa: E402
from recbole.model.fair_recommender.pfcn_dmf import PFCN_DMF  # noqa: E402
from recbole.model.fair_recommender.pfcn_mlp import PFCN_MLP  # noqa: E402
from recbole.model.fair_recommender.pfcn_pmf import PFCN_PMF  # noqa: E402


class _Cfg(dict):
    def __getitem__(self, k):
        return self.get(k, None)


class _FakeDataset:
    def __init__(self, n_users, n_items, feats):
        self._n = {"user_id": n_users, "item_id": n_items}
        cols = {"user_id": torch.arange(n_users)}
        cols.update({k: torch.from_numpy(v) for k, v in feats.items()})
        self._uf = Interaction(cols)

    def num(self, field):
        return self._n[field]

    def get_user_feature(self):
        return self._uf


class RecordedDropout(nn.Module):
    def __init__(self, p, queue):
        super().__init__()
        self.p, self.queue = p, queue

    def forward(self, x):
        if self.p == 0.0:
            return x
        m = self.queue.pop(0)
        assert m.shape == x.shape, (m.shape, x.shape)
        return x * (m / (1.0 - self.p))


def patch_dropout(mlp, p, queue):
    seq = mlp.mlp_layers
    for k, mod in enumerate(seq):
        if isinstance(mod, (nn.Dropout, RecordedDropout)):
            seq[k] = RecordedDropout(p, queue)


def dump_state(model, mode, prefix, out):
    for k, v in model.state_dict().items():
        out[f"{prefix}.model.{k}"] = v.detach().numpy().copy()
    if mode != "none":
        for idx, mlp in model.filter_layer.items():
            for k, v in mlp.state_dict().items():
                out[f"{prefix}.filter.{idx}.{k}"] = v.detach().numpy().copy()
        for sst, mlp in model.dis_layer_dict.items():
            for k, v in mlp.state_dict().items():
                out[f"{prefix}.dis.{sst}.{k}"] = v.detach().numpy().copy()


ONLY = [s for s in os.environ.get("GOLDEN_ONLY", "").split(",") if s]


def run_case(name, *args, **kwargs):
    if ONLY and name not in ONLY:
        return
    return _run_case(name, *args, **kwargs)


def _run_case(name, cls, mode, attrs, phases, sst_lists, D, B, dis_hidden, seed, p_drop=0.3, lr=1e-3, wd=1e-4,
             dis_weight=10.0, n_users=40, n_items=30, clip=None):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    feats = {"gender": rng.integers(0, 2, size=n_users).astype(np.float32),
             "age": rng.integers(0, 3, size=n_users).astype(np.int64)}
    feats["age"][1:4] = [0, 1, 2]          # every class present among users 1..
    feats["gender"][1:3] = [0.0, 1.0]
    cfg = _Cfg(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device=torch.device("cpu"),
               embedding_size=D, sst_attr_list=list(attrs), filter_mode=mode, dis_dropout=p_drop, dis_weight=dis_weight,
               dis_hidden_size_list=list(dis_hidden), activation="leakyrelu",
               # PFCN_MLP scorer / PFCN_DMF towers: their own dropout is 0 here (the mask path is pinned by the
               # discriminators and by the NFCF vectors); sizes kept small
               dropout=0.0, mlp_hidden_size_list=[8, 4], num_layers=2, mlp_dropout=0.0, mlp_activation="relu",
               dis_activation="leakyrelu")
    model = cls(cfg, _FakeDataset(n_users, n_items, feats))
    out = {"mode": np.array(mode), "model": np.array(cls.__name__), "attrs": np.array(list(attrs)),
           "dis_hidden": np.array(dis_hidden), "hyper": np.array([lr, wd, dis_weight, p_drop]),
           "gender": feats["gender"], "age": feats["age"], "phases": np.array(list(phases)),
           "sst_lists": np.array([",".join(s) for s in sst_lists])}
    queues = {}
    if mode != "none":
        for sst, mlp in model.dis_layer_dict.items():
            queues[sst] = []
            patch_dropout(mlp, p_drop, queues[sst])
    dump_state(model, mode, "init", out)
    if mode == "none":
        opt_f = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)                 # trainer.py:139
        opt_d = None
    else:
        if cls is PFCN_MLP:                                                                   # trainer.py:1193-1198
            groups = [{"params": model.user_embedding.weight}, {"params": model.item_embedding.weight}]
        else:
            groups = [{"params": model.user_embedding_layer.weight}, {"params": model.item_embedding_layer.weight}]
        groups += [{"params": m.parameters()} for m in model.filter_layer.values()]
        if cls is PFCN_BiasedMF:                                                              # trainer.py:1205-1211
            groups += [{"params": model.user_bias.weight}, {"params": model.item_bias.weight}, {"params": model.global_bias}]
        if cls is PFCN_MLP:
            groups += [{"params": model.mlp_layer.parameters()}]
        if cls is PFCN_DMF:                                                                   # trainer.py:1219-1224
            groups += [{"params": model.user_mlp.parameters()}, {"params": model.item_mlp.parameters()}]
        opt_f = torch.optim.Adam(groups, lr=lr, weight_decay=wd)
        opt_d = torch.optim.Adam([{"params": m.parameters()} for m in model.dis_layer_dict.values()], lr=lr, weight_decay=wd)
    T = len(phases)
    cols = {k: [] for k in ("user_id", "item_id", "neg_item_id")}
    losses, masks = [], {sst: [[] for _ in range(T)] for sst in attrs}
    norms = []
    dis_sizes = [D] + list(dis_hidden)
    for t in range(T):
        u = rng.integers(1, n_users, size=B)
        pi = rng.integers(1, n_items, size=B)
        ni = rng.integers(1, n_items, size=B)
        inter = Interaction({"user_id": torch.from_numpy(u), "item_id": torch.from_numpy(pi),
                             "neg_item_id": torch.from_numpy(ni), "gender": torch.from_numpy(feats["gender"][u]),
                             "age": torch.from_numpy(feats["age"][u])})
        sl = list(sst_lists[t]) if mode != "none" else None
        if mode != "none":
            for sst in sl:                       # one discriminator pass per selected attribute, in sst_list order
                for w in dis_sizes:
                    m = (torch.rand(B, w) >= p_drop).float()
                    queues[sst].append(m)
                    masks[sst][t].append(m.numpy().astype(np.uint8))
        if phases[t] == "F":
            opt_f.zero_grad()
            loss = model.calculate_loss(inter, sl)
            loss.backward()
            if clip:      # trainer.py:925-926: model.parameters() = what nn.Module registered (no dict-held MLP)
                norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=clip)))
            opt_f.step()
        else:
            opt_d.zero_grad()
            loss = model.calculate_dis_loss(inter, sl)
            loss.backward()
            if clip:
                norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=clip)))
            opt_d.step()
        losses.append(float(loss.item()))
        for k, v in (("user_id", u), ("item_id", pi), ("neg_item_id", ni)):
            cols[k].append(v)
    dump_state(model, mode, "final", out)
    for k in cols:
        out[k] = np.stack(cols[k]).astype(np.int64)
    out["loss"] = np.array(losses)
    if clip:
        out["clip_max_norm"] = np.array(float(clip))
        out["grad_norm"] = np.array(norms)
    for sst in attrs:
        for t in range(T):
            for l, m in enumerate(masks[sst][t]):
                out[f"mask.{sst}.{t}.{l}"] = m
    # predict on the last batch (sigmoid score, pfcn_biasedmf.py:168-178); discriminator dropout plays no role here
    with torch.no_grad():
        out["predict_last"] = model.predict(inter, list(attrs) if mode != "none" else None).numpy().copy()
    path = os.path.join(HERE, f"pfcn_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: loss {losses[0]:.5f} -> {losses[-1]:.5f}  {os.path.getsize(path) / 1024:.1f} KiB")
    if not clip:
        _run_f64(name, out, cls, cfg, feats, mode, attrs, phases, sst_lists, dis_sizes, p_drop, lr, wd, n_users, n_items)


def _optimizers(cls, model, mode, lr, wd):
    if mode == "none":
        return torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd), None                 # trainer.py:139
    if cls is PFCN_MLP:                                                                   # trainer.py:1193-1198
        groups = [{"params": model.user_embedding.weight}, {"params": model.item_embedding.weight}]
    else:
        groups = [{"params": model.user_embedding_layer.weight}, {"params": model.item_embedding_layer.weight}]
    groups += [{"params": m.parameters()} for m in model.filter_layer.values()]
    if cls is PFCN_BiasedMF:                                                              # trainer.py:1205-1211
        groups += [{"params": model.user_bias.weight}, {"params": model.item_bias.weight}, {"params": model.global_bias}]
    if cls is PFCN_MLP:
        groups += [{"params": model.mlp_layer.parameters()}]
    if cls is PFCN_DMF:                                                                   # trainer.py:1219-1224
        groups += [{"params": model.user_mlp.parameters()}, {"params": model.item_mlp.parameters()}]
    return (torch.optim.Adam(groups, lr=lr, weight_decay=wd),
            torch.optim.Adam([{"params": m.parameters()} for m in model.dis_layer_dict.values()], lr=lr, weight_decay=wd))


def _run_f64(name, z, cls, cfg, feats, mode, attrs, phases, sst_lists, dis_sizes, p_drop, lr, wd, n_users, n_items):
    """<case>_f64.npz: the REFERENCE's model again in float64 from the recorded initial state, batches and dropout masks
    (see _refshim.float64_reference) -- the far end of the band the parity tests accept (tests/test_pfcn_hip.py): an
    element may lie between the reference's fp32 execution (the golden) and this float64 execution of the same steps."""
    with _refshim.float64_reference():
        model = cls(cfg, _FakeDataset(n_users, n_items, feats))
        model.load_state_dict({k[11:]: torch.from_numpy(v).double() for k, v in z.items() if k.startswith("init.model.")})
        queues = {}
        if mode != "none":
            for idx, mlp in model.filter_layer.items():
                mlp.load_state_dict({k[len(f"init.filter.{idx}."):]: torch.from_numpy(v).double() for k, v in z.items()
                                     if k.startswith(f"init.filter.{idx}.")})
            for sst, mlp in model.dis_layer_dict.items():
                mlp.load_state_dict({k[len(f"init.dis.{sst}."):]: torch.from_numpy(v).double() for k, v in z.items()
                                     if k.startswith(f"init.dis.{sst}.")})
                queues[sst] = []
                patch_dropout(mlp, p_drop, queues[sst])
        opt_f, opt_d = _optimizers(cls, model, mode, lr, wd)
        losses = []
        for t in range(len(phases)):
            u = z["user_id"][t]
            inter = Interaction({"user_id": torch.from_numpy(u), "item_id": torch.from_numpy(z["item_id"][t]),
                                 "neg_item_id": torch.from_numpy(z["neg_item_id"][t]),
                                 "gender": torch.from_numpy(feats["gender"][u]).double(), "age": torch.from_numpy(feats["age"][u])})
            sl = list(sst_lists[t]) if mode != "none" else None
            if mode != "none":
                for sst in sl:
                    for l in range(len(dis_sizes)):
                        queues[sst].append(torch.from_numpy(z[f"mask.{sst}.{t}.{l}"].astype(np.float64)))
            opt = opt_f if phases[t] == "F" else opt_d
            opt.zero_grad()
            loss = model.calculate_loss(inter, sl) if phases[t] == "F" else model.calculate_dis_loss(inter, sl)
            assert loss.dtype == torch.float64
            loss.backward()
            opt.step()
            losses.append(float(loss.item()))
        keep = {}
        dump_state(model, mode, "final", keep)
        keep = {k: v.astype(np.float32) for k, v in keep.items()}
        keep["loss"] = np.array(losses)
    path = os.path.join(HERE, f"pfcn_{name}_f64.npz")
    np.savez_compressed(path, **keep)
    worst = max(float(np.abs(keep[k].astype(np.float64) - z[k]).max()) for k in keep if k != "loss")
    print(f"{path}: {len(keep)} arrays, max |reference float64 - reference float32| = {worst:.2e}")


def main():
    g = ("gender",)
    ga = ("gender", "age")
    run_case("pmf_none", PFCN_PMF, "none", g, "FFFFFF", [g] * 6, D=8, B=32, dis_hidden=(16, 8), seed=1)
    run_case("pmf_sm", PFCN_PMF, "sm", g, "FFDDFD", [g] * 6, D=8, B=32, dis_hidden=(16, 8), seed=2)
    run_case("pmf_cm2", PFCN_PMF, "cm", ga, "FDFDFD", [ga, ga, g, g, ("age",), ("age",)], D=8, B=32, dis_hidden=(16, 8), seed=3)
    run_case("pmf_sm2", PFCN_PMF, "sm", ga, "FDFD", [ga, ga, ("age",), ("age",)], D=8, B=32, dis_hidden=(16, 8), seed=4)
    run_case("bmf_none", PFCN_BiasedMF, "none", g, "FFFFFF", [g] * 6, D=8, B=32, dis_hidden=(16, 8), seed=5)
    run_case("bmf_sm", PFCN_BiasedMF, "sm", g, "FFDDFD", [g] * 6, D=8, B=32, dis_hidden=(16, 8), seed=6)
    run_case("mlp_none", PFCN_MLP, "none", g, "FFFF", [g] * 4, D=8, B=32, dis_hidden=(16, 8), seed=8)
    run_case("mlp_sm", PFCN_MLP, "sm", g, "FFDDFD", [g] * 6, D=8, B=32, dis_hidden=(16, 8), seed=9)
    run_case("dmf_none", PFCN_DMF, "none", g, "FFFF", [g] * 4, D=8, B=32, dis_hidden=(16, 8), seed=10, wd=1e-3)
    run_case("dmf_cm2", PFCN_DMF, "cm", ga, "FDFDFD", [ga, ga, g, g, ("age",), ("age",)], D=8, B=32, dis_hidden=(16, 8), seed=11, wd=1e-3)
    run_case("bmf_sm_d64", PFCN_BiasedMF, "sm", g, "FDF", [g] * 3, D=64, B=96, dis_hidden=(128, 256, 128, 128, 64, 32), seed=7)
    # BASELINE.json configs[2]'s width: embedding_size 128 with the full-size discriminator, sm and cm
    run_case("bmf_sm_d128", PFCN_BiasedMF, "sm", g, "FDFD", [g] * 4, D=128, B=200, dis_hidden=(128, 256, 128, 128, 64, 32), seed=12)
    # config clip_grad_norm = {max_norm: ...} (trainer.py:925-926), small enough to bite on every filter step
    run_case("bmf_sm_clip", PFCN_BiasedMF, "sm", g, "FFDDFD", [g] * 6, D=8, B=32, dis_hidden=(16, 8), seed=14, clip=0.05)
    run_case("mlp_sm_clip", PFCN_MLP, "sm", g, "FDFD", [g] * 4, D=8, B=32, dis_hidden=(16, 8), seed=15, clip=0.05)
    run_case("bmf_cm_d128", PFCN_BiasedMF, "cm", ga, "FDFD", [ga, ga, g, g], D=128, B=200, dis_hidden=(128, 256, 128, 128, 64, 32), seed=13)


if __name__ == "__main__":
    main()
