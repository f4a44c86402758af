#!/usr/bin/env python3
"""Golden vectors for the NFCF hot path by RUNNING THE REFERENCE (build container only).

Per case: initial state_dict, T batches (user, item, label, sst), the dropout masks actually applied (the
reference draws them from torch's global RNG inside nn.Dropout; here each nn.Dropout of the reference's
MLPLayers is swapped for a module that applies a RECORDED Bernoulli mask with the same 1/(1-p) scaling, so that
inputs and outputs are both known), loss per step, and parameters + Adam moments after selected steps of
`zero_grad -> calculate_loss -> backward -> optim.Adam(model.parameters(), lr, weight_decay).step()`
(trainer.py:139,183-196).  Finetune cases go through NFCF.reset_params (nfcf.py:49-67) with a pretrain
checkpoint written by the pretrain model.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from recbole.data.interaction import Interaction  # noqa: E402
from recbole.model.fair_recommender.nfcf import NFCF  # noqa: E402


class _Cfg(dict):
    def __getitem__(self, k):
        return self.get(k, None)


class _FakeDataset:
    def __init__(self, n_users, n_items, gender):
        self._n = {"user_id": n_users, "item_id": n_items}
        self._uf = Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(gender)})

    def num(self, field):
        return self._n[field]

    def get_user_feature(self):
        return self._uf


class RecordedDropout(nn.Module):
    """Applies mask/(1-p) from a queue filled by the generator (same arithmetic as F.dropout in training)."""

    def __init__(self, p, queue):
        super().__init__()
        self.p, self.queue = p, queue

    def forward(self, x):
        if self.p == 0.0:
            return x
        m = self.queue.pop(0)
        assert m.shape == x.shape
        return x * (m / (1.0 - self.p))


def patch_dropout(model, p, queue):
    seq = model.mlp_layers.mlp_layers
    for k, mod in enumerate(seq):
        if isinstance(mod, (nn.Dropout, RecordedDropout)):
            seq[k] = RecordedDropout(p, queue)


ONLY = [s for s in os.environ.get("GOLDEN_ONLY", "").split(",") if s]


def run_case(name, *args, **kwargs):
    if ONLY and name not in ONLY:
        return
    return _run_case(name, *args, **kwargs)


def _run_case(name, stage, n_users, n_items, D, hidden, B, T, snaps, lr, wd, p_drop, fair_weight, seed, clip=None):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    gender = rng.integers(0, 2, size=n_users).astype(np.float32)
    base = dict(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device=torch.device("cpu"),
                LABEL_FIELD="label", embedding_size=D, mlp_hidden_size=list(hidden), dropout=p_drop,
                sst_attr_list=["gender"], fair_weight=fair_weight)
    ds = _FakeDataset(n_users, n_items, gender)
    out = {"hidden": np.array(hidden), "hyper": np.array([lr, wd, fair_weight, p_drop], dtype=np.float64),
           "stage": np.array(stage), "gender": gender}
    model = NFCF(_Cfg(base, load_pretrain_path=None), ds)
    if stage == "finetune":
        # a short pretrain so that the checkpoint is not just noise, then the reference's own reset_params
        opt0 = torch.optim.Adam(model.parameters(), lr=1e-2)
        q0 = []
        patch_dropout(model, 0.0, q0)
        for _ in range(5):
            u = torch.from_numpy(rng.integers(1, n_users, size=B))
            i = torch.from_numpy(rng.integers(1, n_items, size=B))
            lab = torch.from_numpy((rng.random(B) < 0.5).astype(np.float32))
            opt0.zero_grad()
            model.calculate_loss(Interaction({"user_id": u, "item_id": i, "label": lab})).backward()
            opt0.step()
        out["pretrain_user_embedding"] = model.user_embedding.weight.detach().numpy().copy()
        tmp = tempfile.NamedTemporaryFile(suffix=".pth", delete=False)
        tmp.close()
        torch.save({"state_dict": model.state_dict()}, tmp.name)
        model = NFCF(_Cfg(base, load_pretrain_path=tmp.name), ds)   # -> reset_params (nfcf.py:49-67)
        os.unlink(tmp.name)
    queue = []
    patch_dropout(model, p_drop, queue)
    for k, v in model.state_dict().items():
        out["init." + k] = v.detach().numpy().copy()
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)   # trainer.py:139
    us, its, labs, ss, masks, losses = [], [], [], [], [], []
    norms = []
    sizes = [2 * D] + list(hidden)
    for t in range(T):
        u = rng.integers(1, n_users, size=B)
        pool = rng.integers(1, n_items, size=max(2, B // 4))          # duplicates among items (DF statistics)
        i = pool[rng.integers(0, len(pool), size=B)]
        lab = (rng.random(B) < 0.6).astype(np.float32)
        s = gender[u]
        step_masks = []
        if p_drop > 0:
            for w in sizes:
                m = (torch.rand(B, w) >= p_drop).float()
                queue.append(m)
                step_masks.append(m.numpy())
        inter = Interaction({"user_id": torch.from_numpy(u), "item_id": torch.from_numpy(i),
                             "label": torch.from_numpy(lab), "gender": torch.from_numpy(s)})
        opt.zero_grad()
        loss = model.calculate_loss(inter)
        losses.append(float(loss.item()))
        loss.backward()
        if t == 0:
            with torch.no_grad():
                out["grad_step1.mlp0"] = model.mlp_layers.mlp_layers[1].weight.grad.numpy().copy()
                out["grad_step1.item"] = model.item_embedding.weight.grad.numpy().copy()
        if clip:      # config clip_grad_norm (trainer.py:194-195): every parameter of NFCF is in model.parameters()
            norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=clip)))
        opt.step()
        us.append(u); its.append(i); labs.append(lab); ss.append(s); masks.append(step_masks)
        if (t + 1) in snaps:
            for k, v in model.state_dict().items():
                out[f"after{t + 1}." + k] = v.detach().numpy().copy()
            for pname, prm in model.named_parameters():
                if prm in opt.state and "exp_avg" in opt.state[prm]:
                    out[f"after{t + 1}.m." + pname] = opt.state[prm]["exp_avg"].numpy().copy()
    out.update(user_id=np.stack(us).astype(np.int64), item_id=np.stack(its).astype(np.int64), label=np.stack(labs),
               sst=np.stack(ss), loss=np.array(losses), snaps=np.array(sorted(snaps)))
    if clip:
        out["clip_max_norm"] = np.array(float(clip))
        out["grad_norm"] = np.array(norms)
    if p_drop > 0:
        for li in range(len(sizes)):
            out[f"mask{li}"] = np.stack([masks[t][li] for t in range(T)]).astype(np.uint8)
    # predict() on the last batch in eval mode (dropout off), nfcf.py:112-115
    model.eval()
    patch_dropout(model, 0.0, [])
    with torch.no_grad():
        out["predict_last"] = model.predict(inter).numpy().copy()
    path = os.path.join(HERE, f"nfcf_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: loss[0]={losses[0]:.6f} loss[-1]={losses[-1]:.6f} {os.path.getsize(path) / 1024:.1f} KiB")
    _run_f64(name, out, base, ds, stage, sizes, T, snaps, lr, wd, p_drop, clip)


def _run_f64(name, z, base, ds, stage, sizes, T, snaps, lr, wd, p_drop, clip):
    """<case>_f64.npz: the REFERENCE's NFCF run again in float64 from the recorded initial state, batches and dropout masks
    (see _refshim.float64_reference).  The parity tests accept a parameter element that lies between the reference's fp32
    execution (the golden) and this one: Adam divides a gradient by its own magnitude, so where a gradient nearly cancels
    the fp32 run's rounding noise moves the element by a visible fraction of lr (tests/golden/noise_floor.py)."""
    with _refshim.float64_reference():
        model = NFCF(_Cfg(base, load_pretrain_path=None), ds)
        if stage == "finetune":       # the state reset_params left (z["init.*"]): frozen user table, fairness term on
            model.load_pretrain_path = "recorded"
            model.user_embedding.weight.requires_grad = False
        model.load_state_dict({k[5:]: torch.from_numpy(v).double() for k, v in z.items() if k.startswith("init.")})
        queue = []
        patch_dropout(model, p_drop, queue)
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
        keep = {}
        for t in range(T):
            if p_drop > 0:
                for li in range(len(sizes)):
                    queue.append(torch.from_numpy(z[f"mask{li}"][t].astype(np.float64)))
            inter = Interaction({"user_id": torch.from_numpy(z["user_id"][t]), "item_id": torch.from_numpy(z["item_id"][t]),
                                 "label": torch.from_numpy(z["label"][t]).double(), "gender": torch.from_numpy(z["sst"][t]).double()})
            opt.zero_grad()
            loss = model.calculate_loss(inter)
            assert loss.dtype == torch.float64
            loss.backward()
            if clip:
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=clip)
            opt.step()
            if (t + 1) in snaps:
                for k, v in model.state_dict().items():
                    keep[f"after{t + 1}." + k] = v.detach().numpy().astype(np.float32)
    path = os.path.join(HERE, f"nfcf_{name}_f64.npz")
    np.savez_compressed(path, **keep)
    worst = max(float(np.abs(keep[k].astype(np.float64) - z[k]).max()) for k in keep)
    print(f"{path}: {len(keep)} arrays, max |reference float64 - reference float32| = {worst:.2e}")


def main():
    c = dict(n_users=50, n_items=40, B=64, lr=1e-3)
    run_case("pretrain", "pretrain", D=8, hidden=(16, 8), T=10, snaps=(1, 10), wd=1e-6, p_drop=0.0, fair_weight=0.1, seed=1, **c)
    run_case("pretrain_dropout", "pretrain", D=8, hidden=(16, 8), T=6, snaps=(6,), wd=1e-6, p_drop=0.2, fair_weight=0.1, seed=2, **c)
    run_case("pretrain_wd", "pretrain", D=8, hidden=(16, 8), T=30, snaps=(30,), wd=1e-3, p_drop=0.0, fair_weight=0.1, seed=3, **c)
    run_case("pretrain_d64", "pretrain", D=64, hidden=(128, 64), T=4, snaps=(4,), wd=1e-6, p_drop=0.2, fair_weight=0.1, seed=4, **c)
    run_case("finetune", "finetune", D=8, hidden=(16, 8), T=10, snaps=(1, 10), wd=1e-6, p_drop=0.0, fair_weight=0.1, seed=5, **c)
    run_case("finetune_dropout", "finetune", D=8, hidden=(16, 8), T=6, snaps=(6,), wd=1e-6, p_drop=0.2, fair_weight=0.5, seed=6, **c)
    run_case("pretrain_clip", "pretrain", D=8, hidden=(16, 8), T=6, snaps=(6,), wd=1e-6, p_drop=0.0, fair_weight=0.1, seed=10, clip=0.05, **c)
    run_case("finetune_clip", "finetune", D=8, hidden=(16, 8), T=6, snaps=(6,), wd=1e-6, p_drop=0.2, fair_weight=0.5, seed=11, clip=0.05, **c)
    run_case("finetune_d64", "finetune", D=64, hidden=(128, 64), T=4, snaps=(4,), wd=1e-6, p_drop=0.0, fair_weight=0.1, seed=7, **c)
    # BASELINE.json configs[4]'s width: embedding_size 256, mlp_hidden_size [128, 64], B = 200
    c256 = dict(n_users=120, n_items=90, B=200, lr=1e-3)
    run_case("pretrain_d256", "pretrain", D=256, hidden=(128, 64), T=4, snaps=(4,), wd=1e-6, p_drop=0.2, fair_weight=0.1, seed=8, **c256)
    run_case("finetune_d256", "finetune", D=256, hidden=(128, 64), T=4, snaps=(4,), wd=1e-6, p_drop=0.0, fair_weight=0.1, seed=9, **c256)


if __name__ == "__main__":
    main()
