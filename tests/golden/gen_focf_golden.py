#!/usr/bin/env python3
"""Generate golden vectors for the FOCF hot path by RUNNING THE REFERENCE (build container only).

What is recorded (fixture kinds 1 and 2), per case:
  inputs : initial user/item tables, T batches (user_id, item_id, rating, sst) and the hyper-parameters
  outputs: the reference's loss per step (FOCF.calculate_loss, focf.py:152-169), the dense gradients of
           step 1, and parameters + Adam moments after selected steps of the reference loop
           `zero_grad -> calculate_loss -> backward -> optimizer.step` (trainer.py:183-196) with
           `optim.Adam(params, lr, weight_decay)` exactly as Trainer._build_optimizer builds it
           (trainer.py:139).

The output .npz files are data (inputs + expected outputs); no reference source text is stored.
Run:  python tests/golden/gen_focf_golden.py     (needs /root/reference; not available on the GPU box)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()

import torch  # noqa: E402
from recbole.data.interaction import Interaction  # noqa: E402
from recbole.model.fair_recommender.focf import FOCF  # noqa: E402


class _Cfg(dict):
    """config[k] -> None for missing keys, like recbole Config.__getitem__ (configurator.py:405-409)."""

    def __getitem__(self, k):
        return self.get(k, None)


class _FakeDataset:
    def __init__(self, n_users, n_items, max_rating):
        self._n = {"user_id": n_users, "item_id": n_items}
        self.inter_feat = {"rating": torch.tensor([1.0, float(max_rating)])}

    def num(self, field):
        return self._n[field]


def make_batches(rng, T, B, n_users, n_items, mode):
    """Synthetic (user,item,rating,sst) batches. Row 0 is [PAD] and is only used by the 'pad' mode."""
    gender = rng.integers(0, 2, size=n_users).astype(np.float32)
    us, its, rs, ss = [], [], [], []
    for t in range(T):
        lo = 0 if mode == "pad" else 1
        u = rng.integers(lo, n_users, size=B)
        if mode == "grouped":  # few distinct items, many duplicates (FOCFDataLoader-like batches)
            pool = rng.integers(1, n_items, size=max(2, B // 16))
            i = pool[rng.integers(0, len(pool), size=B)]
        else:
            i = rng.integers(lo, n_items, size=B)
        r = rng.integers(1, 6, size=B).astype(np.float32)
        s = gender[u].copy()
        if mode == "single_group":
            s[:] = 1.0
        if mode == "nonparity" and len(np.unique(s)) < 2:
            s[0] = 1.0 - s[0]
        us.append(u); its.append(i); rs.append(r); ss.append(s)
    return (np.stack(us).astype(np.int64), np.stack(its).astype(np.int64),
            np.stack(rs), np.stack(ss))


def run_case(name, objective, n_users, n_items, D, B, T, snaps, lr, wd, fair_weight, seed, mode="uniform", clip=None):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    cfg = _Cfg(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", device=torch.device("cpu"),
               embedding_size=D, RATING_FIELD="rating", sst_attr_list=["gender"], fair_weight=fair_weight,
               fair_objective=objective)
    model = FOCF(cfg, _FakeDataset(n_users, n_items, 5.0))
    # xavier-normal on a [N, D] table gives ~1e-1 entries at these sizes; keep the reference's init
    out = {"U0": model.user_embedding_layer.weight.detach().numpy().copy(),
           "I0": model.item_embedding_layer.weight.detach().numpy().copy()}
    u, i, r, s = make_batches(rng, T, B, n_users, n_items, mode if objective != "nonparity" else "nonparity")
    out.update(user_id=u, item_id=i, rating=r, sst=s)
    out["hyper"] = np.array([lr, wd, fair_weight, 0.9, 0.999, 1e-8, 5.0], dtype=np.float64)
    out["objective"] = np.array(objective)
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)  # trainer.py:139
    losses, norms = [], []
    for t in range(T):
        inter = Interaction({"user_id": torch.from_numpy(u[t]), "item_id": torch.from_numpy(i[t]),
                             "rating": torch.from_numpy(r[t]), "gender": torch.from_numpy(s[t])})
        opt.zero_grad()
        loss = model.calculate_loss(inter)
        losses.append(float(loss.item()))
        loss.backward()
        if t == 0:
            out["pred_step1"] = model.forward(inter["user_id"], inter["item_id"])[0].detach().numpy().copy()
            out["gradU_step1"] = model.user_embedding_layer.weight.grad.numpy().copy()
            out["gradI_step1"] = model.item_embedding_layer.weight.grad.numpy().copy()
        if clip:      # config `clip_grad_norm`, trainer.py:194-195
            from torch.nn.utils.clip_grad import clip_grad_norm_
            norms.append(float(clip_grad_norm_(model.parameters(), **clip)))
        opt.step()
        if (t + 1) in snaps:
            for tag, p in (("U", model.user_embedding_layer.weight), ("I", model.item_embedding_layer.weight)):
                st = opt.state[p]
                out[f"{tag}_after{t + 1}"] = p.detach().numpy().copy()
                out[f"m{tag}_after{t + 1}"] = st["exp_avg"].numpy().copy()
                out[f"v{tag}_after{t + 1}"] = st["exp_avg_sq"].numpy().copy()
    out["loss"] = np.array(losses, dtype=np.float64)
    if clip:
        out["clip_max_norm"] = np.array(float(clip["max_norm"]))
        out["grad_norm"] = np.array(norms, dtype=np.float64)      # what clip_grad_norm_ returned: the norm BEFORE clipping
    out["snaps"] = np.array(sorted(snaps), dtype=np.int64)
    # predict() on the last batch with the final weights (focf.py:145-150) pins row a6
    with torch.no_grad():
        out["predict_last"] = model.predict(inter).numpy().copy()
    path = os.path.join(HERE, f"focf_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: loss[0]={losses[0]:.6f} loss[-1]={losses[-1]:.6f}  {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    common = dict(n_users=50, n_items=40, lr=1e-3, wd=1e-3, fair_weight=1.0)
    for k, obj in enumerate(["none", "value", "absolute", "under", "over", "nonparity"]):
        run_case(obj, obj, D=8, B=64, T=12, snaps=(1, 2, 12), seed=100 + k, **common)
    # D = 64 / 128 / 256 and B not a multiple of 64
    run_case("value_d64", "value", D=64, B=100, T=12, snaps=(1, 12), seed=200, **common)
    run_case("value_d128", "value", D=128, B=37, T=4, snaps=(4,), seed=201, **common)
    run_case("value_d256", "absolute", D=256, B=64, T=3, snaps=(3,), seed=202, **common)
    # edge cases: duplicates-heavy item-grouped batches, a single-group batch, [PAD] row 0 in the batch
    run_case("value_grouped", "value", D=8, B=128, T=6, snaps=(1, 6), seed=300, mode="grouped", **common)
    run_case("value_single_group", "value", D=8, B=64, T=3, snaps=(3,), seed=301, mode="single_group", **common)
    run_case("value_pad", "value", D=8, B=64, T=3, snaps=(3,), seed=302, mode="pad", **common)
    # long horizon: many steps between touches of a row exercises the lazy Adam catch-up (SURVEY §7 hard part 1)
    run_case("value_long", "value", n_users=400, n_items=300, D=8, B=16, T=120, snaps=(1, 60, 120), lr=1e-3,
             wd=1e-3, fair_weight=0.5, seed=400)
    run_case("none_wd0", "none", n_users=60, n_items=50, D=8, B=16, T=40, snaps=(40,), lr=1e-3, wd=0.0,
             fair_weight=0.0, seed=401)
    # config clip_grad_norm (trainer.py:194-195): always active (0.05), and active on some steps only (D = 64, grouped)
    run_case("value_clip", "value", D=8, B=64, T=12, snaps=(1, 12), seed=500, clip=dict(max_norm=0.05, norm_type=2),
             **common)
    run_case("grouped_clip", "absolute", D=64, B=100, T=8, snaps=(1, 8), seed=501, mode="grouped",
             clip=dict(max_norm=1.2, norm_type=2), **common)


if __name__ == "__main__":
    main()
