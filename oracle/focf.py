"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's FOCF hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
The product path (recbole-fairrec_amd/) never does: it fails loudly when the HIP library is missing.

Parity pin: checked against golden vectors produced by running the reference itself in the build
container (tests/golden/gen_focf_golden.py -> tests/golden/focf_*.npz; test: tests/test_oracle_focf.py).
The reference ships no tests / known-answer vectors of its own, so those generated
fixtures are the pin.

Arithmetic is fp32 torch-CPU, the same third-party arithmetic the reference runs on
(torch 2.10 CPU kernels): each function cites the reference lines it restates.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

OBJECTIVES = ("none", "value", "absolute", "under", "over", "nonparity")


def forward(U: torch.Tensor, I: torch.Tensor, user: torch.Tensor, item: torch.Tensor):
    """FOCF.forward, focf.py:136-143: gather both rows, row-wise dot product."""
    ue = U[user]
    ie = I[item]
    return (ue * ie).sum(dim=-1), ue, ie


def item_group_means(pred, rating, sst, item):
    """FOCF.get_item_ratings, focf.py:75-91.

    Column = rank of the sst value among the values PRESENT in this batch; row = rank of the item id
    among the distinct item ids of the batch.  Returns (mean pred, mean rating) as [K,2] with the
    reference's `count + 1e-5` denominator.
    """
    _, g = torch.unique(sst, return_inverse=True)
    _, k = torch.unique(item, return_inverse=True)
    K = int(k.max().item()) + 1 if k.numel() else 0
    sp = torch.zeros((K, 2), dtype=pred.dtype, device=pred.device)
    st = torch.zeros((K, 2), dtype=pred.dtype, device=pred.device)
    cn = torch.zeros((K, 2), dtype=pred.dtype, device=pred.device)
    sp.index_put_((k, g), pred, accumulate=True)
    st.index_put_((k, g), rating, accumulate=True)
    cn.index_put_((k, g), torch.ones_like(pred), accumulate=True)
    cn = cn + 1e-5
    return sp / cn, st / cn


def fairness_term(objective: str, pred, rating, sst, item):
    """focf.py:93-134 -- value / absolute / under / over / nonparity unfairness (smooth-L1, beta=1, mean)."""
    objective = objective.strip().lower()
    if objective == "nonparity":  # focf.py:127-134
        vals = torch.unique(sst)
        a = pred[sst == vals[0]].mean()
        b = pred[sst == vals[1]].mean()
        return F.smooth_l1_loss(a, b)
    P, T = item_group_means(pred, rating, sst, item)
    if objective == "value":       # focf.py:93-99
        d = P - T
    elif objective == "absolute":  # focf.py:101-107
        d = (P - T).abs()
    elif objective == "under":     # focf.py:109-116
        d = torch.clamp(T - P, min=0.0)
    elif objective == "over":      # focf.py:118-125
        d = torch.clamp(P - T, min=0.0)
    else:
        raise ValueError(objective)
    x = (d[:, 0] - d[:, 1]).abs()
    return F.smooth_l1_loss(x, torch.zeros_like(x))


def loss(objective: str, fair_weight: float, U, I, user, item, rating, sst):
    """FOCF.calculate_loss, focf.py:152-169: MSE(pred, rating) + fair_weight * fairness."""
    pred, _, _ = forward(U, I, user, item)
    out = F.mse_loss(pred, rating)
    if objective.strip().lower() != "none":
        out = out + fair_weight * fairness_term(objective, pred, rating, sst, item)
    return out, pred


def predict(U, I, user, item, max_rating: float):
    """FOCF.predict, focf.py:145-150."""
    pred, _, _ = forward(U, I, user, item)
    return torch.clamp(pred, min=0.0, max=max_rating) / max_rating


def adam_dense_step_(p, g, m, v, step: int, lr: float, wd: float, b1=0.9, b2=0.999, eps=1e-8):
    """One dense Adam step with coupled L2, in the op order of torch/optim/adam.py `_single_tensor_adam`
    (the CPU default that `optim.Adam(params, lr, weight_decay)` of trainer.py:139 runs).  In place.
    Scalars (bias corrections, step size) are computed in Python double like torch does."""
    if wd != 0:
        g = g.add(p, alpha=wd)
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    denom = (v.sqrt() / bc2_sqrt).add_(eps)
    p.addcdiv_(m, denom, value=-step_size)


def train(objective: str, U0, I0, user, item, rating, sst, lr, wd, fair_weight, snaps=(), use_torch_adam=True,
          b1=0.9, b2=0.999, eps=1e-8, clip_max_norm=None) -> Dict[str, np.ndarray]:
    """The reference step loop (trainer.py:183-196) on T recorded batches.

    user/item/rating/sst: arrays [T, B].  Returns loss per step, dense grads of step 1 and snapshots
    of (table, exp_avg, exp_avg_sq) after the steps listed in `snaps`.
    `use_torch_adam=False` swaps torch.optim.Adam for `adam_dense_step_` (checks the restated formula)."""
    U = torch.tensor(np.asarray(U0), dtype=torch.float32, requires_grad=True)
    I = torch.tensor(np.asarray(I0), dtype=torch.float32, requires_grad=True)
    opt = torch.optim.Adam([U, I], lr=lr, weight_decay=wd, betas=(b1, b2), eps=eps) if use_torch_adam else None
    mom = {id(t): (torch.zeros_like(t), torch.zeros_like(t)) for t in (U, I)}
    out: Dict[str, np.ndarray] = {}
    losses, norms = [], []
    T = len(user)
    for t in range(T):
        u = torch.as_tensor(np.asarray(user[t]), dtype=torch.int64)
        i = torch.as_tensor(np.asarray(item[t]), dtype=torch.int64)
        r = torch.as_tensor(np.asarray(rating[t]), dtype=torch.float32)
        s = torch.as_tensor(np.asarray(sst[t]), dtype=torch.float32)
        U.grad = None
        I.grad = None
        l, pred = loss(objective, fair_weight, U, I, u, i, r, s)
        losses.append(float(l.item()))
        l.backward()
        if t == 0:
            out["pred_step1"] = pred.detach().numpy().copy()
            out["gradU_step1"] = U.grad.numpy().copy()
            out["gradI_step1"] = I.grad.numpy().copy()
        if clip_max_norm:      # config `clip_grad_norm` (trainer.py:194-195): norm over all parameters, 2-norm
            norms.append(float(torch.nn.utils.clip_grad_norm_([U, I], clip_max_norm, norm_type=2)))
        if opt is not None:
            opt.step()
        else:
            with torch.no_grad():
                for tns in (U, I):
                    adam_dense_step_(tns, tns.grad, mom[id(tns)][0], mom[id(tns)][1], t + 1, lr, wd, b1, b2, eps)
        if (t + 1) in snaps:
            for tag, tns in (("U", U), ("I", I)):
                if opt is not None:
                    mm, vv = opt.state[tns]["exp_avg"], opt.state[tns]["exp_avg_sq"]
                else:
                    mm, vv = mom[id(tns)]
                out[f"{tag}_after{t + 1}"] = tns.detach().numpy().copy()
                out[f"m{tag}_after{t + 1}"] = mm.numpy().copy()
                out[f"v{tag}_after{t + 1}"] = vv.numpy().copy()
    out["loss"] = np.array(losses, dtype=np.float64)
    out["grad_norm"] = np.array(norms, dtype=np.float64)
    return out


class CpuTrainerBaseline:
    """The reference's CPU step (dense autograd gradient + stock dense torch.optim.Adam) as a timed
    baseline for bench.py's `cpu_baseline` leg ("port" kind: the reference's own Python never leaves the
    build container).  Same arithmetic as `train` above."""

    def __init__(self, n_users, n_items, D, lr, wd, fair_weight, objective, seed=2020, threads: Optional[int] = None):
        if threads:
            torch.set_num_threads(threads)
        g = torch.Generator().manual_seed(seed)
        std_u = math.sqrt(2.0 / (n_users + D))
        std_i = math.sqrt(2.0 / (n_items + D))
        self.U = (torch.randn(n_users, D, generator=g) * std_u).requires_grad_()
        self.I = (torch.randn(n_items, D, generator=g) * std_i).requires_grad_()
        self.opt = torch.optim.Adam([self.U, self.I], lr=lr, weight_decay=wd)
        self.objective, self.fair_weight = objective, fair_weight

    def step(self, u, i, r, s) -> float:
        self.opt.zero_grad()
        l, _ = loss(self.objective, self.fair_weight, self.U, self.I, u, i, r, s)
        v = l.item()
        l.backward()
        self.opt.step()
        return v
