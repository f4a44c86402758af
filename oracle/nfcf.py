"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's NFCF hot path
(SURVEY.md §8 a17-a19).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Parity pin: golden vectors produced by running the reference itself (tests/golden/gen_nfcf_golden.py ->
tests/golden/nfcf_*.npz; test: tests/test_oracle_nfcf.py).  fp32 torch-CPU, the reference's own arithmetic.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def mlp_forward(x, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], p_drop: float = 0.0,
                masks: Optional[Sequence[torch.Tensor]] = None):
    """MLPLayers([2D, h..., 1], dropout) of layers.py:56-85 with activation 'relu' and no BN:
    per layer Dropout -> Linear -> ReLU, INCLUDING the last (1-wide) layer."""
    for l, (W, b) in enumerate(zip(weights, biases)):
        if p_drop > 0.0 and masks is not None:
            x = x * (masks[l] / (1.0 - p_drop))
        x = torch.relu(F.linear(x, W, b))
    return x


def forward(U, I, weights, biases, user, item, p_drop=0.0, masks=None):
    """NFCF.forward, nfcf.py:69-74: sigmoid(MLP(cat(U[u], I[i])))."""
    x = torch.cat((U[user], I[item]), -1)
    return torch.sigmoid(mlp_forward(x, weights, biases, p_drop, masks).squeeze(-1))


def differential_fairness(score, label, sst, item):
    """NFCF.get_differential_fairness, nfcf.py:76-97, on the label == 1 rows:
    M[k,g] = (sum score + 1/K) / (count + 1); eps_k = max_{g<g'} |log M[k,g] - log M[k,g']|; mean over items."""
    pos = label == 1
    score, sst, item = score[pos], sst[pos], item[pos]
    sv, g = torch.unique(sst, return_inverse=True)
    iv, k = torch.unique(item, return_inverse=True)
    K, G = len(iv), len(sv)
    sm = torch.zeros((K, G), dtype=score.dtype)
    nm = torch.zeros((K, G), dtype=score.dtype)
    sm.index_put_((k, g), score, accumulate=True)
    nm.index_put_((k, g), torch.ones_like(score), accumulate=True)
    sm = (sm + 1.0 / K) / (nm + 1.0)
    eps = torch.zeros(K, dtype=score.dtype)
    for a in range(G):
        for b in range(a + 1, G):
            e = (torch.log(sm[:, a]) - torch.log(sm[:, b])).abs()
            eps = torch.where(e > eps, e, eps)
    return eps.mean()


def loss(stage, fair_weight, U, I, weights, biases, user, item, label, sst, p_drop=0.0, masks=None):
    """NFCF.calculate_loss, nfcf.py:99-110: BCE, plus fair_weight * differential fairness when fine-tuning."""
    out = forward(U, I, weights, biases, user, item, p_drop, masks)
    l = F.binary_cross_entropy(out, label)
    if stage == "finetune":
        l = l + fair_weight * differential_fairness(out, label, sst, item)
    return l, out


def reset_user_embedding(user_emb: torch.Tensor, sst: torch.Tensor) -> torch.Tensor:
    """The de-biasing projection of NFCF.reset_params, nfcf.py:53-65 (rows 1.. only; row 0 is [PAD]):
    v_B = (mean_g0 - mean_g1)/||.||;  U[1:] -= (U[1:] . v_B) v_B."""
    vals = torch.unique(sst)
    e = user_emb[1:].clone()
    m1 = e[sst == vals[0]].mean(dim=0)
    m2 = e[sst == vals[1]].mean(dim=0)
    vb = (m1 - m2) / torch.linalg.norm(m1 - m2, keepdim=True)
    e = e - torch.mul(e, vb).sum(dim=1, keepdim=True) * vb
    out = user_emb.clone()
    out[1:] = e
    return out


def train(z: Dict[str, np.ndarray], snaps=()) -> Dict[str, np.ndarray]:
    """The reference step loop on the batches (and dropout masks) recorded in a golden file `z`."""
    stage = str(z["stage"])
    lr, wd, fw, p = (float(x) for x in z["hyper"])
    U = torch.tensor(z["init.user_embedding.weight"], requires_grad=(stage == "pretrain"))
    I = torch.tensor(z["init.item_embedding.weight"], requires_grad=True)
    n_layers = len(z["hidden"]) + 1
    Ws = [torch.tensor(z[f"init.mlp_layers.mlp_layers.{3 * l + 1}.weight"], requires_grad=True) for l in range(n_layers)]
    bs = [torch.tensor(z[f"init.mlp_layers.mlp_layers.{3 * l + 1}.bias"], requires_grad=True) for l in range(n_layers)]
    params = ([U] if stage == "pretrain" else []) + [I] + [t for pair in zip(Ws, bs) for t in pair]
    opt = torch.optim.Adam(params, lr=lr, weight_decay=wd)
    out: Dict[str, np.ndarray] = {}
    losses, norms = [], []
    for t in range(len(z["user_id"])):
        u, i = torch.tensor(z["user_id"][t]), torch.tensor(z["item_id"][t])
        lab, s = torch.tensor(z["label"][t]), torch.tensor(z["sst"][t])
        masks = [torch.tensor(z[f"mask{l}"][t]).float() for l in range(n_layers)] if p > 0 else None
        opt.zero_grad()
        l, _ = loss(stage, fw, U, I, Ws, bs, u, i, lab, s, p, masks)
        losses.append(float(l.item()))
        l.backward()
        if t == 0:
            out["grad_step1.mlp0"] = Ws[0].grad.numpy().copy()
            out["grad_step1.item"] = I.grad.numpy().copy()
        if "clip_max_norm" in z:      # trainer.py:194-195
            norms.append(float(torch.nn.utils.clip_grad_norm_(params, max_norm=float(z["clip_max_norm"]))))
        opt.step()
        if (t + 1) in snaps:
            out[f"after{t + 1}.user_embedding.weight"] = U.detach().numpy().copy()
            out[f"after{t + 1}.item_embedding.weight"] = I.detach().numpy().copy()
            for l in range(n_layers):
                out[f"after{t + 1}.mlp_layers.mlp_layers.{3 * l + 1}.weight"] = Ws[l].detach().numpy().copy()
                out[f"after{t + 1}.mlp_layers.mlp_layers.{3 * l + 1}.bias"] = bs[l].detach().numpy().copy()
    out["loss"] = np.array(losses)
    if norms:
        out["grad_norm"] = np.array(norms)
    return out
