"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's FairGo_PMF hot path
(FairGo_GCN's finetune stage is the same code).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.

Parity pin: golden vectors produced by running the reference itself (tests/golden/gen_fairgo_golden.py ->
tests/golden/fairgo_*.npz; test: tests/test_oracle_fairgo.py).  fp32 torch-CPU, the reference's own arithmetic.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F


def norm_rating_matrix(n_users, n_items, tu, ti, tr) -> torch.Tensor:
    """get_norm_rating_matrix, fairgo_pmf.py:102-129: L = D^-1 A, A = weighted bipartite adjacency (values = ratings),
    row sums + 1e-7."""
    N = n_users + n_items
    A = sp.coo_matrix((np.concatenate([tr, tr]), (np.concatenate([tu, ti + n_users]), np.concatenate([ti + n_users, tu]))),
                      shape=(N, N), dtype=np.float32).tocsr()
    diag = 1.0 / (np.asarray(A.sum(axis=1)).flatten() + 1e-7)
    L = sp.coo_matrix(sp.diags(diag) * A)
    return torch.sparse_coo_tensor(np.stack([L.row, L.col]), L.data.astype(np.float32), (N, N)).coalesce()


class MLP:
    """MLPLayers(layers, activation='leakyrelu') without BN / dropout: Linear -> LeakyReLU per layer, last included."""

    def __init__(self, z, prefix):
        self.W, self.b = [], []
        l = 0
        while f"{prefix}.mlp_layers.{3 * l + 1}.weight" in z:
            self.W.append(torch.tensor(z[f"{prefix}.mlp_layers.{3 * l + 1}.weight"]).requires_grad_())
            self.b.append(torch.tensor(z[f"{prefix}.mlp_layers.{3 * l + 1}.bias"]).requires_grad_())
            l += 1

    def params(self):
        return [t for pair in zip(self.W, self.b) for t in pair]

    def __call__(self, x):
        for W, b in zip(self.W, self.b):
            x = F.leaky_relu(F.linear(x, W, b), 0.01)
        return x

    def export(self, prefix, out):
        for l, (W, b) in enumerate(zip(self.W, self.b)):
            out[f"{prefix}.mlp_layers.{3 * l + 1}.weight"] = W.detach().numpy().copy()
            out[f"{prefix}.mlp_layers.{3 * l + 1}.bias"] = b.detach().numpy().copy()


class Model:
    def __init__(self, z):
        self.aggr = str(z["aggr"])
        self.attrs = [str(a) for a in z["attrs"]]
        self.lr, self.wd, self.fair_weight = (float(x) for x in z["hyper"])
        self.n_layers = int(z["n_layers"])
        t = lambda k: torch.tensor(z["init.model." + k]).requires_grad_()
        self.U, self.I = t("user_embedding_layer.weight"), t("item_embedding_layer.weight")
        self.n_users, self.n_items = self.U.shape[0], self.I.shape[0]
        self.aggr_W = [t(f"aggr_layer.{k}.weight") for k in (0, 2, 4)]
        self.aggr_b = [t(f"aggr_layer.{k}.bias") for k in (0, 2, 4)]
        self.filters = {a: MLP(z, f"init.filter.{a}") for a in self.attrs}
        self.dis = {a: MLP(z, f"init.dis.{a}") for a in self.attrs}
        self.sst_size = {"gender": 2, "age": 3}
        vs = torch.tensor(z["vs_weights"], dtype=torch.float32)
        self.vs = vs / vs.sum()
        self.L = torch.sparse_coo_tensor(np.stack([z["L_row"], z["L_col"]]), z["L_val"],
                                         (self.n_users + self.n_items,) * 2).coalesce()
        self.stage = "pretrain"

    def forward(self, sst_list=None):
        """fairgo_pmf.py:160-172: finetune = sum of the selected filters over the WHOLE table / number of ALL filters."""
        E = torch.cat([self.U, self.I], 0)
        if self.stage == "finetune":
            sl = self.attrs if sst_list is None else sst_list
            tmp = None
            for s in sl:
                tmp = self.filters[s](E) if tmp is None else tmp + self.filters[s](E)
            E = tmp / len(self.filters)
        return E[:self.n_users], E[self.n_users:]

    def aggr_layer(self, x):
        x = F.leaky_relu(F.linear(x, self.aggr_W[0], self.aggr_b[0]), 0.01)
        x = F.leaky_relu(F.linear(x, self.aggr_W[1], self.aggr_b[1]), 0.01)
        return F.linear(x, self.aggr_W[2], self.aggr_b[2])

    def dis_loss(self, user, labels, sst_list):
        """calculate_dis_loss, fairgo_pmf.py:190-238 (note: the multi-class LOCAL term applies CrossEntropy to
        sigmoid(logits), the node term to the raw logits)."""
        ua, ia = self.forward(sst_list)
        node = ua[user]
        H = torch.cat([ua, ia], 0)
        hs = []
        for _ in range(self.n_layers):
            H = torch.sparse.mm(self.L, H)
            hs.append(H)
        lva = self.aggr == "LVA" and self.n_layers > 1
        if self.n_layers == 1:
            G = hs[0]
        elif self.aggr == "WAP":
            G = torch.stack(hs, 1).mean(1)
        elif self.aggr == "LBA":
            G = self.aggr_layer(torch.cat(hs, 1))
        if not lva:
            local = G[:self.n_users][user]
        else:
            locals_ = [h[:self.n_users][user] for h in hs]
        node_l, local_l = 0.0, 0.0
        for s in sst_list:
            d, lab = self.dis[s], labels[s]
            if self.sst_size[s] == 2:
                y = lab.float().unsqueeze(1)
                node_l = node_l + F.binary_cross_entropy(torch.sigmoid(d(node)), y)
                if lva:
                    for w, x in zip(self.vs, locals_):
                        local_l = local_l + w * F.binary_cross_entropy(torch.sigmoid(d(x)), y)
                else:
                    local_l = local_l + F.binary_cross_entropy(torch.sigmoid(d(local)), y)
            else:
                node_l = node_l + F.cross_entropy(d(node), lab.long())
                if lva:
                    for w, x in zip(self.vs, locals_):
                        local_l = local_l + w * F.cross_entropy(torch.sigmoid(d(x)), lab.long())
                else:
                    local_l = local_l + F.cross_entropy(torch.sigmoid(d(local)), lab.long())
        return node_l + local_l

    def loss(self, user, item, rating, labels, sst_list):
        ua, ia = self.forward(sst_list)
        mse = F.mse_loss((ua[user] * ia[item]).sum(-1), rating)
        if self.stage == "finetune":
            return mse - self.fair_weight * self.dis_loss(user, labels, sst_list)
        return mse

    def predict(self, user, item, max_rating=5.0):
        ua, ia = self.forward()
        return torch.clamp((ua[user] * ia[item]).sum(1), min=0.0, max=max_rating) / max_rating


def train(z) -> Dict[str, np.ndarray]:
    m = Model(z)
    opt_p = torch.optim.Adam([m.U, m.I], lr=m.lr, weight_decay=m.wd)
    dis_params = [p for a in m.attrs for p in m.dis[a].params()]
    if m.aggr == "LBA":
        dis_params += [t for pair in zip(m.aggr_W, m.aggr_b) for t in pair]
    opt_d = torch.optim.Adam(dis_params, lr=m.lr, weight_decay=m.wd)
    opt_f = torch.optim.Adam([p for a in m.attrs for p in m.filters[a].params()], lr=m.lr, weight_decay=m.wd)
    losses = []
    for t, ph in enumerate(str(x) for x in z["phases"]):
        u, i, r = torch.tensor(z["user_id"][t]), torch.tensor(z["item_id"][t]), torch.tensor(z["rating"][t])
        labels = {"gender": torch.tensor(z["gender"][z["user_id"][t]]), "age": torch.tensor(z["age"][z["user_id"][t]])}
        sl = [s for s in str(z["sst_lists"][t]).split(",") if s]
        if ph == "P":
            m.stage = "pretrain"
            opt, l = opt_p, None
            opt.zero_grad()
            l = m.loss(u, i, r, labels, None)
        else:
            m.stage = "finetune"
            opt = opt_f if ph == "F" else opt_d
            opt.zero_grad()
            l = m.loss(u, i, r, labels, sl) if ph == "F" else m.dis_loss(u, labels, sl)
        losses.append(float(l.item()))
        l.backward()
        opt.step()
    out = {"loss": np.array(losses), "final.model.user_embedding_layer.weight": m.U.detach().numpy().copy(),
           "final.model.item_embedding_layer.weight": m.I.detach().numpy().copy()}
    for k, (W, b) in zip((0, 2, 4), zip(m.aggr_W, m.aggr_b)):
        out[f"final.model.aggr_layer.{k}.weight"] = W.detach().numpy().copy()
        out[f"final.model.aggr_layer.{k}.bias"] = b.detach().numpy().copy()
    for a in m.attrs:
        m.filters[a].export(f"final.filter.{a}", out)
        m.dis[a].export(f"final.dis.{a}", out)
    with torch.no_grad():
        out["predict_last"] = m.predict(u, i).numpy().copy()
    return out


# ---- FairGo_GCN pretrain stage: PARITY UNPINNED -----------------------------------------------------------------------
# The reference calls torch_geometric.nn.GCN (fairgo_gcn.py:20, :52-57, :176), which is neither pinned by the reference
# nor installed in this image, so nothing below could be checked against the real thing.  It restates PyG's published
# GCNConv / BasicGNN semantics (default arguments) in dense torch, as the checker of the HIP path for that stage.
def gcn_a_hat(n_users, n_items, tu, ti, tr) -> torch.Tensor:
    """Dense Ahat = Dhat^-1/2 (A + I) Dhat^-1/2 for the reference's edge list (both directions of every rating)."""
    N = n_users + n_items
    A = torch.zeros((N, N), dtype=torch.float64)
    tu_t, ti_t, tr_t = torch.as_tensor(tu), torch.as_tensor(ti) + n_users, torch.as_tensor(tr, dtype=torch.float64)
    A.index_put_((tu_t, ti_t), tr_t, accumulate=True)
    A.index_put_((ti_t, tu_t), tr_t, accumulate=True)
    A += torch.eye(N, dtype=torch.float64)
    deg = A.sum(dim=1)
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    return (dis[:, None] * A * dis[None, :]).to(torch.float32)


def gcn_forward(x, a_hat, weights, biases, act=torch.relu, dropout_masks=None, p=0.0):
    """BasicGNN forward: conv -> act -> dropout between layers, nothing after the last; conv = Ahat (X W^T) + b."""
    for k, (W, b) in enumerate(zip(weights, biases)):
        x = a_hat @ (x @ W.t()) + b
        if k == len(weights) - 1:
            break
        x = act(x)
        if dropout_masks is not None:
            x = x * dropout_masks[k] / (1.0 - p)
    return x


def gcn_pretrain_steps(U0, I0, weights, biases, a_hat, users, items, ratings, lr, wd):
    """T steps of MSE(sum(E_u[u] * E_i[i]), rating) with E = GCN(cat(U, I)) and torch.optim.Adam over U, I and the GCN
    parameters (optimizer_pretrain, trainer.py:850-855); returns the losses and the final parameters."""
    U, I = torch.nn.Parameter(U0.clone()), torch.nn.Parameter(I0.clone())
    Ws = [torch.nn.Parameter(w.clone()) for w in weights]
    bs = [torch.nn.Parameter(b.clone()) for b in biases]
    opt = torch.optim.Adam([U, I] + Ws + bs, lr=lr, weight_decay=wd)
    n_users = U0.shape[0]
    losses = []
    for u, i, r in zip(users, items, ratings):
        opt.zero_grad()
        E = gcn_forward(torch.cat([U, I], 0), a_hat, Ws, bs)
        loss = torch.nn.functional.mse_loss((E[u] * E[i + n_users]).sum(-1), r)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses, U.data, I.data, [w.data for w in Ws], [b.data for b in bs]
