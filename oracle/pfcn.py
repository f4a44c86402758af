"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's PFCN_PMF / PFCN_BiasedMF hot
path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Parity pin: golden vectors produced by running the reference itself (tests/golden/gen_pfcn_golden.py ->
tests/golden/pfcn_*.npz; test: tests/test_oracle_pfcn.py).  fp32 torch-CPU, the reference's own arithmetic.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


class MLP:
    """MLPLayers(layers, dropout, 'leakyrelu', bn=True) of layers.py:56-85 as plain tensors: per layer
    Dropout -> Linear -> BatchNorm1d (batch statistics: the dict-held MLPs are never in eval mode, App. B-3)
    -> LeakyReLU(0.01), the last layer included."""

    def __init__(self, z, prefix: str, p_drop: float):
        self.p = p_drop
        self.lin_w, self.lin_b, self.bn_w, self.bn_b, self.rm, self.rv = [], [], [], [], [], []
        l = 0
        while f"{prefix}.mlp_layers.{4 * l + 1}.weight" in z:
            g = lambda s: torch.tensor(z[f"{prefix}.mlp_layers.{s}"])
            self.lin_w.append(g(f"{4 * l + 1}.weight").requires_grad_())
            self.lin_b.append(g(f"{4 * l + 1}.bias").requires_grad_())
            self.bn_w.append(g(f"{4 * l + 2}.weight").requires_grad_())
            self.bn_b.append(g(f"{4 * l + 2}.bias").requires_grad_())
            self.rm.append(g(f"{4 * l + 2}.running_mean"))
            self.rv.append(g(f"{4 * l + 2}.running_var"))
            l += 1

    def params(self) -> List[torch.Tensor]:
        out = []
        for l in range(len(self.lin_w)):
            out += [self.lin_w[l], self.lin_b[l], self.bn_w[l], self.bn_b[l]]   # nn.Module.parameters() order
        return out

    def __call__(self, x, masks=None):
        for l in range(len(self.lin_w)):
            if self.p > 0 and masks is not None:
                x = x * (masks[l] / (1.0 - self.p))
            x = F.linear(x, self.lin_w[l], self.lin_b[l])
            x = F.batch_norm(x, self.rm[l], self.rv[l], self.bn_w[l], self.bn_b[l], training=True, momentum=0.1, eps=1e-5)
            x = F.leaky_relu(x, 0.01)
        return x

    def export(self, prefix: str, out: Dict[str, np.ndarray]):
        for l in range(len(self.lin_w)):
            out[f"{prefix}.mlp_layers.{4 * l + 1}.weight"] = self.lin_w[l].detach().numpy().copy()
            out[f"{prefix}.mlp_layers.{4 * l + 1}.bias"] = self.lin_b[l].detach().numpy().copy()
            out[f"{prefix}.mlp_layers.{4 * l + 2}.weight"] = self.bn_w[l].detach().numpy().copy()
            out[f"{prefix}.mlp_layers.{4 * l + 2}.bias"] = self.bn_b[l].detach().numpy().copy()
            out[f"{prefix}.mlp_layers.{4 * l + 2}.running_mean"] = self.rm[l].numpy().copy()
            out[f"{prefix}.mlp_layers.{4 * l + 2}.running_var"] = self.rv[l].numpy().copy()


class PlainMLP:
    """MLPLayers without BatchNorm (PFCN_MLP's scorer, PFCN_DMF's towers): Dropout(0) -> Linear -> activation per layer."""

    def __init__(self, z, prefix: str, act):
        self.act = act
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
            x = self.act(F.linear(x, W, b))
        return x

    def export(self, prefix, out):
        for l, (W, b) in enumerate(zip(self.W, self.b)):
            out[f"{prefix}.mlp_layers.{3 * l + 1}.weight"] = W.detach().numpy().copy()
            out[f"{prefix}.mlp_layers.{3 * l + 1}.bias"] = b.detach().numpy().copy()


class Model:
    def __init__(self, z):
        self.kind, self.mode = str(z["model"]), str(z["mode"])
        self.attrs = [str(a) for a in z["attrs"]]
        self.lr, self.wd, self.dis_weight, self.p = (float(x) for x in z["hyper"])
        t = lambda k: torch.tensor(z["init.model." + k]).requires_grad_()
        self.uname = "user_embedding" if self.kind == "PFCN_MLP" else "user_embedding_layer"
        self.iname = "item_embedding" if self.kind == "PFCN_MLP" else "item_embedding_layer"
        self.U, self.I = t(self.uname + ".weight"), t(self.iname + ".weight")
        self.biased = self.kind == "PFCN_BiasedMF"
        self.scorer = PlainMLP(z, "init.model.mlp_layer", torch.relu) if self.kind == "PFCN_MLP" else None
        self.user_mlp = PlainMLP(z, "init.model.user_mlp", torch.relu) if self.kind == "PFCN_DMF" else None
        self.item_mlp = PlainMLP(z, "init.model.item_mlp", torch.relu) if self.kind == "PFCN_DMF" else None
        if self.biased:
            self.bu, self.bi, self.gb = t("user_bias.weight"), t("item_bias.weight"), t("global_bias")
        self.filters: Dict[int, MLP] = {}
        self.dis: Dict[str, MLP] = {}
        self.sst_size = {"gender": 2, "age": 3}
        if self.mode != "none":
            n_f = len(self.attrs) if self.mode == "cm" else 2 ** len(self.attrs) - 1
            for i in range(1, n_f + 1):
                self.filters[i] = MLP(z, f"init.filter.{i}", 0.0)
            for a in self.attrs:
                self.dis[a] = MLP(z, f"init.dis.{a}", self.p)
            # _get_filter_info, pfcn_biasedmf.py:67-84
            if self.mode == "cm":
                self.sst_dict = {a: i + 1 for i, a in enumerate(self.attrs)}
            else:
                self.sst_dict = {a: 2 ** i for i, a in enumerate(self.attrs)}

    def filter_params(self):
        ps = [self.U, self.I]
        for f in self.filters.values():
            ps += f.params()
        if self.biased:
            ps += [self.bu, self.bi, self.gb]
        return ps + self.base_mlp_params()

    def base_mlp_params(self):
        ps = []
        for m in (self.scorer, self.user_mlp, self.item_mlp):
            if m is not None:
                ps += m.params()
        return ps

    def all_params(self):        # nn.Module.parameters() of the reference model with filter_mode none
        if self.biased:
            return [self.gb, self.U, self.bu, self.I, self.bi]
        return [self.U, self.I] + self.base_mlp_params()

    def registered_params(self):   # nn.Module.parameters() of the reference model: the dict-held filter / discriminator
        ps = [self.U, self.I]      # MLPs are not among them (pfcn_biasedmf.py:110-142)
        if self.biased:
            ps += [self.bu, self.bi, self.gb]
        return ps + self.base_mlp_params()

    def dis_params(self):
        ps = []
        for d in self.dis.values():
            ps += d.params()
        return ps

    def user_embed(self, user, sst_list):
        """forward(), pfcn_biasedmf.py:144-166: none: raw rows; sm: ONE filter picked by the bit-mask sum; cm: sum of the
        selected attributes' filters divided by the number of ALL filters (App. B-2)."""
        ue = self.U[user]
        if self.user_mlp is not None:          # PFCN_DMF: tower before the filter (pfcn_dmf.py:150-151)
            ue = self.user_mlp(ue)
        if self.mode == "none":
            return ue
        if self.mode == "sm":
            return self.filters[sum(self.sst_dict[s] for s in sst_list)](ue)
        tmp = None
        for s in sst_list:
            e = self.filters[self.sst_dict[s]](ue)
            tmp = e if tmp is None else tmp + e
        return tmp / len(self.filters)

    def dis_loss(self, user, sst_list, labels, masks):
        """calculate_dis_loss, pfcn_biasedmf.py:202-218 (calls forward again)."""
        ue = self.user_embed(user, sst_list)
        total = 0.0
        for s in sst_list:
            y = self.dis[s](ue, masks[s] if masks else None)
            if self.sst_size[s] == 2:
                total = total + F.binary_cross_entropy(torch.sigmoid(y), labels[s].to(y.dtype).unsqueeze(1))
            else:
                total = total + F.cross_entropy(y, labels[s].long())
        return total

    def loss(self, user, pos, neg, sst_list, labels, masks):
        """calculate_loss, pfcn_biasedmf.py:180-200 incl. the [B] + [B,1] -> [B,B] broadcast of BiasedMF (App. B-1)."""
        ue = self.user_embed(user, sst_list)
        pe, ne = self.I[pos], self.I[neg]
        if self.kind == "PFCN_MLP":            # pfcn_mlp.py:185-186: scorer on cat(user, item), BPR on [B,1] scores
            ps, ns = self.scorer(torch.cat((ue, pe), 1)), self.scorer(torch.cat((ue, ne), 1))
        elif self.kind == "PFCN_DMF":          # pfcn_dmf.py:189-194: item tower, cosine similarity * 10
            ps = F.cosine_similarity(ue, self.item_mlp(pe)) * 10
            ns = F.cosine_similarity(ue, self.item_mlp(ne)) * 10
        else:
            ps, ns = (ue * pe).sum(-1), (ue * ne).sum(-1)
        if self.biased:
            ps = ps + self.bu[user] + self.bi[pos] + self.gb
            ns = ns + self.bu[user] + self.bi[neg] + self.gb
        bpr = -torch.log(1e-10 + torch.sigmoid(ps - ns)).mean()
        if self.mode != "none":
            return bpr - self.dis_weight * self.dis_loss(user, sst_list, labels, masks)
        return bpr

    def predict(self, user, item, sst_list):
        ue = self.user_embed(user, sst_list)
        if self.kind == "PFCN_MLP":
            return torch.sigmoid(self.scorer(torch.cat((ue, self.I[item]), 1)))
        if self.kind == "PFCN_DMF":
            return torch.sigmoid(F.cosine_similarity(ue, self.item_mlp(self.I[item])))
        s = (ue * self.I[item]).sum(-1, keepdim=True)
        if self.biased:
            s = s + self.bu[user] + self.bi[item] + self.gb
        return torch.sigmoid(s)


def train(z) -> Dict[str, np.ndarray]:
    m = Model(z)
    opt_f = torch.optim.Adam(m.all_params() if m.mode == "none" else m.filter_params(), lr=m.lr, weight_decay=m.wd)
    opt_d = torch.optim.Adam(m.dis_params(), lr=m.lr, weight_decay=m.wd) if m.mode != "none" else None
    phases = [str(p) for p in z["phases"]]
    losses, norms = [], []
    n_dis_layers = len(z["dis_hidden"]) + 1
    for t, ph in enumerate(phases):
        u, pi, ni = (torch.tensor(z[k][t]) for k in ("user_id", "item_id", "neg_item_id"))
        labels = {"gender": torch.tensor(z["gender"][z["user_id"][t]]), "age": torch.tensor(z["age"][z["user_id"][t]])}
        sl = [s for s in str(z["sst_lists"][t]).split(",") if s] if m.mode != "none" else None
        masks = None
        if m.mode != "none":
            masks = {s: [torch.tensor(z[f"mask.{s}.{t}.{l}"]).to(m.U.dtype) for l in range(n_dis_layers)] for s in sl}
        opt = opt_f if ph == "F" else opt_d
        opt.zero_grad()
        l = m.loss(u, pi, ni, sl, labels, masks) if ph == "F" else m.dis_loss(u, sl, labels, masks)
        losses.append(float(l.item()))
        l.backward()
        if "clip_max_norm" in z:       # trainer.py:925-926
            norms.append(float(torch.nn.utils.clip_grad_norm_(m.registered_params(), max_norm=float(z["clip_max_norm"]))))
        opt.step()
    out: Dict[str, np.ndarray] = {"loss": np.array(losses)}
    if norms:
        out["grad_norm"] = np.array(norms)
    out[f"final.model.{m.uname}.weight"] = m.U.detach().numpy().copy()
    out[f"final.model.{m.iname}.weight"] = m.I.detach().numpy().copy()
    for name, mlp in (("mlp_layer", m.scorer), ("user_mlp", m.user_mlp), ("item_mlp", m.item_mlp)):
        if mlp is not None:
            mlp.export(f"final.model.{name}", out)
    if m.biased:
        out["final.model.user_bias.weight"] = m.bu.detach().numpy().copy()
        out["final.model.item_bias.weight"] = m.bi.detach().numpy().copy()
        out["final.model.global_bias"] = m.gb.detach().numpy().copy()
    for i, f in m.filters.items():
        f.export(f"final.filter.{i}", out)
    for a, d in m.dis.items():
        d.export(f"final.dis.{a}", out)
    with torch.no_grad():
        out["predict_last"] = m.predict(u, pi, m.attrs if m.mode != "none" else None).numpy().copy()
    return out


def bpr_outer(a: torch.Tensor, c: torch.Tensor, chunk: int = 512):
    """PFCN_BiasedMF's BPR under the `[B] + [B,1] -> [B,B]` broadcast of pfcn_biasedmf.py:192-195 with BPRLoss
    (loss.py:45-47): mean over all (i, j) of -log(1e-10 + sigmoid(a_j + c_i)), where a = (u.pos - u.neg) per row j and
    c = (pos item bias - neg item bias) per row i.  The reference materialises the [B, B] matrix; this restatement walks
    it in row chunks in float64 and returns (loss, dLoss/da, dLoss/dc).  Test infrastructure."""
    a64, c64 = a.double(), c.double()
    B = a64.numel()
    total = torch.zeros((), dtype=torch.float64)
    da, dc = torch.zeros(B, dtype=torch.float64), torch.zeros(B, dtype=torch.float64)
    for lo in range(0, B, chunk):
        x = a64[None, :] + c64[lo:lo + chunk, None]               # rows i = lo.., columns j
        s = torch.sigmoid(x)
        total += -torch.log(1e-10 + s).sum()
        g = -(s * (1 - s)) / (1e-10 + s)                           # d(-log(1e-10 + sigmoid(x))) / dx
        da += g.sum(0)
        dc[lo:lo + chunk] = g.sum(1)
    n = float(B) * float(B)
    return total / n, da / n, dc / n
