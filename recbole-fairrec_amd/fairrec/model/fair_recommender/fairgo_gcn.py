"""FairGo_GCN (fairgo_gcn.py): FairGo_PMF whose PRETRAIN stage passes the whole embedding table through a 2-layer GCN
before scoring (fairgo_gcn.py:52-57, :173-176); the finetune stage is line-identical to FairGo_PMF.

The reference builds that GCN with `torch_geometric.nn.GCN`, a third-party module it pins nowhere and that this image
does not have.  **Parity of the pretrain stage is therefore UNPINNED**: what is implemented here is PyG's PUBLISHED
algorithm (torch_geometric.nn.models.BasicGNN / GCN and torch_geometric.nn.conv.GCNConv with default arguments),
restated:

    GCNConv(in, out):  X' = Ahat (X W^T) + b,   Ahat = Dhat^-1/2 (A + I) Dhat^-1/2,
                       A = weighted adjacency (edge (j -> i) with weight w adds w to A[i, j]; one self loop of weight 1 per
                       node), Dhat[i] = sum_j (A + I)[i, j] (in-degree incl. the self loop), 0 where the degree is 0;
                       W: Linear without bias, glorot-uniform; b: zeros
    GCN(in, hidden, out, L layers, dropout p, act): conv -> act -> dropout(p) between layers, nothing after the last one
                       (jk = None), layer widths in -> hidden -> ... -> hidden -> out

and its oracle (`oracle/fairgo.py::gcn_forward`) restates the same in dense torch.  The graph is the reference's
(fairgo_gcn.py:59-66): both directions of every training rating, weight = the rating.

Because the GCN mixes every row into every other, the pretrain gradient of the embedding tables is dense: they are updated
by the dense fused Adam (fr_adam_dense_multi) like the reference's optimizer_pretrain, not through the lazy tables.
Ahat X is the SpMM kernel (symmetric, so the backward uses the same matrix), X W^T the fp32-MFMA linear kernel.
"""
import math

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...engine import GenericEngine
from ...functional import CsrMatrix, Mse, RowDot, RowGather, SplitRows, SpMM
from ..layers import _HipMLP
from .fairgo_pmf import FairGo_PMF


class _Lin(nn.Module):
    """torch_geometric.nn.dense.linear.Linear(in, out, bias=False, weight_initializer='glorot')."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        a = math.sqrt(6.0 / (in_channels + out_channels))            # torch_geometric.nn.inits.glorot
        nn.init.uniform_(self.weight, -a, a)


class _GCNConv(nn.Module):
    """Parameter names as in PyG (`lin.weight`, `bias`) so that checkpoints line up."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = _Lin(in_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.register_buffer("_zero_bias", torch.zeros(out_channels), persistent=False)

    def forward(self, x, a_hat: CsrMatrix):
        xw = _HipMLP.apply(x, None, 0, 0.0, None, None, None, self.lin.weight, self._zero_bias)     # X W^T (no bias)
        return SpMM.apply(xw, a_hat) + self.bias


class _GCN(nn.Module):
    ACTS = {"relu": F.relu, "leakyrelu": F.leaky_relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "elu": F.elu}

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, act):
        super().__init__()
        dims = [in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels]
        self.convs = nn.ModuleList(_GCNConv(dims[k], dims[k + 1]) for k in range(num_layers))
        self.dropout = float(dropout or 0.0)
        self.act = self.ACTS[(act or "relu").lower()]

    def forward(self, x, a_hat):
        for k, conv in enumerate(self.convs):
            x = conv(x, a_hat)
            if k == len(self.convs) - 1:
                break
            x = F.dropout(self.act(x), p=self.dropout, training=self.training)
        return x


def gcn_norm_matrix(n_users, n_items, rating_coo) -> sp.csr_matrix:
    """Ahat of torch_geometric.nn.conv.gcn_conv.gcn_norm for the reference's edge list (fairgo_gcn.py:59-66)."""
    N = n_users + n_items
    R = rating_coo.tocoo()
    rows = np.concatenate([R.row, R.col + n_users])          # targets i
    cols = np.concatenate([R.col + n_users, R.row])          # sources j     (edge list is symmetric)
    w = np.concatenate([R.data, R.data]).astype(np.float64)
    A = sp.coo_matrix((w, (rows, cols)), shape=(N, N)).tocsr() + sp.identity(N, format="csr")
    deg = np.asarray(A.sum(axis=1)).ravel()
    dis = np.where(deg > 0, 1.0 / np.sqrt(np.where(deg > 0, deg, 1.0)), 0.0)
    return (sp.diags(dis) @ A @ sp.diags(dis)).tocsr().astype(np.float32)


class FairGo_GCN(FairGo_PMF):
    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.gcn = _GCN(self.embedding_size, config['hidden_channels'] or 32, self.embedding_size,
                        config['gcn_n_layers'] or 2, config['gcn_dropout'], config['gcn_act'])
        self._a_hat_host = gcn_norm_matrix(self.n_users, self.n_items, self.rating_matrix)
        self._a_hat = None

    # --- engine: the tables are DENSE parameters of optimizer_pretrain here (trainer.py:850-855 adds gcn's too) -------
    def hip_engine(self) -> GenericEngine:
        uw = self.user_embedding_layer.weight
        if self._engine is None or self._engine._dense["user_embedding_layer.weight"].p.data_ptr() != uw.data_ptr():
            if self.replicas is not None:      # one replica per GPU, batch sharded: the tables are dense parameters here
                from ...replicated_engine import ReplicatedGenericEngine      # (the GCN makes their gradient dense), so
                eng = ReplicatedGenericEngine(uw.device)                      # the flat all-reduce covers them too
            else:
                eng = GenericEngine(uw.device)
            eng.add_dense("user_embedding_layer.weight", uw, group='pretrain')
            eng.add_dense("item_embedding_layer.weight", self.item_embedding_layer.weight, group='pretrain')
            for n, p in self.gcn.named_parameters():
                eng.add_dense(f"gcn.{n}", p, group='pretrain')
            for s, mlp in self.filter_layer_dict.items():
                for n, p in mlp.named_parameters():
                    eng.add_dense(f"filter.{s}.{n}", p, group='filter')
            for s, mlp in self.dis_layer_dict.items():
                for n, p in mlp.named_parameters():
                    eng.add_dense(f"dis.{s}.{n}", p, group='dis')
            if self.aggr_method == 'LBA':
                for n, p in self.aggr_layer.named_parameters():
                    eng.add_dense(f"aggr_layer.{n}", p, group='dis')
            self._engine = eng
            self._L = CsrMatrix(self._norm_csr_host, uw.device)
            self._a_hat = CsrMatrix(self._a_hat_host, uw.device)
        return self._engine

    def get_ego_embeddings(self):
        self.hip_engine()
        if self.train_stage == 'pretrain' and torch.is_grad_enabled():
            return torch.cat([self.user_embedding_layer.weight, self.item_embedding_layer.weight], dim=0)
        if self.train_stage == 'finetune':                 # frozen tables: the concatenation is kept (FairGo_PMF)
            return super().get_ego_embeddings()
        return torch.cat([self.user_embedding_layer.weight.data, self.item_embedding_layer.weight.data], dim=0)

    def _filtered_table(self, sst_list, grad_at_z=False):
        if self.train_stage == 'pretrain':                 # fairgo_gcn.py:175-176
            return self.gcn(self.get_ego_embeddings(), self._a_hat)
        return super()._filtered_table(sst_list, grad_at_z=grad_at_z)

    def calculate_loss(self, interaction, sst_list=None):
        if self.train_stage == 'finetune':
            return super().calculate_loss(interaction, sst_list)
        eng = self.hip_engine()
        user = interaction[self.USER_ID].to(eng.device)
        item = interaction[self.ITEM_ID].to(eng.device)
        rating = interaction[self.RATING].to(eng.device, torch.float32)
        E = self._filtered_table(None)                      # whole table through the GCN (fairgo_gcn.py:187-196)
        rows = RowGather.apply(E, torch.cat([user, item + self.n_users]), eng.err_flag)
        u_rows, i_rows = SplitRows.apply(rows, user.numel())
        return Mse.apply(RowDot.apply(u_rows, i_rows), rating)
