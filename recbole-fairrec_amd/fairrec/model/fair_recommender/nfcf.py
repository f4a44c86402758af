"""NFCF (neural fair collaborative filtering) on the MI355X hot path.

Plugin surface of recbole/model/fair_recommender/nfcf.py:16-115: class name, ctor `(config, dataset)`, attributes
`user_embedding`, `item_embedding`, `mlp_layers`, config keys `embedding_size, mlp_hidden_size, dropout,
sst_attr_list, fair_weight, load_pretrain_path, LABEL_FIELD`.  Embeddings are lazy-Adam tables (csrc/table.hip),
the scorer runs on the fp32-MFMA kernels (csrc/mlp.hip), sigmoid + BCE + differential fairness on csrc/nfcf.hip.
"""
from __future__ import annotations


import torch
import torch.nn as nn

from ... import _C
from ...engine import GenericEngine
from ...utils.enum_type import InputType
from ..abstract_recommender import FairRecommender
from ..layers import MLPLayers


class _NfcfLoss(torch.autograd.Function):
    """loss = BCE(sigmoid(y), label) [+ fair_weight * DF]  via fr_nfcf_loss; backward scales the stored dLoss/dy."""

    @staticmethod
    def forward(ctx, y, label, sst, fair_weight, item_table, err_flag, sharded=None):
        """`sharded` = (row-sharded engine, item table name): the differential-fairness term is evaluated on the GLOBAL
        batch (ShardedGenericEngine.global_item_df) instead of on this rank's rows."""
        lib = _C.lib()
        B = y.numel()
        y = y.contiguous().view(-1)
        dev = y.device
        out = torch.empty(B, dtype=torch.float32, device=dev)
        dy = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty(3, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.fr_nfcf_loss_workspace_bytes(B), dtype=torch.uint8, device=dev)
        iws = item_table._ws if item_table is not None else None
        _C.check(lib.fr_nfcf_loss(y.data_ptr(), label.data_ptr(), _C.ptr(sst) if sharded is None else None, B, fair_weight,
                                  _C.ptr(iws), iws.numel() if iws is not None else 0,
                                  item_table.dim if item_table is not None else 1,
                                  out.data_ptr(), dy.data_ptr(), loss.data_ptr(), ws.data_ptr(), ws.numel(),
                                  err_flag.data_ptr(), _C.current_stream()), "fr_nfcf_loss")
        if sharded is not None:
            sharded[0].global_item_df(sharded[1], out, label, sst, fair_weight, dy, loss)
        ctx.save_for_backward(dy)
        ctx.shape = y.shape
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)      # no zero tensor for the gradient of `out` (one fill launch per step)
        return loss[0], out

    @staticmethod
    def backward(ctx, g_loss, g_out):
        (dy,) = ctx.saved_tensors
        if g_loss is None:
            return None, None, None, None, None, None, None
        if _C.is_one(g_loss):               # GraphedStep's seed: nothing to scale by
            return dy.view(-1, 1), None, None, None, None, None, None
        return (dy * g_loss).view(-1, 1), None, None, None, None, None, None


class NFCF(FairRecommender):
    input_type = InputType.POINTWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.LABEL = config['LABEL_FIELD']
        self.embedding_size = config['embedding_size']
        self.mlp_hidden_size = config['mlp_hidden_size']
        self.dropout = config['dropout']
        self.sst_attr = config['sst_attr_list'][0]
        self.fair_weight = config['fair_weight']
        self.load_pretrain_path = config['load_pretrain_path']

        self.user_embedding = nn.Embedding(self._table_rows(self.n_users), self.embedding_size)   # default N(0,1) init, as nfcf.py:38-39
        self.item_embedding = nn.Embedding(self._table_rows(self.n_items), self.embedding_size)
        self.mlp_layers = MLPLayers([2 * self.embedding_size] + list(self.mlp_hidden_size) + [1], self.dropout)
        self._engine = None
        if self.load_pretrain_path is not None:
            self.reset_params(self.load_pretrain_path, dataset.get_user_feature()[1:])

    def reset_params(self, pretrain_path, user_data):
        """nfcf.py:49-67: load the pre-trained weights, project the gender direction out of the user table
        (rows 1..), freeze it, re-initialise the item table.  One-off, whole-table, host-driven torch ops."""
        checkpoint = torch.load(pretrain_path, weights_only=False)
        self.load_state_dict(checkpoint['state_dict'], strict=False)
        dev = self.user_embedding.weight.device
        sst = user_data[self.sst_attr].to(dev)
        vals = torch.unique(sst)
        if self.shard is None:
            e = self.user_embedding.weight.data[1:].clone()
            m1 = e[sst == vals[0]].mean(dim=0)
            m2 = e[sst == vals[1]].mean(dim=0)
            vb = (m1 - m2) / torch.linalg.norm(m1 - m2, keepdim=True)
            self.user_embedding.weight.data[1:] = e - torch.mul(e, vb).sum(dim=1, keepdim=True) * vb
        else:
            # the checkpoint holds this rank's shard; the two group means need every rank's rows: one all-reduce of
            # two D-vectors + two counts.  Global row of local row l: l * world + rank; row 0 = [PAD].
            import torch.distributed as dist
            rank, world = self.shard
            w = self.user_embedding.weight.data
            rows = torch.arange(w.shape[0], device=dev) * world + rank
            real = rows >= 1
            s_loc = torch.zeros(w.shape[0], dtype=sst.dtype, device=dev)
            s_loc[real] = sst[rows[real] - 1]
            g0, g1 = real & (s_loc == vals[0]), real & (s_loc == vals[1])
            red = torch.cat([w[g0].sum(0), w[g1].sum(0), g0.sum().reshape(1).to(w.dtype), g1.sum().reshape(1).to(w.dtype)])
            dist.all_reduce(red)
            D = w.shape[1]
            m1, m2 = red[:D] / red[2 * D], red[D:2 * D] / red[2 * D + 1]
            vb = (m1 - m2) / torch.linalg.norm(m1 - m2, keepdim=True)
            e = w[real].clone()
            w[real] = e - torch.mul(e, vb).sum(dim=1, keepdim=True) * vb
        self.user_embedding.weight.requires_grad = False
        self.item_embedding = nn.Embedding(self._table_rows(self.n_items), self.embedding_size).to(dev)

    # --- engine -------------------------------------------------------------------------------------------
    def hip_engine(self) -> GenericEngine:
        uw = self.user_embedding.weight
        if self._engine is None or self._engine._tables["user_embedding.weight"].weight.data_ptr() != uw.data_ptr():
            if self.shard is None:
                eng = GenericEngine(uw.device)
                eng.add_table("user_embedding.weight", uw, trainable=uw.requires_grad)
                eng.add_table("item_embedding.weight", self.item_embedding.weight, trainable=True)
            else:
                from ...sharded_engine import ShardedGenericEngine
                eng = ShardedGenericEngine(uw.device)
                eng.add_table("user_embedding.weight", uw, trainable=uw.requires_grad, n_rows_global=self.n_users)
                eng.add_table("item_embedding.weight", self.item_embedding.weight, trainable=True,
                              n_rows_global=self.n_items)
            for name, p in self.mlp_layers.named_parameters():
                eng.add_dense("mlp_layers." + name, p)
            self._engine = eng
        return self._engine

    # --- plugin surface -------------------------------------------------------------------------------------
    def _score_logits(self, user, item):
        eng = self.hip_engine()
        ue, ie = eng.lookup_pair("user_embedding.weight", user, "item_embedding.weight", item)
        return self.mlp_layers(ue, ie)                     # [B, 1], after the last ReLU

    def forward(self, user, item):
        return torch.sigmoid(self._score_logits(user, item).squeeze(-1))

    def calculate_loss(self, interaction):
        eng = self.hip_engine()
        dev = eng.device
        user, item = interaction[self.USER_ID], interaction[self.ITEM_ID]
        label = interaction[self.LABEL].to(dev, torch.float32).contiguous()
        y = self._score_logits(user, item)
        finetune = self.load_pretrain_path is not None
        sst = interaction[self.sst_attr].to(dev, torch.float32).contiguous() if finetune else None
        if finetune and self.shard is not None:
            # row-sharded tables: the fairness term of the GLOBAL batch (per-group sums reduced on the items' owners)
            loss, _ = _NfcfLoss.apply(y, label, sst, float(self.fair_weight or 0.0), None, eng.err_flag,
                                      (eng, "item_embedding.weight"))
            return loss
        item_table = eng.batch_segments("item_embedding.weight") if finetune else None
        loss, _ = _NfcfLoss.apply(y, label, sst, float(self.fair_weight or 0.0), item_table, eng.err_flag)
        return loss

    def predict(self, interaction):
        return self.forward(interaction[self.USER_ID], interaction[self.ITEM_ID])

    def state_dict(self, *args, **kwargs):
        if self._engine is not None:
            self._engine.flush()
        return super().state_dict(*args, **kwargs)
