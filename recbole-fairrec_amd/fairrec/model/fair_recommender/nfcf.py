"""NFCF (neural fair collaborative filtering) on the MI355X hot path.

Plugin surface of recbole/model/fair_recommender/nfcf.py:16-115: class name, ctor `(config, dataset)`, attributes
`user_embedding`, `item_embedding`, `mlp_layers`, config keys `embedding_size, mlp_hidden_size, dropout,
sst_attr_list, fair_weight, load_pretrain_path, LABEL_FIELD`.  Embeddings are lazy-Adam tables (csrc/table.hip),
the scorer runs on the fp32-MFMA kernels (csrc/mlp.hip), sigmoid + BCE + differential fairness on csrc/nfcf.hip.
"""
from __future__ import annotations

import ctypes
import os

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


def _r4(n):
    return (n + 3) // 4 * 4


class _NfcfFused(torch.autograd.Function):
    """calculate_loss (nfcf.py:99-110) on gathered rows as one autograd node: fr_scorer_fwd (the three layers, their dropouts,
    sigmoid, BCE: one launch) + fr_nfcf_loss_tail (differential fairness, loss) forward; fr_scorer_bwd (one launch) + the two
    weight-gradient products + fr_parts_sum backward.  Same values as MLPLayers + _NfcfLoss layer by layer -- same dropout
    pattern included (the four offsets below are `_Drop`'s) -- up to the order of the fp32 sums."""

    @staticmethod
    def _desc(k0, k1, params, p, seed, B):
        W1, b1, W2, b2, W3, b3 = params
        n1, n2 = W1.shape[0], W2.shape[0]
        o1 = _r4(B * k0)
        o2 = o1 + _r4(B * k1)
        return _C.FrScorer(k0, k1, n1, n2, W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3.data_ptr(),
                           b3.data_ptr(), p, seed, 0, o1, o2, o2 + _r4(B * n1))

    @staticmethod
    def supported(k0, k1, params, p) -> bool:
        if len(params) != 6 or params[4].shape[0] != 1:
            return False
        return bool(_C.lib().fr_scorer_supported(_NfcfFused._desc(k0, k1, params, p, 0, 1)))

    @staticmethod
    def forward(ctx, x0, x1, label, sst, fair_weight, item_table, err_flag, drop, *params):
        lib, st = _C.lib(), _C.current_stream()
        x0, x1 = x0.contiguous(), x1.contiguous()
        params = tuple(t.contiguous() for t in params)
        B, k0, k1 = x0.shape[0], x0.shape[1], x1.shape[1]
        n1, n2 = params[0].shape[0], params[2].shape[0]
        dev = x0.device
        p, seed, state = drop if drop is not None else (0.0, 0, None)
        d = _NfcfFused._desc(k0, k1, params, p, seed, B)
        f32 = dict(dtype=torch.float32, device=dev)
        nblk = lib.fr_scorer_blocks(B)
        x0d = torch.empty_like(x0) if p > 0 else None
        x1d = torch.empty_like(x1) if p > 0 else None
        used = torch.empty(1, dtype=torch.int64, device=dev) if p > 0 else None
        h1, h2 = torch.empty((B, n1), **f32), torch.empty((B, n2), **f32)
        y, out, dy = torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32)
        part = torch.empty(3 * nblk, **f32)        # bce_part | mm_part
        loss = torch.empty(3, **f32)
        _C.check(lib.fr_scorer_fwd(ctypes.byref(d), x0.data_ptr(), x1.data_ptr(), B, _C.ptr(state), _C.ptr(used), _C.ptr(state),
                                   _C.ptr(x0d), _C.ptr(x1d), h1.data_ptr(), h2.data_ptr(), y.data_ptr(), label.data_ptr(),
                                   _C.ptr(sst), out.data_ptr(), dy.data_ptr(), part.data_ptr(), part[nblk:].data_ptr(),
                                   loss.data_ptr() if item_table is None else None, st), "fr_scorer_fwd")
        iws = item_table._ws if item_table is not None else None
        if iws is not None:     # (without the fairness term the forward launch has closed the loss itself)
            ws = torch.empty(lib.fr_nfcf_loss_workspace_bytes(B), dtype=torch.uint8, device=dev)
            _C.check(lib.fr_nfcf_loss_tail(label.data_ptr(), _C.ptr(sst), B, fair_weight, _C.ptr(iws),
                                           iws.numel(), item_table.dim, out.data_ptr(), dy.data_ptr(), loss.data_ptr(),
                                           part.data_ptr(), part[nblk:].data_ptr(), nblk, ws.data_ptr(), ws.numel(),
                                           err_flag.data_ptr(), st), "fr_nfcf_loss_tail")
        ctx.save_for_backward(dy, y, h1, h2, x0d if p > 0 else x0, x1d if p > 0 else x1, used, *params)
        ctx.meta = (k0, k1, p, seed, B, nblk)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        return loss[0], out

    @staticmethod
    def backward(ctx, g_loss, g_out):
        n_in = 8
        if g_loss is None:
            return (None,) * (n_in + 6)
        lib, st = _C.lib(), _C.current_stream()
        dy, y, h1, h2, xa, xb, used, *params = ctx.saved_tensors
        k0, k1, p, seed, B, nblk = ctx.meta
        n1, n2 = params[0].shape[0], params[2].shape[0]
        dev = dy.device
        d = _NfcfFused._desc(k0, k1, params, p, seed, B)
        f32 = dict(dtype=torch.float32, device=dev)
        gscale = None if _C.is_one(g_loss) else g_loss.detach().to(torch.float32).reshape(1).contiguous()
        dz1, dz2, dz3 = torch.empty((B, n1), **f32), torch.empty((B, n2), **f32), torch.empty((B, 1), **f32)
        dx0 = torch.empty((B, k0), **f32) if ctx.needs_input_grad[0] else None
        dx1 = torch.empty((B, k1), **f32) if ctx.needs_input_grad[1] else None
        w3part = torch.empty((nblk, n2 + 1), **f32)
        _C.check(lib.fr_scorer_bwd(ctypes.byref(d), dy.data_ptr(), _C.ptr(gscale), y.data_ptr(), h1.data_ptr(), h2.data_ptr(), B,
                                   _C.ptr(used), dz1.data_ptr(), dz2.data_ptr(), dz3.data_ptr(), _C.ptr(dx0), _C.ptr(dx1),
                                   w3part.data_ptr(), st), "fr_scorer_bwd")
        grads = [None] * 6
        if any(ctx.needs_input_grad[n_in:]):
            # the three layers' weight gradients: every product in one launch, every slab sum in a second
            dW1, db1, dW2, db2 = (torch.empty_like(t) for t in params[:4])
            w3 = torch.empty(n2 + 1, **f32)
            jobs = (_C.FrWgradJob * 3)(
                _C.FrWgradJob(dz1.data_ptr(), xa.data_ptr(), k0, xb.data_ptr(), k1, n1, dW1.data_ptr(), db1.data_ptr(), None, 0),
                _C.FrWgradJob(dz2.data_ptr(), h1.data_ptr(), n1, None, 0, n2, dW2.data_ptr(), db2.data_ptr(), None, 0),
                _C.FrWgradJob(None, None, n2 + 1, None, 0, 1, w3.data_ptr(), None, w3part.data_ptr(), nblk))
            ws = torch.empty(lib.fr_linear_bwd_weight_multi_workspace_bytes(jobs, 3, B), dtype=torch.uint8, device=dev)
            _C.check(lib.fr_linear_bwd_weight_multi(jobs, 3, B, ws.data_ptr(), ws.numel(), st), "fr_linear_bwd_weight_multi")
            grads = [dW1, db1, dW2, db2, w3[:n2].view(1, n2), w3[n2:]]
        return (dx0, dx1, None, None, None, None, None, None, *grads)


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

    FUSED = os.environ.get("FAIRREC_NFCF_LAYERED") is None     # A/B switch: the layer-by-layer form of round 3

    def _fused_scorer(self) -> bool:
        """Does calculate_loss take the two-launch scorer (csrc/scorer.hip)?  [2 D, n1, n2, 1] with ReLU, no BatchNorm, no
        recorded dropout masks, widths that are multiples of 32 within the kernel's LDS budget; anything else runs layer by
        layer on the same device (fairrec/model/layers.py)."""
        mlp = self.mlp_layers
        act = mlp.activation.lower() if isinstance(mlp.activation, str) else mlp.activation
        if not self.FUSED or mlp.use_bn or act != "relu" or mlp.forced_masks is not None:
            return False
        lins = mlp.linears()
        p = float(mlp.dropout) if mlp.training else 0.0
        return len(lins) == 3 and _NfcfFused.supported(self.embedding_size, self.embedding_size,
                                                       [t for lin in lins for t in (lin.weight, lin.bias)], p)

    def calculate_loss(self, interaction):
        eng = self.hip_engine()
        dev = eng.device
        user, item = interaction[self.USER_ID], interaction[self.ITEM_ID]
        label = interaction[self.LABEL].to(dev, torch.float32).contiguous()
        finetune = self.load_pretrain_path is not None
        sst = interaction[self.sst_attr].to(dev, torch.float32).contiguous() if finetune else None
        if self.shard is None and self._fused_scorer():
            mlp = self.mlp_layers
            ue, ie = eng.lookup_pair("user_embedding.weight", user, "item_embedding.weight", item)
            p = float(mlp.dropout) if mlp.training else 0.0
            drop = (p, mlp._drop_seed(), mlp._drop_state(dev)) if p > 0.0 else None
            item_table = eng.batch_segments("item_embedding.weight") if finetune else None
            params = [t for lin in mlp.linears() for t in (lin.weight, lin.bias)]
            loss, _ = _NfcfFused.apply(ue, ie, label, sst, float(self.fair_weight or 0.0), item_table, eng.err_flag, drop, *params)
            return loss
        y = self._score_logits(user, item)
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
