"""Shared machinery of the PFCN family on the MI355X hot path (PFCN_PMF, PFCN_BiasedMF).

PFCN = adversarially filtered user embeddings (Li et al., "Towards personalized fairness based on causal notion").
Plugin surface of recbole/model/fair_recommender/pfcn_biasedmf.py:24-242 / pfcn_pmf.py: attributes
`user_embedding_layer`, `item_embedding_layer`, [`user_bias`, `item_bias`, `global_bias`], `filter_layer` (dict),
`dis_layer_dict` (dict), `sst_dict`, `sst_size`; methods `forward / calculate_loss / calculate_dis_loss / predict /
get_sst_embed`; config keys `embedding_size, sst_attr_list, filter_mode, dis_dropout, dis_weight,
dis_hidden_size_list, activation`.  As in the reference the filter / discriminator MLPs live in plain dicts: they are
not in `parameters()` / `state_dict()`, and `model.eval()` never reaches them (BatchNorm keeps using batch statistics,
SURVEY.md App. B-3).  Embedding rows come from lazy-Adam tables, MLPs run on the fp32-MFMA kernels, dot products /
BPR / BCE / CE on csrc/pfcn.hip.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ...engine import GenericEngine
from ...functional import Bpr, BprBroadcastGlobal, BprBroadcastPacked, RowDot, RowDotPair, SigmoidBce, SoftmaxCe, SplitRows, SubScaled
from ...utils.enum_type import InputType
from ..abstract_recommender import FairRecommender
from ..layers import MLPLayers


class PFCNBase(FairRecommender):
    input_type = InputType.PAIRWISE
    biased = False
    user_table_attr = "user_embedding_layer"     # attribute / state_dict names of the two tables
    item_table_attr = "item_embedding_layer"

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.embedding_size = config['embedding_size']
        self.sst_attrs = config['sst_attr_list']
        self.filter_mode = config['filter_mode'].lower()
        try:
            assert self.filter_mode in ('cm', 'sm', 'none')
        except AssertionError:
            raise AssertionError('filter_mode must be cm, sm or none')
        self.filter_num, self.sst_dict = self._get_filter_info()
        self.sst_size = self._get_sst_size(dataset.get_user_feature())
        if self.filter_mode != 'none':
            self.dis_drop_out = config['dis_dropout']
            self.dis_weight = config['dis_weight']
            self.dis_hidden_size_list = config['dis_hidden_size_list']
        self.activation = config['activation']

        setattr(self, self.user_table_attr, nn.Embedding(self._table_rows(self.n_users), self.embedding_size))
        if self.biased:
            self.user_bias = nn.Embedding(self._table_rows(self.n_users), 1)
        setattr(self, self.item_table_attr, nn.Embedding(self._table_rows(self.n_items), self.embedding_size))
        self._build_base_layers(config)
        if self.biased:
            self.item_bias = nn.Embedding(self._table_rows(self.n_items), 1)
            self.global_bias = nn.Parameter(torch.tensor(0.1))
        if self.filter_mode != 'none':
            self.filter_layer = self.init_filter()
            self.dis_layer_dict = self.init_dis_layer()
        self._engine = None

    # --- hooks of the base recommender (PMF: nothing; MLP: scorer; DMF: towers + cosine) ------------------------------
    def _build_base_layers(self, config):
        pass

    def _base_dense_modules(self):
        """{state_dict prefix: registered MLP module} trained by optimizer_filter / the default optimizer."""
        return {}

    def _user_tower(self, rows):
        return rows

    def _item_tower(self, rows):
        return rows

    def _score(self, user_embed, item_embed):
        return RowDot.apply(user_embed, item_embed)

    def _filter_activation(self):
        return self.activation

    def _dis_activation(self):
        return self.activation

    @property
    def _utab(self):
        return self.user_table_attr + ".weight"

    @property
    def _itab(self):
        return self.item_table_attr + ".weight"

    # --- construction (pfcn_biasedmf.py:67-142) ---------------------------------------------------------------
    def _get_filter_info(self):
        if self.filter_mode == 'cm':
            return len(self.sst_attrs), {sst: i + 1 for i, sst in enumerate(self.sst_attrs)}
        if self.filter_mode == 'sm':
            return 2 ** len(self.sst_attrs) - 1, {sst: 2 ** i for i, sst in enumerate(self.sst_attrs)}
        return 0, {}

    def _get_sst_size(self, user_feature):
        sst_size = {}
        for sst in self.sst_attrs:
            if sst not in user_feature.columns:
                raise ValueError(f'{sst} sensitive attribute not in user feature')
            sst_size[sst] = len(user_feature[sst][1:].unique())
        return sst_size

    def init_filter(self):
        D = self.embedding_size
        return {i + 1: MLPLayers([D, D * 2, D], activation=self._filter_activation(), bn=True, init_method='norm').to(self.device)
                for i in range(self.filter_num)}

    def init_dis_layer(self):
        D = self.embedding_size
        out = {}
        for sst in self.sst_attrs:
            output_dim = self.sst_size[sst]
            if output_dim == 2:
                output_dim = 1
            out[sst] = MLPLayers([D] + list(self.dis_hidden_size_list) + [output_dim], dropout=self.dis_drop_out,
                                 activation=self._dis_activation(), bn=True, init_method='norm').to(self.device)
        return out

    # --- engine: which optimizer owns what (trainer.py:1201-1235) ---------------------------------------------------
    def hip_engine(self) -> GenericEngine:
        uw = getattr(self, self.user_table_attr).weight
        if self._engine is None or self._engine._tables[self._utab].weight.data_ptr() != uw.data_ptr():
            g = 'filter' if self.filter_mode != 'none' else None
            if self.shard is None:
                eng = GenericEngine(uw.device)
                tab = lambda name, w, n: eng.add_table(name, w, group=g)
            else:   # row-sharded tables, replicated MLPs; BatchNorm statistics are per-rank (fairrec/sharded_engine.py)
                from ...sharded_engine import ShardedGenericEngine
                eng = ShardedGenericEngine(uw.device)
                tab = lambda name, w, n: eng.add_table(name, w, group=g, n_rows_global=n)
            tab(self._utab, uw, self.n_users)
            tab(self._itab, getattr(self, self.item_table_attr).weight, self.n_items)
            for prefix, mod in self._base_dense_modules().items():
                for n, p in mod.named_parameters():
                    eng.add_dense(f"{prefix}.{n}", p, group=g)
            if self.biased:
                tab("user_bias.weight", self.user_bias.weight, self.n_users)
                tab("item_bias.weight", self.item_bias.weight, self.n_items)
                eng.add_dense("global_bias", self.global_bias, group=g)
            if self.filter_mode != 'none':
                for i, mlp in self.filter_layer.items():
                    for n, p in mlp.named_parameters():
                        eng.add_dense(f"filter.{i}.{n}", p, group='filter')
                for sst, mlp in self.dis_layer_dict.items():
                    for n, p in mlp.named_parameters():
                        eng.add_dense(f"dis.{sst}.{n}", p, group='dis')
            self._engine = eng
        return self._engine

    # --- forward pieces -------------------------------------------------------------------------------------------
    def _filter(self, user_embed, sst_list, passes=1):
        """pfcn_biasedmf.py:149-164: sm = ONE filter picked by the bit-mask sum of the selected attributes;
        cm = sum of the selected attributes' filters divided by the number of ALL filters."""
        if self.filter_mode == 'none':
            return user_embed
        if self.filter_mode == 'sm':
            return self.filter_layer[sum(self.sst_dict[s] for s in sst_list)](user_embed, passes=passes)
        tmp = None
        for s in sst_list:
            e = self.filter_layer[self.sst_dict[s]](user_embed, passes=passes)
            tmp = e if tmp is None else tmp + e
        return tmp / len(self.filter_layer)

    def forward(self, user, item=None, sst_list=None):
        eng = self.hip_engine()
        user_embed = self._filter(self._user_tower(eng.lookup(self._utab, user)), sst_list)
        item_embed = self._item_tower(eng.lookup(self._itab, item)) if item is not None else None
        return user_embed, item_embed

    def _dis_terms(self, user_embed, interaction, sst_list, frozen=False):
        """`frozen`: inside the filter pass the discriminators only pass the gradient on to the filters and embeddings;
        their own parameter gradients would be dropped unread by the next zero_grad (engine.zero_grad)."""
        eng = self.hip_engine()
        total = None            # (0.0 + loss would be a launch of its own)
        for sst in sst_list:
            y = self.dis_layer_dict[sst](user_embed, frozen=frozen)
            label = interaction[sst].to(eng.device)
            if self.sst_size[sst] == 2:
                term = SigmoidBce.apply(y, label.float())
            else:
                term = SoftmaxCe.apply(y, label.long(), eng.err_flag)
            total = term if total is None else total + term
        return total if total is not None else 0.0

    def calculate_dis_loss(self, interaction, sst_list):
        """pfcn_biasedmf.py:202-218, called on its own in the discriminator phase: only the discriminators are
        trained there, so the user rows are read without building a gradient path (the reference lets autograd fill
        U.grad and then never uses it); the filters still run forward -- their BatchNorm statistics advance."""
        eng = self.hip_engine()
        user = interaction[self.USER_ID]
        with torch.no_grad():
            user_embed = self._filter(self._user_tower(eng.lookup(self._utab, user)), sst_list)
        return self._dis_terms(user_embed, interaction, sst_list)

    def calculate_loss(self, interaction, sst_list=None):
        eng = self.hip_engine()
        user = interaction[self.USER_ID]
        pos_item, neg_item = interaction[self.POS_ITEM_ID], interaction[self.NEG_ITEM_ID]
        B = user.numel()
        ue_raw = eng.lookup(self._utab, user)
        # The reference runs the user side twice per filter step (forward() again inside calculate_dis_loss,
        # pfcn_biasedmf.py:209): same rows, same filters, both results used.  When nothing on that path is random (identity
        # tower; the filters have no dropout) it is evaluated once with the BatchNorm bookkeeping of two passes
        # (MLPLayers.forward, `passes`).
        once = self.filter_mode != 'none' and type(self)._user_tower is PFCNBase._user_tower
        user_embed = self._filter(self._user_tower(ue_raw), sst_list, passes=2 if once else 1)
        items = torch.cat([pos_item.to(eng.device), neg_item.to(eng.device)])     # one gather for both id lists
        ie = eng.lookup(self._itab, items)
        # plain row dots against rows of ONE lookup: both scores in one launch each way (functional.RowDotPair: RowDot's values,
        # the user rows' two gradients handed to autograd unsummed and in RowDot's order); FAIRREC_PFCN_ROWDOT_SEPARATE=1 or a
        # tower / scorer of its own: the pair of calls
        pair = (type(self)._score is PFCNBase._score and type(self)._item_tower is PFCNBase._item_tower
                and os.environ.get("FAIRREC_PFCN_ROWDOT_SEPARATE", "0") != "1")
        if pair:
            scores = RowDotPair.apply(user_embed, user_embed, ie)
        else:
            ie_pos, ie_neg = SplitRows.apply(ie, B)       # (autograd's slices cost five launches on the way back)
            pos_e, neg_e = self._item_tower(ie_pos), self._item_tower(ie_neg)
            dp, dn = self._score(user_embed, pos_e), self._score(user_embed, neg_e)
            scores = None
        if self.biased:
            ub = eng.lookup("user_bias.weight", user)
            ib = eng.lookup("item_bias.weight", items)
            if scores is None:
                scores = torch.cat([dp, dn])
            # packed columns: the differences and all four gradient columns come out of the loss kernel (bit-identical to
            # BprBroadcast on the slices, without its ten elementwise launches)
            if self.shard is not None:      # row-sharded tables: the broadcast runs over the GLOBAL batch (one all-gather)
                bpr_loss = BprBroadcastGlobal.apply(scores, ub, ib, self.global_bias, eng)
            else:
                bpr_loss = BprBroadcastPacked.apply(scores, ub, ib, self.global_bias)
        else:
            if scores is not None:
                dp, dn = SplitRows.apply(scores, B)
            bpr_loss = Bpr.apply(dp, dn)
        if self.filter_mode != 'none':
            # the reference calls forward() a second time inside calculate_dis_loss (pfcn_biasedmf.py:209): same rows,
            # filters applied again (BatchNorm statistics advance twice), gradient flows through both passes
            again = user_embed if once else self._filter(self._user_tower(ue_raw), sst_list)
            dis_loss = self._dis_terms(again, interaction, sst_list, frozen=True)
            return SubScaled.apply(bpr_loss, dis_loss, self.dis_weight)     # one launch each way; bpr_loss's gradient is the seed itself
        return bpr_loss

    def predict(self, interaction, sst_list=None):
        eng = self.hip_engine()
        user, item = interaction[self.USER_ID], interaction[self.ITEM_ID]
        with torch.no_grad():
            ue, ie = self.forward(user, item, sst_list)
            score = self._predict_score(ue, ie)
            if self.biased:
                score = score + eng.lookup("user_bias.weight", user) + eng.lookup("item_bias.weight", item) + self.global_bias
            return torch.sigmoid(score)

    def _predict_score(self, ue, ie):
        return RowDot.apply(ue, ie).unsqueeze(-1)

    def get_sst_embed(self, user_data, sst_list=None):
        ret = {}
        idx = torch.arange(1, self.n_users)
        sst_list = self.sst_attrs if self.filter_mode == 'none' else sst_list
        for sst in sst_list:
            ret[sst] = user_data[sst][idx - 1]
        with torch.no_grad():
            ret['embedding'], _ = self.forward(idx.to(self.device), None, sst_list)
        return ret

    def state_dict(self, *args, **kwargs):
        if self._engine is not None:
            self._engine.flush()
        return super().state_dict(*args, **kwargs)
