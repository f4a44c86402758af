"""FairGo_PMF (Wu et al., "Learning fair representations for recommendation: a graph-based perspective") on the MI355X
hot path.

Plugin surface of recbole/model/fair_recommender/fairgo_pmf.py:21-268: attributes `user_embedding_layer`,
`item_embedding_layer`, `dis_layer_dict`, `filter_layer_dict` (plain dicts), `aggr_layer`, mutable `train_stage`;
methods `forward / calculate_loss / calculate_dis_loss / predict / full_sort_predict / get_sst_embed`; config keys
`n_layers, activation, embedding_size, dis_hidden_size_list, filter_hidden_size_list, sst_attr_list, fair_weight,
load_pretrain_weight, aggr_method, vs_weights`.

pretrain  : plain MF regression on lazy-Adam tables (gather -> row dot -> MSE -> duplicate-summed update).
finetune  : embeddings frozen; per step, as in the reference, the filter MLPs run over the WHOLE [n_users+n_items, D]
            table on the fp32-MFMA kernels, `n_layers` CSR SpMMs with L = D^-1 A produce the local (graph) embeddings,
            WAP / LBA / LVA aggregate them, discriminators score node and local embeddings.
"""
from __future__ import annotations

import os

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from ... import _C
from ...engine import GenericEngine
from ...functional import CsrMatrix, GatherAndSpMMSel, Mse, RowDot, RowGather, SigmoidBce, SoftmaxCe, SplitRows, SpMM, SpMMSel
from ...utils.enum_type import InputType
from ..abstract_recommender import FairRecommender
from ..layers import ACT_CODES, MLPLayers, _HipMLP, activation_layer


class FairGo_PMF(FairRecommender):
    input_type = InputType.POINTWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        if self.shard is not None:
            # the finetune stage filters and propagates the WHOLE frozen tables every step (fairgo_pmf.py:175-199): it wants
            # replicas (exact), not shards; the pretrain stage alone could shard like PFCN_PMF
            raise NotImplementedError('FairGo reads whole tables every finetune step: run it with replicated tables '
                                      '(data_parallel: True), not row_sharded')
        self.RATING = config['RATING_FIELD']
        self.n_layers = config['n_layers']
        self.act = config['activation']
        self.embedding_size = config['embedding_size']
        self.dis_hidden_size_list = config['dis_hidden_size_list']
        self.filter_hidden_size_list = config['filter_hidden_size_list']
        self.sst_attrs = config['sst_attr_list']
        self.fair_weight = config['fair_weight']
        self.load_pretrain_weight = config['load_pretrain_weight']
        self.train_stage = None
        self.aggr_method = config['aggr_method'].upper()
        self.vs_weights = None
        self.frontier_mode = config['fairgo_frontier'] if config['fairgo_frontier'] is not None else 'auto'
        if config['vs_weights'] is not None:
            vs = torch.tensor(config['vs_weights'], dtype=torch.float32)
            self.vs_weights = vs / vs.sum()
            if self.aggr_method == 'LVA':
                assert self.n_layers == len(self.vs_weights), 'n_layers should be equal to length of vs_weights'
        self.max_rating = dataset.inter_feat[self.RATING].max()
        self.rating_matrix = dataset.inter_matrix(form='coo', value_field=self.RATING).astype(np.float32)
        self.sst_size = self._get_sst_size(dataset.get_user_feature())

        self.user_embedding_layer = nn.Embedding(self.n_users, self.embedding_size, padding_idx=0)
        self.item_embedding_layer = nn.Embedding(self.n_items, self.embedding_size, padding_idx=0)
        if self.load_pretrain_weight:
            self.user_embedding_layer.weight.data.copy_(torch.from_numpy(dataset.get_preload_weight('uid')))
            self.item_embedding_layer.weight.data.copy_(torch.from_numpy(dataset.get_preload_weight('iid')))
        self.dis_layer_dict = self.init_dis_layers()
        self.filter_layer_dict = self.init_filter_layers()
        D = self.embedding_size
        self.aggr_layer = nn.Sequential(nn.Linear(self.n_layers * D, D), activation_layer(self.act), nn.Linear(D, D),
                                        activation_layer(self.act), nn.Linear(D, D))
        self._norm_csr_host = self.get_norm_rating_matrix()      # scipy CSR of L = D^-1 A (one-off host preprocessing)
        self._L = None
        self._engine = None
        # discriminator phase: filtered table + propagated embeddings per attribute subset, valid while the filters rest
        self._dis_cache = {}

    # --- construction ---------------------------------------------------------------------------------------------
    def _get_sst_size(self, user_feature):
        sst_size = {}
        for sst in self.sst_attrs:
            if sst not in user_feature.columns:
                raise ValueError(f'{sst} sensitive attribute not in user feature')
            sst_size[sst] = len(user_feature[sst][1:].unique())
        return sst_size

    def get_norm_rating_matrix(self):
        """fairgo_pmf.py:102-129: A = weighted bipartite adjacency of the training ratings, L = diag(1/(rowsum+1e-7)) A."""
        N = self.n_users + self.n_items
        R = self.rating_matrix.tocoo()
        A = sp.coo_matrix((np.concatenate([R.data, R.data]),
                           (np.concatenate([R.row, R.col + self.n_users]), np.concatenate([R.col + self.n_users, R.row]))),
                          shape=(N, N), dtype=np.float32).tocsr()
        diag = 1.0 / (np.asarray(A.sum(axis=1)).flatten() + 1e-7)
        return (sp.diags(diag.astype(np.float32)) * A).tocsr().astype(np.float32)

    def init_dis_layers(self):
        out = {}
        for sst in self.sst_attrs:
            od = self.sst_size[sst]
            od = 1 if od == 2 else od
            out[sst] = MLPLayers([self.embedding_size] + list(self.dis_hidden_size_list) + [od], activation=self.act).to(self.device)
        return out

    def init_filter_layers(self):
        return {sst: MLPLayers([self.embedding_size] + list(self.filter_hidden_size_list) + [self.embedding_size],
                               activation=self.act).to(self.device) for sst in self.sst_attrs}

    # --- engine: optimizer_pretrain / optimizer_filter / optimizer_dis (trainer.py:837-847) ---------------------------
    def hip_engine(self) -> GenericEngine:
        uw = self.user_embedding_layer.weight
        if self._engine is None or self._engine._tables["user_embedding_layer.weight"].weight.data_ptr() != uw.data_ptr():
            if self.replicas is not None:      # one replica per GPU, batch sharded (fairrec/replicated_engine.py)
                from ...replicated_engine import ReplicatedGenericEngine
                eng = ReplicatedGenericEngine(uw.device)
            else:
                eng = GenericEngine(uw.device)
            eng.add_table("user_embedding_layer.weight", uw, group='pretrain')
            eng.add_table("item_embedding_layer.weight", self.item_embedding_layer.weight, group='pretrain')
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
        return self._engine

    # --- forward pieces -----------------------------------------------------------------------------------------------
    def get_ego_embeddings(self):
        """cat(user table, item table) (fairgo_pmf.py:131-140).  The finetune stage freezes both tables (trainer.py:857-862), so the
        concatenation of one step is the next step's: it is kept while neither table moved (torch's version counters for
        in-place writes such as load_state_dict, the engine's step count of the 'pretrain' group for the kernels' writes) --
        11 GB of copy traffic per filter step at BASELINE configs[3]."""
        eng = self.hip_engine()
        eng.flush()      # no-op unless the pretrain stage left rows behind the optimizer step
        uw, iw = self.user_embedding_layer.weight, self.item_embedding_layer.weight
        if self.train_stage != 'finetune':
            return torch.cat([uw.data, iw.data], dim=0)
        key = (uw.data_ptr(), iw.data_ptr(), uw._version, iw._version, eng.group_version('pretrain'))
        c = getattr(self, '_ego_cache', None)
        if c is None or c[0] != key:
            c = self._ego_cache = (key, torch.cat([uw.data, iw.data], dim=0))
        return c[1]

    def _top_fold_ok(self, sst_list) -> bool:
        """May the filtered table's ONE consumer take the top activation's derivative into its own backward launches
        (MLPLayers.forward(grad_at_z=True) + GatherAndSpMMSel(act=...))?  One filter whose output IS the table (no sum, no
        division), a plain activation on its top layer, no BatchNorm, no dropout.  FAIRREC_FAIRGO_ACT_SEPARATE=1: never."""
        if os.environ.get("FAIRREC_FAIRGO_ACT_SEPARATE") is not None or self.train_stage != 'finetune':
            return False
        sst_list = self.sst_attrs if sst_list is None else sst_list
        if len(self.filter_layer_dict) != 1 or len(sst_list) != 1:
            return False
        mlp = self.filter_layer_dict[sst_list[0]]
        name = mlp.activation.lower() if isinstance(mlp.activation, str) else mlp.activation
        return (not mlp.use_bn and ACT_CODES[name] != 0 and mlp.forced_masks is None
                and not (mlp.training and float(mlp.dropout) > 0.0))

    def _filtered_table(self, sst_list, grad_at_z=False):
        E = self.get_ego_embeddings()
        if self.train_stage == 'finetune':
            if sst_list is None:
                sst_list = self.sst_attrs
            tmp = None
            for sst in sst_list:
                e = self.filter_layer_dict[sst](E, grad_at_z=grad_at_z)
                tmp = e if tmp is None else tmp + e
            # (x / 1 is x, bit for bit, and so is its gradient: with ONE filter the whole-table division and its backward --
            # 22 GB of traffic per filter step at BASELINE configs[3] -- are skipped)
            E = tmp if len(self.filter_layer_dict) == 1 else tmp / len(self.filter_layer_dict)
        return E

    def forward(self, sst_list=None):
        E = self._filtered_table(sst_list)
        return torch.split(E, [self.n_users, self.n_items])

    def _aggr(self, x):
        code = ACT_CODES[self.act.lower()]
        lins = [m for m in self.aggr_layer if isinstance(m, nn.Linear)]
        x = _HipMLP.apply(x, None, code, 0.0, None, None, None, lins[0].weight, lins[0].bias)
        x = _HipMLP.apply(x, None, code, 0.0, None, None, None, lins[1].weight, lins[1].bias)
        return _HipMLP.apply(x, None, 0, 0.0, None, None, None, lins[2].weight, lins[2].bias)

    def _propagate(self, E):
        """H_1 = L E, H_2 = L H_1, ... (fairgo_pmf.py:198-201) and what the aggregation needs of them as whole tables:
        WAP: their mean; LBA: their concatenation (aggr_layer itself is row-wise: it runs on the batch's rows only --
        the same rows of the same function as the reference's whole-table pass); LVA: the layers themselves."""
        H, hs = E, []
        for _ in range(self.n_layers):
            H = SpMM.apply(H, self._L)
            hs.append(H)
        if self.n_layers == 1:
            return [hs[0]]
        if self.aggr_method == 'WAP':
            return [torch.stack(hs, dim=1).mean(dim=1)]
        if self.aggr_method == 'LBA':
            return [torch.cat(hs, dim=1)]
        return hs

    # --- frontier-restricted propagation (SURVEY.md section 7 hard part 3; fairgo_pmf.py:196-216) -------------------------
    FRONTIER_MIN_NNZ = 1 << 22      # `fairgo_frontier: auto`: graphs below this are cheaper to propagate whole (and captured)
    FRONTIER_MAX_SHARE = 0.5        # a frontier that covers more of the graph than this gains nothing: whole tables then

    def use_frontier(self) -> bool:
        """Whether a filter step propagates only the rows the batch can see.  Config `fairgo_frontier`: True / False / 'auto'
        (default: on for graphs of >= 4 M nonzeros).  Such a step has data-dependent shapes (the frontier of every batch is
        its own) and is launched eagerly: `step_capturable`."""
        mode = self.frontier_mode
        if isinstance(mode, str) and mode.lower() == 'auto':
            return int(self._norm_csr_host.nnz) >= self.FRONTIER_MIN_NNZ
        return bool(mode)

    def step_capturable(self, loss_name: str) -> bool:
        """Trainer hook (fairrec/trainer: `_graphed_step`): can a step on this loss function be captured as a hipGraph?"""
        return not (self.train_stage == 'finetune' and loss_name == 'calculate_loss' and self.use_frontier())

    def _frontier(self, user):
        """Row sets S_1 .. S_n (sorted, distinct, int32) with their inverse maps over the N graph rows (-1 elsewhere): the
        rows of H_l = L H_(l-1) that the batch's local embeddings depend on.  S_n = the batch's users; S_(l-1) = S_n plus the
        columns of the rows S_l (every layer's rows of the batch's users are aggregated, fairgo_pmf.py:204-216).  None when
        the frontier covers so much of the graph that whole-table products are as cheap."""
        L = self._L
        ip, col, _ = L.fwd
        N = L.shape[0]
        dev = ip.device
        if os.environ.get("FAIRREC_FRONTIER_TORCH") is not None:
            return self._frontier_torch(user)
        # A set is a bitmap over the graph rows (csrc/frontier.hip): mark the batch's users, OR in the columns of a set's rows,
        # and turn a bitmap into (ascending row ids, rank map) with one popcount launch, one cumulative sum and one scatter --
        # 5 launches per layer on 1.4 MB bitmaps where the torch form below ran ~15 over 11 M-entry maps and the 5.7 M
        # neighbour ids of a BASELINE configs[3] batch.  One host read per layer: the size of the set.
        lib, st = _C.lib(), _C.current_stream()
        eng = self.hip_engine()
        nw = (N + 31) // 32
        bits = torch.zeros(nw, dtype=torch.int32, device=dev)
        ids = user.to(dev, torch.int64).contiguous()
        _C.check(lib.fr_frontier_mark(ids.data_ptr(), ids.numel(), N, bits.data_ptr(), eng.err_flag.data_ptr(), st), "fr_frontier_mark")

        def listed(bits):
            cnt = torch.empty(nw, dtype=torch.int32, device=dev)
            _C.check(lib.fr_frontier_count(bits.data_ptr(), N, cnt.data_ptr(), st), "fr_frontier_count")
            incl = torch.cumsum(cnt, 0, dtype=torch.int32)
            n = int(incl[-1].item())
            rows = torch.empty(n, dtype=torch.int32, device=dev)
            pos = torch.empty(N, dtype=torch.int32, device=dev)
            _C.check(lib.fr_frontier_scatter(bits.data_ptr(), incl.data_ptr(), N, rows.data_ptr(), pos.data_ptr(), st),
                     "fr_frontier_scatter")
            return rows, pos

        out = []
        for l in range(self.n_layers, 0, -1):
            rows, pos = listed(bits)
            if l < self.n_layers and rows.numel() > self.FRONTIER_MAX_SHARE * N:
                return None
            out.append((rows, pos, bits))
            if l > 1:
                r64 = rows.to(torch.int64)
                tot = int((ip[r64 + 1] - ip[r64]).sum().item())
                if tot > self.FRONTIER_MAX_SHARE * col.numel():
                    return None
                bits = bits.clone()             # (the batch's users stay marked: S_(l-1) includes S_n)
                _C.check(lib.fr_frontier_expand(ip.data_ptr(), col.data_ptr(), rows.data_ptr(), rows.numel(), bits.data_ptr(), st),
                         "fr_frontier_expand")
        out.reverse()                           # S_1 first
        return out

    def _frontier_torch(self, user):
        """The same sets with stock torch ops (the form of round 4; `FAIRREC_FRONTIER_TORCH=1`, and what the tests hold the
        kernels against)."""
        L = self._L
        ip, col, _ = L.fwd
        N = L.shape[0]
        dev = ip.device
        # (distinct sorted ids through a flag per graph row and `nonzero`, not through torch.unique: a sort of the 5.7 M
        # neighbour ids of a BASELINE configs[3] batch costs several ms per step, marking 11 M flags a fraction of one)
        flags = torch.zeros(N, dtype=torch.bool, device=dev)
        flags[user.to(dev, torch.int64)] = True
        base = flags.nonzero().squeeze(1)
        sets, cur = [], base
        for l in range(self.n_layers, 0, -1):
            sets.append(cur)
            if l > 1:
                lo = ip[cur]
                deg = ip[cur + 1] - lo
                tot = int(deg.sum().item())
                if tot > self.FRONTIER_MAX_SHARE * col.numel():
                    return None
                start = torch.repeat_interleave(lo - (torch.cumsum(deg, 0) - deg), deg)
                nb = col[start + torch.arange(tot, device=dev)].to(torch.int64)
                flags[nb] = True            # (the batch's users stay marked: S_(l-1) includes S_n)
                cur = flags.nonzero().squeeze(1)
                if cur.numel() > self.FRONTIER_MAX_SHARE * N:
                    return None
        out = []
        for rows in reversed(sets):                      # S_1 first
            pos = torch.full((N,), -1, dtype=torch.int32, device=dev)
            pos[rows] = torch.arange(rows.numel(), dtype=torch.int32, device=dev)
            # bit c of the bitmap = (pos[c] >= 0): what the product kernels test before they touch the 4-byte map (the rows
            # are distinct, so the words are plain sums of distinct powers of two)
            bits = torch.zeros((N + 31) // 32, dtype=torch.int32, device=dev)
            bits.index_add_(0, rows >> 5, (torch.ones_like(rows) << (rows & 31)).to(torch.int32))
            out.append((rows.to(torch.int32), pos, bits))
        return out

    def _propagate_rows(self, E, user, fr=None, H1=None):
        """[H_1[user], ..., H_n[user]] ([B, D] each) through products over the frontier's rows only; None = not worth it.
        `fr`, `H1`: the frontier and its first layer's rows when the caller has them already (calculate_loss)."""
        fr = self._frontier(user) if fr is None else fr
        if fr is None:
            return None
        eng = self.hip_engine()
        H, prev, rows_out = E, (None, None, None), []
        for l, (rows, pos, bits) in enumerate(fr):
            H = H1 if (l == 0 and H1 is not None) else SpMMSel.apply(H, self._L, rows, pos, prev[0], prev[1], bits, prev[2])
            prev = (rows, pos, bits)
            rows_out.append(RowGather.apply(H, pos[user].to(torch.int64), eng.err_flag))
        return rows_out

    def _dis_terms(self, E, interaction, sst_list, props=None, node=None, frontier=None):
        """calculate_dis_loss, fairgo_pmf.py:190-238, on an already filtered whole table E.  `node` = E[user] when the caller
        gathered those rows already (the filter step's rating term does): one gather -- and one dense [N, D] gradient with its
        zero fill and scatter, 5.6 GB at BASELINE configs[3] -- instead of two."""
        eng = self.hip_engine()
        user = interaction[self.USER_ID].to(eng.device)
        if node is None:
            node = RowGather.apply(E, user, eng.err_flag)
        lva = self.aggr_method == 'LVA' and self.n_layers > 1
        layer_rows = None
        if frontier is not None:
            layer_rows = self._propagate_rows(E, user, *frontier)
        elif props is None and self.use_frontier() and not torch.cuda.is_current_stream_capturing():
            layer_rows = self._propagate_rows(E, user)
        if layer_rows is not None:
            # the same aggregation as _propagate's, row-wise on the batch's rows (mean / concatenation / per-layer weights act
            # on a row at a time, so the batch's rows of the aggregated table ARE the aggregate of the layers' batch rows)
            if lva:
                locals_ = layer_rows
            elif self.n_layers == 1:
                local = layer_rows[0]
            elif self.aggr_method == 'WAP':
                local = torch.stack(layer_rows, dim=1).mean(dim=1)
            else:
                local = torch.cat(layer_rows, dim=1)
        else:
            props = self._propagate(E) if props is None else props
            if lva:
                locals_ = [RowGather.apply(h, user, eng.err_flag) for h in props]
            else:
                local = RowGather.apply(props[0], user, eng.err_flag)
        if lva:
            if self.vs_weights.device != eng.device:       # once: no host-to-device copy inside a captured step
                self.vs_weights = self.vs_weights.to(eng.device)
            vs = self.vs_weights
        else:
            if self.aggr_method == 'LBA' and self.n_layers > 1:
                local = self._aggr(local)
        node_l, local_l = 0.0, 0.0
        for sst in sst_list:
            d = self.dis_layer_dict[sst]
            label = interaction[sst].to(eng.device)
            if self.sst_size[sst] == 2:
                node_l = node_l + SigmoidBce.apply(d(node), label.float())
                if lva:
                    for i in range(self.n_layers):
                        local_l = local_l + vs[i] * SigmoidBce.apply(d(locals_[i]), label.float())
                else:
                    local_l = local_l + SigmoidBce.apply(d(local), label.float())
            else:
                # the reference feeds sigmoid(logits) to CrossEntropy for the LOCAL term only (fairgo_pmf.py:233-235)
                node_l = node_l + SoftmaxCe.apply(d(node), label.long(), eng.err_flag)
                if lva:
                    for i in range(self.n_layers):
                        local_l = local_l + vs[i] * SoftmaxCe.apply(torch.sigmoid(d(locals_[i])), label.long(), eng.err_flag)
                else:
                    local_l = local_l + SoftmaxCe.apply(torch.sigmoid(d(local)), label.long(), eng.err_flag)
        return node_l + local_l

    def _filters_version(self):
        eng = self.hip_engine()
        return (self.train_stage, eng.group_version('filter'), eng.group_version('pretrain'))

    def begin_dis_phase(self, sst_list):
        """Trainer hook, called before a pass with optimizer_dis (trainer.py:893-896).  The filters and the tables do not
        move during that pass, so the filtered whole table and its `n_layers` graph propagations -- two whole-table MLP
        passes and `n_layers` SpMMs per STEP in the reference -- are the same tensors for every batch of the pass: they
        are computed once here, into buffers that keep their addresses (a step captured as a hipGraph reads them)."""
        key = tuple(sst_list)
        with torch.no_grad():
            E = self._filtered_table(list(sst_list))
            props = self._propagate(E)
        c = self._dis_cache.get(key)
        if c is None or c["E"].shape != E.shape:
            c = self._dis_cache[key] = {"E": E.clone(), "props": [h.clone() for h in props]}
        else:
            c["E"].copy_(E)
            for dst, src in zip(c["props"], props):
                dst.copy_(src)
        c["version"] = self._filters_version()

    def calculate_dis_loss(self, interaction, sst_list):
        """Discriminator phase: only the discriminators (+ aggr_layer) train, so the filters run without a gradient path."""
        c = self._dis_cache.get(tuple(sst_list))
        if c is not None and c["version"] == self._filters_version():
            return self._dis_terms(c["E"], interaction, sst_list, c["props"])
        with torch.no_grad():
            E = self._filtered_table(sst_list)
        return self._dis_terms(E, interaction, sst_list)

    def calculate_loss(self, interaction, sst_list=None):
        eng = self.hip_engine()
        user = interaction[self.USER_ID].to(eng.device)
        item = interaction[self.ITEM_ID].to(eng.device)
        rating = interaction[self.RATING].to(eng.device, torch.float32)
        B = user.numel()
        if self.train_stage != 'finetune':
            ue = eng.lookup("user_embedding_layer.weight", user)
            ie = eng.lookup("item_embedding_layer.weight", item)
            # padding_idx = 0: row 0 is zero and never receives a gradient (fairgo_pmf.py:60-61)
            ue = ue * (user != 0).unsqueeze(1)
            ie = ie * (item != 0).unsqueeze(1)
            return Mse.apply(RowDot.apply(ue, ie), rating)
        idx = torch.cat([user, item + self.n_users])
        fr = self._frontier(user) if self.use_frontier() and not torch.cuda.is_current_stream_capturing() else None
        # With a frontier the filtered table has ONE consumer below, which can hand the filter the gradient at its top layer's
        # pre-activation from its own two launches: the whole-table pass through act' (fr_act_bwd: 2.8 of 33.9 ms at BASELINE
        # configs[3]) is not run at all.
        fold = fr is not None and self._top_fold_ok(sst_list)
        E = self._filtered_table(sst_list, grad_at_z=fold)
        if fr is not None:
            # the filtered table's two uses -- the batch's rows and the first propagation layer -- as ONE autograd node, so
            # that dLoss/dE is written once (functional.GatherAndSpMMSel) instead of zero-filled, scattered into and added
            act = 0
            if fold:
                mlp = self.filter_layer_dict[(self.sst_attrs if sst_list is None else sst_list)[0]]
                act = ACT_CODES[mlp.activation.lower() if isinstance(mlp.activation, str) else mlp.activation]
            rows, H1 = GatherAndSpMMSel.apply(E, idx, eng.err_flag, self._L, fr[0][0], fr[0][1], fr[0][2], act)
            frontier = (fr, H1)
        else:
            rows, frontier = RowGather.apply(E, idx, eng.err_flag), None
        u_rows, i_rows = SplitRows.apply(rows, B)      # (autograd's slices: a zero fill, a copy and an add each on the way back)
        mse = Mse.apply(RowDot.apply(u_rows, i_rows), rating)
        # the reference's calculate_dis_loss runs forward() a second time (fairgo_pmf.py:205-206): the same values from the
        # same parameters, so ONE filtered table serves both terms -- its gradient is the sum of the two uses, which is
        # what the two backward passes through the filters add up to (linear in dLoss/dE; rounding-level difference)
        fair = self._dis_terms(E, interaction, sst_list, node=u_rows, frontier=frontier)
        return mse - self.fair_weight * fair

    def predict(self, interaction):
        eng = self.hip_engine()
        user = interaction[self.USER_ID].to(eng.device)
        item = interaction[self.ITEM_ID].to(eng.device)
        with torch.no_grad():
            E = self._filtered_table(None)
            rows = RowGather.apply(E, torch.cat([user, item + self.n_users]), eng.err_flag)
            B = user.numel()
            scores = RowDot.apply(rows[:B], rows[B:])
            return torch.clamp(scores, min=0., max=float(self.max_rating)) / float(self.max_rating)

    def full_sort_predict(self, interaction):
        user = interaction[self.USER_ID].to(self.hip_engine().device)
        with torch.no_grad():
            ua, ia = self.forward()
            pred = torch.matmul(ua[user], ia.transpose(0, 1))
            return torch.clamp(pred.view(-1), min=0., max=float(self.max_rating)) / float(self.max_rating)

    def get_sst_embed(self, user_data, sst_list=None):
        ret = {}
        idx = torch.arange(1, self.n_users)
        sst_list = self.sst_attrs if sst_list is None else sst_list
        for sst in sst_list:
            ret[sst] = user_data[sst][idx - 1]
        with torch.no_grad():
            ua, _ = self.forward()
        ret['embedding'] = ua[idx.to(ua.device)]
        return ret

    def state_dict(self, *args, **kwargs):
        if self._engine is not None:
            self._engine.flush()
        return super().state_dict(*args, **kwargs)
