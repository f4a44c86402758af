"""Collector for full-sort ranking evaluation on the device: the `full` branch of recbole/evaluator/collector.py:131-205.

Per user batch it receives the masked score matrix [users, n_items] (column 0 and the users' history already -inf,
trainer.py:435-437), the batch's positives and the batch Interaction, and keeps -- as device tensors -- what the metrics
need: `rec.topk` (hit flags of the top-max(topk) items | number of positives), `rec.positive_score`, `data.positive_i`
and `data.<sst>`.  The reference builds a dense [users, items] 0/1 matrix per batch to mark the positives; here the hit
flags come from binary searches in the sorted positive keys (fr_eval_hits).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .. import _C


class Collector:
    def __init__(self, config):
        self.config = config
        topk = config['topk'] or [10]
        self.topk = [topk] if isinstance(topk, int) else list(topk)
        self.sst = list(config['sst_attr_list'] or [])
        self.full = 'full' in (config['eval_args'] or {}).get('mode', 'full')
        self._parts: Dict[str, List[torch.Tensor]] = {}
        self._data: Dict[str, object] = {}

    def data_collect(self, train_data):
        """collector.py:80-97: what the exposure metrics need from the training data -- the catalogue size and how often
        every item occurs in training (a dense count vector here, a Counter in the reference)."""
        ds = train_data.dataset
        items = ds.inter_feat[ds.iid_field].to(torch.int64)
        self._data['data.num_items'] = int(ds.item_num)
        self._data['data.count_items'] = torch.bincount(items.reshape(-1), minlength=int(ds.item_num)).to(
            torch.device(self.config['device']))

    def _add(self, key, t):
        self._parts.setdefault(key, []).append(t)

    @staticmethod
    def _host_topk(dense: torch.Tensor, K: int) -> torch.Tensor:
        """`torch.topk(dense, K, dim=-1).indices` as the CPU backend orders EQUAL values (fr_topk_like_torch_cpu): the reference
        ranks on the host (collector.py:149), and where a list hangs on an exact tie -- an untrained scorer that clamps, a
        saturated sigmoid -- which candidates enter it and in which order is that kernel's doing (csrc/topk_host.hip).  `dense`:
        the tied users' rows of the reference's [users, n_items] matrix (float32, any device)."""
        rows = dense.to("cpu", torch.float32).contiguous()
        out = torch.empty((rows.shape[0], K), dtype=torch.int64)
        _C.check(_C.lib().fr_topk_like_torch_cpu(rows.data_ptr(), rows.shape[0], rows.shape[1], K, out.data_ptr(), None),
                 "fr_topk_like_torch_cpu")
        return out

    def eval_batch_collect(self, scores: torch.Tensor, interaction, positive_u: torch.Tensor, positive_i: torch.Tensor):
        lib = _C.lib()
        U, n_items = scores.shape
        K = min(max(self.topk), n_items)
        positive_u, positive_i = positive_u.to(scores.device, torch.int64), positive_i.to(scores.device, torch.int64)
        vals, topk_idx = torch.topk(scores, min(K + 1, n_items), dim=-1)
        topk_idx = topk_idx[:, :K].contiguous()
        # rows whose list is decided by an exact tie (inside the list, or at its end): ranked as the reference's host kernel does
        tied = (vals[:, 1:] == vals[:, :-1]).any(dim=1).nonzero().view(-1)
        if tied.numel():
            topk_idx[tied] = self._host_topk(scores[tied], K).to(scores.device)
        keys = torch.sort(positive_u * n_items + positive_i).values
        rec = torch.empty((U, K + 1), dtype=torch.int32, device=scores.device)
        _C.check(lib.fr_eval_hits(topk_idx.data_ptr(), U, K, n_items, keys.data_ptr(), keys.numel(), rec.data_ptr(),
                                  _C.current_stream()), "fr_eval_hits")
        self._add('rec.topk', rec)
        self._add('rec.items', topk_idx)
        self._add('rec.positive_score', scores[positive_u, positive_i])
        self._add('data.positive_i', positive_i)
        for s in self.sst:
            if s in interaction:
                self._add('data.' + s, interaction[s].to(scores.device)[positive_u])

    def eval_batch_collect_candidates(self, origin_scores: torch.Tensor, row_idx: torch.Tensor, interaction,
                                      positive_u: torch.Tensor, positive_i: torch.Tensor, n_items: int):
        """The negative-sampling (`uni100`) branch (collector.py:131-205 with full == False, fed by
        trainer.py:440-456): `origin_scores[r]` is the prediction of row r of `interaction` (user row `row_idx[r]` of the
        batch, item `interaction[item][r]`); a user's candidates are its positives and sampled negatives.  The reference
        scatters them into a dense [users, n_items] -inf matrix; here a (row, item) -> score map is a sorted key array,
        the ranking is a sort over the distinct candidates, and every lookup the reference makes in the dense matrix
        -- including the ones that land on -inf -- is a binary search (missing = -inf).  `rec.negative_score` /
        `data.negative_i` / `data.<sst>` follow the reference's row arithmetic literally: rows [P, 2P) of the batch and
        the first P rows, P = number of positives in the batch."""
        lib = _C.lib()
        dev = origin_scores.device
        iid = self.config['ITEM_ID_FIELD']
        items = interaction[iid].to(dev, torch.int64)
        row_idx = row_idx.to(dev, torch.int64)
        positive_u, positive_i = positive_u.to(dev, torch.int64), positive_i.to(dev, torch.int64)
        U = int(positive_u[-1].item()) + 1                                    # batch_user_num, trainer.py:453
        K = max(self.topk)
        if K <= 62 and (row_idx.numel() < 2 or bool((row_idx[1:] >= row_idx[:-1]).all())):
            # The evaluation loaders emit a batch user by user: a user's candidates are a contiguous segment, and the whole
            # ranking is ONE launch, a wave per user (fr_eval_topk_segments); every lookup the reference makes in its dense
            # -inf matrix is a scan of the user's segment (fr_eval_lookup_segments).  No sort of the batch's candidates.
            items = items.contiguous()
            sc = origin_scores.reshape(-1).to(torch.float32).contiguous()
            seg = torch.searchsorted(row_idx, torch.arange(U + 1, device=dev, dtype=torch.int64)).contiguous()
            topk_idx = torch.empty((U, K), dtype=torch.int64, device=dev)
            flags = torch.empty(U, dtype=torch.int32, device=dev)
            st = _C.current_stream()
            _C.check(lib.fr_eval_topk_segments(seg.data_ptr(), U, items.data_ptr(), sc.data_ptr(), K, topk_idx.data_ptr(),
                                               flags.data_ptr(), st), "fr_eval_topk_segments")
            # users whose list hangs on an exact tie (or who have fewer than K + 1 distinct candidates): their rows of the reference's
            # dense -inf matrix are ranked on the host, in torch.topk's CPU order (csrc/topk_host.hip)
            tied = flags.nonzero().view(-1)
            if tied.numel():
                slot = torch.full((U,), -1, dtype=torch.int64, device=dev)
                slot[tied] = torch.arange(tied.numel(), device=dev)
                sel = slot[row_idx] >= 0
                dense = torch.full((tied.numel(), n_items), -float('inf'), dtype=torch.float32, device=dev)
                dense[slot[row_idx[sel]], items[sel]] = sc[sel]
                topk_idx[tied] = self._host_topk(dense, K).to(dev)

            def lookup(rows, its):
                rows, its = rows.contiguous(), its.contiguous()
                out = torch.empty(rows.numel(), dtype=torch.float32, device=dev)
                _C.check(lib.fr_eval_lookup_segments(seg.data_ptr(), U, items.data_ptr(), sc.data_ptr(), rows.data_ptr(),
                                                     its.data_ptr(), rows.numel(), out.data_ptr(), st), "fr_eval_lookup_segments")
                return out.to(origin_scores.dtype)
            return self._finish_candidates(topk_idx, lookup, interaction, items, positive_u, positive_i, n_items, U, K, dev)
        keys, order = torch.sort(row_idx * n_items + items, stable=True)
        first = torch.ones_like(keys, dtype=torch.bool)
        first[1:] = keys[1:] != keys[:-1]
        ckeys, cscore = keys[first], origin_scores.view(-1)[order][first]     # distinct candidates (equal pairs, equal score)

        def lookup(rows, its):
            q = rows * n_items + its
            p = torch.searchsorted(ckeys, q).clamp_(max=ckeys.numel() - 1)
            return torch.where(ckeys[p] == q, cscore[p], torch.full_like(cscore[p], -float('inf')))

        # (the general form, for a batch whose rows are not grouped by user, or K > 62)
        # ranking among a user's candidates: stable sort by score (descending) inside each row
        o1 = torch.sort(cscore, descending=True, stable=True).indices
        o2 = torch.sort((ckeys // n_items)[o1], stable=True).indices
        ranked = o1[o2]                                                        # candidates by (row asc, score desc)
        rrow = (ckeys // n_items)[ranked]
        start = torch.searchsorted(rrow, torch.arange(U, device=dev))
        rank = torch.arange(ranked.numel(), device=dev) - start[rrow]
        topk_idx = torch.zeros((U, K), dtype=torch.int64, device=dev)          # short lists are padded with [PAD] item 0
        keep = rank < K
        topk_idx[rrow[keep], rank[keep]] = (ckeys % n_items)[ranked][keep]
        # users whose list is decided by an exact tie between candidates (ranks j - 1 and j equal, j <= K), or who have fewer
        # than K candidates (the rest of the list is then the host kernel's pick among the -inf entries): their rows of the
        # reference's dense -inf matrix are ranked on the host, in torch.topk's CPU order
        sr = cscore[ranked]
        pair = (rrow[1:] == rrow[:-1]) & (sr[1:] == sr[:-1]) & (rank[1:] <= K)
        cnt = torch.bincount(rrow, minlength=U)
        tied = torch.unique(torch.cat([rrow[1:][pair], (cnt < min(K + 1, n_items)).nonzero().view(-1)]))
        if tied.numel():
            slot = torch.full((U,), -1, dtype=torch.int64, device=dev)
            slot[tied] = torch.arange(tied.numel(), device=dev)
            crow = ckeys // n_items
            sel = slot[crow] >= 0
            dense = torch.full((tied.numel(), n_items), -float('inf'), dtype=torch.float32, device=dev)
            dense[slot[crow[sel]], (ckeys % n_items)[sel]] = cscore[sel].to(torch.float32)
            topk_idx[tied] = self._host_topk(dense, K).to(dev)
        return self._finish_candidates(topk_idx, lookup, interaction, items, positive_u, positive_i, n_items, U, K, dev)

    def _finish_candidates(self, topk_idx, lookup, interaction, items, positive_u, positive_i, n_items, U, K, dev):
        lib = _C.lib()
        pos_keys = torch.sort(positive_u * n_items + positive_i).values
        rec = torch.empty((U, K + 1), dtype=torch.int32, device=dev)
        _C.check(lib.fr_eval_hits(topk_idx.data_ptr(), U, K, n_items, pos_keys.data_ptr(), pos_keys.numel(), rec.data_ptr(),
                                  _C.current_stream()), "fr_eval_hits")
        P = positive_u.numel()
        self._add('rec.topk', rec)
        self._add('rec.items', topk_idx)
        self._add('rec.positive_score', lookup(positive_u, positive_i))
        self._add('data.positive_i', positive_i)
        neg_items = items[P:2 * P]
        self._add('rec.negative_score', lookup(positive_u[:neg_items.numel()], neg_items))
        self._add('data.negative_i', neg_items)
        for s in self.sst:
            if s in interaction:
                self._add('data.' + s, interaction[s].to(dev)[:P])

    def get_data_struct(self) -> Dict[str, torch.Tensor]:
        out = {k: torch.cat(v, dim=0) for k, v in self._parts.items()}
        out.update(self._data)
        self._parts = {}
        return out
