"""Evaluation metrics on the device with the result keys of recbole/evaluator/metrics.py:
'hit@k', 'mrr@k', 'ndcg@k', 'recall@k', 'precision@k' and '<Name> Unfairness of sensitive attribute <sst>' /
'Differential Fairness of sensitive attribute <sst>', rounded to `metric_decimal_place`.

Inputs are the arrays the reference's Collector gathers (collector.py:131-205), as device tensors:
  rec_topk       [users, max(topk) + 1]  hit flags of the ranked list | number of positives
  pos_score, pos_i                        score and item id of every positive (user, item) pair
  neg_score, neg_i                        (uni100-style modes) the first len(pos) negatives of the batch order
  sst[<name>]                             the user's attribute value per positive pair
The torch ops here are index plumbing (sort / unique of the id columns); every reduction is a HIP kernel
(csrc/metrics.hip: fr_topk_metrics, fr_group_sums, fr_fair_metrics_from_stats).
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch

from .. import _C


def topk_metrics(rec_topk: torch.Tensor, topk: Iterable[int]) -> Dict[str, float]:
    lib = _C.lib()
    rec = rec_topk.to(torch.int32).contiguous()
    U, K = rec.shape[0], rec.shape[1] - 1
    out = torch.empty(6 * K, dtype=torch.float64, device=rec.device)
    ws = torch.empty(lib.fr_topk_metrics_workspace_bytes(U, K), dtype=torch.uint8, device=rec.device)
    _C.check(lib.fr_topk_metrics(rec.data_ptr(), U, K, out.data_ptr(), ws.data_ptr(), ws.numel(), _C.current_stream()),
             "fr_topk_metrics")
    vals = out.view(6, K).cpu()
    res = {}
    for m, name in enumerate(("hit", "mrr", "ndcg", "recall", "precision", "map")):
        for k in topk:
            res[f"{name}@{k}"] = float(vals[m, k - 1])
    return res


def _group_index(sst: torch.Tensor):
    """np.unique(sst_value, return_inverse=True): group index = rank of the value among the values present."""
    vals, inv = torch.unique(sst, sorted=True, return_inverse=True)
    return int(vals.numel()), inv.to(torch.int32).contiguous()


def _group_sums(items: Optional[torch.Tensor], group, value, wtrue, G):
    """stats [K, G, 3] over the distinct items (items = None: one segment)."""
    lib = _C.lib()
    dev = value.device
    n = value.numel()
    if items is None:
        perm, seg_start, K = None, torch.tensor([0, n], dtype=torch.int64, device=dev), 1
    else:
        keys, perm = torch.sort(items, stable=True)
        _, counts = torch.unique_consecutive(keys, return_counts=True)
        K = counts.numel()
        seg_start = torch.zeros(K + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=seg_start[1:])
    stats = torch.empty((K, G, 3), dtype=torch.float64, device=dev)
    _C.check(lib.fr_group_sums(_C.ptr(perm), seg_start.data_ptr(), K, group.data_ptr(), value.contiguous().data_ptr(),
                               _C.ptr(wtrue), G, stats.data_ptr(), _C.current_stream()), "fr_group_sums")
    return stats, K


def _from_stats(stats, K, G):
    lib = _C.lib()
    out = torch.empty(5, dtype=torch.float64, device=stats.device)
    ws = torch.empty(lib.fr_fair_metrics_workspace_bytes(K), dtype=torch.uint8, device=stats.device)
    _C.check(lib.fr_fair_metrics_from_stats(stats.data_ptr(), K, G, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                            _C.current_stream()), "fr_fair_metrics_from_stats")
    return out.cpu()


def fairness_metrics(pos_score, pos_i, sst: Dict[str, torch.Tensor], neg_score=None, neg_i=None, mode="full",
                     value_type=True) -> Dict[str, float]:
    """NonParity + DifferentialFairness per attribute; Value / Absolute / Under / Over unfairness on the FIRST attribute
    (as the reference's classes read `sst_attr_list[0]`, metrics.py:905)."""
    res = {}
    pos_score = pos_score.to(torch.float32).contiguous()
    first = True
    for name, col in sst.items():
        G, gidx = _group_index(col)
        if G < 2:
            raise ValueError(f'there is only one value for {name} sensitive attribute')
        # NonParity (metrics.py:864-882): |difference| of the two group means, population std for more groups
        st, _ = _group_sums(None, gidx, pos_score, None, G)
        means = (st[0, :, 0] / st[0, :, 1]).cpu()
        res[f'NonParity Unfairness of sensitive attribute {name}'] = float(
            (means[0] - means[1]).abs() if G == 2 else means.std(unbiased=False))
        st, K = _group_sums(pos_i, gidx, pos_score, None, G)
        res[f'Differential Fairness of sensitive attribute {name}'] = float(_from_stats(st, K, G)[4])
        if first and value_type:
            if G != 2:
                raise ValueError('sensitive attribute must be binary')
            P = pos_score.numel()
            if mode != 'full' and neg_i is not None:
                # the j-th negative is attributed to the group of the j-th positive's user (metrics.py:957-960)
                m = min(neg_i.numel(), P)
                items = torch.cat([pos_i, neg_i[:m]])
                value = torch.cat([pos_score, neg_score[:m].to(torch.float32)])
                group = torch.cat([gidx, gidx[:m]])
                wtrue = torch.cat([torch.ones(P, device=value.device), torch.zeros(m, device=value.device)])
            else:
                items, value, group, wtrue = pos_i, pos_score, gidx, torch.ones(P, device=pos_score.device)
            st, K = _group_sums(items, group.contiguous(), value.contiguous(), wtrue.contiguous(), 2)
            v = _from_stats(st, K, 2)
            for q, label in enumerate(("Value", "Absolute", "Underestimation", "Overestimation")):
                res[f'{label} Unfairness of sensitive attribute {name}'] = float(v[q])
        first = False
    return res


def gini_index(rec_items: torch.Tensor, num_items: int, topk) -> Dict[str, float]:
    """GiniIndex@k (metrics.py:638-662) of the item exposure in the top-k lists: counts by bincount, no host loop."""
    res = {}
    for k in topk:
        cnt = torch.bincount(rec_items[:, :k].reshape(-1), minlength=num_items)
        cnt = torch.sort(cnt[cnt > 0]).values.to(torch.float64)
        idx = torch.arange(num_items - cnt.numel() + 1, num_items + 1, device=cnt.device, dtype=torch.float64)
        res[f'giniindex@{k}'] = float(((2 * idx - num_items - 1) * cnt).sum() / (rec_items.shape[0] * k) / num_items)
    return res


def popularity_percentage(rec_items: torch.Tensor, count_items: torch.Tensor, topk, popularity_ratio=None) -> Dict[str, float]:
    """PopularityPercentage@k (metrics.py:749-821): popular = top `ratio` fraction of the items that occur in training,
    ordered by (count, id) descending, or the items with count >= ratio when ratio > 1."""
    ratio = 0.1 if popularity_ratio is None or popularity_ratio <= 0 else popularity_ratio
    present = torch.nonzero(count_items > 0).view(-1)
    if ratio > 1:
        popular = count_items >= ratio
    else:
        cnt = count_items[present]
        order = torch.sort(cnt * (count_items.numel() + 1) + present, descending=True).indices   # (count, id) descending
        popular = torch.zeros_like(count_items, dtype=torch.bool)
        popular[present[order[:max(int(present.numel() * ratio), 1)]]] = True
    hit = popular[rec_items].to(torch.float64)
    avg = (hit.cumsum(dim=1) / torch.arange(1, hit.shape[1] + 1, device=hit.device, dtype=torch.float64)).mean(dim=0).cpu()
    return {f'popularitypercentage@{k}': float(avg[k - 1]) for k in topk}


def _mean_at_k(values: torch.Tensor, name: str, topk) -> Dict[str, float]:
    """mean over users of cumsum(values)[:k] / k (the `metric_info` / `topk_result` pair of the exposure metrics)."""
    v = values.to(torch.float64)
    avg = (v.cumsum(dim=1) / torch.arange(1, v.shape[1] + 1, device=v.device, dtype=torch.float64)).mean(dim=0).cpu()
    return {f'{name}@{k}': float(avg[k - 1]) for k in topk}


def item_coverage(rec_items, num_items, topk) -> Dict[str, float]:
    """ItemCoverage@k (metrics.py:438-482): distinct recommended items / catalogue size."""
    return {f'itemcoverage@{k}': float(torch.unique(rec_items[:, :k]).numel() / num_items) for k in topk}


def average_popularity(rec_items, count_items, topk) -> Dict[str, float]:
    """AveragePopularity@k (metrics.py:484-551): mean training popularity of the recommended items."""
    return _mean_at_k(count_items[rec_items], 'averagepopularity', topk)


def shannon_entropy(rec_items, topk) -> Dict[str, float]:
    """ShannonEntropy@k (metrics.py:553-606): -sum p log p over the recommended items, divided by their number (as the
    reference does)."""
    res = {}
    for k in topk:
        cnt = torch.unique(rec_items[:, :k], return_counts=True)[1].to(torch.float64)
        p = cnt / (rec_items.shape[0] * k)
        res[f'shannonentropy@{k}'] = float((-p * torch.log(p)).sum() / cnt.numel())
    return res


def tail_percentage(rec_items, count_items, topk, tail_ratio=None) -> Dict[str, float]:
    """TailPercentage@k (metrics.py:664-747): share of long-tail items; tail = the bottom `ratio` fraction of the items
    that occur in training ordered by (count, id) ascending, or the items with count <= ratio when ratio > 1."""
    ratio = 0.1 if tail_ratio is None or tail_ratio <= 0 else tail_ratio
    present = torch.nonzero(count_items > 0).view(-1)
    tail = torch.zeros_like(count_items, dtype=torch.bool)
    if ratio > 1:
        tail[present[count_items[present] <= ratio]] = True
    else:
        order = torch.sort(count_items[present] * (count_items.numel() + 1) + present).indices       # (count, id) ascending
        tail[present[order[:max(int(present.numel() * ratio), 1)]]] = True
    return _mean_at_k(tail[rec_items], 'tailpercentage', topk)


class Evaluator:
    """recbole/evaluator/evaluator.py: metric names from `config['metrics']` -> one result dict."""

    TOPK = {"hit", "mrr", "ndcg", "recall", "precision", "map"}
    EXPOSURE = {"giniindex", "popularitypercentage", "itemcoverage", "averagepopularity", "shannonentropy", "tailpercentage"}
    FAIR = {"nonparityunfairness", "valueunfairness", "absoluteunfairness", "underunfairness", "overunfairness",
            "differentialfairness"}

    def __init__(self, config):
        self.config = config
        self.metrics = [m.lower() for m in (config['metrics'] or [])]
        self.topk = config['topk'] or [10]
        if isinstance(self.topk, int):
            self.topk = [self.topk]
        self.decimal_place = config['metric_decimal_place'] if config['metric_decimal_place'] is not None else 4
        self.mode = (config['eval_args'] or {}).get('mode', 'full')
        unknown = [m for m in self.metrics if m not in self.TOPK | self.FAIR | self.EXPOSURE]
        if unknown:
            raise NotImplementedError(f'metrics {unknown} are not on the device path')

    def evaluate(self, collected: Dict[str, torch.Tensor]) -> Dict[str, float]:
        res = {}
        if self.TOPK & set(self.metrics):
            allk = topk_metrics(collected['rec.topk'], self.topk)
            res.update({k: v for k, v in allk.items() if k.split('@')[0] in self.metrics})
        if "giniindex" in self.metrics:
            res.update(gini_index(collected['rec.items'], collected['data.num_items'], self.topk))
        if "popularitypercentage" in self.metrics:
            res.update(popularity_percentage(collected['rec.items'], collected['data.count_items'], self.topk,
                                             self.config['popularity_ratio']))
        if "itemcoverage" in self.metrics:
            res.update(item_coverage(collected['rec.items'], collected['data.num_items'], self.topk))
        if "averagepopularity" in self.metrics:
            res.update(average_popularity(collected['rec.items'], collected['data.count_items'], self.topk))
        if "shannonentropy" in self.metrics:
            res.update(shannon_entropy(collected['rec.items'], self.topk))
        if "tailpercentage" in self.metrics:
            res.update(tail_percentage(collected['rec.items'], collected['data.count_items'], self.topk,
                                       self.config['tail_ratio']))
        if self.FAIR & set(self.metrics):
            sst = {s: collected['data.' + s] for s in self.config['sst_attr_list']}
            fair = fairness_metrics(collected['rec.positive_score'], collected['data.positive_i'], sst,
                                    collected.get('rec.negative_score'), collected.get('data.negative_i'), self.mode,
                                    value_type=bool({"valueunfairness", "absoluteunfairness", "underunfairness",
                                                     "overunfairness"} & set(self.metrics)))
            want = {"nonparityunfairness": "NonParity", "valueunfairness": "Value Unfairness",
                    "absoluteunfairness": "Absolute", "underunfairness": "Underestimation",
                    "overunfairness": "Overestimation", "differentialfairness": "Differential"}
            for k, v in fair.items():
                if any(k.startswith(p) for m, p in want.items() if m in self.metrics):
                    res[k] = v
        return {k: round(v, self.decimal_place) for k, v in res.items()}
