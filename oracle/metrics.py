"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

numpy restatement of the reference's evaluation metrics on collected arrays (recbole/evaluator/metrics.py):
Hit :40-66, MRR :68-98, Recall :140-162, NDCG :164-204, Precision :206-232 (all on the `rec.topk` matrix
[users, max(topk) + 1] = hit flags of the ranked list | number of positives), and the fairness metrics
NonParityUnfairness :823-882, Value/Absolute/Under/OverUnfairness :884-1267, DifferentialFairness :1269-1342.
Vectorised; pinned by tests/test_oracle_metrics.py against golden vectors produced by the reference's own classes
(tests/golden/gen_metrics_golden.py).
"""
import numpy as np


def topk_metrics(rec_topk, topk):
    """{'hit@k', 'mrr@k', 'ndcg@k', 'recall@k', 'precision@k', 'map@k'} means over users (base_metric.py:61-77)."""
    rec_topk = np.asarray(rec_topk)
    pos = rec_topk[:, :-1].astype(bool)
    pos_len = rec_topk[:, -1].astype(np.int64)
    U, K = pos.shape
    csum = np.cumsum(pos, axis=1)
    ranks = np.arange(1, K + 1)
    hit = (csum > 0).astype(np.float64)
    first = pos.argmax(axis=1)
    has = pos[np.arange(U), first]
    mrr = np.where((ranks[None, :] > first[:, None]) & has[:, None], 1.0 / (first[:, None] + 1), 0.0)
    recall = csum / pos_len[:, None]
    precision = csum / ranks[None, :]
    disc = 1.0 / np.log2(ranks + 1.0)
    dcg = np.cumsum(np.where(pos, disc[None, :], 0.0), axis=1)
    idcg_all = np.cumsum(disc)
    idcg_len = np.minimum(pos_len, K)
    idcg = idcg_all[np.minimum(ranks[None, :], idcg_len[:, None]) - 1]
    ndcg = dcg / idcg
    # MAP, metrics.py:127-139: sum_{j <= k, hit j} Precision@j over min(k, min(|positives|, K)) (all K when there are none)
    sum_pre = np.cumsum(precision * pos, axis=1)
    lens = np.minimum(pos_len, K)
    denom = np.where(lens[:, None] > 0, np.minimum(ranks[None, :], np.maximum(lens, 1)[:, None]), K)
    ap = sum_pre / denom
    out = {}
    for name, val in (("hit", hit), ("mrr", mrr), ("ndcg", ndcg), ("recall", recall), ("precision", precision), ("map", ap)):
        avg = val.mean(axis=0)
        for k in topk:
            out[f"{name}@{k}"] = float(avg[k - 1])
    return out


def nonparity(score, sst_value):
    """metrics.py:864-882: |mean_g0 - mean_g1| for a binary attribute, std of the group means otherwise."""
    vals = np.unique(sst_value)
    if len(vals) < 2:
        raise ValueError("there is only one value for the sensitive attribute")
    means = [np.mean(score[sst_value == s]) for s in vals]
    return float(np.abs(means[0] - means[1])) if len(vals) == 2 else float(np.std(means))


def _item_group_tables(pos_score, pos_iids, neg_score, neg_iids, sst_value):
    """The [items, 2] tables of metrics.py:948-975: predicted and true means per (item, group); with negatives the j-th
    negative is attributed to the group of the j-th positive's user (zip with sst_indices from the start)."""
    vals, sst_idx = np.unique(sst_value, return_inverse=True)
    if len(vals) != 2:
        raise ValueError("sensitive attribute must be binary")
    full = neg_iids is None
    items = pos_iids if full else np.concatenate((pos_iids, neg_iids))
    _, iid_idx = np.unique(items, return_inverse=True)
    K, P = iid_idx.max() + 1, len(pos_iids)
    pred, num, true = np.zeros((K, 2)), np.zeros((K, 2)), np.zeros((K, 2))
    np.add.at(pred, (iid_idx[:P], sst_idx), pos_score)
    np.add.at(num, (iid_idx[:P], sst_idx), 1.0)
    np.add.at(true, (iid_idx[:P], sst_idx), 1.0)
    if not full:
        m = min(len(neg_iids), len(sst_idx))
        np.add.at(pred, (iid_idx[P:P + m], sst_idx[:m]), neg_score[:m])
        np.add.at(num, (iid_idx[P:P + m], sst_idx[:m]), 1.0)
    num += 1e-5
    return pred / num, true / num


def value_type_unfairness(kind, pos_score, pos_iids, neg_score, neg_iids, sst_value):
    pred, true = _item_group_tables(pos_score, pos_iids, neg_score, neg_iids, sst_value)
    if kind == "value":
        d = pred - true
    elif kind == "absolute":
        d = np.abs(pred - true)
    elif kind == "under":
        d = np.where(true - pred > 0, true - pred, 0)
    elif kind == "over":
        d = np.where(pred - true > 0, pred - true, 0)
    else:
        raise ValueError(kind)
    return float(np.mean(np.abs(d[:, 0] - d[:, 1])))


def differential_fairness(score, iids, sst_value):
    """metrics.py:1311-1342 (float32 tables, concentration parameter 1, alpha = 1 / #items)."""
    vals, sst_idx = np.unique(sst_value, return_inverse=True)
    _, iid_idx = np.unique(iids, return_inverse=True)
    K, G = iid_idx.max() + 1, len(vals)
    s, c = np.zeros((K, G)), np.zeros((K, G))
    np.add.at(s, (iid_idx, sst_idx), score.astype(np.float64))
    np.add.at(c, (iid_idx, sst_idx), 1.0)
    table = ((s + 1.0 / K) / (c + 1.0)).astype(np.float32)
    eps = np.zeros(K, dtype=np.float32)
    for i in range(G):
        for j in range(i + 1, G):
            eps = np.maximum(eps, np.abs(np.log(table[:, i]) - np.log(table[:, j])))
    return float(eps.mean())


def gini_index(rec_items, num_items, topk):
    """metrics.py:638-662: Gini index of the item exposure in the top-k lists."""
    out = {}
    for k in topk:
        cnt = np.sort(np.unique(np.asarray(rec_items)[:, :k].ravel(), return_counts=True)[1])
        idx = np.arange(num_items - len(cnt) + 1, num_items + 1)
        total = rec_items.shape[0] * k
        out[f"giniindex@{k}"] = float(np.sum((2 * idx - num_items - 1) * cnt) / total / num_items)
    return out


def popularity_percentage(rec_items, train_items, topk, popularity_ratio=None):
    """metrics.py:749-821: share of "popular" items in the top-k lists; popular = the top `ratio` fraction of the items
    that occur in training ordered by (count, id) descending (ratio <= 1) or the items with count >= ratio (> 1)."""
    ratio = 0.1 if popularity_ratio is None or popularity_ratio <= 0 else popularity_ratio
    items, cnt = np.unique(train_items, return_counts=True)
    if ratio > 1:
        popular = items[cnt >= ratio]
    else:
        order = np.lexsort((items, cnt))[::-1]
        popular = items[order[:max(int(len(items) * ratio), 1)]]
    hit = np.isin(np.asarray(rec_items), popular)
    avg = (hit.cumsum(axis=1) / np.arange(1, hit.shape[1] + 1)).mean(axis=0)
    return {f"popularitypercentage@{k}": float(avg[k - 1]) for k in topk}


def all_metrics(z, topk, mode, sst_attrs):
    """Every metric of a golden case, with the reference's result keys."""
    out = topk_metrics(z["rec_topk"], topk)
    neg_s, neg_i = (None, None) if mode == "full" else (z["neg_score"], z["neg_i"])
    for sst in sst_attrs:
        out[f"NonParity Unfairness of sensitive attribute {sst}"] = nonparity(z["pos_score"], z[sst])
        out[f"Differential Fairness of sensitive attribute {sst}"] = differential_fairness(z["pos_score"], z["pos_i"], z[sst])
    if len(sst_attrs) == 1:
        s = sst_attrs[0]
        for kind, label in (("value", "Value"), ("absolute", "Absolute"), ("under", "Underestimation"),
                            ("over", "Overestimation")):
            out[f"{label} Unfairness of sensitive attribute {s}"] = value_type_unfairness(
                kind, z["pos_score"], z["pos_i"], neg_s, neg_i, z[s])
    return out
