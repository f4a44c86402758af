"""GPU: full-sort ranking evaluation on the device -- Collector + Evaluator against golden vectors produced by the
reference's own Collector / Evaluator, the full-sort evaluation loader against a brute-force construction, and the
whole thing through run_recbole."""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN, "collector_full_*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[10:-4] for p in CASES])
def test_collector_and_evaluator_match_reference_golden(path):
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.evaluator import Collector, Evaluator
    z = np.load(path)
    ratio, tail = float(z["popularity_ratio"]), float(z["tail_ratio"])
    cfg = Config(config_dict={"metrics": [str(m) for m in z["metrics"]], "topk": [int(k) for k in z["topk"]],
                              "metric_decimal_place": 10, "sst_attr_list": ["gender"], "eval_args": {"mode": "full"},
                              "device": "cuda", "popularity_ratio": None if ratio < 0 else ratio,
                              "tail_ratio": None if tail < 0 else tail})
    col, ev = Collector(cfg), Evaluator(cfg)
    d = lambda k: torch.from_numpy(z[k]).cuda()
    # what Collector.data_collect(train_data) provides: catalogue size and the items' training popularity
    col._data["data.num_items"] = int(z["n_items"])
    col._data["data.count_items"] = torch.bincount(d("train_items"), minlength=int(z["n_items"]))
    for b in range(int(z["n_batches"])):
        scores = d(f"scores{b}")
        scores[:, 0] = -float("inf")
        scores[d(f"hist_u{b}"), d(f"hist_i{b}")] = -float("inf")
        inter = Interaction({"user_id": d(f"users{b}"), "gender": d(f"gender{b}")})
        col.eval_batch_collect(scores, inter, d(f"pos_u{b}"), d(f"pos_i{b}"))
    struct = col.get_data_struct()
    for key in ("rec.topk", "rec.items", "rec.positive_score", "data.positive_i", "data.gender"):   # collected by the reference only when a
        if "collected." + key in z.files:                                     # registered metric needs them
            np.testing.assert_array_equal(struct[key].cpu().numpy(), z["collected." + key])
    ref = json.loads(str(z["result_json"]))
    got = ev.evaluate(struct)
    assert set(got) == set(ref)
    for k, v in ref.items():
        tol = 5e-6 if "Differential" in k or "NonParity" in k else 1e-8
        assert abs(got[k] - v) <= tol * max(1.0, abs(v)), (k, got[k], v)


def test_full_sort_eval_loader_against_brute_force():
    from fairrec.config import Config
    from fairrec.data.dataloader import FullSortEvalDataLoader
    from fairrec.data.dataset import InteractionDataset
    from fairrec.data.interaction import Interaction
    from fairrec.sampler import Sampler
    rng = np.random.default_rng(0)
    n_users, n_items = 30, 20
    cfg = Config(config_dict={"eval_batch_size": 7 * n_items, "device": "cuda", "eval_args": {"mode": "full"}})
    users = Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(rng.integers(0, 2, n_users).astype(np.float32))})

    def ds(n):
        return InteractionDataset(cfg, Interaction({"user_id": torch.from_numpy(rng.integers(1, n_users, n)),
                                                    "item_id": torch.from_numpy(rng.integers(1, n_items, n))}),
                                  users, n_users, n_items)
    train, test = ds(150), ds(60)
    sampler = Sampler(["train", "test"], [train, test], device="cuda").set_phase("test")
    dl = FullSortEvalDataLoader(cfg, test, sampler)
    assert dl.step == 7
    tr = set(zip(train.inter_feat["user_id"].tolist(), train.inter_feat["item_id"].tolist()))
    te = set(zip(test.inter_feat["user_id"].tolist(), test.inter_feat["item_id"].tolist()))
    seen_users = []
    for user_df, (hu, hi), pu, pi in dl:
        uids = user_df["user_id"].tolist()
        seen_users += uids
        assert user_df["gender"].is_cuda and len(uids) <= 7
        got_pos = set((uids[r], it) for r, it in zip(pu.tolist(), pi.tolist()))
        got_hist = set((uids[r], it) for r, it in zip(hu.tolist(), hi.tolist()))
        assert got_pos == {p for p in te if p[0] in uids}
        assert got_hist == {p for p in tr if p[0] in uids} - te       # used in an earlier phase, not a positive now
    assert seen_users == sorted({u for u, _ in te})


def test_run_recbole_full_sort_evaluation(tmp_path):
    """FOCF (has full_sort_predict) and NFCF (scored pair by pair through predict) end to end with ranking + fairness
    metrics; PFCN_PMF with filters: one result per attribute subset."""
    from fairrec.quick_start import run_recbole
    common = {"epochs": 1, "train_batch_size": 512, "synthetic_users": 120, "synthetic_items": 80,
              "synthetic_interactions": 3000, "device": "cuda", "checkpoint_dir": str(tmp_path), "embedding_size": 16,
              "eval_args": {"mode": "full"}, "topk": [5, 10], "valid_metric": "ndcg@10", "valid_metric_bigger": True,
              "metrics": ["NDCG", "Recall", "Hit", "MRR", "DifferentialFairness", "ValueUnfairness", "NonParityUnfairness"],
              "sst_attr_list": ["gender"], "eval_batch_size": 4096, "metric_decimal_place": 4}
    out = run_recbole(model="FOCF", config_dict=dict(common, fair_objective="value"))
    res = out["test_result"]
    assert {"ndcg@10", "recall@5", "hit@10", "mrr@5", "Differential Fairness of sensitive attribute gender",
            "Value Unfairness of sensitive attribute gender"} <= set(res)
    assert all(np.isfinite(v) for v in res.values()) and 0.0 <= res["hit@10"] <= 1.0
    assert out["best_valid_score"] == out["best_valid_result"]["ndcg@10"]
    out = run_recbole(model="NFCF", config_dict=dict(common, mlp_hidden_size=[16, 8], load_pretrain_path=None,
                                                     LABEL_FIELD="label"))
    assert 0.0 <= out["test_result"]["ndcg@10"] <= 1.0
    out = run_recbole(model="PFCN_PMF", config_dict=dict(common, filter_mode="sm", dis_hidden_size_list=[16, 8],
                                                         train_epoch_interval=1))
    assert list(out["test_result"]) == ["sm-['gender']"] and "ndcg@10" in out["test_result"]["sm-['gender']"]


UNI_CASES = sorted(glob.glob(os.path.join(GOLDEN, "collector_uni_*.npz")))


@pytest.mark.parametrize("path", UNI_CASES, ids=[os.path.basename(p)[14:-4] for p in UNI_CASES])
def test_uni_collector_matches_reference_golden_including_its_quirks(path):
    """Negative-sampling (`uni100`) branch: candidate lists instead of the dense -inf matrix; the reference's row
    arithmetic for `rec.negative_score` (rows [P, 2P) of the batch) is reproduced, -inf / nan results included."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.evaluator import Collector, Evaluator
    z = np.load(path)
    n_items = int(z["n_items"])
    cfg = Config(config_dict={"metrics": [str(m) for m in z["metrics"]], "topk": [int(k) for k in z["topk"]],
                              "metric_decimal_place": 10, "sst_attr_list": ["gender"],
                              "eval_args": {"mode": f"uni{int(z['n_neg'])}"}, "device": "cuda"})
    col, ev = Collector(cfg), Evaluator(cfg)
    assert not col.full
    d = lambda k: torch.from_numpy(z[k]).cuda()
    for b in range(int(z["n_batches"])):
        inter = Interaction({"item_id": d(f"items{b}"), "gender": d(f"sst{b}")})
        col.eval_batch_collect_candidates(d(f"origin{b}"), d(f"row_idx{b}"), inter, d(f"pos_u{b}"), d(f"pos_i{b}"), n_items)
    struct = col.get_data_struct()
    for key in ("rec.topk", "rec.positive_score", "data.positive_i", "rec.negative_score", "data.negative_i", "data.gender"):
        if "collected." + key in z.files:
            np.testing.assert_array_equal(struct[key].cpu().numpy(), z["collected." + key])
    ref = json.loads(str(z["result_json"]))
    got = ev.evaluate(struct)
    assert set(got) == set(ref)
    for k, v in ref.items():
        if np.isnan(v):
            assert np.isnan(got[k]), (k, got[k])
        else:
            tol = 5e-6 if "Differential" in k or "NonParity" in k else 1e-8
            assert abs(got[k] - v) <= tol * max(1.0, abs(v)), (k, got[k], v)


def test_neg_sample_eval_loader_against_the_reference_recipe():
    """uniN loader: users in id order, per user [positives | N negatives per positive], negatives from consecutive
    single-user sample_by_user_ids calls on ONE numpy stream (oracle), batch size rule of general_dataloader.py:100-117."""
    from fairrec.config import Config
    from fairrec.data.dataloader import NegSampleEvalDataLoader
    from fairrec.data.dataset import InteractionDataset
    from fairrec.data.interaction import Interaction
    from fairrec.sampler import DeviceRandomState, Sampler
    from oracle import sampler as OS
    rng = np.random.default_rng(2)
    n_users, n_items, N = 25, 60, 7
    cfg = Config(config_dict={"eval_batch_size": 200, "device": "cuda", "eval_args": {"mode": f"uni{N}"}})
    users = Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(rng.integers(0, 2, n_users).astype(np.float32))})
    mk = lambda n: InteractionDataset(cfg, Interaction({"user_id": torch.from_numpy(rng.integers(1, n_users, n)),
                                                        "item_id": torch.from_numpy(rng.integers(1, n_items, n))}),
                                      users, n_users, n_items)
    train, test = mk(200), mk(50)
    tu, ti = test.inter_feat["user_id"].numpy().copy(), test.inter_feat["item_id"].numpy().copy()
    used = [set() for _ in range(n_users)]
    for a, b in list(zip(train.inter_feat["user_id"].tolist(), train.inter_feat["item_id"].tolist())) + list(zip(tu, ti)):
        used[a].add(int(b))
    rs = DeviceRandomState("cuda", 77)
    sampler = Sampler(["train", "test"], [train, test], device="cuda", random_state=rs).set_phase("test")
    dl = NegSampleEvalDataLoader(cfg, test, sampler)
    order = np.argsort(tu, kind="stable")
    su, si = tu[order], ti[order]
    uid_list = np.unique(su)
    sizes = sorted((np.bincount(su, minlength=n_users)[uid_list] * (1 + N)).tolist(), reverse=True)
    step, tot = 1, sizes[0]
    for k in range(1, len(sizes)):
        if tot + sizes[k] > 200:
            break
        step, tot = k + 1, tot + sizes[k]
    assert dl.step == step
    ors = OS.MT19937(77)
    seen = 0
    for b, (inter, row_idx, pu, pi) in enumerate(dl):
        uids = uid_list[b * step:(b + 1) * step]
        exp_u, exp_i, exp_row, exp_pu, exp_pi = [], [], [], [], []
        for r, u in enumerate(uids):
            pos = si[su == u]
            neg = OS.sample_by_key_ids(ors, np.full(len(pos), u), N, used, n_items)
            exp_u += [u] * (len(pos) * (1 + N))
            exp_i += list(pos) + list(neg)
            exp_row += [r] * (len(pos) * (1 + N))
            exp_pu += [r] * len(pos)
            exp_pi += list(pos)
        np.testing.assert_array_equal(inter["user_id"].cpu().numpy(), exp_u)
        np.testing.assert_array_equal(inter["item_id"].cpu().numpy(), exp_i)
        np.testing.assert_array_equal(row_idx.cpu().numpy(), exp_row)
        np.testing.assert_array_equal(pu.cpu().numpy(), exp_pu)
        np.testing.assert_array_equal(pi.cpu().numpy(), exp_pi)
        np.testing.assert_array_equal(inter["gender"].cpu().numpy(), users["gender"].numpy()[np.array(exp_u)])
        seen += len(uids)
    assert seen == len(uid_list)


def test_run_recbole_uni_evaluation(tmp_path):
    from fairrec.quick_start import run_recbole
    out = run_recbole(model="PFCN_PMF", config_dict={
        "epochs": 1, "train_batch_size": 512, "synthetic_users": 120, "synthetic_items": 300, "synthetic_interactions": 3000,
        "device": "cuda", "checkpoint_dir": str(tmp_path), "embedding_size": 16, "eval_args": {"mode": "uni20"},
        "topk": [5, 10], "valid_metric": "ndcg@10", "valid_metric_bigger": True, "filter_mode": "none",
        "metrics": ["NDCG", "Recall", "Hit", "MRR", "DifferentialFairness", "NonParityUnfairness"],
        "sst_attr_list": ["gender"], "eval_batch_size": 2048, "metric_decimal_place": 4})
    res = out["test_result"]["none"]
    assert 0.0 <= res["ndcg@10"] <= 1.0 and np.isfinite(res["Differential Fairness of sensitive attribute gender"])


def test_run_recbole_fairgo_with_the_reference_test_yaml_metrics(tmp_path):
    """FairGo_PMF end to end with the metric list and evaluation mode of the reference's own test.yaml (uni100 there;
    uni20 on this small catalogue): pretrain + finetune, validation per epoch, test results per stage."""
    from fairrec.quick_start import run_recbole
    out = run_recbole(model="FairGo_PMF", config_dict={
        "epochs": 2, "pretrain_epochs": 2, "train_epoch_interval": 1, "train_batch_size": 512, "synthetic_users": 150,
        "synthetic_items": 300, "synthetic_interactions": 4000, "device": "cuda", "checkpoint_dir": str(tmp_path),
        "embedding_size": 16, "n_layers": 2, "dis_hidden_size_list": [16, 8, 4], "filter_hidden_size_list": [32, 16],
        "aggr_method": "LBA", "vs_weights": [4, 1], "fair_weight": 0.1, "weight_decay": 1e-4,
        "eval_args": {"mode": "uni20"}, "topk": [5], "valid_metric": "ndcg@5", "valid_metric_bigger": True,
        "metrics": ["NDCG", "Recall", "Hit", "MRR", "DifferentialFairness", "GiniIndex", "PopularityPercentage",
                    "ValueUnfairness", "AbsoluteUnfairness", "UnderUnfairness", "OverUnfairness", "NonParityUnfairness"],
        "sst_attr_list": ["gender"], "eval_batch_size": 2048, "metric_decimal_place": 4, "neg_sampling": None})
    res = out["test_result"]
    assert {"pretrain-ndcg@5", "finetune-ndcg@5", "finetune-giniindex@5", "finetune-popularitypercentage@5",
            "finetune-Differential Fairness of sensitive attribute gender"} <= set(res)
    assert 0.0 <= res["finetune-ndcg@5"] <= 1.0 and 0.0 <= res["finetune-giniindex@5"] <= 1.0


@pytest.mark.parametrize("K", [1, 5, 20])
def test_segment_topk_kernel_against_dense_rows(K):
    """fr_eval_topk_segments / fr_eval_lookup_segments (the uniN ranking as one launch, a wave per user) against the reference's
    recipe spelled out: scatter a user's candidates into a dense -inf row (trainer.py:441-456), torch.topk on the CPU
    (collector.py:149).  Segments of 0 to 700 candidates, items drawn twice, users with fewer than K + 1 distinct candidates,
    score ties inside and at the end of a list: where the kernel flags nothing its list must EQUAL the reference's, and it must
    flag exactly the users whose first K + 1 distinct scores contain an equal pair (bit 0) or are fewer than K + 1 (bit 1)."""
    from fairrec import _C
    lib = _C.lib()
    g = torch.Generator().manual_seed(K)
    n_items, U = 500, 300
    counts = torch.randint(0, 8, (U,), generator=g) * 101
    counts[5], counts[6], counts[7] = 3, K, 0
    counts[8] = 707
    seg = torch.zeros(U + 1, dtype=torch.int64)
    seg[1:] = torch.cumsum(counts, 0)
    n = int(seg[-1])
    items = torch.randint(1, n_items, (n,), generator=g)
    scores = torch.rand(n, generator=g)
    rows = torch.repeat_interleave(torch.arange(U), counts)
    quant = (rows % 3 == 0)                                    # a third of the users: few distinct score values -> ties
    scores[quant] = (scores[quant] * 6).floor() / 6
    for u in range(U):                                         # an item drawn twice scores the same twice
        a, b = int(seg[u]), int(seg[u + 1])
        first = {}
        for j in range(a, b):
            it = int(items[j])
            if it in first:
                scores[j] = scores[first[it]]
            else:
                first[it] = j
    d_seg, d_items, d_scores = seg.cuda(), items.cuda(), scores.cuda()
    topk = torch.empty((U, K), dtype=torch.int64, device="cuda")
    flags = torch.empty(U, dtype=torch.int32, device="cuda")
    _C.check(lib.fr_eval_topk_segments(d_seg.data_ptr(), U, d_items.data_ptr(), d_scores.data_ptr(), K, topk.data_ptr(),
                                       flags.data_ptr(), _C.current_stream()), "fr_eval_topk_segments")
    topk, flags = topk.cpu(), flags.cpu()
    for u in range(U):
        a, b = int(seg[u]), int(seg[u + 1])
        dense = torch.full((n_items,), -float("inf"))
        dense[items[a:b]] = scores[a:b]
        vals, idx = torch.topk(dense, K + 1)
        distinct = int((dense > -float("inf")).sum())
        short = distinct < K + 1
        m = min(distinct, K + 1)
        tie = bool((vals[1:m] == vals[:m - 1]).any()) if m >= 2 else False
        assert int(flags[u]) == (1 if tie else 0) | (2 if short else 0), (u, int(flags[u]), tie, short)
        if not tie and not short:
            assert topk[u].tolist() == idx[:K].tolist(), u
        elif not tie:
            assert topk[u, :min(distinct, K)].tolist() == idx[:min(distinct, K)].tolist() and (topk[u, distinct:] == 0).all(), u
    # lookups: present pairs, absent pairs, the second copy of a doubled item
    q_rows = torch.randint(0, U, (4000,), generator=g)
    q_items = torch.randint(1, n_items, (4000,), generator=g)
    out = torch.empty(4000, dtype=torch.float32, device="cuda")
    d_qr, d_qi = q_rows.cuda(), q_items.cuda()
    _C.check(lib.fr_eval_lookup_segments(d_seg.data_ptr(), U, d_items.data_ptr(), d_scores.data_ptr(), d_qr.data_ptr(),
                                         d_qi.data_ptr(), 4000, out.data_ptr(), _C.current_stream()), "fr_eval_lookup_segments")
    out = out.cpu()
    for q in range(0, 4000, 7):
        u, it = int(q_rows[q]), int(q_items[q])
        a, b = int(seg[u]), int(seg[u + 1])
        hit = (items[a:b] == it).nonzero()
        want = float(scores[a + int(hit[0])]) if len(hit) else -float("inf")
        assert float(out[q]) == want, (q, u, it)


def test_segment_lookup_hands_back_a_score_that_is_not_a_number():
    """A diverged scorer's NaN sits in the reference's dense matrix and comes back out of it; the segment lookup must not lose
    it to a maximum over its lanes.  Also: a query count that is not a multiple of the lanes per query, a one-candidate segment."""
    from fairrec import _C
    lib = _C.lib()
    seg = torch.tensor([0, 1, 40, 40, 141], dtype=torch.int64)
    items = torch.cat([torch.tensor([7]), torch.arange(1, 40), torch.arange(200, 301)])
    scores = torch.arange(items.numel(), dtype=torch.float32) * 0.25 - 3.0
    scores[0] = float("nan")
    scores[17] = float("nan")
    scores[140] = float("inf")
    q_rows = torch.tensor([0, 0, 1, 1, 1, 2, 3, 3, 3, 1, 3], dtype=torch.int64)
    q_items = torch.tensor([7, 8, 17, 1, 39, 5, 200, 300, 301, 40, 250], dtype=torch.int64)
    out = torch.full((q_rows.numel(),), 123.0, device="cuda")
    d = [t.cuda() for t in (seg, items, scores, q_rows, q_items)]
    _C.check(lib.fr_eval_lookup_segments(d[0].data_ptr(), 4, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                         q_rows.numel(), out.data_ptr(), _C.current_stream()), "fr_eval_lookup_segments")
    got = out.cpu().numpy()
    dense = np.full((4, 400), -np.inf, dtype=np.float32)
    rows = np.repeat(np.arange(4), np.diff(seg.numpy()))
    dense[rows, items.numpy()] = scores.numpy()
    want = dense[q_rows.numpy(), q_items.numpy()]
    assert np.array_equal(got, want, equal_nan=True), (got, want)
    assert np.isnan(got[0]) and np.isnan(got[2]) and got[1] == -np.inf and got[7] == np.inf
    # ... and the ranking hands a user with a NaN candidate to the host (torch.topk ranks NaN first): flag bit 0
    topk = torch.empty((4, 5), dtype=torch.int64, device="cuda")
    flags = torch.empty(4, dtype=torch.int32, device="cuda")
    _C.check(lib.fr_eval_topk_segments(d[0].data_ptr(), 4, d[1].data_ptr(), d[2].data_ptr(), 5, topk.data_ptr(), flags.data_ptr(),
                                       _C.current_stream()), "fr_eval_topk_segments")
    assert flags.cpu().tolist() == [1 | 2, 1, 2, 0]
    assert topk[3].cpu().tolist() == [300, 299, 298, 297, 296]


def test_segment_topk_at_baseline_sizes_against_the_dense_matrix():
    """fr_eval_topk_segments at the sizes of BASELINE configs[1]'s evaluation (100 001 items, 2 800 users x (1-3 positives + 100
    negatives each), K = 20): the reference's recipe -- candidates scattered into a dense [users, n_items] matrix of -inf,
    torch.topk of it (trainer.py:441-456, collector.py:149) -- on the device; random fp32 scores, so no list hangs on a tie and
    the lists must be EQUAL; items drawn twice carry one score."""
    from fairrec import _C
    lib = _C.lib()
    g = torch.Generator(device="cuda").manual_seed(3)
    U, n_items, K = 2800, 100_001, 20
    counts = 101 + torch.randint(0, 3, (U,), device="cuda", generator=g)
    seg = torch.zeros(U + 1, dtype=torch.int64, device="cuda")
    seg[1:] = torch.cumsum(counts, 0)
    n = int(seg[-1])
    rows = torch.repeat_interleave(torch.arange(U, device="cuda"), counts)
    items = torch.randint(1, n_items, (n,), device="cuda", generator=g)
    scores = torch.rand(n, device="cuda", generator=g)
    dense = torch.full((U, n_items), -float("inf"), device="cuda")
    dense[rows, items] = scores                                  # (a duplicate: one of its scores, as in the reference)
    scores = dense[rows, items].contiguous()                     # ... which both copies then carry
    want = torch.topk(dense, K, dim=-1).indices
    topk = torch.empty((U, K), dtype=torch.int64, device="cuda")
    flags = torch.empty(U, dtype=torch.int32, device="cuda")
    _C.check(lib.fr_eval_topk_segments(seg.data_ptr(), U, items.data_ptr(), scores.data_ptr(), K, topk.data_ptr(), flags.data_ptr(),
                                       _C.current_stream()), "fr_eval_topk_segments")
    clean = flags == 0
    assert int(clean.sum()) >= U - 2                             # (an exact fp32 tie among 20 of 100 random scores: ~1e-4 per user)
    assert torch.equal(topk[clean], want[clean])
    # the lookups: every candidate's own score comes back
    q = torch.randint(0, n, (50_000,), device="cuda", generator=g)
    out = torch.empty(q.numel(), dtype=torch.float32, device="cuda")
    qr, qi = rows[q].contiguous(), items[q].contiguous()
    _C.check(lib.fr_eval_lookup_segments(seg.data_ptr(), U, items.data_ptr(), scores.data_ptr(), qr.data_ptr(), qi.data_ptr(),
                                         q.numel(), out.data_ptr(), _C.current_stream()), "fr_eval_lookup_segments")
    assert torch.equal(out, dense[qr, qi])
