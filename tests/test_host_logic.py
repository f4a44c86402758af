"""CPU: host-side plumbing (Interaction, Config, dataloaders, Adam scalar table, utils)."""
import math
import os

import numpy as np
import pytest
import torch

from fairrec.config import Config
from fairrec.data.dataloader import FOCFDataLoader, TrainDataLoader
from fairrec.data.dataset import InteractionDataset, synthetic_dataset
from fairrec.data.interaction import Interaction, cat_interactions
from fairrec.optim import adam_step_scalars
from fairrec.utils import early_stopping, get_model, get_trainer

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_interaction_contract():
    it = Interaction({"a": torch.arange(6), "b": torch.arange(6).float() * 2})
    assert len(it) == 6 and it.columns == ["a", "b"] and "a" in it
    assert torch.equal(it[2:4]["a"], torch.tensor([2, 3]))
    assert torch.equal(it[[0, 5]]["b"], torch.tensor([0.0, 10.0]))
    assert torch.equal(it.repeat(2)["a"], torch.arange(6).repeat(2))
    assert torch.equal(it.repeat_interleave(2)["a"], torch.arange(6).repeat_interleave(2))
    it2 = Interaction({"a": torch.tensor([3, 1, 2]), "b": torch.tensor([0.0, 1.0, 2.0])})
    it2.sort("a")
    assert it2["b"].tolist() == [1.0, 2.0, 0.0]
    assert len(cat_interactions([it, it])) == 12
    it.update(Interaction({"c": torch.zeros(6)}))
    it.drop("c")
    with pytest.raises(ValueError):
        it.drop("c")
    assert it.to("cpu")["a"].device.type == "cpu"


def test_config_priority_and_missing_keys(tmp_path):
    f = tmp_path / "c.yaml"
    f.write_text("embedding_size: 32\nweight_decay: 1e-6\nfair_objective: value\n")
    c = Config(model="FOCF", config_file_list=[str(f)], config_dict={"embedding_size": 16})
    assert c["embedding_size"] == 16            # dict > file > model yaml
    assert c["weight_decay"] == 1e-6 and isinstance(c["weight_decay"], float)   # 1e-6 parses as float
    assert c["fair_objective"] == "value" and c["fair_weight"] == 1.0
    assert c["not_a_key"] is None               # configurator.py:405-409
    assert c["learner"] == "adam" and c["learning_rate"] == 0.001


def test_valid_metric_bigger_follows_the_metric_name():
    """configurator.py:306-307: `valid_metric_bigger` is derived from the metric (False for the reference's
    smaller-is-better classes), whatever a config says -- FOCF.yaml's `valid_metric: NDCG@5` must be maximised."""
    from fairrec.config import Config
    for metric, bigger in (("NDCG@5", True), ("rmse", False), ("MRR@10", True), ("ValueUnfairness", False),
                           ("DifferentialFairness", False), ("GiniIndex@10", False), ("Recall@20", True)):
        for given in (True, False, None):
            d = {"valid_metric": metric, "device": "cpu"}
            if given is not None:
                d["valid_metric_bigger"] = given
            assert Config(model=None, config_dict=d)["valid_metric_bigger"] is bigger, (metric, given)


def test_adam_scalars_match_torch_formula():
    tab = adam_step_scalars(1e-3, 0.9, 0.999, 40000)
    for j in (1, 2, 10, 1000, 20000, 40000):
        assert tab[4 * j] == np.float32(1e-3 / (1 - 0.9 ** j))
        assert tab[4 * j + 1] == np.float32(1 / math.sqrt(1 - 0.999 ** j))
    assert tab[4 * 40000] == np.float32(1e-3) and tab[4 * 40000 + 1] == np.float32(1.0)   # saturated


def test_train_dataloader_covers_dataset_once():
    c = Config(model="FOCF", config_dict={"train_batch_size": 128})
    ds = synthetic_dataset(c, 50, 40, 1000)
    dl = TrainDataLoader(c, ds, shuffle=True)
    torch.manual_seed(0)
    seen = torch.cat([b["user_id"] * 1000 + b["item_id"] for b in dl])
    assert len(dl) == 8 and seen.numel() == 1000
    b = next(iter(dl))
    assert set(b.columns) == {"user_id", "item_id", "rating", "gender"}
    assert torch.equal(b["gender"], ds.user_feat["gender"][b["user_id"]])


@pytest.mark.parametrize("per_call", [1, 3, 100])
def test_loaders_hand_out_runs_of_batches_equal_to_iterating_them(per_call):
    """`take(n)` (what Trainer._train_epoch feeds `model.train_steps` with: the next n batches as ONE Interaction) against plain
    iteration, for the shuffled fixed-size loader and for the item-complete one: same rows in the same order, same batch
    boundaries (a short last batch; ragged item-complete sizes), the same generator consumption, epoch after epoch."""
    c = Config(model="FOCF", config_dict={"train_batch_size": 128})
    for kind in ("plain", "item_complete"):
        def make():
            ds = synthetic_dataset(c, 50, 40, 1000, seed=3)
            return TrainDataLoader(c, ds, shuffle=True) if kind == "plain" else FOCFDataLoader(c, ds)
        a, b = make(), make()
        assert b.sliceable
        torch.manual_seed(1); np.random.seed(1)
        want = [[bt for bt in a] for _ in range(2)]
        after_iterating = (torch.get_rng_state().clone(), np.random.get_state()[1].copy(), np.random.get_state()[2])
        torch.manual_seed(1); np.random.seed(1)
        for epoch in range(2):
            iter(b)
            got_rows, got_sizes = [], []
            while True:
                run = b.take(per_call)
                if run is None:
                    break
                inter, size = run
                sizes = [size] * -(-len(inter) // size) if isinstance(size, int) else list(size)
                if isinstance(size, int):
                    sizes[-1] = len(inter) - size * (len(sizes) - 1)
                assert sum(sizes) == len(inter) and len(sizes) <= per_call
                got_rows.append(inter)
                got_sizes += sizes
            assert got_sizes == [len(bt) for bt in want[epoch]], kind
            for col in ("user_id", "item_id", "rating", "gender"):
                np.testing.assert_array_equal(torch.cat([r[col] for r in got_rows]).numpy(),
                                              torch.cat([bt[col] for bt in want[epoch]]).numpy(), err_msg=f"{kind} {col}")
        # ... and both ways of walking two epochs took the same draws from both generators
        assert torch.equal(torch.get_rng_state(), after_iterating[0])
        st = np.random.get_state()
        assert np.array_equal(st[1], after_iterating[1]) and st[2] == after_iterating[2]


def test_focf_dataloader_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN, "dataloader_focf.npz"))
    c = Config(model="FOCF", config_dict={"train_batch_size": int(z["step"])})
    inter = Interaction({"user_id": torch.from_numpy(z["user_id"]), "item_id": torch.from_numpy(z["item_id"]),
                         "rating": torch.from_numpy(z["rating"])})
    ds = InteractionDataset(c, inter, None, n_users=100, n_items=int(z["item_num"]))
    dl = FOCFDataLoader(c, ds)
    np.random.seed(int(z["np_seed"]))
    it = iter(dl)
    for b in range(4):
        batch = next(it)
        for col in ("user", "item", "rating"):
            np.testing.assert_array_equal(batch[f"{col}_id" if col != "rating" else "rating"].numpy(),
                                          z[f"batch{b}_{col}"])


def test_dispatch_and_early_stopping():
    assert get_model("FOCF").__name__ == "FOCF"
    assert get_trainer(None, "FOCF").__name__ == "Trainer"     # no FOCFTrainer -> plain Trainer (utils.py:76-94)
    with pytest.raises(ValueError):
        get_model("NoSuchModel")
    assert early_stopping(0.5, 0.4, 3, 10, bigger=True) == (0.5, 0, False, True)
    assert early_stopping(0.3, 0.4, 10, 10, bigger=True) == (0.4, 11, True, False)
    assert early_stopping(0.3, 0.4, 0, 10, bigger=False) == (0.3, 0, False, True)


def test_no_cpu_fallback():
    from fairrec import _C
    from fairrec.model.fair_recommender.focf import FocfEngine
    with pytest.raises(_C.FairrecError):
        FocfEngine(torch.zeros(4, 8), torch.zeros(4, 8), "none", 0.0, 5.0)


def test_nfcf_reset_params_matches_reference_golden(tmp_path):
    """NFCF.reset_params (nfcf.py:49-67): de-biasing projection of the pre-trained user table, freeze, item re-init."""
    from fairrec.model.fair_recommender.nfcf import NFCF
    z = np.load(os.path.join(GOLDEN, "nfcf_finetune.npz"))
    n_users, D = z["pretrain_user_embedding"].shape
    n_items = z["init.item_embedding.weight"].shape[0]

    class DS:
        def num(self, f):
            return {"user_id": n_users, "item_id": n_items}[f]

        def get_user_feature(self):
            return Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(z["gender"])})

    ck = tmp_path / "pre.pth"
    torch.save({"state_dict": {"user_embedding.weight": torch.tensor(z["pretrain_user_embedding"])}}, ck)
    cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": [16, 8], "device": "cpu",
                                            "load_pretrain_path": str(ck)})
    m = NFCF(cfg, DS())
    np.testing.assert_allclose(m.user_embedding.weight.detach().numpy(), z["init.user_embedding.weight"], rtol=1e-6, atol=1e-7)
    assert not m.user_embedding.weight.requires_grad and m.item_embedding.weight.requires_grad
    assert get_model("NFCF") is NFCF and get_trainer(None, "NFCF").__name__ == "Trainer"


def test_exchange_capacity_fits_the_owner_sort():
    """An owner sorts the G*cap ids it receives for one table in one launch (FR_SORT_MAX); the bench shapes
    (B = 8192 per rank, G = 1, 2, 4, 8) must fit, and the mean fill B/G must leave head-room."""
    from fairrec.sharded import SORT_MAX, exchange_capacity
    for G in (1, 2, 3, 4, 8, 16):
        for B in (64, 2048, 8192):
            cap = exchange_capacity(B, G, 2.0)
            assert G * cap <= max(SORT_MAX, G) and cap >= min(B, -(-B // G))
    assert exchange_capacity(8192, 8, 2.0) == 2048 and exchange_capacity(8192, 1, 2.0) == 8192
    from fairrec.sharded import capacity_is_sort_bound
    # at G = 8, B = 8192 the owner-side sort bounds the capacity (the overflow message must not suggest capacity_factor)
    assert capacity_is_sort_bound(8192, 8, 2.0) and not capacity_is_sort_bound(8192, 8, 1.5)
    assert not capacity_is_sort_bound(1024, 2, 2.0)


def test_recbole_import_alias_resolves_to_the_native_package():
    """User code written against the reference's package path keeps working (plugin surface by name)."""
    import fairrec.model.abstract_recommender as native
    from recbole.model.abstract_recommender import FairRecommender
    from recbole.quick_start import run_recbole
    from recbole.trainer import PFCN_BiasedMFTrainer, Trainer
    from recbole.utils import InputType, get_model, get_trainer
    assert FairRecommender is native.FairRecommender and callable(run_recbole)
    assert get_model("FOCF").__name__ == "FOCF" and get_model("FOCF").input_type == InputType.POINTWISE
    assert get_trainer(None, "PFCN_BiasedMF") is PFCN_BiasedMFTrainer and get_trainer(None, "FOCF") is Trainer
    import pytest
    with pytest.raises(ImportError):
        import recbole.utils.wandblogger  # noqa: F401  (not part of the hot path, not provided)


def _e2e(case):
    z = np.load(os.path.join(GOLDEN, f"e2e_{case}.npz"))
    import json
    c = {k: v for k, v in json.loads(str(z["config"])).items() if v is not None or k == "neg_sampling"}
    c.pop("seed", None)
    cfg = Config(model=str(z["model"]), config_dict=dict(c, device="cpu"))
    cols = {k[6:]: torch.from_numpy(z[k].astype(np.int64) if k.endswith("_id") else z[k].astype(np.float32))
            for k in z.files if k.startswith("train.")}
    users = {k[10:]: torch.from_numpy(z[k].astype(np.int64) if k.endswith("_id") else z[k].astype(np.float32))
             for k in z.files if k.startswith("user_feat.")}
    ds = InteractionDataset(cfg, Interaction(cols), Interaction(users), int(z["n_users"]), int(z["n_items"]))
    return z, cfg, ds


def test_focf_loader_reproduces_the_reference_run_on_ml100k():
    """BASELINE.json configs[0] on the host: from numpy's generator state at the start of the reference's trainer.fit, the
    item-complete batcher yields the reference FOCFDataLoader's 80 batches of its 2-epoch ml-100k run, row for row
    (tests/golden/gen_e2e_golden.py; focf_dataloader.py:37-51)."""
    z, cfg, ds = _e2e("focf_ml100k")
    dl = FOCFDataLoader(cfg, ds)
    np.random.set_state(("MT19937", z["rng.np_key"].astype(np.uint32), int(z["rng.np_pos"]), 0, 0.0))
    t = 0
    for epoch in range(2):
        for batch in dl:
            for col in ("user_id", "item_id", "rating", "gender"):
                np.testing.assert_array_equal(batch[col].numpy().astype(np.int64), z[f"step{t}.{col}"].astype(np.int64),
                                              err_msg=f"batch {t} column {col}")
            t += 1
    assert t == len(z["kind"]) == 80


@pytest.mark.parametrize("case", ["pfcn_biasedmf_sm", "fairgo_pmf_wap", "nfcf_pretrain"])
def test_epoch_shuffles_reproduce_the_reference_run(case):
    """The positive rows of every batch of the reference trainers' 2-epoch runs (pair-wise: all rows; point-wise: the first
    half) from torch's CPU generator state at the start of trainer.fit: one torch.randperm per pass over the loader
    (general_dataloader.py:59-60 -> interaction.py:293-297), also through the filter / discriminator alternation."""
    from fairrec.utils.enum_type import InputType
    z, cfg, ds = _e2e(case)
    dl = TrainDataLoader(cfg, ds, sampler=None, shuffle=True)
    pointwise = cfg["MODEL_INPUT_TYPE"] == InputType.POINTWISE
    dl.step = int(cfg["train_batch_size"]) // (2 if pointwise else 1)          # what a sampler would make it (:41-50)
    torch.set_rng_state(torch.from_numpy(z["rng.torch"]))
    t, T = 0, len(z["kind"])
    while t < T:
        for batch in dl:
            ref_u, ref_i = z[f"step{t}.user_id"], z[f"step{t}.item_id"]
            n = len(ref_u) // 2 if pointwise else len(ref_u)
            np.testing.assert_array_equal(batch["user_id"].numpy(), ref_u[:n].astype(np.int64), err_msg=f"step {t}")
            np.testing.assert_array_equal(batch["item_id"].numpy(), ref_i[:n].astype(np.int64), err_msg=f"step {t}")
            if pointwise:      # the negatives' rows repeat the positives' users (abstract_dataloader.py:190-198)
                np.testing.assert_array_equal(ref_u[n:], ref_u[:n])
            t += 1
    assert t == T


def test_default_sweep_period_of_a_one_column_table_is_capped():
    """A table is swept about one batch of rows per step (period = n_rows / batch); a bias column is swept 64 rows per wave,
    so at that period a 10 M-row bias would be 128 waves of 1221 serial replayed steps each: its period is capped at 64
    (fairrec/optim.py::LazyTable.default_sweep; any period gives the same values)."""
    from fairrec.optim import LazyTable
    t = object.__new__(LazyTable)
    t.n_rows, t.dim = 10_000_001, 128
    assert t.default_sweep(8192) == 1221
    t.dim = 1
    assert t.default_sweep(8192) == 64
    t.n_rows = 1000
    assert t.default_sweep(8192) == 8


def test_torch_generator_state_blob_round_trip():
    """fairrec/sampler/torch_stream.py reads torch's CPU generator state (key words + position) out of get_rng_state()'s blob
    and writes an advanced one back: reading what was written gives the same words, a blob rewritten with its own content
    is the blob, and the position convention (left == 1 <=> regenerate first) holds right after seeding and inside a block."""
    import numpy as np
    import torch
    from fairrec.sampler import torch_stream as ts
    torch.manual_seed(11)
    blob = torch.get_rng_state()
    key, pos = ts._read(blob)
    assert pos == 624 and key.dtype == np.uint32 and key.shape == (624,)        # freshly seeded: nothing generated yet
    # (a seeded generator holds left = 1, next = 0; one whose draws ended on a block boundary left = 1, next = 624 -- both
    # regenerate before the next draw, and only the second form is ever written back)
    again = ts._read(ts._write(blob, key, pos))
    assert again[1] == 624 and np.array_equal(again[0], key)
    torch.randperm(10)                                                           # 9 draws: regenerates, then position 9
    blob2 = torch.get_rng_state()
    key2, pos2 = ts._read(blob2)
    assert pos2 == 9 and not np.array_equal(key2, key)
    assert torch.equal(ts._write(blob2, key2, pos2), blob2)
    # a CPU "device" takes the host path and is torch.randperm itself
    torch.manual_seed(5)
    a = torch.randperm(100)
    torch.manual_seed(5)
    assert torch.equal(ts.randperm(100, "cpu"), a)


def _numpy_epoch(uniq, indptr, step, pr, pr_end):
    """focf_dataloader.py:37-51 as the reference writes it: a boolean mask over the item ids and np.random.choice per pick."""
    picks, ends = [], []
    while pr < pr_end:
        is_select = np.ones(uniq.size, dtype=bool)
        cnt = 0
        while cnt < step and is_select.any():
            iid = np.random.choice(uniq[is_select], 1, False)[0]
            is_select[np.searchsorted(uniq, iid)] = False
            cnt += indptr[iid + 1] - indptr[iid]
            picks.append(iid)
        ends.append(len(picks))
        pr += step
    return np.array(picks, dtype=np.int64), np.array(ends, dtype=np.int64)


@pytest.mark.parametrize("seed,n_items,n_rows,step", [(0, 50, 600, 64), (1, 1300, 9000, 512), (2, 7, 40, 100),
                                                     (3, 700, 700, 16), (2020, 1, 5, 3), (5, 4000, 30000, 2048),
                                                     (7, 100000, 16384, 8192)],     # BASELINE configs[1]'s item count and batch
                         ids=["tiny", "ragged", "exhausted", "unit", "one_item", "mid", "baseline_items"])
def test_compose_epoch_makes_numpys_draws(seed, n_items, n_rows, step):
    """fr_focf_compose_epoch against numpy itself: same picks, same batch boundaries, the generator left at the same
    position (the next draws of numpy agree) -- including batches that exhaust the candidates (step > rows), a single
    candidate (no draw) and many regenerations of the 624 words."""
    from fairrec import _C
    lib = _C.lib()
    g = np.random.RandomState(seed)
    items = np.sort(g.randint(1, n_items + 1, n_rows))
    uniq = np.unique(items).astype(np.int64)
    indptr = np.searchsorted(items, np.arange(n_items + 2), side="left").astype(np.int64)
    np.random.seed(seed + 77)
    np.random.rand(seed % 5)                                    # an odd position in the 624 words
    st0 = np.random.get_state()
    want_picks, want_ends = _numpy_epoch(uniq, indptr, step, 0, n_rows)
    want_next = np.random.randint(0, 1 << 30, 8)
    state = np.empty(625, dtype=np.uint32)
    state[:624], state[624] = st0[1], st0[2]
    picks = np.empty(want_picks.size + 3, dtype=np.int64)
    ends = np.empty(want_ends.size, dtype=np.int64)
    got = np.zeros(1, dtype=np.int64)
    # too little room: refused, and the state is where it was
    before = state.copy()
    rc = lib.fr_focf_compose_epoch(state.ctypes.data, uniq.ctypes.data, uniq.size, indptr.ctypes.data, step, 0, n_rows,
                                   picks.ctypes.data, max(want_picks.size - 1, 0), ends.ctypes.data, ends.size, got.ctypes.data)
    assert rc != 0 and np.array_equal(state, before)
    _C.check(lib.fr_focf_compose_epoch(state.ctypes.data, uniq.ctypes.data, uniq.size, indptr.ctypes.data, step, 0, n_rows,
                                       picks.ctypes.data, picks.size, ends.ctypes.data, ends.size, got.ctypes.data), "compose")
    assert got[0] == want_ends.size
    assert np.array_equal(ends, want_ends) and np.array_equal(picks[:ends[-1]], want_picks)
    np.random.set_state(('MT19937', state[:624].copy(), int(state[624]), st0[3], st0[4]))
    assert np.array_equal(np.random.randint(0, 1 << 30, 8), want_next)


def test_host_topk_orders_equal_values_as_torch_cpu_topk_does():
    """fr_topk_like_torch_cpu against torch.topk itself (CPU backend: the kernel the reference's evaluation ranks with,
    collector.py:149) on rows where ties decide the list: few distinct values, the dense -inf rows of the uni100 protocol
    (trainer.py:441-456) with clamped scores, rows that are half -inf, and both branches of ATen's kernel (k * 64 <= n:
    partial_sort; otherwise nth_element + sort) on either side of the switch."""
    import ctypes
    from fairrec import _C
    lib = _C.lib()
    rng = np.random.default_rng(0)
    for n, k in ((120, 5), (1683, 5), (319, 5), (320, 5), (321, 5), (64, 1), (65, 1), (1000, 10), (50, 50), (7, 3)):
        for trial in range(40):
            nr, kind = 6, trial % 4
            if kind == 0:
                rows = rng.integers(0, 3, (nr, n)).astype(np.float32)
            elif kind == 1:
                rows = np.full((nr, n), -np.inf, np.float32)
                for r in range(nr):
                    c = rng.choice(n, min(n, max(k, int(rng.integers(k, min(n, 110) + 1)))), replace=False)
                    rows[r, c] = rng.integers(0, 2, c.size) * rng.random(c.size).astype(np.float32).round(1)
            elif kind == 2:
                rows = np.zeros((nr, n), np.float32)
                rows[:, rng.choice(n, n // 2, replace=False)] = -np.inf
            else:
                rows = rng.random((nr, n)).astype(np.float32).round(2)
            idx, val = np.zeros((nr, k), np.int64), np.zeros((nr, k), np.float32)
            assert lib.fr_topk_like_torch_cpu(rows.ctypes.data, nr, n, k, idx.ctypes.data, val.ctypes.data) == 0
            tv, ti = torch.topk(torch.from_numpy(rows), k, dim=-1)
            np.testing.assert_array_equal(idx, ti.numpy(), err_msg=f"n {n} k {k} kind {kind}")
            np.testing.assert_array_equal(val, tv.numpy())


def test_item_complete_batches_that_no_step_takes_are_seen_before_training():
    """FOCFDataLoader's batches are sized by the data (focf_dataloader.py:37-51): with one item rated 17 000 times a batch holds
    more rows than a FOCF step sorts (FR_SORT_MAX); quick_start sees that from the item degrees when it builds the loaders
    (and then trains on fixed-size batches, with a warning) instead of failing in mid-epoch."""
    from fairrec import _C
    from fairrec.quick_start import worst_item_complete_batch
    cfg = Config(model="FOCF", config_dict={"train_batch_size": 2048, "device": "cpu"})
    g = torch.Generator().manual_seed(1)
    i = torch.cat([torch.full((17000,), 7, dtype=torch.int64), torch.randint(1, 50, (3000,), generator=g)])
    u = torch.randint(1, 100, (i.numel(),), generator=g)
    ds = InteractionDataset(cfg, Interaction({"user_id": u, "item_id": i, "rating": torch.ones(i.numel())}),
                            Interaction({"user_id": torch.arange(100), "gender": torch.zeros(100)}), 100, 50)
    worst = worst_item_complete_batch(cfg, ds)
    assert worst >= 17000 + 2047 and worst > _C.FR_SORT_MAX
    ok = InteractionDataset(cfg, Interaction({"user_id": u[17000:], "item_id": i[17000:], "rating": torch.ones(3000)}),
                            Interaction({"user_id": torch.arange(100), "gender": torch.zeros(100)}), 100, 50)
    assert worst_item_complete_batch(cfg, ok) <= _C.FR_SORT_MAX
