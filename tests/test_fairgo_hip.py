"""GPU parity: FairGo_PMF (pretrain on lazy tables; finetune = whole-table filter MLPs on MFMA, CSR SpMM, WAP/LBA/LVA,
node + local discriminators, three optimizers) vs the reference's golden vectors, through the plugin surface."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu
CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fairgo_*.npz")))


class _DS:
    def __init__(self, n_users, n_items, z):
        from fairrec.data.interaction import Interaction
        self._n = {"user_id": n_users, "item_id": n_items}
        self._uf = Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(z["gender"]),
                                "age": torch.from_numpy(z["age"])})
        self.inter_feat = {"rating": torch.from_numpy(z["train_rating"])}
        self._coo = sp.coo_matrix((z["train_rating"], (z["train_user"], z["train_item"])), shape=(n_users, n_items))

    def num(self, f):
        return self._n[f]

    def get_user_feature(self):
        return self._uf

    def inter_matrix(self, form="coo", value_field=None):
        return self._coo


@pytest.mark.parametrize("replicated", [False, True], ids=["single", "data_parallel"])
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_fairgo_training_matches_reference_golden(path, replicated, request):
    if replicated:     # the data-parallel engine (replicated tables, fairrec/replicated_engine.py) as a 1-rank RCCL world:
        request.getfixturevalue("rccl_world1")      # every collective call runs; 2 ranks are covered over gloo
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.optim import FusedLazyAdam
    from fairrec.utils import get_model
    z = np.load(path)
    attrs = [str(a) for a in z["attrs"]]
    lr, wd, fw = (float(x) for x in z["hyper"])
    n_users, D = z["init.model.user_embedding_layer.weight"].shape
    n_items = z["init.model.item_embedding_layer.weight"].shape[0]
    name = str(z["model"])       # FairGo_GCN cases: the reference's FairGo_GCN class in its finetune stage (fairgo_gcn.py:173-250)
    cfg = Config(model=name, config_dict={
        "embedding_size": D, "sst_attr_list": attrs, "aggr_method": str(z["aggr"]), "n_layers": int(z["n_layers"]),
        "filter_hidden_size_list": [int(h) for h in z["filter_hidden"]], "dis_hidden_size_list": [int(h) for h in z["dis_hidden"]],
        "vs_weights": [float(v) for v in z["vs_weights"]], "fair_weight": fw, "device": "cuda",
        "data_parallel": replicated})
    model = get_model(name)(cfg, _DS(n_users, n_items, z))
    # the reference's L = D^-1 A, entry for entry
    L = model._norm_csr_host.tocoo()
    order = np.lexsort((L.col, L.row))
    np.testing.assert_array_equal(L.row[order], z["L_row"])
    np.testing.assert_array_equal(L.col[order], z["L_col"])
    np.testing.assert_allclose(L.data[order], z["L_val"], rtol=1e-6)
    model.load_state_dict({k[11:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.model.")}, strict=name != "FairGo_GCN")
    model = model.to("cuda")
    for s in attrs:
        model.filter_layer_dict[s].load_state_dict({k[len(f"init.filter.{s}."):]: torch.tensor(z[k]) for k in z.files
                                                    if k.startswith(f"init.filter.{s}.")})
        model.dis_layer_dict[s].load_state_dict({k[len(f"init.dis.{s}."):]: torch.tensor(z[k]) for k in z.files
                                                 if k.startswith(f"init.dis.{s}.")})
    eng = model.hip_engine()
    opts = {ph: FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=2, group=g)
            for ph, g in (("P", "pretrain"), ("F", "filter"), ("D", "dis"))}
    losses = []
    for t, ph in enumerate(str(x) for x in z["phases"]):
        u = z["user_id"][t]
        inter = Interaction({"user_id": torch.tensor(u), "item_id": torch.tensor(z["item_id"][t]),
                             "rating": torch.tensor(z["rating"][t]), "gender": torch.tensor(z["gender"][u]),
                             "age": torch.tensor(z["age"][u])}).to("cuda")
        sl = [s for s in str(z["sst_lists"][t]).split(",") if s]
        model.train_stage = "pretrain" if ph == "P" else "finetune"
        opts[ph].zero_grad()
        if ph == "P":
            loss = model.calculate_loss(inter, None)
        elif ph == "F":
            loss = model.calculate_loss(inter, sl)
        else:
            loss = model.calculate_dis_loss(inter, sl)
        losses.append(loss.detach().reshape(1).clone())
        loss.backward()
        opts[ph].step()
    np.testing.assert_allclose(torch.cat(losses).cpu().numpy(), z["loss"], rtol=1e-4, atol=1e-6)
    worst = [0.0, ""]

    def close(a, ref, what):
        # north star: 1e-4 relative; absolute floor 1e-6 of the tensor's scale (no BatchNorm in FairGo's MLPs, so no
        # tensor here is noise-amplified: the measured worst error is printed below)
        a = a.detach().cpu().numpy().astype(np.float64)
        ratio = np.abs(a - ref) / (1e-4 * np.abs(ref) + 1e-6 * max(1e-2, float(np.abs(ref).max())))
        if ratio.max() > worst[0]:
            worst[:] = [float(ratio.max()), what]
        assert ratio.max() <= 1.0, (what, float(np.abs(a - ref).max()), float(ratio.max()))

    for k, v in model.state_dict().items():
        if not k.startswith("gcn."):        # the pretrain stage's GCN: not in the reference-side placeholder, untouched here
            close(v, z["final.model." + k], k)
    for s in attrs:
        for k, v in model.filter_layer_dict[s].state_dict().items():
            close(v, z[f"final.filter.{s}.{k}"], f"filter.{s}.{k}")
        for k, v in model.dis_layer_dict[s].state_dict().items():
            close(v, z[f"final.dis.{s}.{k}"], f"dis.{s}.{k}")
    print("worst |err| / tolerance:", round(worst[0], 3), worst[1])
    eng.check_device_errors()
    np.testing.assert_allclose(model.predict(inter).cpu().numpy(), z["predict_last"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("replicated", [False, True], ids=["single", "data_parallel"])
def test_fairgo_full_batch_at_the_baseline_width(replicated, request):
    """BASELINE.json configs[3]'s step shape -- FairGo_GCN finetune (fairgo_gcn.py:173-250), WAP, embedding_size 128, filters
    [128, 64], discriminators [16, 8, 4], n_layers 2, B = 8192 -- on tables scaled down to what the CPU oracle evaluates in
    seconds (20 001 users x 5 001 items, 20 training ratings per user): filter and discriminator steps in the trainer's
    order through the HIP path against oracle/fairgo.py (pinned to the reference's goldens, tests/test_oracle_fairgo.py),
    every tensor of the filter / discriminator MLPs, the per-step losses and the last predictions compared."""
    if replicated:
        request.getfixturevalue("rccl_world1")
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import fairgo as O
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.optim import FusedLazyAdam
    from fairrec.utils import get_model
    n_users, n_items, D, B = 20_001, 5_001, 128, 8192
    filt, dis, phases = (128, 64), (16, 8, 4), "FDFDF"
    rng = np.random.default_rng(21)
    torch.manual_seed(21)
    gender = rng.integers(0, 2, n_users).astype(np.float32)
    tu = np.repeat(np.arange(1, n_users), 20)
    ti = rng.integers(1, n_items, tu.size)
    pair = np.unique(tu.astype(np.int64) * n_items + ti)          # distinct (user, item) pairs, as a rating matrix has
    tu, ti = pair // n_items, pair % n_items
    tr = rng.integers(1, 6, tu.size).astype(np.float32)
    z = {"model": np.array("FairGo_GCN"), "aggr": np.array("WAP"), "attrs": np.array(["gender"]), "phases": np.array(list(phases)),
         "sst_lists": np.array(["gender"] * len(phases)), "hyper": np.array([1e-3, 1e-4, 0.1]), "n_layers": np.array(2),
         "filter_hidden": np.array(filt), "dis_hidden": np.array(dis), "vs_weights": np.array([4, 1], dtype=np.float32),
         "gender": gender, "age": np.zeros(n_users, dtype=np.int64), "train_user": tu, "train_item": ti, "train_rating": tr}
    cfg = Config(model="FairGo_GCN", config_dict={
        "embedding_size": D, "sst_attr_list": ["gender"], "aggr_method": "WAP", "n_layers": 2,
        "filter_hidden_size_list": list(filt), "dis_hidden_size_list": list(dis), "vs_weights": [4.0, 1.0], "fair_weight": 0.1,
        "device": "cuda", "data_parallel": replicated})
    model = get_model("FairGo_GCN")(cfg, _DS(n_users, n_items, z))
    # a pretrained-looking state (N(0, 0.1) tables: scores of O(1), nothing saturated), the reference's default init of the MLPs
    with torch.no_grad():
        model.user_embedding_layer.weight.normal_(0, 0.1)
        model.item_embedding_layer.weight.normal_(0, 0.1)
    for k, v in model.state_dict().items():
        if not k.startswith("gcn."):
            z["init.model." + k] = v.detach().cpu().numpy().copy()
    for s, m in model.filter_layer_dict.items():
        for k, v in m.state_dict().items():
            z[f"init.filter.{s}.{k}"] = v.detach().cpu().numpy().copy()
    for s, m in model.dis_layer_dict.items():
        for k, v in m.state_dict().items():
            z[f"init.dis.{s}.{k}"] = v.detach().cpu().numpy().copy()
    L = O.norm_rating_matrix(n_users, n_items, tu, ti, tr)
    z["L_row"], z["L_col"], z["L_val"] = L.indices()[0].numpy(), L.indices()[1].numpy(), L.values().numpy()
    sel = rng.integers(0, tu.size, size=(len(phases), B))
    z["user_id"], z["item_id"], z["rating"] = tu[sel], ti[sel], tr[sel]
    ref = O.train(z)
    model = model.to("cuda")
    model.train_stage = "finetune"
    eng = model.hip_engine()
    opts = {ph: FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-4, group=g) for ph, g in (("F", "filter"), ("D", "dis"))}
    losses = []
    for t, ph in enumerate(phases):
        u = z["user_id"][t]
        inter = Interaction({"user_id": torch.tensor(u), "item_id": torch.tensor(z["item_id"][t]),
                             "rating": torch.tensor(z["rating"][t]), "gender": torch.tensor(gender[u])}).to("cuda")
        opts[ph].zero_grad()
        loss = model.calculate_loss(inter, ["gender"]) if ph == "F" else model.calculate_dis_loss(inter, ["gender"])
        losses.append(loss.detach().reshape(1).clone())
        loss.backward()
        opts[ph].step()
    np.testing.assert_allclose(torch.cat(losses).cpu().numpy(), ref["loss"], rtol=1e-4, atol=1e-6)
    worst = [0.0, ""]
    for kind, d in (("filter", model.filter_layer_dict), ("dis", model.dis_layer_dict)):
        for k, v in d["gender"].state_dict().items():
            a, b = v.detach().cpu().numpy().astype(np.float64), ref[f"final.{kind}.gender.{k}"]
            ratio = np.abs(a - b) / (1e-4 * np.abs(b) + 1e-6 * max(1e-2, float(np.abs(b).max())))
            if ratio.max() > worst[0]:
                worst[:] = [float(ratio.max()), f"{kind}.{k}"]
            assert ratio.max() <= 1.0, (kind, k, float(np.abs(a - b).max()), float(ratio.max()))
    print("worst |err| / tolerance:", round(worst[0], 3), worst[1])
    for k in ("user_embedding_layer.weight", "item_embedding_layer.weight"):         # frozen in the finetune stage
        np.testing.assert_array_equal(model.state_dict()[k].cpu().numpy(), z["init.model." + k])
    eng.check_device_errors()
    np.testing.assert_allclose(model.predict(inter).cpu().numpy(), ref["predict_last"], rtol=1e-4, atol=1e-6)


def test_fairgo_trainer_pretrain_then_finetune(tmp_path):
    """FairGoTrainer: pretrain epochs with optimizer_pretrain, checkpoint, stage switch, alternating finetune epochs."""
    from fairrec.config import Config
    from fairrec.data.dataloader import TrainDataLoader
    from fairrec.data.dataset import InteractionDataset
    from fairrec.data.interaction import Interaction
    from fairrec.utils import get_model, get_trainer, init_seed
    init_seed(3)
    n_users, n_items, n = 40, 30, 300
    g = torch.Generator().manual_seed(2)
    inter = Interaction({"user_id": torch.randint(1, n_users, (n,), generator=g), "item_id": torch.randint(1, n_items, (n,), generator=g),
                         "rating": torch.randint(1, 6, (n,), generator=g).float()})
    users = Interaction({"user_id": torch.arange(n_users), "gender": (torch.rand(n_users, generator=g) < 0.5).float()})
    users["gender"][1:3] = torch.tensor([0.0, 1.0])
    cfg = Config(model="FairGo_PMF", dataset="synth", config_dict={
        "embedding_size": 16, "aggr_method": "WAP", "n_layers": 2, "filter_hidden_size_list": [16, 8], "dis_hidden_size_list": [8, 4],
        "train_batch_size": 100, "epochs": 2, "pretrain_epochs": 2, "train_epoch_interval": 1, "device": "cuda",
        "checkpoint_dir": str(tmp_path), **({"graph_train_step": False} if os.environ.get("FAIRREC_TEST_NO_GRAPH") else {})})

    class DS(InteractionDataset):
        def inter_matrix(self, form="coo", value_field=None):
            return sp.coo_matrix((self.inter_feat["rating"].numpy(), (self.inter_feat["user_id"].numpy(),
                                                                       self.inter_feat["item_id"].numpy())), shape=(n_users, n_items))

    ds = DS(cfg, inter, users, n_users, n_items)
    model = get_model("FairGo_PMF")(cfg, ds).to("cuda")
    trainer = get_trainer(None, "FairGo_PMF")(cfg, model)
    assert type(trainer).__name__ == "FairGo_PMFTrainer" and model.train_stage == "pretrain"
    w0 = model.user_embedding_layer.weight.detach().clone()
    # (round 3 reported "Training loss is nan" here as an xfail, 1-2 of 40 processes after a data-parallel test in the same
    # process.  Root cause, round 4: RowGather's backward cleared its dense gradient with hipMemsetAsync, which a stream capture
    # turns into a memset NODE, and this runtime executes such a node on the FIRST launch of the graph only
    # (scratch/memset_node.py: 19 of 20 replays left the buffer as the block's previous tenant had left it) -- every replay
    # after the first fed garbage rows of dLoss/dE into all filter weights; garbage that happened to hold a NaN bit pattern
    # made it visible.  The clear is a kernel now (csrc/graph.hip), and the graphed run below must equal its eager twin.)
    trainer.fit(TrainDataLoader(cfg, ds, shuffle=False), valid_data=None, verbose=False, saved=True)
    assert model.train_stage == "finetune"
    eng = model.hip_engine()
    assert eng._tables["user_embedding_layer.weight"].step == 2 * 3           # 2 pretrain epochs x 3 batches, then frozen
    assert not torch.equal(model.user_embedding_layer.weight, w0)
    assert all(d.step == 2 * 3 for k, d in eng._dense.items() if k.startswith("filter."))   # every finetune epoch (interval 1)
    assert os.path.exists(trainer.saved_pretrain_model_file)
    assert get_model("FairGo_GCN").__mro__[1].__name__ == "FairGo_PMF"
    if cfg["graph_train_step"]:
        # the same fit with every step launched eagerly, from the same seed: captured steps must not change a single weight
        init_seed(3)
        cfg_e = Config(model="FairGo_PMF", dataset="synth", config_dict=dict(
            {k: cfg[k] for k in ("embedding_size", "aggr_method", "n_layers", "filter_hidden_size_list", "dis_hidden_size_list",
                                 "train_batch_size", "epochs", "pretrain_epochs", "train_epoch_interval")},
            device="cuda", checkpoint_dir=str(tmp_path / "eager"), graph_train_step=False))
        ds_e = DS(cfg_e, inter, users, n_users, n_items)
        model_e = get_model("FairGo_PMF")(cfg_e, ds_e).to("cuda")
        get_trainer(None, "FairGo_PMF")(cfg_e, model_e).fit(TrainDataLoader(cfg_e, ds_e, shuffle=False), valid_data=None,
                                                           verbose=False, saved=True)
        for kind in ("filter_layer_dict", "dis_layer_dict"):
            for (k, a), (_, b) in zip(getattr(model, kind)["gender"].state_dict().items(),
                                      getattr(model_e, kind)["gender"].state_dict().items()):
                assert torch.isfinite(a).all(), (kind, k)
                np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-7, err_msg=f"{kind}.{k}")
    # the discriminator pass reads the filtered table and its propagations from the per-pass cache (begin_dis_phase): the
    # same bits as recomputing them per step, and a filter step invalidates it
    b = next(iter(TrainDataLoader(cfg, ds, shuffle=False))).to("cuda")
    model._dis_cache.clear()
    with torch.no_grad():
        plain = model.calculate_dis_loss(b, ["gender"]).clone()
    model.begin_dis_phase(["gender"])
    c = model._dis_cache[("gender",)]
    assert c["version"] == model._filters_version()
    with torch.no_grad():
        assert torch.equal(model.calculate_dis_loss(b, ["gender"]), plain)
    trainer.optimizer_filter.zero_grad()
    model.calculate_loss(b, ["gender"]).backward()
    trainer.optimizer_filter.step()
    assert c["version"] != model._filters_version()          # stale now: the next call recomputes
    with torch.no_grad():
        moved = model.calculate_dis_loss(b, ["gender"])
        # (the loss itself sits at 2 ln 2 for an untrained discriminator and may round to the same float: the recomputed
        # filtered table is what must have moved with the filters, and the loss must be the one of THAT table)
        E_now = model._filtered_table(["gender"])
        assert not torch.equal(E_now, c["E"])
        assert torch.equal(moved, model._dis_terms(E_now, b, ["gender"]))
    # resume of the finetune checkpoint (trainer.py:807-834): optimizer_filter / optimizer_dis come back
    ck = torch.load(trainer.saved_model_file, weights_only=False)
    cfg2 = Config(model="FairGo_PMF", dataset="synth", config_dict=dict(
        {k: cfg[k] for k in ("embedding_size", "aggr_method", "n_layers", "filter_hidden_size_list", "dis_hidden_size_list",
                             "train_batch_size", "epochs", "pretrain_epochs", "train_epoch_interval", "checkpoint_dir")},
        device="cuda", pretrain_model_file_path=trainer.saved_pretrain_model_file))
    model2 = get_model("FairGo_PMF")(cfg2, ds).to("cuda")
    trainer2 = get_trainer(None, "FairGo_PMF")(cfg2, model2)
    trainer2.resume_checkpoint(trainer.saved_model_file)
    assert trainer2.start_epoch == ck["epoch"] + 1
    eng2 = model2.hip_engine()
    for k, st in list(ck["optimizer_filter"]["state"].items()) + list(ck["optimizer_dis"]["state"].items()):
        if k in eng2._dense:
            assert eng2._dense[k].step == int(st["step"])
            assert torch.equal(eng2._dense[k].m, st["exp_avg"].to("cuda"))
    assert torch.equal(model2.user_embedding_layer.weight, model.state_dict()["user_embedding_layer.weight"])


def test_fairgo_gcn_pretrain_matches_the_restated_pyg_gcn(tmp_path):
    """FairGo_GCN pretrain stage (whole table through a 2-layer GCN, dense Adam on tables + GCN) against the oracle's
    dense restatement of PyG's published GCNConv / BasicGNN.  PARITY UNPINNED: torch_geometric is not pinned by the
    reference and not installed, so this checks the HIP path against the restatement only."""
    import scipy.sparse as sp
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.model.fair_recommender.fairgo_gcn import FairGo_GCN, gcn_norm_matrix
    from fairrec.optim import FusedLazyAdam
    from oracle import fairgo as OF
    rng = np.random.default_rng(0)
    n_users, n_items, D, B, T = 30, 25, 16, 64, 6
    pairs = rng.choice((n_users - 1) * (n_items - 1), size=200, replace=False)
    tu, ti = pairs // (n_items - 1) + 1, pairs % (n_items - 1) + 1
    tr = rng.integers(1, 6, size=200).astype(np.float32)

    class DS:
        inter_feat = {"rating": torch.from_numpy(tr)}

        def num(self, f):
            return {"user_id": n_users, "item_id": n_items}[f]

        def get_user_feature(self):
            return Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(rng.integers(0, 2, n_users).astype(np.float32))})

        def inter_matrix(self, form="coo", value_field=None):
            return sp.coo_matrix((tr, (tu, ti)), shape=(n_users, n_items))
    cfg = Config(model="FairGo_GCN", config_dict={
        "embedding_size": D, "n_layers": 2, "dis_hidden_size_list": [8, 4], "filter_hidden_size_list": [16, 8],
        "sst_attr_list": ["gender"], "aggr_method": "WAP", "fair_weight": 0.1, "hidden_channels": 8, "gcn_n_layers": 2,
        "gcn_dropout": 0.0, "gcn_act": "relu", "device": "cuda", "load_pretrain_weight": False, "activation": "leakyrelu"})
    model = FairGo_GCN(cfg, DS()).to("cuda").train()
    model.train_stage = "pretrain"
    a_hat = OF.gcn_a_hat(n_users, n_items, tu, ti, tr)
    np.testing.assert_allclose(gcn_norm_matrix(n_users, n_items, DS().inter_matrix()).toarray(), a_hat.numpy(), rtol=1e-6, atol=1e-7)
    U0, I0 = model.user_embedding_layer.weight.data.cpu().clone(), model.item_embedding_layer.weight.data.cpu().clone()
    Ws = [c.lin.weight.data.cpu().clone() for c in model.gcn.convs]
    bs = [c.bias.data.cpu().clone() for c in model.gcn.convs]
    assert [tuple(w.shape) for w in Ws] == [(8, D), (D, 8)]
    sel = rng.integers(0, 200, size=(T, B))
    users, items = torch.from_numpy(tu[sel]), torch.from_numpy(ti[sel])
    ratings = torch.from_numpy(tr[sel])
    ref_loss, Ur, Ir, Wr, br = OF.gcn_pretrain_steps(U0, I0, Ws, bs, a_hat, users, items, ratings, 1e-2, 1e-4)
    opt = FusedLazyAdam(model.hip_engine(), lr=1e-2, weight_decay=1e-4, group="pretrain")
    losses = []
    for t in range(T):
        inter = Interaction({"user_id": users[t], "item_id": items[t], "rating": ratings[t]}).to("cuda")
        opt.zero_grad()
        loss = model.calculate_loss(inter, None)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    np.testing.assert_allclose(losses, ref_loss, rtol=1e-4)
    for got, ref in ((model.user_embedding_layer.weight, Ur), (model.item_embedding_layer.weight, Ir),
                     (model.gcn.convs[0].lin.weight, Wr[0]), (model.gcn.convs[1].lin.weight, Wr[1]),
                     (model.gcn.convs[0].bias, br[0]), (model.gcn.convs[1].bias, br[1])):
        a, b = got.data.cpu().numpy(), ref.numpy()
        assert (np.abs(a - b) <= 2e-4 * np.abs(b) + 2e-6).all(), np.abs(a - b).max()
    model.eval()
    with torch.no_grad():          # predict in the pretrain stage also goes through the GCN (dropout off)
        pr = model.predict(Interaction({"user_id": users[0], "item_id": items[0]}).to("cuda")).cpu()
    E = OF.gcn_forward(torch.cat([Ur, Ir], 0), a_hat, Wr, br)
    ref_pr = torch.clamp((E[users[0]] * E[items[0] + n_users]).sum(-1), 0, 5.0) / 5.0
    np.testing.assert_allclose(pr.numpy(), ref_pr.numpy(), rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("aggr,n_layers,D", [("WAP", 2, 64), ("LBA", 2, 128), ("LVA", 2, 16), ("WAP", 3, 64), ("WAP", 1, 64)])
def test_frontier_restricted_propagation_equals_whole_table(aggr, n_layers, D):
    """SURVEY.md section 7 hard part 3 (fairgo_pmf.py:196-216): a filter step needs H_l = L H_(l-1) only on the batch's users
    and, below the top layer, on what those rows read.  fr_spmm_csr_sel computes exactly those rows, term for term in the
    whole-table product's order, so the loss must be the SAME BITS and every filter / discriminator gradient equal up to
    the order in which autograd adds the (identical) contributions to dLoss/dE."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.utils import get_model
    n_users, n_items, B = 3001, 801, 160
    rng = np.random.default_rng(4)
    torch.manual_seed(4)
    tu = np.repeat(np.arange(1, n_users), 6)
    ti = rng.integers(1, n_items, tu.size)
    pair = np.unique(tu.astype(np.int64) * n_items + ti)
    tu, ti = pair // n_items, pair % n_items
    tr = rng.integers(1, 6, tu.size).astype(np.float32)
    gender = rng.integers(0, 2, n_users).astype(np.float32)
    z = {"gender": gender, "age": np.zeros(n_users, dtype=np.int64), "train_user": tu, "train_item": ti, "train_rating": tr}
    u = rng.integers(1, n_users, B)
    u[:8] = u[8:16]                                     # duplicate users in the batch
    sel = np.array([rng.choice(np.flatnonzero(tu == x)) for x in u])
    inter = Interaction({"user_id": torch.tensor(u), "item_id": torch.tensor(ti[sel]), "rating": torch.tensor(tr[sel]),
                         "gender": torch.tensor(gender[u])}).to("cuda")
    results = {}
    for mode in (False, True, "separate"):
        # "separate": the frontier again, with the top activation's derivative as the filter's own whole-table pass
        # (FAIRREC_FAIRGO_ACT_SEPARATE=1) instead of inside the table gradient's two launches (fr_spmm_csr_sel_act +
        # fr_row_scatter_add_act): the same products in the same order, so the same BITS in every gradient
        separate, mode = mode == "separate", bool(mode)
        if separate:
            os.environ["FAIRREC_FAIRGO_ACT_SEPARATE"] = "1"
        else:
            os.environ.pop("FAIRREC_FAIRGO_ACT_SEPARATE", None)
        torch.manual_seed(11)
        cfg = Config(model="FairGo_PMF", config_dict={
            "embedding_size": D, "sst_attr_list": ["gender"], "aggr_method": aggr, "n_layers": n_layers,
            "filter_hidden_size_list": [32, 16], "dis_hidden_size_list": [16, 8], "vs_weights": [3.0, 1.0, 1.0][:n_layers],
            "fair_weight": 0.1, "device": "cuda", "fairgo_frontier": mode})
        model = get_model("FairGo_PMF")(cfg, _DS(n_users, n_items, z)).to("cuda")
        model.FRONTIER_MAX_SHARE = 1.0                  # (this small graph's 2-hop frontier is most of it: restrict anyway)
        model.train_stage = "finetune"
        assert model.use_frontier() is mode and model.step_capturable("calculate_loss") is (not mode)
        assert model.step_capturable("calculate_dis_loss")
        assert model._top_fold_ok(["gender"]) is (not separate)
        params = list(model.filter_layer_dict["gender"].parameters()) + list(model.dis_layer_dict["gender"].parameters()) + \
            (list(model.aggr_layer.parameters()) if aggr == "LBA" else [])
        loss = model.calculate_loss(inter, ["gender"])
        loss.backward()
        with torch.no_grad():
            dis = model.calculate_dis_loss(inter, ["gender"])
        results["separate" if separate else mode] = (loss.detach().clone(), dis.clone(), [p.grad.clone() for p in params])
        if mode and not separate:
            fr = model._frontier(inter["user_id"])
            assert len(fr) == n_layers and fr[-1][0].numel() == len(set(u.tolist()))
            if n_layers > 1:
                assert fr[0][0].numel() > fr[-1][0].numel()
            # the bitmap kernels (csrc/frontier.hip) against the same sets built with stock torch ops: row lists, rank maps, bitmaps
            ref = model._frontier_torch(inter["user_id"])
            assert len(ref) == len(fr)
            for (r0, p0, b0), (r1, p1, b1) in zip(fr, ref):
                assert r0.dtype == r1.dtype == torch.int32 and torch.equal(r0, r1)
                assert torch.equal(p0, p1) and torch.equal(b0, b1)
        model.hip_engine().check_device_errors()
    os.environ.pop("FAIRREC_FAIRGO_ACT_SEPARATE", None)
    (l1, d1, g1), (l2, d2, g2) = results[True], results["separate"]
    assert torch.equal(l1, l2) and torch.equal(d1, d2)
    for k, (a, b) in enumerate(zip(g1, g2)):
        assert torch.equal(a, b), (k, float((a - b).abs().max()))
    (l0, d0, g0), (l1, d1, g1) = results[False], results[True]
    assert torch.equal(l0, l1), (float(l0), float(l1))
    assert torch.equal(d0, d1)
    for a, b in zip(g0, g1):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-5 * scale + 1e-12, (float((a - b).abs().max()), scale)
    assert any(float(g.abs().max()) > 0 for g in g0)
