"""GPU: the plugin surface end to end (Config -> FOCF model -> Trainer.fit) against the oracle's step loop."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(tmp_path, objective="value", epochs=2, bs=200):
    from fairrec.config import Config
    from fairrec.data.dataloader import TrainDataLoader
    from fairrec.data.dataset import synthetic_dataset
    from fairrec.utils import get_model, get_trainer, init_seed
    cfg = Config(model="FOCF", config_dict={
        "train_batch_size": bs, "embedding_size": 16, "fair_objective": objective, "fair_weight": 0.7,
        "epochs": epochs, "device": "cuda", "checkpoint_dir": str(tmp_path), "learning_rate": 1e-3})
    init_seed(3)
    ds = synthetic_dataset(cfg, 300, 120, 2000, seed=11)
    train = TrainDataLoader(cfg, ds, shuffle=False)
    model = get_model("FOCF")(cfg, ds).to(cfg["device"])
    trainer = get_trainer(None, "FOCF")(cfg, model)
    return cfg, ds, train, model, trainer


def test_fit_matches_oracle_loss_curve_and_weights(tmp_path):
    from oracle import focf as O
    cfg, ds, train, model, trainer = _setup(tmp_path)
    U0 = model.user_embedding_layer.weight.detach().cpu().numpy().copy()
    I0 = model.item_embedding_layer.weight.detach().cpu().numpy().copy()
    batches = [b for b in train]
    T = len(batches)
    ragged = [len(b) for b in batches]
    assert ragged[-1] != ragged[0] or T * ragged[0] == 2000   # last batch may be short: ragged input path
    best, _ = trainer.fit(train, valid_data=None, verbose=False, saved=False)
    # oracle on the same 2 epochs of batches
    seq = batches * 2
    ref_losses, U, I = [], torch.tensor(U0, requires_grad=True), torch.tensor(I0, requires_grad=True)
    opt = torch.optim.Adam([U, I], lr=1e-3, weight_decay=1e-3)
    for b in seq:
        opt.zero_grad()
        l, _ = O.loss("value", 0.7, U, I, b["user_id"], b["item_id"], b["rating"], b["gender"])
        ref_losses.append(l.item())
        l.backward()
        opt.step()
    got = [trainer.train_loss_dict[e] for e in range(2)]
    want = [sum(ref_losses[:T]), sum(ref_losses[T:])]
    np.testing.assert_allclose(got, want, rtol=1e-4)
    sd = model.state_dict()      # flushes the lazy tables
    for name, ref in (("user_embedding_layer.weight", U), ("item_embedding_layer.weight", I)):
        a, b = sd[name].cpu().numpy(), ref.detach().numpy()
        assert (np.abs(a - b) <= 1e-4 * np.abs(b) + 1e-6).all(), name
    # optimizer state in torch.optim.Adam's layout
    st = trainer.optimizer.state_dict()["state"]["user_embedding_layer.weight"]
    np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), opt.state[U]["exp_avg"].numpy(), rtol=1e-4, atol=1e-7)
    assert int(st["step"]) == 2 * T


def _fit_pair(tmp_path, per_call, objective="value", n_inter=4100, bs=256, epochs=2, wd=1e-3):
    from fairrec.config import Config
    from fairrec.data.dataloader import TrainDataLoader
    from fairrec.data.dataset import synthetic_dataset
    from fairrec.utils import get_model, get_trainer, init_seed
    cfg = Config(model="FOCF", config_dict={
        "train_batch_size": bs, "embedding_size": 64, "fair_objective": objective, "fair_weight": 0.7, "weight_decay": wd,
        "epochs": epochs, "device": "cuda", "checkpoint_dir": str(tmp_path), "learning_rate": 1e-2,
        "train_steps_per_call": per_call, "lazy_adam_sweep_period": 4})
    init_seed(5)
    ds = synthetic_dataset(cfg, 700, 300, n_inter, seed=13).to("cuda")
    train = TrainDataLoader(cfg, ds, shuffle=True)
    model = get_model("FOCF")(cfg, ds).to("cuda")
    trainer = get_trainer(None, "FOCF")(cfg, model)
    return train, model, trainer


@pytest.mark.parametrize("objective,wd", [("value", 0.0), ("value", 1e-3), ("nonparity", 0.0)])
def test_epoch_issued_by_the_library_equals_the_per_batch_loop(tmp_path, objective, wd):
    """Trainer._train_epoch over a sliceable loader hands runs of `train_steps_per_call` batches to the library
    (TrainDataLoader.take -> FOCF.train_steps -> fr_focf_steps_many).  Same shuffles (one torch.randperm per epoch, the
    reference's consumer), same batches (a short last one included), same launches: without weight decay parameters,
    optimizer state and the epoch losses EQUAL the per-batch loop's (`train_steps_per_call: 0`) bit for bit; with it, up to
    the rounding of a replay cut in another place (tests/test_focf_hip.py::test_steps_many_equals_the_per_batch_staged_loop).
    nonparity needs a batch-wide value between loss and update: the trainer must notice and take the per-batch loop itself."""
    out = []
    for per_call in (0, 5, 256):
        train, model, trainer = _fit_pair(tmp_path, per_call, objective, wd=wd)
        calls = []
        eng = model.hip_engine()
        orig = eng.steps_many
        eng.steps_many = lambda *a, **k: calls.append(1) or orig(*a, **k)
        trainer.fit(train, valid_data=None, verbose=False, saved=False)
        n_steps = 2 * len(train)
        assert eng.U.step == n_steps
        want_calls = 0 if (per_call == 0 or objective == "nonparity") else 2 * -(-len(train) // per_call)
        assert len(calls) == want_calls
        sd = model.state_dict()
        out.append((sd["user_embedding_layer.weight"].clone(), sd["item_embedding_layer.weight"].clone(),
                    eng.U.m.clone(), eng.I.v.clone(), [trainer.train_loss_dict[e] for e in range(2)]))
    for o in out[1:]:
        for x, y in zip(o[:4], out[0][:4]):
            if wd == 0.0:
                assert torch.equal(x, y)
            else:
                torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-6 * float(y.abs().max()) + 1e-12)
        if wd == 0.0:
            assert o[4] == out[0][4]
        else:
            np.testing.assert_allclose(o[4], out[0][4], rtol=1e-5)


def test_focf_epoch_keeps_its_launch_and_sync_budget(tmp_path):
    """What the library-issued epoch must keep: about ONE launch per optimizer step (the step kernel; two stage launches
    per run of 256 steps on top) and ONE host synchronisation per epoch (the loss total read at its end)."""
    import warnings
    from fairrec import _C
    train, model, trainer = _fit_pair(tmp_path, 256, n_inter=256 * 300, epochs=1)
    trainer._train_epoch(train, 0)                     # allocations, first-use work
    torch.cuda.synchronize()
    _C.prof_reset()
    _C.prof_enable(True)
    torch.cuda.set_sync_debug_mode("warn")
    try:
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            trainer._train_epoch(train, 1)
    finally:
        torch.cuda.set_sync_debug_mode("default")
        _C.prof_enable(False)
    torch.cuda.synchronize()
    launches = sum(n for _, n in _C.prof_read().values())
    steps = len(train)
    assert steps == 300 and launches <= 1.2 * steps, _C.prof_read()
    syncs = [w for w in seen if "synchronizing" in str(w.message).lower()]
    # the epoch's shuffle index crossing to the device (a pageable copy, before anything is queued) and the read at its end
    # (loss total + device error word in one)
    assert len(syncs) <= 2, [str(w.message) for w in syncs]


def test_loss_backward_step_surface_and_checkpoint_roundtrip(tmp_path):
    cfg, ds, train, model, trainer = _setup(tmp_path, objective="absolute", epochs=1)
    b = next(iter(train)).to("cuda")
    trainer.optimizer.zero_grad()
    loss = model.calculate_loss(b)
    assert loss.dim() == 0 and loss.requires_grad and not torch.isnan(loss)
    loss.backward()                      # legal, as trainer.py:193 needs
    trainer.optimizer.step()
    p = model.predict(b)
    assert p.shape == (len(b),) and float(p.min()) >= 0 and float(p.max()) <= 1
    fs = model.full_sort_predict(b[:3])
    assert fs.shape == (3 * ds.item_num,)
    trainer._save_checkpoint(0, verbose=False)
    w = model.user_embedding_layer.weight.detach().clone()
    cfg2, ds2, train2, model2, trainer2 = _setup(tmp_path, objective="absolute", epochs=1)
    trainer2.resume_checkpoint(trainer.saved_model_file)
    assert torch.equal(model2.user_embedding_layer.weight, w)
    assert model2.hip_engine().U.step == 1 and trainer2.start_epoch == 1
    # both copies continue identically
    for m, t in ((model, trainer), (model2, trainer2)):
        m.calculate_loss(b).backward()
        t.optimizer.step()
    assert torch.equal(model.state_dict()["user_embedding_layer.weight"],
                       model2.state_dict()["user_embedding_layer.weight"])


def test_unsupported_paths_fail_loudly(tmp_path):
    from fairrec.config import Config
    from fairrec.data.dataset import synthetic_dataset
    from fairrec.trainer import Trainer
    from fairrec.utils import get_model
    cfg = Config(model="FOCF", config_dict={"device": "cuda", "learner": "sgd", "checkpoint_dir": str(tmp_path)})
    ds = synthetic_dataset(cfg, 30, 20, 100)
    model = get_model("FOCF")(cfg, ds).to("cuda")
    with pytest.raises(NotImplementedError):
        Trainer(cfg, model)
    with pytest.raises(ValueError):
        cfg2 = Config(model="FOCF", config_dict={"device": "cuda", "fair_objective": "bogus"})
        get_model("FOCF")(cfg2, ds)


def test_evaluate_load_best_model_after_a_non_saving_epoch(tmp_path):
    """evaluate(load_best_model=True) must score exactly the checkpointed weights (reference trainer.py:478-483), also
    when the epochs after the checkpoint left rows of the lazy-Adam tables behind the optimizer step: loading must not
    let their missed zero-gradient steps be replayed on top of the loaded weights."""
    cfg, ds, train, model, trainer = _setup(tmp_path, epochs=1)
    trainer._train_epoch(train, 0)
    trainer._save_checkpoint(0, verbose=False)
    trainer._train_epoch(train, 1)                 # an epoch that saves nothing: rows stay behind `step`
    eng = model.hip_engine()
    assert eng.U._dirty or eng.I._dirty
    ckpt = torch.load(trainer.saved_model_file, weights_only=False)
    trainer._load_for_eval(True, None)
    b = next(iter(train)).to("cuda")
    got = model.predict(b)
    fs = model.full_sort_predict(b[:4])
    cfg2, ds2, train2, fresh, _ = _setup(tmp_path, epochs=1)
    fresh.load_state_dict(ckpt["state_dict"])
    want = fresh.predict(b)
    assert torch.equal(got, want)
    assert torch.equal(fs, fresh.full_sort_predict(b[:4]))
    assert torch.equal(model.user_embedding_layer.weight, ckpt["state_dict"]["user_embedding_layer.weight"].to("cuda"))


GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _focf_from_fixture(z, tmp_path):
    from fairrec.config import Config
    from fairrec.data.dataset import synthetic_dataset
    from fairrec.utils import get_model, get_trainer
    lr, wd, fw = (float(x) for x in z["hyper"][:3])
    cfg = Config(model="FOCF", config_dict={"embedding_size": int(z["U0"].shape[1]), "fair_objective": "value",
                                            "fair_weight": fw, "learning_rate": lr, "weight_decay": wd, "device": "cuda",
                                            "checkpoint_dir": str(tmp_path), "epochs": 1})
    ds = synthetic_dataset(cfg, z["U0"].shape[0], z["I0"].shape[0], 100, seed=1)
    model = get_model("FOCF")(cfg, ds).to("cuda")
    return cfg, model, get_trainer(None, "FOCF")(cfg, model)


def test_resume_from_a_reference_written_checkpoint(tmp_path):
    """f-4: a checkpoint as the REFERENCE's Trainer._save_checkpoint writes it (trainer.py:221-240; fixture = its key names
    and tensors, tests/golden/gen_checkpoint_golden.py) must load through resume_checkpoint -- torch.optim.Adam's
    integer-keyed state included -- and the resumed run must continue like the reference did."""
    import json
    z = np.load(os.path.join(GOLDEN, "checkpoint_focf.npz"))
    cfg, model, trainer = _focf_from_fixture(z, tmp_path)
    sd_keys = json.loads(str(z["state_dict_keys"]))
    assert sd_keys == list(model.state_dict().keys())                    # same parameter names, same order
    opt_keys = json.loads(str(z["opt_state_keys"]))
    ck = {"config": None, "epoch": int(z["epoch"]), "cur_step": int(z["cur_step"]),
          "best_valid_score": float(z["best_valid_score"]), "other_parameter": json.loads(str(z["other_parameter"])),
          "state_dict": {k: torch.from_numpy(z["sd::" + k]) for k in sd_keys},
          "optimizer": {"state": {int(k): {n: torch.from_numpy(np.asarray(z[f"opt::{k}::{n}"])) for n in names}
                                  for k, names in opt_keys.items()},
                        "param_groups": json.loads(str(z["opt_param_groups"]))}}
    assert sorted(k for k in ck if k != "config") == json.loads(str(z["ck_keys"]))
    path = str(tmp_path / "FOCF-ref.pth")
    torch.save(ck, path)
    trainer.resume_checkpoint(path)
    assert trainer.start_epoch == int(z["epoch"]) + 1 and trainer.cur_step == int(z["cur_step"])
    eng = model.hip_engine()
    assert eng.U.step == eng.I.step == int(z["opt::0::step"]) == 3
    assert torch.equal(eng.U.m.cpu(), ck["optimizer"]["state"][0]["exp_avg"])
    assert torch.equal(eng.I.v.cpu(), ck["optimizer"]["state"][1]["exp_avg_sq"])
    t = z["user_id"].shape[0] - 1
    cols = [torch.tensor(z[k][t], device="cuda") for k in ("user_id", "item_id", "rating", "sst")]
    loss, _ = eng.forward(*cols)
    loss = float(loss[0])
    eng.backward_adam()
    eng.flush()
    assert abs(loss - float(z["loss_next"])) <= 1e-4 * abs(float(z["loss_next"]))
    for got, want in ((eng.U.weight, z["U_next"]), (eng.I.weight, z["I_next"])):
        a = got.cpu().numpy()
        assert (np.abs(a - want) <= 1e-4 * np.abs(want) + 1e-6).all()
    # ... and the other way: the optimizer state in torch's own layout, which a stock torch.optim.Adam (= the reference's
    # optimizer, trainer.py:139) loads as it is
    names = [n for n, _ in model.named_parameters()]
    sd = trainer.optimizer.state_dict(param_names=names)
    assert sorted(sd["state"]) == [0, 1] and sd["param_groups"][0]["params"] == [0, 1]
    for st in sd["state"].values():
        assert sorted(st) == ["exp_avg", "exp_avg_sq", "step"] and float(st["step"]) == 4.0
    ref_params = [torch.nn.Parameter(p.detach().cpu().clone()) for _, p in model.named_parameters()]
    ref_opt = torch.optim.Adam(ref_params, lr=1e-3, weight_decay=1e-3)
    ref_opt.load_state_dict({"state": {k: {n: v.cpu() if torch.is_tensor(v) else v for n, v in st.items()}
                                       for k, st in sd["state"].items()}, "param_groups": sd["param_groups"]})
    assert torch.equal(ref_opt.state[ref_params[0]]["exp_avg"], eng.U.m.cpu())


def test_focf_dataloader_device_resident_matches_reference_golden():
    """f-3: the item-complete batcher with the interaction table RESIDENT ON THE DEVICE (what the trainer feeds from)
    yields the reference's batches (tests/golden/gen_dataloader_golden.py ran focf_dataloader.py:37-51), row for row."""
    from fairrec.config import Config
    from fairrec.data.dataloader import FOCFDataLoader
    from fairrec.data.dataset import InteractionDataset
    from fairrec.data.interaction import Interaction
    z = np.load(os.path.join(GOLDEN, "dataloader_focf.npz"))
    c = Config(model="FOCF", config_dict={"train_batch_size": int(z["step"]), "device": "cuda"})
    inter = Interaction({"user_id": torch.from_numpy(z["user_id"]), "item_id": torch.from_numpy(z["item_id"]),
                         "rating": torch.from_numpy(z["rating"])})
    users = Interaction({"user_id": torch.arange(100), "gender": (torch.arange(100) % 2).float()})
    ds = InteractionDataset(c, inter, users, n_users=100, n_items=int(z["item_num"]))
    ds.to("cuda")
    dl = FOCFDataLoader(c, ds)
    # (seeded the way `init_seed` does: numpy AND the device mirrors of its stream -- in a process that holds a mirror, the
    # loader's picks are drawn inside host_numpy_stream(), i.e. from the one stream the device sampler shares)
    from fairrec.sampler.sampler import seed_all
    np.random.seed(int(z["np_seed"]))
    seed_all(int(z["np_seed"]))
    it = iter(dl)
    for b in range(4):
        batch = next(it)
        assert batch["user_id"].device.type == "cuda"
        for col, key in (("user", "user_id"), ("item", "item_id"), ("rating", "rating")):
            np.testing.assert_array_equal(batch[key].cpu().numpy(), z[f"batch{b}_{col}"])
        np.testing.assert_array_equal(batch["gender"].cpu().numpy(), (z[f"batch{b}_user"] % 2).astype(np.float32))


def test_full_sort_predict_values_match_the_oracle(tmp_path):
    """a6, value level (focf.py:171-178): clamp(U[user] @ I^T, 0, max) / max over ALL items, after training steps that leave
    rows of both lazy tables behind the optimizer step (no flush by the caller) -- against oracle.focf.predict on the
    reference's weights after the same steps (golden focf_value: `U_after*`, produced by the reference's own run)."""
    from fairrec.data.interaction import Interaction
    from oracle import focf as O
    z = np.load(os.path.join(GOLDEN, "focf_value.npz"))
    cfg, model, trainer = _focf_from_fixture(z, tmp_path)
    with torch.no_grad():
        model.user_embedding_layer.weight.copy_(torch.from_numpy(z["U0"]))
        model.item_embedding_layer.weight.copy_(torch.from_numpy(z["I0"]))
    T = max(int(s) for s in z["snaps"])
    for t in range(T):
        b = Interaction({"user_id": torch.tensor(z["user_id"][t]), "item_id": torch.tensor(z["item_id"][t]),
                         "rating": torch.tensor(z["rating"][t]), "gender": torch.tensor(z["sst"][t])}).to("cuda")
        trainer.optimizer.zero_grad()
        model.calculate_loss(b).backward()
        trainer.optimizer.step()
    eng = model.hip_engine()
    assert eng.U._dirty and eng.I._dirty                       # rows are behind: full_sort_predict has to see them caught up
    n_users, n_items = z["U0"].shape[0], z["I0"].shape[0]
    users = torch.tensor([1, 2, 5, n_users - 1, 7, 7], device="cuda")
    got = model.full_sort_predict(Interaction({"user_id": users})).cpu().numpy().reshape(len(users), n_items)
    Uref, Iref = torch.from_numpy(z[f"U_after{T}"]), torch.from_numpy(z[f"I_after{T}"])
    want = np.stack([O.predict(Uref, Iref, torch.full((n_items,), int(u)), torch.arange(n_items), 5.0).numpy()
                     for u in users.cpu()])
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6)
    assert 0.0 <= got.min() and got.max() <= 1.0 and (got > 0).any() and (got < 1).any()
    # ... and pair by pair it is what predict() gives for the same (user, item)
    pair = model.predict(Interaction({"user_id": users.repeat_interleave(n_items), "item_id": torch.arange(n_items, device="cuda").repeat(len(users))}))
    np.testing.assert_allclose(got.reshape(-1), pair.cpu().numpy(), rtol=1e-5, atol=1e-6)
    eng.check_device_errors()


def test_nan_loss_names_the_step_it_first_appeared_at(tmp_path):
    """trainer.py:192, :286-288 raise 'Training loss is nan' at the step whose loss is NaN; here the losses are summed on the
    device and read once per epoch -- the sum carries a sticky record of the FIRST bad step (fr_loss_accumulate / the one-launch
    steps' loss_acc[3:5]), so the same ValueError names the step the reference would have stopped at."""
    from fairrec import _C
    acc = torch.zeros(8, device="cuda")
    for x in ([1.0, 2.0], [3.0, float("nan")], [float("nan"), 1.0], [1.0, 1.0]):
        p = torch.tensor(x, device="cuda")
        _C.check(_C.lib().fr_loss_accumulate(p.data_ptr(), 2, acc.data_ptr(), _C.current_stream()), "fr_loss_accumulate")
    a = acc.cpu().tolist()
    assert a[3] == 4.0 and a[4] == 2.0 and a[0] != a[0] and a[1] != a[1]
    # the FOCF trainer loop (one launch per step, loss reduced by the next launch): a NaN rating in the 4th batch of the epoch
    cfg, ds, train, model, trainer = _setup(tmp_path, epochs=1)
    ds.inter_feat["rating"][3 * 200 + 5] = float("nan")
    with pytest.raises(ValueError, match=r"Training loss is nan \(first at step 4 of this pass\)"):
        trainer.fit(TrainDataLoader_(cfg, ds), valid_data=None, verbose=False, saved=False)
    # ... and a generic-engine model (NFCF: losses accumulated by fr_loss_accumulate, captured steps included)
    from fairrec.config import Config
    from fairrec.data.dataset import synthetic_dataset
    from fairrec.utils import get_model, get_trainer, init_seed
    cfg2 = Config(model="NFCF", config_dict={"train_batch_size": 128, "embedding_size": 16, "mlp_hidden_size": [16, 8], "dropout": 0.0,
                                             "epochs": 1, "device": "cuda", "checkpoint_dir": str(tmp_path), "neg_sampling": None,
                                             "load_pretrain_path": None})
    init_seed(3)
    ds2 = synthetic_dataset(cfg2, 200, 80, 1024, seed=5)
    ds2.inter_feat.update(type(ds2.inter_feat)({"label": (ds2.inter_feat["rating"] >= 3).float()}))
    ds2.inter_feat["label"][5 * 128 + 7] = float("nan")
    model2 = get_model("NFCF")(cfg2, ds2).to("cuda")
    trainer2 = get_trainer(None, "NFCF")(cfg2, model2)
    with pytest.raises(ValueError, match=r"first at step 6 of this pass"):
        trainer2.fit(TrainDataLoader_(cfg2, ds2), valid_data=None, verbose=False, saved=False)


def TrainDataLoader_(cfg, ds):
    from fairrec.data.dataloader import TrainDataLoader
    return TrainDataLoader(cfg, ds, shuffle=False)
