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
