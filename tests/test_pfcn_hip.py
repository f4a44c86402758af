"""GPU parity: PFCN_PMF / PFCN_BiasedMF (filters + discriminators with BatchNorm on the MFMA kernels, BPR incl. the
[B,B] broadcast form, two optimizers) vs the reference's golden vectors, through the plugin surface."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CASES = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pfcn_*.npz"))
               if not p.endswith("_f64.npz"))   # <case>_f64.npz: the case's float64 companion (the reference in float64: gen_pfcn_golden.py::_run_f64)


class _DS:
    def __init__(self, n_users, n_items, z):
        from fairrec.data.interaction import Interaction
        self._n = {"user_id": n_users, "item_id": n_items}
        self._uf = Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(z["gender"]),
                                "age": torch.from_numpy(z["age"])})

    def num(self, f):
        return self._n[f]

    def get_user_feature(self):
        return self._uf


def _load_mlp(mlp, z, prefix):
    sd = {k[len(prefix) + 1:]: torch.tensor(z[k]) for k in (z.files if hasattr(z, "files") else z) if k.startswith(prefix + ".")}
    mlp.load_state_dict(sd)


@pytest.mark.parametrize("sharded", [False, True], ids=["single", "row_sharded"])
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_pfcn_training_matches_reference_golden(path, sharded, request):
    if sharded:     # the row-sharded engine as a 1-rank RCCL world: same goldens (fairrec/sharded_engine.py)
        request.getfixturevalue("rccl_world1")
    f64 = path[:-4] + "_f64.npz"
    _run_case(np.load(path), sharded, exact=np.load(f64) if os.path.exists(f64) else None)


def _run_case(z, sharded=False, noise=None, exact=None):
    """One recorded PFCN run (a golden .npz or a dict of the same layout) through the plugin surface on the GPU.
    noise: optional {key: absolute self-noise of the reference arithmetic on THIS case} (measured by running the CPU
    restatement with another reduction order), added to the tolerances."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.optim import FusedLazyAdam
    from fairrec.utils import get_model
    name, mode = str(z["model"]), str(z["mode"])
    attrs = [str(a) for a in z["attrs"]]
    lr, wd, dis_weight, p = (float(x) for x in z["hyper"])
    utab = "user_embedding" if name == "PFCN_MLP" else "user_embedding_layer"
    itab = "item_embedding" if name == "PFCN_MLP" else "item_embedding_layer"
    n_users, D = z[f"init.model.{utab}.weight"].shape
    n_items = z[f"init.model.{itab}.weight"].shape[0]
    cfg = Config(model=name, config_dict={"embedding_size": D, "sst_attr_list": attrs, "filter_mode": mode,
                                          "dis_hidden_size_list": [int(h) for h in z["dis_hidden"]], "dis_dropout": p,
                                          "dis_weight": dis_weight, "device": "cuda", "dropout": 0.0,
                                          "mlp_hidden_size_list": [8, 4], "num_layers": 2, "mlp_dropout": 0.0,
                                          "mlp_activation": "relu", "dis_activation": "leakyrelu", "activation": "leakyrelu",
                                          "row_sharded": sharded})
    model = get_model(name)(cfg, _DS(n_users, n_items, z))
    model.load_state_dict({k[11:]: torch.tensor(z[k]) for k in (z.files if hasattr(z, "files") else z)
                           if k.startswith("init.model.")})
    model = model.to("cuda")
    if mode != "none":
        for i, mlp in model.filter_layer.items():
            _load_mlp(mlp, z, f"init.filter.{i}")
        for s, mlp in model.dis_layer_dict.items():
            _load_mlp(mlp, z, f"init.dis.{s}")
    eng = model.hip_engine()
    # config clip_grad_norm (trainer.py:925-926): the optimizer clips inside step(), where the gradient exists
    clip = {"max_norm": float(z["clip_max_norm"])} if "clip_max_norm" in z else None
    if mode == "none":
        opt_f, opt_d = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=2, clip_grad_norm=clip), None
    else:
        opt_f = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=2, group="filter", clip_grad_norm=clip)
        opt_d = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=2, group="dis", clip_grad_norm=clip)
    n_dis = len(z["dis_hidden"]) + 1
    losses = []
    for t, ph in enumerate(str(x) for x in z["phases"]):
        u = z["user_id"][t]
        inter = Interaction({"user_id": torch.tensor(u), "item_id": torch.tensor(z["item_id"][t]),
                             "neg_item_id": torch.tensor(z["neg_item_id"][t]), "gender": torch.tensor(z["gender"][u]),
                             "age": torch.tensor(z["age"][u])}).to("cuda")
        sl = [s for s in str(z["sst_lists"][t]).split(",") if s] if mode != "none" else None
        if mode != "none":
            for s in sl:
                model.dis_layer_dict[s].forced_masks = [torch.tensor(z[f"mask.{s}.{t}.{l}"]) for l in range(n_dis)]
        opt = opt_f if ph == "F" else opt_d
        opt.zero_grad()
        loss = model.calculate_loss(inter, sl) if ph == "F" else model.calculate_dis_loss(inter, sl)
        losses.append(loss.detach().reshape(1).clone())
        loss.backward()
        opt.step()
        if clip and ph == "F":
            # the norm over model.parameters() = embeddings, biases, registered MLPs.  (On a "D" step the reference measures
            # the embeddings' gradient too -- the stale clipped one of the last filter step plus what the discriminator pass
            # added -- and scales it; the next filter step zeroes it unread and the discriminators are not in
            # model.parameters(), so that call changes no state and its norm is not reproduced here.)
            np.testing.assert_allclose(float(opt.last_grad_norm), z["grad_norm"][t], rtol=1e-4)
    noise = noise or {}
    got_loss = torch.cat(losses).cpu().numpy().astype(np.float64)
    assert (np.abs(got_loss - z["loss"]) <= 1e-4 * np.abs(z["loss"]) + 1e-6 + noise.get("loss", 0.0)).all(), \
        (got_loss, z["loss"], noise.get("loss"))
    sd = model.state_dict()
    steps = len(z["phases"])

    # |a - ref| <= 1e-4 |ref| + floor(kind of tensor).  The floors are the reference's OWN fp32 noise: the same torch ops
    # with 1 instead of 8 threads (another reduction order) move a tensor by this much (tests/golden/noise_floor.py,
    # measured on the full-size discriminators at D = 128 / 64), doubled; a tighter test would reject the reference
    # against itself.
    FLOOR = {"Linear weight": 2e-5,        # self-noise 8.7e-6 at scale 4e-2
             "BatchNorm gamma": 1e-6,      # 2.4e-7 at scale 1
             "BatchNorm beta": 2e-4,       # 1.0e-4 at scale 2e-3: Adam turns a cancelling column sum into +-lr-sized steps
             "BatchNorm running_var": 1e-6}
    worst, band_used = {}, {}

    def kind_of(k):
        parts = k.split(".")
        if len(parts) >= 2 and parts[-2].isdigit():
            lin = int(parts[-2]) % 4 == 1
            if parts[-1] == "weight":
                return "Linear weight" if lin else "BatchNorm gamma"
            if parts[-1] == "bias":
                return "Linear bias feeding BatchNorm" if lin else "BatchNorm beta"
            return "BatchNorm " + parts[-1]
        return "table"

    def close(a, ref, what, kind=None):
        a = a.detach().cpu().numpy().astype(np.float64)
        kind = kind or kind_of(what)
        if kind in ("Linear bias feeding BatchNorm", "BatchNorm running_mean"):
            # A Linear bias that feeds BatchNorm has an exactly-zero true gradient (the batch mean is subtracted again);
            # what reaches Adam is rounding noise whose SIGN Adam turns into +-lr steps (self-noise 3.5e-3 at lr 1e-3,
            # larger than the values), and running_mean averages it: not comparable, and neither can influence an
            # output.  What CAN be asserted: they moved by no more than Adam can move anything, lr per step.
            assert float(np.abs(a - ref).max()) <= 2 * steps * lr + 1e-6, (what, float(np.abs(a - ref).max()))
            return
        if kind == "BatchNorm num_batches_tracked":
            return
        floor = FLOOR.get(kind, 1e-6 * max(1.0, float(np.abs(ref).max())))      # tables: 1e-6 of the tensor's scale
        floor = floor + noise.get(what, 0.0)
        # An element must lie between the REFERENCE's fp32 execution (the golden) and the REFERENCE's float64 execution of the
        # same steps (<case>_f64.npz, written by tests/golden/gen_pfcn_golden.py::_run_f64 from the reference's own model
        # code under _refshim.float64_reference), give or take the tolerance: the HIP GEMMs are closer to exact than
        # torch's CPU sgemm (scratch/linear_acc.py), and on the few dozen Linear-weight elements whose gradient nearly
        # cancels the reference's own rounding has carried ITS fp32 run up to 9e-5 from its float64 run
        # (tests/golden/noise_floor.py) -- exactly the elements and the distance at which this path differs from the golden.
        ref64 = exact["final." + what] if exact is not None and ("final." + what) in exact.files else None
        if ref64 is None:
            dist = np.abs(a - ref)
        else:
            ref64 = ref64.astype(np.float64)
            lo, hi = np.minimum(ref, ref64), np.maximum(ref, ref64)
            dist = np.maximum(np.maximum(lo - a, a - hi), 0.0)
            # how many elements NEED the band (outside the plain tolerance around the fp32 golden, inside the band): they
            # must be the exception -- at most 0.5 % of a tensor, never fewer than 4 allowed (measured: <= 0.2 %)
            used = (np.abs(a - ref) > 1e-4 * np.abs(ref) + floor) & (dist <= 1e-4 * np.abs(ref) + floor)
            band_used[what] = int(used.sum())
            assert used.sum() <= max(4, 0.005 * used.size), (what, kind, int(used.sum()), used.size)
        ratio = dist / (1e-4 * np.abs(ref) + floor)
        if noise and kind != "table" and ratio.max() > 1.0:
            # Self-noise runs only (synthetic full-batch cases): an element whose gradient cancels to rounding noise takes
            # Adam steps of +-lr by the SIGN of that noise (the first step is lr * g / |g|) -- in the reference as well, on
            # other elements from run to run.  Such elements may be off by whole steps; they must be few.
            out = ratio > 1.0
            assert out.sum() <= max(2, 0.005 * out.size) and float(np.abs(a - ref)[out].max()) <= 2 * steps * lr + 1e-6, \
                (what, kind, float(out.mean()), float(np.abs(a - ref).max()))
            ratio = np.where(out, 0.0, ratio)
        worst[kind] = max(worst.get(kind, 0.0), float(ratio.max()))
        assert ratio.max() <= 1.0, (what, kind, float(np.abs(a - ref).max()), float(ratio.max()), int((ratio > 1.0).sum()),
                                    ratio.size)

    for k, v in sd.items():
        close(v, z["final.model." + k], "model." + k, kind=None if ".mlp_layers." in k else "table")
    if mode != "none":
        for i, mlp in model.filter_layer.items():
            for k, v in mlp.state_dict().items():
                if not k.endswith("num_batches_tracked"):
                    close(v, z[f"final.filter.{i}.{k}"], f"filter.{i}.{k}")
        for s, mlp in model.dis_layer_dict.items():
            for k, v in mlp.state_dict().items():
                if not k.endswith("num_batches_tracked"):
                    close(v, z[f"final.dis.{s}.{k}"], f"dis.{s}.{k}")
    print("worst |err| / tolerance per kind:", {k: round(v, 3) for k, v in worst.items()})
    print("elements that needed the fp32..float64 band:", {k: v for k, v in band_used.items() if v} or "none")
    eng.check_device_errors()
    pr = model.predict(inter, attrs if mode != "none" else None).cpu().numpy()
    np.testing.assert_allclose(pr, z["predict_last"], rtol=1e-4, atol=1e-6 + noise.get("predict_last", 0.0))


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipGraph"])
def test_pfcn_trainer_alternating_schedule(tmp_path, graph):
    """PFCNTrainer: mask draw per epoch, filter pass every `train_epoch_interval` epochs, discriminator pass always;
    with `graph_train_step` every (phase, attribute subset) runs as replays of its own captured step."""
    from fairrec.config import Config
    from fairrec.data.dataloader import TrainDataLoader
    from fairrec.data.dataset import InteractionDataset
    from fairrec.data.interaction import Interaction
    from fairrec.utils import get_model, get_trainer, init_seed
    init_seed(7)
    n_users, n_items, n = 60, 50, 600
    g = torch.Generator().manual_seed(1)
    inter = Interaction({"user_id": torch.randint(1, n_users, (n,), generator=g),
                         "item_id": torch.randint(1, n_items, (n,), generator=g),
                         "neg_item_id": torch.randint(1, n_items, (n,), generator=g)})
    users = Interaction({"user_id": torch.arange(n_users), "gender": (torch.rand(n_users, generator=g) < 0.5).float(),
                         "age": torch.randint(0, 3, (n_users,), generator=g)})
    users["age"][1:4] = torch.tensor([0, 1, 2])
    cfg = Config(model="PFCN_BiasedMF", config_dict={
        "embedding_size": 16, "sst_attr_list": ["gender", "age"], "filter_mode": "cm", "dis_hidden_size_list": [16, 8],
        "train_batch_size": 120, "epochs": 3, "train_epoch_interval": 2, "device": "cuda", "checkpoint_dir": str(tmp_path),
        "graph_train_step": graph})
    ds = InteractionDataset(cfg, inter, users, n_users, n_items)
    model = get_model("PFCN_BiasedMF")(cfg, ds).to("cuda")
    trainer_cls = get_trainer(None, "PFCN_BiasedMF")
    assert trainer_cls.__name__ == "PFCN_BiasedMFTrainer"
    trainer = trainer_cls(cfg, model)
    trainer.fit(TrainDataLoader(cfg, ds, shuffle=True), valid_data=None, verbose=False, saved=True)
    l = trainer.train_loss_dict
    assert set(l) == {0, 1, 2} and all(np.isfinite(v) for v in l.values())
    eng = model.hip_engine()
    eng.sync_steps()                         # graph mode keeps the step counters on the device
    steps_per_epoch = 5                      # 600 / 120
    assert eng._tables["user_embedding_layer.weight"].step == 2 * steps_per_epoch          # filter epochs 0 and 2
    dis_steps = {d.step for k, d in eng._dense.items() if k.startswith("dis.")}
    # a discriminator steps in the epochs whose random mask selected its attribute (at least one attribute per epoch)
    assert all(s % steps_per_epoch == 0 and s <= 3 * steps_per_epoch for s in dis_steps) and max(dis_steps) >= steps_per_epoch
    assert eng._dense["global_bias"].step == 2 * steps_per_epoch    # zero gradient, but present: it still steps (App. B-1)
    ck = torch.load(trainer.saved_model_file, weights_only=False)
    assert {"optimizer_filter", "optimizer_dis", "state_dict"} <= set(ck)
    # resume (trainer.py:1156-1186 as intended): both optimizers' lazy state comes back, training continues from there
    init_seed(7)
    model2 = get_model("PFCN_BiasedMF")(cfg, ds).to("cuda")
    trainer2 = trainer_cls(cfg, model2)
    trainer2.resume_checkpoint(trainer.saved_model_file)
    assert trainer2.start_epoch == ck["epoch"] + 1
    eng2 = model2.hip_engine()
    eng2.sync_steps()
    t2 = eng2._tables["user_embedding_layer.weight"]
    assert t2.step == int(ck["optimizer_filter"]["state"]["user_embedding_layer.weight"]["step"]) > 0
    assert torch.equal(t2.m, ck["optimizer_filter"]["state"]["user_embedding_layer.weight"]["exp_avg"].to("cuda"))
    assert torch.equal(model2.user_embedding_layer.weight, ck["state_dict"]["user_embedding_layer.weight"].to("cuda"))
    for k, st in ck["optimizer_dis"]["state"].items():
        assert eng2._dense[k].step == int(st["step"])
    cfg_none = Config(model="PFCN_BiasedMF", config_dict={
        "embedding_size": 16, "sst_attr_list": ["gender", "age"], "filter_mode": "none", "train_batch_size": 120,
        "epochs": 1, "device": "cuda", "checkpoint_dir": str(tmp_path / "none")})
    os.makedirs(str(tmp_path / "none"), exist_ok=True)
    m3 = get_model("PFCN_BiasedMF")(cfg_none, ds).to("cuda")
    t3 = trainer_cls(cfg_none, m3)
    t3.fit(TrainDataLoader(cfg_none, ds, shuffle=True), valid_data=None, verbose=False, saved=True)
    m4 = get_model("PFCN_BiasedMF")(cfg_none, ds).to("cuda")
    t4 = trainer_cls(cfg_none, m4)
    t4.resume_checkpoint(t3.saved_model_file)
    assert m4.hip_engine()._tables["user_embedding_layer.weight"].step == 5


@pytest.mark.parametrize("mode", ["sm", "cm"])
def test_pfcn_biasedmf_full_batch_at_the_baseline_width(mode):
    """BASELINE.json configs[2]'s step shape -- PFCN_BiasedMF, embedding_size 128, the full-size discriminator
    [128, 256, 128, 128, 64, 32] with dropout 0.3, B = 8192 (the [B,B] BPR broadcast: 67 M terms) -- on tables scaled down to
    what the CPU oracle's dense Adam sweeps in seconds: filter / discriminator / filter steps of the HIP path against
    oracle/pfcn.py (pinned to the reference's goldens), every tensor compared."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import pfcn as O
    from fairrec.config import Config
    from fairrec.utils import get_model
    n_users, n_items, D, B = 100_001, 20_001, 128, 8192
    hidden, p = (128, 256, 128, 128, 64, 32), 0.3
    attrs = ["gender"] if mode == "sm" else ["gender", "age"]
    phases, lists = "FDF", [attrs, attrs, attrs[:1]]
    g = torch.Generator().manual_seed(31)
    z = {"model": np.array("PFCN_BiasedMF"), "mode": np.array(mode), "attrs": np.array(attrs), "dis_hidden": np.array(hidden),
         "hyper": np.array([1e-3, 1e-4, 10.0, p]), "gender": (torch.rand(n_users, generator=g) < 0.5).float().numpy(),
         "age": torch.randint(0, 3, (n_users,), generator=g).numpy(), "phases": np.array(list(phases)),
         "sst_lists": np.array([",".join(s) for s in lists])}
    torch.manual_seed(9)
    cfg = Config(model="PFCN_BiasedMF", config_dict={"embedding_size": D, "sst_attr_list": attrs, "filter_mode": mode,
                                                     "dis_hidden_size_list": list(hidden), "dis_dropout": p, "dis_weight": 10.0,
                                                     "device": "cpu", "activation": "leakyrelu", "dis_activation": "leakyrelu"})
    m0 = get_model("PFCN_BiasedMF")(cfg, _DS(n_users, n_items, z))
    for k, v in m0.state_dict().items():
        z["init.model." + k] = (v * 0.1 if "embedding" in k else v).detach().numpy().copy()
    for i, mlp in m0.filter_layer.items():
        for k, v in mlp.state_dict().items():
            z[f"init.filter.{i}.{k}"] = v.detach().numpy().copy()
    for s, mlp in m0.dis_layer_dict.items():
        for k, v in mlp.state_dict().items():
            z[f"init.dis.{s}.{k}"] = v.detach().numpy().copy()
    T = len(phases)
    z["user_id"] = torch.randint(1, n_users, (T, B), generator=g).numpy()
    z["item_id"] = torch.randint(1, n_items, (T, B), generator=g).numpy()
    z["neg_item_id"] = torch.randint(1, n_items, (T, B), generator=g).numpy()
    sizes = [D] + list(hidden)
    for t in range(T):
        for s in lists[t]:
            for l, w in enumerate(sizes):
                z[f"mask.{s}.{t}.{l}"] = (torch.rand(B, w, generator=g) >= p).float().numpy()
    ref = O.train(z)
    # the reference's own fp32 noise on this case: the same ops with one thread (another reduction order).  After a
    # discriminator step the 8192-row BatchNorm column sums differ in the last bits, Adam turns that into +-lr-sized moves
    # of betas and biases, and the next loss moves by ~6e-4 relative (x dis_weight 10) -- in the reference itself.
    alts = []
    for nt in (1, 2, 4):
        torch.set_num_threads(nt)
        alts.append(O.train(z))
    torch.set_num_threads(8)
    # (a handful of runs are samples of a sign-flip process, not a bound: allow twice their spread around the reference)
    def spread(k):
        vals = np.stack([np.asarray(o[k], dtype=np.float64) for o in alts + [ref]])
        return 2.0 * (vals.max(0) - vals.min(0))
    noise = {k[6:] if k.startswith("final.") else k: float(spread(k).max()) for k in ref if k != "loss"}
    noise["loss"] = spread("loss")
    z.update(ref)
    _run_case(z, sharded=False, noise=noise)


def test_bpr_outer_rect_matches_float64_torch():
    """fr_bpr_outer_rect (rows c x columns a of PFCN_BiasedMF's broadcast loss, the piece a row-sharded step evaluates of
    the GLOBAL batch's matrix) against the term matrix in float64; Na != Nc, neither a multiple of the tile sizes."""
    from fairrec.sharded_engine import HipTableOps
    g = torch.Generator().manual_seed(4)
    Na, Nc = 1000, 2300
    a, c = torch.randn(Na, generator=g), torch.randn(Nc, generator=g) * 0.5
    inv = 1.0 / (Na * Nc)
    x = (a[None, :] + c[:, None]).double()
    sig = torch.sigmoid(x)
    d = -(sig * (1 - sig)) / (1e-10 + sig) * inv
    ops = HipTableOps()
    loss, da, dc = torch.zeros(1, device="cuda"), torch.zeros(Na, device="cuda"), torch.zeros(Nc, device="cuda")
    hold = {}
    ops.bpr_outer_rect(a.cuda(), c.cuda(), inv, loss, da, dc, hold)
    np.testing.assert_allclose(float(loss), float((-torch.log(1e-10 + sig)).sum() * inv), rtol=2e-5)
    np.testing.assert_allclose(da.cpu().numpy(), d.sum(0).float().numpy(), rtol=2e-4, atol=1e-9)
    np.testing.assert_allclose(dc.cpu().numpy(), d.sum(1).float().numpy(), rtol=2e-4, atol=1e-9)
    da2 = torch.zeros(Na, device="cuda")
    ops.bpr_outer_rect(a.cuda(), c.cuda(), inv, loss, da2, None, hold)        # either gradient may be left out
    assert torch.equal(da, da2)


@pytest.mark.parametrize("switch", ["FAIRREC_PFCN_ROWDOT_SEPARATE", "FAIRREC_APPLY_SEPARATE"], ids=["row_dots", "apply_grad"])
@pytest.mark.parametrize("name", ["PFCN_BiasedMF", "PFCN_PMF"])
def test_paired_row_dots_equal_the_two_separate_calls_bit_for_bit(name, switch, monkeypatch):
    """(`apply_grad`: the same comparison for the optimizer step's table launches -- the user and the item table, and their two
    bias columns, share a launch each (fr_table_apply_grad_two, each table with its own id count) unless
    FAIRREC_APPLY_SEPARATE is set.)  calculate_loss scores the positive and the negative item rows in one launch each way (functional.RowDotPair) and
    hands autograd the user rows' two gradients unsummed; the pair of RowDot calls it replaces
    (FAIRREC_PFCN_ROWDOT_SEPARATE=1, the reference's two torch.mul(u, i).sum(-1) nodes, pfcn_pmf.py:182-183) must give the
    same bits: every loss, every table row and every filter / discriminator parameter after filter, discriminator and
    filter steps (dropout off so that the two runs see the same masks)."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.optim import FusedLazyAdam
    from fairrec.utils import get_model
    n_users, n_items, D, B = 3001, 801, 128, 1024
    g = torch.Generator().manual_seed(5)
    z = {"gender": (torch.rand(n_users, generator=g) < 0.5).float().numpy(), "age": torch.randint(0, 3, (n_users,), generator=g).numpy()}
    batches = [(torch.randint(1, n_users, (B,), generator=g), torch.randint(1, n_items, (B,), generator=g),
                torch.randint(1, n_items, (B,), generator=g)) for _ in range(3)]

    def run(separate):
        if separate:
            monkeypatch.setenv(switch, "1")
        else:
            monkeypatch.delenv(switch, raising=False)
        torch.manual_seed(17)
        cfg = Config(model=name, config_dict={"embedding_size": D, "sst_attr_list": ["gender"], "filter_mode": "sm",
                                              "dis_hidden_size_list": [32, 16], "dis_dropout": 0.0, "dis_weight": 10.0,
                                              "device": "cuda", "activation": "leakyrelu", "dis_activation": "leakyrelu"})
        model = get_model(name)(cfg, _DS(n_users, n_items, z)).to("cuda")
        eng = model.hip_engine()
        opt_f = FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-4, sweep_period=2, group="filter")
        opt_d = FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-4, sweep_period=2, group="dis")
        losses = []
        for t, ph in enumerate("FDF"):
            u, i, n = batches[t]
            inter = Interaction({"user_id": u, "item_id": i, "neg_item_id": n, "gender": torch.tensor(z["gender"][u.numpy()]),
                                 "age": torch.tensor(z["age"][u.numpy()])}).to("cuda")
            opt = opt_f if ph == "F" else opt_d
            opt.zero_grad()
            loss = model.calculate_loss(inter, ["gender"]) if ph == "F" else model.calculate_dis_loss(inter, ["gender"])
            losses.append(float(loss))
            loss.backward()
            opt.step()
        out = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        for i, mlp in model.filter_layer.items():
            out.update({f"filter.{i}.{k}": v.detach().cpu().clone() for k, v in mlp.state_dict().items()})
        for s, mlp in model.dis_layer_dict.items():
            out.update({f"dis.{s}.{k}": v.detach().cpu().clone() for k, v in mlp.state_dict().items()})
        return losses, out

    la, a = run(separate=True)
    lb, b = run(separate=False)
    assert la == lb, (la, lb)
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k].double() - b[k].double()).abs().max()))
