"""GPU parity: NFCF (lazy tables + fp32-MFMA scorer + BCE / differential-fairness head) vs the reference's golden
vectors, through the plugin surface (model.calculate_loss -> loss.backward() -> optimizer.step())."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CASES = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "nfcf_*.npz"))
               if not p.endswith("_f64.npz"))   # <case>_f64.npz: the REFERENCE's float64 execution of the case (gen_nfcf_golden.py)


class _DS:
    def __init__(self, n_users, n_items, gender):
        from fairrec.data.interaction import Interaction
        self._n = {"user_id": n_users, "item_id": n_items}
        self._uf = Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(gender)})

    def num(self, f):
        return self._n[f]

    def get_user_feature(self):
        return self._uf


@pytest.mark.parametrize("sharded", [False, True], ids=["single", "row_sharded"])
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_nfcf_training_matches_reference_golden(path, tmp_path, sharded, request):
    if sharded:     # the row-sharded engine as a 1-rank RCCL world: same goldens (fairrec/sharded_engine.py)
        request.getfixturevalue("rccl_world1")
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.model.fair_recommender.nfcf import NFCF
    f64 = path[:-4] + "_f64.npz"
    _run_case(np.load(path), sharded, exact=np.load(f64) if os.path.exists(f64) else None)


def _run_case(z, sharded=False, exact=None):
    """One recorded NFCF run (a golden .npz or a dict of the same layout) through the plugin surface on the GPU."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.model.fair_recommender.nfcf import NFCF
    from fairrec.optim import FusedLazyAdam
    stage = str(z["stage"])
    lr, wd, fw, p = (float(x) for x in z["hyper"])
    n_users, D = z["init.user_embedding.weight"].shape
    n_items = z["init.item_embedding.weight"].shape[0]
    cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": [int(h) for h in z["hidden"]],
                                            "dropout": p, "fair_weight": fw, "device": "cuda", "load_pretrain_path": None,
                                            "row_sharded": sharded})
    model = NFCF(cfg, _DS(n_users, n_items, z["gender"]))
    init = {k[5:]: torch.tensor(z[k]) for k in (z.files if hasattr(z, "files") else z) if k.startswith("init.")}
    if stage == "finetune":     # the state reset_params produced in the reference (its projection is pinned in the oracle test)
        model.load_pretrain_path = "reference-checkpoint"
        model.user_embedding.weight.requires_grad = False
    model.load_state_dict(init)
    model = model.to("cuda")
    model.train()
    # config clip_grad_norm (trainer.py:194-195): clipped inside step(), where the embedding gradient exists
    clip = {"max_norm": float(z["clip_max_norm"])} if "clip_max_norm" in z else None
    opt = FusedLazyAdam(model.hip_engine(), lr=lr, weight_decay=wd, sweep_period=3, clip_grad_norm=clip)
    n_layers = len(z["hidden"]) + 1
    snaps = set(int(s) for s in z["snaps"])
    losses = []
    for t in range(len(z["user_id"])):
        inter = Interaction({"user_id": torch.tensor(z["user_id"][t]), "item_id": torch.tensor(z["item_id"][t]),
                             "label": torch.tensor(z["label"][t]), "gender": torch.tensor(z["sst"][t])}).to("cuda")
        if p > 0:
            model.mlp_layers.forced_masks = [torch.tensor(z[f"mask{l}"][t]) for l in range(n_layers)]
        opt.zero_grad()
        loss = model.calculate_loss(inter)
        losses.append(loss.detach().reshape(1).clone())
        loss.backward()
        opt.step()
        if clip:
            np.testing.assert_allclose(float(opt.last_grad_norm), z["grad_norm"][t], rtol=1e-4)
        if (t + 1) in snaps:
            sd = model.state_dict()
            for k, v in sd.items():
                ref = z[f"after{t + 1}." + k]
                a = v.cpu().numpy()
                # between the reference's fp32 execution (the golden) and the reference's float64 execution of the same
                # steps (tests/golden/gen_nfcf_golden.py::_run_f64), give or take the tolerance -- see tests/test_pfcn_hip.py
                key = f"after{t + 1}." + k
                r64 = exact[key] if exact is not None and key in exact.files else ref
                lo, hi = np.minimum(ref, r64), np.maximum(ref, r64)
                dist = np.maximum(np.maximum(lo - a, a - hi), 0.0)
                tol = 1e-4 * np.abs(ref) + 2e-6
                used = (np.abs(a - ref) > tol) & (dist <= tol)          # elements that NEED the band: the exception
                assert used.sum() <= max(4, 0.005 * used.size), (k, t + 1, int(used.sum()), used.size)
                out = dist > tol
                # Adam divides a gradient by its own magnitude: where a weight gradient nearly cancels, ITS rounding noise
                # (different in every correct implementation) moves the element by a visible fraction of lr.  One element
                # per 10 000 (at least two) may therefore leave the band, by at most 0.25 % of what Adam can move anything
                # (steps * lr); measured: 1-2 of the 65 536 elements of the [128, 512] first-layer weight at D = 256, by
                # 6e-6 .. 9e-6 (which ones depends on the GEMM's summation order).
                assert out.sum() <= max(2, out.size // 10000) and (not out.any() or dist[out].max() <= 2.5e-3 * (t + 1) * lr), \
                    (k, t + 1, np.abs(a - ref).max(), np.abs(a - r64).max(), int(out.sum()))
    np.testing.assert_allclose(torch.cat(losses).cpu().numpy(), z["loss"], rtol=1e-4)
    model.hip_engine().check_device_errors()
    model.eval()
    model.mlp_layers.forced_masks = None
    with torch.no_grad():
        pr = model.predict(inter).cpu().numpy()
    np.testing.assert_allclose(pr, z["predict_last"], rtol=1e-4, atol=1e-6)


def test_optimizer_state_of_a_model_with_a_frozen_table_loads_into_stock_adam():
    """f-4: NFCF finetune freezes the user table (nfcf.py:62-63); the reference builds its Adam on ALL of model.parameters()
    (trainer.py:139) and torch maps saved state to parameters by POSITION, checking the group length -- so the state written
    for a reference `optimizer.load_state_dict` must list every parameter's index and keep state only for the stepped ones."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.model.fair_recommender.nfcf import NFCF
    from fairrec.optim import FusedLazyAdam
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "nfcf_finetune.npz"))
    lr, wd, fw, p = (float(x) for x in z["hyper"])
    n_users, D = z["init.user_embedding.weight"].shape
    n_items = z["init.item_embedding.weight"].shape[0]
    cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": [int(h) for h in z["hidden"]], "dropout": 0.0,
                                            "fair_weight": fw, "device": "cuda", "load_pretrain_path": None})
    model = NFCF(cfg, _DS(n_users, n_items, z["gender"]))
    model.load_pretrain_path = "reference-checkpoint"
    model.user_embedding.weight.requires_grad = False
    model.load_state_dict({k[5:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.")})
    model = model.to("cuda").train()
    opt = FusedLazyAdam(model.hip_engine(), lr=lr, weight_decay=wd, sweep_period=3)
    for t in range(2):
        inter = Interaction({"user_id": torch.tensor(z["user_id"][t]), "item_id": torch.tensor(z["item_id"][t]),
                             "label": torch.tensor(z["label"][t]), "gender": torch.tensor(z["sst"][t])}).to("cuda")
        opt.zero_grad()
        model.calculate_loss(inter).backward()
        opt.step()
    names = [n for n, _ in model.named_parameters()]
    sd = opt.state_dict(param_names=names)
    frozen = names.index("user_embedding.weight")
    assert sd["param_groups"][0]["params"] == list(range(len(names)))
    assert frozen not in sd["state"] and sorted(sd["state"]) == [k for k in range(len(names)) if k != frozen]
    ref_params = [torch.nn.Parameter(q.detach().cpu().clone(), requires_grad=q.requires_grad) for _, q in model.named_parameters()]
    ref_opt = torch.optim.Adam(ref_params, lr=lr, weight_decay=wd)
    ref_opt.load_state_dict({"state": {k: {n: (v.cpu() if torch.is_tensor(v) else v) for n, v in st.items()}
                                       for k, st in sd["state"].items()}, "param_groups": sd["param_groups"]})
    item = names.index("item_embedding.weight")
    eng = model.hip_engine()
    assert torch.equal(ref_opt.state[ref_params[item]]["exp_avg"], eng._tables["item_embedding.weight"].m.cpu())
    assert float(ref_opt.state[ref_params[item]]["step"]) == 2.0 and ref_params[frozen] not in ref_opt.state


def test_nfcf_full_batch_at_the_baseline_width():
    """BASELINE.json configs[4]'s step shape -- embedding_size 256, mlp_hidden_size [128, 64], B = 8192, finetune (user table
    frozen, differential-fairness term) and pretrain -- on tables scaled down to what the CPU oracle's dense Adam sweeps in
    seconds (200 001 x 50 001): three steps of the HIP path against oracle/nfcf.py (pinned to the reference's goldens),
    every row of both tables compared, also the rows no batch touched."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import nfcf as O
    from fairrec.config import Config
    from fairrec.model.fair_recommender.nfcf import NFCF
    n_users, n_items, D, B, T, hidden = 200_001, 50_001, 256, 8192, 3, (128, 64)
    g = torch.Generator().manual_seed(12)
    gender = (torch.rand(n_users, generator=g) < 0.5).float().numpy()
    for stage, p in (("finetune", 0.0), ("pretrain", 0.2)):
        torch.manual_seed(5)
        cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": list(hidden), "dropout": p,
                                                "fair_weight": 0.1, "device": "cpu", "load_pretrain_path": None})
        m0 = NFCF(cfg, _DS(n_users, n_items, gender))
        z = {"stage": np.array(stage), "hyper": np.array([1e-3, 1e-6, 0.1, p]), "hidden": np.array(hidden), "gender": gender,
             "snaps": np.array([T])}
        for k, v in m0.state_dict().items():
            z["init." + k] = (v * 0.1 if "embedding" in k else v).detach().numpy().copy()     # N(0,1) tables saturate the sigmoid
        u = torch.randint(1, n_users, (T, B), generator=g)
        i = torch.randint(1, n_items // 8, (T, B), generator=g)                  # items repeat: segments of several members
        r = torch.randint(1, 6, (T, B), generator=g).float()
        z.update(user_id=u.numpy(), item_id=i.numpy(), label=(r >= 3).float().numpy(), sst=gender[u.numpy()])
        if p > 0:
            sizes = [2 * D] + list(hidden)
            for l, w in enumerate(sizes):
                z[f"mask{l}"] = (torch.rand(T, B, w, generator=g) >= p).to(torch.uint8).numpy()
        ref = O.train(z, snaps=(T,))
        z.update({k: v for k, v in ref.items() if k.startswith("after") or k == "loss"})
        U, I = torch.tensor(ref[f"after{T}.user_embedding.weight"]), torch.tensor(ref[f"after{T}.item_embedding.weight"])
        n_l = len(hidden) + 1
        Ws = [torch.tensor(ref[f"after{T}.mlp_layers.mlp_layers.{3 * l + 1}.weight"]) for l in range(n_l)]
        bs = [torch.tensor(ref[f"after{T}.mlp_layers.mlp_layers.{3 * l + 1}.bias"]) for l in range(n_l)]
        z["predict_last"] = O.forward(U, I, Ws, bs, u[-1], i[-1]).numpy()
        _run_case(z, sharded=False)


@pytest.mark.parametrize("sharded", [False, True], ids=["single", "row_sharded"])
def test_reset_params_on_the_device_then_finetune(tmp_path, sharded, request):
    """NFCF.reset_params (nfcf.py:49-67) with the model on the GPU: the pre-trained checkpoint is loaded, the gender direction
    projected out of the user table on the device (row-sharded: the two group means through one all-reduce, here as a 1-rank
    RCCL world), the table frozen, the item table re-initialised -- against the state the reference's reset_params left
    (golden `init.*`), and the finetune steps then run from THAT state (not from a state loaded out of the golden) to the
    reference's first snapshot."""
    if sharded:
        request.getfixturevalue("rccl_world1")
    from fairrec.config import Config
    from fairrec.model.fair_recommender.nfcf import NFCF
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "nfcf_finetune.npz"))
    n_users, D = z["pretrain_user_embedding"].shape
    n_items = z["init.item_embedding.weight"].shape[0]
    ck = tmp_path / "pre.pth"
    torch.save({"state_dict": {"user_embedding.weight": torch.tensor(z["pretrain_user_embedding"])}}, ck)
    lr, wd, fw, p = (float(x) for x in z["hyper"])
    cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": [int(h) for h in z["hidden"]], "dropout": p,
                                            "fair_weight": fw, "device": "cuda", "load_pretrain_path": str(ck),
                                            "row_sharded": sharded})
    with torch.device("cuda"):
        model = NFCF(cfg, _DS(n_users, n_items, z["gender"]))
    model = model.to("cuda")
    assert model.user_embedding.weight.is_cuda and not model.user_embedding.weight.requires_grad
    assert model.item_embedding.weight.requires_grad
    np.testing.assert_allclose(model.user_embedding.weight.detach().cpu().numpy(), z["init.user_embedding.weight"],
                               rtol=1e-5, atol=1e-6)
    # the rest of the state the reference started its finetune stage from (fresh item table, scorer), then its steps
    init = {k[5:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.") and "user_embedding" not in k}
    model.load_state_dict(init, strict=False)
    from fairrec.data.interaction import Interaction
    from fairrec.optim import FusedLazyAdam
    model.train()
    opt = FusedLazyAdam(model.hip_engine(), lr=lr, weight_decay=wd, sweep_period=3)
    n_layers = len(z["hidden"]) + 1
    first = min(int(s) for s in z["snaps"])
    losses = []
    for t in range(first):
        inter = Interaction({"user_id": torch.tensor(z["user_id"][t]), "item_id": torch.tensor(z["item_id"][t]),
                             "label": torch.tensor(z["label"][t]), "gender": torch.tensor(z["sst"][t])}).to("cuda")
        if p > 0:
            model.mlp_layers.forced_masks = [torch.tensor(z[f"mask{l}"][t]) for l in range(n_layers)]
        opt.zero_grad()
        loss = model.calculate_loss(inter)
        losses.append(float(loss))
        loss.backward()
        opt.step()
    np.testing.assert_allclose(losses, z["loss"][:first], rtol=1e-4)
    for k, v in model.state_dict().items():
        ref = z[f"after{first}." + k]
        np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=2e-4, atol=1e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)


def test_global_df_kernels_match_their_cpu_doubles():
    """fr_nfcf_df_pack / _owner / _apply (the differential-fairness term of a row-sharded step on the GLOBAL batch) against
    the CPU doubles that tests/test_sharded_engine_gloo.py runs the exchange schedule with (and proves equal to the
    single-process step at 2 and 4 ranks), on a 4-rank layout: this process plays one requester and one owner, the records
    "received" are another draw of the same shape (no collective here; the 1-rank RCCL goldens run the whole path)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_sharded_engine_gloo import _Ops, _Table
    from fairrec.optim import AdamHyper, LazyTable
    from fairrec.sharded_engine import HipTableOps
    cpu, hip = _Ops(), HipTableOps()
    g = torch.Generator().manual_seed(3)
    G, B, n_items, D = 4, 600, 97, 64
    cap, S, off = 320, 640, 320                   # the item half of a packed [G, 2 * cap] exchange
    # requester side: slots of this rank's B rows, counted per owner like fr_bucket_by_owner does
    item = torch.randint(1, n_items, (B,), generator=g)
    cnt = [0] * G
    slot = torch.zeros(B, dtype=torch.int32)
    for b, it in enumerate(item.tolist()):
        o = it % G
        slot[b] = o * S + off + cnt[o]
        cnt[o] += 1
    out = torch.rand(B, generator=g) * 0.98 + 0.01
    label = (torch.rand(B, generator=g) < 0.6).float()
    sst = (torch.rand(B, generator=g) < 0.5).float()
    n = G * (cap + 1)
    rec_c, rec_g = torch.zeros(n * 2), torch.zeros(n * 2, device="cuda")
    ws = hip.df_workspace(B, G * cap, "cuda")
    cpu.df_pack(out, label, sst, slot, S, off, cap, G, rec_c, None)
    hip.df_pack(out.cuda(), label.cuda(), sst.cuda(), slot.cuda(), S, off, cap, G, rec_g, ws)
    assert torch.equal(rec_c, rec_g.cpu())
    # owner side: ids received from the four requesters (padding -1), their records, the owner's sorted segments
    ids = torch.full((G, cap), -1, dtype=torch.int64)
    recv = torch.zeros(G, cap + 1, 2)
    for q in range(G):
        m = int(torch.randint(150, cap, (1,), generator=g))
        ids[q, :m] = torch.randint(0, 25, (m,), generator=g)                   # local rows: many members per item
        pos = torch.rand(m, generator=g) < 0.6
        recv[q, :m, 0] = torch.where(pos, torch.rand(m, generator=g) * 0.98 + 0.01, torch.full((m,), -1.0))
        recv[q, :m, 1] = (torch.rand(m, generator=g) < 0.5).float()
        recv[q, cap] = torch.tensor([0.0, 1.0])
    recv[2, cap] = torch.tensor([1.0, 1.0])                                     # a sender whose positives are all of one group
    tab_c = _Table(torch.zeros(25, D))
    tab_c.ids = ids.view(-1)
    tab_g = LazyTable(torch.zeros(25, D, device="cuda"))
    tab_g.ensure_state()
    hyper = AdamHyper(1e-3, 0.0, device="cuda")
    rows = torch.empty(G * cap, D, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    tab_g.gather_train_into(hyper, ids.view(-1).cuda().data_ptr(), G * cap, rows.data_ptr(), 0, 0, err)
    rep_c, rep_g = torch.zeros(n * 4), torch.zeros(n * 4, device="cuda")
    for rep in range(2):                                                        # the arrival tickets are left clean
        cpu.df_owner(tab_c, recv.view(-1), G, cap, rep_c, None, B, None)
        hip.df_owner(tab_g, recv.view(-1).cuda(), G, cap, rep_g, ws, B, err)
        a, b = rep_c.view(G, cap + 1, 4), rep_g.cpu().view(G, cap + 1, 4)
        used = (ids >= 0) & (recv[:, :cap, 0] >= 0)
        assert torch.equal(torch.signbit(a[:, :cap, 0][used]), torch.signbit(b[:, :cap, 0][used]))
        np.testing.assert_allclose(b[:, :cap][used].numpy(), a[:, :cap][used].numpy(), rtol=1e-6)
        assert torch.equal(a[:, cap], b[:, cap])                                # K_owner, smin, smax
    assert int(err.item()) == 0
    # requester side again: a reply of this rank's layout
    reply = torch.zeros(G, cap + 1, 4)
    reply[:, :cap, 0] = torch.rand(G, cap, generator=g) * 3
    reply[:, :cap, 1] = torch.rand(G, cap, generator=g) * 3
    reply[:, :cap, 2:] = torch.randint(0, 5, (G, cap, 2), generator=g).float()
    flip = torch.rand(G, cap, generator=g) < 0.3
    reply[:, :cap, 0] = torch.where(flip, -reply[:, :cap, 0], reply[:, :cap, 0])
    reply[:, cap] = torch.tensor([[11.0, 0.0, 1.0, 0.0], [9.0, 0.0, 1.0, 0.0], [0.0, 0.0, 1.0, 0.0], [14.0, 0.0, 1.0, 0.0]])
    dy0 = torch.randn(B, generator=g) * 1e-3
    for fw in (0.1, 0.0):
        dy_c, dy_g = dy0.clone(), dy0.clone().cuda()
        l_c, l_g = torch.tensor([0.7, 0.7, 0.0]), torch.tensor([0.7, 0.7, 0.0], device="cuda")
        cpu.df_apply(reply.view(-1), slot, S, off, cap, G, out, label, sst, fw, float(G), dy_c, l_c, None)
        hip.df_apply(reply.view(-1).cuda(), slot.cuda(), S, off, cap, G, out.cuda(), label.cuda(), sst.cuda(), fw, float(G), dy_g,
                     l_g, ws)
        np.testing.assert_allclose(dy_g.cpu().numpy(), dy_c.numpy(), rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(l_g.cpu().numpy(), l_c.numpy(), rtol=2e-5)
    assert int(ws.view(torch.int32)[-64:].abs().sum().item()) == 0             # tickets back at zero


def test_global_df_kernels_skip_rows_that_found_no_slot():
    """fr_bucket_by_owner writes slot -1 for a row whose owner's bucket is full (skewed item ids; the overflow error bit voids
    the step at the epoch's check).  Until then nothing may be written or read through such a slot: fr_nfcf_df_pack used to
    store its record 8 bytes in front of the exchange buffer and fr_nfcf_df_apply to read its reply from there."""
    from fairrec.sharded_engine import HipTableOps
    hip = HipTableOps()
    g = torch.Generator().manual_seed(5)
    G, B = 4, 300
    cap, S, off = 64, 128, 64
    slot = torch.full((B,), -1, dtype=torch.int32)
    cnt = [0] * G
    for b in range(B):                         # every owner's bucket overflows: B / G = 75 rows for cap = 64 slots
        o = b % G
        if cnt[o] < cap:
            slot[b] = o * S + off + cnt[o]
            cnt[o] += 1
    assert int((slot < 0).sum()) == B - G * cap
    out = torch.rand(B, generator=g) * 0.98 + 0.01
    label = torch.ones(B)
    sst = (torch.rand(B, generator=g) < 0.5).float()
    n = G * (cap + 1)
    guard = 64
    buf = torch.full((guard + n * 2 + guard,), 7.25, device="cuda")             # guard bands around the record buffer
    rec = buf[guard:guard + n * 2]
    rec.zero_()
    ws = hip.df_workspace(B, G * cap, "cuda")
    hip.df_pack(out.cuda(), label.cuda(), sst.cuda(), slot.cuda(), S, off, cap, G, rec, ws)
    torch.cuda.synchronize()
    assert bool((buf[:guard] == 7.25).all()) and bool((buf[-guard:] == 7.25).all())
    r = rec.cpu().view(G, cap + 1, 2)
    ok = slot >= 0
    want = torch.zeros(G, cap, 2)
    want[(slot[ok] // S).long(), (slot[ok] % S - off).long()] = torch.stack([out[ok], sst[ok]], dim=1)
    assert torch.equal(r[:, :cap], want)
    # apply: rows without a slot take no fairness gradient (and read nothing)
    reply = torch.zeros(G, cap + 1, 4)
    reply[:, :cap, :2] = torch.rand(G, cap, 2, generator=g) * 3
    reply[:, :cap, 2:] = torch.randint(1, 5, (G, cap, 2), generator=g).float()
    reply[:, cap] = torch.tensor([5.0, 0.0, 1.0, 0.0])
    rbuf = torch.full((guard * 4 + n * 4 + guard * 4,), float("nan"), device="cuda")
    rbuf[guard * 4:guard * 4 + n * 4] = reply.view(-1).cuda()
    dy0 = torch.randn(B, generator=g) * 1e-3
    dy = dy0.clone().cuda()
    loss = torch.tensor([0.7, 0.7, 0.0], device="cuda")
    hip.df_apply(rbuf[guard * 4:guard * 4 + n * 4], slot.cuda(), S, off, cap, G, out.cuda(), label.cuda(), sst.cuda(), 0.1, float(G), dy,
                 loss, ws)
    d = dy.cpu()
    assert bool(torch.isfinite(d).all()) and bool(torch.isfinite(loss).all())
    assert torch.equal(d[~ok], dy0[~ok]) and not torch.equal(d[ok], dy0[ok])


@pytest.mark.parametrize("D,hidden,p,finetune,B", [(64, (128, 64), 0.2, False, 1000), (256, (128, 64), 0.2, True, 8192),
                                                   (32, (32, 32), 0.0, True, 77), (128, (96, 64), 0.5, False, 4097),
                                                   (64, (64, 32), 0.3, True, 32)],
                         ids=["d64_pretrain", "d256_finetune", "d32_tail_rows", "d128_n96", "one_tile"])
def test_fused_scorer_matches_the_layered_form(D, hidden, p, finetune, B):
    """csrc/scorer.hip (one forward + one backward launch) against the layer-by-layer form of round 3 on the same model,
    batches and dropout stream (device-drawn pattern: same seed, counter and offsets => the same keep masks): per-step losses,
    sigmoid scores, the scorer's weight gradients of the first step, and every parameter after three optimizer steps."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.model.fair_recommender.nfcf import NFCF
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, T = 3001, 701, 3
    g = torch.Generator().manual_seed(4)
    gender = (torch.rand(n_users, generator=g) < 0.5).float().numpy()
    u = torch.randint(1, n_users, (T, B), generator=g)
    i = torch.randint(1, n_items, (T, B), generator=g)
    label = (torch.rand(T, B, generator=g) < 0.6).float()
    models = []
    for fused in (True, False):
        torch.manual_seed(9)
        cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": list(hidden), "dropout": p, "fair_weight": 0.3,
                                                "device": "cuda", "load_pretrain_path": None})
        m = NFCF(cfg, _DS(n_users, n_items, gender))
        with torch.no_grad():
            m.user_embedding.weight.mul_(0.3)
            m.item_embedding.weight.mul_(0.3)
            for lin in m.mlp_layers.linears():
                lin.bias.add_(0.05)        # biases that matter (the default init is small)
        if finetune:
            m.load_pretrain_path = "a-checkpoint"
            m.user_embedding.weight.requires_grad = False
        m = m.to("cuda").train()
        m.FUSED = fused
        m.mlp_layers._seed = 1234567
        assert m._fused_scorer() == fused
        models.append(m)
    outs = []
    for m in models:
        opt = FusedLazyAdam(m.hip_engine(), lr=1e-3, weight_decay=1e-6, sweep_period=2)
        rec = {"loss": [], "grads": None}
        for t in range(T):
            inter = Interaction({"user_id": u[t], "item_id": i[t], "label": label[t], "gender": torch.from_numpy(gender)[u[t]]}).to("cuda")
            opt.zero_grad()
            loss = m.calculate_loss(inter)
            rec["loss"].append(float(loss))
            loss.backward()
            if t == 0:
                rec["grads"] = [q.grad.detach().clone() for q in m.mlp_layers.parameters()]
            opt.step()
        m.hip_engine().check_device_errors()
        rec["state"] = {k: v.detach().clone() for k, v in m.state_dict().items()}
        with torch.no_grad():
            m.eval()
            rec["predict"] = m.predict(Interaction({"user_id": u[0], "item_id": i[0]}).to("cuda")).clone()
        outs.append(rec)
    a, b = outs
    np.testing.assert_allclose(a["loss"], b["loss"], rtol=2e-5)
    for ga, gb in zip(a["grads"], b["grads"]):
        scale = float(gb.abs().max())
        assert float((ga - gb).abs().max()) <= 2e-5 * scale + 1e-9, (ga.shape, float((ga - gb).abs().max()), scale)
    for k in a["state"]:
        va, vb = a["state"][k], b["state"][k]
        assert float((va - vb).abs().max()) <= 2e-4 * float(vb.abs().max()) + 1e-7, k      # (Adam normalises: rounding-level
        # differences of a gradient come back at the scale of lr)
    torch.testing.assert_close(a["predict"], b["predict"], rtol=1e-4, atol=1e-6)
