"""GPU parity: the HIP FOCF path vs the reference's golden vectors and vs the oracle (through the C ABI)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "focf_*.npz")))
RTOL, ATOL = 1e-4, 1e-6   # north_star: 1e-4 relative fp32; atol covers near-zero entries (SURVEY §7 hard part 1)


def _close(a, b, what, rtol=RTOL, atol=ATOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    bad = np.abs(a - b) > rtol * np.abs(b) + atol
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} outside tolerance, max abs diff "
                           f"{np.abs(a - b).max():.3e}, worst rel {np.max(np.abs(a - b) / (np.abs(b) + 1e-12)):.3e}")


def _engine(z, sweep):
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    lr, wd, fw = (float(x) for x in z["hyper"][:3])
    U = torch.tensor(z["U0"], device="cuda")
    I = torch.tensor(z["I0"], device="cuda")
    eng = FocfEngine(U, I, str(z["objective"]), fw, 5.0)
    FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=sweep)
    return eng


@pytest.mark.parametrize("sweep", [0, 3, None, "lookahead", "lookahead2", "fused", "fused_ahead", "fused_nosweep", "fused_runs",
                                   "fused_runs_ahead"],
                         ids=["nosweep", "sweep3", "sweepdefault", "lookahead", "lookahead2", "fused", "fused_ahead",
                              "fused_nosweep", "runs", "runs_ahead"])
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_training_matches_reference_golden(path, sweep):
    z = np.load(path)
    # "lookahead": the next batch's index sort runs one step ahead on a side stream, stamps the batch's rows and carries
    # the sweep slice of the step in flight
    lookahead = {"lookahead": 1, "lookahead2": 5, "fused_ahead": 5, "fused_runs_ahead": 5}.get(sweep, 0)   # coming batches announced
    # "fused*": the whole step as ONE launch (fr_focf_step); the loss of a step is reduced by the next step's launch
    fused = isinstance(sweep, str) and sweep.startswith("fused")
    eng = _engine(z, 0 if sweep == "fused_nosweep" else (3 if (lookahead or fused) else sweep))
    eng.defer_loss = fused
    # "runs*": the one-launch step for item-complete batches (fr_focf_step_runs: a workgroup per chunk of the item-sorted
    # order, csrc/focf_runs.hip) -- on EVERY golden, item-complete or not: what it computes does not depend on the shape
    eng.item_runs = isinstance(sweep, str) and "runs" in sweep
    snaps = set(int(s) for s in z["snaps"])
    T = z["user_id"].shape[0]
    dev = "cuda"
    losses = []
    cols = {k: torch.tensor(z[k], device=dev) for k in ("user_id", "item_id", "rating", "sst")}
    for t in range(T):
        u, i, r, s = (cols[k][t] for k in ("user_id", "item_id", "rating", "sst"))
        nxt = [(cols["user_id"][j], cols["item_id"][j], cols["sst"][j], cols["rating"][j])
               for j in range(t + 1, t + 1 + lookahead) if j < T]
        nxt = (nxt[0] if lookahead == 1 else nxt) if nxt else None
        loss, pred = eng.forward(u, i, r, s, want_pred=(t == 0 and not fused), next_batch=nxt)
        losses.append(loss if fused else loss.clone())
        if t == 0 and not fused:
            _close(pred.cpu().numpy(), z["pred_step1"], "pred step 1", atol=1e-6)
        if "clip_max_norm" in z:      # config clip_grad_norm: the optimizer's step() calls this before the backward launch
            norm = eng.clip_grad_norm(float(z["clip_max_norm"]))
            _close(float(norm[0]), z["grad_norm"][t], f"gradient norm step {t + 1}")
        eng.backward_adam()
        if (t + 1) in snaps:
            eng.flush()
            for tag, tab in (("U", eng.U), ("I", eng.I)):
                _close(tab.weight.cpu().numpy(), z[f"{tag}_after{t + 1}"], f"{tag} after {t + 1}")
                # moments: absolute floor scaled to the tensor (entries are differences of cancelling terms)
                mref, vref = z[f"m{tag}_after{t + 1}"], z[f"v{tag}_after{t + 1}"]
                _close(tab.m.cpu().numpy(), mref, f"m{tag} after {t + 1}", atol=1e-6 * np.abs(mref).max())
                _close(tab.v.cpu().numpy(), vref, f"v{tag} after {t + 1}", atol=1e-6 * np.abs(vref).max())
    eng.finish()
    if fused and "clip_max_norm" not in z and str(z["objective"]) != "nonparity":
        assert eng._prev is None and eng.U.step == T
        np.testing.assert_allclose(float(eng.loss_acc[0]), float(np.sum(z["loss"], dtype=np.float64)), rtol=1e-4)
    got = torch.stack(losses).cpu().numpy()[:, 0]
    _close(got, z["loss"], "loss curve", atol=1e-6)
    eng.check_device_errors()
    # predict on the last batch with final weights (focf.py:145-150)
    p = eng.predict(u, i).cpu().numpy()
    _close(p, z["predict_last"], "predict", atol=1e-6)


def test_lazy_equals_flush_every_step():
    """Catch-up at next touch == catching up everything after every step (size-independent property).
    With weight decay the replay runs on scaled moments (m/k1, v/k2), so the two schedules differ by the
    rounding of the scale round trips: a few ulp, far below the parity tolerance; without weight decay
    they are the same sequence of fp32 operations and must agree bit for bit."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "focf_value_long.npz"))
    for wd_zero in (False, True):
        a, b = _engine(z, 0), _engine(z, 0)
        if wd_zero:
            from fairrec.optim import FusedLazyAdam
            for e in (a, b):
                FusedLazyAdam(e, lr=1e-3, weight_decay=0.0, sweep_period=0)
        for t in range(40):
            cols = [torch.tensor(z[k][t], device="cuda") for k in ("user_id", "item_id", "rating", "sst")]
            la, _ = a.forward(*cols)
            lb, _ = b.forward(*cols)
            a.backward_adam()
            b.backward_adam()
            b.flush()
            if wd_zero:
                assert torch.equal(la, lb)
        a.flush()
        for x, y in ((a.U.weight, b.U.weight), (a.I.weight, b.I.weight), (a.U.m, b.U.m), (a.U.v, b.U.v)):
            if wd_zero:
                assert torch.equal(x, y)
            else:
                torch.testing.assert_close(x, y, rtol=2e-5, atol=1e-8 * float(y.abs().max()) + 1e-12)


def _synthetic(n_users, n_items, B, T, seed, item_dist):
    g = torch.Generator().manual_seed(seed)
    u = torch.randint(1, n_users, (T, B), generator=g)
    if item_dist == "uniform":
        i = torch.randint(1, n_items, (T, B), generator=g)
    elif item_dist == "zipf":      # hot items: segments of every length, short and long
        i = (torch.rand((T, B), generator=g) ** 3 * (n_items - 1)).long() + 1
    else:                           # a few items only: every segment is long (the chain path inside a fused step)
        i = torch.randint(1, 9, (T, B), generator=g)
    u[:, 1] = u[:, 0]               # a repeated user in every batch ...
    i[:, 3] = i[:, 2]
    u[:, 3] = u[:, 2]               # ... and a repeated (user, item) pair
    r = torch.randint(1, 6, (T, B), generator=g).float()
    gender = torch.randint(0, 2, (n_users,), generator=g).float()
    return u.cuda(), i.cuda(), r.cuda(), gender[u].cuda()


@pytest.mark.parametrize("item_dist", ["uniform", "zipf", "few"])
@pytest.mark.parametrize("dim", [64, 100, 256])
@pytest.mark.parametrize("objective", ["none", "value", "nonparity"])
def test_lookahead_sweep_equals_plain_chain(objective, dim, item_dist):
    """The look-ahead prepare (sort + stamps + the sweep slice riding in the sort launch, rows brought to the state
    BEFORE the step) against the plain chain (sort inside forward, sweeper beside the backward kernel) on the same
    batches: same arithmetic per row except where a row is replayed in one stretch instead of two (rounding of the moment
    scaling: a few ulp)."""
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, B, T = 3001, 1501, 1000, 14
    u, i, r, s = _synthetic(n_users, n_items, B, T, 11, item_dist)
    g = torch.Generator().manual_seed(5)
    U0 = (torch.randn(n_users, dim, generator=g) * 0.1).cuda()
    I0 = (torch.randn(n_items, dim, generator=g) * 0.1).cuda()
    engs = []
    modes = (0, 1, 6, 7)        # coming batches announced to the engine; 7: the whole step as one launch (fr_focf_step)
    for ahead in modes:
        eng = FocfEngine(U0.clone(), I0.clone(), objective, 0.5, 5.0)
        FusedLazyAdam(eng, lr=1e-2, weight_decay=1e-3, sweep_period=4)
        eng.defer_loss = ahead >= 6      # loss reduced by the backward launch (read after backward_adam below)
        eng.item_runs = ahead == 1       # item row replayed once per workgroup of the gather kernel (a hint only)
        engs.append(eng)
    fused_losses = []
    for t in range(T):
        out = []
        for eng, ahead in zip(engs, modes):
            nxt = [(u[j], i[j], s[j], r[j]) for j in range(t + 1, t + 1 + ahead) if j < T] or None
            loss, pred = eng.forward(u[t], i[t], r[t], s[t], want_pred=ahead != 7, next_batch=nxt)
            eng.backward_adam()
            if ahead == 7:
                fused_losses.append(loss)          # filled by the next step's launch / finish()
                assert (eng._prev is not None) == (objective != "nonparity")
            else:
                out.append((loss.clone(), pred))
        for o in out[1:]:
            torch.testing.assert_close(o[0][:3], out[0][0][:3], rtol=2e-5, atol=1e-7)
            torch.testing.assert_close(o[1], out[0][1], rtol=2e-5, atol=1e-6)
        if t > 0:
            torch.testing.assert_close(fused_losses[t - 1][:3], prev_loss, rtol=2e-5, atol=1e-7)
        prev_loss = out[0][0][:3]
    engs[-1].finish()
    torch.testing.assert_close(fused_losses[-1][:3], prev_loss, rtol=2e-5, atol=1e-7)
    for eng in engs:
        eng.flush()
        eng.check_device_errors()
    a = engs[0]
    for b in engs[1:]:
        for x, y in ((b.U.weight, a.U.weight), (b.I.weight, a.I.weight), (b.U.m, a.U.m), (b.I.m, a.I.m),
                     (b.U.v, a.U.v), (b.I.v, a.I.v)):
            torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-6 * float(y.abs().max()) + 1e-12)


@pytest.mark.parametrize("dim", [64, 100, 128])
@pytest.mark.parametrize("objective", ["none", "value", "absolute"])
def test_in_launch_prepare_equals_sorted_prepare(objective, dim):
    """The one-launch step with the coming batches' index work riding in the step launches (fr_focf_step_staged: claim /
    place stages on the rows' 64-bit words, member lists ordered by the last arriver) against the same step prepared by
    the look-ahead sort (fr_focf_prepare_step): every sum of the shared-row path runs in ascending batch position in both,
    so the tables and moments must be EQUAL BIT FOR BIT; only the reported fairness value may differ in its last bits (the
    items' terms are summed in another order).  The batches hold rows shared by 2..3 interactions, hot items and hot
    users with more than 64 members (the chunked member lists), a queue that announces two batches ahead, steps nobody
    announced and a queue that is cut short (claimed batches that never run)."""
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, B, T = 5001, 2501, 1500, 16
    g = torch.Generator().manual_seed(17)
    u = torch.randint(1, n_users, (T + 4, B), generator=g)
    i = torch.randint(1, n_items, (T + 4, B), generator=g)
    i[:, 100:250] = torch.randint(1, 4, (T + 4, 150), generator=g)       # three hot items: ~50 members each
    i[:, 300:520] = 7                                                    # one item with 220 members
    u[:, 600:700] = 11                                                   # one user with 100 members
    u[:, 640:660] = u[:, 300:320]                                        # ... and users shared between hot rows
    r = torch.randint(1, 6, (T + 4, B), generator=g).float()
    gender = torch.randint(0, 2, (n_users,), generator=g).float()
    u, i, r = u.cuda(), i.cuda(), r.cuda()
    s = gender.cuda()[u]
    U0 = (torch.randn(n_users, dim, generator=g) * 0.1).cuda()
    I0 = (torch.randn(n_items, dim, generator=g) * 0.1).cuda()
    engs = []
    for staged in (False, True):
        eng = FocfEngine(U0.clone(), I0.clone(), objective, 0.5, 5.0)
        FusedLazyAdam(eng, lr=1e-2, weight_decay=1e-3, sweep_period=5)
        eng.defer_loss = True
        eng.staged = staged
        engs.append(eng)
    losses = [[], []]
    for t in range(T):
        if t < 6:
            nxt = [(u[j], i[j], s[j], r[j]) for j in range(t + 1, min(t + 4, T))]      # the queue runs
        elif t == 6:
            nxt = [(u[T + 1], i[T + 1], s[T + 1], r[T + 1]), (u[T + 2], i[T + 2], s[T + 2], r[T + 2])]   # ... announces batches
        elif t < 10:                                                                              #     that never come
            nxt = None                                                                 # nobody announces anything
        else:
            nxt = [(u[j], i[j], s[j], r[j]) for j in range(t + 1, min(t + 3, T))]
        for k, eng in enumerate(engs):
            loss, _ = eng.forward(u[t], i[t], r[t], s[t], next_batch=nxt or None)
            eng.backward_adam()
            assert eng._prev is not None and eng._prev[3] == (k == 1)
            losses[k].append(loss)
        for eng in engs:      # (every step: which steps of a row's replay run in one stretch depends on when the sweeper
            eng.flush()       # meets the row, and that on how far ahead its stamps were written -- a few ulp otherwise)
        a, b = engs
        for name in ("weight", "m", "v"):
            assert torch.equal(getattr(a.U, name), getattr(b.U, name)), "user %s after step %d" % (name, t)
            assert torch.equal(getattr(a.I, name), getattr(b.I, name)), "item %s after step %d" % (name, t)
    for eng in engs:
        eng.finish()
        eng.check_device_errors()
    for la, lb in zip(*losses):
        assert torch.equal(la[1], lb[1])                                     # mse: the same sum
        torch.testing.assert_close(lb[:3], la[:3], rtol=1e-5, atol=1e-7)     # fairness value: another summation order


@pytest.mark.parametrize("B", [1, 2, 63, 65, 257])
def test_in_launch_prepare_with_tiny_and_ragged_batches(B):
    """Batch sizes around the wave and workgroup granules of the stages (one interaction, an odd one out of the last pair,
    a partly filled stage wave), every row shared or none: staged and sorted prepare give the same tables bit for bit."""
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, D, T = 301, 151, 64, 7
    g = torch.Generator().manual_seed(B)
    u = torch.randint(1, n_users, (T, B), generator=g)
    i = torch.randint(1, n_items, (T, B), generator=g)
    i[2] = i[2, 0]                      # one step with a single item (every interaction on the shared path) ...
    u[3] = u[3, 0]                      # ... and one with a single user
    r = torch.randint(1, 6, (T, B), generator=g).float()
    gender = torch.randint(0, 2, (n_users,), generator=g).float()
    u, i, r = u.cuda(), i.cuda(), r.cuda()
    s = gender.cuda()[u]
    U0 = (torch.randn(n_users, D, generator=g) * 0.1).cuda()
    I0 = (torch.randn(n_items, D, generator=g) * 0.1).cuda()
    engs = []
    for staged in (False, True):
        eng = FocfEngine(U0.clone(), I0.clone(), "value", 0.5, 5.0)
        FusedLazyAdam(eng, lr=1e-2, weight_decay=1e-3, sweep_period=3)
        eng.defer_loss = True
        eng.staged = staged
        engs.append(eng)
    for t in range(T):
        nxt = [(u[j], i[j], s[j], r[j]) for j in range(t + 1, min(t + 3, T))] or None
        for eng in engs:
            eng.forward(u[t], i[t], r[t], s[t], next_batch=nxt)
            eng.backward_adam()
            eng.flush()
        a, b = engs
        for name in ("weight", "m", "v"):
            assert torch.equal(getattr(a.U, name), getattr(b.U, name)), "user %s after step %d" % (name, t)
            assert torch.equal(getattr(a.I, name), getattr(b.I, name)), "item %s after step %d" % (name, t)
    for eng in engs:
        eng.finish()
        eng.check_device_errors()
    torch.testing.assert_close(engs[1].loss_acc[:3], engs[0].loss_acc[:3], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("runs", [False, True], ids=["staged", "item_runs"])
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_steps_many_matches_reference_golden(path, runs):
    """The step loop issued by the library (fr_focf_steps_many: one call per run of batches, trainer.py:181-196 with its body
    in C) on every golden it applies to: the runs are cut at the snapshot steps, every per-step loss comes out of the loss
    ring the calls fill."""
    z = np.load(path)
    if "clip_max_norm" in z or str(z["objective"]) == "nonparity":
        pytest.skip("needs a batch-wide value between loss and update: per-batch path")
    eng = _engine(z, 3)
    eng.defer_loss = True
    eng.item_runs = runs          # fr_focf_runs_many: the pipelined item-run steps (what it computes does not depend on the shape)
    assert eng.can_step_many()
    T, B = z["user_id"].shape
    cols = [torch.tensor(z[k], device="cuda").reshape(-1) for k in ("user_id", "item_id", "rating", "sst")]
    snaps = sorted(set(int(s) for s in z["snaps"]) | {T})
    t0 = 0
    for t1 in snaps:
        if t1 > t0:
            assert eng.steps_many(*(c[t0 * B:t1 * B] for c in cols), B) == t1 - t0
        t0 = t1
        if f"U_after{t1}" in z:
            eng.flush()
            for tag, tab in (("U", eng.U), ("I", eng.I)):
                _close(tab.weight.cpu().numpy(), z[f"{tag}_after{t1}"], f"{tag} after {t1}")
                mref, vref = z[f"m{tag}_after{t1}"], z[f"v{tag}_after{t1}"]
                _close(tab.m.cpu().numpy(), mref, f"m{tag} after {t1}", atol=1e-6 * np.abs(mref).max())
                _close(tab.v.cpu().numpy(), vref, f"v{tag} after {t1}", atol=1e-6 * np.abs(vref).max())
    eng.finish()
    assert eng._prev is None and eng.U.step == T and T <= eng.LOSS_SLOTS
    got = eng.loss_ring.cpu().numpy()[(np.arange(T) + 1) % eng.LOSS_SLOTS, 0]
    _close(got, z["loss"], "loss curve", atol=1e-6)
    np.testing.assert_allclose(float(eng.loss_acc[0]), float(np.sum(z["loss"], dtype=np.float64)), rtol=1e-4)
    eng.check_device_errors()


@pytest.mark.parametrize("wd", [0.0, 1e-3], ids=["wd0", "wd1e-3"])
@pytest.mark.parametrize("ragged", [False, True], ids=["uniform", "ragged"])
@pytest.mark.parametrize("objective", ["none", "value"])
def test_steps_many_equals_the_per_batch_staged_loop(objective, ragged, wd):
    """fr_focf_steps_many issues the launches of fr_focf_step_staged: against the per-batch loop fed by a two-deep queue --
    with runs of 1, 2, 3 and many batches (the pipeline starts and drains inside every call), batch sizes that differ inside
    a run, shared rows with long member lists, and per-batch steps taken between two runs (the pending loss handed from one
    form to the other).  Without weight decay a replay is the same fp32 sequence however it is cut, so tables, moments,
    every step's loss and the running total must be EQUAL BIT FOR BIT; with it, a row's missed steps replayed in one stretch
    or in two (the sweeper of step k races the claim of batch k + 2 for the row's stamp, in either form) differ by the
    rounding of the moment scaling: a few ulp."""
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, D, T, B = 4001, 2001, 64, 23, 1200
    g = torch.Generator().manual_seed(29)
    sizes = [B] * T
    if ragged:
        sizes = [int(x) for x in torch.randint(1, B + 1, (T,), generator=g)]
        sizes[4], sizes[5] = 1, 65
    tot = sum(sizes)
    u = torch.randint(1, n_users, (tot,), generator=g)
    i = torch.randint(1, n_items, (tot,), generator=g)
    i[100:320] = 7                       # an item with 220 members in the first batch(es)
    u[50:90] = 11
    r = torch.randint(1, 6, (tot,), generator=g).float()
    gender = torch.randint(0, 2, (n_users,), generator=g).float()
    u, i, r = u.cuda(), i.cuda(), r.cuda()
    s = gender.cuda()[u]
    U0 = (torch.randn(n_users, D, generator=g) * 0.1).cuda()
    I0 = (torch.randn(n_items, D, generator=g) * 0.1).cuda()
    start = np.concatenate(([0], np.cumsum(sizes)))
    cut = lambda c, a, b: c[int(start[a]):int(start[b])]
    engs = []
    for _ in range(2):
        eng = FocfEngine(U0.clone(), I0.clone(), objective, 0.5, 5.0)
        FusedLazyAdam(eng, lr=1e-2, weight_decay=wd, sweep_period=5)
        eng.defer_loss = True
        engs.append(eng)
    a, b = engs
    # a: one batch per call, the queue announcing two ahead INSIDE the same runs b takes (a run starts with nothing staged)
    runs = [(0, 1), (1, 3), (3, 6), (6, 7), (7, 8), (8, 20), (20, T)]
    per_batch_between = {6, 7}           # b takes these two single-batch runs through forward / backward_adam as well
    for lo, hi in runs:
        for t in range(lo, hi):
            nxt = [tuple(cut(c, j, j + 1) for c in (u, i, s, r)) for j in range(t + 1, min(t + 3, hi))] or None
            a.forward(cut(u, t, t + 1), cut(i, t, t + 1), cut(r, t, t + 1), cut(s, t, t + 1), next_batch=nxt)
            a.backward_adam()
        if lo in per_batch_between:
            b.forward(cut(u, lo, hi), cut(i, lo, hi), cut(r, lo, hi), cut(s, lo, hi))
            b.backward_adam()
        else:
            n = b.steps_many(cut(u, lo, hi), cut(i, lo, hi), cut(r, lo, hi), cut(s, lo, hi),
                             sizes[lo:hi] if ragged else B)
            assert n == hi - lo
        assert a.U.step == b.U.step == hi
    for eng in engs:
        eng.flush()
        eng.check_device_errors()
    for name in ("weight", "m", "v", "last"):
        for ta, tb, tag in ((a.U, b.U, "user "), (a.I, b.I, "item ")):
            x, y = getattr(ta, name), getattr(tb, name)
            if wd == 0.0 or name == "last":
                assert torch.equal(x, y), tag + name
            else:
                torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-6 * float(y.abs().max()) + 1e-12, msg=tag + name)
    if wd == 0.0:
        assert torch.equal(a.loss_acc, b.loss_acc)
        assert torch.equal(a.loss_ring, b.loss_ring)
    else:
        torch.testing.assert_close(a.loss_ring, b.loss_ring, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(a.loss_acc, b.loss_acc, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("item_dist", ["uniform", "zipf"])
def test_in_launch_prepare_at_the_baseline_size_over_a_sweep_period(item_dist):
    """BASELINE.json configs[1] at its real size, more steps than one sweep period (every row is replayed and swept at least
    once, every generation of row words is reused many times): the engine with the in-launch prepare and the one with the
    sorted prepare must report the SAME running squared error to the bit (every prediction of every step equal to fp32
    rounding of the same sums) and hold the same tables up to the rounding of a replay run in one stretch or in two."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    dev = torch.device("cuda")
    T = 150
    u, i, r, s = (t.to(dev) for t in bench.synth_batches(T, bench.BATCH, bench.N_USERS, bench.N_ITEMS, 7, item_dist))
    engs = []
    for staged in (False, True):
        U, I = bench.xavier_tables(bench.N_USERS, bench.N_ITEMS, bench.DIM, 3, dev)
        eng = FocfEngine(U, I, bench.OBJECTIVE, bench.FAIR_WEIGHT, 5.0)
        FusedLazyAdam(eng, lr=bench.LR, weight_decay=bench.WD)
        eng.defer_loss = True
        eng.staged = staged
        engs.append(eng)
    rows = [(u[k], i[k], s[k], r[k]) for k in range(T)]
    for k in range(T):
        for eng in engs:
            eng.forward(u[k], i[k], r[k], s[k], next_batch=rows[k + 1:k + 21] or None)
            eng.backward_adam()
    for eng in engs:
        eng.finish()
        eng.flush()
        eng.check_device_errors()
    a, b = engs
    assert torch.equal(a.loss_acc[1], b.loss_acc[1]), (a.loss_acc.tolist(), b.loss_acc.tolist())       # sum of the steps' MSE
    torch.testing.assert_close(b.loss_acc[:3], a.loss_acc[:3], rtol=1e-5, atol=1e-6)
    for tab in ("U", "I"):
        for name in ("weight", "m", "v"):
            x, y = getattr(getattr(a, tab), name), getattr(getattr(b, tab), name)
            # (rows no batch touches wobble around zero by Adam steps of ~1e-7; a different split of their replay moves them by
            # a fraction of one such step)
            tol = 2e-3 * bench.LR if name == "weight" else 1e-3 * float(x.abs().max())
            assert float((x - y).abs().max()) <= tol, (tab, name, float((x - y).abs().max()), tol)


def test_full_size_steps_match_the_oracle():
    """BASELINE.json configs[1] at its real size (1 000 001 users x 100 001 items, D = 64, B = 8192, Adam lr 1e-3 wd 1e-3,
    fair_objective value): a few optimizer steps of the HIP path -- look-ahead sorts, sweeper, lazy replay -- against the
    oracle's dense step (the reference's arithmetic: autograd gradient + stock torch.optim.Adam over both whole tables).
    Every row of both tables is compared after a flush, so rows that were never in a batch (pure weight-decay replay)
    are checked too."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle.focf import CpuTrainerBaseline
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, D, B, T = 1_000_001, 100_001, 64, 8192, 5
    ref = CpuTrainerBaseline(n_users, n_items, D, 1e-3, 1e-3, 0.5, "value", seed=7, threads=8)
    eng = FocfEngine(ref.U.detach().clone().cuda(), ref.I.detach().clone().cuda(), "value", 0.5, 5.0)
    FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-3, sweep_period=3)     # short period: the sweeper visits every row
    eng.defer_loss = True
    g = torch.Generator().manual_seed(3)
    u = torch.randint(1, n_users, (T, B), generator=g)
    i = torch.randint(1, n_items, (T, B), generator=g)
    r = torch.randint(1, 6, (T, B), generator=g).float()
    gender = (torch.rand(n_users, generator=g) < 0.5).float()
    s = gender[u]
    ud, idv, rd, sd = u.cuda(), i.cuda(), r.cuda(), s.cuda()
    for t in range(T):
        want = ref.step(u[t], i[t], r[t], s[t])
        coming = [(ud[j], idv[j], sd[j], rd[j]) for j in range(t + 1, T)] or None
        loss, _ = eng.forward(ud[t], idv[t], rd[t], sd[t], next_batch=coming)
        eng.backward_adam()
        assert eng._prev is not None          # the fused one-launch step (fr_focf_step) is what runs here
        eng.finish()
        assert abs(float(loss[0]) - want) <= 1e-4 * abs(want), (t, float(loss[0]), want)
    eng.flush()
    eng.check_device_errors()
    _close(eng.U.weight.cpu().numpy(), ref.U.detach().numpy(), "user table after %d steps" % T)
    _close(eng.I.weight.cpu().numpy(), ref.I.detach().numpy(), "item table after %d steps" % T)
    st = ref.opt.state[ref.U]
    _close(eng.U.m.cpu().numpy(), st["exp_avg"].numpy(), "user exp_avg", atol=1e-6 * float(st["exp_avg"].abs().max()))
    _close(eng.U.v.cpu().numpy(), st["exp_avg_sq"].numpy(), "user exp_avg_sq",
           atol=1e-6 * float(st["exp_avg_sq"].abs().max()))


def test_prefetch_queue_over_epochs_matches_plain_steps():
    """The trainer's usage pattern over several epochs: a prefetch queue of up to 10 announced batches that drains at
    every epoch end, a flush (evaluation / checkpoint) between epochs, item-complete batches of varying size.  The tables
    must equal those of an engine that is stepped plainly (no look-ahead at all)."""
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, D = 4001, 601, 64
    g = torch.Generator().manual_seed(21)
    U0 = (torch.randn(n_users, D, generator=g) * 0.05).cuda()
    I0 = (torch.randn(n_items, D, generator=g) * 0.05).cuda()
    gender = torch.randint(0, 2, (n_users,), generator=g).float()
    engs = []
    # 0: plain steps; 1: prefetch queue, three-launch chain (item-complete hint, the one-launch runs step switched off);
    # 2: prefetch queue, one-launch step; 3: prefetch queue, one-launch step for item-complete batches (fr_focf_step_runs)
    for k in range(4):
        eng = FocfEngine(U0.clone(), I0.clone(), "value", 0.3, 5.0)
        FusedLazyAdam(eng, lr=5e-3, weight_decay=1e-3, sweep_period=7)
        eng.defer_loss = k >= 1
        eng.item_runs = k in (1, 3)
        if k == 1:
            eng.RUNS = False
        engs.append(eng)
    for epoch in range(3):
        batches = []
        for _ in range(23 + 5 * epoch):
            B = int(torch.randint(700, 1100, (1,), generator=g))
            items = torch.randint(1, n_items, (12,), generator=g)
            i = items[torch.arange(B) * 12 // B]                    # runs of equal items, like FOCFDataLoader's batches
            u = torch.randint(1, n_users, (B,), generator=g)
            r = torch.randint(1, 6, (B,), generator=g).float()
            batches.append((u.cuda(), i.cuda(), r.cuda(), gender[u].cuda()))
        for t, (u, i, r, s) in enumerate(batches):
            engs[0].forward(u, i, r, s)
            engs[0].backward_adam()
            queue = [(b[0], b[1], b[3], b[2]) for b in batches[t + 1:t + 11]] or None
            for eng in engs[1:]:
                eng.forward(u, i, r, s, next_batch=queue)
                eng.backward_adam()
            assert engs[1]._prev is None and engs[2]._prev is not None
            assert engs[3]._prev is not None or engs[3]._pipe is not None      # (pipelined: the step's item runs are pending)
        for eng in engs:
            eng.flush()                                              # evaluation / checkpoint between epochs
            eng.check_device_errors()
        assert not engs[1]._prep and not engs[2]._prep and not engs[3]._prep, "the queue must be empty at an epoch end"
    a = engs[0]
    for b in engs[1:]:
        for x, y in ((b.U.weight, a.U.weight), (b.I.weight, a.I.weight), (b.U.v, a.U.v), (b.I.m, a.I.m)):
            # the schedules split a row's replay into different stretches (rounding of the moment scaling): the parity
            # tolerance, with the absolute floor scaled to the tensor
            torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-6 * float(y.abs().max()) + 1e-12)


def test_optimizer_step_clips_like_the_reference_loop():
    """config `clip_grad_norm` reaches FusedLazyAdam, whose step() clips before the backward launch -- same numbers as
    clipping by hand (the golden cases above do that) -- and other norm types / engines without a norm pass are refused."""
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "focf_value_clip.npz"))
    lr, wd, fw = (float(x) for x in z["hyper"][:3])
    mn = float(z["clip_max_norm"])
    engs = []
    for cfg in (dict(max_norm=mn, norm_type=2), None):
        eng = FocfEngine(torch.tensor(z["U0"], device="cuda"), torch.tensor(z["I0"], device="cuda"), "value", fw, 5.0)
        opt = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=3, clip_grad_norm=cfg)
        engs.append((eng, opt))
    for t in range(4):
        cols = [torch.tensor(z[k][t], device="cuda") for k in ("user_id", "item_id", "rating", "sst")]
        for eng, opt in engs:
            eng.forward(*cols)
            if opt.clip is None:
                eng.clip_grad_norm(mn)
            opt.step()
    for eng, _ in engs:
        eng.flush()
    assert torch.equal(engs[0][0].U.weight, engs[1][0].U.weight) and torch.equal(engs[0][0].I.weight, engs[1][0].I.weight)
    with pytest.raises(NotImplementedError):
        FusedLazyAdam(engs[0][0], lr=lr, weight_decay=wd, clip_grad_norm=dict(max_norm=1.0, norm_type=1))


def test_deferred_loss_with_interleaved_engines():
    """FR_FOCF_DEFER_LOSS: the backward launch reduces the loss of ITS workspace's batch, also when another engine's
    forward ran in between."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "focf_value.npz"))
    ref, a, b = _engine(z, 3), _engine(z, 3), _engine(z, 3)
    a.defer_loss = b.defer_loss = True
    b.fused_step = False          # a: one-launch step (loss reduced by the next step / finish()); b: three-launch chain
    for t in range(6):
        cols = [torch.tensor(z[k][t], device="cuda") for k in ("user_id", "item_id", "rating", "sst")]
        lr, _ = ref.forward(*cols)
        lr = lr.clone()
        ref.backward_adam()
        la, _ = a.forward(*cols)
        lb, _ = b.forward(*cols)
        a.backward_adam()
        b.backward_adam()
        a.finish()
        torch.testing.assert_close(la[:3], lr[:3], rtol=1e-6, atol=0)
        torch.testing.assert_close(lb[:3], lr[:3], rtol=1e-6, atol=0)


def test_two_host_threads_with_their_own_workspaces():
    """The C ABI keeps no host-side state per call (the deferred-loss record lives in the workspace): two host threads,
    each driving its own engine on its own stream, forced to interleave launch by launch, get what a serial run gets."""
    import threading
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "focf_value.npz"))
    T = 6
    cols = [[torch.tensor(z[k][t], device="cuda") for k in ("user_id", "item_id", "rating", "sst")] for t in range(T)]

    def run(eng, losses, gate=None):
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.default_stream())
        with torch.cuda.stream(st):
            for t in range(T):
                if gate is not None:
                    gate.wait()          # both threads are between the same two launches
                loss, _ = eng.forward(*cols[t])
                if gate is not None:
                    gate.wait()
                eng.backward_adam()
                losses.append(loss)
            eng.finish()
            eng.flush()
        st.synchronize()

    results = {}
    for mode in ("serial", "threads"):
        engs = [_engine(z, 3), _engine(z, 3)]
        engs[1].fused_step = False       # one engine per launch shape: one-launch step, three-launch chain
        for e in engs:
            e.defer_loss = True
        torch.cuda.synchronize()
        out = [[], []]
        if mode == "serial":
            for e, o in zip(engs, out):
                run(e, o)
        else:
            gate = threading.Barrier(2)
            errs = []

            def guarded(e, o):
                try:
                    run(e, o, gate)
                except Exception as ex:   # a failing thread must not leave the other one at the barrier
                    errs.append(ex)
                    gate.abort()

            th = [threading.Thread(target=guarded, args=(e, o)) for e, o in zip(engs, out)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            assert not errs, errs
        torch.cuda.synchronize()
        results[mode] = [(e.U.weight.clone(), e.I.weight.clone(), float(e.loss_acc[0])) for e in engs]
    for k in range(2):
        torch.testing.assert_close(results["threads"][k][0], results["serial"][k][0], rtol=0, atol=0)
        torch.testing.assert_close(results["threads"][k][1], results["serial"][k][1], rtol=0, atol=0)
    assert results["threads"][0][2] == results["serial"][0][2]
    np.testing.assert_allclose(results["serial"][0][2], float(np.sum(z["loss"][:T], dtype=np.float64)), rtol=1e-4)


def test_device_error_flags():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "focf_value.npz"))
    eng = _engine(z, 0)
    u = torch.tensor(z["user_id"][0], device="cuda")
    i = torch.tensor(z["item_id"][0], device="cuda")
    r = torch.tensor(z["rating"][0], device="cuda")
    s = torch.tensor(z["sst"][0], device="cuda").clone()
    s[:3] = torch.tensor([2.0, 3.0, 4.0], device="cuda")   # >2 groups: the reference raises IndexError at focf.py:86
    eng.forward(u, i, r, s)
    with pytest.raises(IndexError):
        eng.check_device_errors()
    u2 = u.clone()
    u2[0] = 10 ** 6
    eng.forward(u2, i, r, torch.tensor(z["sst"][0], device="cuda"))
    with pytest.raises(IndexError):
        eng.check_device_errors()


@pytest.mark.parametrize("staged", [False, True])
def test_lookahead_stamps_are_the_apply_steps(staged):
    """Every batch of a fused-step loop is stamped with the optimizer step at which it is applied -- prepared by a side
    launch many steps ahead, by the stages riding in the two step launches before its own, or in line; first batch of an
    epoch (nothing announced before it) included: the sweeper tells a batch's rows from its own by that stamp, and the start
    order of a step's sweeper tasks is built for the stamped step."""
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, D, B = 3001, 501, 64, 512
    g = torch.Generator().manual_seed(5)
    eng = FocfEngine((torch.randn(n_users, D, generator=g) * 0.05).cuda(), (torch.randn(n_items, D, generator=g) * 0.05).cuda(),
                     "value", 0.5, 5.0)
    FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-3, sweep_period=5)
    eng.defer_loss = True
    eng.staged = staged
    gender = torch.randint(0, 2, (n_users,), generator=g).float()
    seen = []
    for epoch in range(2):
        batches = []
        for _ in range(45):
            u = torch.randint(1, n_users, (B,), generator=g)
            batches.append((u.cuda(), torch.randint(1, n_items, (B,), generator=g).cuda(),
                            torch.randint(1, 6, (B,), generator=g).float().cuda(), gender[u].cuda()))
        for t, (u, i, r, s) in enumerate(batches):
            queue = [(b[0], b[1], b[3], b[2]) for b in batches[t + 1:t + 21]] or None
            eng.forward(u, i, r, s, next_batch=queue)
            assert eng._stash is not None, "the one-launch step must be taken"
            stamp = eng._stash[6]
            seen.append((stamp[0] if staged else stamp, eng.U.step + 1))      # (staged: stamp and generation of row words)
            eng.backward_adam()
        eng.flush()
    eng.check_device_errors()
    assert all(a == b for a, b in seen), [x for x in seen if x[0] != x[1]][:5]


def _item_complete_batches(n_users, n_items, T, target, seed, degree=(20, 140)):
    """Batches as FOCFDataLoader forms them (focf_dataloader.py:37-51): random items, ALL interactions of each, until >= target
    rows; users recur under several items of a batch; one item much longer than a stage-2 pass of the kernel."""
    rng = np.random.default_rng(seed)
    gender = rng.integers(0, 2, n_users).astype(np.float32)
    out = []
    for t in range(T):
        us, its = [], []
        picked = rng.permutation(np.arange(1, n_items))
        k = 0
        while sum(len(x) for x in us) < target:
            deg = int(rng.integers(*degree)) if k != 1 else 330          # (a run of 330 members: four passes at D = 64)
            us.append(rng.choice(np.arange(1, n_users), size=deg, replace=False))
            its.append(np.full(deg, picked[k]))
            k += 1
        u, i = np.concatenate(us), np.concatenate(its)
        r = rng.integers(1, 6, u.size).astype(np.float32)
        out.append(tuple(torch.tensor(x, device="cuda") for x in (u, i, r, gender[u])))
    return out


@pytest.mark.parametrize("objective", ["value", "none", "under"])
@pytest.mark.parametrize("dim", [64, 128, 40])
def test_runs_step_matches_chain_and_is_bit_reproducible(objective, dim):
    """fr_focf_step_runs against the three-launch chain on item-complete batches of ragged sizes (what FOCFDataLoader feeds):
    the same sums in the same order, so the tables agree to the rounding of where a replay is cut in two (a few ulp); and
    against ITSELF, with and without announced batches: torch.equal -- no result may depend on which chunk arrives last."""
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, T = 1501, 401, 9
    batches = _item_complete_batches(n_users, n_items, T, 900, seed=dim)
    g = torch.Generator().manual_seed(1)
    U0 = (torch.randn(n_users, dim, generator=g) * 0.1).cuda()
    I0 = (torch.randn(n_items, dim, generator=g) * 0.1).cuda()

    def run(mode):
        eng = FocfEngine(U0.clone(), I0.clone(), objective, 0.7, 5.0)
        FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-3, sweep_period=4)
        eng.defer_loss = True
        eng.item_runs = True
        eng.PIPE = False          # the two-launch form itself (the pipelined form has its own tests below)
        eng.fused_step = mode != "chain"
        for t, (u, i, r, s) in enumerate(batches):
            nxt = [(b[0], b[1], b[3], b[2]) for b in batches[t + 1:t + 4]] if mode == "runs_ahead" else None
            eng.forward(u, i, r, s, next_batch=nxt or None)
            eng.backward_adam()
        eng.finish()
        eng.flush()
        eng.check_device_errors()
        return eng

    chain, a, b, c = run("chain"), run("runs"), run("runs"), run("runs_ahead")
    for x, y, z in ((a.U.weight, b.U.weight, c.U.weight), (a.I.weight, b.I.weight, c.I.weight), (a.U.m, b.U.m, c.U.m),
                    (a.I.v, b.I.v, c.I.v)):
        assert torch.equal(x, y)
    for x, z in ((a.U.weight, c.U.weight), (a.I.weight, c.I.weight)):      # announced ahead: rows may be swept at other steps
        torch.testing.assert_close(x, z, rtol=2e-5, atol=1e-8)
    for x, y in ((a.U.weight, chain.U.weight), (a.I.weight, chain.I.weight), (a.U.m, chain.U.m), (a.I.m, chain.I.m),
                 (a.U.v, chain.U.v), (a.I.v, chain.I.v)):
        torch.testing.assert_close(x, y, rtol=2e-5, atol=1e-8 * float(y.abs().max()) + 1e-12)
    np.testing.assert_allclose(a.loss_acc.cpu().numpy()[:3], chain.loss_acc.cpu().numpy()[:3], rtol=1e-5)


@pytest.mark.parametrize("objective,dim", [("value", 64), ("none", 64), ("under", 128), ("value", 40)])
def test_pipelined_runs_step_equals_the_two_launch_step(objective, dim):
    """fr_focf_step_runs_pipe (the item runs of batch k - 1 and the gather of batch k in ONE launch) against fr_focf_step_runs
    on item-complete batches whose consecutive members share most of their users and some items -- every shared row goes
    through the in-launch hand-off (the finisher publishes, the gather waits): the tables agree to the rounding of where a
    replay is cut (rows are swept at other steps), pipelined runs agree with each other bit for bit -- a row taken too early
    would miss a whole Adam step -- whether the batches were announced or not, with the pipeline drained in the middle of
    the loop (finish / predict / flush) or not."""
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, T = 1501, 401, 12
    batches = _item_complete_batches(n_users, n_items, T, 900, seed=dim + 7)
    # some items in consecutive batches as well (an epoch boundary does that): batch t + 1 starts with the last run of batch t
    rng = np.random.default_rng(5)
    linked = [batches[0]]
    for t in range(1, T):
        pu, pi_, pr, ps = linked[-1]
        last_item = pi_[-1]
        n_last = int((pi_ == last_item).sum())
        u, i, r, s = batches[t]
        keep = i != last_item
        linked.append((torch.cat([pu[-n_last:], u[keep]]), torch.cat([pi_[-n_last:], i[keep]]),
                       torch.cat([pr[-n_last:].flip(0), r[keep]]), torch.cat([ps[-n_last:], s[keep]])))
    batches = linked
    g = torch.Generator().manual_seed(1)
    U0 = (torch.randn(n_users, dim, generator=g) * 0.1).cuda()
    I0 = (torch.randn(n_items, dim, generator=g) * 0.1).cuda()

    def run(mode):
        eng = FocfEngine(U0.clone(), I0.clone(), objective, 0.7, 5.0)
        FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-3, sweep_period=4)
        eng.defer_loss = True
        eng.item_runs = True
        eng.PIPE = mode != "runs"
        seen = []
        for t, (u, i, r, s) in enumerate(batches):
            nxt = [(b[0], b[1], b[3], b[2]) for b in batches[t + 1:t + 4]] if mode == "pipe_ahead" else None
            eng.forward(u, i, r, s, next_batch=nxt or None)
            eng.backward_adam()
            if mode == "pipe_drained":
                if t == 3:
                    eng.finish()
                if t == 6:
                    seen.append(eng.predict(u[:50], i[:50]).clone())
                if t == 8:
                    eng.flush()
        eng.finish()
        eng.flush()
        eng.check_device_errors()
        return eng

    ref = run("runs")
    runs = [run("pipe"), run("pipe"), run("pipe"), run("pipe_ahead"), run("pipe_drained")]
    a = runs[0]
    for other in runs[1:3]:
        for x, y in ((a.U.weight, other.U.weight), (a.I.weight, other.I.weight), (a.U.m, other.U.m), (a.I.m, other.I.m),
                     (a.U.v, other.U.v), (a.I.v, other.I.v)):
            assert torch.equal(x, y)
    for other in runs:
        for x, y in ((other.U.weight, ref.U.weight), (other.I.weight, ref.I.weight)):
            torch.testing.assert_close(x, y, rtol=2e-5, atol=1e-8 * float(y.abs().max()) + 1e-12)
        # (the moments of a row swept at another step differ where their sums cancel: a looser floor; a missed Adam step
        # would move a weight by lr = 1e-3 and a first moment by a tenth of a gradient)
        for x, y in ((other.U.m, ref.U.m), (other.I.m, ref.I.m), (other.U.v, ref.U.v), (other.I.v, ref.I.v)):
            torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-6 * float(y.abs().max()) + 1e-14)
        np.testing.assert_allclose(other.loss_acc.cpu().numpy()[:4], ref.loss_acc.cpu().numpy()[:4], rtol=1e-5)


@pytest.mark.parametrize("wd", [0.0, 1e-3], ids=["wd0", "wd1e-3"])
@pytest.mark.parametrize("objective", ["none", "value"])
def test_runs_many_equals_the_per_batch_pipelined_loop(objective, wd):
    """fr_focf_runs_many (the library issues the pipelined item-run steps of a run of item-complete batches, their sorted
    prepare one group ahead on its side stream) against the per-batch loop (forward / backward_adam, batches announced
    ahead): 27 ragged batches -- more than three prepare groups -- whose neighbours share most users and one item run, cut
    into runs of 1, 2, 9 and 15 batches with the pipeline handed from one call to the next, a drain (flush) in between.
    Without weight decay: bit for bit (tables, moments, every loss); with it: to the rounding of where a replay is cut."""
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, T, D = 1501, 401, 27, 64
    batches = _item_complete_batches(n_users, n_items, T, 700, seed=23)
    linked = [batches[0]]
    for t in range(1, T):
        pu, pi_, pr, ps = linked[-1]
        n_last = int((pi_ == pi_[-1]).sum())
        u, i, r, s = batches[t]
        keep = i != pi_[-1]
        linked.append((torch.cat([pu[-n_last:], u[keep]]), torch.cat([pi_[-n_last:], i[keep]]),
                       torch.cat([pr[-n_last:].flip(0), r[keep]]), torch.cat([ps[-n_last:], s[keep]])))
    batches = linked
    sizes = [int(b[0].numel()) for b in batches]
    cols = [torch.cat([b[j] for b in batches]).contiguous() for j in range(4)]
    start = np.concatenate(([0], np.cumsum(sizes)))
    g = torch.Generator().manual_seed(1)
    U0 = (torch.randn(n_users, D, generator=g) * 0.1).cuda()
    I0 = (torch.randn(n_items, D, generator=g) * 0.1).cuda()
    engs = []
    for _ in range(2):
        eng = FocfEngine(U0.clone(), I0.clone(), objective, 0.7, 5.0)
        FusedLazyAdam(eng, lr=1e-2, weight_decay=wd, sweep_period=4)
        eng.defer_loss = True
        eng.item_runs = True
        engs.append(eng)
    a, b = engs
    assert b.can_step_many()
    for t, (u, i, r, s) in enumerate(batches):
        a.forward(u, i, r, s, next_batch=[(x[0], x[1], x[3], x[2]) for x in batches[t + 1:t + 4]] or None)
        a.backward_adam()
        if t == 11:
            a.flush()
    for lo, hi in ((0, 1), (1, 3), (3, 12), (12, T)):
        n = b.steps_many(*(c[int(start[lo]):int(start[hi])] for c in cols), sizes[lo:hi])
        assert n == hi - lo and b.U.step == hi and b._pipe is not None
        if hi == 12:
            b.flush()
    for eng in engs:
        eng.flush()
        eng.check_device_errors()
        assert eng._pipe is None and eng._prev is None
    for name in ("weight", "m", "v", "last"):
        for ta, tb, tag in ((a.U, b.U, "user "), (a.I, b.I, "item ")):
            x, y = getattr(ta, name), getattr(tb, name)
            if wd == 0.0 or name == "last":
                assert torch.equal(x, y), tag + name
            else:
                torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-6 * float(y.abs().max()) + 1e-12, msg=tag + name)
    if wd == 0.0:
        assert torch.equal(a.loss_ring, b.loss_ring) and torch.equal(a.loss_acc, b.loss_acc)
    else:
        torch.testing.assert_close(a.loss_ring, b.loss_ring, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(a.loss_acc, b.loss_acc, rtol=1e-5, atol=1e-7)


def test_pipelined_runs_step_at_full_batch_size():
    """The same comparison at B = 8192 on tables large enough that a launch holds every kind of workgroup at once (sweeper
    slice, 80 item runs, 1024 gather workgroups) and a few dozen users recur from one batch to the next: 40 steps pipelined
    against 40 steps of the two-launch form, then the pipelined run against itself."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    n_users, n_items, B, T, D = 300_001, 30_001, 8192, 40, 64
    u, i, r, s = (t.cuda() for t in bench.synth_batches(T, B, n_users, n_items, 11, "grouped"))
    g = torch.Generator().manual_seed(2)
    U0 = (torch.randn(n_users, D, generator=g) * 0.05).cuda()
    I0 = (torch.randn(n_items, D, generator=g) * 0.05).cuda()

    def run(pipe, ahead):
        eng = FocfEngine(U0.clone(), I0.clone(), "value", 0.5, 5.0)
        FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-3)
        eng.defer_loss = True
        eng.item_runs = True
        eng.PIPE = pipe
        rows = [(u[k], i[k], s[k], r[k]) for k in range(T)]
        for k in range(T):
            eng.forward(u[k], i[k], r[k], s[k], next_batch=(rows[k + 1:k + 9] or None) if ahead else None)
            eng.backward_adam()
        eng.flush()
        eng.check_device_errors()
        return eng

    ref, a, b, c = run(False, False), run(True, False), run(True, False), run(True, True)
    for x, y in ((a.U.weight, b.U.weight), (a.I.weight, b.I.weight), (a.U.m, b.U.m), (a.I.v, b.I.v)):
        assert torch.equal(x, y)
    # (batches announced ahead are stamped by a side-stream launch while a step's sweeper reads the stamps: which rows a
    # sweep leaves to their batch -- hence where their replays are cut -- depends on timing there, pipelined or not)
    for o in (a, c):
        for x, y in ((o.U.weight, ref.U.weight), (o.I.weight, ref.I.weight)):
            # (19 M elements, some of them crossing zero: a floor of 1e-7 of the largest weight; a missed step is 1e-3)
            torch.testing.assert_close(x, y, rtol=2e-5, atol=1e-7 * float(y.abs().max()) + 1e-12)
        np.testing.assert_allclose(o.loss_acc.cpu().numpy()[:4], ref.loss_acc.cpu().numpy()[:4], rtol=1e-5)
