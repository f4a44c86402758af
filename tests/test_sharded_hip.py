"""GPU: the HIP kernels of the row-sharded step (bucket / padded gather / score / owner fairness / grads / apply)
against the reference's golden vectors, run as a 1-rank "world" over RCCL (world_size 1 exercises every kernel and
every collective call; the multi-rank exchange schedule itself is covered by tests/test_sharded_gloo.py)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def pg(rccl_world1):
    yield


@pytest.mark.parametrize("schedule", ["requester", "item_owner"])
@pytest.mark.parametrize("lookahead", [False, True], ids=["inline", "lookahead"])
@pytest.mark.parametrize("case", ["focf_none", "focf_value", "focf_absolute", "focf_under", "focf_over",
                                  "focf_value_grouped", "focf_value_d128", "focf_value_pad", "focf_value_long",
                                  "focf_nonparity"])
def test_sharded_hip_matches_reference_golden(pg, case, lookahead, schedule):
    from fairrec.sharded import ShardedFocfEngine, ShardedFocfEngineV2
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    lr, wd, fw = (float(x) for x in z["hyper"][:3])
    v2 = schedule == "item_owner"     # interactions routed to the item owners: 2 dependent all-to-alls per step
    if v2 and str(z["objective"]) == "nonparity":
        pytest.skip("non-parity runs on the requester-computes schedule")
    eng = (ShardedFocfEngineV2 if v2 else ShardedFocfEngine)(
        torch.tensor(z["U0"], device="cuda"), torch.tensor(z["I0"], device="cuda"), str(z["objective"]), fw, lr, wd,
        capacity_factor=1.0)
    snaps = set(int(s) for s in z["snaps"])
    losses = []
    T = z["user_id"].shape[0]
    batches = [[torch.tensor(z[k][t], device="cuda") for k in ("user_id", "item_id", "rating", "sst")] for t in range(T)]
    for t in range(T):
        nxt = None
        if lookahead and t + 1 < T and t % 4 != 3:   # the next step's bucket / id exchange / sort runs on a side stream
            nxt = (batches[t + 1][0], batches[t + 1][1], batches[t + 1][3], batches[t + 1][2])
        loss, pred = eng.forward(*batches[t], next_batch=nxt)
        losses.append(loss.reshape(1).clone())
        if t == 0 and pred is not None:      # (the item-owner schedule leaves the scores with the item owners)
            np.testing.assert_allclose(pred.cpu().numpy(), z["pred_step1"], rtol=1e-4, atol=1e-6)
        eng.backward_adam()
        if (t + 1) in snaps:
            eng.flush()
            for tag, tab in (("U", eng.U), ("I", eng.I)):
                a, b = tab.weight.cpu().numpy(), z[f"{tag}_after{t + 1}"]
                assert (np.abs(a - b) <= 1e-4 * np.abs(b) + 1e-6).all(), (tag, t + 1, np.abs(a - b).max())
    np.testing.assert_allclose(torch.cat(losses).cpu().numpy(), z["loss"], rtol=1e-4)
    eng.check_device_errors()


def test_bucket_by_owner_bit_exact():
    """Dense [G, cap] layout and the shared [G, 2*cap+1] layout (second list + the (min, max) pair of an aux column in
    the trailing slot of every chunk)."""
    from fairrec.sharded import HipOps
    ops = HipOps("cuda")
    g = torch.Generator().manual_seed(5)
    for M, G, cap, shared in ((1, 1, 1, False), (100, 2, 80, True), (8192, 8, 1200, True), (8192, 4, 8192, False),
                              (5000, 3, 100, True), (8192, 8, 2048, True)):
        idx = torch.randint(0, 10 ** 6, (M,), generator=g, dtype=torch.int64)
        aux = torch.randn(M, generator=g) if shared else None
        stride, offset, aux_slot = (2 * cap + 1, cap, 2 * cap) if shared else (cap, 0, 0)
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        send = torch.full((G * stride,), -7, dtype=torch.int64, device="cuda")
        slot = torch.empty(M, dtype=torch.int32, device="cuda")
        counts = torch.empty(G, dtype=torch.int32, device="cuda")
        ops.bucket_by_owner(idx.cuda(), G, cap, stride, offset, send, slot, counts, aux.cuda() if shared else None,
                            aux_slot, err)
        send, slot, counts = send.cpu().numpy(), slot.cpu().numpy(), counts.cpu().numpy()
        exp_send = np.full(G * stride, -7, dtype=np.int64)          # slots of the other list stay untouched
        for o in range(G):
            exp_send[o * stride + offset:o * stride + offset + cap] = -1
        exp_slot = np.full(M, -1, dtype=np.int32)
        fill = np.zeros(G, dtype=np.int64)
        overflow = False
        for j, r in enumerate(idx.numpy()):
            o = r % G
            if fill[o] < cap:
                exp_send[o * stride + offset + fill[o]] = r // G
                exp_slot[j] = o * stride + offset + fill[o]
                fill[o] += 1
            else:
                overflow = True
        if shared:
            pair = np.array([aux.min().item(), aux.max().item()], dtype=np.float32).view(np.int64)[0]
            for o in range(G):
                exp_send[o * stride + aux_slot] = pair
        np.testing.assert_array_equal(send, exp_send)
        np.testing.assert_array_equal(slot, exp_slot)
        np.testing.assert_array_equal(counts, fill)
        assert bool(int(err.item()) & 4) == overflow


def test_sort_ignores_padding_slots():
    from tests_helpers import sort_segments
    idx = torch.tensor([5, -1, 3, 5, -1, 0, 3], dtype=torch.int64).cuda()
    perm, seg_start, seg_row, nseg, err = sort_segments(idx, 10)
    assert err == 0 and nseg == 3
    assert seg_row[:3].tolist() == [0, 3, 5]
    assert seg_start[:4].tolist() == [0, 1, 3, 5]        # 5 real ids; the two -1 slots belong to no segment
    assert perm[:5].tolist() == [5, 2, 6, 0, 3]


def test_item_owner_kernels_match_their_cpu_doubles():
    """The kernels only the item-owner schedule uses, on the layouts of a 4-rank world (the 1-rank golden runs above see one
    chunk only), against the pure-torch doubles the gloo tests run the same schedule with (tests/cpu_ops.py)."""
    from cpu_ops import CpuOps, TAIL
    from fairrec.sharded import HipOps
    hip, cpu = HipOps("cuda"), CpuOps()
    g = torch.Generator().manual_seed(11)
    G, M, cap, D, n_users, n_items = 4, 3000, 900, 64, 50_000, 7_000
    RS = 4 * cap + 1

    def pair(shape, dtype, fill):
        return torch.full(shape, fill, dtype=dtype), torch.full(shape, fill, dtype=dtype, device="cuda")

    def same(a, b, **kw):
        np.testing.assert_array_equal(a.numpy(), b.cpu().numpy(), **kw)

    item = torch.randint(0, n_items, (M,), generator=g, dtype=torch.int64)
    user = torch.randint(0, n_users, (M,), generator=g, dtype=torch.int64)
    rating = torch.rand(M, generator=g) * 4 + 1
    sst = torch.randint(0, 2, (M,), generator=g).float()

    # stage 1 (sender): items to their owners with the interaction's record beside each
    send_c, send_g = pair((G * RS,), torch.int64, -7)
    slot_c, slot_g = pair((M,), torch.int32, 0)
    cnt_c, cnt_g = pair((G,), torch.int32, 0)
    err_c, err_g = pair((1,), torch.int32, 0)
    cpu.bucket_by_owner(item, G, cap, RS, 0, send_c, slot_c, cnt_c, sst, 4 * cap, err_c)
    hip.bucket_by_owner(item.cuda(), G, cap, RS, 0, send_g, slot_g, cnt_g, sst.cuda(), 4 * cap, err_g)
    cpu.pack_records(slot_c, user, rating, sst, cap, send_c)
    hip.pack_records(slot_g, user.cuda(), rating.cuda(), sst.cuda(), cap, send_g)
    real = torch.zeros(G * RS, dtype=torch.bool)                # record fields beside an empty slot are not defined
    for o in range(G):
        k = int(cnt_c[o])
        for f in range(4):
            real[o * RS + f * cap:o * RS + f * cap + k] = True
        real[o * RS:o * RS + cap] = True
        real[o * RS + 4 * cap] = True
    same(send_c[real], send_g.cpu()[real])
    same(slot_c, slot_g); same(cnt_c, cnt_g)

    # stage 2 (owner): what arrived -> id lists with holes, the users' requests, the distinct-item count
    recv = torch.where(real, send_c, torch.zeros((), dtype=torch.int64))
    n = G * cap
    outs_c = [torch.zeros(n, dtype=torch.int64), torch.zeros(n, dtype=torch.int64), torch.zeros(n, dtype=torch.int32),
              torch.zeros(n), torch.zeros(n), torch.zeros(G, dtype=torch.int64)]
    outs_g = [t.cuda() for t in outs_c]
    cpu.unpack_records(recv, G, cap, *outs_c)
    hip.unpack_records(recv.cuda(), G, cap, *outs_g)
    for a, b in zip(outs_c, outs_g):
        same(a, b)
    iid, uid = outs_c[0], outs_c[1]
    ureq_c, ureq_g = pair((G * cap,), torch.int64, -7)
    uslot_c, uslot_g = pair((n,), torch.int32, 0)
    cpu.bucket_sparse(uid, G, cap, cap, 0, ureq_c, uslot_c, cnt_c, err_c)
    hip.bucket_sparse(uid.cuda(), G, cap, cap, 0, ureq_g, uslot_g, cnt_g, err_g)
    same(ureq_c, ureq_g); same(uslot_c, uslot_g); same(cnt_c, cnt_g)
    assert int(err_c.item()) == int(err_g.item())
    k_c, k_g = pair((1,), torch.float32, 0.0)
    bitmap = torch.zeros((n_items + 31) // 32, dtype=torch.int32, device="cuda")
    count = torch.zeros(1, dtype=torch.int32, device="cuda")
    for rep in range(2):                                        # the bitmap and the counter are left clean
        cpu.count_distinct(iid, n_items, None, None, k_c)
        hip.count_distinct(iid.cuda(), n_items, bitmap, count, k_g)
        same(k_c, k_g)
    assert int(bitmap.abs().sum().item()) == 0 and int(count.item()) == 0

    # stage 3 (owner): predictions and gradient rows from two row buffers
    rows_u, rows_i = torch.randn(n, D, generator=g) * 0.1, torch.randn(n, D, generator=g) * 0.1
    islot = outs_c[2]
    pred_c, pred_g = pair((n,), torch.float32, 0.0)
    coef_c, coef_g = pair((n,), torch.float32, 0.0)
    rec_c, rec_g = pair((G * 3 * cap,), torch.float32, 0.0)
    sq_c, sq_g = pair(((n + 3) // 4,), torch.float32, 0.0)      # one partial per 4 positions
    cpu.shard_score2(rows_u, rows_i, uslot_c, islot, outs_c[3], outs_c[4], G * M, pred_c, coef_c, rec_c, cap, cap, 0, None, sq_c)
    hip.shard_score2(rows_u.cuda(), rows_i.cuda(), uslot_g, islot.cuda(), outs_g[3], outs_g[4], G * M, pred_g, coef_g, rec_g,
                     cap, cap, 0, None, sq_g)
    np.testing.assert_allclose(pred_g.cpu().numpy(), pred_c.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(coef_g.cpu().numpy(), coef_c.numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(rec_g.cpu().numpy(), rec_c.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(float(sq_g.sum().item()), float(sq_c[0]), rtol=1e-5)

    reply = torch.randn(G * (cap + TAIL), generator=g)
    k_all = torch.tensor([700.0, 650.0, 712.0, 3.0])
    rep_c, rep_g = reply.clone(), reply.cuda()
    sums_c, sums_g = pair((2,), torch.float32, 0.0)
    cpu.post_fair(rep_c, k_all, G, cap, sums_c)
    hip.post_fair(rep_g, k_all.cuda(), G, cap, sums_g)
    same(rep_c, rep_g); same(sums_c, sums_g)
    for fair in (True, False):
        loss_c, loss_g = pair((3,), torch.float32, 0.0)
        cpu.loss_finish(sums_c, k_all, G, G * M, 0.3, fair, loss_c)
        hip.loss_finish(sums_g, k_all.cuda(), G, G * M, 0.3, fair, loss_g)
        np.testing.assert_allclose(loss_g.cpu().numpy(), loss_c.numpy(), rtol=1e-6)

    gu_c, gu_g = pair((n, D), torch.float32, 0.0)
    gi_c, gi_g = pair((n, D), torch.float32, 0.0)
    lo_c, lo_g = pair((3,), torch.float32, 0.0)
    cpu.shard_grads2(rows_u, rows_i, uslot_c, islot, coef_c, rep_c, G, G * M, 0.3, lo_c, cap, cap, 0, gu_c, gi_c)
    hip.shard_grads2(rows_u.cuda(), rows_i.cuda(), uslot_g, islot.cuda(), coef_g, rep_g, G, G * M, 0.3, lo_g, cap, cap, 0,
                     gu_g, gi_g)
    touched_u, touched_i = uslot_c[(uslot_c >= 0) & (islot >= 0)].long(), islot[(uslot_c >= 0) & (islot >= 0)].long()
    np.testing.assert_allclose(gu_g.cpu().numpy()[touched_u], gu_c.numpy()[touched_u], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(gi_g.cpu().numpy()[touched_i], gi_c.numpy()[touched_i], rtol=2e-5, atol=1e-9)


# ---- the HIP kernels of the row-sharded step with G > 1 ranks -- on ONE GPU -----------------------------------------------------
class _ThreadWorld:
    """G "ranks" as G threads of this process, all on cuda:0, with torch.distributed's calls in fairrec/sharded.py served from
    shared memory: what a rank would send is left in a slot, a barrier, every rank copies its share.  Exactly one thread runs at
    any time (a baton handed over inside the collectives), so the library sees the single-threaded call sequence it is built
    for, and the device is synchronised around every exchange (ranks produce on side streams).  It stands for RCCL only in WHAT
    moves; what it proves is the kernels' side of a multi-rank step: owner buckets with several chunks, padded gathers, records
    for other owners, replies -- the layouts a 1-rank world never produces."""

    ReduceOp = dist.ReduceOp

    def __init__(self, G):
        import threading
        self.G, self.local = G, threading.local()
        self.slots = [None] * G
        self.barrier = threading.Barrier(G)
        self.baton = threading.Lock()
        self.have, self.current = [False] * G, None

    def get_world_size(self, group=None):
        return self.G

    def _rank(self):
        # (a rank's backward pass calls the collectives from autograd's device thread: it acts for whoever holds the baton)
        r = getattr(self.local, "rank", None)
        return self.current if r is None else r

    def get_rank(self, group=None):
        return self._rank()

    def _wait(self):
        r = self._rank()
        self.have[r] = False
        self.baton.release()
        self.barrier.wait(timeout=120)    # (BrokenBarrierError when another rank failed or never came: without the baton)
        self.baton.acquire()
        self.have[r], self.current = True, r

    def _meet(self, mine):
        torch.cuda.synchronize()
        self.slots[self._rank()] = mine
        self._wait()

    def _part(self):
        torch.cuda.synchronize()
        self._wait()

    def all_to_all_single(self, out, inp, group=None):
        self._meet(inp)
        r, n = self._rank(), inp.shape[0] // self.G
        for j in range(self.G):
            out[j * n:(j + 1) * n].copy_(self.slots[j][r * n:(r + 1) * n])
        self._part()

    def all_reduce(self, t, op=None, group=None):
        self._meet(t.clone())
        acc = self.slots[0].clone()
        for j in range(1, self.G):
            acc = torch.maximum(acc, self.slots[j]) if op == dist.ReduceOp.MAX else acc + self.slots[j]
        t.copy_(acc)
        self._part()

    def all_gather_into_tensor(self, out, inp, group=None):
        self._meet(inp)
        n = inp.shape[0]
        for j in range(self.G):
            out[j * n:(j + 1) * n].copy_(self.slots[j])
        self._part()

    def run(self, fn):
        import threading
        results, errors = [None] * self.G, []

        def body(rank):
            self.local.rank = rank
            self.baton.acquire()
            self.have[rank], self.current = True, rank
            try:
                results[rank] = fn(rank)
            except BaseException as e:       # noqa: BLE001 -- reported below; the others must not wait for this rank forever
                errors.append((rank, e))
                self.barrier.abort()
            finally:
                if self.have[rank]:
                    self.baton.release()

        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.G)]
        for t in threads:
            t.daemon = True          # (a rank that hangs must not keep the test process alive)
            t.start()
        for t in threads:
            t.join(timeout=600)
            assert not t.is_alive(), "a rank of the thread world did not finish"
        real = [(r, e) for r, e in errors if not isinstance(e, __import__("threading").BrokenBarrierError)]
        if real or errors:
            raise (real or errors)[0][1]
        return results


@pytest.mark.parametrize("world,schedule,objective,hot",
                         [(2, "requester", "value", False), (4, "requester", "value", True), (4, "requester", "nonparity", True),
                          (8, "requester", "value", False), (2, "item_owner", "under", False), (4, "item_owner", "value", True)])
def test_hip_kernels_of_a_multi_rank_step_match_the_oracle(world, schedule, objective, hot, monkeypatch):
    """tests/test_sharded_gloo.py's check -- the sharded step on the concatenated global batch equals the single-process oracle,
    losses and both tables -- with the HIP kernels instead of their CPU doubles and `_ThreadWorld` instead of gloo: G = 2, 4, 8
    ranks on one GPU, both schedules, a Zipf-hot owner, look-ahead and inline index work."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import fairrec.sharded as S
    from oracle import focf as O
    from test_sharded_gloo import _case
    tw = _ThreadWorld(world)
    monkeypatch.setattr(S, "dist", tw)
    z = _case(hot, world)
    T, B = 6, z["user_id"].shape[1] // world
    Engine = S.ShardedFocfEngineV2 if schedule == "item_owner" else S.ShardedFocfEngine
    U0, I0 = torch.tensor(z["U0"]), torch.tensor(z["I0"])

    def rank_fn(rank):
        eng = Engine(S.shard_of(U0, rank, world).cuda(), S.shard_of(I0, rank, world).cuda(), objective, 0.8, 1e-3, 1e-3,
                     capacity_factor=4.0 if hot else 1.5)
        sl = slice(rank * B, (rank + 1) * B)
        batches = [[torch.tensor(z[k][t][sl]).cuda() for k in ("user_id", "item_id", "rating", "sst")] for t in range(T)]
        losses = []
        for t in range(T):
            nxt = (batches[t + 1][0], batches[t + 1][1], batches[t + 1][3], batches[t + 1][2]) if t + 1 < T and t % 3 != 2 else None
            loss, _ = eng.forward(*batches[t], next_batch=nxt)
            losses.append(float(loss))
            eng.backward_adam()
        eng.flush()
        eng.check_device_errors()
        torch.cuda.synchronize()
        return {"U": eng.U.weight.cpu(), "I": eng.I.weight.cpu(), "loss": losses}

    parts = tw.run(rank_fn)
    Bg = B * world
    ref = O.train(objective, z["U0"], z["I0"], z["user_id"][:T, :Bg], z["item_id"][:T, :Bg], z["rating"][:T, :Bg],
                  z["sst"][:T, :Bg], 1e-3, 1e-3, 0.8, snaps=(T,))
    for r in range(world):
        np.testing.assert_allclose(parts[r]["loss"], ref["loss"], rtol=1e-4)
    for tag in ("U", "I"):
        full = np.zeros_like(ref[f"{tag}_after{T}"])
        for r in range(world):
            full[r::world] = parts[r][tag].numpy()
        b = ref[f"{tag}_after{T}"]
        assert (np.abs(full - b) <= 1e-4 * np.abs(b) + 1e-6).all(), (tag, np.abs(full - b).max())


def test_hip_kernels_eight_ranks_item_complete_batches_drop_nothing(monkeypatch):
    """tests/test_sharded_gloo.py::test_eight_ranks_item_complete_batches_drop_nothing with the HIP kernels: item-complete
    batches whose items live on two of eight owners overflow the default exchange capacity; the engine notices on the device
    (bucket fill counters, a MAX all-reduce), doubles the capacity, buckets again -- and every step equals the single-device
    step on the concatenated batch."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import fairrec.sharded as S
    from oracle import focf as O
    from test_sharded_gloo import _item_complete_case
    world = 8
    tw = _ThreadWorld(world)
    monkeypatch.setattr(S, "dist", tw)
    z = _item_complete_case(world)
    T, B = z["user_id"].shape[0], z["user_id"].shape[1] // world
    U0, I0 = torch.tensor(z["U0"]), torch.tensor(z["I0"])

    def rank_fn(rank):
        eng = S.ShardedFocfEngine(S.shard_of(U0, rank, world).cuda(), S.shard_of(I0, rank, world).cuda(), "value", 0.8, 1e-3,
                                  1e-3, capacity_factor=2.0)
        sl = slice(rank * B, (rank + 1) * B)
        batches = [[torch.tensor(z[k][t][sl]).cuda() for k in ("user_id", "item_id", "rating", "sst")] for t in range(T)]
        cap0, losses = eng.capacity(B), []
        for t in range(T):
            nxt = (batches[t + 1][0], batches[t + 1][1], batches[t + 1][3], batches[t + 1][2]) if t + 1 < T and t % 2 == 0 else None
            loss, _ = eng.forward(*batches[t], next_batch=nxt)
            losses.append(float(loss))
            eng.backward_adam()
        eng.flush()
        eng.check_device_errors()
        torch.cuda.synchronize()
        return {"U": eng.U.weight.cpu(), "I": eng.I.weight.cpu(), "loss": losses, "cap0": cap0, "cap1": eng.capacity(B),
                "steps": eng.step_count}

    parts = tw.run(rank_fn)
    ref = O.train("value", z["U0"], z["I0"], z["user_id"], z["item_id"], z["rating"], z["sst"], 1e-3, 1e-3, 0.8, snaps=(T,))
    for r in range(world):
        assert parts[r]["steps"] == T and parts[r]["cap1"] > parts[r]["cap0"]
        np.testing.assert_allclose(parts[r]["loss"], ref["loss"], rtol=1e-4)
    for tag in ("U", "I"):
        full = np.zeros_like(ref[f"{tag}_after{T}"])
        for r in range(world):
            full[r::world] = parts[r][tag].numpy()
        b = ref[f"{tag}_after{T}"]
        assert (np.abs(full - b) <= 1e-4 * np.abs(b) + 1e-6).all(), (tag, np.abs(full - b).max())


@pytest.mark.parametrize("world,mode", [(2, "single"), (2, "pair"), (4, "pair"), (8, "pair"), (2, "pair_frozen_user"),
                                        (3, "pair_clip")])
def test_hip_generic_sharded_engine_with_several_ranks_equals_single_process(world, mode, monkeypatch):
    """tests/test_sharded_engine_gloo.py::test_generic_engine_equals_single_process with the HIP table kernels and `_ThreadWorld`:
    the row-sharded GenericEngine (NFCF / PFCN on sharded tables: differentiable sharded lookup, gradient rows back to their
    owners, one flat all-reduce of the replicated dense gradients, the global gradient norm) on G = 2, 3, 4, 8 ranks of one GPU
    against torch autograd on the full tables + the oracle's dense Adam on the concatenated batch."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import fairrec.sharded_engine as SE
    from fairrec.optim import AdamHyper
    from oracle import focf as O
    NU, NI, D, B, T, LR, WD, CLIP = 3701, 2303, 64, 512, 5, 1e-2, 1e-3, 0.05
    g = torch.Generator().manual_seed(11)
    U0, I0 = torch.randn(NU, D, generator=g) * 0.3, torch.randn(NI, D, generator=g) * 0.3
    w0, b0 = torch.randn(D, generator=g) * 0.5, torch.zeros(1)
    u = torch.randint(1, NU, (T, world * B), generator=g)
    i = torch.randint(1, NI, (T, world * B), generator=g)
    r = torch.randn(T, world * B, generator=g)

    def loss_fn(ue, ie, w, b, rr):
        return (((ue * ie) * w).sum(-1) + b - rr).pow(2).mean()

    tw = _ThreadWorld(world)
    monkeypatch.setattr(SE, "dist", tw)
    frozen = mode.startswith("pair_frozen_user")

    def rank_fn(rank):
        eng = SE.ShardedGenericEngine("cuda")
        Us, Is = U0[rank::world].clone().cuda(), I0[rank::world].clone().cuda()
        eng.add_table("U", torch.nn.Parameter(Us, requires_grad=not frozen), trainable=not frozen, n_rows_global=NU)
        eng.add_table("I", torch.nn.Parameter(Is), n_rows_global=NI)
        w, b = torch.nn.Parameter(w0.clone().cuda()), torch.nn.Parameter(b0.clone().cuda())
        eng.add_dense("w", w)
        eng.add_dense("b", b)
        eng.hyper = AdamHyper(LR, WD, device="cuda")
        for t_ in eng._tables.values():
            t_.ensure_state()
        losses, norms = [], []
        sl = slice(rank * B, (rank + 1) * B)
        for t in range(T):
            eng.zero_grad()
            ut, it = u[t][sl].cuda(), i[t][sl].cuda()
            if mode.startswith("single"):
                ue, ie = eng.lookup("U", ut), eng.lookup("I", it)
            else:
                ue, ie = eng.lookup_pair("U", ut, "I", it)
            loss = loss_fn(ue, ie, w, b, r[t][sl].cuda())
            # (autograd's one device thread would serve every rank of this process: rank A's backward waiting in a collective
            # would keep rank B's from ever starting -- the backward pass runs on the rank's own thread here)
            with torch.autograd.set_multithreading_enabled(False):
                loss.backward()
            if mode.endswith("_clip"):
                norms.append(float(eng.clip_grad_norm(CLIP)))
            eng.backward_adam()
            losses.append(float(loss))
        eng.flush()
        torch.cuda.synchronize()
        return {"U": eng._tables["U"].weight.cpu(), "I": eng._tables["I"].weight.cpu(), "w": w.data.cpu(), "b": b.data.cpu(),
                "loss": losses, "norm": norms}

    parts = tw.run(rank_fn)
    P = [torch.nn.Parameter(x.clone()) for x in (U0, I0, w0, b0)]
    if frozen:
        P[0].requires_grad_(False)
    ms, vs = [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P]
    ref_loss, ref_norm = [], []
    for t in range(T):
        for p in P:
            p.grad = None
        loss = loss_fn(P[0][u[t]], P[1][i[t]], P[2], P[3], r[t])
        loss.backward()
        if mode.endswith("_clip"):
            ref_norm.append(float(torch.nn.utils.clip_grad_norm_([p for p in P if p.grad is not None], CLIP)))
        for k, p in enumerate(P):
            if p.grad is not None:
                O.adam_dense_step_(p.data, p.grad, ms[k], vs[k], t + 1, LR, WD)
        ref_loss.append(float(loss))
    np.testing.assert_allclose(np.mean([p["loss"] for p in parts], axis=0), ref_loss, rtol=1e-4)
    if mode.endswith("_clip"):
        assert min(ref_norm) > CLIP
        for q in range(world):
            np.testing.assert_allclose(parts[q]["norm"], ref_norm, rtol=1e-4)
    for tag, ref in (("U", P[0]), ("I", P[1])):
        full = torch.zeros_like(ref.data)
        for q in range(world):
            full[q::world] = parts[q][tag]
        # (Adam at lr = 1e-2 turns a last-bit difference of a near-zero gradient sum -- G partial sums scaled 1/G instead of one
        # sum -- into a few 1e-6 of a step: 1 element of 237 k sat at 3.6e-6 with G = 8)
        a_, b_ = full.numpy(), ref.data.numpy()
        off = np.abs(a_ - b_) > 1e-4 * np.abs(b_) + 1e-5
        # ... and a gradient element near zero whose sign the summation order decides moves its weight by up to 2 lr in Adam's
        # first steps: a handful of elements per table may sit further out (2 of 237 k at 3e-5 with G = 3 and clipping)
        assert off.mean() < 1e-4 and np.abs(a_ - b_).max() < 5e-3, (tag, int(off.sum()), float(np.abs(a_ - b_).max()))
    for q in range(world):
        np.testing.assert_allclose(parts[q]["w"].numpy(), P[2].data.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(parts[q]["b"].numpy(), P[3].data.numpy(), rtol=1e-4, atol=1e-5)
    assert torch.equal(parts[0]["w"], parts[1]["w"])
