"""CPU, world_size 2 over gloo: the exchange schedule of the row-sharded GenericEngine (fairrec/sharded_engine.py:
differentiable sharded lookup, gradient rows back to the owners scaled 1/G, flat all-reduce of the replicated dense
gradients) with a CPU test double for the kernels equals a single-process run on the concatenated batch (torch autograd
on the full tables + the oracle's dense Adam)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NU, NI, D, B, T, LR, WD = 37, 23, 8, 24, 5, 1e-2, 1e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(world=2):
    g = torch.Generator().manual_seed(11)
    U0, I0 = torch.randn(NU, D, generator=g) * 0.3, torch.randn(NI, D, generator=g) * 0.3
    w0, b0 = torch.randn(D, generator=g) * 0.5, torch.zeros(1)
    u = torch.randint(1, NU, (T, world * B), generator=g)
    i = torch.randint(1, NI, (T, world * B), generator=g)
    r = torch.randn(T, world * B, generator=g)
    return U0, I0, w0, b0, u, i, r


def _loss(ue, ie, w, b, r):
    return (((ue * ie) * w).sum(-1) + b - r).pow(2).mean()


class _Table:                      # CPU double of a lazy table shard: dense oracle Adam, torch indexing
    def __init__(self, weight, trainable=True):
        self.weight, self.trainable = weight, trainable
        self.n_rows, self.dim = weight.shape
        self.m, self.v = torch.zeros_like(weight), torch.zeros_like(weight)
        self.step, self._pending, self._grad_rows, self.ids = 0, None, None, None

    def ensure_state(self):
        pass


class _Ops:                        # CPU double of HipTableOps
    def bucket(self, idx, G, cap, send, slot, counts, err):
        send.fill_(-1)
        counts.zero_()
        for j, r in enumerate(idx.tolist()):
            o, k = r % G, int(counts[r % G])
            assert k < cap
            send[o * cap + k], slot[j] = r // G, o * cap + k
            counts[o] += 1

    def gather_train(self, table, hyper, ids, M, rows, err):
        table.ids = ids.clone()
        ok = ids >= 0
        rows.zero_()
        rows[ok] = table.weight[ids[ok]]
        table._pending = (M, None)

    def gather(self, table, hyper, ids, M, rows, err):
        ok = ids >= 0
        rows.zero_()
        rows[ok] = table.weight[ids[ok]]

    def unbucket_rows(self, src, slot, M, D, out):
        out.copy_(src[slot.long()])

    def bucket_rows(self, src, scale, slot, M, D, dst):
        dst[slot.long()] = src * scale[:, None]

    def apply_grad(self, table, hyper, M, rows, grads, sweep):
        from oracle import focf as O
        g = torch.zeros_like(table.weight)
        ok = table.ids >= 0
        g.index_add_(0, table.ids[ok], grads[ok])
        table.step += 1
        O.adam_dense_step_(table.weight, g, table.m, table.v, table.step, hyper.lr, hyper.weight_decay)
        table._pending = None

    # --- one table's part of a packed exchange buffer: chunk g = `chunk` slots at offset `off`, chunks `stride` apart
    def bucket_at(self, idx, G, cap, stride, off, send, slot, counts, err):
        counts.zero_()
        sv = send.view(G, stride)
        sv[:, off:off + cap] = -1
        for j, r in enumerate(idx.tolist()):
            o, k = r % G, int(counts[r % G])
            assert k < cap
            sv[o, off + k], slot[j] = r // G, o * stride + off + k
            counts[o] += 1

    def gather_train_at(self, table, hyper, ids, off, M, chunk, stride, rows, err):
        G = M // chunk
        mine = ids.view(G, stride)[:, off:off + chunk].reshape(-1).clone()
        table.ids = mine
        ok = mine >= 0
        out = torch.zeros((M, table.dim))
        out[ok] = table.weight[mine[ok]]
        rows.view(G, stride, table.dim)[:, off:off + chunk] = out.view(G, chunk, table.dim)
        table._pending = (M, None)

    def apply_grad_at(self, table, hyper, M, rows, grads, off, chunk, stride, sweep):
        G = M // chunk
        g = grads.view(G, stride, table.dim)[:, off:off + chunk].reshape(M, table.dim)
        self.apply_grad(table, hyper, M, None, g, sweep)

    def adam_dense(self, p, g, m, v, hyper, step):
        from oracle import focf as O
        O.adam_dense_step_(p, g.clone(), m, v, step, hyper.lr, hyper.weight_decay)

    def bpr_outer_rect(self, a, c, inv, loss, da, dc, ws_holder):      # CPU double of fr_bpr_outer_rect
        x = (a[None, :] + c[:, None]).double()
        sig = torch.sigmoid(x)
        loss[0] = float((-torch.log(1e-10 + sig)).sum() * inv)
        d = -(sig * (1 - sig)) / (1e-10 + sig) * inv
        if da is not None:
            da.copy_(d.sum(0).float())
        if dc is not None:
            dc.copy_(d.sum(1).float())

    # --- CPU doubles of fr_nfcf_df_pack / _owner / _apply (include/fairrec_hip.h: buffers [G, cap + 1, 2] and [G, cap + 1, 4])
    def df_workspace(self, B, n_slots, device):
        return torch.zeros(8, dtype=torch.uint8)

    def df_pack(self, out, label, sst, slot, S, off, cap, G, rec, ws):
        r = rec.view(G, cap + 1, 2)
        o, k = (slot // S).long(), (slot % S - off).long()
        pos = label == 1
        r[o, k, 0] = torch.where(pos, out, torch.full_like(out, -1.0))
        r[o, k, 1] = sst
        r[:, cap, 0] = sst[pos].min() if pos.any() else float("inf")
        r[:, cap, 1] = sst[pos].max() if pos.any() else float("-inf")

    def df_owner(self, table, rec, G, cap, reply, ws, B, err):
        r, rp = rec.view(G, cap + 1, 2), reply.view(G, cap + 1, 4)
        smin, smax = float(r[:, cap, 0].min()), float(r[:, cap, 1].max())
        ids = table.ids.view(-1)                                    # [G * cap] received ids, -1 = padding
        score, s = r[:, :cap, 0].reshape(-1), r[:, :cap, 1].reshape(-1)
        valid = (ids >= 0) & (score >= 0)
        flat = rp[:, :cap].reshape(-1, 4)
        K = 0
        for it in torch.unique(ids[valid]).tolist():
            mem = torch.nonzero(valid & (ids == it)).view(-1)       # ascending slot = rank, then batch position
            g0 = s[mem] == smin
            st = torch.tensor([float(score[mem][g0].sum()), float(score[mem][~g0].sum()), float(g0.sum()), float((~g0).sum())])
            flat[mem] = st
            flat[mem[0], 0] = -st[0] if st[0] > 0 else -0.0          # the sign bit: this member reports the item's eps
            K += 1
        rp[:, :cap] = flat.view(G, cap, 4)
        rp[:, cap] = torch.tensor([float(K), smin, smax, 0.0])

    def df_apply(self, reply, slot, S, off, cap, G, out, label, sst, fair_weight, scale, dy, loss, ws):
        rp = reply.view(G, cap + 1, 4)
        K, smin, smax = float(rp[:, cap, 0].sum()), float(rp[0, cap, 1]), float(rp[0, cap, 2])
        if K <= 0 or smin == smax:
            loss[2] = 0.0
            return
        st = rp[(slot // S).long(), (slot % S - off).long()]
        pos = label == 1
        rep = torch.signbit(st[:, 0]) & pos
        S0, S1, n0, n1 = st[:, 0].abs(), st[:, 1], st[:, 2], st[:, 3]
        M0, M1 = (S0 + 1.0 / K) / (n0 + 1.0), (S1 + 1.0 / K) / (n1 + 1.0)
        d = torch.log(M0) - torch.log(M1)
        g = torch.where(sst == smin, fair_weight * torch.sign(d) / K / M0 / (n0 + 1.0),
                        -fair_weight * torch.sign(d) / K / M1 / (n1 + 1.0))
        dy += torch.where(pos, scale * g * out * (1 - out), torch.zeros_like(out))
        df = scale * float(d.abs()[rep].sum()) / K
        loss[0] += fair_weight * df
        loss[2] = df


def _worker(rank, world, port, out_dir, mode="single"):
    for p in (ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairrec.optim import AdamHyper
        from fairrec.sharded_engine import ShardedGenericEngine
        U0, I0, w0, b0, u, i, r = _data(world)
        eng = ShardedGenericEngine("cpu", ops=_Ops())
        Us, Is = U0[rank::world].clone(), I0[rank::world].clone()
        frozen = mode.startswith("pair_frozen_user")      # NFCF finetune: the user table is read-only
        eng.add_table("U", torch.nn.Parameter(Us, requires_grad=not frozen), table=_Table(Us, trainable=not frozen),
                      n_rows_global=NU)
        eng.add_table("I", torch.nn.Parameter(Is), table=_Table(Is), n_rows_global=NI)
        w, b = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(b0.clone())
        eng.add_dense("w", w)
        eng.add_dense("b", b)
        eng.hyper = AdamHyper(LR, WD, device="cpu")
        losses, norms = [], []
        for t in range(T):
            sl = slice(rank * B, (rank + 1) * B)
            eng.zero_grad()
            if mode.startswith("single"):
                ue, ie = eng.lookup("U", u[t][sl]), eng.lookup("I", i[t][sl])
            else:       # both tables through ONE packed exchange per direction
                ue, ie = eng.lookup_pair("U", u[t][sl], "I", i[t][sl])
            loss = _loss(ue, ie, w, b, r[t][sl])
            loss.backward()
            if mode.endswith("_clip"):      # what FusedLazyAdam.step() does when the config clips
                norms.append(float(eng.clip_grad_norm(CLIP)))
            eng.backward_adam()
            losses.append(float(loss))
        torch.save({"U": Us, "I": Is, "w": w.data, "b": b.data, "loss": losses, "norm": norms},
                   os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


CLIP = 0.05     # well below the gradient norms of the toy problem: every step is clipped


@pytest.mark.parametrize("world,mode", [(2, "single"), (2, "pair"), (4, "pair"), (2, "pair_frozen_user"),
                                        (2, "single_clip"), (3, "pair_clip"), (2, "pair_frozen_user_clip")])
def test_generic_engine_equals_single_process(tmp_path, world, mode):
    from oracle import focf as O
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    U0, I0, w0, b0, u, i, r = _data(world)
    P = [torch.nn.Parameter(x.clone()) for x in (U0, I0, w0, b0)]
    if mode.startswith("pair_frozen_user"):
        P[0].requires_grad_(False)
    ms, vs = [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P]
    ref_loss, ref_norm = [], []
    for t in range(T):
        for p in P:
            p.grad = None
        loss = _loss(P[0][u[t]], P[1][i[t]], P[2], P[3], r[t])
        loss.backward()
        if mode.endswith("_clip"):
            ref_norm.append(float(torch.nn.utils.clip_grad_norm_([p for p in P if p.grad is not None], CLIP)))
        for k, p in enumerate(P):
            if p.grad is not None:
                O.adam_dense_step_(p.data, p.grad, ms[k], vs[k], t + 1, LR, WD)
        ref_loss.append(float(loss))
    parts = [torch.load(os.path.join(str(tmp_path), f"r{q}.pt")) for q in range(world)]
    # the global loss is the mean of the local means (equal batch sizes)
    np.testing.assert_allclose(np.mean([p["loss"] for p in parts], axis=0), ref_loss, rtol=1e-5)
    if mode.endswith("_clip"):      # every rank measures the norm of the GLOBAL batch's gradient
        assert min(ref_norm) > CLIP
        for q in range(world):
            np.testing.assert_allclose(parts[q]["norm"], ref_norm, rtol=2e-5)
    for tag, ref in (("U", P[0]), ("I", P[1])):
        full = torch.zeros_like(ref.data)
        for q in range(world):
            full[q::world] = parts[q][tag]
        np.testing.assert_allclose(full.numpy(), ref.data.numpy(), rtol=2e-5, atol=1e-7)
    for q in range(world):      # replicas stay identical and equal the single-process parameters
        np.testing.assert_allclose(parts[q]["w"].numpy(), P[2].data.numpy(), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(parts[q]["b"].numpy(), P[3].data.numpy(), rtol=2e-5, atol=1e-7)
    assert torch.equal(parts[0]["w"], parts[1]["w"])


def _reset_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairrec.config import Config
        from fairrec.data.interaction import Interaction
        from fairrec.model.fair_recommender.nfcf import NFCF
        z = np.load(os.path.join(ROOT, "tests", "golden", "nfcf_finetune.npz"))
        n_users, D_ = z["pretrain_user_embedding"].shape
        n_items = z["init.item_embedding.weight"].shape[0]

        class DS:
            def num(self, f):
                return {"user_id": n_users, "item_id": n_items}[f]

            def get_user_feature(self):
                return Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(z["gender"])})

        ck = os.path.join(out_dir, f"pre{rank}.pth")       # a rank's checkpoint holds its shard (row = local * world + rank)
        torch.save({"state_dict": {"user_embedding.weight": torch.tensor(z["pretrain_user_embedding"])[rank::world].contiguous()}},
                   ck)
        cfg = Config(model="NFCF", config_dict={"embedding_size": D_, "mlp_hidden_size": [int(h) for h in z["hidden"]],
                                                "device": "cpu", "load_pretrain_path": ck, "row_sharded": True})
        m = NFCF(cfg, DS())
        np.save(os.path.join(out_dir, f"user{rank}.npy"), m.user_embedding.weight.detach().numpy())
        assert not m.user_embedding.weight.requires_grad
        assert m.item_embedding.weight.shape[0] == len(range(rank, n_items, world))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reset_params_on_row_sharded_tables_matches_the_reference(tmp_path, world):
    """NFCF.reset_params with the user table row-sharded: the two group means (and counts) of the gender projection come from
    one all-reduce over the ranks' shards; every rank's shard must equal the rows it owns of the table the
    reference's reset_params produced (golden), the [PAD] row 0 untouched."""
    mp.spawn(_reset_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(ROOT, "tests", "golden", "nfcf_finetune.npz"))
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f"user{rank}.npy"))
        np.testing.assert_allclose(got, z["init.user_embedding.weight"][rank::world], rtol=1e-5, atol=1e-6)


FW = 0.3


def _df_data(world):
    g = torch.Generator().manual_seed(23)
    lab = (torch.rand(T, world * B, generator=g) < 0.6).float()
    gender = (torch.rand(NU, generator=g) < 0.5).float()
    gender[1:3] = torch.tensor([0.0, 1.0])
    return lab, gender


def _df_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch.nn.functional as F
        from fairrec.optim import AdamHyper
        from fairrec.sharded_engine import ShardedGenericEngine
        U0, I0, w0, b0, u, i, r = _data(world)
        lab, gender = _df_data(world)
        eng = ShardedGenericEngine("cpu", ops=_Ops())
        Us, Is = U0[rank::world].clone(), I0[rank::world].clone()
        eng.add_table("U", torch.nn.Parameter(Us, requires_grad=False), table=_Table(Us, trainable=False), n_rows_global=NU)
        eng.add_table("I", torch.nn.Parameter(Is), table=_Table(Is), n_rows_global=NI)
        w, b = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(b0.clone())
        eng.add_dense("w", w)
        eng.add_dense("b", b)
        eng.hyper = AdamHyper(LR, WD, device="cpu")
        losses = []
        for t in range(T):
            sl = slice(rank * B, (rank + 1) * B)
            eng.zero_grad()
            ue, ie = eng.lookup_pair("U", u[t][sl], "I", i[t][sl])          # NFCF finetune: the user table is frozen
            y = ((ue * ie) * w).sum(-1) + b
            out = torch.sigmoid(y)
            label, sst = lab[t][sl], gender[u[t][sl]]
            dy, l3 = torch.zeros(B), torch.zeros(3)
            eng.global_item_df("I", out.detach(), label, sst, FW, dy, l3)
            loss = F.binary_cross_entropy(out, label) + (y * dy).sum() - (y * dy).sum().detach() + l3[0]
            loss.backward()
            eng.backward_adam()
            losses.append(float(loss))
        torch.save({"I": Is, "w": w.data, "b": b.data, "loss": losses}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_global_differential_fairness_equals_single_process(tmp_path, world):
    """NFCF finetune shape on row-sharded tables: BCE + fair_weight * differential fairness (nfcf.py:76-97).  M[k, g], K and
    the mean of eps are statistics of the GLOBAL batch: with `global_item_df` (records to the items' owners, per-group sums
    back, K and the groups present in the tails) the G-rank step must equal the single-process step on the concatenated batch
    -- same losses (mean of the ranks' losses), same item table, same dense parameters."""
    import torch.nn.functional as F
    from oracle import focf as O
    from oracle import nfcf as ON
    mp.spawn(_df_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    U0, I0, w0, b0, u, i, r = _data(world)
    lab, gender = _df_data(world)
    P = [torch.nn.Parameter(x.clone()) for x in (I0, w0, b0)]
    ms, vs = [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P]
    ref_loss = []
    for t in range(T):
        for p in P:
            p.grad = None
        out = torch.sigmoid(((U0[u[t]] * P[0][i[t]]) * P[1]).sum(-1) + P[2])
        loss = F.binary_cross_entropy(out, lab[t]) + FW * ON.differential_fairness(out, lab[t], gender[u[t]], i[t])
        loss.backward()
        for k, p in enumerate(P):
            O.adam_dense_step_(p.data, p.grad, ms[k], vs[k], t + 1, LR, WD)
        ref_loss.append(float(loss))
    parts = [torch.load(os.path.join(str(tmp_path), f"r{q}.pt")) for q in range(world)]
    np.testing.assert_allclose(np.mean([p["loss"] for p in parts], axis=0), ref_loss, rtol=1e-5)
    full = torch.zeros_like(I0)
    for q in range(world):
        full[q::world] = parts[q]["I"]
    np.testing.assert_allclose(full.numpy(), P[0].data.numpy(), rtol=2e-5, atol=1e-7)
    for q in range(world):
        np.testing.assert_allclose(parts[q]["w"].numpy(), P[1].data.numpy(), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(parts[q]["b"].numpy(), P[2].data.numpy(), rtol=2e-5, atol=1e-7)


def _bpr_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairrec.optim import AdamHyper
        from fairrec.sharded_engine import ShardedGenericEngine
        U0, I0, w0, b0, u, i, r = _data(world)
        g = torch.Generator().manual_seed(31)
        bias0 = torch.randn(NI, 1, generator=g) * 0.2
        neg = torch.randint(1, NI, (T, world * B), generator=g)
        eng = ShardedGenericEngine("cpu", ops=_Ops())
        Us, Is, Bs = U0[rank::world].clone(), I0[rank::world].clone(), bias0[rank::world].clone()
        eng.add_table("U", torch.nn.Parameter(Us), table=_Table(Us), n_rows_global=NU)
        eng.add_table("I", torch.nn.Parameter(Is), table=_Table(Is), n_rows_global=NI)
        eng.add_table("bias", torch.nn.Parameter(Bs), table=_Table(Bs), n_rows_global=NI)
        eng.hyper = AdamHyper(LR, WD, device="cpu")
        losses = []
        for t in range(T):
            sl = slice(rank * B, (rank + 1) * B)
            eng.zero_grad()
            items = torch.cat([i[t][sl], neg[t][sl]])
            ue, ie, ib = eng.lookup("U", u[t][sl]), eng.lookup("I", items), eng.lookup("bias", items).reshape(-1)
            a = (ue * ie[:B]).sum(-1) - (ue * ie[B:]).sum(-1)
            c = ib[:B] - ib[B:]
            loss_v, da, dc = eng.global_bpr_broadcast(a.detach().contiguous(), c.detach().contiguous())
            loss = (a * da).sum() - (a * da).sum().detach() + (c * dc).sum() - (c * dc).sum().detach() + loss_v[0]
            loss.backward()
            eng.backward_adam()
            losses.append(float(loss))
        torch.save({"U": Us, "I": Is, "bias": Bs, "loss": losses}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_global_bpr_broadcast_equals_single_process(tmp_path, world):
    """PFCN_BiasedMF's training loss (pfcn_biasedmf.py:192-195: `[B] + [B, 1] -> [B, B]`, mean over all pairs (i, j)) on
    row-sharded tables: with `global_bpr_broadcast` the G-rank step equals the single-process step on the concatenated batch
    -- the (G B)^2 matrix of the GLOBAL batch, not G matrices of B^2."""
    from oracle import focf as O
    mp.spawn(_bpr_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    U0, I0, w0, b0, u, i, r = _data(world)
    g = torch.Generator().manual_seed(31)
    bias0 = torch.randn(NI, 1, generator=g) * 0.2
    neg = torch.randint(1, NI, (T, world * B), generator=g)
    P = [torch.nn.Parameter(x.clone()) for x in (U0, I0, bias0)]
    ms, vs = [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P]
    ref_loss = []
    for t in range(T):
        for p in P:
            p.grad = None
        ue = P[0][u[t]]
        a = (ue * P[1][i[t]]).sum(-1) - (ue * P[1][neg[t]]).sum(-1)                      # [G B]
        c = P[2][i[t]] - P[2][neg[t]]                                                   # [G B, 1]
        loss = -torch.log(1e-10 + torch.sigmoid(a + c)).mean()                          # the reference's broadcast
        loss.backward()
        for k, p in enumerate(P):
            O.adam_dense_step_(p.data, p.grad, ms[k], vs[k], t + 1, LR, WD)
        ref_loss.append(float(loss))
    parts = [torch.load(os.path.join(str(tmp_path), f"r{q}.pt")) for q in range(world)]
    np.testing.assert_allclose(np.mean([p["loss"] for p in parts], axis=0), ref_loss, rtol=1e-5)
    for tag, ref in (("U", P[0]), ("I", P[1]), ("bias", P[2])):
        full = torch.zeros_like(ref.data)
        for q in range(world):
            full[q::world] = parts[q][tag]
        np.testing.assert_allclose(full.numpy(), ref.data.numpy(), rtol=2e-5, atol=1e-7, err_msg=tag)
