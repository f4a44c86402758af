"""CPU, world_size 2 over gloo: the data-parallel schedule of fairrec/replicated_engine.py (FairGo, BASELINE.json
configs[3]: replicated tables, sharded batch, one flat all-reduce of the replicated dense gradients; in the pretrain stage
all-gathered ids / gradient rows so that every replica applies the update of the global batch) with a CPU double for the
kernels equals a single-process run on the concatenated batch (torch autograd + the oracle's dense Adam)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_sharded_engine_gloo import D, LR, NI, NU, T, WD, B, _data, _free_port, _loss

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Table:                      # CPU double of a replicated lazy table: dense oracle Adam, torch indexing
    def __init__(self, weight, trainable=True):
        self.weight, self.trainable = weight, trainable
        self.n_rows, self.dim = weight.shape
        self.m, self.v = torch.zeros_like(weight), torch.zeros_like(weight)
        self.step, self._pending, self._grad_rows, self.ids = 0, None, None, None

    def ensure_state(self):
        pass


class _Ops:
    def gather_train(self, table, hyper, ids, M, rows, err):
        table.ids = ids.clone()
        rows.copy_(table.weight[ids])
        table._pending = (M, None)

    def gather(self, table, hyper, ids, M, rows, err):
        rows.copy_(table.weight[ids])

    def apply_grad(self, table, hyper, M, rows, grads, sweep):
        from oracle import focf as O
        g = torch.zeros_like(table.weight)
        g.index_add_(0, table.ids, grads)
        table.step += 1
        O.adam_dense_step_(table.weight, g, table.m, table.v, table.step, hyper.lr, hyper.weight_decay)
        table._pending = None

    def adam_dense(self, p, g, m, v, hyper, step):
        from oracle import focf as O
        O.adam_dense_step_(p, g.clone(), m, v, step, hyper.lr, hyper.weight_decay)


CLIP = 0.05     # well below the gradient norms of the toy problem: every step is clipped


def _worker(rank, world, port, out_dir, frozen, clip=False, two_groups=False):
    for p in (ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairrec.optim import AdamHyper
        from fairrec.replicated_engine import ReplicatedGenericEngine
        U0, I0, w0, b0, u, i, r = _data()
        eng = ReplicatedGenericEngine("cpu", ops=_Ops())
        Ur, Ir = U0.clone(), I0.clone()                       # full replicas
        eng.add_table("U", torch.nn.Parameter(Ur), table=_Table(Ur, trainable=not frozen))
        eng.add_table("I", torch.nn.Parameter(Ir), table=_Table(Ir, trainable=not frozen))
        w, b = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(b0.clone())
        eng.add_dense("w", w, group="ga" if two_groups else None)
        eng.add_dense("b", b, group="gb" if two_groups else None)
        stepping = "ga" if two_groups else None
        eng.hyper = AdamHyper(LR, WD, device="cpu")
        losses, norms = [], []
        for t in range(T):
            sl = slice(rank * B, (rank + 1) * B)
            eng.zero_grad()
            loss = _loss(eng.lookup("U", u[t][sl]), eng.lookup("I", i[t][sl]), w, b, r[t][sl])
            loss.backward()
            if clip:      # what FusedLazyAdam.step() does when the config clips
                norms.append(float(eng.clip_grad_norm(CLIP, stepping)))
            eng.backward_adam(stepping)
            losses.append(float(loss))
        torch.save({"U": Ur, "I": Ir, "w": w.data, "b": b.data, "loss": losses, "norm": norms}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _reference(frozen, clip=False, stepped=(0, 1, 2, 3)):
    from oracle import focf as O
    U0, I0, w0, b0, u, i, r = _data()
    P = [torch.nn.Parameter(x.clone(), requires_grad=not (frozen and k < 2)) for k, x in enumerate((U0, I0, w0, b0))]
    ms, vs = [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P]
    ref_loss, ref_norm = [], []
    for t in range(T):
        for p in P:
            p.grad = None
        loss = _loss(P[0][u[t]], P[1][i[t]], P[2], P[3], r[t])
        loss.backward()
        if clip:
            ref_norm.append(float(torch.nn.utils.clip_grad_norm_([p for p in P if p.grad is not None], CLIP)))
        for k, p in enumerate(P):
            if p.grad is not None and k in stepped:
                O.adam_dense_step_(p.data, p.grad, ms[k], vs[k], t + 1, LR, WD)
        ref_loss.append(float(loss))
    return (P, ref_loss, ref_norm) if clip else (P, ref_loss)


def _check(tmp_path, frozen):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), frozen), nprocs=world, join=True)
    P, ref_loss = _reference(frozen)
    parts = [torch.load(os.path.join(str(tmp_path), f"r{q}.pt")) for q in range(world)]
    np.testing.assert_allclose(np.mean([p["loss"] for p in parts], axis=0), ref_loss, rtol=1e-5)
    for q in range(world):      # every replica equals the single-process parameters ...
        for tag, ref in (("U", P[0]), ("I", P[1]), ("w", P[2]), ("b", P[3])):
            np.testing.assert_allclose(parts[q][tag].numpy(), ref.data.numpy(), rtol=2e-5, atol=1e-7, err_msg=tag)
    for tag in ("U", "I", "w", "b"):      # ... and the replicas are bit-identical to each other
        assert torch.equal(parts[0][tag], parts[1][tag]), tag


def test_two_replicas_with_frozen_tables_equal_single_process(tmp_path):
    """The finetune stage of FairGo: tables frozen, only the replicated dense parameters train."""
    _check(tmp_path, frozen=True)


def test_two_replicas_training_the_tables_equal_single_process(tmp_path):
    """The pretrain stage: the tables train on the global batch (all-gathered ids and gradient rows)."""
    _check(tmp_path, frozen=False)


@pytest.mark.parametrize("frozen", [False, True], ids=["tables_train", "tables_frozen"])
def test_clip_grad_norm_on_replicas_is_the_single_process_clip(tmp_path, frozen):
    """config clip_grad_norm with data_parallel replicas: every replica must measure the norm of the GLOBAL batch's gradient
    (tables: the all-gathered rows, duplicates summed; dense: after the flat all-reduce) and apply ONE coefficient -- the
    step of torch.nn.utils.clip_grad_norm_ on the concatenated batch."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), frozen, True), nprocs=world, join=True)
    P, ref_loss, ref_norm = _reference(frozen, clip=True)
    assert min(ref_norm) > CLIP
    parts = [torch.load(os.path.join(str(tmp_path), f"r{q}.pt")) for q in range(world)]
    np.testing.assert_allclose(np.mean([p["loss"] for p in parts], axis=0), ref_loss, rtol=1e-5)
    for q in range(world):
        np.testing.assert_allclose(parts[q]["norm"], ref_norm, rtol=2e-5)
        for tag, ref in (("U", P[0]), ("I", P[1]), ("w", P[2]), ("b", P[3])):
            np.testing.assert_allclose(parts[q][tag].numpy(), ref.data.numpy(), rtol=2e-5, atol=1e-7, err_msg=tag)
    for tag in ("U", "I", "w", "b"):
        assert torch.equal(parts[0][tag], parts[1][tag]), tag


def test_clip_grad_norm_on_replicas_with_two_optimizer_groups(tmp_path):
    """Per-group optimizers + clip_grad_norm (the PFCN / FairGo trainers): the norm covers model parameters the stepping
    optimizer does not own (torch's clip_grad_norm_(model.parameters()) does), so their gradients must be averaged over
    the replicas as well -- measured from each rank's local gradient they gave every rank its own coefficient."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), True, True, True), nprocs=world, join=True)
    P, ref_loss, ref_norm = _reference(True, clip=True, stepped=(2,))
    parts = [torch.load(os.path.join(str(tmp_path), f"r{q}.pt")) for q in range(world)]
    for q in range(world):
        np.testing.assert_allclose(parts[q]["norm"], ref_norm, rtol=2e-5)
        np.testing.assert_allclose(parts[q]["w"].numpy(), P[2].data.numpy(), rtol=2e-5, atol=1e-7)
        np.testing.assert_array_equal(parts[q]["b"].numpy(), P[3].data.numpy())          # never stepped
    assert parts[0]["norm"] == parts[1]["norm"] and torch.equal(parts[0]["w"], parts[1]["w"])
