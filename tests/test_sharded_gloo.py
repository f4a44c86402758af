"""CPU, world_size 2 over gloo: the exchange schedule of the row-sharded FOCF step (fairrec/sharded.py) with a
CPU test double for the kernels reproduces the single-process oracle on the concatenated global batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, objective, out_dir):
    for p in (ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_ops import CpuOps
        from fairrec.sharded import ShardedFocfEngine, shard_of
        z = np.load(os.path.join(ROOT, "tests", "golden", "focf_value_d64.npz"))
        U0, I0 = torch.tensor(z["U0"]), torch.tensor(z["I0"])
        eng = ShardedFocfEngine(shard_of(U0, rank, world), shard_of(I0, rank, world), objective, 0.8, 1e-3, 1e-3,
                                ops=CpuOps(), capacity_factor=1.5)
        T, B = 6, z["user_id"].shape[1] // world
        losses = []
        sl = slice(rank * B, (rank + 1) * B)
        batches = [[torch.tensor(z[k][t][sl]) for k in ("user_id", "item_id", "rating", "sst")] for t in range(T)]
        for t in range(T):
            # every other step names its successor: both the look-ahead and the inline index path are exercised
            nxt = (batches[t + 1][0], batches[t + 1][1], batches[t + 1][3]) if t + 1 < T and t % 3 != 2 else None
            loss, _ = eng.forward(*batches[t], next_batch=nxt)
            losses.append(float(loss))
            eng.backward_adam()
        assert int(eng.err.item()) == 0
        torch.save({"U": eng.U.weight, "I": eng.I.weight, "loss": losses}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("objective", ["none", "value", "under", "nonparity"])
def test_two_rank_schedule_matches_oracle(tmp_path, objective):
    from oracle import focf as O
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, objective, str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(ROOT, "tests", "golden", "focf_value_d64.npz"))
    T, B = 6, (z["user_id"].shape[1] // world) * world
    ref = O.train(objective, z["U0"], z["I0"], z["user_id"][:T, :B], z["item_id"][:T, :B], z["rating"][:T, :B],
                  z["sst"][:T, :B], 1e-3, 1e-3, 0.8, snaps=(T,))
    parts = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt")) for r in range(world)]
    np.testing.assert_allclose(parts[0]["loss"], ref["loss"], rtol=1e-5)
    np.testing.assert_allclose(parts[1]["loss"], ref["loss"], rtol=1e-5)
    for tag in ("U", "I"):
        full = np.zeros_like(ref[f"{tag}_after{T}"])
        for r in range(world):
            full[r::world] = parts[r][tag].numpy()
        np.testing.assert_allclose(full, ref[f"{tag}_after{T}"], rtol=1e-5, atol=1e-7)
