"""bench.py's N-rank FOCF step loop at the BASELINE sizes (B = 8192 per rank, 1 000 001 x 100 001, D = 64) with G ranks as
threads on ONE GPU (tests/test_sharded_hip.py::_ThreadWorld): no RCCL, no timing claim -- a functional run of what `bench.py
--gpus G` executes per rank (shards, look-ahead, both schedules), with the device error word and the losses checked.
python scratch/thread_world_bench.py [G] [steps] [schedule]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "tests")]
import torch
import bench
import fairrec.sharded as S
from test_sharded_hip import _ThreadWorld

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
schedule = sys.argv[3] if len(sys.argv) > 3 else "requester"
tw = _ThreadWorld(G)
S.dist = tw
dev = torch.device("cuda", 0)


def rank_fn(rank):
    u, i, r, s = (t.to(dev) for t in bench.synth_batches(K, bench.BATCH, bench.N_USERS, bench.N_ITEMS, bench.SEED + rank, "uniform"))
    g = torch.Generator(device="cpu").manual_seed(bench.SEED + 1 + 1000 * rank)
    Us = (torch.randn(S.shard_rows(bench.N_USERS, rank, G), bench.DIM, generator=g) * math.sqrt(2.0 / (bench.N_USERS + bench.DIM))).to(dev)
    Is = (torch.randn(S.shard_rows(bench.N_ITEMS, rank, G), bench.DIM, generator=g) * math.sqrt(2.0 / (bench.N_ITEMS + bench.DIM))).to(dev)
    Eng = S.ShardedFocfEngineV2 if schedule == "item_owner" else S.ShardedFocfEngine
    eng = Eng(Us, Is, bench.OBJECTIVE, bench.FAIR_WEIGHT, bench.LR, bench.WD)
    losses = []
    for k in range(K):
        nxt = (u[k + 1], i[k + 1], s[k + 1], r[k + 1]) if k + 1 < K else None
        loss, _ = eng.forward(u[k], i[k], r[k], s[k], next_batch=nxt)
        losses.append(float(loss))
        eng.backward_adam()
    eng.flush()
    eng.check_device_errors()
    torch.cuda.synchronize()
    return losses, eng.capacity(bench.BATCH), float(eng.U.weight.abs().max()), bool(torch.isfinite(eng.U.weight).all() and torch.isfinite(eng.I.weight).all())


t0 = time.time()
out = tw.run(rank_fn)
print(f"G={G} schedule={schedule} steps={K}: {time.time() - t0:.1f} s")
for rank, (losses, cap, umax, finite) in enumerate(out):
    print(f"rank {rank}: loss {losses[0]:.5f} .. {losses[-1]:.5f}  capacity {cap}  max|U| {umax:.4f}  finite {finite}")
assert all(o[3] for o in out)
assert all(abs(o[0][-1] - out[0][0][-1]) <= 1e-5 * abs(out[0][0][-1]) for o in out), "ranks disagree on the global loss"
print("ok")
