"""Workload for rocprofv3 counter passes: 60 aged + 120 fused FOCF steps of the bench workload (fr_focf_step)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import bench
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam
dev = torch.device("cuda")
K = int(os.environ.get("PMC_STEPS", "260"))
BATCH = int(os.environ.get("PMC_BATCH", bench.BATCH))          # (a tiny batch leaves the sweeper waves alone in the launch)
SWEEP = os.environ.get("PMC_SWEEP")                              # "0": no sweeper (then the rows' replays grow step by step)
u, i, r, s = (t.to(dev) for t in bench.synth_batches(K, BATCH, bench.N_USERS, bench.N_ITEMS, bench.SEED,
                                                     os.environ.get("PMC_ITEM_DIST", "uniform")))
U, I = bench.xavier_tables(bench.N_USERS, bench.N_ITEMS, bench.DIM, bench.SEED, dev)
eng = FocfEngine(U, I, bench.OBJECTIVE, bench.FAIR_WEIGHT, 5.0)
FusedLazyAdam(eng, lr=bench.LR, weight_decay=bench.WD, sweep_period=int(SWEEP) if SWEEP is not None else None)
eng.defer_loss = True
rows = [(u[k], i[k], s[k], r[k]) for k in range(K)]
for k in range(K):
    eng.forward(u[k], i[k], r[k], s[k], next_batch=rows[k + 1:k + 11] or None)
    eng.backward_adam()
eng.finish()
torch.cuda.synchronize()
print("done", K, "steps; sweep", eng._sweep(BATCH), "loss", float(eng.loss_ring[eng.loss_slot][0]))
