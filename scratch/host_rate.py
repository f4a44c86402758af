"""How fast the host enqueues FOCF steps (eager launches): enqueue time vs completion time per step."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import bench
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam
dev = torch.device("cuda")
K = 400
u, i, r, s = (t.to(dev) for t in bench.synth_batches(K + 30, bench.BATCH, bench.N_USERS, bench.N_ITEMS, bench.SEED))
U, I = bench.xavier_tables(bench.N_USERS, bench.N_ITEMS, bench.DIM, bench.SEED, dev)
eng = FocfEngine(U, I, bench.OBJECTIVE, bench.FAIR_WEIGHT, 5.0)
FusedLazyAdam(eng, lr=bench.LR, weight_decay=bench.WD)
eng.defer_loss = True
rows = [(u[k], i[k], s[k], r[k]) for k in range(K + 30)]
for k in range(150):
    eng.forward(u[k], i[k], r[k], s[k], next_batch=rows[k + 1:k + 21]); eng.backward_adam()
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(150, 350):
    eng.forward(u[k], i[k], r[k], s[k], next_batch=rows[k + 1:k + 21]); eng.backward_adam()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f us/step, complete %.2f us/step" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
