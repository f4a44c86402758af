"""Calibration + measurement run for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one counter per pass).
1) table_flush_kernel on a 1M x 64 table with every row exactly 1 step stale: reads N*(3*D*4 + 4) bytes
   (p,m,v rows + last) and writes the same -> known byte count in OUR access pattern (dword per lane, 256-B rows).
2) 340 FOCF steps of the bench workload, launched as bench.py launches them (one-launch step, 16 batches prepared ahead)."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import bench
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import AdamHyper, FusedLazyAdam, LazyTable
N, D = 1_000_000, 64
hyper = AdamHyper(lr=1e-3, weight_decay=1e-3, device="cuda")
tab = LazyTable(torch.randn(N, D, device="cuda") * 0.01)
tab.ensure_state(); tab.m.normal_(std=1e-3); tab.v.uniform_(1e-7, 1e-5)
for rep in range(3):
    tab.last.fill_(1); tab.step = 2; tab._dirty = True
    tab.flush(hyper)
torch.cuda.synchronize()
print("flush known bytes per launch: read", N * (3 * D * 4 + 4), "write", N * (3 * D * 4 + 4))
dev = torch.device("cuda")
K = 340
u, i, r, s = (t.to(dev) for t in bench.synth_batches(K, bench.BATCH, bench.N_USERS, bench.N_ITEMS, bench.SEED))
U, I = bench.xavier_tables(bench.N_USERS, bench.N_ITEMS, bench.DIM, bench.SEED, dev)
eng = FocfEngine(U, I, bench.OBJECTIVE, bench.FAIR_WEIGHT, 5.0)
FusedLazyAdam(eng, lr=bench.LR, weight_decay=bench.WD)
eng.defer_loss = True
rows = [(u[k], i[k], s[k], r[k]) for k in range(K)]
for k in range(K):
    eng.forward(u[k], i[k], r[k], s[k], next_batch=rows[k + 1:k + 21] or None); eng.backward_adam()
eng.finish()
torch.cuda.synchronize()
print("done", K, "steps; sweep", eng._sweep(bench.BATCH))
