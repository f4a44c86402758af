"""Wave time lines of the item-run finisher (library built with -DFR_RUN_DIAG=5): per item, wave 0 and the last wave, 8 stamps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recbole-fairrec_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import bench as Bn
from fairrec import _C
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam
dev = torch.device("cuda", 0)
U, I = Bn.xavier_tables(Bn.N_USERS, Bn.N_ITEMS, Bn.DIM, Bn.SEED, dev)
eng = FocfEngine(U, I, "value", 1.0, 5.0)
FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-3, sweep_period=int(os.environ.get("SWEEP", "123")))
eng.defer_loss = True; eng.item_runs = True
n = 160
u, i, r, s = (t.to(dev) for t in Bn.synth_batches(n, Bn.BATCH, Bn.N_USERS, Bn.N_ITEMS, 7, "grouped"))
rows = [(u[j], i[j], s[j], r[j]) for j in range(n)]
for k in range(n):
    eng.forward(u[k], i[k], r[k], s[k], next_batch=rows[k + 1:k + 9] or None)
    eng.backward_adam()
torch.cuda.synchronize()
ws = eng._prev[0]
lay_off = None
# task_rec offset inside the workspace: recompute the layout like focf_layout (256-byte aligned takes)
B, D = Bn.BATCH, Bn.DIM
off = 0
def take(nb):
    global off
    o = off; off = (off + nb + 255) // 256 * 256; return o
Bp = B + 1
take(32 * 4)
for _ in range(4): take(Bp * 4)
take(4)
for _ in range(4): take(Bp * 4)
take(16); take(Bp * 4); take(Bp * 4)
ngb = (B * 64 + 255) // 256; nfb = (B * 16 + 1023) // 1024
take(ngb * 4); take(nfb * 4 * 4); take(4); take(((2 * B + 3) // 4) * 4)
take(Bp * 16); take(Bp * 16)
o_task = take(Bp * 16)
raw = ws[o_task:o_task + 82 * 2 * 8 * 8].view(torch.int64).cpu().numpy().reshape(82, 2, 8).astype(np.float64) / 100.0   # us
t0 = raw[:, :, 0].min()
rel = raw - t0
names = ["start", "lvl3 issued", "stats done", "barrier A", "members done", "barrier B", "grad done", "end"]
for w in (0, 1):
    print("wave", "0" if w == 0 else "last")
    for j, nm in enumerate(names):
        x = rel[:, w, j]
        print(f"  {nm:14s} median {np.median(x):7.2f}  p90 {np.percentile(x, 90):7.2f}  max {x.max():7.2f}")
d = rel[:, 0, 1:] - rel[:, 0, :-1]
print("wave 0 phase durations (median):", [round(float(np.median(d[:, j])), 2) for j in range(7)])
d = rel[:, 1, 1:] - rel[:, 1, :-1]
print("last wave phase durations (median):", [round(float(np.median(d[:, j])), 2) for j in range(7)])
sys.stdout.flush(); os._exit(0)
