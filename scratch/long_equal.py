"""Staged vs sorted prepare over many steps at the bench sizes: the tables must agree to rounding (every row, after a flush)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import torch
import bench
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam
dev = torch.device("cuda")
T = int(os.environ.get("T", "400"))
for dist in ("uniform", "zipf"):
    u, i, r, s = (t.to(dev) for t in bench.synth_batches(T, bench.BATCH, bench.N_USERS, bench.N_ITEMS, 7, dist))
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
        eng.finish(); eng.flush(); eng.check_device_errors()
    a, b = engs
    out = []
    for name in ("weight", "m", "v"):
        for tab in ("U", "I"):
            x, y = getattr(getattr(a, tab), name), getattr(getattr(b, tab), name)
            d = (x - y).abs()
            scale = x.abs().max().item()
            out.append(f"{tab}.{name}: max|d| {d.max().item():.3e} (scale {scale:.3e}) rows>1e-5*scale: {int((d.amax(1) > 1e-5 * scale).sum())}")
    print(dist, "loss_acc", a.loss_acc[:3].tolist(), b.loss_acc[:3].tolist())
    print("   " + "\n   ".join(out))
