import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import numpy as np, torch
import bench
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam
n_users, n_items, B, T, D = 300_001, 30_001, 8192, int(os.environ.get("T", 40)), 64
u, i, r, s = (t.cuda() for t in bench.synth_batches(T, B, n_users, n_items, 11, "grouped"))
g = torch.Generator().manual_seed(2)
U0 = (torch.randn(n_users, D, generator=g) * 0.05).cuda()
I0 = (torch.randn(n_items, D, generator=g) * 0.05).cuda()
def run(pipe, ahead=True):
    eng = FocfEngine(U0.clone(), I0.clone(), "value", 0.5, 5.0)
    FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-3)
    eng.defer_loss = True; eng.item_runs = True; eng.PIPE = pipe
    rows = [(u[k], i[k], s[k], r[k]) for k in range(T)]
    for k in range(T):
        eng.forward(u[k], i[k], r[k], s[k], next_batch=(rows[k + 1:k + 9] or None) if ahead else None)
        eng.backward_adam()
    eng.flush(); eng.check_device_errors()
    return eng
ref, a, b = run(False), run(True), run(True)
for name, x, y in (("U", a.U.weight, b.U.weight), ("I", a.I.weight, b.I.weight)):
    d = (x - y).abs().amax(1)
    bad = d.nonzero().squeeze(1)
    print(name, "rows differing a vs b:", bad.numel(), "max", float(d.max()))
    if bad.numel():
        ids = (u if name == "U" else i)
        for rr in bad[:8].tolist():
            steps = [k for k in range(T) if bool((ids[k] == rr).any())]
            print("   row", rr, "diff", float(d[rr]), "in batches", steps, " a-ref", float((x[rr] - (ref.U.weight if name == 'U' else ref.I.weight)[rr]).abs().max()),
                  " b-ref", float((y[rr] - (ref.U.weight if name == 'U' else ref.I.weight)[rr]).abs().max()))
for name, x, y in (("U", a.U.weight, ref.U.weight), ("I", a.I.weight, ref.I.weight)):
    d = (x - y).abs().amax(1)
    print(name, "a vs ref: rows > 1e-6:", int((d > 1e-6).sum()), "max", float(d.max()))
# no look-ahead
a2, b2 = run(True, False), run(True, False)
print("no look-ahead: equal", bool(torch.equal(a2.U.weight, b2.U.weight)), bool(torch.equal(a2.I.weight, b2.I.weight)))
