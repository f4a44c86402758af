import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam
z = np.load(os.path.join(ROOT, "tests/golden/focf_value_long.npz"))
lr, wd, fw = (float(x) for x in z["hyper"][:3])
for sweep in (0,):
    eng = FocfEngine(torch.tensor(z["U0"], device="cuda"), torch.tensor(z["I0"], device="cuda"), "value", fw, 5.0)
    FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=sweep)
    for t in range(60):
        cols = [torch.tensor(z[k][t], device="cuda") for k in ("user_id", "item_id", "rating", "sst")]
        eng.forward(*cols); eng.backward_adam()
    eng.flush()
    for tag, tab in (("U", eng.U), ("I", eng.I)):
        a = tab.weight.cpu().numpy(); b = z[f"{tag}_after60"]
        d = np.abs(a - b); bad = np.argwhere(d > 1e-4 * np.abs(b) + 1e-6)
        print(tag, "max abs", d.max(), "bad", bad.tolist())
        for r, c in bad:
            key = "user_id" if tag == "U" else "item_id"
            print("  elem", r, c, "got", a[r, c], "ref", b[r, c], "row touched at", [t + 1 for t in range(60) if r in z[key][t]],
                  "p0", z[f"{tag}0"][r, c])
            print("  row got", a[r], "\n  row ref", b[r])
            print("  m got", tab.m.cpu().numpy()[r], "\n  m ref", z[f"m{tag}_after60"][r])
            print("  v got", tab.v.cpu().numpy()[r], "\n  v ref", z[f"v{tag}_after60"][r])
