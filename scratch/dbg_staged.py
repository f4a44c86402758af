"""Staged vs sorted prepare, step by step: first step at which the flushed tables differ by more than ulps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import torch
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam

def run(scn, objective="none", dim=64, hot=True, T=16):
    n_users, n_items, B = 5001, 2501, 1500
    g = torch.Generator().manual_seed(17)
    u = torch.randint(1, n_users, (T + 4, B), generator=g)
    i = torch.randint(1, n_items, (T + 4, B), generator=g)
    if hot >= 1:
        i[:, 100:250] = torch.randint(1, 4, (T + 4, 150), generator=g)
    if hot >= 2:
        i[:, 300:520] = 7
    if hot >= 3:
        u[:, 600:700] = 11
    if hot >= 4:
        u[:, 640:660] = u[:, 300:320]
    r = torch.randint(1, 6, (T + 4, B), generator=g).float()
    gender = torch.randint(0, 2, (n_users,), generator=g).float()
    u, i, r = u.cuda(), i.cuda(), r.cuda()
    s = gender.cuda()[u]
    U0 = (torch.randn(n_users, dim, generator=g) * 0.1).cuda()
    I0 = (torch.randn(n_items, dim, generator=g) * 0.1).cuda()
    engs = []
    for staged in (False, True):
        eng = FocfEngine(U0.clone(), I0.clone(), objective, 0.5, 5.0)
        FusedLazyAdam(eng, lr=1e-2, weight_decay=1e-3, sweep_period=5)
        eng.defer_loss = True
        eng.staged = staged
        engs.append(eng)
    for t in range(T):
        if scn == "queue":
            nxt = [(u[j], i[j], s[j], r[j]) for j in range(t + 1, min(t + 4, T))]
        elif scn == "none":
            nxt = None
        elif scn == "cut":
            nxt = [(u[j], i[j], s[j], r[j]) for j in range(t + 1, min(t + 4, T))] if t != 6 else \
                [(u[T + 1], i[T + 1], s[T + 1], r[T + 1]), (u[T + 2], i[T + 2], s[T + 2], r[T + 2])]
        for k, eng in enumerate(engs):
            eng.forward(u[t], i[t], r[t], s[t], next_batch=nxt or None)
            eng.backward_adam()
        if os.environ.get("EVERY", "1") == "1":
            for eng in engs:
                eng.flush()
            a, b = engs
            du = (a.U.weight - b.U.weight).abs().max().item()
            di = (a.I.weight - b.I.weight).abs().max().item()
            if du > 1e-5 or di > 1e-5:
                bad_u = ((a.U.weight - b.U.weight).abs().amax(1) > 1e-5).nonzero().flatten().tolist()
                bad_i = ((a.I.weight - b.I.weight).abs().amax(1) > 1e-5).nonzero().flatten().tolist()
                cu = {x: int((u[t] == x).sum()) for x in bad_u[:8]}
                ci = {x: int((i[t] == x).sum()) for x in bad_i[:8]}
                print(f"  scn={scn} obj={objective} dim={dim} hot={hot}: step {t} du={du:.2e} di={di:.2e} bad users {len(bad_u)} {cu} bad items {len(bad_i)} {ci}")
                return
    for eng in engs:
        eng.flush(); eng.check_device_errors()
    a, b = engs
    print(f"  scn={scn} obj={objective} dim={dim} hot={hot}: ok  equal={torch.equal(a.U.weight, b.U.weight) and torch.equal(a.I.weight, b.I.weight)} "
          f"max du {(a.U.weight - b.U.weight).abs().max().item():.2e} di {(a.I.weight - b.I.weight).abs().max().item():.2e}")

for scn in ("none", "queue", "cut"):
    for hot in (0, 1, 2, 3, 4):
        run(scn, hot=hot)
run("queue", objective="value", hot=4)
run("queue", objective="value", hot=4, dim=128)
os.environ["EVERY"] = "0"
run("queue", hot=0); run("queue", hot=4); run("none", hot=4)
