"""Vectorised version of sched_sim.simulate (same model): state arrays [CU, slot, simd]."""
import numpy as np
from sched_sim import make_tasks, NS_PER_ROWSTEP, pair_cost, ages
import sched_sim as S

def simulate(work, kinds, wg=4, slots=6, front=(1.8, 1.2), back=(3.1, 0.3), fmax=0.62, dt=0.05, ncu=256, lanes=4):
    """work [n_wg, lanes] us of SIMD time per wave; kinds [n_wg] 0 = interaction, 1 = sweeper (latencies)."""
    n_wg = len(work)
    F = np.zeros((ncu, slots, lanes)); W = np.zeros((ncu, slots, lanes)); Bk = np.zeros((ncu, slots, lanes))
    occ = np.zeros((ncu, slots), bool)
    ends = np.zeros((ncu, lanes))
    front, back = np.asarray(front), np.asarray(back)
    nxt, t, done, rr = 0, 0.0, 0, 0
    while done < n_wg:
        free = ~occ
        while nxt < n_wg and free.any():
            # one pass: each CU with a free slot takes one workgroup, in round-robin order from rr
            order = (rr + np.arange(ncu)) % ncu
            has = free.any(1)[order]
            cs = order[has][: n_wg - nxt]
            if len(cs) == 0:
                break
            sl = free[cs].argmax(1)
            k = kinds[nxt:nxt + len(cs)]
            F[cs, sl] = front[k][:, None]; W[cs, sl] = work[nxt:nxt + len(cs)]; Bk[cs, sl] = back[k][:, None]
            occ[cs, sl] = True
            free[cs, sl] = False
            nxt += len(cs)
            rr = (cs[-1] + 1) % ncu
        active = occ[:, :, None] & (F <= 0) & (W > 0)
        nact = active.sum(1)                                     # [cu, simd]
        share = np.where(nact > 0, np.minimum(fmax, 1.0 / np.maximum(nact, 1)), 0.0)
        fin = occ[:, :, None] & (F <= 0) & (W <= 0)
        Bk = np.where(fin & (Bk > 0), Bk - dt, Bk)
        W = np.where(active, W - share[:, None, :] * dt, W)
        F = np.where(occ[:, :, None] & (F > 0), F - dt, F)
        t += dt
        wave_done = (F <= 0) & (W <= 0) & (Bk <= 0)
        wg_done = occ & wave_done.all(2)
        if wg_done.any():
            done += int(wg_done.sum())
            cu_idx = np.nonzero(wg_done.any(1))[0]
            ends[cu_idx] = t
            occ &= ~wg_done
    return t, ends

def groups(w, wg=4):
    pad = (-len(w)) % wg
    return np.concatenate([w, np.zeros(pad)]).reshape(-1, wg)

def run(name, wi, ws, lead_first=True, order=None, **kw):
    gi, gs = groups(wi, kw.get("lanes", 4)), groups(ws, kw.get("lanes", 4))
    if order is not None:
        work, kinds = order(gi, gs)
    elif lead_first:
        work, kinds = np.concatenate([gi, gs]), np.concatenate([np.zeros(len(gi), int), np.ones(len(gs), int)])
    else:
        work, kinds = np.concatenate([gs, gi]), np.concatenate([np.ones(len(gs), int), np.zeros(len(gi), int)])
    t, ends = simulate(work, kinds, **kw)
    e = ends.reshape(-1)
    print(f"{name:44s} makespan {t:5.1f}  CU ends p10 {np.percentile(e,10):.1f} med {np.median(e):.1f} p90 {np.percentile(e,90):.1f}  work/SIMD {(wi.sum()+ws.sum())/1024:.1f}", flush=True)
    return t

if __name__ == "__main__":
    rng = lambda: np.random.default_rng(1)
    wi, ws = make_tasks(rng())
    run("current", wi, ws)
    run("sweep first", wi, ws, lead_first=False)
    wi2, ws2 = make_tasks(rng(), sort_sweep=True)
    run("sorted sweep", wi2, ws2)
    def merged(gi, gs):     # one list, longest workgroup first
        work = np.concatenate([gi, gs]); kinds = np.concatenate([np.zeros(len(gi), int), np.ones(len(gs), int)])
        o = np.argsort(-work.sum(1), kind="stable")
        return work[o], kinds[o]
    run("one LPT list (inter + sorted sweep)", wi2, ws2, order=merged)
    run("sorted sweep, back latency 1.5", wi2, ws2, back=(1.5, 0.3))
    run("sorted sweep, 8 slots", wi2, ws2, slots=8)
    run("sorted sweep, no latencies", wi2, ws2, front=(0, 0), back=(0, 0))
    run("current, no latencies", wi, ws, front=(0, 0), back=(0, 0))
    wi3, ws3 = make_tasks(rng(), sort_sweep=True, inter_overhead=0.85e3 / 2.19 * 1e-3)
    run("sorted sweep, half inter overhead+back1.5", wi3, ws3, back=(1.5, 0.3))
