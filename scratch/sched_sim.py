"""Toy model of one focf_step_kernel launch: 256 CUs x 4 SIMDs, workgroups of 4 waves (one per SIMD) handed out in index
order to CUs with a free slot, every SIMD shared by its resident waves that have VALU work (processor sharing, a single
chain can use at most FMAX of a SIMD).  Used to rank launch orders before building them (diagnostics only)."""
import numpy as np, sys

NS_PER_ROWSTEP = 34 / 2.19      # ns of SIMD time per paired row-step (34 cycles at the measured 2.19 GHz)
SINGLE = 41 / 34.0              # an unpaired row-step costs this much more
FMAX = 0.62                     # share of a SIMD one dependent chain can use
DT = 0.05                       # us

def ages(rng, n, p, S=123):
    a = rng.geometric(p, n)
    return np.minimum(a, S)

def pair_cost(a, b):            # us of SIMD time: common part paired, the rest alone
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    return (2 * lo + (hi - lo) * SINGLE) * NS_PER_ROWSTEP * 1e-3

def make_tasks(rng, sort_sweep=False, repair=False, inter_overhead=1.7e3 / 2.19 * 1e-3):
    B = 8192
    # interactions: 2 per wave; user ages geometric p=0.0082 capped, item ages p=0.079; LPT order in 8 classes
    # a batch row was last brought up to date by an earlier batch or by the sweeper, whichever came later
    au = np.minimum(ages(rng, B, 0.0082), rng.integers(0, 123, B))
    ai = np.minimum(ages(rng, B, 0.079), rng.integers(0, 123, B))
    est = 7 * np.maximum(au, ai) + 2 * np.minimum(au, ai)
    cls = 7 - np.minimum(7, est * 8 // (9 * 123 + 1))
    order = np.argsort(cls, kind="stable")
    au, ai = au[order], ai[order]
    w_inter = pair_cost(au[0::2], au[1::2]) + pair_cost(ai[0::2], ai[1::2]) + inter_overhead
    # sweeper: N/S rows of each table, adjacent pairs
    nu, ni = 8130, 814
    su, si = ages(rng, nu, 0.0082), ages(rng, ni, 0.079)
    if repair:                   # pair rows of similar age
        su, si = np.sort(su)[::-1], np.sort(si)[::-1]
    w_sw = np.concatenate([pair_cost(su[0::2], su[1::2]), pair_cost(si[0::2], si[1::2])])
    if sort_sweep:
        w_sw = np.sort(w_sw)[::-1]
    return w_inter, w_sw

def simulate(w_inter, w_sw, lead_first=True, wg=4, slots=6, front=(1.8, 1.2), back=(3.1, 0.3), verbose=False):
    # task arrays in dispatch order
    n_i, n_s = len(w_inter), len(w_sw)
    pad = lambda x: np.concatenate([x, np.zeros((-len(x)) % wg)])
    wi, ws = pad(w_inter).reshape(-1, wg), pad(w_sw).reshape(-1, wg)
    kinds = np.concatenate([np.zeros(len(wi), int), np.ones(len(ws), int)]) if lead_first else np.concatenate([np.ones(len(ws), int), np.zeros(len(wi), int)])
    work = np.concatenate([wi, ws]) if lead_first else np.concatenate([ws, wi])
    n_wg = len(work)
    NCU = 256
    # per resident WG state: per-CU list of (remaining front latency, remaining work, remaining back latency) per wave
    cu_slots = [[] for _ in range(NCU)]
    nxt, t, done_wg, rr = 0, 0.0, 0, 0
    ends = np.zeros((NCU, 4))
    while done_wg < n_wg:
        # dispatch: one workgroup per CU per pass, round robin from where the last pass stopped
        progressed = True
        while progressed and nxt < n_wg:
            progressed = False
            for d in range(NCU):
                c = (rr + d) % NCU
                if nxt < n_wg and len(cu_slots[c]) < slots:
                    k = kinds[nxt]
                    cu_slots[c].append([np.full(wg, front[k]), work[nxt].copy(), np.full(wg, back[k]), k])
                    nxt += 1
                    progressed = True
                    last_c = c
            if progressed:
                rr = (last_c + 1) % NCU
        # advance DT
        for c in range(NCU):
            sl = cu_slots[c]
            if not sl:
                continue
            F = np.array([s[0] for s in sl]); W = np.array([s[1] for s in sl]); Bk = np.array([s[2] for s in sl])
            active = (F <= 0) & (W > 0)
            nact = active.sum(0)                                     # per SIMD
            share = np.where(nact > 0, np.minimum(FMAX, 1.0 / np.maximum(nact, 1)), 0.0)
            W2 = np.where(active, W - share[None, :] * DT, W)
            F2 = np.where(F > 0, F - DT, F)
            fin = (F <= 0) & (W <= 0)
            B2 = np.where(fin & (Bk > 0), Bk - DT, Bk)
            keep = []
            for j, s in enumerate(sl):
                s[0], s[1], s[2] = F2[j], W2[j], B2[j]
                wave_done = (s[0] <= 0) & (s[1] <= 0) & (s[2] <= 0)
                newly = wave_done & (ends[c] < t + DT)               # record last end per simd
                if wave_done.all():
                    done_wg += 1
                    ends[c] = np.maximum(ends[c], t + DT)
                else:
                    keep.append(s)
            cu_slots[c] = keep
        t += DT
    return t, ends

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for name, kw, skw in (("current", {}, {}), ("sorted sweeper tasks", dict(sort_sweep=True), {}),
                          ("re-paired by age + sorted", dict(sort_sweep=True, repair=True), {}),
                          ("sweepers first", {}, dict(lead_first=False)),
                          ("half the interaction overhead", dict(inter_overhead=0.85e3 / 2.19 * 1e-3), dict(back=(1.5, 0.3))),
                          ("sorted + repaired + half overhead", dict(sort_sweep=True, repair=True, inter_overhead=0.85e3 / 2.19 * 1e-3), dict(back=(1.5, 0.3)))):
        wi, ws = make_tasks(np.random.default_rng(1), **kw)
        t, ends = simulate(wi, ws, **skw)
        tot = (wi.sum() + ws.sum()) / 1024
        print(f"{name:36s} makespan {t:5.1f} us   mean SIMD work {tot:5.1f} us  (inter {wi.sum() / 1024:4.1f} sweep {ws.sum() / 1024:4.1f})")

def report(name, wi, ws, **skw):
    t, ends = simulate(wi, ws, **skw)
    e = ends.reshape(-1)
    print(f"{name:36s} makespan {t:5.1f}  SIMD ends p10 {np.percentile(e,10):.1f} med {np.median(e):.1f} p90 {np.percentile(e,90):.1f}  work/SIMD {(wi.sum()+ws.sum())/1024:.1f}")
