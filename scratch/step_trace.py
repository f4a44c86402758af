"""Timeline of the waves of ONE fr_focf_step launch (library built with -DFR_STEP_TRACE=1: `make VARIANT=trace
EXTRA=-DFR_STEP_TRACE=1`, run with FAIRREC_HIP_LIB=scratch/lib/libfairrec_hip_trace.so).  Prints a summary and, with
TRACE_OUT=<file.npz>, saves the raw per-wave records (start, end, role, hw id, phase stamps) for offline analysis."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import bench
from fairrec import _C
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam
dev = torch.device("cuda")
K = int(os.environ.get("TRACE_STEPS", "200"))
dist = os.environ.get("TRACE_ITEM_DIST", "uniform")
PIPE = int(os.environ.get("TRACE_PIPE", "24"))      # batches announced past the last step: the traced launch is a steady-state one
u, i, r, s = (t.to(dev) for t in bench.synth_batches(K + PIPE, bench.BATCH, bench.N_USERS, bench.N_ITEMS, bench.SEED, dist))
U, I = bench.xavier_tables(bench.N_USERS, bench.N_ITEMS, bench.DIM, bench.SEED, dev)
eng = FocfEngine(U, I, bench.OBJECTIVE, bench.FAIR_WEIGHT, 5.0)
FusedLazyAdam(eng, lr=bench.LR, weight_decay=bench.WD, sweep_period=int(os.environ.get("SWEEP", "0")) or None)
eng.defer_loss = True
eng.item_runs = dist == "grouped"
rows = [(u[k], i[k], s[k], r[k]) for k in range(K + PIPE)]
for k in range(K):
    eng.forward(u[k], i[k], r[k], s[k], next_batch=rows[k + 1:k + 21] or None)
    eng.backward_adam()
torch.cuda.synchronize()
lib = ctypes.CDLL(_C.LIB_PATH)
n = 16384
both = np.zeros((2 * n, 4), dtype=np.uint64)
rc = lib.fr_debug_step_trace(both.ctypes.data_as(ctypes.c_void_p), n)
assert rc == 0
buf, phs = both[:n], both[n:]
keep = buf[:, 1] > 0
buf, phs = buf[keep], phs[keep]
if os.environ.get("TRACE_OUT"):
    np.savez_compressed(os.environ["TRACE_OUT"], buf=buf, phs=phs, wq=np.nonzero(keep)[0])
t0 = buf[:, 0].min()
st = (buf[:, 0] - t0).astype(np.float64) / 100.0   # us
en = (buf[:, 1] - t0).astype(np.float64) / 100.0
role = buf[:, 2]
print("waves", len(buf), "kernel span %.2f us" % en.max())
for rl, name in ((1, "sweeper"), (2, "interaction")):
    m = role == rl
    if not m.any():
        continue
    d = en[m] - st[m]
    print(f"{name:12s} n={m.sum():6d} start: min {st[m].min():6.2f} median {np.median(st[m]):6.2f} p90 {np.percentile(st[m], 90):6.2f} max {st[m].max():6.2f} | "
          f"dur: min {d.min():6.2f} median {np.median(d):6.2f} p90 {np.percentile(d, 90):6.2f} max {d.max():6.2f} | end: median {np.median(en[m]):6.2f} p90 {np.percentile(en[m], 90):6.2f} max {en[m].max():6.2f}")
# resident waves over time
ts = np.linspace(0, en.max(), 40)
occ = [(int(((st <= t) & (en > t) & (role == 1)).sum()), int(((st <= t) & (en > t) & (role == 2)).sum())) for t in ts]
print("t(us): resident sweeper / interaction waves")
print("  ".join(f"{t:.1f}:{a}/{b}" for t, (a, b) in zip(ts, occ)))
# per-SIMD: when its last wave ends, how many waves it ran (the hardware id: simd 5:4, cu 11:8, sh 12, se 15:13; xcc in the high word)
hw = buf[:, 3]
simd = ((hw >> 4) & 3) | (((hw >> 8) & 0xf) << 2) | (((hw >> 12) & 1) << 6) | (((hw >> 13) & 7) << 7) | (((hw >> 32) & 0xf) << 10)
ids, inv = np.unique(simd, return_inverse=True)
last_end = np.zeros(len(ids)); nw = np.zeros(len(ids), dtype=int); busy = np.zeros(len(ids))
np.maximum.at(last_end, inv, en); np.add.at(nw, inv, 1); np.add.at(busy, inv, en - st)
print(f"SIMDs seen {len(ids)}; waves per SIMD min {nw.min()} median {np.median(nw):.0f} max {nw.max()}; last wave of a SIMD ends: "
      f"p10 {np.percentile(last_end, 10):.1f} median {np.median(last_end):.1f} p90 {np.percentile(last_end, 90):.1f} max {last_end.max():.1f} us")

cyc = (phs[:, 0] & np.uint64((1 << 40) - 1)).astype(np.float64)
dur = (buf[:, 1] - buf[:, 0]).astype(np.float64) * 10.0          # ns
ok = dur > 2000
print("shader clock over a wave's life (s_memtime / s_memrealtime): median %.3f GHz  p10 %.3f  p90 %.3f" %
      tuple(np.percentile(cyc[ok] / dur[ok], [50, 10, 90])))
m = (role == 2) & (phs[:, 1] > 0)
l0 = (phs[m, 0] >> np.uint64(40)).astype(np.float64) / 100.0          # start -> records
l1 = (phs[m, 1] - buf[m, 0]).astype(np.float64) / 100.0 - l0        # records -> rows
rp = (phs[m, 2] - phs[m, 1]).astype(np.float64) / 100.0
fin = (buf[m, 1] - phs[m, 2]).astype(np.float64) / 100.0
steps = phs[m, 3].astype(np.float64)
print("interaction phases (us): level-1 %.2f/%.2f  level-2 %.2f/%.2f  replay %.2f/%.2f  finish %.2f/%.2f  (median/p90); row-steps median %.0f mean %.0f max %.0f"
      % (np.median(l0), np.percentile(l0, 90), np.median(l1), np.percentile(l1, 90), np.median(rp), np.percentile(rp, 90),
         np.median(fin), np.percentile(fin, 90), np.median(steps), steps.mean(), steps.max()))
ok = steps > 20
print("replay us per row-step: median %.4f  p10 %.4f p90 %.4f" % (np.median(rp[ok] / steps[ok]), np.percentile(rp[ok] / steps[ok], 10), np.percentile(rp[ok] / steps[ok], 90)))
raise SystemExit
m = (role == 2) & (phs[:, 0] > 0)
l1 = (phs[m, 0] - buf[m, 0]).astype(np.float64) / 100.0
l2 = (phs[m, 1] - phs[m, 0]).astype(np.float64) / 100.0
rp = (phs[m, 2] - phs[m, 1]).astype(np.float64) / 100.0
fin = (buf[m, 1] - phs[m, 2]).astype(np.float64) / 100.0
steps = phs[m, 3].astype(np.float64)
print("interaction phases (us): level-1 %.2f/%.2f  level-2 %.2f/%.2f  replay %.2f/%.2f  finish %.2f/%.2f  (median/p90); row-steps median %.0f mean %.0f max %.0f"
      % (np.median(l1), np.percentile(l1, 90), np.median(l2), np.percentile(l2, 90), np.median(rp), np.percentile(rp, 90),
         np.median(fin), np.percentile(fin, 90), np.median(steps), steps.mean(), steps.max()))
ok = steps > 20
print("replay us per row-step: median %.4f  p10 %.4f p90 %.4f" % (np.median(rp[ok] / steps[ok]), np.percentile(rp[ok] / steps[ok], 10), np.percentile(rp[ok] / steps[ok], 90)))
