"""Wave time lines of consecutive CHAINED step launches (library built with -DFR_STEP_TRACE=1:
scratch/build_step_variant.sh trace -DFR_STEP_TRACE=1; FAIRREC_HIP_LIB=scratch/lib/libfairrec_hip_trace.so).
usage: python scratch/chain_trace.py [steps-in-the-call] [first traced launch]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import bench
from fairrec import _C
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
T0 = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda")
if os.environ.get("PROBE_STREAM") == "1":      # a created stream instead of the legacy default stream
    torch.cuda.set_stream(torch.cuda.Stream())
U, I = bench.xavier_tables(bench.N_USERS, bench.N_ITEMS, bench.DIM, 3, dev)
eng = FocfEngine(U, I, bench.OBJECTIVE, bench.FAIR_WEIGHT, 5.0)
FusedLazyAdam(eng, lr=bench.LR, weight_decay=bench.WD)
eng.defer_loss = True
n_age = eng._sweep(bench.BATCH)
B = bench.BATCH
cols = [t.to(dev).reshape(-1) for t in bench.synth_batches(n_age + K, B, bench.N_USERS, bench.N_ITEMS, 11, os.environ.get("TRACE_ITEM_DIST", "uniform"))]
cut = lambda a, b: [c[a * B:b * B] for c in cols]
lib = ctypes.CDLL(_C.LIB_PATH)
eng.steps_many(*cut(0, n_age), B)
torch.cuda.synchronize()
assert lib.fr_debug_set_trace_step(eng.U.step + 1 + T0) == 0
eng.steps_many(*cut(n_age, n_age + K), B)
torch.cuda.synchronize()
eng.finish(); eng.check_device_errors()
n = 65536
both = np.zeros((2 * n, 4), dtype=np.uint64)
assert lib.fr_debug_step_trace(both.ctypes.data_as(ctypes.c_void_p), n) == 0
buf, phs = both[:n], both[n:]
t00 = None
names = {0: "workgroup 0", 1: "sweeper", 2: "interaction", 3: "stage"}
for L in range(4):
    b = buf[16384 * L:16384 * (L + 1)]
    b = b[b[:, 1] > 0]
    if not len(b):
        continue
    if t00 is None:
        t00 = b[:, 0].min()
    st = (b[:, 0] - t00).astype(np.float64) / 100.0
    en = (b[:, 1] - t00).astype(np.float64) / 100.0
    print(f"launch +{L}: waves {len(b)}  first start {st.min():7.2f}  last end {en.max():7.2f}  span {en.max() - st.min():6.2f} us")
    for rl in (0, 3, 1, 2):
        m = b[:, 2] == rl
        if not m.any():
            continue
        d = en[m] - st[m]
        print(f"   {names[rl]:12s} n={m.sum():5d} start min/med/p90/max {st[m].min():7.2f} {np.median(st[m]):7.2f} {np.percentile(st[m], 90):7.2f} {st[m].max():7.2f} | "
              f"dur med/p90/max {np.median(d):6.2f} {np.percentile(d, 90):6.2f} {d.max():6.2f} | end med/p90/max {np.median(en[m]):7.2f} {np.percentile(en[m], 90):7.2f} {en[m].max():7.2f}")
if os.environ.get("TRACE_OUT"):
    np.savez_compressed(os.environ["TRACE_OUT"], buf=buf, phs=phs)
# per-XCD end of each traced launch (hardware id word: xcc in the high half)
for L in range(4):
    b = buf[16384 * L:16384 * (L + 1)]
    b = b[b[:, 1] > 0]
    if not len(b):
        continue
    xcc = (b[:, 3] >> np.uint64(32)).astype(np.int64) & 0xf
    en = (b[:, 1] - t00).astype(np.float64) / 100.0
    st = (b[:, 0] - t00).astype(np.float64) / 100.0
    ends = [en[xcc == x].max() for x in range(8) if (xcc == x).any()]
    starts = [st[xcc == x].min() for x in range(8) if (xcc == x).any()]
    print(f"launch +{L}: per-XCD first start " + " ".join(f"{s:6.2f}" for s in starts) + " | last end " + " ".join(f"{e:6.2f}" for e in ends) +
          f" | mean of the XCDs' ends {np.mean(ends):.2f}, last {max(ends):.2f}")
