"""Role spans of the last fr_focf_step_runs_pipe launch (library built with -DFR_PIPE_TRACE, FAIRREC_HIP_LIB=...): bench.py's
grouped workload for a few hundred steps, then first start / last end of the loss, sweeper, item-run and gather waves."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
os.environ["FAIRREC_FOCF_PIPE"] = "1"
import numpy as np, torch
import bench
from fairrec import _C
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam
dev = torch.device("cuda")
K = 260
u, i, r, s = (t.to(dev) for t in bench.synth_batches(K + 24, bench.BATCH, bench.N_USERS, bench.N_ITEMS, bench.SEED, "grouped"))
U, I = bench.xavier_tables(bench.N_USERS, bench.N_ITEMS, bench.DIM, bench.SEED, dev)
eng = FocfEngine(U, I, bench.OBJECTIVE, bench.FAIR_WEIGHT, 5.0)
FusedLazyAdam(eng, lr=bench.LR, weight_decay=bench.WD)
eng.defer_loss = True
eng.item_runs = True
rows = [(u[k], i[k], s[k], r[k]) for k in range(K + 24)]
raw = ctypes.CDLL(_C.LIB_PATH)
buf = np.zeros(64, dtype=np.uint64)
raw.fr_debug_pipe_trace(buf.ctypes.data_as(ctypes.c_void_p), 1)
for k in range(K):
    eng.forward(u[k], i[k], r[k], s[k], next_batch=rows[k + 1:k + 21] or None)
    eng.backward_adam()
torch.cuda.synchronize()
raw.fr_debug_pipe_trace(buf.ctypes.data_as(ctypes.c_void_p), 0)
NONE = np.uint64(2**64 - 1)
for slot in range(8):
    b8 = buf[slot * 8:slot * 8 + 8]
    if all(b8[2 * q] == NONE for q in range(4)):
        continue
    t0 = min(int(b8[2 * q]) for q in range(4) if b8[2 * q] != NONE)
    line = f"slot {slot}: "
    for q, name in enumerate(("loss", "sweeper", "item runs", "gather")):
        if b8[2 * q] != NONE:
            line += f"{name} {(int(b8[2 * q]) - t0) / 100:5.2f}..{(int(b8[2 * q + 1]) - t0) / 100:6.2f}  "
    print(line)
sys.stdout.flush()
os._exit(0)
