"""Isolated duration of the look-ahead prepare (sort + LPT) of 8 batches: nothing else on the GPU."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "recbole-fairrec_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
from fairrec import _C
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam
dev = torch.device("cuda", 0)
u, i, r, s = (t.to(dev) for t in bench.synth_batches(64, bench.BATCH, bench.N_USERS, bench.N_ITEMS, 1))
U, I = bench.xavier_tables(bench.N_USERS, bench.N_ITEMS, bench.DIM, 1, dev)
eng = FocfEngine(U, I, bench.OBJECTIVE, bench.FAIR_WEIGHT, 5.0)
FusedLazyAdam(eng, lr=bench.LR, weight_decay=bench.WD)
eng.defer_loss = True
_C.prof_reset(); _C.prof_enable(True)
for rep in range(6):
    eng._prep.clear()
    bt = [(u[k], i[k], s[k], r[k]) for k in range(rep * 8, rep * 8 + 8)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.prepare_many(bt)
    torch.cuda.synchronize()
_C.prof_enable(False)
print(_C.prof_read())
import ctypes
L = _C.lib()
if hasattr(L, "fr_debug_lpt_stamps"):
    st = (ctypes.c_ulonglong * 8)()
    L.fr_debug_lpt_stamps(st)
    s = list(st)
    print("lpt ticks: rec+last", s[1] - s[0], "classify", s[2] - s[1], "scan", s[3] - s[2], "rewrite", s[4] - s[3], "total", s[4] - s[0])
