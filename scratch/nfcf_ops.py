"""Which python line issues which aten op in one eager NFCF finetune step (diagnostic)."""
import os, sys, types, traceback
from collections import Counter
import torch
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import bench

rows = Counter()
on = [False]
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        if on[0]:
            fr = [f for f in traceback.extract_stack() if "fairrec" in f.filename or f.filename.endswith("bench.py")]
            where = f"{fr[-1].filename.split('recbole-fairrec_amd/')[-1]}:{fr[-1].lineno}" if fr else "?"
            rows[(str(func), where)] += 1
        return func(*args, **(kwargs or {}))

args = types.SimpleNamespace(nfcf_users=100001, nfcf_items=10001, steps=1, warmup=3, no_graph=True)
dev = torch.device("cuda:0")
import time
orig = time.perf_counter
def pc():
    on[0] = not on[0]      # bench_nfcf calls perf_counter exactly around the timed steps
    return orig()
bench.time.perf_counter = pc
with Log():
    bench.bench_nfcf(args, 0, 1, dev)
for (n, w), c in sorted(rows.items(), key=lambda t: t[0][1]):
    print(f"{c:4d} {n:40s} {w}")
