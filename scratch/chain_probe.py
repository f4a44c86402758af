"""fr_focf_steps_many at the BASELINE sizes: host issue time and device time of one call of K steps (FAIRREC_FOCF_CHAIN=0/1).
usage: python scratch/chain_probe.py [K] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "recbole-fairrec_amd"))
import torch
import bench
from fairrec.model.fair_recommender.focf import FocfEngine
from fairrec.optim import FusedLazyAdam

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dist = sys.argv[3] if len(sys.argv) > 3 else "uniform"
dev = torch.device("cuda")
if os.environ.get("PROBE_STREAM") == "1":      # a created stream instead of the legacy default stream
    torch.cuda.set_stream(torch.cuda.Stream())
U, I = bench.xavier_tables(bench.N_USERS, bench.N_ITEMS, bench.DIM, 3, dev)
eng = FocfEngine(U, I, bench.OBJECTIVE, bench.FAIR_WEIGHT, 5.0)
FusedLazyAdam(eng, lr=bench.LR, weight_decay=bench.WD)
eng.defer_loss = True
n_age = eng._sweep(bench.BATCH)
cols = [t.to(dev).reshape(-1) for t in bench.synth_batches(n_age + K * calls, bench.BATCH, bench.N_USERS, bench.N_ITEMS, 11, dist)]
B = bench.BATCH
cut = lambda a, b: [c[a * B:b * B] for c in cols]
eng.steps_many(*cut(0, n_age), B)
torch.cuda.synchronize()
for c in range(calls):
    lo = n_age + c * K
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.steps_many(*cut(lo, lo + K), B)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"chain={os.environ.get('FAIRREC_FOCF_CHAIN', '1')} call {c}: host issue {(t1 - t0) / K * 1e6:.2f} us/step, "
          f"device done {(t2 - t0) / K * 1e6:.2f} us/step", flush=True)
eng.finish()
eng.check_device_errors()
print("loss_acc", eng.loss_acc[:5].tolist())
