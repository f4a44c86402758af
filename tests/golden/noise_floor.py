#!/usr/bin/env python3
"""How much of a PFCN / FairGo golden tensor is the reference's own fp32 noise?  Runs the CPU restatement of the reference
step loop (oracle/, pinned to the goldens) with 1 thread and with 8 threads, i.e. the same torch ops with a different
reduction order, and once more in float64 (every fp32 array of the case widened: the same algorithm in near-exact
arithmetic) -- and prints, per kind of parameter tensor, the largest difference between the 1- and 8-thread runs and
between the fp32 run and the float64 run (how far the reference's own rounding has carried it from the exact result of
its algorithm: an implementation that rounds differently -- e.g. more accurately -- lands that far from the golden),
next to the tensor's scale.  The parity tests' per-tensor absolute floors (tests/test_pfcn_hip.py) are taken from this table: a tolerance below
it would reject the reference against itself.  Usage: python tests/golden/noise_floor.py [pfcn_bmf_sm_d128 ...]"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pfcn as O  # noqa: E402

names = sys.argv[1:] or ["pfcn_bmf_sm_d128", "pfcn_bmf_cm_d128", "pfcn_bmf_sm_d64"]
for name in names:
    z = np.load(os.path.join(HERE, name + ".npz"))
    outs = []
    for nt in (1, 8):
        torch.set_num_threads(nt)
        outs.append(O.train(z))
    z64 = {k: (z[k].astype(np.float64) if z[k].dtype == np.float32 else z[k]) for k in z.files}
    exact = O.train(z64)
    print(name)
    worst = {}
    worst64 = {}
    count64 = {}
    for k in outs[0]:
        if not k.startswith("final."):
            continue
        parts = k.split(".")
        kind = "other"
        if parts[-1] == "bias" and parts[-2].isdigit():
            kind = "Linear bias feeding BatchNorm" if int(parts[-2]) % 4 == 1 else "BatchNorm beta"
        elif parts[-1] in ("running_mean", "running_var"):
            kind = "BatchNorm " + parts[-1]
        elif parts[-1] == "weight" and parts[-2].isdigit():
            kind = "Linear weight" if int(parts[-2]) % 4 == 1 else "BatchNorm gamma"
        d = float(np.abs(outs[0][k] - outs[1][k]).max())
        s = float(np.abs(z[k]).max())
        if d > worst.get(kind, (0, 0, ""))[0]:
            worst[kind] = (d, s, k)
        e = np.abs(outs[1][k].astype(np.float64) - exact[k])
        if float(e.max()) > worst64.get(kind, (0, 0, ""))[0]:
            worst64[kind] = (float(e.max()), s, k)
        if kind == "Linear weight":      # elements beyond the parity test's bound 1e-4 |ref| + 2e-5
            over = e > 1e-4 * np.abs(exact[k]) + 2e-5
            count64[k] = (int(over.sum()), over.size, float(e.max()))
    for kind, (d, s, k) in sorted(worst.items()):
        print(f"   {kind:32s} max |1 thread - 8 threads| = {d:.3e}   (scale {s:.3e})   {k}")
    for kind, (d, s, k) in sorted(worst64.items()):
        print(f"   {kind:32s} max |fp32 - float64|       = {d:.3e}   (scale {s:.3e})   {k}")
    for k, (n, size, mx) in sorted(count64.items()):
        if n:
            print(f"   {k}: {n} of {size} elements of the fp32 run are beyond 1e-4 |x| + 2e-5 of the float64 run (max {mx:.2e})")
