#!/usr/bin/env python3
"""How much of a PFCN / FairGo golden tensor is the reference's own fp32 noise?  Runs the CPU restatement of the reference
step loop (oracle/, pinned to the goldens) twice -- 1 thread and 8 threads, i.e. the same torch ops with a different
reduction order -- and prints, per parameter tensor, the largest difference between the two runs next to the tensor's
scale.  The parity tests' per-tensor absolute floors (tests/test_pfcn_hip.py) are taken from this table: a tolerance below
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
    print(name)
    worst = {}
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
    for kind, (d, s, k) in sorted(worst.items()):
        print(f"   {kind:32s} max |1 thread - 8 threads| = {d:.3e}   (scale {s:.3e})   {k}")
