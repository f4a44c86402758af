#!/usr/bin/env python3
"""Companion fixtures <case>_f64.npz for the PFCN goldens: the SAME step sequence (the case's recorded batches, dropout masks
and initial state) run through the CPU restatement of the reference (oracle/pfcn.py, pinned to the goldens by
tests/test_oracle_pfcn.py) with every fp32 array of the case widened to float64, i.e. the reference's algorithm in
near-exact arithmetic.  Why: Adam divides a gradient by its own magnitude, so an element whose gradient nearly cancels
(the Linear weights in front of a BatchNorm) moves by the SIGN PATTERN of rounding noise; the reference's fp32 run sits up
to 9e-5 from its own float64 run on a few dozen such elements (tests/golden/noise_floor.py prints the table).  An
implementation that rounds differently from torch's CPU GEMM -- e.g. more accurately -- lands near the float64 result on
those elements, so the parity test accepts an element that lies between the two executions of the reference (plus the
usual tolerance on either side) instead of pretending the fp32 run is exact.
Data only: parameter tensors of the final state of the float64 run (stored rounded to fp32: 1e-7 relative).  Usage: python tests/golden/gen_pfcn_exact64.py [case ...]"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pfcn as O  # noqa: E402

names = sys.argv[1:] or sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "pfcn_*.npz"))
                               if not p.endswith("_f64.npz"))
for name in names:
    z = np.load(os.path.join(HERE, name + ".npz"))
    z64 = {k: (z[k].astype(np.float64) if z[k].dtype == np.float32 else z[k]) for k in z.files}
    out = O.train(z64)
    keep = {k: np.asarray(v, dtype=np.float32) for k, v in out.items() if k.startswith("final.") or k == "loss"}
    path = os.path.join(HERE, name + "_f64.npz")
    np.savez_compressed(path, **keep)
    worst = max(float(np.abs(out[k] - z[k]).max()) for k in keep if k in z.files and k != "loss")
    print(f"{path}: {len(keep)} arrays, {os.path.getsize(path) / 1024:.1f} KiB, max |float64 - golden| = {worst:.2e}")
