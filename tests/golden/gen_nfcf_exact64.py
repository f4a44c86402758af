#!/usr/bin/env python3
"""Companion fixtures nfcf_<case>_f64.npz: the NFCF goldens' step sequences run through the pinned CPU restatement
(oracle/nfcf.py) with every fp32 array widened to float64 -- the reference's algorithm in near-exact arithmetic; see
gen_pfcn_exact64.py for why the parity tests accept a parameter element that lies between the reference's fp32 and float64
executions.  Data only (parameter snapshots, stored rounded to fp32).  Usage: python tests/golden/gen_nfcf_exact64.py [case ...]"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import nfcf as O  # noqa: E402

names = sys.argv[1:] or sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "nfcf_*.npz"))
                               if not p.endswith("_f64.npz"))
for name in names:
    z = np.load(os.path.join(HERE, name + ".npz"))
    z64 = {k: (z[k].astype(np.float64) if z[k].dtype == np.float32 else z[k]) for k in z.files}
    out = O.train(z64, snaps=tuple(int(s) for s in z["snaps"]))
    keep = {k: np.asarray(v, dtype=np.float32) for k, v in out.items() if k.startswith("after")}
    path = os.path.join(HERE, name + "_f64.npz")
    np.savez_compressed(path, **keep)
    worst = max(float(np.abs(out[k] - z[k]).max()) for k in keep if k in z.files)
    print(f"{path}: {len(keep)} arrays, {os.path.getsize(path) / 1024:.1f} KiB, max |float64 - golden| = {worst:.2e}")
