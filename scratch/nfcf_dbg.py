import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import torch
import bench
from fairrec.model.fair_recommender import nfcf as N
orig = N._NfcfFused.backward
def bw(ctx, g_loss, g_out):
    pass
    return orig(ctx, g_loss, g_out)
N._NfcfFused.backward = staticmethod(bw)
import argparse
sys.argv = ["bench.py", "--workload", "nfcf100m", "--nfcf-users", "1000001", "--nfcf-items", "100001", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph"]
bench.main()
import ctypes, numpy as np
from fairrec import _C
raw = ctypes.CDLL(_C.LIB_PATH)
if hasattr(raw, "fr_debug_scorer_trace"):
    buf = np.zeros(64, dtype=np.uint64)
    raw.fr_debug_scorer_trace(buf.ctypes.data_as(ctypes.c_void_p))
    t = (buf[16:23] - buf[16]).astype(np.float64)
    print("bwd stamps (cycles): enter, counter, dz2, w3part, dz1, dX products, end:", t.tolist())
