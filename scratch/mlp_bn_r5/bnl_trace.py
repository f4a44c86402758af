"""Phase stamps of one bnl_fwd launch (diagnostic library: make -C recbole-fairrec_amd/csrc VARIANT=bnltrace EXTRA=-DBNL_TRACE=1;
run with FAIRREC_HIP_LIB=scratch/lib/libfairrec_hip_bnltrace.so)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "recbole-fairrec_amd"))
from fairrec.model.layers import MLPLayers
from fairrec import _C

os.environ["FAIRREC_BN_FUSED"] = "1"
lib = _C.lib()
raw = ctypes.CDLL(_C.LIB_PATH)
M = int(os.environ.get("M", 8192))
widths = [int(v) for v in os.environ.get("WIDTHS", "128,256").split(",")]
mlp = MLPLayers(widths, activation="leakyrelu", bn=True, init_method="norm").cuda().train()
x = torch.randn(M, widths[0], device="cuda")
with torch.no_grad():
    for _ in range(3):
        mlp(x)
torch.cuda.synchronize()
# the last launch that stamped was bnl_fwd of the top layer (bnl_out does not stamp)
n = (M + 31) // 32
buf = np.zeros(n * 8, dtype=np.uint64)
assert raw.fr_bnl_trace_read(buf.ctypes.data_as(ctypes.c_void_p), n * 8) == 0
t = buf.reshape(n, 8).astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["start", "A formed", "product done", "stored", "arrived", "folded", "ticket drawn"]
for k, nm in enumerate(names):
    col = us[:, k][t[:, k] > 0]
    if col.size:
        print("%-14s n=%4d  min %7.2f  median %7.2f  max %7.2f us" % (nm, col.size, col.min(), np.median(col), col.max()))
