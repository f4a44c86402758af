"""Phase stamps of sort_segments_kernel (library built with -DFR_SORT_STAMPS, FAIRREC_HIP_LIB=...): M ids below n_rows."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import numpy as np, torch
from fairrec import _C
lib = _C.lib(); raw = ctypes.CDLL(_C.LIB_PATH)
M, n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 100001
dev = torch.device("cuda")
idx = torch.randint(1, n_rows, (M,), device=dev)
perm = torch.empty(M + 1, dtype=torch.int32, device=dev); ss = torch.empty(M + 1, dtype=torch.int32, device=dev)
sr = torch.empty(M + 1, dtype=torch.int32, device=dev); ns = torch.empty(4, dtype=torch.int32, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
for _ in range(5):
    _C.check(lib.fr_sort_segments(idx.data_ptr(), M, n_rows, perm.data_ptr(), ss.data_ptr(), sr.data_ptr(), None, ns.data_ptr(), err.data_ptr(), _C.current_stream()), "sort")
    torch.cuda.synchronize()
buf = np.zeros(16, dtype=np.uint64)
raw.fr_debug_sort_stamps(buf.ctypes.data_as(ctypes.c_void_p))
t = (buf - buf[0]).astype(np.float64) / 100.0
names = ["enter", "keys in LDS", "p0 ranks", "p0 scan", "p0 scatter", "p1 ranks", "p1 scan", "p1 scatter", "p2 ranks", "p2 scan", "p2 scatter", "-", "heads", "seg scan", "written"]
for k in range(15):
    if buf[k]:
        print(f"{names[k]:12s} {t[k]:7.2f} us")
