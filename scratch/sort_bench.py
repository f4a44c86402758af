import sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
from fairrec import _C
dev = "cuda"
def run(M, n_rows, reps=50):
    idx = torch.randint(0, n_rows, (M,), dtype=torch.int64, device=dev)
    perm = torch.empty(M + 1, dtype=torch.int32, device=dev); ss = torch.empty_like(perm); sr = torch.empty_like(perm); so = torch.empty_like(perm)
    ns = torch.zeros(1, dtype=torch.int32, device=dev); err = torch.zeros(1, dtype=torch.int32, device=dev)
    st = _C.current_stream()
    f = lambda: _C.lib().fr_sort_segments(idx.data_ptr(), M, n_rows, perm.data_ptr(), ss.data_ptr(), sr.data_ptr(), so.data_ptr(), ns.data_ptr(), err.data_ptr(), st)
    for _ in range(5): f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    print(f"M={M:6d} n_rows={n_rows:10d}  {a.elapsed_time(b) / reps * 1e3:8.1f} us/launch")
for M in (1024, 2048, 4096, 8192, 16384):
    for n in (200, 60000, 1_000_001, 100_000_001):
        run(M, n)
