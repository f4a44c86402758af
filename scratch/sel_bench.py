"""fr_spmm_csr_sel on the all-rows backward shape of BASELINE configs[3] (synthetic regular graph: N rows x deg nonzeros, random
columns; S selected columns).  usage: sel_bench.py [N] [deg] [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import torch
from fairrec import _C
lib = _C.lib()
dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_002
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 36
S = int(sys.argv[3]) if len(sys.argv) > 3 else 157_893
D = 128
g = torch.Generator(device="cuda").manual_seed(1)
indptr = torch.arange(N + 1, device=dev, dtype=torch.int64) * deg
col = torch.randint(0, N, (N * deg,), device=dev, generator=g, dtype=torch.int32)
val = torch.rand(N * deg, device=dev, generator=g)
rows = torch.randperm(N, device=dev, generator=g)[:S].sort().values
pos = torch.full((N,), -1, dtype=torch.int32, device=dev)
pos[rows] = torch.arange(S, dtype=torch.int32, device=dev)
bits = torch.zeros((N + 31) // 32, dtype=torch.int32, device=dev)
bits.index_add_(0, rows >> 5, (torch.ones_like(rows) << (rows & 31)).to(torch.int32))
dY = torch.randn(S, D, device=dev, generator=g)
dX = torch.empty(N, D, device=dev)
def run(use_bits):
    _C.check(lib.fr_spmm_csr_sel(indptr.data_ptr(), col.data_ptr(), val.data_ptr(), dY.data_ptr(), None, N, pos.data_ptr(),
                                 bits.data_ptr() if use_bits else None, D, dX.data_ptr(), _C.current_stream()), "sel")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print(f"N={N} deg={deg} S={S}: with bitmap {timeit(lambda: run(True)):.2f} ms, map only {timeit(lambda: run(False)):.2f} ms", flush=True)
ref = dX.clone()
run(False)
print("same result:", bool(torch.equal(ref, dX)), " zero fill alone:", f"{timeit(lambda: dX.zero_()):.2f} ms",
      " col pass alone (sum):", f"{timeit(lambda: col.sum()):.2f} ms")
