"""Times fr_scorer_fwd / fr_scorer_bwd alone (hip events around 200 launches each).  usage: scorer_bench.py [B ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import torch
from fairrec import _C
lib = _C.lib()
dev = torch.device("cuda")
D, n1, n2 = int(os.environ.get("SC_D", 256)), 128, 64
Bs = [int(x) for x in sys.argv[1:]] or [8192]
for B in Bs:
    for p in (0.0, 0.2):
        g = torch.Generator(device="cuda").manual_seed(1)
        r = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.1
        x0, x1 = r(B, D), r(B, D)
        W1, b1, W2, b2, W3, b3 = r(n1, 2 * D), r(n1), r(n2, n1), r(n2), r(1, n2), r(1)
        o1 = B * D; o2 = 2 * o1
        d = _C.FrScorer(D, D, n1, n2, W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3.data_ptr(), b3.data_ptr(), p, 77, 0, o1, o2, o2 + B * n1)
        state = torch.zeros(2, dtype=torch.int64, device=dev); used = torch.zeros(1, dtype=torch.int64, device=dev)
        x0d, x1d = torch.empty_like(x0), torch.empty_like(x1)
        h1, h2, y = torch.empty(B, n1, device=dev), torch.empty(B, n2, device=dev), torch.empty(B, device=dev)
        label = (torch.rand(B, device=dev) < 0.5).float(); sst = (torch.rand(B, device=dev) < 0.5).float()
        out, dy = torch.empty(B, device=dev), torch.empty(B, device=dev)
        nblk = lib.fr_scorer_blocks(B); part = torch.empty(3 * nblk, device=dev)
        dz1, dz2, dz3 = torch.empty(B, n1, device=dev), torch.empty(B, n2, device=dev), torch.empty(B, device=dev)
        dx0, dx1 = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
        w3p = torch.empty(nblk, n2 + 1, device=dev)
        st = _C.current_stream()
        def fwd():
            _C.check(lib.fr_scorer_fwd(ctypes.byref(d), x0.data_ptr(), x1.data_ptr(), B, state.data_ptr(), used.data_ptr(), state.data_ptr(),
                                       x0d.data_ptr(), x1d.data_ptr(), h1.data_ptr(), h2.data_ptr(), y.data_ptr(), label.data_ptr(), sst.data_ptr(),
                                       out.data_ptr(), dy.data_ptr(), part.data_ptr(), part[nblk:].data_ptr(), None, st), "fwd")
        def bwd(both):
            _C.check(lib.fr_scorer_bwd(ctypes.byref(d), dy.data_ptr(), None, y.data_ptr(), h1.data_ptr(), h2.data_ptr(), B, used.data_ptr(),
                                       dz1.data_ptr(), dz2.data_ptr(), dz3.data_ptr(), dx0.data_ptr() if both else None, dx1.data_ptr(), w3p.data_ptr(), st), "bwd")
        def timeit(fn, n=200):
            for _ in range(10): fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / n * 1e3
        def lat(fn, n=50):
            ts = []
            for _ in range(n):
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            ts.sort()
            return ts[len(ts) // 2]
        print(f"   isolated launches (median): fwd {lat(fwd):.2f}  bwd(item half) {lat(lambda: bwd(False)):.2f}  bwd(both) {lat(lambda: bwd(True)):.2f}")
        print(f"B={B} D={D} p={p}: fwd {timeit(fwd):.2f} us  bwd(item half) {timeit(lambda: bwd(False)):.2f} us  bwd(both) {timeit(lambda: bwd(True)):.2f} us", flush=True)
