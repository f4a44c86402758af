"""Phase stamps of workgroup 0 of scorer_fwd_kernel (library built with -DFR_SC_TRACE=1, FAIRREC_HIP_LIB=scratch/lib/libfairrec_hip_sctrace.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import numpy as np, torch
from fairrec import _C
lib = _C.lib()
raw = ctypes.CDLL(_C.LIB_PATH)
dev = torch.device("cuda")
D, n1, n2, B = 256, 128, 64, int(os.environ.get("SC_B", 8192))
for p in (0.0, 0.2):
    r = lambda *s: torch.randn(*s, device=dev) * 0.1
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
    for it in range(5):
        _C.check(lib.fr_scorer_fwd(ctypes.byref(d), x0.data_ptr(), x1.data_ptr(), B, state.data_ptr(), used.data_ptr(), state.data_ptr(),
                                   x0d.data_ptr(), x1d.data_ptr(), h1.data_ptr(), h2.data_ptr(), y.data_ptr(), label.data_ptr(), sst.data_ptr(),
                                   out.data_ptr(), dy.data_ptr(), part.data_ptr(), part[nblk:].data_ptr(), None, _C.current_stream()), "fwd")
        torch.cuda.synchronize()
    buf = np.zeros(64, dtype=np.uint64)
    assert raw.fr_debug_scorer_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = (buf[:15] - buf[0]).astype(np.float64)
    names = ["enter", "counter", "X staged", "W0 in LDS"] + [f"step {k}" for k in range(8)] + ["H1 dropped", "H2 dropped", "end"]
    print(f"p={p}: shader cycles since entry (and delta)")
    for k in range(15):
        print(f"   {names[k]:12s} {t[k]:9.0f}  (+{t[k] - t[k - 1] if k else 0:7.0f})")
    dz1, dz2, dz3 = torch.empty(B, n1, device=dev), torch.empty(B, n2, device=dev), torch.empty(B, device=dev)
    dx1 = torch.empty(B, D, device=dev); w3p = torch.empty(nblk, n2 + 1, device=dev)
    for it in range(3):
        _C.check(lib.fr_scorer_bwd(ctypes.byref(d), dy.data_ptr(), None, y.data_ptr(), h1.data_ptr(), h2.data_ptr(), B, used.data_ptr(),
                                   dz1.data_ptr(), dz2.data_ptr(), dz3.data_ptr(), None, dx1.data_ptr(), w3p.data_ptr(), _C.current_stream()), "bwd")
        torch.cuda.synchronize()
    assert raw.fr_debug_scorer_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    print("   bwd stamps (cycles): enter, counter, dz2, w3part, dz1, dX products, end:", (buf[16:23] - buf[16]).astype(np.float64).tolist())
    sys.stdout.flush()
