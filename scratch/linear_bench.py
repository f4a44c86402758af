"""Per-shape device time of the fp32-MFMA linear kernels (fr_linear_fwd / bwd_input / bwd_weight) and their fraction of the
157 TFLOP/s fp32 matrix peak."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
from fairrec import _C
lib = _C.lib()
st = lambda: _C.current_stream()
shapes = [(8192, 512, 128), (8192, 128, 64), (8192, 64, 1), (8192, 128, 256), (8192, 256, 128), (8192, 128, 128),
          (8192, 64, 32), (1100002, 128, 128), (1100002, 128, 64), (1100002, 64, 128)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for M, K, N in shapes:
    X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.zeros(N, device="cuda")
    Y = torch.empty(M, N, device="cuda"); dY = torch.randn(M, N, device="cuda"); dX = torch.empty(M, K, device="cuda")
    dW = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    ws = torch.empty(lib.fr_linear_bwd_weight_workspace_bytes(M, N, K), dtype=torch.uint8, device="cuda")
    f = lambda: lib.fr_linear_fwd(X.data_ptr(), K, None, 0, None, 1.0, W.data_ptr(), (None if os.environ.get("NOBIAS") else b.data_ptr()), M, N, int(os.environ.get("ACT", "2")), Y.data_ptr(), st())
    bi = lambda: lib.fr_linear_bwd_input(dY.data_ptr(), Y.data_ptr(), 0, W.data_ptr(), None, 1.0, M, N, dX.data_ptr(), K, None, 0, st())
    bw = lambda: lib.fr_linear_bwd_weight(dY.data_ptr(), Y.data_ptr(), 0, X.data_ptr(), K, None, 0, None, 1.0, M, N, dW.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), st())
    fl = 2.0 * M * K * N
    t = [timeit(f), timeit(bi), timeit(bw)]
    if os.environ.get("KTIME"):   # device time of the kernels themselves (HIP events around each launch)
        for k, fn in enumerate((f, bi, bw)):
            _C.prof_reset(); _C.prof_enable(True)
            for _ in range(20): fn()
            torch.cuda.synchronize(); _C.prof_enable(False)
            t[k] = sum(v[0] for v in _C.prof_read().values()) / 20 * 1e3
    # correctness spot check against torch
    f(); ref = torch.nn.functional.leaky_relu(X @ W.t() + b, 0.01)
    err = float((Y - ref).abs().max() / ref.abs().max())
    bi(); bw()
    e2 = float((dX - dY @ W).abs().max() / (dY @ W).abs().max())
    e3 = float((dW - dY.t() @ X).abs().max() / (dY.t() @ X).abs().max())
    e4 = float((db - dY.sum(0)).abs().max() / dY.sum(0).abs().max())
    err = max(err, e2, e3, e4)
    print(f"[{M},{K}]->{N}: fwd {t[0]:8.1f} us ({fl / t[0] / 1e6 / 157:5.1%} of 157 TF)  bwd_in {t[1]:8.1f} us ({fl / t[1] / 1e6 / 157:5.1%})  "
          f"bwd_w {t[2]:8.1f} us ({fl / t[2] / 1e6 / 157:5.1%})   max rel err {err:.1e}", flush=True)
