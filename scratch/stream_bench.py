"""The forward / input-gradient products over many rows: the streaming form (csrc/mlp_stream.hip) against the macro-tile kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "recbole-fairrec_amd"))
from fairrec import _C
lib = _C.lib()
REPS = int(os.environ.get("REPS", 10))
st = _C.current_stream()
SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ.get("SHAPES", "1800000x128x128,1800000x128x64,1800000x64x128,11000000x128x128").split(",")]
for (M, K, N) in SHAPES:
    X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    Y = torch.empty(M, N, device="cuda"); dX = torch.empty(M, K, device="cuda")
    for form in ("stream", "tiles"):
        if form == "tiles":
            os.environ["FAIRREC_LINEAR_NO_STREAM"] = "1"
        else:
            os.environ.pop("FAIRREC_LINEAR_NO_STREAM", None)
        wsz = lib.fr_linear_bwd_weight_workspace_bytes(M, N, K)
        ws = torch.empty(wsz, dtype=torch.uint8, device="cuda"); dW = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
        for name, call in (("bwd_weight", lambda: lib.fr_linear_bwd_weight(Y.data_ptr(), Y.data_ptr(), 0, X.data_ptr(), K, None, 0, None, 1.0, M, N, dW.data_ptr(), db.data_ptr(), ws.data_ptr(), wsz, st)),
                           ("fwd", lambda: lib.fr_linear_fwd(X.data_ptr(), K, None, 0, None, 1.0, W.data_ptr(), b.data_ptr(), M, N, 2, Y.data_ptr(), st)),
                           ("bwd_input", lambda: lib.fr_linear_bwd_input(Y.data_ptr(), Y.data_ptr(), 0, W.data_ptr(), None, 1.0, M, N, dX.data_ptr(), K, None, 0, st))):
            for _ in range(3):
                call()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(REPS):
                call()
            e.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(e) / REPS
            print("M=%9d K=%3d N=%3d %-7s %-9s %7.3f ms  %6.1f TFLOP/s  %6.2f TB/s (in + out once)" %
                  (M, K, N, form, name, ms, 2.0 * M * N * K / ms / 1e9, (M * (K + N) * 4) / ms / 1e9), flush=True)
    del X, Y, dX
