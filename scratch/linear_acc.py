"""fp32 accuracy of the dense-layer kernels against float64, fast (LDS-DMA) and general forms."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
from fairrec import _C
lib = _C.lib()
st = lambda: _C.current_stream()
torch.manual_seed(0)
for M, K, N in [(8192, 128, 128), (8192, 256, 128), (8192, 512, 128)]:
    X = torch.randn(M, K, device="cuda") * 0.1
    W = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda") * 0.01
    dY = torch.randn(M, N, device="cuda")
    dY = dY - dY.mean(0, keepdim=True)          # BatchNorm-like: columns sum to ~0
    Y = torch.empty(M, N, device="cuda"); dX = torch.empty(M, K, device="cuda"); dW = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    ws = torch.empty(lib.fr_linear_bwd_weight_workspace_bytes(M, N, K), dtype=torch.uint8, device="cuda")
    lib.fr_linear_fwd(X.data_ptr(), K, None, 0, None, 1.0, W.data_ptr(), b.data_ptr(), M, N, 0, Y.data_ptr(), st())
    lib.fr_linear_bwd_input(dY.data_ptr(), Y.data_ptr(), 0, W.data_ptr(), None, 1.0, M, N, dX.data_ptr(), K, None, 0, st())
    lib.fr_linear_bwd_weight(dY.data_ptr(), Y.data_ptr(), 0, X.data_ptr(), K, None, 0, None, 1.0, M, N, dW.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), st())
    Xd, Wd, dYd = X.double(), W.double(), dY.double()
    refs = {"fwd": Xd @ Wd.t() + b.double(), "bwd_in": dYd @ Wd, "bwd_w": dYd.t() @ Xd, "db": dYd.sum(0)}
    got = {"fwd": Y, "bwd_in": dX, "bwd_w": dW, "db": db}
    tor = {"fwd": X @ W.t() + b, "bwd_in": dY @ W, "bwd_w": dY.t() @ X, "db": dY.sum(0)}
    out = []
    for k in refs:
        e = (got[k].double() - refs[k]).abs(); et = (tor[k].double() - refs[k]).abs(); sc = refs[k].abs().max()
        out.append(f"{k}: max {float(e.max() / sc):.1e} mean {float(e.mean() / sc):.1e} (torch fp32: {float(et.max() / sc):.1e} / {float(et.mean() / sc):.1e})")
    print(f"[{M},{K}]->{N}  " + "  ".join(out))
