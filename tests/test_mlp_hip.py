"""GPU: the fp32-MFMA dense-layer kernels against stock torch (same op, fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {0: lambda x: x, 1: torch.relu, 2: lambda x: F.leaky_relu(x, 0.01), 3: torch.sigmoid, 4: torch.tanh}


def _lib():
    from fairrec import _C
    return _C


@pytest.mark.parametrize("M,k0,k1,N,act,drop", [
    (64, 16, 0, 16, 1, False), (100, 8, 8, 16, 1, True), (8192, 64, 64, 128, 1, True), (8192, 128, 0, 64, 2, False),
    (777, 64, 0, 1, 1, False), (513, 37, 11, 70, 3, True), (2048, 256, 256, 128, 4, False), (33, 5, 0, 3, 0, True)])
def test_linear_forward_and_backward_match_torch(M, k0, k1, N, act, drop):
    _C = _lib()
    lib = _C.lib()
    g = torch.Generator().manual_seed(M + N)
    K = k0 + k1
    # asymmetric data: a transposed or swapped fragment map cannot pass
    x0 = torch.randn(M, k0, generator=g)
    x1 = torch.randn(M, k1, generator=g) * 0.5 + 0.1 if k1 else None
    W = torch.randn(N, K, generator=g) * 0.3
    b = torch.randn(N, generator=g)
    mask = (torch.rand(M, K, generator=g) >= 0.3) if drop else None
    scale = 1.0 / 0.7 if drop else 1.0
    dY = torch.randn(M, N, generator=g)
    # --- torch reference
    X = torch.cat([x0, x1], 1) if k1 else x0
    Xr = X.clone().requires_grad_()
    Wr, br = W.clone().requires_grad_(), b.clone().requires_grad_()
    Xe = Xr * (mask.float() * scale) if drop else Xr
    Yr = ACTS[act](F.linear(Xe, Wr, br))
    Yr.backward(dY)
    # --- HIP
    dev = "cuda"
    d = lambda t: None if t is None else t.to(dev).contiguous()
    x0d, x1d, Wd, bd, dYd = d(x0), d(x1), d(W), d(b), d(dY)
    md = d(mask.to(torch.uint8)) if drop else None
    Y = torch.empty(M, N, device=dev)
    st = _C.current_stream()
    _C.check(lib.fr_linear_fwd(x0d.data_ptr(), k0, _C.ptr(x1d), k1, _C.ptr(md), scale, Wd.data_ptr(), bd.data_ptr(), M, N,
                               act, Y.data_ptr(), st), "fwd")
    tol = dict(rtol=2e-4, atol=2e-5 * max(1.0, float(Yr.abs().max())))
    torch.testing.assert_close(Y.cpu(), Yr.detach(), **tol)
    dx0 = torch.empty(M, k0, device=dev)
    dx1 = torch.empty(M, k1, device=dev) if k1 else None
    _C.check(lib.fr_linear_bwd_input(dYd.data_ptr(), Y.data_ptr(), act, Wd.data_ptr(), _C.ptr(md), scale, M, N,
                                     dx0.data_ptr(), k0, _C.ptr(dx1), k1, st), "bwd_input")
    dX = torch.cat([dx0, dx1], 1).cpu() if k1 else dx0.cpu()
    torch.testing.assert_close(dX, Xr.grad, rtol=2e-4, atol=2e-5 * max(1.0, float(Xr.grad.abs().max())))
    ws = torch.empty(lib.fr_linear_bwd_weight_workspace_bytes(M, N, K), dtype=torch.uint8, device=dev)
    dW = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev)
    _C.check(lib.fr_linear_bwd_weight(dYd.data_ptr(), Y.data_ptr(), act, x0d.data_ptr(), k0, _C.ptr(x1d), k1, _C.ptr(md),
                                      scale, M, N, dW.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), st), "bwd_w")
    torch.testing.assert_close(dW.cpu(), Wr.grad, rtol=3e-4, atol=3e-5 * max(1.0, float(Wr.grad.abs().max())))
    torch.testing.assert_close(db.cpu(), br.grad, rtol=3e-4, atol=3e-5 * max(1.0, float(br.grad.abs().max())))
    # bit-reproducible (fixed reduction order)
    dW2 = torch.empty_like(dW)
    _C.check(lib.fr_linear_bwd_weight(dYd.data_ptr(), Y.data_ptr(), act, x0d.data_ptr(), k0, _C.ptr(x1d), k1, _C.ptr(md),
                                      scale, M, N, dW2.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), st), "bwd_w")
    assert torch.equal(dW, dW2)
