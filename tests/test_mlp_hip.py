"""GPU: the fp32-MFMA dense-layer kernels against stock torch (same op, fp32)."""
import os

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
    (777, 64, 0, 1, 1, False), (513, 37, 11, 70, 3, True), (2048, 256, 256, 128, 4, False), (33, 5, 0, 3, 0, True),
    # act = 0, no mask, widths multiples of 32: all three products take their LDS-DMA form (csrc/mlp_glds.hip) -- rows that
    # are not a multiple of 32 (tail tiles, reduction tail of the weight gradient), two-block input, split reductions
    (1000, 64, 64, 96, 0, False), (8192, 256, 0, 128, 0, False), (50, 32, 0, 32, 0, False), (4100, 128, 0, 256, 0, False),
    (8192, 512, 0, 128, 0, False)])
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


@pytest.mark.parametrize("M,K,act,need_dx", [(8192, 64, 1, True), (777, 128, 3, True), (5000, 512, 0, False), (1, 64, 4, True),
                                             (8192, 64, 2, False)])
def test_one_output_layer_forward_and_fused_backward_match_torch(M, K, act, need_dx):
    """N == 1 (a scorer's / discriminator's last layer): fr_linear_fwd takes its row-dot form, fr_linear_n1_bwd gives dW, db
    and dX in one pass; shapes the form does not cover are refused so that the caller takes the general calls."""
    _C = _lib()
    lib = _C.lib()
    g = torch.Generator().manual_seed(M + K)
    X = torch.randn(M, K, generator=g)
    W = torch.randn(1, K, generator=g) * 0.3
    b = torch.randn(1, generator=g)
    dY = torch.randn(M, 1, generator=g)
    Xr, Wr, br = X.clone().requires_grad_(), W.clone().requires_grad_(), b.clone().requires_grad_()
    Yr = ACTS[act](F.linear(Xr, Wr, br))
    Yr.backward(dY)
    Xd, Wd, bd, dYd = X.cuda(), W.cuda(), b.cuda(), dY.cuda()
    st = _C.current_stream()
    Y = torch.empty(M, 1, device="cuda")
    _C.check(lib.fr_linear_fwd(Xd.data_ptr(), K, None, 0, None, 1.0, Wd.data_ptr(), bd.data_ptr(), M, 1, act, Y.data_ptr(), st),
             "fwd")
    torch.testing.assert_close(Y.cpu(), Yr.detach(), rtol=2e-4, atol=2e-5 * max(1.0, float(Yr.abs().max())))
    ws = torch.empty(lib.fr_linear_bwd_weight_workspace_bytes(M, 1, K), dtype=torch.uint8, device="cuda")
    dW, db = torch.empty(1, K, device="cuda"), torch.empty(1, device="cuda")
    dX = torch.empty(M, K, device="cuda") if need_dx else None
    for rep in range(2):
        out = torch.empty_like(dW)
        _C.check(lib.fr_linear_n1_bwd(dYd.data_ptr(), Y.data_ptr(), act, Xd.data_ptr(), K, Wd.data_ptr(), M, 0.0, _C.ptr(dX),
                                      out.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), st), "n1_bwd")
        assert rep == 0 or torch.equal(out, dW)                     # fixed reduction order
        dW = out
    torch.testing.assert_close(dW.cpu(), Wr.grad, rtol=3e-4, atol=3e-5 * max(1.0, float(Wr.grad.abs().max())))
    torch.testing.assert_close(db.cpu(), br.grad, rtol=3e-4, atol=3e-5 * max(1.0, float(br.grad.abs().max())))
    if need_dx:
        torch.testing.assert_close(dX.cpu(), Xr.grad, rtol=2e-4, atol=2e-5 * max(1.0, float(Xr.grad.abs().max())))
    # not this form: K not a multiple of 64
    rc = lib.fr_linear_n1_bwd(dYd.data_ptr(), Y.data_ptr(), act, Xd.data_ptr(), 48, Wd.data_ptr(), M, 0.0, None, dW.data_ptr(),
                              db.data_ptr(), ws.data_ptr(), ws.numel(), st)
    assert rc == -3                                                 # FR_EUNSUPPORTED


@pytest.mark.parametrize("M,N,K", [(8192, 64, 128), (1000, 128, 512), (8192, 1, 64), (333, 1, 128)])
def test_input_gradient_taken_on_through_a_dropped_relu(M, N, K):
    """fr_linear_bwd_input_relu / fr_linear_n1_bwd(relu_scale): the layer's input is Xd = relu(z) o keep; the result is the
    gradient at z, i.e. what fr_linear_bwd_input followed by fr_act_bwd_dropped give."""
    _C = _lib()
    lib = _C.lib()
    g = torch.Generator().manual_seed(M + N + K)
    z = torch.randn(M, K, generator=g).requires_grad_()
    keep = (torch.rand(M, K, generator=g) >= 0.4).float() / 0.6
    Xd = torch.relu(z) * keep
    W = (torch.randn(N, K, generator=g) * 0.3)
    dY = torch.randn(M, N, generator=g)
    (Xd @ W.t()).backward(dY)
    Xdd, Wd, dYd = Xd.detach().cuda(), W.cuda(), dY.cuda()
    dA = torch.empty(M, K, device="cuda")
    st = _C.current_stream()
    if N == 1:
        ws = torch.empty(lib.fr_linear_bwd_weight_workspace_bytes(M, 1, K), dtype=torch.uint8, device="cuda")
        dW, db = torch.empty(1, K, device="cuda"), torch.empty(1, device="cuda")
        _C.check(lib.fr_linear_n1_bwd(dYd.data_ptr(), dYd.data_ptr(), 0, Xdd.data_ptr(), K, Wd.data_ptr(), M, 1 / 0.6,
                                      dA.data_ptr(), dW.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), st), "n1_bwd")
        torch.testing.assert_close(dW.cpu(), (dY.t() @ Xd.detach()), rtol=3e-4, atol=3e-4)
    else:
        _C.check(lib.fr_linear_bwd_input_relu(dYd.data_ptr(), Wd.data_ptr(), M, N, K, Xdd.data_ptr(), 1 / 0.6, dA.data_ptr(), st),
                 "bwd_input_relu")
    torch.testing.assert_close(dA.cpu(), z.grad, rtol=2e-4, atol=2e-5 * max(1.0, float(z.grad.abs().max())))
    rc = lib.fr_linear_bwd_input_relu(dYd.data_ptr(), Wd.data_ptr(), M, 24, K, Xdd.data_ptr(), 1 / 0.6, dA.data_ptr(), st)
    assert rc == -3


@pytest.mark.parametrize("act", [1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(8192, 64, 128), (1000, 128, 512), (333, 32, 32)])
def test_input_gradient_taken_on_through_the_activation_below(M, N, K, act):
    """fr_linear_bwd_input_act: dA = (dY W) o act'(Yin) -- what fr_linear_bwd_input followed by the fr_act_bwd of the layer
    below give, BIT FOR BIT (the product is rounded, then multiplied, in both), and torch's autograd of act(z) @ W^T."""
    _C = _lib()
    lib = _C.lib()
    g = torch.Generator().manual_seed(M + N + K + act)
    z = torch.randn(M, K, generator=g).requires_grad_()
    Yin = ACTS[act](z)
    W = (torch.randn(N, K, generator=g) * 0.3)
    dY = torch.randn(M, N, generator=g)
    (Yin @ W.t()).backward(dY)
    Yd, Wd, dYd = Yin.detach().cuda(), W.cuda(), dY.cuda()
    st = _C.current_stream()
    dA = torch.empty(M, K, device="cuda")
    _C.check(lib.fr_linear_bwd_input_act(dYd.data_ptr(), Wd.data_ptr(), M, N, K, Yd.data_ptr(), act, dA.data_ptr(), st), "bwd_input_act")
    dX = torch.empty(M, K, device="cuda")
    two = torch.empty(M, K, device="cuda")
    _C.check(lib.fr_linear_bwd_input(dYd.data_ptr(), dYd.data_ptr(), 0, Wd.data_ptr(), None, 1.0, M, N, dX.data_ptr(), K, None, 0, st),
             "bwd_input")
    _C.check(lib.fr_act_bwd(dX.data_ptr(), Yd.data_ptr(), act, dX.numel(), two.data_ptr(), st), "act_bwd")
    assert torch.equal(dA, two)
    torch.testing.assert_close(dA.cpu(), z.grad, rtol=2e-4, atol=2e-5 * max(1.0, float(z.grad.abs().max())))
    assert lib.fr_linear_bwd_input_act(dYd.data_ptr(), Wd.data_ptr(), M, 24, K, Yd.data_ptr(), act, dA.data_ptr(), st) == -3


@pytest.mark.parametrize("act", ["relu", "leakyrelu", "tanh"])
def test_mlp_backward_with_the_activation_folded_into_the_product_below_equals_the_separate_pass(act, monkeypatch):
    """MLPLayers.backward: a hidden layer's activation derivative rides in the epilogue of the input-gradient product of the
    layer above (fr_linear_bwd_input_act) instead of a pass of its own (FAIRREC_ACT_BWD_SEPARATE=1 restores it): every
    gradient equal bit for bit, one launch and one [M, width] pass less per hidden layer."""
    from fairrec import _C
    from fairrec.model.layers import MLPLayers
    torch.manual_seed(3)
    mlp = MLPLayers([128, 128, 64, 32], activation=act).cuda()
    x = torch.randn(4096, 128, device="cuda")
    grads, passes = [], []
    lib = _C.lib()
    real = lib.fr_act_bwd
    calls = []
    monkeypatch.setattr(lib, "fr_act_bwd", lambda *a: calls.append(1) or real(*a))
    for separate in (True, False):
        if separate:
            monkeypatch.setenv("FAIRREC_ACT_BWD_SEPARATE", "1")
        else:
            monkeypatch.delenv("FAIRREC_ACT_BWD_SEPARATE")
        xi = x.clone().requires_grad_()
        for p in mlp.parameters():
            p.grad = None
        del calls[:]
        mlp(xi).square().sum().backward()
        torch.cuda.synchronize()
        passes.append(len(calls))
        grads.append([xi.grad.clone()] + [p.grad.clone() for p in mlp.parameters()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    assert passes == [3, 1], passes          # only the top layer's activation still takes a pass of its own


def test_act_bwd_through_a_relu_dropped_in_place():
    """fr_act_bwd_dropped: Yd = relu(z) o keep, gradient at z = dY o keep o relu'(z) without z or the keep pattern."""
    _C = _lib()
    lib = _C.lib()
    g = torch.Generator().manual_seed(5)
    z = torch.randn(513, 64, generator=g).requires_grad_()
    keep = (torch.rand(513, 64, generator=g) >= 0.3).float() / 0.7
    Yd = torch.relu(z) * keep
    dY = torch.randn(513, 64, generator=g)
    Yd.backward(dY)
    out = torch.empty(513, 64, device="cuda")
    dYd, Ydd = dY.cuda(), Yd.detach().cuda()
    _C.check(lib.fr_act_bwd_dropped(dYd.data_ptr(), Ydd.data_ptr(), 1 / 0.7, Yd.numel(), out.data_ptr(),
                                    _C.current_stream()), "act_bwd_dropped")
    torch.testing.assert_close(out.cpu(), z.grad, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("act", [1, 2, 3, 4])
def test_act_bwd_prepass(act):
    """fr_act_bwd: dY o act'(Y), the pre-pass that lets a layer's two backward products run with act = 0."""
    _C = _lib()
    lib = _C.lib()
    g = torch.Generator().manual_seed(act)
    X = torch.randn(513, 64, generator=g).requires_grad_()
    Y = ACTS[act](X)
    dY = torch.randn(513, 64, generator=g)
    Y.backward(dY)
    Yd, dYd = Y.detach().cuda(), dY.cuda()
    out = torch.empty_like(Yd)
    _C.check(lib.fr_act_bwd(dYd.data_ptr(), Yd.data_ptr(), act, Yd.numel(), out.data_ptr(), _C.current_stream()), "act_bwd")
    torch.testing.assert_close(out.cpu(), X.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,N,act", [(32, 8, 2), (128, 64, 1), (300, 70, 2), (8192, 256, 2), (40000, 33, 0), (1, 5, 1)])
def test_batchnorm_forward_and_backward_match_torch(M, N, act):
    """Row-chunked BatchNorm1d (stats + apply launches) against nn.BatchNorm1d in training mode, including a
    far-from-zero column mean (the Chan fold must not cancel) and the running statistics."""
    _C = _lib()
    lib = _C.lib()
    g = torch.Generator().manual_seed(M * 7 + N)
    Z = torch.randn(M, N, generator=g) * (torch.rand(N, generator=g) * 3 + 0.1) + torch.randn(N, generator=g) * 50
    gamma = torch.rand(N, generator=g) + 0.5
    beta = torch.randn(N, generator=g)
    dY = torch.randn(M, N, generator=g)
    bn = torch.nn.BatchNorm1d(N)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    if M > 1:
        Zr = Z.clone().requires_grad_()
        pre = bn(Zr)
        Yr = ACTS[act](pre)
        # an output within rounding of the activation's kink may take the other slope: no gradient through those
        smooth = (pre.detach().abs() > 1e-4) if act in (1, 2) else torch.ones_like(dY, dtype=torch.bool)
        assert smooth.float().mean() > 0.999
        dY = dY * smooth
        Yr.backward(dY)
    dev = "cuda"
    Zd, gd, bd, dYd = Z.to(dev), gamma.to(dev), beta.to(dev), dY.to(dev)
    rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    Y, xh, inv = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev), torch.empty(N, device=dev)
    ws = torch.empty(lib.fr_bn_workspace_bytes(M, N), dtype=torch.uint8, device=dev)
    st = _C.current_stream()
    _C.check(lib.fr_bn_fwd(Zd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), M, N, act,
                           Y.data_ptr(), xh.data_ptr(), inv.data_ptr(), ws.data_ptr(), ws.numel(), st), "bn_fwd")
    if M == 1:      # torch refuses a single row in training mode; the kernel yields xhat = 0
        assert torch.equal(xh.cpu(), torch.zeros(1, N))
        return
    torch.testing.assert_close(Y.cpu(), Yr.detach(), rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(rm.cpu(), bn.running_mean, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rv.cpu(), bn.running_var, rtol=1e-4, atol=1e-6)
    dZ, dg, db = torch.empty(M, N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
    _C.check(lib.fr_bn_bwd(dYd.data_ptr(), Y.data_ptr(), act, xh.data_ptr(), inv.data_ptr(), gd.data_ptr(), M, N,
                           dZ.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), st), "bn_bwd")
    s = float(Zr.grad.abs().max())
    torch.testing.assert_close(dZ.cpu(), Zr.grad, rtol=5e-4, atol=2e-4 * max(s, 1e-3))
    torch.testing.assert_close(dg.cpu(), bn.weight.grad, rtol=5e-4, atol=1e-4 * max(1.0, float(bn.weight.grad.abs().max())))
    torch.testing.assert_close(db.cpu(), bn.bias.grad, rtol=5e-4, atol=1e-4 * max(1.0, float(bn.bias.grad.abs().max())))


@pytest.mark.parametrize("D", [8, 64, 128, 256, 100])
def test_spmm_csr_matches_torch_sparse(D):
    """fr_spmm_csr (vector-gather variants for D = 64/128/256, strided otherwise) against torch.sparse.mm, with rows
    longer than one 64-nonzero batch and empty rows."""
    import scipy.sparse as sp
    _C = _lib()
    lib = _C.lib()
    rng = np.random.default_rng(D)
    n = 500
    dens = sp.random(n, n, density=0.05, random_state=3, format="lil", dtype=np.float32)
    dens[7, :] = rng.random(n).astype(np.float32)          # a 500-nonzero row
    dens[11, :] = 0                                        # an empty row
    csr = dens.tocsr()
    X = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32))
    ref = torch.sparse.mm(torch.sparse_csr_tensor(torch.from_numpy(csr.indptr.astype(np.int64)),
                                                  torch.from_numpy(csr.indices.astype(np.int64)),
                                                  torch.from_numpy(csr.data), size=(n, n)), X)
    dev = "cuda"
    ip = torch.from_numpy(csr.indptr.astype(np.int64)).to(dev)
    ci = torch.from_numpy(csr.indices.astype(np.int32)).to(dev)
    va = torch.from_numpy(csr.data).to(dev)
    Xd = X.to(dev)
    Y = torch.empty(n, D, device=dev)
    _C.check(lib.fr_spmm_csr(ip.data_ptr(), ci.data_ptr(), va.data_ptr(), Xd.data_ptr(), n, D, Y.data_ptr(),
                             _C.current_stream()), "spmm")
    torch.testing.assert_close(Y.cpu(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B", [200, 8192])
def test_bpr_outer_at_the_baseline_batch(B):
    """fr_bpr_outer (PFCN_BiasedMF's [B] + [B,1] -> [B,B] BPR, pfcn_biasedmf.py:192-195) at BASELINE.json configs[2]'s
    B = 8192 -- 67 M exp/log pairs, nothing of size B^2 stored -- against the float64 restatement of the reference's
    expression (oracle/pfcn.py::bpr_outer, which walks the matrix in row chunks)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle.pfcn import bpr_outer
    from fairrec.functional import BprBroadcast
    g = torch.Generator().manual_seed(B)
    dp, dn = torch.randn(B, generator=g) * 2.0, torch.randn(B, generator=g) * 2.0
    bp, bn = torch.randn(B, 1, generator=g) * 0.5, torch.randn(B, 1, generator=g) * 0.5
    want, wda, wdc = bpr_outer(dp - dn, (bp - bn).view(-1))
    leaves = [t.cuda().requires_grad_(True) for t in (dp, dn, torch.zeros(B, 1), bp, bn, torch.zeros(1))]
    loss = BprBroadcast.apply(*leaves)
    loss.backward()
    assert abs(float(loss) - float(want)) <= 1e-5 * abs(float(want))
    scale = float(wda.abs().max())
    for got, ref in ((leaves[0].grad, wda), (leaves[1].grad, -wda), (leaves[3].grad.view(-1), wdc),
                     (leaves[4].grad.view(-1), -wdc)):
        err = (got.cpu().double() - ref).abs()
        assert bool((err <= 1e-4 * ref.abs() + 1e-6 * scale).all()), float(err.max())
    assert float(leaves[2].grad.abs().max()) == 0.0 and float(leaves[5].grad.abs().max()) == 0.0   # App. B-1: exactly zero


@pytest.mark.parametrize("B,seeded", [(200, False), (8192, True)])
def test_packed_bpr_and_repeated_row_dots_equal_the_unpacked_forms(B, seeded):
    """RowDotRep + BprBroadcastPacked (PFCN_BiasedMF's score / loss on one [2B] lookup of [pos | neg] items) against
    RowDot x 2 + BprBroadcast on the slices: same loss, same gradients for the user rows, item rows, item biases; exactly
    zero for user bias and global bias.  `seeded`: backward() from the cached one the graphed step uses (no scaling launches)
    and from an arbitrary upstream gradient."""
    from fairrec import _C
    from fairrec.functional import BprBroadcast, BprBroadcastPacked, RowDot, RowDotRep
    g = torch.Generator().manual_seed(B)
    D = 64
    ue = (torch.randn(B, D, generator=g) * 0.3).cuda()
    ie = (torch.randn(2 * B, D, generator=g) * 0.3).cuda()
    ub = torch.randn(B, 1, generator=g).cuda()
    ib = (torch.randn(2 * B, 1, generator=g) * 0.5).cuda()
    gb = torch.zeros(1).cuda()

    def leaves():
        return [t.clone().requires_grad_(True) for t in (ue, ie, ub, ib, gb)]

    a = leaves()
    loss_a = BprBroadcast.apply(RowDot.apply(a[0], a[1][:B]), RowDot.apply(a[0], a[1][B:]), a[2], a[3][:B], a[3][B:], a[4])
    b = leaves()
    loss_b = BprBroadcastPacked.apply(RowDotRep.apply(b[0], b[1]), b[2], b[3], b[4])
    if seeded:
        loss_a.backward(_C.one("cuda"))
        loss_b.backward(_C.one("cuda"))
    else:
        (loss_a * 0.37).backward()
        (loss_b * 0.37).backward()
    assert abs(float(loss_a) - float(loss_b)) <= 1e-6 * abs(float(loss_a))
    for x, y in zip(a, b):
        torch.testing.assert_close(y.grad, x.grad, rtol=1e-5, atol=1e-7 * max(1.0, float(x.grad.abs().max())))
    assert float(b[2].grad.abs().max()) == 0.0 and float(b[4].grad.abs().max()) == 0.0


@pytest.mark.parametrize("M,k0,k1,N", [(8192, 256, 0, 128), (8192, 256, 256, 128), (1000, 64, 64, 96), (50, 32, 0, 32),
                                        (4100, 128, 0, 256), (8192, 128, 0, 64), (200, 128, 0, 128), (333, 96, 32, 160)])
def test_shared_tile_gemms_are_bit_identical_to_the_wave_private_form(M, k0, k1, N):
    """The LDS-DMA kernels with operand chunks shared by the four waves of a 64 x 64 macro tile (csrc/mlp_glds.hip,
    linear_glds64_kernel) accumulate every output in the order of the wave-private kernel: forward, input gradient and
    weight gradient must come out bit for bit the same (odd tile counts, two-block inputs, split reductions included)."""
    _C = _lib()
    lib = _C.lib()
    g = torch.Generator().manual_seed(M + N + k0)
    K = k0 + k1
    x0 = torch.randn(M, k0, generator=g).cuda()
    x1 = (torch.randn(M, k1, generator=g) * 0.5).cuda() if k1 else None
    W = (torch.randn(N, K, generator=g) * 0.3).cuda()
    b = torch.randn(N, generator=g).cuda()
    dY = torch.randn(M, N, generator=g).cuda()
    st = _C.current_stream()

    def products():
        Y = torch.empty(M, N, device="cuda")
        _C.check(lib.fr_linear_fwd(x0.data_ptr(), k0, _C.ptr(x1), k1, None, 1.0, W.data_ptr(), b.data_ptr(), M, N, 1,
                                   Y.data_ptr(), st), "fwd")
        dx0 = torch.empty(M, k0, device="cuda")
        dx1 = torch.empty(M, k1, device="cuda") if k1 else None
        _C.check(lib.fr_linear_bwd_input(dY.data_ptr(), dY.data_ptr(), 0, W.data_ptr(), None, 1.0, M, N, dx0.data_ptr(), k0,
                                         _C.ptr(dx1), k1, st), "bwd_input")
        ws = torch.empty(lib.fr_linear_bwd_weight_workspace_bytes(M, N, K), dtype=torch.uint8, device="cuda")
        dW, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
        _C.check(lib.fr_linear_bwd_weight(dY.data_ptr(), dY.data_ptr(), 0, x0.data_ptr(), k0, _C.ptr(x1), k1, None, 1.0, M, N,
                                          dW.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), st), "bwd_w")
        torch.cuda.synchronize()
        return [Y, dx0] + ([dx1] if k1 else []) + [dW, db]

    shared = products()
    os.environ["FAIRREC_LINEAR_NO_SHARED"] = "1"
    try:
        private = products()
    finally:
        del os.environ["FAIRREC_LINEAR_NO_SHARED"]
    for a, c in zip(shared, private):
        assert torch.equal(a, c)
    X = torch.cat([x0, x1], 1) if k1 else x0
    torch.testing.assert_close(shared[0], torch.relu(X @ W.t() + b), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("M,N,k0,k1", [(8192, 128, 128, 0), (1000, 40, 64, 32), (33, 256, 32, 0), (4097, 64, 96, 96)],
                         ids=["full", "ragged_n40", "two_tiles", "ragged_two_blocks"])
def test_bn_statistics_from_the_product_epilogue_equal_the_statistics_launch(M, N, k0, k1):
    """fr_linear_fwd_bnstats + fr_bn_fwd_ex(have_stats = 1) against fr_linear_fwd + fr_bn_fwd: Z bit for bit (the same
    product), the BatchNorm outputs, xhat, 1/std and the running statistics to rounding (the column sums of a 32-row chunk are
    taken in another order) -- ragged last tiles and column counts that are no multiple of 32 included."""
    from fairrec import _C
    lib = _C.lib()
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x0 = torch.randn(M, k0, device=dev, generator=g)
    x1 = torch.randn(M, k1, device=dev, generator=g) if k1 else None
    W = torch.randn(N, k0 + k1, device=dev, generator=g) * 0.1
    b = torch.randn(N, device=dev, generator=g)
    gamma = torch.rand(N, device=dev, generator=g) + 0.5
    beta = torch.randn(N, device=dev, generator=g)
    st = _C.current_stream()
    outs = []
    for epilogue in (True, False):
        Z = torch.empty(M, N, device=dev)
        ws = torch.zeros(lib.fr_bn_workspace_bytes(M, N), dtype=torch.uint8, device=dev)
        rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
        Y, xh, inv = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev), torch.empty(N, device=dev)
        if epilogue:
            _C.check(lib.fr_linear_fwd_bnstats(x0.data_ptr(), k0, _C.ptr(x1), k1, W.data_ptr(), b.data_ptr(), M, N, Z.data_ptr(),
                                               ws.data_ptr(), ws.numel(), st), "fr_linear_fwd_bnstats")
        else:
            _C.check(lib.fr_linear_fwd(x0.data_ptr(), k0, _C.ptr(x1), k1, None, 1.0, W.data_ptr(), b.data_ptr(), M, N, 0, Z.data_ptr(),
                                       st), "fr_linear_fwd")
        nbt = torch.full((1,), 5, dtype=torch.int64, device=dev)
        _C.check(lib.fr_bn_fwd_ex(Z.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), M, N, 2,
                                  Y.data_ptr(), xh.data_ptr(), inv.data_ptr(), ws.data_ptr(), ws.numel(), 1 if epilogue else 0,
                                  None, 0.0, 0, 0, None, None, None, nbt.data_ptr(), 2, st), "fr_bn_fwd_ex")
        torch.cuda.synchronize()
        assert int(nbt.item()) == 7        # the layer's batch counter moves by `passes` in the fold launch
        outs.append((Z, Y, xh, inv, rm, rv))
    a, r = outs
    assert torch.equal(a[0], r[0])
    for x, y in zip(a[1:], r[1:]):
        torch.testing.assert_close(x, y, rtol=2e-5, atol=2e-6)
    ref = torch.nn.functional.batch_norm(r[0].double(), None, None, gamma.double(), beta.double(), True, 0.1, 1e-5)
    torch.testing.assert_close(a[1].double(), torch.nn.functional.leaky_relu(ref, 0.01), rtol=1e-4, atol=1e-5)


def _bn_mlp_run(mlp, x, w_out, need_dx=True, frozen=False):
    xi = x.clone().requires_grad_(need_dx)
    for p in mlp.parameters():
        p.grad = None
    y = mlp(xi, frozen=frozen)
    (y * w_out).sum().backward()
    torch.cuda.synchronize()
    out = {"y": y.detach().clone(), "dx": xi.grad.clone() if need_dx else None}
    for n, p in mlp.named_parameters():
        out["g." + n] = None if p.grad is None else p.grad.clone()
    for n, b in mlp.named_buffers():
        out["b." + n] = b.clone()
    return out


@pytest.mark.parametrize("M,K,N", [(40000, 128, 128), (32768, 128, 64), (33333, 64, 128), (50001, 96, 128), (70000, 32, 64)])
def test_streaming_products_are_bit_identical_to_the_macro_tile_kernels(M, K, N, monkeypatch):
    """csrc/mlp_stream.hip (many rows, 64 / 128 output columns: persistent workgroups, weights resident in LDS, the input rows
    streamed once by loader waves) against the macro-tile kernels (FAIRREC_LINEAR_NO_STREAM=1): the forward product with every
    activation, the input gradient plain and taken on through the activation below -- the same sums in the same order."""
    _C = _lib()
    lib = _C.lib()
    st = _C.current_stream()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    X = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) * 0.2
    b = torch.randn(N, device="cuda", generator=g)
    dY = torch.randn(M, N, device="cuda", generator=g)
    Yin = torch.rand(M, K, device="cuda", generator=g) - 0.3
    outs = {}
    for form in ("stream", "tiles"):
        if form == "tiles":
            monkeypatch.setenv("FAIRREC_LINEAR_NO_STREAM", "1")
        res = []
        for act in (0, 1, 2, 3, 4):
            Y = torch.full((M, N), float("nan"), device="cuda")
            _C.check(lib.fr_linear_fwd(X.data_ptr(), K, None, 0, None, 1.0, W.data_ptr(), b.data_ptr(), M, N, act, Y.data_ptr(), st), "fwd")
            res.append(Y)
        dX = torch.full((M, K), float("nan"), device="cuda")
        _C.check(lib.fr_linear_bwd_input(dY.data_ptr(), dY.data_ptr(), 0, W.data_ptr(), None, 1.0, M, N, dX.data_ptr(), K, None, 0, st), "bwd")
        res.append(dX)
        for act in (1, 2, 4):
            dA = torch.full((M, K), float("nan"), device="cuda")
            _C.check(lib.fr_linear_bwd_input_act(dY.data_ptr(), W.data_ptr(), M, N, K, Yin.data_ptr(), act, dA.data_ptr(), st), "bwd_act")
            res.append(dA)
        torch.cuda.synchronize()
        outs[form] = res
    for a, c in zip(outs["stream"], outs["tiles"]):
        assert not torch.isnan(c).any()
        assert torch.equal(a, c), float((a - c).abs().max())
    ref = torch.nn.functional.linear(X.double(), W.double(), b.double())
    torch.testing.assert_close(outs["stream"][0].double(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(70000, 128, 128), (65536, 128, 64), (100001, 64, 128), (66001, 64, 64), (300000, 128, 128)])
def test_streaming_weight_gradient_is_bit_identical_to_the_macro_tile_kernel(M, N, K, monkeypatch):
    """csrc/mlp_stream.hip's weight gradient (one workgroup per row split forms ALL of dY^T X: both operands read once) against
    the macro-tile kernel (FAIRREC_LINEAR_NO_STREAM=1): the same splits, the same slabs, so dW and db bit for bit."""
    _C = _lib()
    lib = _C.lib()
    st = _C.current_stream()
    g = torch.Generator(device="cuda").manual_seed(M + N)
    X = torch.randn(M, K, device="cuda", generator=g)
    dY = torch.randn(M, N, device="cuda", generator=g)
    got = {}
    for form in ("stream", "tiles"):
        if form == "tiles":
            monkeypatch.setenv("FAIRREC_LINEAR_NO_STREAM", "1")
        dW = torch.full((N, K), float("nan"), device="cuda")
        db = torch.full((N,), float("nan"), device="cuda")
        ws = torch.empty(lib.fr_linear_bwd_weight_workspace_bytes(M, N, K), dtype=torch.uint8, device="cuda")
        _C.check(lib.fr_linear_bwd_weight(dY.data_ptr(), dY.data_ptr(), 0, X.data_ptr(), K, None, 0, None, 1.0, M, N, dW.data_ptr(),
                                          db.data_ptr(), ws.data_ptr(), ws.numel(), st), "fr_linear_bwd_weight")
        # ... and as a job of the multi-layer call (the path MLPLayers.backward takes)
        dW2 = torch.full((N, K), float("nan"), device="cuda")
        db2 = torch.full((N,), float("nan"), device="cuda")
        jobs = (_C.FrWgradJob * 1)(_C.FrWgradJob(dY.data_ptr(), X.data_ptr(), K, None, 0, N, dW2.data_ptr(), db2.data_ptr(), None, 0))
        wsm = torch.empty(lib.fr_linear_bwd_weight_multi_workspace_bytes(jobs, 1, M), dtype=torch.uint8, device="cuda")
        _C.check(lib.fr_linear_bwd_weight_multi(jobs, 1, M, wsm.data_ptr(), wsm.numel(), st), "fr_linear_bwd_weight_multi")
        torch.cuda.synchronize()
        got[form] = (dW, db, dW2, db2)
    for a, c in zip(got["stream"], got["tiles"]):
        assert not torch.isnan(c).any()
        assert torch.equal(a, c), float((a - c).abs().max())
    ref = dY.double().t() @ X.double()
    torch.testing.assert_close(got["stream"][0].double(), ref, rtol=1e-4, atol=1e-3 * float(ref.abs().max()))
    torch.testing.assert_close(got["stream"][1].double(), dY.double().sum(0), rtol=1e-4, atol=1e-3 * float(M) ** 0.5)


def test_streaming_products_at_the_table_size_of_baseline_config_3():
    """[11 000 002, 128] -- the whole embedding table FairGo's filter MLP runs over (BASELINE.json configs[3]) -- through the
    streaming kernels: size-independent properties instead of a full reference.  The forward product is linear in its input
    (no bias, no activation), sampled rows equal the float64 product, the input gradient is the transpose map (<dY, X W^T> =
    <dY W, X>), and the weight gradient equals dY^T X on a sampled block of rows summed the slow way."""
    _C = _lib()
    lib = _C.lib()
    st = _C.current_stream()
    M, K, N = 11_000_002, 128, 128
    g = torch.Generator(device="cuda").manual_seed(9)
    X1 = torch.randn(M, K, device="cuda", generator=g)
    X2 = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) * 0.1

    def fwd(X):
        Y = torch.empty(M, N, device="cuda")
        _C.check(lib.fr_linear_fwd(X.data_ptr(), K, None, 0, None, 1.0, W.data_ptr(), None, M, N, 0, Y.data_ptr(), st), "fwd")
        return Y

    Y1, Y2 = fwd(X1), fwd(X2)
    X2.add_(X1)                                   # X2 <- X1 + X2
    Y12 = fwd(X2)
    err = (Y12 - (Y1 + Y2)).abs().max()
    assert float(err) <= 1e-4 * float(Y12.abs().max()), float(err)
    rows = torch.randint(0, M, (4096,), device="cuda", generator=g)
    rows[:3] = torch.tensor([0, M - 1, M - 2], device="cuda")          # the ragged last tile included
    ref = X1[rows].double() @ W.double().t()
    torch.testing.assert_close(Y1[rows].double(), ref, rtol=1e-4, atol=1e-5)
    del Y2, Y12
    # input gradient: adjointness
    dY = torch.randn(M, N, device="cuda", generator=g)
    dX = torch.empty(M, K, device="cuda")
    _C.check(lib.fr_linear_bwd_input(dY.data_ptr(), dY.data_ptr(), 0, W.data_ptr(), None, 1.0, M, N, dX.data_ptr(), K, None, 0, st), "bwd")
    lhs = (dY.double() * Y1.double()).sum()
    rhs = (dX.double() * X1.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * max(abs(float(lhs)), float((dY.double() ** 2).sum().sqrt() * (Y1.double() ** 2).sum().sqrt())), \
        (float(lhs), float(rhs))
    torch.testing.assert_close(dX[rows].double(), dY[rows].double() @ W.double(), rtol=1e-4, atol=1e-5)
    # weight gradient: zero everything but a block of rows, whose contribution is known
    blk = slice(7_000_001, 7_000_001 + 3000)
    dYb = torch.zeros_like(dY)
    dYb[blk] = dY[blk]
    dW = torch.empty(N, K, device="cuda")
    db = torch.empty(N, device="cuda")
    ws = torch.empty(lib.fr_linear_bwd_weight_workspace_bytes(M, N, K), dtype=torch.uint8, device="cuda")
    _C.check(lib.fr_linear_bwd_weight(dYb.data_ptr(), dYb.data_ptr(), 0, X1.data_ptr(), K, None, 0, None, 1.0, M, N, dW.data_ptr(),
                                      db.data_ptr(), ws.data_ptr(), ws.numel(), st), "wgrad")
    ref = dY[blk].double().t() @ X1[blk].double()
    torch.testing.assert_close(dW.double(), ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    torch.testing.assert_close(db.double(), dY[blk].double().sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M", [200, 8192])
@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("widths", [[128, 256, 128, 128, 64, 32, 1], [64, 32, 32]], ids=["discriminator", "narrow"])
def test_bn_backward_statistics_from_the_product_epilogue_are_the_three_launches_bits(M, p, widths):
    """fr_linear_bwd_input_bnstats + fr_bn_bwd_ex(have_stats): the input-gradient product of a layer above a BatchNorm layer
    takes the gradient through the dropout between them (pattern regenerated, one Philox call per four columns shared by a
    quad of lanes) and leaves the BatchNorm layer's backward sums per 32-row tile, added in bn_bwd_stats_kernel's own order --
    so every gradient of the MLP must EQUAL the product + dropout + statistics launches' (FAIRREC_BN_BWD_SEPARATE=1), with
    the same dropout pattern (call counter restored), for a batch that ends inside a tile and for a full one."""
    from fairrec.model.layers import MLPLayers
    torch.manual_seed(M + len(widths))
    mlp = MLPLayers(widths, dropout=p, activation="leakyrelu", bn=True).cuda().train()
    x = (torch.randn(M, widths[0], device="cuda") * 0.1).requires_grad_()
    tgt = torch.randn(M, widths[-1], device="cuda")
    st = mlp._drop_state(x.device) if p > 0 else None
    st0 = st.clone() if st is not None else None
    res = {}
    try:
        for mode in ("fused", "separate"):
            if mode == "separate":
                os.environ["FAIRREC_BN_BWD_SEPARATE"] = "1"
            else:
                os.environ.pop("FAIRREC_BN_BWD_SEPARATE", None)
            if st is not None:
                st.copy_(st0)
            for q in mlp.parameters():
                q.grad = None
            x.grad = None
            ((mlp(x) - tgt) ** 2).mean().backward()
            res[mode] = {"x": x.grad.clone(), **{n: q.grad.clone() for n, q in mlp.named_parameters()}}
    finally:
        os.environ.pop("FAIRREC_BN_BWD_SEPARATE", None)
    assert float(res["fused"]["x"].abs().max()) > 0
    for n in res["fused"]:
        assert torch.equal(res["fused"][n], res["separate"][n]), n
