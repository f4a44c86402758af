@pytest.mark.parametrize("widths,p,act,M", [
    ([128, 256, 128], 0.0, "relu", 8192),                       # PFCN's filter
    ([128, 256, 128, 128, 64, 32, 1], 0.3, "leakyrelu", 8192),  # PFCN's discriminator (binary attribute)
    ([64, 32, 7], 0.2, "tanh", 777),                            # a class head, ragged batch
    ([32, 32], 0.0, "sigmoid", 33),
    ([256, 256, 96, 21], 0.5, "relu", 1000),
    ([128, 256, 128], 0.0, "leakyrelu", 200),                   # the filter at the goldens' batch
    ([64, 128, 64, 32], 0.3, "tanh", 4100),
], ids=["filter", "discriminator", "classes7", "tiny", "wide", "filter200", "dropped32"])
@pytest.mark.parametrize("mode", ["train", "frozen", "no_dx"])
def test_bn_chain_equals_the_layered_form(widths, p, act, M, mode, monkeypatch):
    """csrc/mlp_bn.hip (one launch per layer and direction, a layer normalised and dropped by the launch that consumes it)
    against the layered form (the default; the fused one is FAIRREC_BN_FUSED=1) on copies of one module: output, input gradient, every parameter gradient,
    the running statistics and the batch counter -- the same dropout patterns, so the comparison is element for element."""
    import copy
    from fairrec import _C
    from fairrec.model.layers import MLPLayers
    torch.manual_seed(11)
    a = MLPLayers(widths, dropout=p, activation=act, bn=True).cuda().train()
    with torch.no_grad():
        for bn in a.batchnorms():                               # away from the (1, 0) initial values
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
        for lin in a.linears():
            lin.weight.normal_(0, 0.2)
            lin.bias.uniform_(-0.1, 0.1)
    a._drop_seed()
    a._drop_state(torch.device("cuda", torch.cuda.current_device()))
    b = copy.deepcopy(a)
    x = torch.randn(M, widths[0], device="cuda")
    w_out = torch.randn(M, widths[-1], device="cuda")
    lib = _C.lib()
    calls = {"fr_bnl_fwd": 0, "fr_bnl_bwd": 0, "fr_bnl_bwd_top": 0, "fr_bn_fwd_ex": 0, "fr_bn_bwd": 0}
    for name in calls:
        real = getattr(lib, name)
        monkeypatch.setattr(lib, name, (lambda real, name: lambda *args: calls.__setitem__(name, calls[name] + 1) or real(*args))(real, name))
    kw = dict(need_dx=mode != "no_dx", frozen=mode == "frozen")
    monkeypatch.setenv("FAIRREC_BN_FUSED", "1")                   # opt-in: the layered form is the default (it measures faster)
    got = [_bn_mlp_run(a, x, w_out, **kw) for _ in range(2)]      # two passes: the second has another pattern and moved statistics
    L = len(widths) - 1
    assert calls["fr_bnl_fwd"] == 2 * L and calls["fr_bnl_bwd_top"] == 2 and calls["fr_bn_fwd_ex"] == 0 and calls["fr_bn_bwd"] == 0
    assert calls["fr_bnl_bwd"] == 2 * L
    monkeypatch.delenv("FAIRREC_BN_FUSED")
    want = [_bn_mlp_run(b, x, w_out, **kw) for _ in range(2)]
    assert calls["fr_bn_fwd_ex"] == 2 * L
    exact = all(w % 32 == 0 for w in widths)
    for g_, w_ in zip(got, want):
        assert g_.keys() == w_.keys()
        for k in g_:
            if exact and w_[k] is not None:
                # every width a multiple of 32: both forms apply to every layer, and they are the same operations in the same
                # order (csrc/mlp_bn_math.hpp, the products' parts, the statistics' chunks) -- bit for bit, forward and backward
                assert torch.equal(g_[k], w_[k]), (k, float((g_[k].float() - w_[k].float()).abs().max()))
                continue
            if w_[k] is None:
                assert g_[k] is None, k
                continue
            if k.endswith("num_batches_tracked"):
                assert torch.equal(g_[k], w_[k]), k
                continue
            scale = max(float(w_[k].abs().max()), 1e-6)
            wk = k[:-len("bias")] + "weight"
            if k.startswith("g.") and k.endswith(".bias") and w_.get(wk) is not None and w_[wk].dim() == 2:
                # a bias in front of BatchNorm: gradient 0 + rounding noise, in both forms
                assert float((g_[k] - w_[k]).abs().max()) <= 1e-4 * float(w_[wk].abs().max()), k
                continue
            # element for element, except where a pre-activation sits on the kink of relu / leakyrelu: the two forms sum a
            # product in different orders (last-bit differences in Z), a y of +-1e-8 then takes the other slope, and the
            # gradient of that one element -- and, diluted by the batch, of what is summed over it -- differs
            bad = ~torch.isclose(g_[k], w_[k], rtol=2e-4, atol=2e-5 * scale)
            frac = bad.float().mean().item()
            rel = float((g_[k] - w_[k]).norm() / max(float(w_[k].norm()), 1e-12))
            kinked = act in ("relu", "leakyrelu") and k != "y" and not k.startswith("b.")
            n_bad = int(bad.sum())
            assert (n_bad <= (max(2, 5e-4 * bad.numel()) if kinked else 0) and rel <= (2e-3 if kinked else 1e-4)) or scale <= 1e-6, \
                (k, n_bad, bad.numel(), rel)
    if p > 0:
        assert not torch.equal(got[0]["y"], got[1]["y"])


@pytest.mark.parametrize("widths,act,M", [([128, 256, 128], "relu", 4096), ([64, 64, 32, 3], "tanh", 500), ([32, 1], "leakyrelu", 64)])
def test_bn_chain_matches_torch_in_float64(widths, act, M, monkeypatch):
    """The fused BatchNorm MLP against torch in float64 (no dropout): output, gradients, running statistics."""
    from fairrec.model.layers import MLPLayers
    monkeypatch.setenv("FAIRREC_BN_FUSED", "1")
    torch.manual_seed(5)
    mlp = MLPLayers(widths, activation=act, bn=True).cuda().train()
    with torch.no_grad():
        for bn in mlp.batchnorms():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
        for lin in mlp.linears():
            lin.weight.normal_(0, 0.3)
    x = torch.randn(M, widths[0], device="cuda")
    w_out = torch.randn(M, widths[-1], device="cuda")
    got = _bn_mlp_run(mlp, x, w_out)
    xd = x.double().requires_grad_()
    h = xd
    ps = {n: p.detach().double().requires_grad_() for n, p in mlp.named_parameters()}
    lins = [n[:-len(".weight")] for n, p in mlp.named_parameters() if p.dim() == 2]
    fn = {"relu": torch.relu, "tanh": torch.tanh, "leakyrelu": torch.nn.functional.leaky_relu}[act]
    for l, ln in enumerate(lins):
        bn_name = ln.rsplit(".", 1)[0] + "." + str(int(ln.rsplit(".", 1)[1]) + 1)
        h = torch.nn.functional.linear(h, ps[ln + ".weight"], ps[ln + ".bias"])
        h = torch.nn.functional.batch_norm(h, None, None, ps[bn_name + ".weight"], ps[bn_name + ".bias"], training=True, eps=1e-5)
        h = fn(h)
    (h * w_out.double()).sum().backward()
    torch.testing.assert_close(got["y"].double(), h.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got["dx"].double(), xd.grad, rtol=1e-3, atol=1e-5 * max(1.0, float(xd.grad.abs().max())))
    for n, p in ps.items():
        if n.endswith(".bias") and n[:-len(".bias")] in lins:
            continue                                             # a bias in front of BatchNorm: gradient 0 + rounding noise
        torch.testing.assert_close(got["g." + n].double(), p.grad, rtol=1e-3, atol=2e-5 * max(1.0, float(p.grad.abs().max())),
                                   msg=lambda m: f"{n}: {m}")


