"""GPU: dropout drawn on the device with nothing stored (csrc/dropout.hip) and its use by MLPLayers: the properties the
generator must have (keep fraction, determinism, independence of calls / layers / offsets, counter handling) and the
autograd contract (the backward pass sees exactly the pattern of its forward pass).  The arithmetic of dropout -> Linear ->
activation itself is pinned against the reference's goldens with recorded masks in tests/test_mlp_hip.py."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _apply(x, p, seed, off, counter, used=None, tick=None, out=None):
    from fairrec import _C
    out = torch.empty_like(x) if out is None else out
    _C.check(_C.lib().fr_dropout_apply(x.data_ptr(), x.numel(), p, seed, off, counter.data_ptr(), _C.ptr(used), _C.ptr(tick),
                                       out.data_ptr(), _C.current_stream()), "fr_dropout_apply")
    return out


def test_generator_properties():
    dev = "cuda"
    n = 1 << 20
    x = torch.ones(n, device=dev)
    state = torch.zeros(2, dtype=torch.int64, device=dev)
    for p in (0.1, 0.5, 0.0):
        y = _apply(x, p, 1234, 0, state)
        vals = torch.unique(y).cpu().tolist()
        assert vals == ([0.0, pytest.approx(1 / (1 - p))] if p else [1.0])
        keep = (y != 0).float().mean().item()
        assert abs(keep - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-9           # 4 sigma
    a = _apply(x, 0.5, 1234, 0, state)
    assert torch.equal(a, _apply(x, 0.5, 1234, 0, state))                           # a pure function of its arguments
    for other in (_apply(x, 0.5, 1235, 0, state), _apply(x, 0.5, 1234, n, state),
                  _apply(x, 0.5, 1234, 0, torch.ones(2, dtype=torch.int64, device=dev))):
        agree = ((a != 0) == (other != 0)).float().mean().item()
        assert abs(agree - 0.5) < 0.01                                              # ... and unrelated otherwise
    # the pattern is a function of the element's GLOBAL index: a launch at offset o continues the stream
    whole = _apply(x, 0.3, 7, 0, state)
    part = _apply(x[:n // 2], 0.3, 7, n // 2, state)
    assert torch.equal(whole[n // 2:], part)
    # ragged length and in-place
    z = torch.randn(1003, device=dev)[:1001 // 4 * 4 + 1].clone()
    ref = _apply(z, 0.4, 9, 0, state)
    assert torch.equal(_apply(z, 0.4, 9, 0, state, out=z), ref)
    assert int(state[0].item()) == 0 and int(state[1].item()) == 0                  # nobody asked for a tick


def test_counter_ticks_once_per_forward_and_is_recorded():
    dev = "cuda"
    x = torch.ones(300_000, device=dev)
    state = torch.tensor([41, 0], dtype=torch.int64, device=dev)
    used = torch.zeros(1, dtype=torch.int64, device=dev)
    a = _apply(x, 0.5, 5, 0, state, used=used)                      # first launch of a forward: records, does not tick
    b = _apply(x, 0.5, 5, x.numel(), state, tick=state)             # last launch: same counter value, then ticks
    assert used.item() == 41 and state.cpu().tolist() == [42, 0]
    assert torch.equal(a, _apply(x, 0.5, 5, 0, used))               # the backward's view
    assert torch.equal(b, _apply(x, 0.5, 5, x.numel(), used))
    c = _apply(x, 0.5, 5, 0, state, tick=state)                     # next forward: another pattern
    assert state.cpu().tolist() == [43, 0] and not torch.equal(a, c)


@pytest.mark.parametrize("act", ["relu", "tanh"])
@pytest.mark.parametrize("bn", [False, True], ids=["plain", "batchnorm"])
@pytest.mark.parametrize("two_block", [False, True])
def test_mlp_with_device_dropout_equals_torch_with_the_same_patterns(bn, two_block, act):
    """The patterns a forward pass drew are recovered by applying the generator to ones (same seed, counter, offsets); a
    plain torch MLP with exactly those masks must give the same output and the same gradients -- for the layers dropped into
    a copy, the hidden ReLU layers dropped in place (no pattern in the backward pass at all) and the BatchNorm layers."""
    from fairrec.model.layers import MLPLayers
    torch.manual_seed(3)
    dev = "cuda"
    M, widths, p = 256, [64, 32, 32, 1], 0.5
    mlp = MLPLayers(widths, dropout=p, activation=act, bn=bn).to(dev).train()
    x = torch.randn(M, 64, device=dev)
    w_out = torch.randn(M, 1, device=dev)
    state = mlp._drop_state(x.device)
    state[0] = 7
    a = x.clone().requires_grad_(True)
    y = mlp(a[:, :32], a[:, 32:]) if two_block else mlp(a)
    assert state.cpu().tolist() == [8, 0]                          # one tick per forward pass
    (y * w_out).sum().backward()
    got = [a.grad] + [q.grad.clone() for q in mlp.parameters()]
    for q in mlp.parameters():
        q.grad = None

    # the keep-scales of that pass
    ctr = torch.tensor([7, 0], dtype=torch.int64, device=dev)
    off, masks = 0, []
    for l, w in enumerate(widths[:-1]):
        blocks = [32, 32] if (l == 0 and two_block) else [w]
        parts = []
        for k in blocks:
            parts.append(_apply(torch.ones(M, k, device=dev), p, mlp._drop_seed(), off, ctr))
            off += (M * k + 3) // 4 * 4
        masks.append(torch.cat(parts, dim=1))
    b = x.clone().requires_grad_(True)
    h = b
    lins, bns = mlp.linears(), mlp.batchnorms()
    for l, lin in enumerate(lins):
        h = torch.nn.functional.linear(h * masks[l], lin.weight, lin.bias)
        if bn:
            h = torch.nn.functional.batch_norm(h, None, None, bns[l].weight, bns[l].bias, training=True, eps=bns[l].eps)
        h = torch.relu(h) if act == "relu" else torch.tanh(h)
    (h * w_out).sum().backward()
    want = [b.grad] + [q.grad for q in mlp.parameters()]
    np.testing.assert_allclose(y.detach().cpu().numpy(), h.detach().cpu().numpy(), rtol=2e-4, atol=2e-5)
    for g_, w_ in zip(got, want):
        np.testing.assert_allclose(g_.cpu().numpy(), w_.cpu().numpy(), rtol=2e-3,
                                   atol=max(2e-4 * float(w_.abs().max()), 3e-5))   # (a bias in front of BatchNorm: gradient 0 + noise)
    frac = (a.grad == 0).float().mean().item()                      # dropped inputs get no gradient
    assert 0.4 < frac < 0.9

    state[0] = 7                                                    # same counter, same pattern; next counter, another
    y2 = mlp(x[:, :32], x[:, 32:]) if two_block else mlp(x)
    y3 = mlp(x[:, :32], x[:, 32:]) if two_block else mlp(x)
    assert torch.equal(y2, y) and not torch.equal(y3, y)


def test_dropout_inside_a_captured_step_draws_a_new_pattern_per_replay():
    from fairrec.model.layers import MLPLayers
    torch.manual_seed(4)
    dev = "cuda"
    mlp = MLPLayers([64, 64, 1], dropout=0.3).to(dev).train()
    x = torch.randn(512, 64, device=dev)
    with torch.no_grad():
        mlp(x)                                                      # state tensors exist before the capture
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            with torch.no_grad():
                y = mlp(x)
    outs = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        outs.append(y.clone())
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])


def test_copy_many():
    from fairrec import _C
    dev = "cuda"
    g = torch.Generator().manual_seed(1)
    srcs = [torch.randint(0, 255, (n,), generator=g, dtype=torch.uint8).to(dev) for n in (8192 * 8, 8192 * 4, 7, 4097, 1)]
    srcs.append(srcs[0][3:3 + 5001])                                # unaligned source
    dsts = [torch.zeros_like(s) for s in srcs]
    n = len(srcs)
    src = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
    dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
    nb = (ctypes.c_int64 * n)(*[s.numel() for s in srcs])
    _C.check(_C.lib().fr_copy_many(src, dst, nb, n, _C.current_stream()), "fr_copy_many")
    for s, d in zip(srcs, dsts):
        assert torch.equal(s, d)
