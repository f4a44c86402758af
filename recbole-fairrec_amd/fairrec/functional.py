"""Autograd bridges over the HIP scoring / loss kernels (csrc/pfcn.hip, csrc/nfcf.hip): each Function's forward and
backward are single kernel launches through the C ABI; torch only chains them."""
from __future__ import annotations

import torch

from . import _C


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


class RowDot(torch.autograd.Function):
    """torch.mul(a, b).sum(-1) (pfcn_pmf.py:182-183)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
        _C.check(_C.lib().fr_rowdot_fwd(a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], out.data_ptr(),
                                        _C.current_stream()), "fr_rowdot_fwd")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            _C.check(_C.lib().fr_rowdot_bwd(g.data_ptr(), a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], _C.ptr(da),
                                            _C.ptr(db), _C.current_stream()), "fr_rowdot_bwd")
        return da, db


class Bpr(torch.autograd.Function):
    """BPRLoss(pos, neg), loss.py:45-47."""

    @staticmethod
    def forward(ctx, pos, neg):
        pos, neg = pos.contiguous(), neg.contiguous()
        B, dev = pos.numel(), pos.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dpos, dneg = torch.empty_like(pos), torch.empty_like(neg)
        ws = _ws(_C.lib().fr_bpr_workspace_bytes(B, 0), dev)
        _C.check(_C.lib().fr_bpr(pos.data_ptr(), neg.data_ptr(), B, loss.data_ptr(), dpos.data_ptr(), dneg.data_ptr(),
                                 ws.data_ptr(), ws.numel(), _C.current_stream()), "fr_bpr")
        ctx.save_for_backward(dpos, dneg)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dpos, dneg = ctx.saved_tensors
        return dpos * g, dneg * g


class BprBroadcast(torch.autograd.Function):
    """PFCN_BiasedMF's training loss (pfcn_biasedmf.py:192-195): pos/neg scores are `[B] + [B,1]` sums, i.e. a [B,B]
    matrix x_ij = (dp_j - dn_j) + (bpi_i - bni_i); user_bias and global_bias cancel but stay in the graph with an
    exactly-zero gradient, as in the reference (so their optimizer state still steps)."""

    @staticmethod
    def forward(ctx, dp, dn, user_bias, pos_bias, neg_bias, global_bias):
        a = (dp - dn).contiguous()
        c = (pos_bias - neg_bias).reshape(-1).contiguous()
        B, dev = a.numel(), a.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        da, dc = torch.empty_like(a), torch.empty_like(c)
        ws = _ws(_C.lib().fr_bpr_workspace_bytes(B, 1), dev)
        _C.check(_C.lib().fr_bpr_outer(a.data_ptr(), c.data_ptr(), B, loss.data_ptr(), da.data_ptr(), dc.data_ptr(),
                                       ws.data_ptr(), ws.numel(), _C.current_stream()), "fr_bpr_outer")
        ctx.save_for_backward(da, dc)
        ctx.shapes = (user_bias.shape, pos_bias.shape, global_bias.shape)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        da, dc = ctx.saved_tensors
        ub, pb, gb = ctx.shapes
        dev = da.device
        return (da * g, -da * g, torch.zeros(ub, device=dev), (dc * g).reshape(pb), (-dc * g).reshape(pb),
                torch.zeros(gb, device=dev))


class SigmoidBce(torch.autograd.Function):
    """nn.BCELoss()(sigmoid(y), label) (pfcn_biasedmf.py:212-213) -- the BCE leg of fr_nfcf_loss."""

    @staticmethod
    def forward(ctx, y, label):
        lib = _C.lib()
        shape = y.shape
        y = y.contiguous().view(-1)
        label = label.contiguous().view(-1).to(torch.float32)
        B, dev = y.numel(), y.device
        out = torch.empty(B, dtype=torch.float32, device=dev)
        dy = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty(3, dtype=torch.float32, device=dev)
        ws = _ws(lib.fr_nfcf_loss_workspace_bytes(B), dev)
        _C.check(lib.fr_nfcf_loss(y.data_ptr(), label.data_ptr(), None, B, 0.0, None, 0, 1, out.data_ptr(), dy.data_ptr(),
                                  loss.data_ptr(), ws.data_ptr(), ws.numel(), None, _C.current_stream()), "fr_nfcf_loss")
        ctx.save_for_backward(dy)
        ctx.shape = shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dy,) = ctx.saved_tensors
        return (dy * g).view(ctx.shape), None


class SoftmaxCe(torch.autograd.Function):
    """nn.CrossEntropyLoss()(logits, label) (pfcn_biasedmf.py:216)."""

    @staticmethod
    def forward(ctx, logits, label, err_flag):
        logits = logits.contiguous()
        label = label.contiguous().to(torch.int64)
        M, C = logits.shape
        dev = logits.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dl = torch.empty_like(logits)
        ws = _ws(((M + 255) // 256) * 4, dev)
        _C.check(_C.lib().fr_softmax_ce(logits.data_ptr(), label.data_ptr(), M, C, loss.data_ptr(), dl.data_ptr(),
                                        ws.data_ptr(), ws.numel(), _C.ptr(err_flag), _C.current_stream()), "fr_softmax_ce")
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None
