"""Autograd bridges over the HIP scoring / loss kernels (csrc/pfcn.hip, csrc/nfcf.hip): each Function's forward and
backward are single kernel launches through the C ABI; torch only chains them."""
from __future__ import annotations

import torch

from . import _C


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


class SplitRows(torch.autograd.Function):
    """(x[:n], x[n:]) of rows that ONE lookup gathered for two id lists.  Autograd's own slices give each half a backward of
    its own -- a zero-filled [2B, D] buffer, a copy into it, and an add of the two buffers: five launches where the gradient
    is simply the two halves side by side (one `cat`; the same values, x + 0 = x)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.shape = int(n), x.shape
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        ref = ga if ga is not None else gb
        if ga is None:
            ga = ref.new_zeros((ctx.n,) + tuple(ctx.shape[1:]))
        if gb is None:
            gb = ref.new_zeros((ctx.shape[0] - ctx.n,) + tuple(ctx.shape[1:]))
        return torch.cat([ga, gb]), None


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


class RowDotRep(torch.autograd.Function):
    """RowDot with the rows of `a` [A, D] reused by the R = b.shape[0] / A row blocks of `b`: out[r*A + i] = a[i] . b[r*A + i]
    (a user row against its positive and its negative item row, which one lookup gathered as [2B, D])."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        A, D = a.shape
        reps = b.shape[0] // A
        assert b.shape[0] == reps * A and b.shape[1] == D
        out = torch.empty(b.shape[0], dtype=torch.float32, device=a.device)
        _C.check(_C.lib().fr_rowdot_rep_fwd(a.data_ptr(), b.data_ptr(), A, reps, D, out.data_ptr(), _C.current_stream()),
                 "fr_rowdot_rep_fwd")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            _C.check(_C.lib().fr_rowdot_rep_bwd(g.data_ptr(), a.data_ptr(), b.data_ptr(), a.shape[0], b.shape[0] // a.shape[0],
                                                a.shape[1], _C.ptr(da), _C.ptr(db), _C.current_stream()), "fr_rowdot_rep_bwd")
        return da, db


class RowDotPair(torch.autograd.Function):
    """cat(RowDot(a, b[:A]), RowDot(a, b[A:])) for the rows `b` [2A, D] that one lookup gathered for a positive and a negative
    id list: ONE forward launch (fr_rowdot_rep_fwd) instead of two and a `cat`, ONE backward launch (fr_rowdot_rep_bwd_sep: the
    products RowDot's two backward launches form, the gradient of `b` whole instead of two halves glued by autograd).  `a` is
    passed TWICE -- `RowDotPair.apply(a, a, b)` -- so that its two gradients reach autograd unsummed and are added to
    whatever else `a` feeds in the order the two RowDot nodes gave them (the negative's first: the later node runs first);
    summing them here would round once differently (the d128 golden)."""

    @staticmethod
    def forward(ctx, a_neg, a_pos, b):
        a, b = a_pos.contiguous(), b.contiguous()
        A, D = a.shape
        assert a_neg.data_ptr() == a_pos.data_ptr() and b.shape[0] == 2 * A and b.shape[1] == D
        out = torch.empty(2 * A, dtype=torch.float32, device=a.device)
        _C.check(_C.lib().fr_rowdot_rep_fwd(a.data_ptr(), b.data_ptr(), A, 2, D, out.data_ptr(), _C.current_stream()),
                 "fr_rowdot_rep_fwd")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        A, D = a.shape
        want_a = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        da = torch.empty_like(b) if want_a else None        # [pos part; neg part], unsummed
        db = torch.empty_like(b) if ctx.needs_input_grad[2] else None
        if da is not None or db is not None:
            _C.check(_C.lib().fr_rowdot_rep_bwd_sep(g.data_ptr(), a.data_ptr(), b.data_ptr(), A, 2, D, _C.ptr(da), _C.ptr(db),
                                                    _C.current_stream()), "fr_rowdot_rep_bwd_sep")
        da_neg = da[A:] if ctx.needs_input_grad[0] else None
        da_pos = da[:A] if ctx.needs_input_grad[1] else None
        return da_neg, da_pos, db


class SubScaled(torch.autograd.Function):
    """a - alpha * b of two scalar losses: torch.sub's backward negates and scales in two launches; here one."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        ctx.alpha = float(alpha)
        return torch.sub(a, b, alpha=alpha)

    @staticmethod
    def backward(ctx, g):
        return (g if ctx.needs_input_grad[0] else None), (g * (-ctx.alpha) if ctx.needs_input_grad[1] else None), None


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
        if _C.is_one(g):            # GraphedStep's backward seed: nothing to scale by
            return dpos, dneg
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


class BprBroadcastPacked(torch.autograd.Function):
    """BprBroadcast on packed columns: `scores` = [pos | neg] ([2B], RowDotRep's output), `item_bias` = the [2B, 1] lookup of
    [pos items | neg items].  The differences are formed in the kernel and the four gradient columns come out of it, so the
    loss costs no elementwise launches (the unpacked form: 2 subtractions before, 6 products / negations and 2 slice
    gradients after)."""

    @staticmethod
    def forward(ctx, scores, user_bias, item_bias, global_bias):
        scores = scores.contiguous()
        ib = item_bias.contiguous()
        B, dev = scores.numel() // 2, scores.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ds = torch.empty_like(scores)
        dib = torch.empty(2 * B, dtype=torch.float32, device=dev)
        ws = _ws(_C.lib().fr_bpr_workspace_bytes(B, 1), dev)
        sp, bp = scores.data_ptr(), ib.data_ptr()
        _C.check(_C.lib().fr_bpr_outer2(sp, sp + 4 * B, bp, bp + 4 * B, B, loss.data_ptr(), ds.data_ptr(), ds.data_ptr() + 4 * B,
                                        dib.data_ptr(), dib.data_ptr() + 4 * B, ws.data_ptr(), ws.numel(),
                                        _C.current_stream()), "fr_bpr_outer2")
        ctx.save_for_backward(ds, dib)
        ctx.shapes = (user_bias.shape, item_bias.shape, global_bias.shape)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        ds, dib = ctx.saved_tensors
        ub, ibs, gb = ctx.shapes
        dev = ds.device
        if not _C.is_one(g):
            ds, dib = ds * g, dib * g
        return ds, _C.zeros_cached(ub, dev), dib.reshape(ibs), _C.zeros_cached(gb, dev)


class BprBroadcastGlobal(torch.autograd.Function):
    """BprBroadcastPacked on row-sharded tables: the [B] + [B, 1] broadcast runs over the GLOBAL batch (every rank's rows and
    columns: ShardedGenericEngine.global_bpr_broadcast), so a G-rank step is the single-device step on the concatenated
    batch."""

    @staticmethod
    def forward(ctx, scores, user_bias, item_bias, global_bias, engine):
        B = scores.numel() // 2
        ib = item_bias.reshape(-1)
        a = (scores[:B] - scores[B:]).contiguous()
        c = (ib[:B] - ib[B:]).contiguous()
        loss, da, dc = engine.global_bpr_broadcast(a, c)
        ctx.save_for_backward(da, dc)
        ctx.shapes = (user_bias.shape, item_bias.shape, global_bias.shape)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        da, dc = ctx.saved_tensors
        ub, ibs, gb = ctx.shapes
        dev = da.device
        if not _C.is_one(g):
            da, dc = da * g, dc * g
        return (torch.cat([da, -da]), torch.zeros(ub, device=dev), torch.cat([dc, -dc]).reshape(ibs), torch.zeros(gb, device=dev),
                None)


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
        return (dy if _C.is_one(g) else dy * g).view(ctx.shape), None


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
        return (dl if _C.is_one(g) else dl * g), None, None


class Mse(torch.autograd.Function):
    """nn.MSELoss()(pred, target) (fairgo_pmf.py:182)."""

    @staticmethod
    def forward(ctx, pred, target):
        pred, target = pred.contiguous(), target.contiguous().to(torch.float32)
        B, dev = pred.numel(), pred.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dp = torch.empty_like(pred)
        ws = _ws(((B + 255) // 256) * 4, dev)
        _C.check(_C.lib().fr_mse(pred.data_ptr(), target.data_ptr(), B, loss.data_ptr(), dp.data_ptr(), ws.data_ptr(),
                                 ws.numel(), _C.current_stream()), "fr_mse")
        ctx.save_for_backward(dp)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return (dp if _C.is_one(g) else dp * g), None


class CsrMatrix:
    """A sparse matrix and its transpose as device CSR arrays (indptr int64, col int32, val fp32)."""

    def __init__(self, scipy_csr, device):
        import numpy as np
        self.shape = scipy_csr.shape
        self.fwd = self._pack(scipy_csr, device)
        self.bwd = self._pack(scipy_csr.transpose().tocsr(), device)

    @staticmethod
    def _pack(m, device):
        import numpy as np
        m.sort_indices()
        return (torch.from_numpy(m.indptr.astype(np.int64)).to(device), torch.from_numpy(m.indices.astype(np.int32)).to(device),
                torch.from_numpy(m.data.astype(np.float32)).to(device))


class SpMM(torch.autograd.Function):
    """Y = L X (torch.sparse.mm, fairgo_pmf.py:198); backward dX = L^T dY with the pre-built transpose."""

    @staticmethod
    def forward(ctx, X, L: CsrMatrix):
        X = X.contiguous()
        Y = torch.empty((L.shape[0], X.shape[1]), dtype=torch.float32, device=X.device)
        ip, col, val = L.fwd
        _C.check(_C.lib().fr_spmm_csr(ip.data_ptr(), col.data_ptr(), val.data_ptr(), X.data_ptr(), L.shape[0], X.shape[1],
                                      Y.data_ptr(), _C.current_stream()), "fr_spmm_csr")
        ctx.L = L
        return Y

    @staticmethod
    def backward(ctx, dY):
        L = ctx.L
        dY = dY.contiguous()
        dX = torch.empty((L.shape[1], dY.shape[1]), dtype=torch.float32, device=dY.device)
        ip, col, val = L.bwd
        _C.check(_C.lib().fr_spmm_csr(ip.data_ptr(), col.data_ptr(), val.data_ptr(), dY.data_ptr(), L.shape[1], dY.shape[1],
                                      dX.data_ptr(), _C.current_stream()), "fr_spmm_csr")
        return dX, None


class SpMMSel(torch.autograd.Function):
    """Rows of Y = L X where the batch can see them (fr_spmm_csr_sel): Y[i] = sum_j L[rows[i], j] * X[xmap[j]] with the
    nonzeros in CSR order and nonzeros whose `xmap` entry is -1 skipped.  `rows` / `rpos` (int32 [R] / its inverse map over
    the rows of L, -1 elsewhere) select the output rows; `xrows` / `xmap` (int32 [n_x] / inverse over the columns) say which
    columns the rows of a COMPACT X stand for; either pair may be None (all rows / X is the whole table).  The backward is the
    same kernel on the CSR of L^T with the two pairs swapped.  Kept terms are added in SpMM's order and skipped terms are the
    ones SpMM adds as exact zeros, so values and gradients are those of the whole-table product restricted to the rows."""

    @staticmethod
    def forward(ctx, X, L: CsrMatrix, rows, rpos, xrows, xmap, rbits=None, xbits=None):
        """`rbits` / `xbits`: the bitmaps of `rpos` / `xmap` (bit c set where the map is >= 0; fr_spmm_csr_sel)."""
        X = X.contiguous()
        n_out = L.shape[0] if rows is None else rows.numel()
        Y = torch.empty((n_out, X.shape[1]), dtype=torch.float32, device=X.device)
        ip, col, val = L.fwd
        _C.check(_C.lib().fr_spmm_csr_sel(ip.data_ptr(), col.data_ptr(), val.data_ptr(), X.data_ptr(), _C.ptr(rows), n_out,
                                          _C.ptr(xmap), _C.ptr(xbits), X.shape[1], Y.data_ptr(), _C.current_stream()),
                 "fr_spmm_csr_sel")
        ctx.L, ctx.sel, ctx.n_x = L, (rows, rpos, xrows, xmap, rbits), X.shape[0]
        return Y

    @staticmethod
    def backward(ctx, dY):
        L = ctx.L
        rows, rpos, xrows, xmap, rbits = ctx.sel
        dY = dY.contiguous()
        dX = torch.empty((ctx.n_x, dY.shape[1]), dtype=torch.float32, device=dY.device)
        ip, col, val = L.bwd
        _C.check(_C.lib().fr_spmm_csr_sel(ip.data_ptr(), col.data_ptr(), val.data_ptr(), dY.data_ptr(), _C.ptr(xrows), ctx.n_x,
                                          _C.ptr(rpos), _C.ptr(rbits), dY.shape[1], dX.data_ptr(), _C.current_stream()),
                 "fr_spmm_csr_sel")
        return dX, None, None, None, None, None, None, None


class GatherAndSpMMSel(torch.autograd.Function):
    """(X[idx], SpMMSel(X, L, rows, rpos)) of ONE whole-table operand as one autograd node: the two uses of X meet in the
    backward as ONE dense [N, D] gradient -- fr_spmm_csr_sel writes it (every row: zeros where the batch sees nothing), then
    fr_row_scatter_add adds the gathered rows' gradients, duplicates summed in ascending position -- where two nodes cost a
    zero fill, a second dense tensor and autograd's addition of the two (three more whole-table passes).  The same two
    addends per element: the same bits."""

    @staticmethod
    def forward(ctx, X, idx, err_flag, L: CsrMatrix, rows, rpos, rbits=None, act=0):
        """`act` != 0: X is the output of that activation (an MLP built with `grad_at_z=True`) and the backward returns the
        gradient at the activation's INPUT -- (L^T dY + scatter(g_rows)) o act'(X) -- from the same two launches
        (fr_spmm_csr_sel_act / fr_row_scatter_add_act) instead of a third whole-table pass in the MLP's backward."""
        X = X.contiguous()
        idx = idx.contiguous().to(torch.int64)
        M, (N, D) = idx.numel(), X.shape
        out = torch.empty((M, D), dtype=torch.float32, device=X.device)
        _C.check(_C.lib().fr_row_gather(X.data_ptr(), idx.data_ptr(), M, N, D, out.data_ptr(), _C.ptr(err_flag),
                                        _C.current_stream()), "fr_row_gather")
        Y = torch.empty((rows.numel(), D), dtype=torch.float32, device=X.device)
        ip, col, val = L.fwd
        _C.check(_C.lib().fr_spmm_csr_sel(ip.data_ptr(), col.data_ptr(), val.data_ptr(), X.data_ptr(), rows.data_ptr(),
                                          rows.numel(), None, None, D, Y.data_ptr(), _C.current_stream()), "fr_spmm_csr_sel")
        if act:
            ctx.save_for_backward(idx, X)
        else:
            ctx.save_for_backward(idx)
        ctx.meta = (N, D, err_flag, L, rpos, rbits, int(act))
        return out, Y

    @staticmethod
    def backward(ctx, g_rows, dY):
        idx = ctx.saved_tensors[0]
        N, D, err, L, rpos, rbits, act = ctx.meta
        lib, st = _C.lib(), _C.current_stream()
        dX = torch.empty((N, D), dtype=torch.float32, device=idx.device)
        ip, col, val = L.bwd
        if act and (dY is None or g_rows is None):      # (not FairGo's step: both uses carry a gradient there)
            raise _C.FairrecError("GatherAndSpMMSel(act=...): both outputs must receive a gradient")
        if act:
            X = ctx.saved_tensors[1]
            M = idx.numel()
            # the rows the scatter adds to are scaled by IT, after the addition: a bitmap of them for the product to leave alone
            skip = torch.zeros((N + 31) // 32, dtype=torch.int32, device=idx.device)
            _C.check(lib.fr_frontier_mark(idx.data_ptr(), M, N, skip.data_ptr(), _C.ptr(err), st), "fr_frontier_mark")
            dY = dY.contiguous()
            _C.check(lib.fr_spmm_csr_sel_act(ip.data_ptr(), col.data_ptr(), val.data_ptr(), dY.data_ptr(), None, N, rpos.data_ptr(),
                                             _C.ptr(rbits), D, dX.data_ptr(), X.data_ptr(), act, skip.data_ptr(), st),
                     "fr_spmm_csr_sel_act")
            g_rows = g_rows.contiguous()
            ws = _ws(lib.fr_row_scatter_workspace_bytes(M), dX.device)
            _C.check(lib.fr_row_scatter_add_act(g_rows.data_ptr(), idx.data_ptr(), M, N, D, dX.data_ptr(), ws.data_ptr(), ws.numel(),
                                                X.data_ptr(), act, _C.ptr(err), st), "fr_row_scatter_add_act")
            return dX, None, None, None, None, None, None, None
        if dY is None:
            dX.zero_()
        else:
            dY = dY.contiguous()
            _C.check(lib.fr_spmm_csr_sel(ip.data_ptr(), col.data_ptr(), val.data_ptr(), dY.data_ptr(), None, N,
                                         rpos.data_ptr(), _C.ptr(rbits), D, dX.data_ptr(), st), "fr_spmm_csr_sel")
        if g_rows is not None:
            g_rows = g_rows.contiguous()
            M = idx.numel()
            ws = _ws(lib.fr_row_scatter_workspace_bytes(M), dX.device)
            _C.check(lib.fr_row_scatter_add(g_rows.data_ptr(), idx.data_ptr(), M, N, D, dX.data_ptr(), ws.data_ptr(),
                                            ws.numel(), _C.ptr(err), st), "fr_row_scatter_add")
        return dX, None, None, None, None, None, None, None


class RowGather(torch.autograd.Function):
    """X[idx] on a whole-table activation (fairgo_pmf.py:178-179); backward = dense gradient with duplicates summed in
    ascending batch position (fixed order, no atomics)."""

    @staticmethod
    def forward(ctx, X, idx, err_flag):
        X = X.contiguous()
        idx = idx.contiguous().to(torch.int64)
        M, (N, D) = idx.numel(), X.shape
        out = torch.empty((M, D), dtype=torch.float32, device=X.device)
        _C.check(_C.lib().fr_row_gather(X.data_ptr(), idx.data_ptr(), M, N, D, out.data_ptr(), _C.ptr(err_flag),
                                        _C.current_stream()), "fr_row_gather")
        ctx.save_for_backward(idx)
        ctx.shape, ctx.err = (N, D), err_flag
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        N, D = ctx.shape
        g = g.contiguous()
        M = idx.numel()
        dX = torch.empty((N, D), dtype=torch.float32, device=g.device)
        ws = _ws(_C.lib().fr_row_scatter_workspace_bytes(M), g.device)
        _C.check(_C.lib().fr_row_scatter_sum(g.data_ptr(), idx.data_ptr(), M, N, D, dX.data_ptr(), ws.data_ptr(), ws.numel(),
                                             _C.ptr(ctx.err), _C.current_stream()), "fr_row_scatter_sum")
        return dX, None, None
