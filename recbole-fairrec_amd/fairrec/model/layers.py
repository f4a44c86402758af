"""`MLPLayers` with the constructor, module structure and parameter names of recbole/model/layers.py:30-85
(per layer: Dropout -> Linear -> [BatchNorm1d] -> activation, the last layer included), so state_dict keys
(`mlp_layers.<3l+1>.weight` ...) interchange with the reference.  The arithmetic runs on the fp32-MFMA kernels of
csrc/mlp.hip through one autograd Function; there is no torch fallback for CUDA tensors and no CPU path.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
from torch.nn.init import normal_

from .. import _C

ACT_CODES = {None: 0, "none": 0, "relu": 1, "leakyrelu": 2, "sigmoid": 3, "tanh": 4}


def activation_layer(activation_name='relu', emb_dim=None):
    """Same name -> module mapping as the reference's activation_layer (layers.py:88-118)."""
    if activation_name is None:
        return None
    name = activation_name.lower()
    if name == 'sigmoid':
        return nn.Sigmoid()
    if name == 'tanh':
        return nn.Tanh()
    if name == 'relu':
        return nn.ReLU()
    if name == 'leakyrelu':
        return nn.LeakyReLU()
    if name == 'none':
        return None
    raise NotImplementedError("activation function {} is not implemented".format(activation_name))


class _Drop:
    """Dropout of one MLP forward pass drawn by csrc/dropout.hip: nothing stored, the pattern is a function of
    (seed, call counter, element offset).  `state` = device int64 {counter, ticket}; the LAST apply of a forward pass
    advances the counter on the device (replay-safe inside a hipGraph); `used` records the value for the backward pass."""

    def __init__(self, seed: int, state: torch.Tensor, p: float, n_launches: int):
        self.seed, self.state, self.p, self.left = seed & (2 ** 64 - 1), state, p, n_launches
        self.used = torch.empty(1, dtype=torch.int64, device=state.device)
        self.off = 0
        self.first = True

    def take(self, n: int):
        """Books one launch that drops n elements: (offset, `used` pointer or None, tick-state pointer or None)."""
        off = self.off
        self.left -= 1
        used = self.used.data_ptr() if self.first else None
        tick = self.state.data_ptr() if self.left == 0 else None
        self.first = False
        self.off += (n + 3) // 4 * 4
        return off, used, tick

    def forward(self, x: torch.Tensor, out: torch.Tensor) -> int:
        """out = dropout(x); returns the element offset of this launch (what `again` needs)."""
        off, used, tick = self.take(x.numel())
        _C.check(_C.lib().fr_dropout_apply(x.data_ptr(), x.numel(), self.p, self.seed, off, self.state.data_ptr(), used, tick,
                                           out.data_ptr(), _C.current_stream()), "fr_dropout_apply")
        return off

    def forward2(self, x0, out0, x1, out1):
        """The two blocks of a first layer's input in one launch; returns their offsets."""
        o0, o1 = self.off, self.off + (x0.numel() + 3) // 4 * 4
        self.left -= 2
        _C.check(_C.lib().fr_dropout_apply2(x0.data_ptr(), x0.numel(), o0, out0.data_ptr(), x1.data_ptr(), x1.numel(), o1,
                                            out1.data_ptr(), self.p, self.seed, self.state.data_ptr(),
                                            self.used.data_ptr() if self.first else None,
                                            self.state.data_ptr() if self.left == 0 else None, _C.current_stream()),
                 "fr_dropout_apply2")
        self.first = False
        self.off = o1 + (x1.numel() + 3) // 4 * 4
        return o0, o1

    def again2(self, g0, off0, g1, off1):
        _C.check(_C.lib().fr_dropout_apply2(g0.data_ptr(), g0.numel(), off0, g0.data_ptr(), g1.data_ptr(), g1.numel(), off1,
                                            g1.data_ptr(), self.p, self.seed, self.used.data_ptr(), None, None,
                                            _C.current_stream()), "fr_dropout_apply2")

    def again(self, g: torch.Tensor, off: int):
        """g *= the keep pattern the forward launch at `off` drew (in place)."""
        _C.check(_C.lib().fr_dropout_apply(g.data_ptr(), g.numel(), self.p, self.seed, off, self.used.data_ptr(), None, None,
                                           g.data_ptr(), _C.current_stream()), "fr_dropout_apply")


class _HipMLP(torch.autograd.Function):
    """y = MLP([x0 | x1]) on the HIP kernels: per layer [dropout ->] fr_linear_fwd [-> fr_bn_fwd] with the activation fused
    into the last of the two; saves the post-activation outputs and the BatchNorm statistics.

    Dropout comes in two forms.  `masks` (tests: recorded patterns as fp32 keep-scales) multiply a layer's input, or go to
    the general kernels as bytes when the input's width does not suit the fast GEMMs.  `drop` (training) draws the pattern
    on the device with nothing stored: the input of the first layer is dropped into a copy and its gradient through the
    regenerated pattern; a hidden layer's input is the previous layer's OUTPUT, dropped in place when that layer ends in a
    plain ReLU -- the backward pass then needs neither the pattern nor the undropped output (fr_act_bwd_dropped)."""

    @staticmethod
    def forward(ctx, x0, x1, act, p_drop, masks, drop, bn_buffers, *params):
        lib = _C.lib()
        st = _C.current_stream()
        # bit 8 of `act` (MLPLayers.forward(grad_at_z=True)): the ONE consumer of the output hands back the gradient at the top
        # layer's pre-activation, i.e. already through act' (functional.GatherAndSpMMSel does, inside its own launches)
        top_at_z, act = bool(act & 0x100), act & 0xff
        use_bn = bn_buffers is not None
        if top_at_z and (use_bn or act == 0 or p_drop > 0.0):
            raise _C.FairrecError("grad_at_z: a plain activation on the top layer, no BatchNorm, no dropout")
        per = 4 if use_bn else 2
        n_layers = len(params) // per
        M = x0.shape[0]
        dev = x0.device
        x0 = x0.contiguous()
        x1 = x1.contiguous() if x1 is not None else None
        scale = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
        cur = (x0, x1)
        outs, xhats, invstds = [], [], []
        premul, ins, masks8 = [], [], []
        fused_in = None                       # (dropped input of the next layer, its offset) written by fr_bn_fwd_drop
        drop_off = [None] * n_layers          # (offset of block a, offset of block c) of a regenerated pattern
        dropped_out = [False] * n_layers      # layer l's output was dropped in place for layer l + 1
        def product(a, k0, c, k1, W, b, N, Y, bn_ws):
            """Y = [a | c] W^T + b (+ the layer's activation where no BatchNorm follows); with BatchNorm behind the layer, its
            per-chunk statistics come out of the product's epilogue when the shapes allow (returns 1: fr_bn_fwd_ex skips its
            statistics launch)."""
            if bn_ws is not None:
                rc = lib.fr_linear_fwd_bnstats(a.data_ptr(), k0, _C.ptr(c), k1, W.data_ptr(), b.data_ptr(), M, N, Y.data_ptr(),
                                               bn_ws.data_ptr(), bn_ws.numel(), st)
                if rc == 0:
                    return 1
                if rc != _C.EUNSUPPORTED:
                    _C.check(rc, "fr_linear_fwd_bnstats")
            _C.check(lib.fr_linear_fwd(a.data_ptr(), k0, _C.ptr(c), k1, None, 1.0, W.data_ptr(), b.data_ptr(), M, N,
                                       0 if use_bn else act, Y.data_ptr(), st), "fr_linear_fwd")
            return 0

        for l in range(n_layers):
            W, b = params[per * l].contiguous(), params[per * l + 1].contiguous()
            a, c = cur
            k0, k1 = a.shape[1], (c.shape[1] if c is not None else 0)
            N = W.shape[0]
            Y = torch.empty((M, N), dtype=torch.float32, device=dev)
            bn_ws = torch.empty(lib.fr_bn_workspace_bytes(M, N), dtype=torch.uint8, device=dev) if use_bn else None
            have_stats = 0
            mk = masks[l] if masks is not None else None      # fp32 keep scales [M, K] (0 or 1/(1-p)), or None
            mk8 = None
            if drop is not None and fused_in is not None:
                # the BatchNorm pass of the layer below already wrote this layer's dropped input
                a, c = fused_in[0], None
                drop_off[l] = (fused_in[1], 0)
                fused_in = None
                premul.append(True)
                have_stats = product(a, k0, None, 0, W, b, N, Y, bn_ws)
            elif drop is not None:
                if l > 0 and not use_bn and act == 1 and a.numel() % 4 == 0:
                    drop.forward(a, a)                          # a IS outs[l - 1]
                    dropped_out[l - 1] = True
                    premul.append(False)
                else:
                    ad = torch.empty_like(a)
                    cd, oc = None, 0
                    if c is not None:
                        cd = torch.empty_like(c)
                        oa, oc = drop.forward2(a, ad, c, cd)
                    else:
                        oa = drop.forward(a, ad)
                    a, c, drop_off[l] = ad, cd, (oa, oc)
                    premul.append(True)
                have_stats = product(a, k0, c, k1, W, b, N, Y, bn_ws)
            elif mk is not None and k0 % 32 == 0 and k1 % 32 == 0:
                # the layer's input with dropout applied (both blocks of a two-block input); kept for backward
                a = a * (mk if c is None else mk[:, :k0])
                c = c * mk[:, k0:] if c is not None else None
                premul.append(True)
                have_stats = product(a, k0, c, k1, W, b, N, Y, bn_ws)
            else:
                premul.append(False)
                mk8 = (mk != 0).to(torch.uint8) if mk is not None else None
                if mk8 is None:
                    have_stats = product(a, k0, c, k1, W, b, N, Y, bn_ws)
                else:
                    _C.check(lib.fr_linear_fwd(a.data_ptr(), k0, _C.ptr(c), k1, _C.ptr(mk8), scale, W.data_ptr(), b.data_ptr(), M,
                                               N, 0 if use_bn else act, Y.data_ptr(), st), "fr_linear_fwd")
            ins.append((a, c) if premul[-1] else None)
            masks8.append(mk8)
            if use_bn:
                g, be = params[per * l + 2].contiguous(), params[per * l + 3].contiguous()
                rm, rv, eps, mom, nbt, n_pass = bn_buffers[l]
                Z = Y
                Y = torch.empty_like(Z)
                xh = torch.empty_like(Z)
                inv = torch.empty(N, dtype=torch.float32, device=dev)
                ws = bn_ws
                if drop is not None and l + 1 < n_layers and N % 4 == 0:
                    # ... and the next layer's dropout in the same pass (Yd next to Y, which the backward pass needs)
                    Yd = torch.empty_like(Z)
                    off, used, tick = drop.take(M * N)
                    _C.check(lib.fr_bn_fwd_ex(Z.data_ptr(), g.data_ptr(), be.data_ptr(), eps, mom, _C.ptr(rm), _C.ptr(rv), M, N,
                                              act, Y.data_ptr(), xh.data_ptr(), inv.data_ptr(), ws.data_ptr(), ws.numel(),
                                              have_stats, Yd.data_ptr(), drop.p, drop.seed, off, drop.state.data_ptr(), used, tick,
                                              _C.ptr(nbt), n_pass, st), "fr_bn_fwd_ex")
                    fused_in = (Yd, off)
                else:
                    _C.check(lib.fr_bn_fwd_ex(Z.data_ptr(), g.data_ptr(), be.data_ptr(), eps, mom, _C.ptr(rm), _C.ptr(rv), M, N, act,
                                              Y.data_ptr(), xh.data_ptr(), inv.data_ptr(), ws.data_ptr(), ws.numel(), have_stats,
                                              None, 0.0, 0, 0, None, None, None, _C.ptr(nbt), n_pass, st), "fr_bn_fwd_ex")
                xhats.append(xh)
                invstds.append(inv)
            outs.append(Y)
            cur = (Y, None)
        ctx.act, ctx.scale, ctx.masks, ctx.n_layers, ctx.use_bn = act, scale, masks, n_layers, use_bn
        ctx.premul, ctx.ins, ctx.masks8 = premul, ins, masks8
        ctx.drop, ctx.drop_off, ctx.dropped_out = drop, drop_off, dropped_out
        ctx.has_x1 = x1 is not None
        ctx.top_at_z = top_at_z
        ctx.n_params = len(params)
        ctx.save_for_backward(x0, *([x1] if x1 is not None else []), *params, *outs, *xhats, *invstds)
        return outs[-1]

    @staticmethod
    def backward(ctx, dY):
        lib = _C.lib()
        st = _C.current_stream()
        saved = list(ctx.saved_tensors)
        x0 = saved.pop(0)
        x1 = saved.pop(0) if ctx.has_x1 else None
        L, per = ctx.n_layers, (4 if ctx.use_bn else 2)
        params = saved[:ctx.n_params]
        rest = saved[ctx.n_params:]
        outs, xhats, invstds = rest[:L], rest[L:2 * L], rest[2 * L:3 * L]
        M = x0.shape[0]
        dev = x0.device
        grads: List[Optional[torch.Tensor]] = [None] * ctx.n_params
        dY = dY.contiguous()
        dx0 = dx1 = None
        drop = ctx.drop
        at_z = ctx.top_at_z   # dY already is the gradient at this layer's pre-activation (folded into the layer above / the consumer)
        # A BatchNorm layer's backward statistics may come out of the epilogue of the product above it, which then also takes
        # the gradient through the dropout between the two layers (fr_linear_bwd_input_bnstats: one launch for product,
        # dropout and statistics; FAIRREC_BN_BWD_SEPARATE=1: the three launches).  `bn_stats`: the workspace they are in.
        bn_stats = None
        bn_fuse = (ctx.use_bn and M <= 32768 and ctx.masks is None and os.environ.get("FAIRREC_BN_BWD_SEPARATE") is None
                   and os.environ.get("FAIRREC_LINEAR_NO_GLDS") is None and os.environ.get("FAIRREC_LINEAR_SLOW") is None
                   and os.environ.get("FAIRREC_LINEAR_NO_SHARED") is None and os.environ.get("FAIRREC_BN_FOLD_SEPARATE") is None)
        # weight gradients in the fast form wait here and go out together at the end: every product in ONE launch and every
        # slab sum in a second (fr_linear_bwd_weight_multi) instead of two launches per layer -- the same kernels' bodies on
        # the same arguments, so the same bits
        deferred = []
        defer_ok = (os.environ.get("FAIRREC_LINEAR_NO_GLDS") is None and os.environ.get("FAIRREC_LINEAR_SLOW") is None
                    and os.environ.get("FAIRREC_WGRAD_PER_LAYER") is None)
        for l in range(L - 1, -1, -1):
            W = params[per * l].contiguous()
            Y = outs[l]
            a, c = (outs[l - 1], None) if l > 0 else (x0, x1)
            k0, k1 = a.shape[1], (c.shape[1] if c is not None else 0)
            N, K = W.shape
            fused_here = False                    # this layer's input-gradient product carried the dropout and the statistics below
            mk = ctx.masks8[l]                    # byte mask for the general kernels (None: no dropout, or pre-multiplied)
            scale = ctx.scale
            if ctx.premul[l]:                     # X o mask*scale was formed in the forward: the products see a plain input
                (a, c), scale = ctx.ins[l], 1.0
            elif drop is not None:
                scale = 1.0                       # the input is the previous layer's output, dropped in place
            act = ctx.act
            if at_z:
                Y, act, at_z = dY, 0, False
            elif ctx.dropped_out[l]:
                # Y is relu(z) o keep and dY the gradient with respect to it: one pass gives the gradient at z
                dA = torch.empty_like(Y)
                _C.check(lib.fr_act_bwd_dropped(dY.data_ptr(), Y.data_ptr(), ctx.scale, M * N, dA.data_ptr(), st),
                         "fr_act_bwd_dropped")
                dY, Y, act = dA, dA, 0
            elif ctx.use_bn:   # through activation + BatchNorm first; the linear layer then sees a plain gradient
                g = params[per * l + 2].contiguous()
                dZ = torch.empty_like(Y)
                dg = torch.empty(N, dtype=torch.float32, device=dev)
                dbt = torch.empty(N, dtype=torch.float32, device=dev)
                if bn_stats is not None:      # the statistics are there already: the apply launch alone
                    ws, bn_stats = bn_stats, None
                    _C.check(lib.fr_bn_bwd_ex(dY.data_ptr(), Y.data_ptr(), act, xhats[l].data_ptr(), invstds[l].data_ptr(),
                                              g.data_ptr(), M, N, dZ.data_ptr(), dg.data_ptr(), dbt.data_ptr(), ws.data_ptr(),
                                              ws.numel(), 1, st), "fr_bn_bwd_ex")
                else:
                    ws = torch.empty(lib.fr_bn_workspace_bytes(M, N), dtype=torch.uint8, device=dev)
                    _C.check(lib.fr_bn_bwd(dY.data_ptr(), Y.data_ptr(), act, xhats[l].data_ptr(), invstds[l].data_ptr(),
                                           g.data_ptr(), M, N, dZ.data_ptr(), dg.data_ptr(), dbt.data_ptr(), ws.data_ptr(),
                                           ws.numel(), st), "fr_bn_bwd")
                if ctx.needs_input_grad[7 + per * l + 2] or ctx.needs_input_grad[7 + per * l + 3]:
                    grads[per * l + 2], grads[per * l + 3] = dg, dbt
                dY, Y, act = dZ, dZ, 0
            elif act != 0 and mk is None and N % 32 == 0 and K % 32 == 0 and k0 % 32 == 0:
                # one pass through the activation's derivative, shared by the two products (which then take their fast form)
                dA = torch.empty_like(Y)
                _C.check(lib.fr_act_bwd(dY.data_ptr(), Y.data_ptr(), act, M * N, dA.data_ptr(), st), "fr_act_bwd")
                dY, Y, act = dA, dA, 0
            # (a module evaluated with `frozen=True` -- a discriminator inside the filter pass -- asks for no weight gradients)
            need_w = ctx.needs_input_grad[7 + per * l] or ctx.needs_input_grad[7 + per * l + 1]
            need = lib.fr_linear_bwd_weight_workspace_bytes(M, N, K) if need_w else 16
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            dW = torch.empty_like(W) if need_w else None
            db = torch.empty(N, dtype=torch.float32, device=dev) if need_w else None
            need0 = l > 0 or ctx.needs_input_grad[0]
            need1 = l == 0 and ctx.has_x1 and ctx.needs_input_grad[1]
            da = dc = None
            if need_w and N == 1 and c is None and mk is None and K % 64 == 0 and K <= 512:
                # a layer with one output: both products in one pass over its input
                da = torch.empty((M, k0), dtype=torch.float32, device=dev) if need0 else None
                at_z = l > 0 and ctx.dropped_out[l - 1]        # ... and on through the dropped ReLU below, in the same pass
                _C.check(lib.fr_linear_n1_bwd(dY.data_ptr(), Y.data_ptr(), act, a.data_ptr(), K, W.data_ptr(), M,
                                              ctx.scale if at_z else 0.0, _C.ptr(da), dW.data_ptr(), db.data_ptr(),
                                              ws.data_ptr(), ws.numel(), st), "fr_linear_n1_bwd")
            else:
                if need_w and defer_ok and act == 0 and mk is None and scale == 1.0 and N % 32 == 0 and K % 32 == 0 and k0 % 32 == 0 \
                        and (dY.data_ptr() | a.data_ptr() | (c.data_ptr() if c is not None else 0)) % 16 == 0:
                    deferred.append((dY, a, k0, c, k1, N, dW, db))
                elif need_w:
                    _C.check(lib.fr_linear_bwd_weight(dY.data_ptr(), Y.data_ptr(), act, a.data_ptr(), k0, _C.ptr(c), k1,
                                                      _C.ptr(mk), scale, M, N, dW.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                                      ws.numel(), st), "fr_linear_bwd_weight")
                if need0 or need1:
                    da = torch.empty((M, k0), dtype=torch.float32, device=dev)
                    dc = torch.empty((M, k1), dtype=torch.float32, device=dev) if k1 else None
                    drop_here = drop is not None and ctx.premul[l]
                    if (bn_fuse and l > 0 and k1 == 0 and act == 0 and mk is None and scale == 1.0 and N % 32 == 0 and K % 32 == 0
                            and not ctx.dropped_out[l - 1] and (drop_here or (drop is None and not ctx.premul[l]))
                            and (not drop_here or ctx.drop_off[l][0] % 4 == 0)
                            and (dY.data_ptr() | W.data_ptr() | da.data_ptr()) % 16 == 0):
                        bn_stats = torch.empty(lib.fr_bn_workspace_bytes(M, K), dtype=torch.uint8, device=dev)
                        _C.check(lib.fr_linear_bwd_input_bnstats(
                            dY.data_ptr(), W.data_ptr(), M, N, K, da.data_ptr(), outs[l - 1].data_ptr(), xhats[l - 1].data_ptr(),
                            ctx.act, bn_stats.data_ptr(), bn_stats.numel(), drop.p if drop_here else 0.0,
                            drop.seed if drop_here else 0, ctx.drop_off[l][0] if drop_here else 0,
                            drop.used.data_ptr() if drop_here else None, st), "fr_linear_bwd_input_bnstats")
                        fused_here = True
                    elif l > 0 and ctx.dropped_out[l - 1] and act == 0 and N % 32 == 0 and K % 32 == 0 and \
                            os.environ.get("FAIRREC_LINEAR_NO_GLDS") is None and os.environ.get("FAIRREC_LINEAR_SLOW") is None:
                        # the input is the dropped ReLU output of the layer below: on through it in the epilogue
                        _C.check(lib.fr_linear_bwd_input_relu(dY.data_ptr(), W.data_ptr(), M, N, K, a.data_ptr(), ctx.scale,
                                                              da.data_ptr(), st), "fr_linear_bwd_input_relu")
                        at_z = True
                    elif (l > 0 and not ctx.use_bn and ctx.act != 0 and act == 0 and mk is None and scale == 1.0 and drop is None
                          and ctx.masks is None and not ctx.premul[l] and not ctx.dropped_out[l - 1] and k1 == 0
                          and N % 32 == 0 and K % 32 == 0 and params[per * (l - 1)].shape[1] % 32 == 0
                          and (l - 1 > 0 or x1 is None or x0.shape[1] % 32 == 0)
                          and os.environ.get("FAIRREC_LINEAR_NO_GLDS") is None and os.environ.get("FAIRREC_LINEAR_SLOW") is None
                          and os.environ.get("FAIRREC_ACT_BWD_SEPARATE") is None):
                        # the input is the activation output of the layer below: on through act' in the epilogue, so that layer
                        # starts from the gradient at its pre-activation (no fr_act_bwd pass of its own; same bits)
                        _C.check(lib.fr_linear_bwd_input_act(dY.data_ptr(), W.data_ptr(), M, N, K, a.data_ptr(), ctx.act,
                                                             da.data_ptr(), st), "fr_linear_bwd_input_act")
                        at_z = True
                    else:
                        _C.check(lib.fr_linear_bwd_input(dY.data_ptr(), Y.data_ptr(), act, W.data_ptr(), _C.ptr(mk), scale,
                                                         M, N, da.data_ptr(), k0, _C.ptr(dc), k1, st), "fr_linear_bwd_input")
            grads[per * l], grads[per * l + 1] = dW, db
            if (need0 or need1) and not fused_here:
                if drop is not None and ctx.premul[l]:      # back through this layer's input dropout: the pattern again
                    if need0 and dc is not None and need1:
                        drop.again2(da, ctx.drop_off[l][0], dc, ctx.drop_off[l][1])
                    elif need0:
                        drop.again(da, ctx.drop_off[l][0])
                    elif dc is not None and need1:
                        drop.again(dc, ctx.drop_off[l][1])
                elif ctx.premul[l]:               # back through the dropout of this layer's input
                    da = da * (ctx.masks[l] if dc is None else ctx.masks[l][:, :k0]) if da is not None else None
                    dc = dc * ctx.masks[l][:, k0:] if dc is not None else None
            if need0 or need1:
                if l > 0:
                    dY = da
                else:
                    dx0, dx1 = (da if need0 else None), (dc if need1 else None)
        for q in range(0, len(deferred), _C.WGRAD_MAX):
            chunk = deferred[q:q + _C.WGRAD_MAX]
            jobs = (_C.FrWgradJob * len(chunk))(*[
                _C.FrWgradJob(dy.data_ptr(), a.data_ptr(), k0, _C.ptr(c), k1, N, dW.data_ptr(), db.data_ptr(), None, 0)
                for (dy, a, k0, c, k1, N, dW, db) in chunk])
            wsm = torch.empty(lib.fr_linear_bwd_weight_multi_workspace_bytes(jobs, len(chunk), M), dtype=torch.uint8, device=dev)
            _C.check(lib.fr_linear_bwd_weight_multi(jobs, len(chunk), M, wsm.data_ptr(), wsm.numel(), st),
                     "fr_linear_bwd_weight_multi")
        return (dx0, dx1, None, None, None, None, None, *grads)


class MLPLayers(nn.Module):
    def __init__(self, layers, dropout=0., activation='relu', bn=False, init_method=None):
        super().__init__()
        self.layers = layers
        self.dropout = dropout
        self.activation = activation
        self.use_bn = bn
        self.init_method = init_method
        if (activation.lower() if isinstance(activation, str) else activation) not in ACT_CODES:
            raise NotImplementedError(f"activation {activation} is not on the HIP path")
        mods = []
        for input_size, output_size in zip(self.layers[:-1], self.layers[1:]):
            mods.append(nn.Dropout(p=self.dropout))
            mods.append(nn.Linear(input_size, output_size))
            if self.use_bn:
                mods.append(nn.BatchNorm1d(num_features=output_size))
            act = activation_layer(self.activation, output_size)
            if act is not None:
                mods.append(act)
        self.mlp_layers = nn.Sequential(*mods)
        self.forced_masks: Optional[Sequence[torch.Tensor]] = None   # tests inject recorded dropout masks here
        if self.init_method is not None:
            self.apply(self.init_weights)

    def init_weights(self, module):
        if isinstance(module, nn.Linear):
            if self.init_method == 'norm':
                normal_(module.weight.data, 0, 0.01)
            if module.bias is not None:
                module.bias.data.fill_(0.0)

    def linears(self) -> List[nn.Linear]:
        return [m for m in self.mlp_layers if isinstance(m, nn.Linear)]

    def batchnorms(self) -> List[nn.BatchNorm1d]:
        return [m for m in self.mlp_layers if isinstance(m, nn.BatchNorm1d)]

    def forward(self, input_feature, second_block=None, passes=1, frozen=False, grad_at_z=False):
        """MLP(cat(input_feature, second_block)); `second_block` avoids materialising the concatenation.
        `passes` = 2 stands for the module being evaluated twice on the same input with both results used (the reference's
        PFCN filter pass does that, pfcn_biasedmf.py:209): without dropout the two evaluations are the same function at the
        same point, so ONE evaluation whose output feeds both consumers gives the same values and the same gradient
        (J^T (g1 + g2)); what differs is the BatchNorm bookkeeping, which advances twice: two momentum updates with the
        same batch statistics are one update with momentum 1 - (1 - m)^2, and the batch counter moves by two.
        `frozen`: the parameters take no gradient from this evaluation (only the input does): a discriminator inside the filter
        pass, whose gradients the reference computes and then never reads.
        `grad_at_z`: the caller promises that the output has exactly ONE consumer in the autograd graph and that this consumer
        returns the gradient at the top layer's PRE-activation (it has multiplied by act'(output) itself): the backward then
        skips its own pass through the top activation's derivative (a whole-table pass for FairGo's filters).
        BatchNorm layers always use batch statistics while `self.training` (and the reference's dict-held PFCN MLPs are
        never switched to eval mode); eval-mode BatchNorm (running statistics) is not on this path."""
        if input_feature.device.type != "cuda":
            raise _C.FairrecError("MLPLayers runs only on a ROCm device; there is no CPU fallback")
        lins, bns = self.linears(), self.batchnorms()
        if self.use_bn and not self.training:
            raise NotImplementedError("eval-mode BatchNorm (running statistics) is not on the HIP path")
        if self.use_bn:
            params = [t for lin, bn in zip(lins, bns) for t in (lin.weight, lin.bias, bn.weight, bn.bias)]
            if passes != 1 and (self.training and float(self.dropout) > 0.0):
                raise ValueError("passes > 1 needs a module without dropout (two evaluations would differ)")
            # (num_batches_tracked moves by `passes` in each layer's fold launch, fr_bn_fwd_ex: no launch of its own;
            # FAIRREC_BN_COUNT_SEPARATE=1: one multi-tensor add per forward instead, for A/B runs)
            own_launch = os.environ.get("FAIRREC_BN_COUNT_SEPARATE") is not None
            bn_buffers = [(bn.running_mean, bn.running_var, float(bn.eps), 1.0 - (1.0 - float(bn.momentum)) ** passes,
                           None if own_launch else bn.num_batches_tracked, int(passes)) for bn in bns]
            if own_launch:
                torch._foreach_add_([bn.num_batches_tracked for bn in bns], int(passes))
        else:
            params = [t for lin in lins for t in (lin.weight, lin.bias)]
            bn_buffers = None
        p = float(self.dropout) if self.training else 0.0
        # Dropout: drawn on the device with nothing stored (csrc/dropout.hip, `_Drop`); recorded patterns (tests) come as fp32
        # "keep scales" (0 or 1/(1-p)) that a layer in the fast form multiplies its input by, the others get as bytes.
        masks = drop = None
        if self.forced_masks is not None:
            scale = 1.0 / (1.0 - p) if p > 0 else 1.0
            masks = [m.to(input_feature.device, torch.float32).contiguous() * scale for m in self.forced_masks]
        elif p > 0.0:
            drop = _Drop(self._drop_seed(), self._drop_state(input_feature.device), p,
                         len(lins) + (1 if second_block is not None else 0))
        if frozen:
            params = [t.detach() for t in params]
        name = self.activation.lower() if isinstance(self.activation, str) else self.activation
        return _HipMLP.apply(input_feature, second_block, ACT_CODES[name] | (0x100 if grad_at_z else 0),
                             p if (masks is not None or drop is not None) else 0.0, masks, drop, bn_buffers, *params)

    _instances = 0

    def _drop_seed(self) -> int:
        """Seed of this module's dropout stream: torch's seed when the module first drops, mixed with the module's
        construction rank (same program + same torch.manual_seed => same patterns)."""
        s = getattr(self, "_seed", None)
        if s is None:
            import torch.distributed as dist
            MLPLayers._instances += 1
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0      # ranks drop independently
            s = self._seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + MLPLayers._instances * 0xD1B54A32D192ED03
                              + rank * 0xA0761D6478BD642F) % 2 ** 64
        return s

    def _drop_state(self, device) -> torch.Tensor:
        st = getattr(self, "_dstate", None)
        if st is None or st.device != device:
            st = self._dstate = torch.zeros(2, dtype=torch.int64, device=device)     # {call counter, ticket}
        return st
