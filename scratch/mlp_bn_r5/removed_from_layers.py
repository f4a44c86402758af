BNL_MAX_WIDTH = 256          # csrc/mlp_bn.hip: BNL_MAXW
BNL_MAX_ROWS = 1 << 16       # one workgroup folds the statistics of all 32-row tiles: a batch-sized M, not a table-sized one


def _bn_chain_ok(x0, x1, masks, params) -> bool:
    """The one-launch-per-layer form of csrc/mlp_bn.hip can take this MLP: BatchNorm behind every layer, one input block, no
    recorded masks, input widths multiples of 32, every width <= 256, a batch-sized M.  It is OPT-IN (`FAIRREC_BN_FUSED=1`):
    measured on MI355X it is the same numbers (bit for bit where every width is a multiple of 32) in a third of the launches
    and 10-40 % SLOWER than the layered form at every batch size from 256 to 8192 -- a launch of it is one workgroup per 32 rows
    walking load -> product -> store -> arrive -> fold as one latency chain at one wave per SIMD (DESIGN.md §4b has the
    phase stamps) -- so the layered form stays the default."""
    if os.environ.get("FAIRREC_BN_FUSED") is None or os.environ.get("FAIRREC_BN_LAYERED") is not None:
        return False
    if x1 is not None or masks is not None:
        return False
    if x0.dim() != 2 or x0.shape[0] > BNL_MAX_ROWS or x0.dtype != torch.float32:
        return False
    k = x0.shape[1]
    for l in range(len(params) // 4):
        n, kk = params[4 * l].shape
        if kk != k or k % 32 != 0 or k > BNL_MAX_WIDTH or n > BNL_MAX_WIDTH:
            return False
        k = n
    return True


class _HipBnMLP(torch.autograd.Function):
    """y = MLP(x) with BatchNorm behind every layer, ONE launch per layer and direction (csrc/mlp_bn.hip): a layer is normalised
    (and dropped) by the launch that consumes it.  Saves per layer the pre-BatchNorm output Z, its folded statistics and -- for
    a layer whose weights train -- the formed input A; the same dropout stream as `_HipMLP` (`_Drop.take` in the same order)."""

    @staticmethod
    def forward(ctx, x0, act, drop, bn_buffers, ticket, *params):
        lib = _C.lib()
        st = _C.current_stream()
        L = len(params) // 4
        M = x0.shape[0]
        dev = x0.device
        x0 = x0.contiguous()
        need_w = [ctx.needs_input_grad[5 + 4 * l] or ctx.needs_input_grad[5 + 4 * l + 1] for l in range(L)]
        widths = [x0.shape[1]] + [params[4 * l].shape[0] for l in range(L)]
        ws = torch.empty(lib.fr_bnl_workspace_bytes(M, max(widths)), dtype=torch.uint8, device=dev)
        p_drop = drop.p if drop is not None else 0.0
        seed = drop.seed if drop is not None else 0
        src = _C.FrBnSrc(x0.data_ptr(), None, None, None, 0, 0.0, 0, 0)
        Zs, fins, As, offs = [], [], [], []
        keep = []          # contiguous copies the launches read
        for l in range(L):
            W, b = params[4 * l].contiguous(), params[4 * l + 1].contiguous()
            g, be = params[4 * l + 2].contiguous(), params[4 * l + 3].contiguous()
            keep += [W, b, g, be]
            N, K = W.shape
            rm, rv, eps, mom, nbt, n_pass = bn_buffers[l]
            off, used, tick = drop.take(M * K) if drop is not None else (0, None, None)
            src.drop_p, src.drop_seed, src.drop_off = p_drop, seed, off
            Z = torch.empty((M, N), dtype=torch.float32, device=dev)
            fin = torch.empty((N, 2), dtype=torch.float32, device=dev)
            A = torch.empty((M, K), dtype=torch.float32, device=dev) if need_w[l] and (l > 0 or drop is not None) else None
            _C.check(lib.fr_bnl_fwd(ctypes.byref(src), M, K, W.data_ptr(), b.data_ptr(), N, Z.data_ptr(), _C.ptr(A), eps, mom,
                                    _C.ptr(rm), _C.ptr(rv), _C.ptr(nbt), n_pass, fin.data_ptr(), ws.data_ptr(), ws.numel(),
                                    ticket.data_ptr(), drop.state.data_ptr() if drop is not None else None, used, tick, st),
                     "fr_bnl_fwd")
            Zs.append(Z)
            fins.append(fin)
            As.append(A)
            offs.append(off)
            src = _C.FrBnSrc(Z.data_ptr(), fin.data_ptr(), g.data_ptr(), be.data_ptr(), act, 0.0, 0, 0)
        Y = torch.empty_like(Zs[-1])
        _C.check(lib.fr_bnl_out(ctypes.byref(src), M, Y.shape[1], Y.data_ptr(), st), "fr_bnl_out")
        ctx.act, ctx.L, ctx.drop, ctx.offs, ctx.need_w, ctx.ticket = act, L, drop, offs, need_w, ticket
        ctx.has_A = [a is not None for a in As]
        ctx.save_for_backward(x0, *params, *Zs, *fins, *[a for a in As if a is not None])
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = _C.lib()
        st = _C.current_stream()
        saved = list(ctx.saved_tensors)
        L, act, drop = ctx.L, ctx.act, ctx.drop
        x0 = saved.pop(0)
        params = [t.contiguous() for t in saved[:4 * L]]
        Zs, fins = saved[4 * L:5 * L], saved[5 * L:6 * L]
        rest = saved[6 * L:]
        As = [rest.pop(0) if h else None for h in ctx.has_A]
        M = x0.shape[0]
        dev = x0.device
        dY = dY.contiguous()
        widths = [x0.shape[1]] + [params[4 * l].shape[0] for l in range(L)]
        ws = torch.empty(lib.fr_bnl_workspace_bytes(M, max(widths)), dtype=torch.uint8, device=dev)
        ticket = ctx.ticket
        grads: List[Optional[torch.Tensor]] = [None] * (4 * L)
        p_drop = drop.p if drop is not None else 0.0
        seed = drop.seed if drop is not None else 0
        used = drop.used.data_ptr() if drop is not None else None

        def src_of(l):
            return _C.FrBnSrc(Zs[l].data_ptr(), fins[l].data_ptr(), params[4 * l + 2].data_ptr(), params[4 * l + 3].data_ptr(),
                              act, 0.0, 0, 0)

        def stat_grads(l, n):
            want = ctx.needs_input_grad[5 + 4 * l + 2] or ctx.needs_input_grad[5 + 4 * l + 3]
            dg = torch.empty(n, dtype=torch.float32, device=dev) if want else None
            db = torch.empty(n, dtype=torch.float32, device=dev) if want else None
            if want:
                grads[4 * l + 2], grads[4 * l + 3] = dg, db
            return dg, db

        N = widths[L]
        G = dY                      # the gradient at a layer's output: the loss's for the top layer, dA of the layer above below
        sums = torch.empty((N, 2), dtype=torch.float32, device=dev)
        dg, db = stat_grads(L - 1, N)
        top = src_of(L - 1)
        _C.check(lib.fr_bnl_bwd_top(dY.data_ptr(), ctypes.byref(top), M, N, sums.data_ptr(), _C.ptr(dg), _C.ptr(db),
                                    ws.data_ptr(), ws.numel(), ticket.data_ptr(), st), "fr_bnl_bwd_top")
        deferred = []
        dx0 = None
        for l in range(L - 1, -1, -1):
            W = params[4 * l]
            N, K = W.shape
            need_in = l > 0 or ctx.needs_input_grad[0]
            if not need_in and not ctx.need_w[l]:
                break
            dZ = torch.empty((M, N), dtype=torch.float32, device=dev) if ctx.need_w[l] else None
            dA = torch.empty((M, K), dtype=torch.float32, device=dev) if need_in else None
            me = src_of(l)
            if l > 0:
                below = src_of(l - 1)
                sums_b = torch.empty((K, 2), dtype=torch.float32, device=dev)
                dg, db = stat_grads(l - 1, K)
            else:
                below = _C.FrBnSrc(None, None, None, None, 0, 0.0, 0, 0)
                sums_b, dg, db = None, None, None
            below.drop_p, below.drop_seed, below.drop_off = p_drop, seed, ctx.offs[l]
            _C.check(lib.fr_bnl_bwd(G.data_ptr(), ctypes.byref(me), sums.data_ptr(), M, N, W.data_ptr(), K, _C.ptr(dZ),
                                    ctypes.byref(below), used, _C.ptr(dA), _C.ptr(sums_b), _C.ptr(dg), _C.ptr(db), ws.data_ptr(),
                                    ws.numel(), ticket.data_ptr(), st), "fr_bnl_bwd")
            if ctx.need_w[l]:
                A = As[l] if As[l] is not None else x0
                dW = torch.empty_like(W)
                dbias = torch.empty(N, dtype=torch.float32, device=dev)
                grads[4 * l], grads[4 * l + 1] = dW, dbias
                if N % 32 == 0 and os.environ.get("FAIRREC_LINEAR_NO_GLDS") is None and os.environ.get("FAIRREC_LINEAR_SLOW") is None:
                    deferred.append((dZ, A, K, None, 0, N, dW, dbias))
                else:
                    w2 = torch.empty(lib.fr_linear_bwd_weight_workspace_bytes(M, N, K), dtype=torch.uint8, device=dev)
                    _C.check(lib.fr_linear_bwd_weight(dZ.data_ptr(), dZ.data_ptr(), 0, A.data_ptr(), K, None, 0, None, 1.0, M, N,
                                                      dW.data_ptr(), dbias.data_ptr(), w2.data_ptr(), w2.numel(), st),
                             "fr_linear_bwd_weight")
            if l > 0:
                G, sums = dA, sums_b
            else:
                dx0 = dA
        for q in range(0, len(deferred), _C.WGRAD_MAX):
            chunk = deferred[q:q + _C.WGRAD_MAX]
            jobs = (_C.FrWgradJob * len(chunk))(*[
                _C.FrWgradJob(dy.data_ptr(), a.data_ptr(), k0, _C.ptr(c), k1, n, dW.data_ptr(), db_.data_ptr(), None, 0)
                for (dy, a, k0, c, k1, n, dW, db_) in chunk])
            wsm = torch.empty(lib.fr_linear_bwd_weight_multi_workspace_bytes(jobs, len(chunk), M), dtype=torch.uint8, device=dev)
            _C.check(lib.fr_linear_bwd_weight_multi(jobs, len(chunk), M, wsm.data_ptr(), wsm.numel(), st),
                     "fr_linear_bwd_weight_multi")
        return (dx0, None, None, None, None, *grads)


