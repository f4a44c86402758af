"""TEST DOUBLE (never shipped, never imported by the product): torch-CPU stand-ins for the HIP kernels the
row-sharded step calls, so that its exchange schedule can run over gloo on a machine without a GPU.  Same buffer
layouts as the kernels (fairrec_hip.h "slot layout").  Table updates use the oracle's dense Adam on every row
(the reference's semantics), not the lazy replay."""
import torch

from oracle import focf as O

TAIL = 3


def phys(n, chunk, stride, off=0):
    """Physical positions of logical slots 0..n-1 of a (chunk, stride) layout starting `off` slots into the buffer."""
    j = torch.arange(n)
    return (j // chunk) * stride + j % chunk + off if chunk else j + off


class CpuTable:
    def __init__(self, weight):
        self.weight = weight
        self.n_rows, self.dim = weight.shape
        self.m = torch.zeros_like(weight)
        self.v = torch.zeros_like(weight)
        self.step = 0
        self.ids = None
        self._ws = None


class CpuOps:
    def make_table(self, weight):
        return CpuTable(weight)

    def bucket_by_owner(self, idx, G, cap, stride, offset, send, slot, counts, aux, aux_slot, err):
        M = idx.numel()
        for o in range(G):
            send[o * stride + offset:o * stride + offset + cap] = -1
        slot.fill_(-1)
        counts.zero_()
        for j in range(M):
            o = int(idx[j]) % G
            k = int(counts[o])
            if k < cap:
                send[o * stride + offset + k] = int(idx[j]) // G
                slot[j] = o * stride + offset + k
                counts[o] += 1
            else:
                err |= 4
        if aux is not None:
            pair = torch.stack([aux.min(), aux.max()]).to(torch.float32).view(torch.int64)
            for o in range(G):
                send[o * stride + aux_slot] = pair[0]

    def bucket_pair(self, idx_a, idx_b, G, cap, stride, off_a, off_b, send, slot_a, slot_b, counts, aux, aux_slot, err):
        self.bucket_by_owner(idx_a, G, cap, stride, off_a, send, slot_a, counts[:G], None, 0, err)
        self.bucket_by_owner(idx_b, G, cap, stride, off_b, send, slot_b, counts[G:], aux, aux_slot, err)

    def gather_train(self, table, hyper, ids, ids_off, M, chunk, stride, rows, err):
        p = phys(M, chunk, stride, ids_off)
        table.ids = ids[p].clone()
        ok = table.ids >= 0
        out = torch.zeros((M, table.dim))
        out[ok] = table.weight[table.ids[ok]]
        rows[p] = out

    def side(self, fork=True):
        import contextlib
        return contextlib.nullcontext()

    def used_on_side(self, *things):
        pass

    def join_side(self):
        pass

    def alloc_ws(self, table, M):
        return None

    def sort_pair(self, ta, tb, ids, off_a, off_b, M, chunk, stride, ws_a, ws_b, err):
        pass

    def gather_train_pair(self, ta, tb, hyper, ids, off_a, off_b, M, chunk, stride, rows, ws_a, ws_b, err):
        self.gather_train(ta, hyper, ids, off_a, M, chunk, stride, rows, err)
        self.gather_train(tb, hyper, ids, off_b, M, chunk, stride, rows, err)

    def apply_grad_pair(self, ta, tb, hyper, M, chunk, stride, rows, grads, off_a, off_b, sweep_a, sweep_b):
        self.apply_grad(ta, hyper, M, chunk, stride, rows, grads, off_a, sweep_a)
        self.apply_grad(tb, hyper, M, chunk, stride, rows, grads, off_b, sweep_b)

    def apply_grad(self, table, hyper, M, chunk, stride, rows, grads, off, sweep):
        p = phys(M, chunk, stride, off)
        g = torch.zeros_like(table.weight)
        ok = table.ids >= 0
        g.index_add_(0, table.ids[ok], grads[p][ok])
        table.step += 1
        O.adam_dense_step_(table.weight, g, table.m, table.v, table.step, hyper.lr, hyper.weight_decay)

    def flush(self, table, hyper):
        pass

    def shard_score(self, rows, slot_u, slot_i, rating, sst, n_global, pred, coef, rec, cap, slot_stride, slot_offset,
                    sq, sq_part):
        su, si = slot_u.long(), slot_i.long()
        p = (rows[su] * rows[si]).sum(-1)
        e = p - rating
        pred.copy_(p)
        coef.copy_(2 * e / n_global)
        sq_part.zero_()
        sq_part[0] = (e * e).sum()
        if sq is not None:
            sq[0] = sq_part[0]
        if rec is not None:
            base = (si // slot_stride) * 3 * cap + (si % slot_stride - slot_offset)
            rec[base] = p
            rec[base + cap] = rating
            rec[base + 2 * cap] = sst

    def shard_fair(self, item_table, n_slots, rec, cap, ids_recv, mm_slot, stride, objective, fair_weight, reply,
                   sq_part, n_sq_part, scratch, err):
        G = n_slots // cap
        pairs = torch.stack([ids_recv[g * stride + mm_slot:g * stride + mm_slot + 1].view(torch.float32) for g in range(G)])
        smin, smax = pairs[:, 0].min(), pairs[:, 1].max()
        ok = item_table.ids >= 0
        pr = phys(n_slots, cap, 3 * cap)
        pred = rec[pr][ok].clone().requires_grad_()
        rating, sst, item = rec[pr + cap][ok], rec[pr + 2 * cap][ok], item_table.ids[ok]
        P, T = O.item_group_means(pred, rating, (sst != smin).float() if smin != smax else sst, item)
        d = {"value": P - T, "absolute": (P - T).abs(), "under": torch.clamp(T - P, min=0),
             "over": torch.clamp(P - T, min=0)}[objective]
        x = (d[:, 0] - d[:, 1]).abs()
        s = torch.nn.functional.smooth_l1_loss(x, torch.zeros_like(x), reduction="sum")
        (fair_weight * s).backward()
        c = torch.zeros(n_slots)
        c[ok] = pred.grad
        reply[phys(n_slots, cap, cap + TAIL)] = c
        for g in range(G):
            t = g * (cap + TAIL) + cap
            reply[t], reply[t + 1], reply[t + 2] = float(P.shape[0]), float(s.detach()), float(sq_part[:n_sq_part].sum())

    def _minmax(self, ids_recv, mm_slot, stride, G):
        pairs = torch.stack([ids_recv[g * stride + mm_slot:g * stride + mm_slot + 1].view(torch.float32) for g in range(G)])
        return pairs[:, 0].min(), pairs[:, 1].max()

    def nonparity_sums(self, pred, sst, ids_recv, mm_slot, stride, G, sq_part, n_sq_part, out5):
        smin, smax = self._minmax(ids_recv, mm_slot, stride, G)
        g0, g1 = sst == smin, (sst == smax) & (sst != smin)
        out5.copy_(torch.stack([sq_part[:n_sq_part].sum(), pred[g0].sum(), g0.sum().float(), pred[g1].sum(),
                                g1.sum().float()]))

    def nonparity_coef(self, coef, sst, ids_recv, mm_slot, stride, G, global5, n_global, fair_weight, loss_out, err):
        smin, smax = self._minmax(ids_recv, mm_slot, stride, G)
        n0, n1 = global5[2], global5[4]
        delta = global5[1] / n0 - global5[3] / n1
        dl = delta.clamp(-1, 1) * fair_weight
        coef[sst == smin] += dl / n0
        coef[(sst == smax) & (sst != smin)] -= dl / n1
        a = delta.abs()
        loss_out[1] = global5[0] / n_global
        loss_out[2] = 0.5 * a * a if a < 1 else a - 0.5
        loss_out[0] = loss_out[1] + fair_weight * loss_out[2]

    def shard_grads(self, rows, slot_u, slot_i, coef, reply, G, n_global, fair_weight, loss_out, cap, slot_stride,
                    slot_offset, grads):
        su, si = slot_u.long(), slot_i.long()
        c = coef.clone()
        if reply is not None:
            tails = torch.stack([reply[g * (cap + TAIL) + cap:g * (cap + TAIL) + cap + TAIL] for g in range(G)])
            K, fs, sqs = tails[:, 0].sum(), tails[:, 1].sum(), tails[:, 2].sum()
            if loss_out is not None:
                loss_out[1] = sqs / n_global
                loss_out[2] = fs / K
                loss_out[0] = loss_out[1] + fair_weight * loss_out[2]
            c = c + reply[(si // slot_stride) * (cap + TAIL) + (si % slot_stride - slot_offset)] / K
        grads[su] = c[:, None] * rows[si]
        grads[si] = c[:, None] * rows[su]

    # --- item-owner-computes schedule (ShardedFocfEngineV2): lists with empty positions, two row buffers ----------------
    def bucket_sparse(self, idx, G, cap, stride, offset, send, slot, counts, err):
        for o in range(G):
            send[o * stride + offset:o * stride + offset + cap] = -1
        slot.fill_(-1)
        counts.zero_()
        for j in range(idx.numel()):
            r = int(idx[j])
            if r < 0:
                continue
            o, k = r % G, int(counts[r % G])
            if k < cap:
                send[o * stride + offset + k] = r // G
                slot[j] = o * stride + offset + k
                counts[o] += 1
            else:
                err |= 4

    def pack_records(self, slot, user, rating, sst, cap, send):
        ok = slot >= 0
        s = slot[ok].long()
        send[s + cap] = user[ok]
        send[s + 2 * cap] = rating[ok].view(torch.int32).to(torch.int64)
        send[s + 3 * cap] = sst[ok].view(torch.int32).to(torch.int64) if sst is not None else 0

    def unpack_records(self, recv, G, cap, iid, uid, islot, rating, sst, mm):
        rv = recv.view(G, 4 * cap + 1)
        it = rv[:, :cap].reshape(-1)
        held = it >= 0
        iid.copy_(it)
        uid.copy_(torch.where(held, rv[:, cap:2 * cap].reshape(-1), torch.full((), -1, dtype=torch.int64)))
        islot.copy_(torch.where(held, torch.arange(G * cap, dtype=torch.int32), torch.full((), -1, dtype=torch.int32)))
        rating.copy_(torch.where(held, rv[:, 2 * cap:3 * cap].reshape(-1).to(torch.int32).view(torch.float32), torch.zeros(())))
        sst.copy_(torch.where(held, rv[:, 3 * cap:4 * cap].reshape(-1).to(torch.int32).view(torch.float32), torch.zeros(())))
        mm.copy_(rv[:, 4 * cap])

    def post_fair(self, reply, k_all, G, cap, sums):
        sums[0], sums[1] = float(reply[cap + 1]), float(reply[cap + 2])
        for g in range(G):
            reply[g * (cap + TAIL) + cap] = k_all[g]

    def loss_finish(self, sums, k_all, G, n_global, fair_weight, fair, loss):
        mse = sums[1] / n_global
        fv = sums[0] / k_all.sum() if fair else torch.zeros(())
        loss[0], loss[1], loss[2] = mse + fair_weight * fv, mse, fv

    def count_distinct(self, ids, n_rows, bitmap, count, out):
        out[0] = float(torch.unique(ids[ids >= 0]).numel())

    def shard_score2(self, rows_u, rows_i, slot_u, slot_i, rating, sst, n_global, pred, coef, rec, cap, slot_stride,
                     slot_offset, sq, sq_part):
        su, si = slot_u.long(), slot_i.long()
        ok = (su >= 0) & (si >= 0)
        p = torch.where(ok, (rows_u[su.clamp(min=0)] * rows_i[si.clamp(min=0)]).sum(-1), torch.zeros(()))
        e = p - rating
        pred.copy_(p)
        coef.copy_(2 * e / n_global)
        sq_part.zero_()
        sq_part[0] = (e * e).sum()
        if sq is not None:
            sq[0] = sq_part[0]
        if rec is not None:
            held = si >= 0
            base = (si[held] // slot_stride) * 3 * cap + (si[held] % slot_stride - slot_offset)
            rec[base] = p[held]
            rec[base + cap] = rating[held]
            rec[base + 2 * cap] = sst[held]

    def shard_grads2(self, rows_u, rows_i, slot_u, slot_i, coef, reply, G, n_global, fair_weight, loss_out, cap,
                     slot_stride, slot_offset, grad_u, grad_i):
        su, si = slot_u.long(), slot_i.long()
        ok = (su >= 0) & (si >= 0)
        su, si = su[ok], si[ok]
        c = coef[ok].clone()
        if reply is not None:
            K = sum(float(reply[g * (cap + TAIL) + cap]) for g in range(G))
            c = c + reply[(si // slot_stride) * (cap + TAIL) + (si % slot_stride - slot_offset)] / K
        grad_u[su] = c[:, None] * rows_i[si]
        grad_i[si] = c[:, None] * rows_u[su]
