"""TEST DOUBLE (never shipped, never imported by the product): torch-CPU stand-ins for the HIP kernels the
row-sharded step calls, so that its exchange schedule can run over gloo on a machine without a GPU.
Table updates use the oracle's dense Adam on every row (the reference's semantics), not the lazy replay."""
import torch

from oracle import focf as O


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

    def bucket_by_owner(self, idx, G, cap, err):
        M = idx.numel()
        send = torch.full((G * cap,), -1, dtype=torch.int64)
        slot = torch.full((M,), -1, dtype=torch.int32)
        counts = torch.zeros(G, dtype=torch.int32)
        for j in range(M):
            o = int(idx[j]) % G
            k = int(counts[o])
            if k < cap:
                send[o * cap + k] = int(idx[j]) // G
                slot[j] = o * cap + k
                counts[o] += 1
            else:
                err |= 4
        return send, slot, counts

    def gather_train(self, table, hyper, ids, err):
        table.ids = ids.clone()
        rows = torch.zeros((ids.numel(), table.dim))
        ok = ids >= 0
        rows[ok] = table.weight[ids[ok]]
        return rows

    def apply_grad(self, table, hyper, grads, sweep):
        g = torch.zeros_like(table.weight)
        ok = table.ids >= 0
        g.index_add_(0, table.ids[ok], grads[ok])
        table.step += 1
        O.adam_dense_step_(table.weight, g, table.m, table.v, table.step, hyper.lr, hyper.weight_decay)

    def flush(self, table, hyper):
        pass

    def shard_score(self, rows_u, rows_i, slot_u, slot_i, rating, sst, n_global, want_rec):
        ue, ie = rows_u[slot_u.long()], rows_i[slot_i.long()]
        pred = (ue * ie).sum(-1)
        err = pred - rating
        coef = 2 * err / n_global
        rec = None
        if want_rec:
            rec = torch.zeros((3, rows_i.shape[0]))
            rec[0, slot_i.long()] = pred
            rec[1, slot_i.long()] = rating
            rec[2, slot_i.long()] = sst
        return pred, coef, rec, (err * err).sum().reshape(1)

    def shard_fair(self, item_table, rec, minmax, objective, fair_weight, err):
        ok = item_table.ids >= 0
        pred = rec[0, ok].clone().requires_grad_()
        rating, sst, item = rec[1, ok], rec[2, ok], item_table.ids[ok]
        P, T = O.item_group_means(pred, rating, (sst != minmax[0]).float() if minmax[0] != minmax[1] else sst, item)
        d = {"value": P - T, "absolute": (P - T).abs(), "under": torch.clamp(T - P, min=0),
             "over": torch.clamp(P - T, min=0)}[objective]
        x = (d[:, 0] - d[:, 1]).abs()
        s = torch.nn.functional.smooth_l1_loss(x, torch.zeros_like(x), reduction="sum")
        (fair_weight * s).backward()
        coef = torch.zeros(rec.shape[1])
        coef[ok] = pred.grad
        return coef, torch.stack([s.detach(), torch.tensor(float(P.shape[0]))])

    def shard_grads(self, rows_u, rows_i, slot_u, slot_i, coef, coef_slots, inv_k):
        su, si = slot_u.long(), slot_i.long()
        c = coef.clone()
        if coef_slots is not None:
            c = c + coef_slots[si] * inv_k
        gu = torch.zeros_like(rows_u)
        gi = torch.zeros_like(rows_i)
        gu[su] = c[:, None] * rows_i[si]
        gi[si] = c[:, None] * rows_u[su]
        return gu, gi
