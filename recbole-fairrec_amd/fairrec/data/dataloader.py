"""Training batch feeds with the iteration contract of recbole/data/dataloader/*.py (len(), iteration yields
`Interaction`s, `.dataset`): the plain shuffled loader (general_dataloader.py:25-65 without negative sampling)
and the item-complete loader FOCF trains with (focf_dataloader.py:5-51).

FOCFDataLoader keeps the reference's draw order (`np.random.choice(candidates, 1, False)` per picked item)
but finds an item's interactions through a CSR built once instead of an O(#inter) `np.where` per pick
(SURVEY.md §8-f row f-3).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .interaction import Interaction


class AbstractDataLoader:
    def __init__(self, config, dataset, shuffle=False):
        self.config, self.dataset, self.shuffle = config, dataset, shuffle
        self.step = int(config['train_batch_size'])
        self.pr = 0

    @property
    def pr_end(self):
        return len(self.dataset)

    def __len__(self):
        return math.ceil(self.pr_end / self.step)

    def __iter__(self):
        if self.shuffle:
            self.dataset.shuffle()
        return self

    def __next__(self):
        if self.pr >= self.pr_end:
            self.pr = 0
            raise StopIteration()
        return self._next_batch_data()


def _prejoin(dataset):
    """Device-resident interactions: attach the user feature columns to the WHOLE interaction table once instead of to
    every batch (dataset.py:1256-1269 joins per batch; the values are the same, and a batch is then a pure slice --
    no gather launch and no host work per step)."""
    feat = dataset.inter_feat
    if dataset.user_feat is not None and feat[dataset.uid_field].device.type == 'cuda':
        dataset.inter_feat = dataset.join(feat)


class TrainDataLoader(AbstractDataLoader):
    """general_dataloader.py:25-65 + NegSampleDataLoader (abstract_dataloader.py:110-198): fixed-size batches in
    (shuffled) dataset order, user features joined in, and -- when `train_neg_sample_args` asks for it and a
    `sampler` is given -- negatives drawn per batch and assembled pair-wise (`neg_<item field>` column) or point-wise
    (negatives appended, `LABEL_FIELD` 1/0).  The sampler is fairrec.sampler.Sampler: ids are drawn on the device from
    the device mirror of numpy's generator, bit-identical to the reference's host draws, so the batch never has to
    leave the GPU (keep `dataset.inter_feat` there: `dataset.to(device)`)."""

    def __init__(self, config, dataset, sampler=None, shuffle=False):
        super().__init__(config, dataset, shuffle=shuffle)
        self.sampler = sampler
        self.uid_field, self.iid_field = dataset.uid_field, dataset.iid_field
        self.neg_sample_args = config['train_neg_sample_args'] or {'strategy': 'none'}
        self.times = 1
        if sampler is not None and self.neg_sample_args['strategy'] == 'by':
            if self.neg_sample_args.get('dynamic', 'none') != 'none':
                raise NotImplementedError('dynamic negative sampling is not on the device path')
            self.neg_sample_num = int(self.neg_sample_args['by'])
            self.dl_format = config['MODEL_INPUT_TYPE']
            from ..utils.enum_type import InputType
            if self.dl_format == InputType.POINTWISE:
                self.times = 1 + self.neg_sample_num
                self.label_field = config['LABEL_FIELD']
            elif self.dl_format == InputType.PAIRWISE:
                self.times = self.neg_sample_num
                self.neg_item_id = config['NEG_PREFIX'] + self.iid_field
            else:
                raise ValueError(f'`neg sampling by` with dl_format [{self.dl_format}] not been implemented.')
            batch_num = max(int(config['train_batch_size']) // self.times, 1)     # general_dataloader.py:41-50
            self.step = batch_num
        else:
            self.sampler = None
        _prejoin(self.dataset)

    def _neg_sampling(self, inter_feat):
        from ..utils.enum_type import InputType
        neg = self.sampler.sample_by_user_ids(inter_feat[self.uid_field], inter_feat[self.iid_field], self.neg_sample_num)
        dev = inter_feat[self.uid_field].device
        neg = neg.to(dev)
        if self.dl_format == InputType.PAIRWISE:                 # abstract_dataloader.py:182-188
            out = inter_feat.repeat(self.times)
            out.update(Interaction({self.neg_item_id: neg}))
            return out
        pos = len(inter_feat)                                    # abstract_dataloader.py:190-198
        out = inter_feat.repeat(self.times)
        out[self.iid_field][pos:] = neg
        labels = torch.zeros(pos * self.times, device=dev)
        labels[:pos] = 1.0
        out.update(Interaction({self.label_field: labels}))
        return out

    def _next_batch_data(self):
        cur = self.dataset[self.pr:self.pr + self.step]
        self.pr += self.step
        if self.sampler is not None:
            cur = self._neg_sampling(cur)
        return self.dataset.join(cur)

    @property
    def sliceable(self):
        """A batch is a slice of the (shuffled) dataset and nothing more: `take` can hand out runs of batches."""
        return self.sampler is None

    def take(self, n_batches):
        """The next `n_batches` batches of this epoch as ONE Interaction (their rows back to back) and the batch size --
        what `n_batches` calls of next() would have yielded, for a consumer that walks a run of batches itself
        (Trainer._train_epoch hands it to `model.train_steps`: one library call per run instead of an interpreter round
        trip per batch).  None when the epoch is over.  The shuffle of `__iter__` has happened by then, so the slices are the reference's batches
        (general_dataloader.py:59-65)."""
        if self.sampler is not None:
            raise TypeError('take(): this loader draws negatives per batch (check `sliceable` first)')
        if self.pr >= self.pr_end:
            self.pr = 0                       # as __next__ does when it raises StopIteration
            return None
        hi = min(self.pr + self.step * int(n_batches), self.pr_end)
        cur = self.dataset.join(self.dataset[self.pr:hi])
        self.pr = hi
        return cur, self.step


def _degree(indptr, ids):
    return indptr[ids + 1] - indptr[ids]


class FOCFDataLoader(AbstractDataLoader):
    """Item-complete batches: keep picking a random not-yet-picked item and append ALL its interactions until
    the batch holds >= train_batch_size rows (focf_dataloader.py:37-51)."""

    def __init__(self, config, dataset, shuffle=False):
        super().__init__(config, dataset, shuffle=False)
        self.ITEM_ID = config['ITEM_ID_FIELD']
        self.dataset.sort(by=self.ITEM_ID)                      # focf_dataloader.py:11
        items = self.dataset.inter_feat[self.ITEM_ID].cpu().numpy()      # the batch composition stays host logic
        self.item_num = self.dataset.item_num
        self.item_uniques = np.unique(items)
        # CSR by item over the item-sorted interaction array
        self.indptr = np.searchsorted(items, np.arange(self.item_num + 1), side="left")
        self._sizes, self._rows, self._row_pr = [], None, 0     # the composed epoch: batch sizes still to come, its rows
        _prejoin(self.dataset)

    def _compose_epoch(self):
        """The interaction index lists of ALL batches of the epoch, drawn in one go.  Which rows form a batch depends on
        nothing but numpy's generator and the item CSR, and nothing else draws from that generator while a training epoch of
        FOCF runs (no negatives, no attribute masks), so drawing the epoch's picks ahead leaves every consumer at the
        reference's position of the stream -- with ONE hand-over between numpy and its device mirror per epoch
        (`host_numpy_stream`: the evaluation loaders' negatives come from the mirror) instead of one per batch."""
        from ..sampler import host_numpy_stream
        from .. import _C
        lib = _C.lib()
        n_batches = max(-(-(self.pr_end - self.pr) // self.step), 0)
        if n_batches == 0:
            return []
        # the picks: csrc/focf_compose.hip makes numpy's draws (`np.random.choice(select_item[is_select], 1, False)` per pick,
        # focf_dataloader.py:41-44) without building the permutation behind each of them
        uniq = np.ascontiguousarray(self.item_uniques, dtype=np.int64)
        indptr = np.ascontiguousarray(self.indptr, dtype=np.int64)
        deg = _degree(indptr, uniq)
        worst = min(int(-(-self.step // max(int(deg.min()), 1))), uniq.size)      # picks of one batch, at most
        usual = int(self.step / max(float(deg.mean()), 1.0)) + 64
        ends = np.empty(n_batches, dtype=np.int64)
        got = np.zeros(1, dtype=np.int64)
        with host_numpy_stream():
            name, key, pos, has_gauss, cached = np.random.get_state()
            state = np.empty(625, dtype=np.uint32)
            for per_batch in (min(2 * usual, worst), worst):         # a failed call leaves `state` where it was
                state[:624], state[624] = key, pos
                picks = np.empty(n_batches * per_batch, dtype=np.int64)
                rc = lib.fr_focf_compose_epoch(state.ctypes.data, uniq.ctypes.data, uniq.size, indptr.ctypes.data,
                                               int(self.step), int(self.pr), int(self.pr_end), picks.ctypes.data, picks.size,
                                               ends.ctypes.data, ends.size, got.ctypes.data)
                if rc == 0:
                    break
            _C.check(rc, "fr_focf_compose_epoch")
            np.random.set_state((name, state[:624].copy(), int(state[624]), has_gauss, cached))
        # the rows of a pick = its CSR range; all ranges of the epoch expanded at once
        picks = picks[:ends[int(got[0]) - 1]]
        lo, n = indptr[picks], _degree(indptr, picks)
        first = np.cumsum(n) - n                                      # where each pick's rows start in the epoch's row list
        rows = np.arange(int(n.sum()), dtype=np.int64) - np.repeat(first - lo, n)
        cuts = first[ends[:int(got[0]) - 1]] if got[0] > 1 else []
        return np.split(rows, cuts)

    def _begin_epoch(self):
        """Compose the epoch and gather its rows ONCE, in batch order (one index crossing to the device and one gather per
        column per epoch, like the plain loader's shuffle): a batch is then a slice, a run of batches a longer slice."""
        parts = self._compose_epoch()
        self._sizes = [len(p) for p in parts][::-1]                     # sizes of the batches still to come, last first
        self._rows = self.dataset.join(self.dataset[np.concatenate(parts)]) if parts else None
        self._row_pr = 0

    def _next_batch_data(self):
        if not self._sizes:
            self._begin_epoch()
        n = self._sizes.pop()
        self.pr += self.step
        lo, self._row_pr = self._row_pr, self._row_pr + n
        return self._rows[lo:lo + n]

    sliceable = True

    def take(self, n_batches):
        """The next `n_batches` batches of this epoch as ONE Interaction (rows back to back) and their sizes (ragged: a
        batch holds whole item histories) -- TrainDataLoader.take for item-complete batches.  None when the epoch is over."""
        if self.pr >= self.pr_end:
            self.pr = 0
            return None
        if not self._sizes:
            self._begin_epoch()
        sizes = [self._sizes.pop() for _ in range(min(int(n_batches), len(self._sizes)))]
        self.pr += self.step * len(sizes)
        lo, self._row_pr = self._row_pr, self._row_pr + sum(sizes)
        return self._rows[lo:self._row_pr], sizes


class FullSortEvalDataLoader:
    """general_dataloader.py:170-262 for non-sequential models: iterates over the users of the evaluation set in id order,
    `max(eval_batch_size // item_num, 1)` users per batch, and yields
        (user_df, (history_u, history_i), positive_u, positive_i)
    user_df = the users' feature rows; history = items to mask (used in an earlier phase, from `sampler.used_ids`, minus
    the positives of this phase); positives = the items of this evaluation set; `*_u` are row indices into the batch.
    Everything is a device tensor built from two CSRs (positives / history per user); positives are listed in ascending
    item order (the reference lists them in Python-set order; every metric is order-invariant in full mode)."""

    def __init__(self, config, dataset, sampler, shuffle=False):
        self.config, self.dataset = config, dataset
        self.uid_field, self.iid_field = dataset.uid_field, dataset.iid_field
        self.device = torch.device(config['device'])
        n_users, n_items = dataset.user_num, dataset.item_num
        u = dataset.inter_feat[self.uid_field].to(self.device, torch.int64)
        i = dataset.inter_feat[self.iid_field].to(self.device, torch.int64)
        pos_keys = torch.unique(u * n_items + i)                                   # sorted by (user, item), distinct
        self.pos_items = pos_keys % n_items
        self.pos_indptr = self._indptr(pos_keys // n_items, n_users)
        used_indptr, used_items, _ = sampler.used_ids                              # CSR of the phase (incl. earlier phases)
        used_users = torch.repeat_interleave(torch.arange(n_users, device=self.device), used_indptr[1:] - used_indptr[:-1])
        used_keys = used_users * n_items + used_items.to(torch.int64)
        hist_keys = used_keys[~torch.isin(used_keys, pos_keys)]                    # history = used - positive
        self.hist_items = hist_keys % n_items
        self.hist_indptr = self._indptr(hist_keys // n_items, n_users)
        self.uid_list = torch.unique(u)                                            # users that have positives, ascending
        self.user_df = dataset.join(Interaction({self.uid_field: self.uid_list}))
        self.step = max(int(config['eval_batch_size'] or 4096) // n_items, 1)
        self.pr = 0

    def _indptr(self, users, n_users):
        counts = torch.bincount(users, minlength=n_users)
        indptr = torch.zeros(n_users + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(counts, 0, out=indptr[1:])
        return indptr

    def __len__(self):
        return math.ceil(len(self.uid_list) / self.step)

    def __iter__(self):
        self.pr = 0
        return self

    def _rows(self, indptr, items, uids):
        lo, hi = indptr[uids], indptr[uids + 1]
        n = hi - lo
        row = torch.repeat_interleave(torch.arange(len(uids), device=self.device), n)
        start = torch.repeat_interleave(lo - torch.cumsum(n, 0) + n, n)
        return row, items[torch.arange(int(n.sum()), device=self.device) + start]

    def __next__(self):
        if self.pr >= len(self.uid_list):
            self.pr = 0
            raise StopIteration()
        sl = slice(self.pr, self.pr + self.step)
        self.pr += self.step
        uids = self.uid_list[sl]
        user_df = Interaction({k: v[sl] for k, v in self.user_df.interaction.items()})
        history = self._rows(self.hist_indptr, self.hist_items, uids)
        positive_u, positive_i = self._rows(self.pos_indptr, self.pos_items, uids)
        return user_df, history, positive_u, positive_i


class NegSampleEvalDataLoader:
    """general_dataloader.py:68-158 with `eval_args.mode: uniN`: users in id order; per user its positives (dataset order
    after a stable sort by user) followed by N sampled negatives per positive, drawn user by user from the numpy-compatible
    device stream (fr_sample_negatives_calls: each user's re-draw rounds complete before the next user draws, exactly like
    the reference's consecutive sample_by_user_ids calls).  Yields (interaction, row_idx, positive_u, positive_i); how many
    users form a batch follows :100-117."""

    def __init__(self, config, dataset, sampler, shuffle=False):
        mode = (config['eval_args'] or {}).get('mode', '')
        if mode[:3] != 'uni':
            raise NotImplementedError(f"evaluation mode [{mode}]: uniN (uniform negatives) or full")
        self.config, self.dataset, self.sampler = config, dataset, sampler
        self.neg_sample_num = int(mode[3:])
        self.times = 1 + self.neg_sample_num
        self.uid_field, self.iid_field = dataset.uid_field, dataset.iid_field
        self.device = torch.device(config['device'])
        dataset.sort(by=self.uid_field, ascending=True)
        u = dataset.inter_feat[self.uid_field].to(self.device, torch.int64)
        self.items = dataset.inter_feat[self.iid_field].to(self.device, torch.int64)
        self.uid_list, counts = torch.unique_consecutive(u, return_counts=True)
        self.counts = counts
        self.start = torch.cumsum(counts, 0) - counts
        inters = sorted((counts * self.times).tolist(), reverse=True)           # :103-113
        batch_size = int(config['eval_batch_size'] or 4096)
        batch_num, size = 1, inters[0]
        for k in range(1, len(inters)):
            if size + inters[k] > batch_size:
                break
            batch_num, size = k + 1, size + inters[k]
        self.step = batch_num
        self.pr = 0
        # the user feature columns every batch is joined with (dataset.py:1256-1269), resident where the batches are built: a
        # host-side table costs a device -> host -> device round trip of the batch's user column per batch (20 ms of a 25 ms batch)
        uf = dataset.user_feat
        self._user_cols = ({k: uf[k].to(self.device) for k in uf.columns if k != self.uid_field} if uf is not None else {})

    def __len__(self):
        return math.ceil(len(self.uid_list) / self.step)

    def __iter__(self):
        self.pr = 0
        return self

    def __next__(self):
        if self.pr >= len(self.uid_list):
            self.pr = 0
            raise StopIteration()
        sl = slice(self.pr, self.pr + self.step)
        self.pr += self.step
        dev, N = self.device, self.neg_sample_num
        uids, P, st = self.uid_list[sl], self.counts[sl], self.start[sl]
        Ub = uids.numel()
        indptr, used_items, _ = self.sampler.used_ids
        neg = self.sampler.rs.sample_calls(1, self.dataset.item_num, uids, P * N, indptr, used_items)
        blk = P * self.times
        blk_off = torch.cumsum(blk, 0) - blk
        rows = int(blk.sum())
        row_idx = torch.repeat_interleave(torch.arange(Ub, device=dev), blk)
        within = torch.arange(rows, device=dev) - blk_off[row_idx]
        is_pos = within < P[row_idx]
        item_col = torch.empty(rows, dtype=torch.int64, device=dev)
        item_col[is_pos] = self.items[(st[row_idx] + within)[is_pos]]
        neg_off = torch.cumsum(P * N, 0) - P * N
        item_col[~is_pos] = neg[(neg_off[row_idx] + within - P[row_idx])[~is_pos]]
        ucol = uids[row_idx]
        inter = Interaction({self.uid_field: ucol, self.iid_field: item_col})
        for k, col in self._user_cols.items():
            inter[k] = col[ucol]
        positive_u = row_idx[is_pos]
        positive_i = item_col[is_pos]
        return inter, row_idx, positive_u, positive_i
