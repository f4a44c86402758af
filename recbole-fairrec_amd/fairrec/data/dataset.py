"""Minimal in-memory dataset with the attributes the fair models and dataloaders read
(`num(field)`, `inter_feat`, `get_user_feature()`, `user_num`, `item_num`, `__len__`, `__getitem__`, `sort`,
`shuffle`): the part of recbole/data/dataset/dataset.py the hot path touches.  Atomic-file loading,
remapping and splitting are out of scope; synthetic generators live here instead.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .interaction import Interaction


class InteractionDataset:
    def __init__(self, config, inter_feat: Interaction, user_feat: Optional[Interaction] = None,
                 n_users: Optional[int] = None, n_items: Optional[int] = None):
        self.config = config
        self.uid_field = config['USER_ID_FIELD']
        self.iid_field = config['ITEM_ID_FIELD']
        self.inter_feat = inter_feat
        self.user_feat = user_feat
        self._num = {
            self.uid_field: int(n_users if n_users is not None else int(inter_feat[self.uid_field].max()) + 1),
            self.iid_field: int(n_items if n_items is not None else int(inter_feat[self.iid_field].max()) + 1),
        }

    def num(self, field):
        if field in self._num:
            return self._num[field]
        col = self.user_feat[field] if self.user_feat is not None and field in self.user_feat else self.inter_feat[field]
        return int(col.max()) + 1

    @property
    def user_num(self):
        return self._num[self.uid_field]

    @property
    def item_num(self):
        return self._num[self.iid_field]

    def get_user_feature(self):
        if self.user_feat is None:
            return Interaction({self.uid_field: torch.arange(self.user_num)})
        return self.user_feat

    def __len__(self):
        return len(self.inter_feat)

    def __getitem__(self, index):
        return self.inter_feat[index]

    def inter_matrix(self, form='coo', value_field=None):
        """dataset.py:1596-1651 (`_create_sparse_matrix`): the user x item interaction matrix as scipy sparse, entries 1 or
        the values of `value_field` (FairGo builds its rating-weighted graph from it, fairgo_pmf.py:102-129)."""
        import scipy.sparse as sp
        u = self.inter_feat[self.uid_field].cpu().numpy()
        i = self.inter_feat[self.iid_field].cpu().numpy()
        if value_field is None:
            data = np.ones(len(u))
        else:
            if value_field not in self.inter_feat:
                raise ValueError(f'Value_field [{value_field}] should be one of `df_feat`\'s features.')
            data = self.inter_feat[value_field].cpu().numpy()
        mat = sp.coo_matrix((data, (u, i)), shape=(self.user_num, self.item_num))
        if form == 'coo':
            return mat
        if form == 'csr':
            return mat.tocsr()
        raise NotImplementedError(f'Sparse matrix format [{form}] has not been implemented.')

    def to(self, device):
        """Keep the interaction and user-feature columns resident on `device` (the batch feed then never leaves it)."""
        self.inter_feat = self.inter_feat.to(device)
        if self.user_feat is not None:
            self.user_feat = self.user_feat.to(device)
        return self

    def sort(self, by, ascending=True):
        self.inter_feat.sort(by=by, ascending=ascending)

    def shuffle(self):
        self.inter_feat.shuffle()

    def join(self, inter: Interaction) -> Interaction:
        """Attach the user feature columns to a batch (what dataset.py:1256-1269 does for the sst attribute)."""
        if self.user_feat is None:
            return inter
        uid = inter[self.uid_field].long()
        for k in self.user_feat.columns:
            if k != self.uid_field and k not in inter:
                col = self.user_feat[k]
                inter[k] = col[uid.to(col.device)].to(uid.device)
        return inter


def synthetic_dataset(config, n_users, n_items, n_inter, seed=2020, item_dist="uniform", sst_field="gender"):
    """Deterministic synthetic (user, item, rating, sensitive-attr) interactions, SURVEY.md §8-d: row 0 is [PAD]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    gender = (torch.rand(n_users, generator=g) < 0.5).to(torch.float32)
    gender[0] = 0.0
    u = torch.randint(1, n_users, (n_inter,), generator=g, dtype=torch.int64)
    if item_dist == "zipf":
        x = torch.rand(n_inter, generator=g)
        i = ((n_items - 1) * x * x).floor().to(torch.int64) + 1
    else:
        i = torch.randint(1, n_items, (n_inter,), generator=g, dtype=torch.int64)
    r = torch.randint(1, 6, (n_inter,), generator=g).to(torch.float32)
    inter = Interaction({config['USER_ID_FIELD']: u, config['ITEM_ID_FIELD']: i, config['RATING_FIELD']: r})
    users = Interaction({config['USER_ID_FIELD']: torch.arange(n_users), sst_field: gender})
    return InteractionDataset(config, inter, users, n_users, n_items)
