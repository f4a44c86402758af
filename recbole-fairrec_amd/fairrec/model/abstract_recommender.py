"""Plugin base classes -- same names, attributes and method contracts as
recbole/model/abstract_recommender.py:23-103 (AbstractRecommender / FairRecommender), so model code
and trainers written against the reference keep working.  Clean-room: only the contract is shared.
"""
from logging import getLogger

import numpy as np
import torch.nn as nn

from ..utils.enum_type import ModelType


class AbstractRecommender(nn.Module):
    """Contract (reference abstract_recommender.py:23-83): calculate_loss / predict / full_sort_predict,
    other_parameter()/load_other_parameter() for non-tensor state, and a __str__ that reports the
    number of trainable parameters."""

    def __init__(self):
        self.logger = getLogger()
        super().__init__()

    def calculate_loss(self, interaction):
        raise NotImplementedError

    def predict(self, interaction):
        raise NotImplementedError

    def full_sort_predict(self, interaction):
        raise NotImplementedError

    def other_parameter(self):
        names = getattr(self, 'other_parameter_name', None)
        return {k: getattr(self, k) for k in names} if names else dict()

    def load_other_parameter(self, para):
        if para is None:
            return
        for k, v in para.items():
            setattr(self, k, v)

    # hook used by fairrec.trainer.Trainer to build the fused optimizer; models that own HIP engines override
    def hip_engine(self):
        return None

    def load_state_dict(self, state_dict, *args, **kwargs):
        """The reference evaluates / resumes exactly the checkpointed weights (trainer.py:258-284, :478-483).  A lazy-Adam
        table may still hold rows that are behind the optimizer step (any epoch that did not end in a flush): left alone,
        their missed zero-gradient steps would later be replayed ON TOP of the loaded weights.  So bring every row up to
        date first (flush: `last == step`, nothing left to replay), then copy."""
        eng = getattr(self, '_engine', None)
        if eng is not None and hasattr(eng, 'flush'):
            eng.flush()
        return super().load_state_dict(state_dict, *args, **kwargs)

    def __str__(self):
        n = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        return super().__str__() + f'\nTrainable parameters: {n}'


class FairRecommender(AbstractRecommender):
    """Reference abstract_recommender.py:86-103: reads USER_ID/ITEM_ID/NEG_PREFIX field names, the table
    sizes from the dataset and the device from the config."""
    type = ModelType.GENERAL

    def __init__(self, config, dataset):
        super().__init__()
        self.USER_ID = config['USER_ID_FIELD']
        self.ITEM_ID = config['ITEM_ID_FIELD']
        self.POS_ITEM_ID = self.ITEM_ID
        self.NEG_ITEM_ID = config['NEG_PREFIX'] + self.ITEM_ID
        self.n_users = dataset.num(self.USER_ID)
        self.n_items = dataset.num(self.ITEM_ID)
        self.device = config['device']
        # `row_sharded: True` under an initialised torch.distributed world: every embedding table of the model holds only
        # the rows r with r mod world == rank (local row r div world), SURVEY.md §8-e
        self.shard = None
        if config['row_sharded']:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError('row_sharded needs an initialised torch.distributed process group')
            self.shard = (dist.get_rank(), dist.get_world_size())
        # `data_parallel: True`: REPLICATED tables, the batch sharded over the ranks, gradients of the replicated
        # parameters averaged per step (fairrec/replicated_engine.py) -- for the models that read whole tables (FairGo)
        self.replicas = None
        if config['data_parallel']:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError('data_parallel needs an initialised torch.distributed process group')
            if self.shard is not None:
                raise ValueError('row_sharded and data_parallel exclude each other')
            self.replicas = (dist.get_rank(), dist.get_world_size())

    def _table_rows(self, n_rows):
        """Rows of an n_rows table this rank holds."""
        if self.shard is None:
            return n_rows
        rank, world = self.shard
        return (n_rows - rank + world - 1) // world if n_rows > rank else 0
