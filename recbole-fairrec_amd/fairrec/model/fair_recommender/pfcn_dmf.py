"""PFCN_DMF: PFCN on a two-tower base model with cosine scoring (reference: recbole/model/fair_recommender/pfcn_dmf.py):
`user_mlp` / `item_mlp` towers ([D]*(num_layers+1), mlp_activation, init 'norm') before the filters, scores =
cosine_similarity * 10 in training, sigmoid(cosine) in predict; filters and discriminators use `dis_activation`."""
import torch

from ...functional import RowDot
from ..layers import MLPLayers
from .pfcn_base import PFCNBase


class PFCN_DMF(PFCNBase):
    biased = False

    def _build_base_layers(self, config):
        self.num_layers = config['num_layers']
        self.mlp_dropout = config['mlp_dropout']
        self.mlp_activation = config['mlp_activation']
        self.dis_activation = config['dis_activation']
        D = self.embedding_size
        mk = lambda: MLPLayers(layers=[D] + [D for _ in range(self.num_layers)], dropout=self.mlp_dropout,
                               activation=self.mlp_activation, init_method='norm')
        self.user_mlp, self.item_mlp = mk(), mk()

    def _base_dense_modules(self):
        return {"user_mlp": self.user_mlp, "item_mlp": self.item_mlp}

    def _filter_activation(self):
        return self.dis_activation

    def _dis_activation(self):
        return self.dis_activation

    def _user_tower(self, rows):
        return self.user_mlp(rows)

    def _item_tower(self, rows):
        return self.item_mlp(rows)

    @staticmethod
    def _cosine(a, b):
        """nn.CosineSimilarity(dim=1, eps=1e-8): a.b / (max(|a|, eps) * max(|b|, eps)); the three row dots are HIP
        launches, the [B]-sized combination is elementwise glue."""
        # clamp before the square root: max(|a|, eps) with a zero (not NaN) gradient for an all-zero row (dead ReLU tower)
        na = torch.sqrt(RowDot.apply(a, a).clamp_min(1e-16))
        nb = torch.sqrt(RowDot.apply(b, b).clamp_min(1e-16))
        return RowDot.apply(a, b) / (na * nb)

    def _score(self, user_embed, item_embed):          # pfcn_dmf.py:193-194
        return self._cosine(user_embed, item_embed) * 10

    def _predict_score(self, ue, ie):
        return self._cosine(ue, ie)
