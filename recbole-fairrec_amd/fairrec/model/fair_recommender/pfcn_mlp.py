"""PFCN_MLP: PFCN on an MLP scorer over cat(user, item) (reference: recbole/model/fair_recommender/pfcn_mlp.py).
Tables are named `user_embedding` / `item_embedding`, the scorer `mlp_layer` (registered, trained by optimizer_filter)."""
from ..layers import MLPLayers
from .pfcn_base import PFCNBase


class PFCN_MLP(PFCNBase):
    biased = False
    user_table_attr = "user_embedding"
    item_table_attr = "item_embedding"

    def _build_base_layers(self, config):
        self.dropout = config['dropout']
        self.mlp_hidden_size_list = config['mlp_hidden_size_list']
        self.mlp_layer = MLPLayers([self.embedding_size * 2] + list(self.mlp_hidden_size_list) + [1], dropout=self.dropout)

    def _base_dense_modules(self):
        return {"mlp_layer": self.mlp_layer}

    def _score(self, user_embed, item_embed):          # pfcn_mlp.py:185-186, [B, 1]; BPR then averages over B
        return self.mlp_layer(user_embed, item_embed).view(-1)

    def _predict_score(self, ue, ie):
        return self.mlp_layer(ue, ie)
