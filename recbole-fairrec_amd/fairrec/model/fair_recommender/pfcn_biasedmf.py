"""PFCN_BiasedMF: PFCN on a biased MF base model (reference: recbole/model/fair_recommender/pfcn_biasedmf.py),
including its [B] + [B,1] -> [B,B] training-score broadcast, computed without the matrix."""
from .pfcn_base import PFCNBase


class PFCN_BiasedMF(PFCNBase):
    biased = True
