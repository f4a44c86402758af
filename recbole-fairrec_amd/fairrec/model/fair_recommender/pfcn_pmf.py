"""PFCN_PMF: PFCN on a plain MF base model (reference: recbole/model/fair_recommender/pfcn_pmf.py)."""
from .pfcn_base import PFCNBase


class PFCN_PMF(PFCNBase):
    biased = False
