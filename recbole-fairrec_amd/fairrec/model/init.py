"""Parameter initialisers with the reference's semantics (recbole/model/init.py:15-31):
xavier-normal for Embedding/Linear weights, zero Linear bias.  One-off host-side torch RNG work."""
import torch.nn as nn
from torch.nn.init import constant_, xavier_normal_, xavier_uniform_


def xavier_normal_initialization(module):
    if isinstance(module, nn.Embedding):
        xavier_normal_(module.weight.data)
    elif isinstance(module, nn.Linear):
        xavier_normal_(module.weight.data)
        if module.bias is not None:
            constant_(module.bias.data, 0)


def xavier_uniform_initialization(module):
    if isinstance(module, nn.Embedding):
        xavier_uniform_(module.weight.data)
    elif isinstance(module, nn.Linear):
        xavier_uniform_(module.weight.data)
        if module.bias is not None:
            constant_(module.bias.data, 0)
