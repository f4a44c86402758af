"""Enumerations the plugin surface exposes (mirror of recbole/utils/enum_type.py names and values)."""
from enum import Enum


class ModelType(Enum):
    GENERAL = 1
    SEQUENTIAL = 2
    CONTEXT = 3
    KNOWLEDGE = 4
    TRADITIONAL = 5
    DECISIONTREE = 6


class InputType(Enum):
    POINTWISE = 1
    PAIRWISE = 2
    LISTWISE = 3


class EvaluatorType(Enum):
    RANKING = 1
    VALUE = 2


class FeatureType(Enum):
    TOKEN = 'token'
    FLOAT = 'float'
    TOKEN_SEQ = 'token_seq'
    FLOAT_SEQ = 'float_seq'
