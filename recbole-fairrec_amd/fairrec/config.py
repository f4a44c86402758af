"""`Config`: dict-like view of the hyper-parameters the hot path consumes.

Same lookup contract as recbole/config/configurator.py (config[k] is None for a missing key, :405-409; priority
later-overrides-earlier: built-in defaults < model yaml < config files < config_dict), without the dataset /
argv machinery (out of scope).  Keys are the reference's (SURVEY.md §8-b list).
"""
from __future__ import annotations

import os
import re
from typing import Dict, Iterable, Optional

import torch
import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))


def _yaml_loader():
    loader = yaml.FullLoader
    # accept 1e-6 style floats like the reference's custom resolver (configurator.py:90-104)
    loader.add_implicit_resolver(
        u'tag:yaml.org,2002:float',
        re.compile(u'''^(?:[-+]?(?:[0-9][0-9_]*)\\.[0-9_]*(?:[eE][-+]?[0-9]+)?|[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\\.[0-9_]+(?:[eE][-+][0-9]+)?|[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+\\.[0-9_]*|[-+]?\\.(?:inf|Inf|INF)
                    |\\.(?:nan|NaN|NAN))$''', re.X), list(u'-+0123456789.'))
    return loader


# the reference's `smaller_metrics` (evaluator/register.py:62: every metric class with `smaller = True`)
SMALLER_METRICS = frozenset(("mae", "rmse", "logloss", "averagepopularity", "giniindex", "nonparityunfairness",
                             "valueunfairness", "absoluteunfairness", "underunfairness", "overunfairness",
                             "differentialfairness"))


class Config:
    def __init__(self, model: Optional[str] = None, dataset: Optional[str] = None,
                 config_file_list: Optional[Iterable[str]] = None, config_dict: Optional[Dict] = None):
        self.final_config_dict: Dict = {}
        self._load(os.path.join(_HERE, "properties", "overall.yaml"))
        if model is not None:
            name = model if isinstance(model, str) else model.__name__
            self._load(os.path.join(_HERE, "properties", "model", name + ".yaml"), required=False)
            self.final_config_dict["model"] = name
        for f in config_file_list or ():
            self._load(f)
        self.final_config_dict.update(config_dict or {})
        if dataset is not None:
            self.final_config_dict["dataset"] = dataset
        self._derive()

    def _load(self, path, required=True):
        if not os.path.exists(path):
            if required:
                raise FileNotFoundError(path)
            return
        with open(path, "r", encoding="utf-8") as f:
            self.final_config_dict.update(yaml.load(f.read(), Loader=_yaml_loader()) or {})

    def _derive(self):
        d = self.final_config_dict
        use_gpu = d.get("use_gpu", True)
        if "device" not in d or d["device"] is None:
            d["device"] = torch.device("cuda" if torch.cuda.is_available() and use_gpu else "cpu")
        elif isinstance(d["device"], str):
            d["device"] = torch.device(d["device"])
        if isinstance(d.get("valid_metric"), str):
            # configurator.py:306-307: always derived from the metric's name (a config cannot set it): False exactly for
            # the metrics the reference's classes declare `smaller = True` (evaluator/metrics.py)
            d["valid_metric_bigger"] = d["valid_metric"].split("@")[0].lower() not in SMALLER_METRICS
        self._set_train_neg_sample_args()
        if d.get("MODEL_INPUT_TYPE") is None and d.get("model") is not None:   # configurator.py:274-275
            try:
                from .utils import get_model
                d["MODEL_INPUT_TYPE"] = get_model(d["model"]).input_type
            except (ValueError, ImportError):
                pass

    def _set_train_neg_sample_args(self):
        """configurator.py:350-373: `neg_sampling: {uniform: 1}` -> train_neg_sample_args."""
        d = self.final_config_dict
        neg_sampling = d.get("neg_sampling")
        if neg_sampling is None:
            d["train_neg_sample_args"] = {"strategy": "none"}
            return
        if not isinstance(neg_sampling, dict):
            raise ValueError(f"neg_sampling:[{neg_sampling}] should be a dict.")
        distribution = list(neg_sampling.keys())[0]
        if distribution not in ["uniform", "popularity"]:
            raise ValueError(f"The distribution [{distribution}] of neg_sampling should in ['uniform', 'popularity']")
        d["train_neg_sample_args"] = {"strategy": "by", "by": neg_sampling[distribution], "distribution": distribution,
                                      "dynamic": neg_sampling.get("dynamic", "none")}

    def __getitem__(self, item):
        return self.final_config_dict.get(item, None)

    def __setitem__(self, key, value):
        if not isinstance(key, str):
            raise TypeError("index must be a str.")
        self.final_config_dict[key] = value

    def __contains__(self, key):
        return key in self.final_config_dict

    def __str__(self):
        return "\n".join(f"{k} = {v}" for k, v in self.final_config_dict.items())

    __repr__ = __str__
