"""Name-based dispatch and small helpers with the contracts of recbole/utils/utils.py
(get_model :51-73, get_trainer :76-94, early_stopping :97-140, calculate_valid_score :143-156,
dict2str :159-169, init_seed :172-189)."""
from __future__ import annotations

import datetime
import importlib
import os
import random

import numpy as np
import torch


def get_local_time():
    return datetime.datetime.now().strftime('%b-%d-%Y_%H-%M-%S')


def ensure_dir(dir_path):
    os.makedirs(dir_path, exist_ok=True)


def get_model(model_name):
    """Model class by name: module `fairrec.model.fair_recommender.<name.lower()>`, attribute `<name>`."""
    mod_name = '.'.join(['fairrec.model.fair_recommender', model_name.lower()])
    if importlib.util.find_spec(mod_name, __name__) is None:
        raise ValueError('`model_name` [{}] is not the name of an existing model.'.format(model_name))
    return getattr(importlib.import_module(mod_name, __name__), model_name)


def get_trainer(model_type, model_name):
    """`<ModelName>Trainer` if fairrec.trainer defines it, else the plain Trainer (reference utils.py:76-94)."""
    trainer_mod = importlib.import_module('fairrec.trainer')
    return getattr(trainer_mod, model_name + 'Trainer', getattr(trainer_mod, 'Trainer'))


def early_stopping(value, best, cur_step, max_step, bigger=True):
    """Returns (best, cur_step, stop_flag, update_flag); strict improvement resets the counter."""
    improved = value > best if bigger else value < best
    if improved:
        return value, 0, False, True
    cur_step += 1
    return best, cur_step, cur_step > max_step, False


def calculate_valid_score(valid_result, valid_metric=None):
    return valid_result[valid_metric] if valid_metric else valid_result['Recall@10']


def dict2str(result_dict):
    return '    '.join(str(metric) + ' : ' + str(value) for metric, value in result_dict.items())


def init_seed(seed, reproducibility=True):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    from ..sampler.sampler import seed_all     # the device mirrors of numpy's global generator
    seed_all(seed)
