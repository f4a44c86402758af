"""Collector for full-sort ranking evaluation on the device: the `full` branch of recbole/evaluator/collector.py:131-205.

Per user batch it receives the masked score matrix [users, n_items] (column 0 and the users' history already -inf,
trainer.py:435-437), the batch's positives and the batch Interaction, and keeps -- as device tensors -- what the metrics
need: `rec.topk` (hit flags of the top-max(topk) items | number of positives), `rec.positive_score`, `data.positive_i`
and `data.<sst>`.  The reference builds a dense [users, items] 0/1 matrix per batch to mark the positives; here the hit
flags come from binary searches in the sorted positive keys (fr_eval_hits).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .. import _C


class Collector:
    def __init__(self, config):
        self.config = config
        topk = config['topk'] or [10]
        self.topk = [topk] if isinstance(topk, int) else list(topk)
        self.sst = list(config['sst_attr_list'] or [])
        mode = (config['eval_args'] or {}).get('mode', 'full')
        if 'full' not in mode:
            raise NotImplementedError(f"evaluation mode [{mode}]: only full-sort evaluation is on the device path")
        self._parts: Dict[str, List[torch.Tensor]] = {}

    def _add(self, key, t):
        self._parts.setdefault(key, []).append(t)

    def eval_batch_collect(self, scores: torch.Tensor, interaction, positive_u: torch.Tensor, positive_i: torch.Tensor):
        lib = _C.lib()
        U, n_items = scores.shape
        K = min(max(self.topk), n_items)
        positive_u, positive_i = positive_u.to(scores.device, torch.int64), positive_i.to(scores.device, torch.int64)
        _, topk_idx = torch.topk(scores, K, dim=-1)
        topk_idx = topk_idx.contiguous()
        keys = torch.sort(positive_u * n_items + positive_i).values
        rec = torch.empty((U, K + 1), dtype=torch.int32, device=scores.device)
        _C.check(lib.fr_eval_hits(topk_idx.data_ptr(), U, K, n_items, keys.data_ptr(), keys.numel(), rec.data_ptr(),
                                  _C.current_stream()), "fr_eval_hits")
        self._add('rec.topk', rec)
        self._add('rec.positive_score', scores[positive_u, positive_i])
        self._add('data.positive_i', positive_i)
        for s in self.sst:
            if s in interaction:
                self._add('data.' + s, interaction[s].to(scores.device)[positive_u])

    def get_data_struct(self) -> Dict[str, torch.Tensor]:
        out = {k: torch.cat(v, dim=0) for k, v in self._parts.items()}
        self._parts = {}
        return out
