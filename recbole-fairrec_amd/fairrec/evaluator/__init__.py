from .metrics import Evaluator, fairness_metrics, topk_metrics  # noqa: F401
