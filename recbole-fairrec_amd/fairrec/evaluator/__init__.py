from .collector import Collector  # noqa: F401
from .metrics import Evaluator, fairness_metrics, topk_metrics  # noqa: F401
