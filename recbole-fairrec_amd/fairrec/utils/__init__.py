from .enum_type import EvaluatorType, FeatureType, InputType, ModelType  # noqa: F401
from .utils import (calculate_valid_score, dict2str, early_stopping, ensure_dir, get_local_time,  # noqa: F401
                    get_model, get_trainer, init_seed)
