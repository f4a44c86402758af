from .enum_type import EvaluatorType, FeatureType, InputType, ModelType  # noqa: F401
