from .trainer import AbstractTrainer, Trainer  # noqa: F401
