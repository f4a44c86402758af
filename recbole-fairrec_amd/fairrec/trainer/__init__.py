from .trainer import (AbstractTrainer, PFCN_BiasedMFTrainer, PFCN_PMFTrainer, PFCNTrainer,  # noqa: F401
                      Trainer)
