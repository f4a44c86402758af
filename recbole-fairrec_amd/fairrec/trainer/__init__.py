from .trainer import (AbstractTrainer, FairGo_GCNTrainer, FairGo_PMFTrainer, FairGoTrainer,  # noqa: F401
                      PFCN_BiasedMFTrainer, PFCN_DMFTrainer, PFCN_MLPTrainer, PFCN_PMFTrainer, PFCNTrainer, Trainer)
