"""fairrec -- MI355X-native training hot path for RecBole-FairRec models (FOCF / PFCN / FairGo / NFCF).

Host side in Python on PyTorch-ROCm (tensors, streams, torch.distributed); all per-step arithmetic in
hand-written HIP kernels behind the C ABI of include/fairrec_hip.h (libfairrec_hip.so).
The class / method / config-key surface mirrors recbole.model.abstract_recommender and
recbole.trainer so that model configs written for the reference drop in unchanged.
"""
__version__ = "0.1.0"
