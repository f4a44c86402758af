"""Workload for `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ...`: eager filter and discriminator passes of the bench's PFCN
workload (bench.py --workload pfcn10m; smaller tables by default: the dense layers do not depend on the table sizes)."""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import bench_workloads as BW
from fairrec.config import Config
from fairrec.optim import FusedLazyAdam
from fairrec.utils import get_model
nu, ni, D = int(os.environ.get("PMC_USERS", 1_000_001)), int(os.environ.get("PMC_ITEMS", 100_001)), 128
dev = torch.device("cuda")
cfg = Config(model="PFCN_BiasedMF", config_dict={"embedding_size": D, "device": "cuda", "filter_mode": "sm"})
ds = BW._DS(nu, ni)
torch.manual_seed(2020)
m = get_model("PFCN_BiasedMF")(cfg, ds).to(dev)
m.train()
eng = m.hip_engine()
of = FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-4, group="filter")
od = FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-4, group="dis")
data = BW._batches(nu, ni, 8, ds._uf["gender"], dev, pair=True)
for k in range(int(os.environ.get("PMC_STEPS", "24"))):
    for opt, fn in ((of, lambda it: m.calculate_loss(it, ["gender"])), (od, lambda it: m.calculate_dis_loss(it, ["gender"]))):
        opt.zero_grad()
        loss = fn(data[k % len(data)])
        loss.backward()
        opt.step()
torch.cuda.synchronize()
print("done")
