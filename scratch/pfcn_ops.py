"""Which aten ops (and from which Python lines) a PFCN filter step launches: torch.profiler on one eager step."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "scratch")]
sys.argv = [sys.argv[0], "none"]
import bench_models as bm
from fairrec.config import Config
from fairrec.data.interaction import Interaction
from fairrec.optim import FusedLazyAdam
from fairrec.utils import get_model
nu, ni, D, B = 1_000_001, 100_001, 128, 8192
cfg = Config(model="PFCN_BiasedMF", config_dict={"embedding_size": D, "device": "cuda", "filter_mode": "sm"})
ds = bm.DS(nu, ni)
m = get_model("PFCN_BiasedMF")(cfg, ds).to("cuda"); m.train()
of = FusedLazyAdam(m.hip_engine(), lr=1e-3, weight_decay=1e-4, group="filter")
od = FusedLazyAdam(m.hip_engine(), lr=1e-3, weight_decay=1e-4, group="dis")
sl = ["gender"]
data = bm.batches(nu, ni, B, 4, pair=True)
inters = []
for d in data:
    d = dict(d); d["gender"] = ds._uf["gender"][d["user_id"]]
    inters.append(Interaction(d).to("cuda"))
def step(k, which):
    opt, fn = (of, lambda it: m.calculate_loss(it, sl)) if which == "F" else (od, lambda it: m.calculate_dis_loss(it, sl))
    opt.zero_grad(); loss = fn(inters[k % 4]); loss.backward(); opt.step()
for k in range(3): step(k, "F"); step(k, "D")
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
for which in ("F", "D"):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(5, which); torch.cuda.synchronize()
    print("=====", which)
    print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=40, max_src_column_width=90))
