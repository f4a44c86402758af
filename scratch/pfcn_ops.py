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
import traceback
from collections import Counter
from torch.utils._python_dispatch import TorchDispatchMode
rows = Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in ("view", "detach", "empty", "slice", "select", "reshape", "squeeze", "expand", "as_strided", "t.default", "alias")):
            fr = [f for f in traceback.extract_stack() if "fairrec" in f.filename]
            where = f"{fr[-1].filename.split('recbole-fairrec_amd/')[-1]}:{fr[-1].lineno}" if fr else "(autograd / torch)"
            shape = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            rows[(name, where, shape)] += 1
        return func(*args, **(kwargs or {}))
for which in ("F", "D"):
    rows.clear()
    with Log():
        step(5, which)
    torch.cuda.synchronize()
    print("=====", which, sum(rows.values()), "ops")
    for (n, w, s), c in sorted(rows.items(), key=lambda t: (t[0][1], t[0][0])):
        print(f"{c:4d} {n:36s} {str(s):18s} {w}")
