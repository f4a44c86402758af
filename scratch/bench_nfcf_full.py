"""NFCF finetune at BASELINE.json cfg 5's FULL size on one MI355X: 100 000 001 x 10 000 001, D = 256, user table frozen
(102 GB resident, no optimizer state), item table trainable (lazy Adam).  Informational."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "scratch")]
from fairrec.config import Config
from fairrec.data.interaction import Interaction
from fairrec.optim import FusedLazyAdam
from fairrec.utils import get_model

nu, ni, D, B = int(os.environ.get("NU", 100_000_001)), int(os.environ.get("NI", 10_000_001)), 256, 8192


class DS:
    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self.gender = (torch.rand(nu, generator=g) < 0.5).float()
        self.inter_feat = {"rating": torch.tensor([1.0, 5.0])}

    def num(self, f):
        return {"user_id": nu, "item_id": ni}[f]

    def get_user_feature(self):
        return Interaction({"user_id": torch.arange(4), "gender": self.gender[:4]})


cfg = Config(model="NFCF", config_dict={"embedding_size": D, "device": "cuda", "load_pretrain_path": None,
                                        "fair_weight": 0.1, "mlp_hidden_size": [128, 64]})
ds = DS()
t0 = time.time()
with torch.device("cuda"):
    m = get_model("NFCF")(cfg, ds)
m = m.to("cuda").train()
m.load_pretrain_path = "finetune"                 # the finetune branch of calculate_loss (differential fairness term)
m.user_embedding.weight.requires_grad = False     # what reset_params does after projecting out the bias direction
opt = FusedLazyAdam(m.hip_engine(), lr=1e-3, weight_decay=1e-6)
torch.cuda.synchronize()
print(f"model + tables on the device in {time.time() - t0:.1f} s, {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
g = torch.Generator().manual_seed(1)
data = []
for _ in range(8):
    u = torch.randint(1, nu, (B,), generator=g)
    r = torch.randint(1, 6, (B,), generator=g).float()
    data.append(Interaction({"user_id": u, "item_id": torch.randint(1, ni, (B,), generator=g), "rating": r,
                             "label": (r >= 3).float(), "gender": ds.gender[u]}).to("cuda"))


def step(k):
    opt.zero_grad()
    loss = m.calculate_loss(data[k % len(data)])
    loss.backward()
    opt.step()
    return loss


for k in range(5):
    step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 40
for k in range(5, 5 + K):
    loss = step(k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"NFCF finetune {nu}x{ni} D={D} B={B}: {dt * 1e3:.3f} ms per step, {B / dt / 1e6:.2f} M interactions/s, "
      f"loss {float(loss):.4f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
