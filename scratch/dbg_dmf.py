import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "tests")]
from test_pfcn_hip import _DS
from fairrec.config import Config
from fairrec.data.interaction import Interaction
from fairrec.optim import FusedLazyAdam
from fairrec.utils import get_model
z = np.load(os.path.join(ROOT, "tests/golden/pfcn_dmf_none.npz"))
cfg = Config(model="PFCN_DMF", config_dict={"embedding_size": 8, "sst_attr_list": ["gender"], "filter_mode": "none", "device": "cuda",
    "num_layers": 2, "mlp_dropout": 0.0, "mlp_activation": "relu", "dis_activation": "leakyrelu", "dis_hidden_size_list": [16, 8]})
model = get_model("PFCN_DMF")(cfg, _DS(40, 30, z))
model.load_state_dict({k[11:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.model.")})
model = model.to("cuda")
eng = model.hip_engine()
opt = FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-3, sweep_period=2)
for t in range(2):
    u = z["user_id"][t]
    inter = Interaction({"user_id": torch.tensor(u), "item_id": torch.tensor(z["item_id"][t]), "neg_item_id": torch.tensor(z["neg_item_id"][t])}).to("cuda")
    opt.zero_grad()
    loss = model.calculate_loss(inter, None)
    print("step", t, "loss", float(loss), "ref", z["loss"][t])
    loss.backward()
    for n, p in model.user_mlp.named_parameters():
        print("  grad", n, None if p.grad is None else (float(p.grad.abs().max()), bool(torch.isnan(p.grad).any())))
    opt.step()
    for n, p in model.user_mlp.named_parameters():
        print("  param", n, float(p.abs().max()), bool(torch.isnan(p).any()), "ref", float(np.abs(z["final.model.user_mlp." + n]).max()))
    print("  U nan", bool(torch.isnan(model.user_embedding_layer.weight).any()))
