"""Probe: the d128 PFCN golden run with the fused BatchNorm MLP vs the layered form: per-step losses, per-tensor differences."""
import os, sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, os.path.join(ROOT, "recbole-fairrec_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_pfcn_hip import _DS, _load_mlp
from fairrec.config import Config
from fairrec.data.interaction import Interaction
from fairrec.optim import FusedLazyAdam
from fairrec.utils import get_model

z = np.load(os.path.join(ROOT, "tests/golden/pfcn_bmf_sm_d128.npz"))


def run(layered, nsteps=4):
    if layered:
        os.environ["FAIRREC_BN_LAYERED"] = "1"
    else:
        os.environ.pop("FAIRREC_BN_LAYERED", None)
    name, mode = str(z["model"]), str(z["mode"])
    attrs = [str(a) for a in z["attrs"]]
    lr, wd, dis_weight, p = (float(x) for x in z["hyper"])
    n_users, D = z["init.model.user_embedding_layer.weight"].shape
    n_items = z["init.model.item_embedding_layer.weight"].shape[0]
    cfg = Config(model=name, config_dict={"embedding_size": D, "sst_attr_list": attrs, "filter_mode": mode,
                                          "dis_hidden_size_list": [int(h) for h in z["dis_hidden"]], "dis_dropout": p,
                                          "dis_weight": dis_weight, "device": "cuda", "dropout": 0.0,
                                          "mlp_hidden_size_list": [8, 4], "num_layers": 2, "mlp_dropout": 0.0,
                                          "mlp_activation": "relu", "dis_activation": "leakyrelu", "activation": "leakyrelu",
                                          "row_sharded": False})
    model = get_model(name)(cfg, _DS(n_users, n_items, z))
    model.load_state_dict({k[11:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.model.")})
    model = model.to("cuda")
    for i, mlp in model.filter_layer.items():
        _load_mlp(mlp, z, f"init.filter.{i}")
    for s, mlp in model.dis_layer_dict.items():
        _load_mlp(mlp, z, f"init.dis.{s}")
    eng = model.hip_engine()
    opt_f = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=2, group="filter")
    opt_d = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=2, group="dis")
    n_dis = len(z["dis_hidden"]) + 1
    losses, grads = [], {}
    for t, ph in enumerate(str(x) for x in z["phases"][:nsteps]):
        u = z["user_id"][t]
        inter = Interaction({"user_id": torch.tensor(u), "item_id": torch.tensor(z["item_id"][t]),
                             "neg_item_id": torch.tensor(z["neg_item_id"][t]), "gender": torch.tensor(z["gender"][u]),
                             "age": torch.tensor(z["age"][u])}).to("cuda")
        sl = [s for s in str(z["sst_lists"][t]).split(",") if s]
        for s in sl:
            model.dis_layer_dict[s].forced_masks = [torch.tensor(z[f"mask.{s}.{t}.{l}"]) for l in range(n_dis)]
        opt = opt_f if ph == "F" else opt_d
        opt.zero_grad()
        loss = model.calculate_loss(inter, sl) if ph == "F" else model.calculate_dis_loss(inter, sl)
        losses.append(float(loss))
        loss.backward()
        for i, mlp in model.filter_layer.items():
            for n, q in mlp.named_parameters():
                if q.grad is not None:
                    grads[f"{t}.filter.{i}.{n}"] = q.grad.detach().clone()
        for s, mlp in model.dis_layer_dict.items():
            for n, q in mlp.named_parameters():
                if q.grad is not None:
                    grads[f"{t}.dis.{s}.{n}"] = q.grad.detach().clone()
        opt.step()
    st = {"model." + k: v.detach().clone() for k, v in model.state_dict().items()}
    for i, mlp in model.filter_layer.items():
        st.update({f"filter.{i}.{k}": v.detach().clone() for k, v in mlp.state_dict().items()})
    for s, mlp in model.dis_layer_dict.items():
        st.update({f"dis.{s}.{k}": v.detach().clone() for k, v in mlp.state_dict().items()})
    return losses, st, grads


for nsteps in (2, 4):
    la, sa, ga = run(False, nsteps)
    lb, sb, gb = run(True, nsteps)
    print("steps", nsteps, "losses chain  ", la)
    print("steps", nsteps, "losses layered", lb, "golden", list(z["loss"][:nsteps]))
    for k in ga:
        d = (ga[k] - gb[k]).abs()
        s = float(gb[k].abs().max())
        if float(d.max()) > 1e-4 * s:
            print("  grad %-44s scale %.3e maxdiff %.3e frac>1e-4 %.4f" % (k, s, float(d.max()), float((d > 1e-4 * s).float().mean())))
    for k in sa:
        if sa[k].dtype != torch.float32:
            continue
        d = (sa[k] - sb[k]).abs()
        if float(d.max()) > 2e-5:
            print("  state %-44s maxdiff %.3e n>2e-5 %d / %d" % (k, float(d.max()), int((d > 2e-5).sum()), d.numel()))
