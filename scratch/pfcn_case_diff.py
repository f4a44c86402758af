"""Final parameters of a recorded PFCN run under FAIRREC_BN_BWD_SEPARATE=1 and without: who moves, by how much."""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo/recbole-fairrec_amd"); sys.path.insert(0, "/root/repo/tests")
import test_pfcn_hip as T
path = sys.argv[1] if len(sys.argv) > 1 else "/root/repo/tests/golden/pfcn_bmf_sm_d128.npz"
z = np.load(path)
def run():
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.optim import FusedLazyAdam
    from fairrec.utils import get_model
    name, mode = str(z["model"]), str(z["mode"])
    attrs = [str(a) for a in z["attrs"]]
    lr, wd, dis_weight, p = (float(x) for x in z["hyper"])
    utab = "user_embedding" if name == "PFCN_MLP" else "user_embedding_layer"
    itab = "item_embedding" if name == "PFCN_MLP" else "item_embedding_layer"
    n_users, D = z[f"init.model.{utab}.weight"].shape
    n_items = z[f"init.model.{itab}.weight"].shape[0]
    cfg = Config(model=name, config_dict={"embedding_size": D, "sst_attr_list": attrs, "filter_mode": mode,
                                          "dis_hidden_size_list": [int(h) for h in z["dis_hidden"]], "dis_dropout": p,
                                          "dis_weight": dis_weight, "device": "cuda", "dropout": 0.0,
                                          "mlp_hidden_size_list": [8, 4], "num_layers": 2, "mlp_dropout": 0.0,
                                          "mlp_activation": "relu", "dis_activation": "leakyrelu", "activation": "leakyrelu",
                                          "row_sharded": False})
    model = get_model(name)(cfg, T._DS(n_users, n_items, z))
    model.load_state_dict({k[11:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.model.")})
    model = model.to("cuda")
    for i, mlp in model.filter_layer.items():
        T._load_mlp(mlp, z, f"init.filter.{i}")
    for s, mlp in model.dis_layer_dict.items():
        T._load_mlp(mlp, z, f"init.dis.{s}")
    eng = model.hip_engine()
    opt_f = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=2, group="filter")
    opt_d = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=2, group="dis")
    n_dis = len(z["dis_hidden"]) + 1
    grads = {}
    for t, ph in enumerate(str(x) for x in z["phases"]):
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
        loss.backward()
        for i, mlp in model.filter_layer.items():
            for n, q in mlp.named_parameters():
                if q.grad is not None:
                    grads[f"t{t}{ph}.filter.{i}.{n}"] = q.grad.clone()
        for s, mlp in model.dis_layer_dict.items():
            for n, q in mlp.named_parameters():
                if q.grad is not None:
                    grads[f"t{t}{ph}.dis.{s}.{n}"] = q.grad.clone()
        opt.step()
    out = {}
    with torch.no_grad():
        ue, _ = model.forward(torch.arange(1, 201).cuda(), torch.arange(1, 201).cuda(), ["gender"])
    out["filtered_user_rows"] = ue.clone()
    for i, mlp in model.filter_layer.items():
        out.update({f"filter.{i}.{n}": q.detach().clone() for n, q in mlp.named_parameters()})
    for s, mlp in model.dis_layer_dict.items():
        out.update({f"dis.{s}.{n}": q.detach().clone() for n, q in mlp.named_parameters()})
    return out, grads
os.environ.pop("FAIRREC_BN_BWD_SEPARATE", None)
a, ga = run()
os.environ["FAIRREC_BN_BWD_SEPARATE"] = "1"
b, gb = run()
print("phases", [str(x) for x in z["phases"]])
for k in ga:
    d = float((ga[k] - gb[k]).abs().max()); s = float(gb[k].abs().max())
    if d > 1e-5 * s:
        print("grad %-44s max|g| %.3e  max diff %.3e" % (k, s, d))
for k in a:
    d = float((a[k] - b[k]).abs().max())
    print("final %-40s max diff %.3e  elements > 1e-4: %d of %d" % (k, d, int(((a[k] - b[k]).abs() > 1e-4).sum()), a[k].numel()))
