"""fr_linear_bwd_input_bnstats + fr_bn_bwd_ex against the three launches, and both against float64 torch on the CPU."""
import os, sys, copy, torch
sys.path.insert(0, "/root/repo/recbole-fairrec_amd")
from fairrec.model.layers import MLPLayers
torch.manual_seed(0)
for widths in ([128, 256, 128, 64, 32], [64, 32, 32]):
    mlp = MLPLayers(widths + [1], dropout=0.0, activation="leakyrelu", bn=True).cuda().train()
    x = torch.randn(8192, widths[0], device="cuda", requires_grad=True)
    # float64 reference: the same modules in torch on the CPU
    ref = torch.nn.Sequential(*[copy.deepcopy(m) for m in mlp.mlp_layers]).double().cpu().train()
    xr = x.detach().double().cpu().requires_grad_()
    yr = ref(xr)
    (yr * yr).sum().backward()
    refg = [xr.grad] + [q.grad for q in ref.parameters()]
    res = {}
    for mode in ("fused", "separate"):
        if mode == "separate":
            os.environ["FAIRREC_BN_BWD_SEPARATE"] = "1"
        else:
            os.environ.pop("FAIRREC_BN_BWD_SEPARATE", None)
        for q in mlp.parameters():
            q.grad = None
        x.grad = None
        y = mlp(x)
        (y * y).sum().backward()
        res[mode] = [x.grad.clone()] + [q.grad.clone() for q in mlp.parameters()]
    names = ["x"] + [n for n, _ in mlp.named_parameters()]
    for k, n in enumerate(names):
        r = refg[k]
        ef = float((res["fused"][k].double().cpu() - r).abs().max())
        es = float((res["separate"][k].double().cpu() - r).abs().max())
        print("%-28s max|ref| %.3e   err fused %.3e   err separate %.3e" % (n, float(r.abs().max()), ef, es))
