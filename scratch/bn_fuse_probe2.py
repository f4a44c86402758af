import os, sys, torch
sys.path.insert(0, "/root/repo/recbole-fairrec_amd")
from fairrec.model.layers import MLPLayers
torch.manual_seed(0)
widths = [128, 256, 128, 128, 64, 32]
mlp = MLPLayers(widths + [1], dropout=0.0, activation="leakyrelu", bn=True).cuda().train()
x = torch.randn(8192, widths[0], device="cuda") * 0.05 + 0.3      # nearly constant columns, like trained embeddings
x.requires_grad_()
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
    torch.nn.functional.binary_cross_entropy_with_logits(y.view(-1), (torch.arange(8192, device="cuda") % 2).float()).backward()
    res[mode] = {n: q.grad.clone() for n, q in mlp.named_parameters()}
for n in res["fused"]:
    if n.endswith("weight") and res["fused"][n].dim() == 2:
        a, b = res["fused"][n], res["separate"][n]
        rel = (a - b).abs() / (b.abs() + 1e-12)
        rowmax = (a - b).abs().max(dim=1).values / (b.abs().max(dim=1).values + 1e-30)
        print(n, tuple(a.shape), "max|g| %.3e  max abs diff %.3e  elements with rel diff > 1e-3: %d  worst row rel %.3e" %
              (float(b.abs().max()), float((a - b).abs().max()), int((rel > 1e-3).sum()), float(rowmax.max())))
