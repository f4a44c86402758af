"""Probe: the fused BatchNorm MLP (csrc/mlp_bn.hip) against the layered form on PFCN's d128 golden shapes."""
import copy, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "recbole-fairrec_amd"))
from fairrec.model.layers import MLPLayers

torch.manual_seed(0)
for M in (200, 8192):
    a = MLPLayers([128, 256, 128], activation="leakyrelu", bn=True, init_method="norm").cuda().train()
    b = copy.deepcopy(a)
    x = torch.randn(M, 128, device="cuda") * 0.1
    for mode in ("no_grad", "grad", "passes2"):
        outs = []
        for mlp, env in ((a, None), (b, "1")):
            if env:
                os.environ["FAIRREC_BN_LAYERED"] = env
            else:
                os.environ.pop("FAIRREC_BN_LAYERED", None)
            if mode == "no_grad":
                with torch.no_grad():
                    y = mlp(x)
            else:
                y = mlp(x.clone().requires_grad_(), passes=2 if mode == "passes2" else 1)
            outs.append(y.detach().clone())
        d = (outs[0] - outs[1]).abs()
        print(M, mode, "max |y| %.3e  max diff %.3e  rows with diff > 1e-6: %d" % (float(outs[1].abs().max()), float(d.max()),
              int((d.max(dim=1).values > 1e-6).sum())), "nan:", bool(torch.isnan(outs[0]).any()))
        for (n1, b1), (n2, b2) in zip(a.named_buffers(), b.named_buffers()):
            if not torch.equal(b1, b2):
                print("   buffer", n1, float((b1.float() - b2.float()).abs().max()))
