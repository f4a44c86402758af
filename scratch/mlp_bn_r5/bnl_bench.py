"""Per-launch times of the fused BatchNorm MLP (csrc/mlp_bn.hip) against the layered form on PFCN's shapes: eager loops timed
with events, and the library's own per-kernel profiler (fr_prof_*) for the fused form's launches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "recbole-fairrec_amd"))
from fairrec.model.layers import MLPLayers
from fairrec import _C

M = int(os.environ.get("M", 8192))
torch.manual_seed(0)
cases = {"filter": ([128, 256, 128], 0.0), "discriminator": ([128, 128, 256, 128, 128, 64, 32, 1], 0.3)}
for name, (widths, p) in cases.items():
    mlp = MLPLayers(widths, dropout=p, activation="leakyrelu", bn=True, init_method="norm").cuda().train()
    x = torch.randn(M, widths[0], device="cuda")
    w = torch.randn(M, widths[-1], device="cuda")
    for form in ("fused", "layered"):
        if form == "fused":
            os.environ["FAIRREC_BN_FUSED"] = "1"
        else:
            os.environ.pop("FAIRREC_BN_FUSED", None)
        for mode in ("fwd", "fwd+bwd"):
            def one():
                if mode == "fwd":
                    with torch.no_grad():
                        return mlp(x)
                xi = x.detach().requires_grad_()
                (mlp(xi) * w).sum().backward()
            for _ in range(5):
                one()
            # one iteration captured in a hipGraph: launch-to-launch time without the host
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                one()
                with torch.cuda.graph(g, stream=s):
                    one()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                g.replay()
            b.record()
            torch.cuda.synchronize()
            print("%-14s %-8s %-8s %8.1f us per pass (graph replay)" % (name, form, mode, a.elapsed_time(b) * 1e3 / 50), flush=True)
