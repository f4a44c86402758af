"""Does memory of a LazyTable workspace get recycled while the library's side-stream sort still writes to it?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import torch
from fairrec import _C
from fairrec.optim import AdamHyper, LazyTable
import fairrec.optim as O
if os.environ.get("NO_RECORD") == "1":
    O._used_on_side_stream = lambda t: None
dev = torch.device("cuda")
bad = 0
for trial in range(20):
    W = torch.randn(5000, 64, device=dev)
    t = LazyTable(W)
    t.ensure_state()
    hyper = AdamHyper(device=dev, cap=8)
    hyper.configure(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3) if hasattr(hyper, "configure") else None
    idx = torch.randint(0, 5000, (4096,), device=dev)
    a = torch.randn(8192, 8192, device=dev)
    torch.cuda.synchronize()
    for _ in range(6):
        a = a @ a * 1e-4          # ~ tens of ms of queued main-stream work: the sort's fork point lies behind it
    rows = t.gather_train(hyper, idx)
    ws_ptr, ws_n = t._ws.data_ptr(), t._ws.numel()
    del rows, t, idx
    x = torch.full((ws_n,), 7, dtype=torch.uint8, device=dev)      # same size: the allocator's first candidate is the freed workspace
    same = x.data_ptr() == ws_ptr
    torch.cuda.synchronize()
    ok = bool((x == 7).all())
    bad += (not ok)
    print(f"trial {trial}: recycled the workspace address: {same}; pattern intact: {ok}")
print("corrupted:", bad)
