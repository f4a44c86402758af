import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
from fairrec import _C
from fairrec.optim import AdamHyper, LazyTable
N, D = 1_000_000, 64
hyper = AdamHyper(lr=1e-3, weight_decay=1e-3, device="cuda")
for D in (64, 128, 256):
  for k in (0, 1, 8, 32, 128, 512):
    tab = LazyTable(torch.randn(N, D, device="cuda") * 0.01)
    tab.ensure_state(); tab.m.normal_(std=1e-3); tab.v.uniform_(1e-7, 1e-5)
    tab.last.fill_(1); tab.step = 1 + k; tab._dirty = True
    tab.flush(hyper)  # warm (does the work once)
    ts = []
    for rep in range(3):
        tab.last.fill_(1); tab._dirty = True
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); tab.flush(hyper); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    t = min(ts)
    wave_steps = N * max(D // 64, 1) * k
    print(f"D={D} k={k:4d}: {t:9.1f} us   bytes/s={(N*D*4*6 + N*8)/t/1e3:7.1f} GB/s" + (f"   cycles per 64-element step per SIMD: {t*1e-6*2.4e9*1024/wave_steps:6.1f}" if k else ""))
