import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
from fairrec import _C
from fairrec.functional import _ws
B = 8192
a = torch.randn(B, device="cuda"); c = torch.randn(B, device="cuda") * 0.1
loss = torch.empty(1, device="cuda"); da, dc = torch.empty_like(a), torch.empty_like(c)
ws = _ws(_C.lib().fr_bpr_workspace_bytes(B, 1), a.device)
def run():
    _C.check(_C.lib().fr_bpr_outer(a.data_ptr(), c.data_ptr(), B, loss.data_ptr(), da.data_ptr(), dc.data_ptr(),
                                   ws.data_ptr(), ws.numel(), _C.current_stream()), "fr_bpr_outer")
for _ in range(5): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"fr_bpr_outer B={B}: {dt*1e6:.1f} us ({B*B/dt/1e9:.1f} G pairs/s), loss {float(loss):.5f}")
