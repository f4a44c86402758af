import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
from fairrec import _C
from fairrec.sampler import DeviceRandomState, Sampler
class DS:
    uid_field, iid_field = "user_id", "item_id"
    def __init__(s, nu, ni, u, i): s.user_num, s.item_num, s.inter_feat = nu, ni, {"user_id": u, "item_id": i}
g = torch.Generator().manual_seed(0)
for nu, ni, per in ((1_000_001, 100_001, 20), (1_000_001, 1_000_001, 20), (944, 1683, 100)):
    u = torch.arange(1, nu).repeat_interleave(per); i = torch.randint(1, ni, (u.numel(),), generator=g)
    rs = DeviceRandomState("cuda", 2020)
    smp = Sampler("train", DS(nu, ni, u, i), device="cuda", random_state=rs).set_phase("train")
    for B in (2048, 8192):
        users = torch.randint(1, nu, (B,), generator=g).cuda()
        for _ in range(3): smp.sample_by_user_ids(users, None, 1)
        _C.prof_reset(); _C.prof_enable(True)
        for _ in range(50): smp.sample_by_user_ids(users, None, 1)
        torch.cuda.synchronize(); _C.prof_enable(False)
        print(nu, ni, per, B, {k: round(ms / n * 1e3, 1) for k, (ms, n) in _C.prof_read().items()}, flush=True)
