import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
from fairrec import _C
from fairrec.sharded import HipOps
ops = HipOps("cuda")
for M, G in ((8192, 1), (8192, 2), (8192, 8), (1024, 1), (2048, 8)):
    cap = min(M, 2 * M // G + 64, 16384 // G)
    S = 2 * cap + 1
    g = torch.Generator().manual_seed(0)
    a = torch.randint(0, 10**6, (M,), generator=g).cuda()
    b = torch.randint(0, 10**5, (M,), generator=g).cuda()
    aux = torch.rand(M, generator=g).cuda()
    send = torch.empty(G * S, dtype=torch.int64, device="cuda")
    sa = torch.empty(M, dtype=torch.int32, device="cuda"); sb = torch.empty_like(sa)
    cnt = torch.empty(2 * G, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    for which in ("pair", "single"):
        _C.prof_reset(); _C.prof_enable(True)
        for _ in range(50):
            if which == "pair":
                ops.bucket_pair(a, b, G, cap, S, 0, cap, send, sa, sb, cnt, aux, 2 * cap, err)
            else:
                ops.bucket_by_owner(a, G, cap, S, 0, send, sa, cnt[:G], None, 0, err)
        torch.cuda.synchronize(); _C.prof_enable(False)
        print(M, G, which, {k: round(ms / n * 1e3, 2) for k, (ms, n) in _C.prof_read().items()}, flush=True)
