"""Device randperm (fr_randperm) against torch.randperm on the host: time per call at the epoch sizes of the bench."""
import os, sys, time
os.environ.setdefault("FAIRREC_RANDPERM_AHEAD", "0")     # one permutation per call: no look-ahead launch beside the timed one
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
from fairrec.sampler.torch_stream import randperm
for n in (1_000_000, 8192 * 1024, 50_000_000):
    torch.manual_seed(0)
    t0 = time.perf_counter(); ref = torch.randperm(n); th = time.perf_counter() - t0
    torch.manual_seed(0)
    randperm(1000, "cuda"); torch.cuda.synchronize()
    torch.manual_seed(0)
    t0 = time.perf_counter(); got = randperm(n, "cuda"); torch.cuda.synchronize(); td = time.perf_counter() - t0
    print(f"n = {n}: host torch.randperm {th * 1e3:.1f} ms, device {td * 1e3:.2f} ms, equal {torch.equal(got.cpu(), ref)}")
