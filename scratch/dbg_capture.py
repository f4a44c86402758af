import os, sys, torch, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29671")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from fairrec.sharded import ShardedFocfEngine
NU, NI, D, B = 100001, 10001, 64, 8192
g = torch.Generator().manual_seed(0)
eng = ShardedFocfEngine((torch.randn(NU, D, generator=g) * 0.01).to(dev), (torch.randn(NI, D, generator=g) * 0.01).to(dev), "value", 0.8, 1e-3, 1e-3)
T = 12
u = torch.randint(1, NU, (T, B), generator=g).to(dev); i = torch.randint(1, NI, (T, B), generator=g).to(dev)
r = torch.randint(1, 6, (T, B), generator=g).float().to(dev); s = (torch.rand(T, B, generator=g) < 0.5).float().to(dev)
mode = sys.argv[1]
def step(k, ahead):
    nxt = (u[k + 1], i[k + 1], s[k + 1]) if ahead and k + 1 < T else None
    eng.forward(u[k], i[k], r[k], s[k], next_batch=nxt); eng.backward_adam()
for k in range(4): step(k, mode != "noahead" and k != 3)
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
        for k in range(4, T):
            step(k, mode == "ahead" and k + 1 < T)
            if mode == "ahead1": pass
print("captured", mode, flush=True)
torch.cuda.current_stream().wait_stream(side)
graph.replay(); torch.cuda.synchronize()
print("replayed", mode, flush=True)
graph.reset(); del graph
os._exit(0)
