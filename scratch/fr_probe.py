"""Does torch's NCCL flight recorder say when the watchdog has RETIRED the eager collectives (what a capture must wait for)?"""
import os, pickle, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29671")
os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "256")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(1024, device="cuda")
for _ in range(5):
    dist.all_reduce(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
from torch._C._distributed_c10d import _dump_nccl_trace
for k in range(400):
    tr = pickle.loads(_dump_nccl_trace())
    ent = tr.get("entries", [])
    left = [e for e in ent if not e.get("retired", True)]
    if k == 0:
        print("keys", list(tr.keys()), "n entries", len(ent), "fields", sorted(ent[0].keys()) if ent else None)
    if not left:
        print(f"all {len(ent)} retired after {1e3 * (time.perf_counter() - t0):.1f} ms ({k} polls)")
        break
    time.sleep(0.005)
else:
    print("never retired:", left[:2])
import sys; sys.stdout.flush(); os._exit(0)
