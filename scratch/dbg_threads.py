import os, time, torch, torch.distributed as dist
def nthreads():
    names = []
    for t in os.listdir("/proc/self/task"):
        try:
            names.append(open(f"/proc/self/task/{t}/comm").read().strip())
        except Exception:
            pass
    return len(names), sorted(set(names))
torch.zeros(1, device="cuda")
print("before init", nthreads())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29633")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(1 << 20, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
print("with group ", nthreads())
dist.destroy_process_group(); torch.cuda.synchronize(); time.sleep(1.0)
print("after destroy", nthreads())
for k in ("NCCL_", "RCCL_", "HSA_", "TORCH_NCCL", "HIP_"):
    print(k, {a: b for a, b in os.environ.items() if a.startswith(k)})
