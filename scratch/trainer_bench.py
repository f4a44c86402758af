"""Throughput of the FOCF Trainer path (dataloader -> Trainer._train_epoch -> FusedLazyAdam) at the bench's sizes."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
from fairrec.config import Config
from fairrec.data.dataloader import TrainDataLoader
from fairrec.data.dataset import synthetic_dataset
from fairrec.utils import get_model, get_trainer, init_seed
NU, NI, B, STEPS = 1_000_001, 100_001, 8192, int(os.environ.get("STEPS", 300))
on_dev = sys.argv[1:] == ["device"]
cfg = Config(model="FOCF", config_dict={"embedding_size": 64, "train_batch_size": B, "device": "cuda", "epochs": 1,
                                        "fair_objective": "value", "weight_decay": 1e-3, "eval_step": 0,
                                        "checkpoint_dir": "/tmp/ck", "sst_attr_list": ["gender"]})
init_seed(2020)
ds = synthetic_dataset(cfg, NU, NI, B * STEPS, seed=1)
if on_dev: ds.to("cuda")
model = get_model("FOCF")(cfg, ds).to("cuda")
trainer = get_trainer(None, "FOCF")(cfg, model)
dl = TrainDataLoader(cfg, ds, shuffle=False)
trainer._train_epoch(dl, 0)          # warm-up epoch
torch.cuda.synchronize(); t0 = time.perf_counter()
loss = trainer._train_epoch(dl, 1)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"FOCF Trainer epoch ({'device' if on_dev else 'host'}-resident dataset): {STEPS} steps, {dt / STEPS * 1e6:.1f} us/step, "
      f"{B * STEPS / dt / 1e6:.1f} M interactions/s, loss {loss}")
if os.environ.get("PROFILE") == "1":
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    trainer._train_epoch(dl, 2); torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
