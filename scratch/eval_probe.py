"""Where an evaluation batch's time goes (uni100 at the BASELINE sizes): loader, predict, collector, evaluator."""
import os, sys, time, types, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "recbole-fairrec_amd"))
import torch
import bench
from fairrec.config import Config
from fairrec.data.dataloader import NegSampleEvalDataLoader
from fairrec.data.dataset import synthetic_dataset
from fairrec.quick_start import split_dataset
from fairrec.sampler import Sampler
from fairrec.utils import get_model, get_trainer, init_seed
from fairrec.evaluator import Collector, Evaluator
dev = torch.device("cuda")
cfg = Config(model="FOCF", config_dict={
    "embedding_size": 64, "train_batch_size": 8192, "device": str(dev), "epochs": 1, "fair_objective": "value",
    "fair_weight": 1.0, "weight_decay": 1e-3, "learning_rate": 1e-3, "checkpoint_dir": tempfile.mkdtemp(),
    "sst_attr_list": ["gender"], "eval_args": {"split": {"RS": [8, 1, 1]}, "group_by": "user", "order": "RO", "mode": "uni100"},
    "metrics": ["NDCG", "Recall", "Hit", "MRR", "DifferentialFairness", "GiniIndex", "PopularityPercentage", "ValueUnfairness",
                "AbsoluteUnfairness", "UnderUnfairness", "OverUnfairness", "NonParityUnfairness"],
    "valid_metric": "NDCG@5", "topk": [5], "popularity_ratio": 0.1, "eval_batch_size": 4096 * 101, "eval_step": 1})
init_seed(2020)
ds = synthetic_dataset(cfg, bench.N_USERS, bench.N_ITEMS, int(sys.argv[1]) if len(sys.argv) > 1 else 500_000, seed=5)
train_set, valid_set, test_set = split_dataset(ds)
phases = Sampler(["train", "valid", "test"], [train_set, valid_set, test_set], "uniform", device=dev)
valid = NegSampleEvalDataLoader(cfg, valid_set, phases.set_phase("valid"))
model = get_model("FOCF")(cfg, train_set).to(dev)
trainer = get_trainer(None, "FOCF")(cfg, model)
trainer._train_data_for_eval = types.SimpleNamespace(dataset=train_set)
trainer.evaluate(valid)
torch.cuda.synchronize()
T = {"next": 0.0, "predict": 0.0, "collect": 0.0, "final": 0.0}
model.eval()
collector, evaluator = Collector(cfg), Evaluator(cfg)
collector.data_collect(trainer._train_data_for_eval)
it = iter(valid)
nb = 0
with torch.no_grad():
    while True:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            interaction, row_idx, positive_u, positive_i = next(it)
        except StopIteration:
            break
        torch.cuda.synchronize(); t1 = time.perf_counter()
        per = int(cfg['eval_batch_size'])
        scores = torch.cat([model.predict(interaction[lo:lo + per]).view(-1) for lo in range(0, len(interaction), per)])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        collector.eval_batch_collect_candidates(scores, row_idx, interaction, positive_u, positive_i, valid.dataset.item_num)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        T["next"] += t1 - t0; T["predict"] += t2 - t1; T["collect"] += t3 - t2; nb += 1
    t0 = time.perf_counter()
    res = evaluator.evaluate(collector.get_data_struct())
    torch.cuda.synchronize()
    T["final"] = time.perf_counter() - t0
print("batches", nb, "rows per batch", len(valid_set) * 101 // max(nb, 1))
for k, v in T.items():
    print(f"{k:8s} {v * 1e3:9.2f} ms total, {v / max(nb, 1) * 1e3:8.3f} ms per batch")

# ---- inside NegSampleEvalDataLoader.__next__: the sampler call alone, then the rest
import numpy as np
ld = valid
sl = slice(0, ld.step)
uids, P = ld.uid_list[sl], ld.counts[sl]
indptr, used_items, _ = ld.sampler.used_ids
for name in ("sample_calls",):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        neg = ld.sampler.rs.sample_calls(1, ld.dataset.item_num, uids, P * ld.neg_sample_num, indptr, used_items)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per call for {uids.numel()} users, {neg.numel()} values")
