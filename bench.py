#!/usr/bin/env python3
"""bench.py -- training interactions/s of the FOCF hot path on MI355X (BASELINE.json configs[1]).

One "step" = one pass of the hot path (sort -> lazy-Adam gather + dot -> fairness term -> loss ->
backward + Adam) over one batch of B synthetic (user, item, rating, sensitive-attr) tuples that are
already resident in HBM.  Workload: FOCF, fair_objective=value, 1 000 001 users x 100 001 items,
embedding_size 64, B = 8192 per GPU, Adam lr 1e-3 with the reference's coupled weight_decay 1e-3
(SURVEY.md §8-d cfg 2; uniform-random pairs = the figure of record).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: spawns torch.distributed.run itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --workload nfcf100m [--gpus N]               (BASELINE.json configs[4]: NFCF finetune, tables row-sharded)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "recbole-fairrec_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_USERS, N_ITEMS, DIM, BATCH = 1_000_001, 100_001, 64, 8192
LR, WD, FAIR_WEIGHT, OBJECTIVE = 1e-3, 1e-3, 1.0, "value"
# idx 8*2 + scalars 4*2 + gather 4*D*2 + Adam read m,v 8*D*2 + write p,m,v 12*D*2
ALGO_BYTES_PER_INTERACTION = 16 + 8 + 4 * DIM * 2 + 8 * DIM * 2 + 12 * DIM * 2   # = 3096 at D = 64
SEED = 2020
RANKS_SEEN = 1           # N > 1: what an RCCL all-reduce of ones returned right after the process group came up (main)


def synth_batches(n_batches, batch, n_users, n_items, seed, item_dist="uniform"):
    """Deterministic synthetic interactions: row 0 is [PAD] and never sampled."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    gender = (torch.rand(n_users, generator=g) < 0.5).to(torch.float32)
    u = torch.randint(1, n_users, (n_batches, batch), generator=g, dtype=torch.int64)
    if item_dist == "zipf":
        x = torch.rand((n_batches, batch), generator=g)
        i = ((n_items - 1) * x * x).floor().to(torch.int64) + 1
    elif item_dist == "grouped":
        # the shape of FOCFDataLoader's item-complete batches (focf_dataloader.py:37-51): a batch is the
        # union of ~100-interaction item histories, K ~ 82 distinct items, hot item rows
        per = 100
        k_items = batch // per + (1 if batch % per else 0)
        picks = torch.randint(1, n_items, (n_batches, k_items), generator=g, dtype=torch.int64)
        i = picks.repeat_interleave(per, dim=1)[:, :batch].contiguous()
    elif item_dist == "unique":
        # no row occurs twice in a batch (diagnostic: every interaction takes the kernel's common path)
        i = torch.stack([torch.randperm(n_items - 1, generator=g)[:batch] + 1 for _ in range(n_batches)])
        u = torch.stack([torch.randperm(n_users - 1, generator=g)[:batch] + 1 for _ in range(n_batches)])
    else:
        i = torch.randint(1, n_items, (n_batches, batch), generator=g, dtype=torch.int64)
    r = torch.randint(1, 6, (n_batches, batch), generator=g).to(torch.float32)
    s = gender[u]
    return u, i, r, s


def focf_shape_block(item_dist, K, W, dev, sweep):
    """The same FOCF step on another batch shape, measured the same way as the headline (fresh engine, optimizer state aged by
    one sweep period, K steps issued by the library's step loop, 256 per call): `grouped` = item-complete batches, the shape the
    reference's own FOCFDataLoader feeds (focf_dataloader.py:37-51; SURVEY.md section 8-d says this run "must also be reported"),
    `zipf` = popularity-skewed items.  Bytes: SURVEY.md section 8-d's UNIQUE-ROW definition for these shapes --
    8 R B + 4 (1 + S) B + sum over tables of distinct rows x D x (4 gathered + 8 Adam state read + 12 written)."""
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam
    U, I = xavier_tables(N_USERS, N_ITEMS, DIM, SEED, dev)
    eng = FocfEngine(U, I, OBJECTIVE, FAIR_WEIGHT, 5.0)
    FusedLazyAdam(eng, lr=LR, weight_decay=WD, sweep_period=sweep)
    eng.defer_loss = True
    eng.item_runs = item_dist == "grouped"          # what the Trainer sets when it is fed by FOCFDataLoader
    ahead = FocfEngine.LOW_WATER + FocfEngine.GROUP
    n_age = eng._sweep(BATCH) if (sweep is None or sweep > 0) else 256
    n = n_age + W + K + W + K + ahead + 4
    u, i, r, s = (t.to(dev) for t in synth_batches(n, BATCH, N_USERS, N_ITEMS, SEED + 31337, item_dist))
    def run(lo, hi):
        for a in range(lo, hi, 256):
            b = min(a + 256, hi)
            eng.steps_many(u[a:b].reshape(-1), i[a:b].reshape(-1), r[a:b].reshape(-1), s[a:b].reshape(-1), BATCH)
    assert eng.can_step_many()
    run(0, n_age + W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n_age + W, n_age + W + K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # ... and the next K steps as per-batch launches captured in ONE hipGraph (how rounds 2-5 timed these shapes)
    base = n_age + W + K
    rows = [(u[j], i[j], s[j], r[j]) for j in range(n)]

    def step(k):
        eng.forward(u[k], i[k], r[k], s[k], next_batch=rows[k + 1:k + 1 + ahead] or None)
        eng.backward_adam()
    for k in range(base, base + W):
        step(k)
    torch.cuda.synchronize()
    eng.prepared_is_complete()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            for k in range(base + W, base + W + K):
                step(k)
            eng.join_prepared()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    graph.replay()
    torch.cuda.synchronize()
    dt_graph = time.perf_counter() - t0
    graph.reset()
    del graph, rows
    eng.check_device_errors()
    lo, hi = n_age + W, n_age + W + K
    uniq_u = sum(int(torch.unique(u[k]).numel()) for k in range(lo, hi)) / K
    uniq_i = sum(int(torch.unique(i[k]).numel()) for k in range(lo, hi)) / K
    step_bytes = 8 * 2 * BATCH + 4 * 2 * BATCH + (uniq_u + uniq_i) * DIM * (4 + 8 + 12)
    gbs = step_bytes / (dt / K) / 1e9
    kind = ("one launch per step (fr_focf_step_runs_pipe: the item runs of the previous batch beside the gather of this one)"
            if eng.item_runs and eng.RUNS and eng.PIPE else
            "two launches per step (fr_focf_step_runs: the gather, then a workgroup per item run)" if eng.item_runs and eng.RUNS
            else ("one launch per step (fr_focf_step_staged)" if eng.staged and not eng.item_runs else "three-launch chain"))
    eng.finish()
    del eng, U, I, u, i, r, s
    torch.cuda.synchronize()
    return {"item_distribution": item_dist, "steps": K, "us_per_step": round(dt / K * 1e6, 2), "launch": "library step loop, 256 steps per call",
            "hipGraph_us_per_step": round(dt_graph / K * 1e6, 2),
            "interactions_per_s": round(K * BATCH / dt, 1), "step": kind,
            "distinct_user_rows_per_batch": round(uniq_u, 1), "distinct_item_rows_per_batch": round(uniq_i, 1),
            "bytes_definition": "SURVEY.md 8-d unique rows: 8*R*B + 4*(1+S)*B + distinct rows * D * (4 + 8 + 12)",
            "algorithmic_bytes_per_step": int(step_bytes), "bytes_per_interaction": round(step_bytes / BATCH, 1),
            "achieved_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}


def trainer_fit_block(dev, headline_us_per_step, steps=1024, sweep=None):
    """What a user of the plugin surface gets: `fairrec.trainer.Trainer._train_epoch` (reference trainer.py:155-204) on the
    BASELINE sizes, through the loaders -- not the bench's own step loop.  `uniform`: TrainDataLoader over a device-resident,
    per-epoch shuffled interaction table; the trainer hands runs of `train_steps_per_call` batches to the library
    (TrainDataLoader.take -> FOCF.train_steps -> fr_focf_steps_many).  `item_complete`: the reference's own batch shape,
    FOCFDataLoader over ~100 interactions per item; the epoch's item picks (host numpy, the reference's draw order) are
    composed before the clock starts and reported separately.  One warm-up epoch first (allocations; it also ages the
    lazy-Adam state over more than a sweep period), then one timed epoch bracketed by synchronisations."""
    from fairrec.config import Config
    from fairrec.data.dataloader import FOCFDataLoader, TrainDataLoader
    from fairrec.data.dataset import InteractionDataset, synthetic_dataset
    from fairrec.data.interaction import Interaction
    from fairrec.utils import get_model, get_trainer, init_seed
    import tempfile
    out = {"path": "Trainer._train_epoch -> loader -> model -> FusedLazyAdam (the plugin surface), one warm-up epoch, one timed"}
    cfg = Config(model="FOCF", config_dict={"embedding_size": DIM, "train_batch_size": BATCH, "device": str(dev), "epochs": 1,
                                            "fair_objective": OBJECTIVE, "fair_weight": FAIR_WEIGHT, "weight_decay": WD,
                                            "learning_rate": LR, "eval_step": 0, "checkpoint_dir": tempfile.mkdtemp(),
                                            "sst_attr_list": ["gender"], "lazy_adam_sweep_period": sweep})

    def timed_epoch(loader, before=None):
        init_seed(SEED)
        model = get_model("FOCF")(cfg, loader.dataset).to(dev)
        trainer = get_trainer(None, "FOCF")(cfg, model)
        trainer._train_epoch(loader, 0)
        if before is not None:
            before()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trainer._train_epoch(loader, 1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n, rows = model.hip_engine().U.step // 2, len(loader.dataset)
        del trainer, model
        torch.cuda.empty_cache()
        return dt, n, rows

    ds = synthetic_dataset(cfg, N_USERS, N_ITEMS, BATCH * steps, seed=SEED + 1).to(dev)
    dt, n, rows = timed_epoch(TrainDataLoader(cfg, ds, shuffle=True))
    out["uniform"] = {"loader": "TrainDataLoader(shuffle=True), dataset resident on the device", "steps": n,
                      "us_per_step": round(dt / n * 1e6, 2), "interactions_per_s": round(rows / dt, 1),
                      "steps_per_library_call": int(cfg["train_steps_per_call"]),
                      "vs_headline": round(dt / n * 1e6 / headline_us_per_step, 3)}
    del ds
    # item-complete: 20 000 of the items carry ~100 interactions each (the shape of SURVEY.md 8-d's grouped run): an epoch of
    # ~250 batches -- rounds 4-5 timed 62 (5000 items), where the epoch's fixed costs (the first group's prepare, the last
    # batches' drain, one host synchronisation) were ~5 us of every step
    g = torch.Generator(device="cpu").manual_seed(SEED + 2)
    n_hot, per = 20000, 100
    hot = torch.randperm(N_ITEMS - 1, generator=g)[:n_hot] + 1
    i = hot.repeat_interleave(per)
    u = torch.randint(1, N_USERS, (i.numel(),), generator=g, dtype=torch.int64)
    r = torch.randint(1, 6, (i.numel(),), generator=g).to(torch.float32)
    gender = (torch.rand(N_USERS, generator=g) < 0.5).to(torch.float32)
    ds = InteractionDataset(cfg, Interaction({"user_id": u, "item_id": i, "rating": r}),
                            Interaction({"user_id": torch.arange(N_USERS), "gender": gender}), N_USERS, N_ITEMS).to(dev)
    loader = FOCFDataLoader(cfg, ds)
    compose = {}

    def precompose():
        t0 = time.perf_counter()
        loader._begin_epoch()
        torch.cuda.synchronize()
        compose["s"] = time.perf_counter() - t0
    dt, n, rows = timed_epoch(loader, precompose)
    out["item_complete"] = {"loader": "FOCFDataLoader (focf_dataloader.py:37-51), %d items x %d interactions, dataset resident "
                                      "on the device; the epoch's item picks composed before the clock starts" % (n_hot, per),
                            "steps": n, "us_per_step": round(dt / n * 1e6, 2), "interactions_per_s": round(rows / dt, 1),
                            "rows_per_batch": round(rows / n, 1),
                            "picks_us_per_batch_host": round(compose["s"] / n * 1e6, 1)}
    return out


def next_rows_block(dev, n_inter=2_000_000):
    """The rows either side of the step, timed at the BASELINE sizes: the ranking evaluation a `fit`
    runs after every epoch (`Trainer.evaluate` on a NegSampleEvalDataLoader, `eval_args.mode: uni100`: 100 sampled negatives
    per positive, scores, top-k, the twelve metrics of the reference's test.yaml) and the device negative sampler on its own
    (bit-exact numpy stream, sampler.py:145-197).  A 1024-step training epoch of the headline takes ~29 ms."""
    from fairrec.config import Config
    from fairrec.data.dataloader import NegSampleEvalDataLoader
    from fairrec.data.dataset import synthetic_dataset
    from fairrec.quick_start import split_dataset
    from fairrec.sampler import Sampler
    from fairrec.utils import get_model, get_trainer, init_seed
    import tempfile
    per_batch_users = 4096
    cfg = Config(model="FOCF", config_dict={
        "embedding_size": DIM, "train_batch_size": BATCH, "device": str(dev), "epochs": 1, "fair_objective": OBJECTIVE,
        "fair_weight": FAIR_WEIGHT, "weight_decay": WD, "learning_rate": LR, "checkpoint_dir": tempfile.mkdtemp(),
        "sst_attr_list": ["gender"], "eval_args": {"split": {"RS": [8, 1, 1]}, "group_by": "user", "order": "RO", "mode": "uni100"},
        "metrics": ["NDCG", "Recall", "Hit", "MRR", "DifferentialFairness", "GiniIndex", "PopularityPercentage", "ValueUnfairness",
                    "AbsoluteUnfairness", "UnderUnfairness", "OverUnfairness", "NonParityUnfairness"],
        "valid_metric": "NDCG@5", "topk": [5], "popularity_ratio": 0.1, "eval_batch_size": per_batch_users * 101, "eval_step": 1})
    init_seed(SEED)
    ds = synthetic_dataset(cfg, N_USERS, N_ITEMS, n_inter, seed=SEED + 3)
    train_set, valid_set, test_set = split_dataset(ds)
    phases = Sampler(["train", "valid", "test"], [train_set, valid_set, test_set], "uniform", device=dev)
    valid = NegSampleEvalDataLoader(cfg, valid_set, phases.set_phase("valid"))
    model = get_model("FOCF")(cfg, train_set).to(dev)
    trainer = get_trainer(None, "FOCF")(cfg, model)
    import types
    trainer._train_data_for_eval = types.SimpleNamespace(dataset=train_set)      # (what fit() hands the exposure metrics)
    trainer.evaluate(valid)                        # warm-up: allocations, the lazy tables' flush
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = trainer.evaluate(valid)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    users, rows = len(valid.uid_list), len(valid_set) * 101
    out = {"eval_uni100": {"path": "Trainer.evaluate -> NegSampleEvalDataLoader (device sampler) -> FOCF.predict -> Collector -> Evaluator",
                           "users": users, "positives": len(valid_set), "scored_rows": rows, "batches": len(valid),
                           "users_per_batch": valid.step, "ms_per_evaluation": round(dt * 1e3, 2),
                           "ms_per_batch": round(dt / max(len(valid), 1) * 1e3, 3), "users_per_s": round(users / dt, 1),
                           "scored_rows_per_s": round(rows / dt, 1), "ndcg@5": float(res["ndcg@5"])}}
    # the sampler on its own: 100 negatives for each of 8192 x 16 users per call, used-item sets of the training phase
    smp = phases.set_phase("train")
    g = torch.Generator(device="cpu").manual_seed(SEED + 4)
    uids = torch.randint(1, N_USERS, (BATCH * 16,), generator=g).to(dev)
    smp.sample_by_user_ids(uids, uids, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        neg = smp.sample_by_user_ids(uids, uids, 100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["sampler"] = {"path": "Sampler.sample_by_user_ids (fr_sample_negatives: numpy's MT19937 stream + masked rejection on the device, "
                              "re-draw rounds against the users' used-item sets)",
                      "item_num": N_ITEMS, "negatives_per_call": int(neg.numel()), "ms_per_call": round(dt / reps * 1e3, 3),
                      "negatives_per_s": round(reps * neg.numel() / dt, 1)}
    del trainer, model, valid, phases
    torch.cuda.empty_cache()
    return out


def other_workloads_block(args, dev, budget_s=None):
    """BASELINE.json configs[2], [3], [4] under the same clock as the FOCF line (the driver runs ONE command): each workload of
    `--workload pfcn10m | fairgo10m | nfcf100m` at its full table size, >= 20 timed steps, reduced to the figures a reader
    compares -- ms per step, interactions/s, the roofline it is bound by and the fraction reached.  A workload that would not
    fit the time budget (FAIRREC_BENCH_WORKLOADS_BUDGET seconds for all three, default 240; FairGo's one-off host preprocessing
    of its 400 M-entry propagation matrix alone takes ~75 s) or that fails is reported with `skipped_reason`, not dropped."""
    import argparse
    import bench_workloads
    budget = float(os.environ.get("FAIRREC_BENCH_WORKLOADS_BUDGET", "240")) if budget_s is None else budget_s
    t_all = time.perf_counter()
    expect = {"pfcn10m": 45.0, "nfcf100m": 60.0, "fairgo10m": 120.0}      # seconds a workload takes here, set-up included
    out = {}
    for name in ("pfcn10m", "nfcf100m", "fairgo10m"):
        left = budget - (time.perf_counter() - t_all)
        if left < expect[name]:
            out[name] = {"skipped_reason": f"{left:.0f} s of the {budget:.0f} s budget left, the workload takes ~{expect[name]:.0f} s "
                                           f"(python bench.py --workload {name} runs it on its own)"}
            continue
        a = argparse.Namespace(**vars(args))
        a.workload, a.steps, a.warmup, a.age = name, 20, 4, 0
        if name == "fairgo10m":
            a.steps, a.warmup = 5, 2
        t0 = time.perf_counter()
        try:
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            d = (bench_nfcf(a, 0, 1, dev) if name == "nfcf100m" else
                 (bench_workloads.bench_pfcn if name == "pfcn10m" else bench_workloads.bench_fairgo)(a, dev))
            r = d["roofline"]
            out[name] = {"workload": d["config"]["workload"], "steps": d["steps"], "ms_per_step": d["ms_per_step"],
                         "interactions_per_s": d["value"], "launch": d["config"].get("launch"),
                         "roofline": {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_of_valu_floor",
                                                            "whole_step_tflops") if r.get(k) is not None},
                         "passes_ms": {k: d["config"][k] for k in ("filter_pass_ms", "dis_pass_ms") if k in d["config"]} or None,
                         "peak_mem_GiB": d["config"].get("peak_mem_GiB"), "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:      # e.g. a box whose GPU cannot hold the tables: the FOCF line still goes out
            out[name] = {"skipped_reason": f"{type(e).__name__}: {e}", "wall_s": round(time.perf_counter() - t0, 1)}
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    return out


def stream_copy_ceiling(device, n_bytes=1 << 30, reps=10):
    """On-box streaming ceiling: device-to-device copy of 1 GiB, read + write bytes per second."""
    src = torch.empty(n_bytes // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    return 2.0 * n_bytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9


def xavier_tables(n_users, n_items, dim, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    U = torch.randn(n_users, dim, generator=g) * math.sqrt(2.0 / (n_users + dim))
    I = torch.randn(n_items, dim, generator=g) * math.sqrt(2.0 / (n_items + dim))
    return U.to(device), I.to(device)


def cpu_baseline(budget_s=20.0):
    """The reference's CPU step (oracle restatement + stock dense torch.optim.Adam) on the SAME workload,
    timed on this host's cores on a bounded sample of steps."""
    from oracle.focf import CpuTrainerBaseline   # checker / baseline only
    # the dense Adam sweep is memory-bound: more threads than memory channels only add contention
    threads = int(os.environ.get("FAIRREC_CPU_THREADS", min(torch.get_num_threads(), 32)))
    base = CpuTrainerBaseline(N_USERS, N_ITEMS, DIM, LR, WD, FAIR_WEIGHT, OBJECTIVE, seed=SEED, threads=threads)
    u, i, r, s = synth_batches(64, BATCH, N_USERS, N_ITEMS, SEED)
    for k in range(2):
        base.step(u[k], i[k], r[k], s[k])
    t0 = time.perf_counter()
    n = 0
    while n < 60 and (time.perf_counter() - t0) < budget_s:
        base.step(u[2 + n], i[2 + n], r[2 + n], s[2 + n])
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n * BATCH / dt, 1), "unit": "interactions/s", "cores": threads, "kind": "port",
            "sample": f"{n} steps of B={BATCH} on the full {N_USERS}x{N_ITEMS}x{DIM} tables, "
                      f"oracle FOCF step + dense torch.optim.Adam, {dt:.1f} s"}


NFCF_ALGO_BYTES = 16 + 8 + 4 * 256 * 2 + 8 * 256 * 1 + 12 * 256 * 1      # SURVEY.md §8-d cfg 5 finetune = 7192 B / interaction


def bench_nfcf(args, rank, world, dev):
    """BASELINE.json configs[4]: NFCF finetune (fair_weight 0.1, mlp_hidden_size [128, 64], user table frozen, item table
    lazy Adam) at --nfcf-users x --nfcf-items, D = 256, B = 8192 per rank; with more than one rank both tables are
    row-sharded over the ranks (fairrec/sharded_engine.py: 3 all-to-alls per lookup + one flat all-reduce of the dense
    gradients).  One step = zero_grad, calculate_loss, backward, optimizer.step() through the plugin surface."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.optim import FusedLazyAdam
    from fairrec.utils import get_model
    nu, ni, D, B = args.nfcf_users, args.nfcf_items, 256, BATCH
    K, W = args.steps, args.warmup

    class DS:
        inter_feat = {"rating": torch.tensor([1.0, 5.0])}

        def num(self, f):
            return {"user_id": nu, "item_id": ni}[f]

        def get_user_feature(self):
            return Interaction({"user_id": torch.arange(4), "gender": torch.tensor([0., 1., 0., 1.])})

    cfg = Config(model="NFCF", config_dict={"embedding_size": D, "device": dev, "load_pretrain_path": None, "fair_weight": 0.1,
                                            "mlp_hidden_size": [128, 64], "row_sharded": world > 1,
                                            "graph_train_step": False})
    torch.manual_seed(SEED)          # the replicated scorer MLP must start identical on every rank
    with torch.device(dev):
        m = get_model("NFCF")(cfg, DS())
    m = m.to(dev).train()
    m.load_pretrain_path = "finetune"                 # the finetune branch of calculate_loss (differential fairness term)
    m.user_embedding.weight.requires_grad = False     # what reset_params leaves behind: the projected user table is frozen
    opt = FusedLazyAdam(m.hip_engine(), lr=LR, weight_decay=1e-6)
    g = torch.Generator().manual_seed(SEED + 1 + rank)
    data = []
    for _ in range(16):
        u = torch.randint(1, nu, (B,), generator=g)
        r = torch.randint(1, 6, (B,), generator=g).float()
        data.append(Interaction({"user_id": u, "item_id": torch.randint(1, ni, (B,), generator=g), "rating": r,
                                 "label": (r >= 3).float(), "gender": (u % 2).float()}).to(dev))

    # single GPU: the trainers' default launch mode (`graph_train_step: True`: the optimizer step captured once as a hipGraph,
    # fairrec/graph.py; capture happens inside the warm-up).  Row-sharded ranks launch eagerly.
    graphed = None
    if world == 1 and not args.no_graph and W >= 4:
        from fairrec.graph import GraphedStep
        graphed = GraphedStep(m.hip_engine(), opt, m.calculate_loss, eager_steps=2)

    def step(k):
        if graphed is not None:
            return graphed(data[k % len(data)])
        opt.zero_grad()
        loss = m.calculate_loss(data[k % len(data)])
        loss.backward()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Age the optimizer state first (set-up, not measured), as the FOCF workload does: a lazily updated row replays the
    # steps since it was last touched, at most one sweep period (n_rows / B steps) -- a fresh table has nothing to replay,
    # which would flatter every step before the first full sweep (1221 steps for the 10 M-row item table).
    n_age = 0
    if getattr(args, "age", 0) >= 0 and world == 1:
        eng_ = m.hip_engine()
        n_age = getattr(args, "age", 0) or max([t.default_sweep(B) for t in eng_._tables.values() if t.trainable] + [0])
        for k in range(n_age):
            step(k)
    for k in range(W):
        step(n_age + k)
    barrier()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        loss = step(n_age + k)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    m.hip_engine().check_device_errors()
    total = K * B * world
    achieved = NFCF_ALGO_BYTES * B / (dt / K) / 1e9
    # the lazy-Adam replay of the trainable (item) table: one row-step per row and step in steady state, 30 SIMD cycles per
    # 64-element fragment at D >= 128 -- the floor an exact-parity step has whatever its memory traffic
    rows_local = sum(t.n_rows for t in m.hip_engine()._tables.values() if t.trainable)
    valu_floor_ms = rows_local * (D / 64) * 30 / (1024 * 2.4e9) * 1e3
    detail = None
    if world == 1:      # per-kernel picture (eager pass with the library's HIP-event profiler, after the timed region)
        import bench_workloads as BW

        def eager(k):
            opt.zero_grad()
            l_ = m.calculate_loss(data[k % len(data)])
            l_.backward()
            opt.step()
        prof = BW._profiled(eager, min(K, 10), n_age + W + K)
        detail = dict(BW._gemm_summary(prof), kernels=BW._kernel_table(prof))
    # Which roof the step sits under depends on the table: every trainable row replays one step per step in steady state (the
    # exact lazy Adam), which at 10 M x 256 is 0.49 ms of VALU issue against 0.007 ms of HBM time for the batch's
    # algorithmic bytes -- there the step is bound by the replay and the HBM fraction says nothing; at 100 k rows the replay is
    # 5 us and the step is the sum of its latency-bound launches, priced against HBM as before.
    hbm_ms = NFCF_ALGO_BYTES * B / (HBM_PEAK_GBS * 1e9) * 1e3
    hbm_block = {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                 "algorithmic_bytes_per_launch": NFCF_ALGO_BYTES * B}
    if valu_floor_ms > hbm_ms:
        rs_peak = 1024 * 2.4e9 / (30 * D / 64)          # row-steps per second the chip's SIMDs can issue
        rs_ach = rows_local / (dt / K)
        roof = {"bound": "valu (exact replay)", "kernel": "table_apply_grad_kernel (gradient rows + the sweep's replayed rows)",
                "achieved": round(rs_ach / 1e9, 3), "peak": round(rs_peak / 1e9, 3), "unit": "G row-steps/s",
                "frac": round(rs_ach / rs_peak, 4), "traffic": None, "hbm": hbm_block}
    else:
        roof = dict({"bound": "hbm", "kernel": "whole step (no dominant kernel: gather, MLP, loss, apply)", "traffic": None},
                    **hbm_block)
    roof.update({"valu_floor_ms_per_step": round(valu_floor_ms, 4), "frac_of_valu_floor": round(valu_floor_ms / (dt / K * 1e3), 4),
                 "gemm": detail})
    out = {
        "metric": "training interactions/sec + achieved HBM GB/s, NFCF finetune emb=256 (BASELINE.json configs[4])",
        "value": round(total / dt, 1), "unit": "interactions/s", "n_gpus": RANKS_SEEN, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"ranks_seen": RANKS_SEEN,
                   "workload": f"NFCF finetune, {nu} users x {ni} items, embedding_size={D}, B={B} per GPU, user table frozen, "
                               "item table lazy Adam lr=1e-3 wd=1e-6, mlp [512,128,64,1], fair_weight 0.1",
                   "tables": f"row-sharded over {world} ranks (owner = row mod {world}), RCCL all-to-all" if world > 1 else "single GPU",
                   "fairness_term": ("differential fairness on the GLOBAL batch (records to the items' owners, per-group sums back: "
                                     "2 all-to-alls); the scorer MLP has no BatchNorm") if world > 1 else "single device",
                   "global_batch": B * world, "launch": "hipGraph step" if graphed is not None else "eager",
                   "aged_steps": n_age,
                   "final_loss": round(float(loss.detach()), 6),
                   "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
        "roofline": roof,
    }
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process --
    before anything in this process has touched the GPU -- relay what they print (rank 0's JSON line) and leave with
    their exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


def dry_launch(args, rank, world):
    """Launch-path check without a GPU: the ranks rendezvous over gloo, all-reduce one number, rank 0 reports."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(1) * (rank + 1)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks_seen": world, "sum_of_ranks": float(t.item()),
                          "workload": args.workload}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="focf", choices=["focf", "nfcf100m", "pfcn10m", "fairgo10m"],
                    help="focf = BASELINE.json configs[1] (the headline metric); nfcf100m = configs[4], NFCF finetune at "
                         "100 000 001 x 10 000 001, D = 256, both tables row-sharded over the ranks; pfcn10m = configs[2] "
                         "(PFCN_BiasedMF sm, MFMA roofline per pass); fairgo10m = configs[3] on one GPU (SpMM GB/s and GEMM "
                         "TFLOP/s) -- the last two in bench_workloads.py")
    ap.add_argument("--users", type=int, default=0, help="pfcn10m / fairgo10m: override the number of users (smaller runs)")
    ap.add_argument("--items", type=int, default=0, help="pfcn10m / fairgo10m: override the number of items")
    ap.add_argument("--dry-launch", action="store_true", help="only prove the N-rank launch path (gloo, no GPU)")
    ap.add_argument("--nfcf-users", type=int, default=100_000_001)
    ap.add_argument("--nfcf-items", type=int, default=10_000_001)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--launch", default="auto", choices=["auto", "library", "graph", "eager"],
                    help="how the timed steps are issued on one GPU: the library's step loop (fr_focf_steps_many / fr_focf_runs_many: "
                         "what Trainer._train_epoch calls), one hipGraph of per-batch steps, or per-batch eager launches.  auto "
                         "(default): the library loop when --steps >= 64, else the hipGraph -- a library call has a fixed cost "
                         "(host preparation of the run + three stage launches, ~0.1 ms) that the trainer spreads over runs of 256 "
                         "steps and a 20-step timed region cannot")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly (= --launch eager; the N > 1 path: no hipGraph)")
    ap.add_argument("--graph-only", action="store_true", help="do not also time the other launch modes (single-GPU default: all three)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shapes", action="store_true", help="skip the grouped / zipf blocks of the default FOCF run")
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the other_workloads block of the default FOCF run (BASELINE configs[2..4] at full size, ~3 min)")
    ap.add_argument("--item-dist", default="uniform", choices=["uniform", "zipf", "grouped", "unique"])
    ap.add_argument("--sweep", type=int, default=None, help="lazy-Adam sweep period (default: auto)")
    ap.add_argument("--force-sharded", action="store_true", help="use the row-sharded engine even on one GPU")
    ap.add_argument("--force-fused", action="store_true",
                    help="--item-dist grouped: take the one-launch step although the batches are item-complete (the engine "
                         "takes the three-launch chain for them by default: ~100 interactions share every item row)")
    ap.add_argument("--age", type=int, default=0,
                    help="optimizer steps run before the warm-up so that the lazy-Adam staleness is stationary "
                         "(default: one sweep period)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.dry_launch:
        return dry_launch(args, rank, world)
    if os.environ.get("FAIRREC_BENCH_SHARE_GPU") == "1":   # test rig only: several ranks on one GPU (if RCCL lets them)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import threading
        import torch.distributed as dist

        # a multi-rank run that stops making progress (a collective that never completes) must not hold the node: report
        # and leave after FAIRREC_BENCH_DEADLINE seconds (default 10 min; the whole run takes about one)
        def _deadline():
            print(f"[bench] rank {rank}: no result after the deadline, giving up", file=sys.stderr, flush=True)
            os._exit(3)

        t = threading.Timer(float(os.environ.get("FAIRREC_BENCH_DEADLINE", "600")), _deadline)
        t.daemon = True
        t.start()
        os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "4096")     # the flight recorder: quiesce_rccl's event-based wait
        dist.init_process_group("nccl", device_id=dev)
        # prove the world this run is in: every rank adds one over RCCL; `n_gpus` in the report IS this number
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)
        global RANKS_SEEN
        RANKS_SEEN = ranks_seen = int(seen.item())
        if ranks_seen != args.gpus or dist.get_world_size() != args.gpus:
            print(f"[bench] --gpus {args.gpus} but the RCCL all-reduce saw {ranks_seen} ranks (world size "
                  f"{dist.get_world_size()})", file=sys.stderr, flush=True)
            os._exit(4)

    from fairrec import _C
    from fairrec.model.fair_recommender.focf import FocfEngine
    from fairrec.optim import FusedLazyAdam

    if args.workload in ("pfcn10m", "fairgo10m"):
        if world != 1:
            raise SystemExit(f"--workload {args.workload} is a single-GPU measurement")
        import bench_workloads
        out = (bench_workloads.bench_pfcn if args.workload == "pfcn10m" else bench_workloads.bench_fairgo)(args, dev)
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
        return
    if args.workload == "nfcf100m":
        out = bench_nfcf(args, rank, world, dev)
        import ctypes
        ctypes.CDLL(None).fflush(None)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            sys.stderr.flush()
            os._exit(0)
        return

    K, W = args.steps, args.warmup
    sharded = world > 1 or args.force_sharded
    # every rank feeds its own B interactions per step; N > 1: ONE optimizer step on the global batch of
    # world*B interactions, tables row-sharded over the ranks (fairrec/sharded.py)
    # single GPU: PIPE more batches than are stepped on, so that the look-ahead queue stays full to the last timed step (the
    # timed region then prepares exactly K batches ahead while it applies K, as any K consecutive steps of an epoch do)
    AHEAD = FocfEngine.LOW_WATER + FocfEngine.GROUP      # batches a dataloader-style queue announces ahead
    PIPE = 0 if (world > 1 or args.force_sharded) else AHEAD + 4
    u, i, r, s = (t.to(dev) for t in synth_batches(K + W + PIPE, BATCH, N_USERS, N_ITEMS, SEED + rank, args.item_dist))
    if not sharded:
        U, I = xavier_tables(N_USERS, N_ITEMS, DIM, SEED, dev)
        eng = FocfEngine(U, I, OBJECTIVE, FAIR_WEIGHT, 5.0)
        FusedLazyAdam(eng, lr=LR, weight_decay=WD, sweep_period=args.sweep)
        eng.defer_loss = True     # every forward below is followed by backward_adam; the loss is read at the end
        eng.item_runs = args.item_dist == "grouped" and not args.force_fused    # what the Trainer sets when fed by FOCFDataLoader
    else:
        from fairrec.sharded import ShardedFocfEngine, ShardedFocfEngineV2, shard_rows
        if not torch.distributed.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29655")
            os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "4096")
            torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        g = torch.Generator(device="cpu").manual_seed(SEED + 1 + 1000 * rank)
        Us = (torch.randn(shard_rows(N_USERS, rank, world), DIM, generator=g) * math.sqrt(2.0 / (N_USERS + DIM))).to(dev)
        Is = (torch.randn(shard_rows(N_ITEMS, rank, world), DIM, generator=g) * math.sqrt(2.0 / (N_ITEMS + DIM))).to(dev)
        # Two exchange schedules.  "requester": 5 collectives per step, 4 of them on the step's dependent chain.  "item_owner":
        # 2 on the chain, but 6 in all (3 run a step ahead).  Collectives of one communicator execute one after the other
        # whatever stream issued them, so a step costs (number of collectives) x (collective time) + the kernels between the
        # dependent ones: until the item-owner schedule's scalars ride in its other exchanges the requester
        # schedule has one collective less, and it is the default.  FAIRREC_SHARD_SCHEDULE=item_owner selects the other.
        schedule = os.environ.get("FAIRREC_SHARD_SCHEDULE", "requester")
        if schedule not in ("item_owner", "requester"):
            raise SystemExit("FAIRREC_SHARD_SCHEDULE must be item_owner or requester")
        Eng = ShardedFocfEngineV2 if schedule == "item_owner" else ShardedFocfEngine
        eng = Eng(Us, Is, OBJECTIVE, FAIR_WEIGHT, LR, WD, sweep_period=args.sweep)
        eng.defer_loss = True       # (V2) the step loop reads no loss: its all-reduce runs after the update

    n_batches = u.shape[0]

    _rows = {}

    def coming(k, ub, ib, sb, rb, stop=None):
        # the next batches, dataloader-style prefetch queue: the engine sorts their id columns ahead on a side stream,
        # GROUP batches per launch (the row views are made once per tensor, not once per step)
        key = id(ub)
        if key not in _rows:
            _rows[key] = [(ub[j], ib[j], sb[j], rb[j]) for j in range(ub.shape[0])]
        rows = _rows[key]
        hi = min(k + 1 + AHEAD, len(rows), stop if stop is not None else len(rows))
        return rows[k + 1:hi] or None

    def step(k, ub=None, ib=None, sb=None, rb=None):
        if ub is None:
            ub, ib, sb, rb = u, i, s, r
        if sharded:   # look-ahead of the index work, not across the warm-up / captured-graph boundary
            nxt = (ub[k + 1], ib[k + 1], sb[k + 1], rb[k + 1]) if k + 1 < ub.shape[0] and k != W - 1 else None
            eng.forward(ub[k], ib[k], rb[k], sb[k], next_batch=nxt)
        else:         # the dataloader-style queue runs through: the warm-up steps already announce the first timed batches
            eng.forward(ub[k], ib[k], rb[k], sb[k], next_batch=coming(k, ub, ib, sb, rb))
        eng.backward_adam()

    # One GPU: the timed steps go through the LIBRARY's step loop -- fr_focf_steps_many (fr_focf_runs_many for item-complete
    # batches), one foreign call per run of up to 256 batches: the launches Trainer._train_epoch issues (TrainDataLoader.take ->
    # FOCF.train_steps), and 1.5-2 us per step faster than a hipGraph replay of the same launches.  `--launch graph | eager` time
    # the per-batch entry points instead; the default run reports all three (config.launch_modes_timed).
    launch_mode = "eager" if args.no_graph else args.launch
    auto_note = None
    if launch_mode == "auto":
        launch_mode = "library" if K >= 64 else "graph"
        if launch_mode == "graph":
            auto_note = (f"--launch auto with {K} timed steps: one hipGraph replay; the library's step loop (the trainer's path, "
                         "timed beside it in launch_modes_timed and over 1024 steps in trainer_fit) pays ~0.1 ms per CALL, "
                         "which its runs of 256 steps absorb and a 20-step region does not")
    if sharded and launch_mode == "library":
        launch_mode = "graph"
    if launch_mode == "library" and not eng.can_step_many():
        launch_mode = "graph"
    LIB_RUN = 256

    def lib_steps(ub, ib, sb, rb, lo, hi):
        B = ub.shape[1]
        for a in range(lo, hi, LIB_RUN):
            b = min(a + LIB_RUN, hi)
            eng.steps_many(ub[a:b].reshape(-1), ib[a:b].reshape(-1), rb[a:b].reshape(-1), sb[a:b].reshape(-1), B)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Age the optimizer state first (set-up, not measured): every row's replay length depends on how long ago it was
    # last touched, and that distribution is stationary only after one full sweep period -- a fresh table has nothing
    # to replay, which would flatter the first `sweep_period` timed steps.
    n_age = 0
    if not sharded and args.age >= 0:
        n_age = args.age if args.age else (eng._sweep(BATCH) if (args.sweep is None or args.sweep > 0) else 256)
        ua, ia, ra, sa = (t.to(dev) for t in synth_batches(n_age, BATCH, N_USERS, N_ITEMS, SEED + 104729 + rank, args.item_dist))
        if launch_mode == "library":
            lib_steps(ua, ia, sa, ra, 0, n_age)
        else:
            for k in range(n_age):
                eng.forward(ua[k], ia[k], ra[k], sa[k], next_batch=coming(k, ua, ia, sa, ra))
                eng.backward_adam()
        torch.cuda.synchronize()
        del ua, ia, ra, sa
    if launch_mode == "library":
        lib_steps(u, i, s, r, 0, W)
    else:
        for k in range(W):
            step(k)
    barrier()
    if not sharded:
        eng.prepared_is_complete()      # (synchronised above) the captured steps do not wait for pre-capture side-stream work

    graph, quiesced = None, None
    if launch_mode == "graph":   # K steps (kernels and, when sharded, the RCCL collectives) captured in one hipGraph
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        if sharded:
            # let the RCCL watchdog thread retire the warm-up collectives before capture begins: it polls their events
            # from another thread, which a capture in progress does not tolerate (event-based: fairrec.graph.quiesce_rccl
            # watches torch's flight recorder until every earlier collective is marked retired)
            from fairrec.graph import quiesce_rccl
            quiesced = quiesce_rccl()
        try:
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    for k in range(W, W + K):
                        step(k)
                    if not sharded:
                        eng.join_prepared()     # the queue runs past the timed steps: its last fork is joined here
        except Exception as e:   # e.g. a collective that refuses capture: fall back to eager launches
            if not sharded:
                raise
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
            # forget the look-ahead that was in flight inside the failed capture and its (now invalid) side stream
            eng._prep_key, eng._armed = None, False
            if hasattr(eng.ops, "_side"):
                del eng.ops._side
        torch.cuda.current_stream().wait_stream(side)

    # (the captured steps carry their optimizer step numbers and stamps as kernel arguments: this graph is replayed ONCE)
    barrier()
    t0 = time.perf_counter()
    if launch_mode == "library":
        lib_steps(u, i, s, r, W, W + K)
    elif graph is not None:
        graph.replay()
    else:
        for k in range(W, W + K):
            step(k)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    eng.check_device_errors()
    loss_last = float(eng.loss_ring[eng.loss_slot][0].item()) if not sharded else float("nan")
    launch = {"library": "library step loop (fr_focf_runs_many)" if (not sharded and eng.item_runs) else "library step loop (fr_focf_steps_many)",
              "graph": "hipGraph" if graph is not None else "eager", "eager": "eager"}[launch_mode]
    other = None
    if not sharded and not args.graph_only and launch_mode != "eager":
        # Single-GPU FOCF: `value` is the launch mode named in config.launch (default: the library's step loop); the same K
        # steps' worth of work in the other launch modes (fresh batches, same distribution) is timed as well and reported in
        # `config.launch_modes_timed`, for orientation only.
        key = {"library": "library_loop_ms_per_step", "graph": "hipGraph_ms_per_step"}[launch_mode]
        other = {key: round(dt / K * 1e3, 5)}

        def timed_graph(ub, ib, sb, rb):      # K per-batch steps behind W warm-up ones, captured in one hipGraph, replayed once
            for k in range(W):
                step(k, ub, ib, sb, rb)
            barrier()
            eng.prepared_is_complete()
            sd = torch.cuda.Stream()
            sd.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(sd):
                with torch.cuda.graph(g, stream=sd, capture_error_mode="thread_local"):
                    for k in range(W, W + K):
                        step(k, ub, ib, sb, rb)
                    eng.join_prepared()
            torch.cuda.current_stream().wait_stream(sd)
            barrier()
            t0 = time.perf_counter()
            g.replay()
            barrier()
            t = time.perf_counter() - t0
            g.reset()
            return t

        if launch_mode != "graph":
            ug, ig, rg, sg = (t.to(dev) for t in synth_batches(K + W + PIPE, BATCH, N_USERS, N_ITEMS, SEED + 6700417 + rank, args.item_dist))
            other["hipGraph_ms_per_step"] = round(timed_graph(ug, ig, sg, rg) / K * 1e3, 5)
            eng.check_device_errors()
            del ug, ig, rg, sg
        # (per-batch eager launches: about one pass in five on a fresh box runs ten times slower than the next one -- 300-450 us per
        # step, host side, seen since round 4 with and without event pairs on the launches; such a pass is taken once more and
        # the line says so)
        retimed = []
        for rep_ in range(2):
            ue, ie, re_, se = (t.to(dev) for t in synth_batches(K, BATCH, N_USERS, N_ITEMS, SEED + 15485863 + rank + 101 * rep_, args.item_dist))
            coming(0, ue, ie, se, re_)      # row views made outside the timed region, as for the graph's batches
            barrier()
            t0 = time.perf_counter()
            for k in range(K):
                eng.forward(ue[k], ie[k], re_[k], se[k], next_batch=coming(k, ue, ie, se, re_))
                eng.backward_adam()
            barrier()
            t_e = time.perf_counter() - t0
            eng.check_device_errors()
            if t_e < 3.0 * dt or rep_:
                break
            retimed.append(round(t_e / K * 1e3, 5))
        other["eager_ms_per_step"] = round(t_e / K * 1e3, 5)
        if retimed:
            other["eager_ms_per_step_first_pass_discarded"] = retimed[0]
        if launch_mode != "library" and eng.can_step_many():
            # ... and the same K steps' worth issued by the LIBRARY's own step loop (what Trainer._train_epoch calls; the stage
            # launches for the run's first batches are inside the clock)
            ul, il, rl, sl = (t.to(dev) for t in synth_batches(K + 4, BATCH, N_USERS, N_ITEMS, SEED + 32452843 + rank, args.item_dist))
            lib_steps(ul, il, sl, rl, 0, 4)      # (first use: the run's ring of workspaces is allocated)
            barrier()
            t0 = time.perf_counter()
            lib_steps(ul, il, sl, rl, 4, K + 4)
            barrier()
            other["library_loop_ms_per_step"] = round((time.perf_counter() - t0) / K * 1e3, 5)
            eng.check_device_errors()

    # ---- per-kernel device time: K more steps, eager, with the library's HIP-event profiler -----------
    roofline = None
    # every rank runs the extra steps (the collectives need all ranks); rank 0 reports
    u2, i2, r2, s2 = (t.to(dev) for t in synth_batches(K, BATCH, N_USERS, N_ITEMS, SEED + 7919 + rank, args.item_dist))
    def profiled_pass(sync_every_step):
        _C.prof_reset()
        _C.prof_enable(rank == 0)
        torch.cuda.synchronize()
        t_pass = time.perf_counter()
        for k in range(K):
            if sharded:
                nxt = (u2[k + 1], i2[k + 1], s2[k + 1], r2[k + 1]) if k + 1 < K else None
            else:
                nxt = coming(k, u2, i2, s2, r2)
            eng.forward(u2[k], i2[k], r2[k], s2[k], next_batch=nxt)
            eng.backward_adam()
            if sync_every_step:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        t_pass = time.perf_counter() - t_pass
        _C.prof_enable(False)
        return (_C.prof_read() if rank == 0 else {}), t_pass

    prof, t_pass = profiled_pass(False)
    # On one stream the step kernels run one after the other, so a launch cannot be longer than the step the hipGraph replay
    # just timed.  Seen in about one run in five on a fresh box: the eager pass with event pairs on every launch runs ten
    # times slower (380 us per 32 us kernel -- the runtime's profiling path, not the kernel).  Such a pass is taken again with
    # the queue kept empty (a synchronise after every step); if that does not help either, the kernel's time is taken to be
    # the whole timed step (an upper bound: the step IS that one launch) and the JSON says so.
    events_ok = True
    if rank == 0 and not sharded:
        def step_kernel_us(pr):
            ms, n = pr.get("focf_step_kernel", (0.0, 0))
            if os.environ.get("FAIRREC_BENCH_TEST_EVENT_FLAKE"):     # (exercises the two branches below in a test run)
                ms *= 12
            return ms / n * 1e3 if n else 0.0
        if step_kernel_us(prof) > 1.3 * dt / K * 1e6:
            print(f"[bench] per-launch events inconsistent with the timed step ({step_kernel_us(prof):.1f} us per launch against "
                  f"{dt / K * 1e6:.1f} us per step); measuring again with an empty queue", file=sys.stderr)
            prof, t_pass = profiled_pass(True)
            if step_kernel_us(prof) > 1.3 * dt / K * 1e6:
                events_ok = False
                ms, n = prof["focf_step_kernel"]
                prof["focf_step_kernel"] = (dt / K * 1e3 * n, n)
    if rank == 0:
        per_kernel = {name: ms / n * 1e3 for name, (ms, n) in prof.items()}   # us per launch
        # dominant = the longest kernel of the dependent chain gather -> fair -> backward_adam; sort_segments runs on
        # 2 CUs, one step ahead and concurrently with that chain (fr_focf_prepare), so it is not on the critical path
        # (sort_segments_kernel and focf_lpt_kernel are the look-ahead prepare: side stream, one launch per 8 batches)
        chain = {k: v for k, v in per_kernel.items() if k not in ("sort_segments_kernel", "focf_lpt_kernel")} or per_kernel
        dom = max(chain, key=chain.get)
        algo_bytes = ALGO_BYTES_PER_INTERACTION * BATCH
        achieved = algo_bytes / (per_kernel[dom] * 1e-6) / 1e9
        traffic, traffic_stale = None, None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            traffic = pmc.get(dom)
            # the figure is a stored one (the last profiles/collect.sh pass): say so when the kernel sources have moved since
            import hashlib
            src = os.path.join(ROOT, "recbole-fairrec_amd", "csrc")
            now = hashlib.sha1(b"".join(open(os.path.join(src, f), "rb").read()
                                        for f in ("focf_step.hip", "focf_ws.hpp", "common.hpp"))).hexdigest()
            traffic_stale = pmc.get("_kernel_source_sha1") != now
            if traffic is not None and traffic_stale:
                print("[bench] roofline.traffic comes from profiles/pmc_traffic.json, collected on OTHER kernel sources than "
                      "the ones running now: re-run profiles/collect.sh", file=sys.stderr)
        copy_gbs = stream_copy_ceiling(dev)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    # `traffic` is not measured by this run: it is the PMC figure of the last profiles/collect.sh pass
                    "traffic_source": "profiles/pmc_traffic.json" if traffic is not None else None,
                    "traffic_stale": traffic_stale if traffic is not None else None,
                    "measured_copy_ceiling": round(copy_gbs, 1), "frac_of_measured_ceiling": round(achieved / copy_gbs, 4),
                    "algorithmic_bytes_per_launch": algo_bytes,
                    # ... plus the launch's slice of the bounded-staleness sweep (rows / period, p m v read and written, last
                    # written): traffic the lazy update owes for the rows no batch touched -- what `traffic` is to be held against
                    "sweep_bytes_per_launch": (0 if sharded or not eng._sweep(BATCH) else
                                               int((N_USERS + N_ITEMS) / eng._sweep(BATCH) * (6 * 4 * DIM + 8))),
                    # the same bytes over the WHOLE timed step (every launch of the step + gaps): the honest figure
                    "whole_step_GBps": round(algo_bytes / (dt / K) / 1e9, 1),
                    "frac_step": round(algo_bytes / (dt / K) / 1e9 / HBM_PEAK_GBS, 4),
                    # What actually bounds the kernel: with the reference's weight_decay != 0 every row of both
                    # tables takes one Adam step per training step, replayed lazily without traffic -- (N_users + N_items)
                    # row-steps of 64 elements per launch at 34 SIMD cycles each (two quarter-rate transcendentals are 32 of
                    # them; measured with table_flush_kernel), on 1024 SIMDs at 2.4 GHz.  Informative only: `frac` stays the
                    # HBM figure the contract asks for.
                    "valu_floor_us_per_launch": (None if sharded else
                                                 round((N_USERS + N_ITEMS) * (DIM / 64) * 34 / (1024 * 2.4e9) * 1e6, 2)),
                    "frac_of_valu_floor": (None if sharded else
                                           round((N_USERS + N_ITEMS) * (DIM / 64) * 34 / (1024 * 2.4e9) * 1e6 / per_kernel[dom], 4)),
                    "kernel_us": {k: round(v, 2) for k, v in sorted(per_kernel.items())},
                    "dominant_rule": ("longest kernel of the dependent chain; sort_segments_kernel runs on 1-2 workgroups "
                                      "concurrently with the gather kernels on a side stream" if sharded else
                                      ("the step IS one kernel (focf_step_kernel), and the index work of the two coming "
                                       "batches rides in it as ~100 extra workgroups (claim / place stages); "
                                       "focf_stage_kernel = those stages on their own launch for the first two batches of a loop"
                                       if getattr(eng, "staged", False) else
                                       "the step IS one kernel (focf_step_kernel); sort_segments_kernel + focf_lpt_kernel run "
                                       "twice per 16 steps on a side stream, ahead of the steps they serve")),
                    "measured": ("" if events_ok else "FALLBACK for focf_step_kernel: the timed step itself (per-launch events "
                                 "were inconsistent in this run, twice); otherwise ") +
                                f"hipExtLaunchKernelGGL start/stop events on every launch, {K} eager steps after the "
                                "timed region (same look-ahead sort overlap as the timed steps)"}

    if rank == 0:
        total = K * BATCH * world
        out = {
            "metric": "training interactions/sec + achieved HBM GB/s, FOCF emb=64 at 1/2/4/8 MI355X",
            "value": round(total / dt, 1), "unit": "interactions/s", "n_gpus": RANKS_SEEN, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"ranks_seen": RANKS_SEEN,      # N > 1: counted by an RCCL all-reduce of ones when the group came up
                       "workload": "FOCF fair_objective=value, 1000001 users x 100001 items, embedding_size=64, "
                                   "B=8192 per GPU, Adam lr=1e-3 weight_decay=1e-3 (BASELINE.json configs[1])",
                       "item_distribution": args.item_dist, "launch": launch, "launch_note": auto_note, "launch_modes_timed": other,
                       "rccl_quiesce_before_capture": quiesced,     # how the wait for the watchdog ended (fairrec.graph.quiesce_rccl)
                       "step": (("item-owner-computes: records and user-row requests exchanged one step ahead; gather -> all-to-all(user rows) -> "
                                 "score / fair / grads -> all-to-all(user gradients) -> apply" if sharded and
                                 schedule == "item_owner" else
                                 "gather / fair / backward_adam chain over 5 all-to-alls") if sharded else
                                ("ONE launch per step and nothing else on the device (fr_focf_step_staged: gather + lazy-Adam "
                                 "replay + dot + fairness + backward + Adam + sweep slice + the claim stage of the batch two "
                                 "steps ahead + the place stage of the next one), the run of launches issued by ONE library call "
                                 "per 256 steps (fr_focf_steps_many, the call Trainer._train_epoch makes)"
                                 if getattr(eng, "staged", False) and launch_mode == "library" else
                                 "ONE launch per step and nothing else on the device (fr_focf_step_staged: gather + lazy-Adam "
                                 "replay + dot + fairness + backward + Adam + sweep slice + the claim stage of the batch two "
                                 "steps ahead + the place stage of the next one)" if getattr(eng, "staged", False) else
                                 "ONE launch per step (fr_focf_step: gather + lazy-Adam replay + dot + fairness + backward + Adam + "
                                 "sweep slice); id columns of 16 coming batches sorted and packed per fork of the side stream")),
                       "lazy_adam_sweep_period": eng._sweep(BATCH) if not sharded else args.sweep,
                       "tables": "row-sharded over %d ranks, RCCL all-to-all" % world if sharded else "single GPU",
                       "global_batch": BATCH * world, "final_loss": round(loss_last, 6) if not sharded else None,
                       "aged_steps": None if sharded else n_age},
            "roofline": roofline,
        }
        if world == 1 and not sharded and args.item_dist == "uniform" and not args.no_shapes:
            # the figure of record is the uniform run above; the item-complete shape FOCF's real loader
            # produces (and a popularity-skewed one) are reported next to it, each under the bytes definition it names
            if graph is not None:
                graph.reset()
                graph = None
            del eng, U, I
            torch.cuda.empty_cache()
            # (informational blocks, each timed over its OWN number of steps -- at least 200, whatever K the headline was asked
            # for: a 20-step graph of the item-complete shape carries one side-stream prepare per 8 batches and reads 14 %
            # slower than any stretch of an epoch does; the block says how many steps it timed)
            out["other_batch_shapes"] = [focf_shape_block(d, max(K, 200), W, dev, args.sweep) for d in ("grouped", "zipf")]
            out["trainer_fit"] = trainer_fit_block(dev, dt / K * 1e6, sweep=args.sweep)
            try:
                out["next_rows"] = next_rows_block(dev)
            except Exception as e:      # (orientation figures: a failure here must not cost the line)
                out["next_rows"] = {"error": f"{type(e).__name__}: {e}"}
            if not args.no_workloads:
                out["other_workloads"] = other_workloads_block(args, dev)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
    if (world > 1 or os.environ.get("FAIRREC_BENCH_NFCF_BLOCK") == "1") and not args.no_shapes:      # (the env: this code path on one GPU)
        # BASELINE.json configs[4] next to the FOCF line: the config the >= 6 x scaling target is stated on (NFCF finetune,
        # 100 000 001 x 10 000 001, D = 256, both tables row-sharded over the ranks).  Every rank runs it; rank 0 reports.
        if graph is not None:
            graph.reset()
            graph = None
        del eng
        torch.cuda.empty_cache()
        try:
            nf = bench_nfcf(args, rank, world, dev)
        except Exception as e:          # e.g. a node whose GPUs cannot hold their shard: the FOCF line still goes out
            nf = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            out["nfcf100m"] = nf
    # The JSON line is the LAST thing on stdout: RCCL's banner sits in C stdio buffers until flushed, so flush first.
    barrier()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if graph is not None:
        graph.reset()
        del graph
        torch.cuda.synchronize()
    if world > 1:
        # No process-group teardown on the multi-rank path: destroying an RCCL communicator that captured graphs used
        # blocked for ~20 min in this image (measured with world_size 1), and nothing runs after the report.
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
