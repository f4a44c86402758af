"""bench.py --workload pfcn10m | fairgo10m: BASELINE.json configs[2] and configs[3] through the plugin surface, each with a
`roofline` block on the bound SURVEY.md §8-d names (MFMA for the PFCN discriminator / filter MLPs; SpMM bytes AND GEMM FLOP
for FairGo).  One "step" = one filter pass + one discriminator pass over a batch of B = 8192 synthetic interactions, the two
optimizer passes the reference's trainers alternate (trainer.py:875-930, :684-736), timed separately.

Timing: K steps of each pass, launched as the trainers launch them (`graph_train_step: True`: one hipGraph per optimizer
step, fairrec/graph.py), bracketed by barrier + synchronize.  The roofline figures come from a second, eager pass with the
library's HIP-event profiler (fr_prof_*): per kernel kind the device time and the algorithmic work it stood for
(fr_prof_read_work: 2 M N K FLOP per dense product, 12 nnz + 8 n_rows D bytes per SpMM).
"""
import json
import math
import os
import time

import numpy as np
import torch

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA, dense
HBM_PEAK_GBS = 8000.0
B = 8192
GEMM_KINDS = ("linear_fwd_kernel", "linear_bwd_input_kernel", "linear_bwd_weight_kernel")


class _DS:
    def __init__(self, nu, ni, graph=None):
        from fairrec.data.interaction import Interaction
        self._n = {"user_id": nu, "item_id": ni}
        g = torch.Generator().manual_seed(0)
        self._uf = Interaction({"user_id": torch.arange(nu), "gender": (torch.rand(nu, generator=g) < 0.5).float()})
        self._uf["gender"][1:3] = torch.tensor([0.0, 1.0])
        self.inter_feat = {"rating": torch.tensor([1.0, 5.0])}
        self._graph = graph

    def num(self, f):
        return self._n[f]

    def get_user_feature(self):
        return self._uf

    def inter_matrix(self, form="coo", value_field=None):
        return self._graph


def _batches(nu, ni, T, gender, dev, pair=False, seed=1):
    from fairrec.data.interaction import Interaction
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(T):
        u = torch.randint(1, nu, (B,), generator=g)
        d = {"user_id": u, "item_id": torch.randint(1, ni, (B,), generator=g)}
        if pair:
            d["neg_item_id"] = torch.randint(1, ni, (B,), generator=g)
        d["rating"] = torch.randint(1, 6, (B,), generator=g).float()
        d["label"] = (d["rating"] >= 3).float()
        d["gender"] = gender[u]
        out.append(Interaction(d).to(dev))
    return out


def _timed(step, K, W, first=0):
    for k in range(W):
        step(first + k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        step(first + k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K


def _profiled(step, K, first=0):
    """Per-kind device time and work over K eager steps: {kind: (ms per step, work per step)}."""
    from fairrec import _C
    _C.prof_reset()
    _C.prof_enable(True)
    torch.cuda.synchronize()
    for k in range(K):
        step(first + k)
    torch.cuda.synchronize()
    _C.prof_enable(False)
    t, w = _C.prof_read(), _C.prof_read_work()
    return {name: (ms / K, n / K, w.get(name, 0.0) / K) for name, (ms, n) in t.items()}


def _gemm_summary(prof):
    ms = sum(prof[k][0] for k in GEMM_KINDS if k in prof)
    flop = sum(prof[k][2] for k in GEMM_KINDS if k in prof)
    n = sum(prof[k][1] for k in GEMM_KINDS if k in prof)
    tf = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {"gemm_ms_per_step": round(ms, 4), "gemm_gflop_per_step": round(flop / 1e9, 3), "gemm_launches_per_step": round(n, 1),
            "gemm_tflops": round(tf, 2), "gemm_frac_of_mfma_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4)}


def _kernel_table(prof):
    return {k: {"ms": round(v[0], 4), "launches": round(v[1], 1)} for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}


def _pmc_note(name):
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
    return json.load(open(f)) if os.path.exists(f) else None


def bench_pfcn(args, dev):
    """BASELINE.json configs[2]: PFCN_BiasedMF, filter_mode sm, one sensitive attribute, embedding 128, discriminator
    [128, 256, 128, 128, 64, 32] with BatchNorm and dropout 0.3, dis_weight 10, 10 000 001 users x 1 000 001 items."""
    from fairrec.config import Config
    from fairrec.graph import GraphedStep
    from fairrec.optim import FusedLazyAdam
    from fairrec.utils import get_model
    nu, ni, D = args.users or 10_000_001, args.items or 1_000_001, 128
    K, W = args.steps, max(args.warmup, 4)
    cfg = Config(model="PFCN_BiasedMF", config_dict={"embedding_size": D, "device": str(dev), "filter_mode": "sm"})
    ds = _DS(nu, ni)
    torch.manual_seed(2020)
    m = get_model("PFCN_BiasedMF")(cfg, ds).to(dev)
    m.train()
    eng = m.hip_engine()
    of = FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-4, group="filter")
    od = FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-4, group="dis")
    sl = ["gender"]
    data = _batches(nu, ni, 16, ds._uf["gender"], dev, pair=True)
    f_loss, d_loss = (lambda it: m.calculate_loss(it, sl)), (lambda it: m.calculate_dis_loss(it, sl))
    gf, gd = GraphedStep(eng, of, f_loss, eager_steps=2), GraphedStep(eng, od, d_loss, eager_steps=2)
    f_step, d_step = (lambda k: gf(data[k % len(data)])), (lambda k: gd(data[k % len(data)]))

    def eager(opt, fn):
        def s(k):
            opt.zero_grad()
            loss = fn(data[k % len(data)])
            loss.backward()
            opt.step()
        return s
    # age the lazy-Adam state by one sweep period of the largest trainable table (as the FOCF workload does): a fresh table
    # has nothing to replay
    n_age = args.age if args.age > 0 else (0 if args.age < 0 else max(t.default_sweep(B) for t in eng._tables.values() if t.trainable))
    for k in range(n_age):
        f_step(k)
    t_f = _timed(f_step, K, W, n_age)
    t_d = _timed(d_step, K, W, n_age)
    pf = _profiled(eager(of, f_loss), min(K, 10), n_age)
    pd = _profiled(eager(od, d_loss), min(K, 10), n_age)
    eng.check_device_errors()
    gf_, gd_ = _gemm_summary(pf), _gemm_summary(pd)
    tot_flop = (gf_["gemm_gflop_per_step"] + gd_["gemm_gflop_per_step"]) * 1e9
    tot_ms = gf_["gemm_ms_per_step"] + gd_["gemm_ms_per_step"]
    tf = tot_flop / (tot_ms * 1e-3) / 1e12
    return {
        "metric": "training interactions/sec + MFMA utilisation, PFCN_BiasedMF sm emb=128 (BASELINE.json configs[2])",
        "value": round(B / (t_f + t_d), 1), "unit": "interactions/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": round((t_f + t_d) * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"PFCN_BiasedMF filter_mode=sm, {nu} users x {ni} items, embedding_size={D}, B={B}, discriminator "
                               "[128,256,128,128,64,32] BatchNorm + dropout 0.3, dis_weight 10, Adam lr=1e-3 wd=1e-4; one step = one "
                               "filter pass + one discriminator pass",
                   "filter_pass_ms": round(t_f * 1e3, 5), "dis_pass_ms": round(t_d * 1e3, 5), "launch": "hipGraph step",
                   "aged_steps": n_age, "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
        "roofline": {"bound": "mfma", "kernel": "linear_fwd / linear_bwd_input / linear_bwd_weight (fp32 MFMA 32x32x2)",
                     "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                     "filter_pass": dict(gf_, gemm_share_of_pass=round(gf_["gemm_ms_per_step"] / (t_f * 1e3), 3)),
                     "dis_pass": dict(gd_, gemm_share_of_pass=round(gd_["gemm_ms_per_step"] / (t_d * 1e3), 3)),
                     "whole_step_tflops": round(tot_flop / (t_f + t_d) / 1e12, 3),
                     "mfma_util_pmc": _pmc_note("pmc_mfma_pfcn.json"),
                     "kernels_filter_pass": _kernel_table(pf), "kernels_dis_pass": _kernel_table(pd),
                     "measured": "hipExt start/stop events on every launch of an eager pass after the timed region; FLOP = 2 M N K "
                                 "per product from the entry points (fr_prof_read_work)"},
    }


def bench_fairgo(args, dev):
    """BASELINE.json configs[3] on ONE GPU (the data-parallel replicas of fairrec/replicated_engine.py run exactly this step on
    their share of the batch): FairGo_GCN finetune stage, WAP, n_layers 2, filters [128, 64], discriminators [16, 8, 4],
    10 000 001 users x 1 000 001 items, embedding 128, 20 training ratings per user (nnz = 2 x 20 x n_users)."""
    import scipy.sparse as sp
    from fairrec.config import Config
    from fairrec.graph import GraphedStep
    from fairrec.optim import FusedLazyAdam
    from fairrec.utils import get_model
    nu, ni, D = args.users or 10_000_001, args.items or 1_000_001, 128
    K, W = args.steps, max(args.warmup, 3)
    rng = np.random.default_rng(0)
    t_host = time.time()
    nnz = 20 * nu
    gu, gi = rng.integers(1, nu, nnz), rng.integers(1, ni, nnz)
    graph = sp.coo_matrix((rng.integers(1, 6, nnz).astype(np.float32), (gu, gi)), shape=(nu, ni))
    del gu, gi
    cfg = Config(model="FairGo_GCN", config_dict={"embedding_size": D, "device": str(dev), "aggr_method": "WAP", "n_layers": 2,
                                                  "filter_hidden_size_list": [128, 64], "dis_hidden_size_list": [16, 8, 4]})
    ds = _DS(nu, ni, graph)
    torch.manual_seed(2020)
    m = get_model("FairGo_GCN")(cfg, ds).to(dev)
    m.train()
    m.train_stage = "finetune"
    eng = m.hip_engine()
    torch.cuda.synchronize()
    t_host = time.time() - t_host
    of = FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-4, group="filter")
    od = FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-4, group="dis")
    sl = ["gender"]
    data = _batches(nu, ni, 8, ds._uf["gender"], dev)
    f_loss, d_loss = (lambda it: m.calculate_loss(it, sl)), (lambda it: m.calculate_dis_loss(it, sl))

    def eager(opt, fn):
        def s(k):
            opt.zero_grad()
            loss = fn(data[k % len(data)])
            loss.backward()
            opt.step()
        return s
    f_step = eager(of, f_loss)        # tens of ms of whole-table kernels: launch overhead is nothing here, no graph needed
    # the filter pass twice: propagating over the batch's frontier only (fr_spmm_csr_sel; `fairgo_frontier: auto` turns it on at
    # this graph's size) and, for reference, over whole tables as the reference's torch.sparse.mm does -- same loss bits
    frontier = m.use_frontier()
    t_whole = None
    if frontier:
        m.frontier_mode = False
        t_whole = _timed(f_step, max(2, min(K, 3)), 2)
        m.frontier_mode = True
    t_f = _timed(f_step, K, W)
    pf = _profiled(f_step, min(K, 3))
    fr_rows = None
    if frontier:
        fr = m._frontier(data[0]["user_id"])
        fr_rows = [int(x[0].numel()) for x in fr] if fr is not None else None
    m.begin_dis_phase(sl)             # what the trainer does before a discriminator pass: filtered table + propagations, once
    gd = GraphedStep(eng, od, d_loss, eager_steps=2)
    t_d = _timed(lambda k: gd(data[k % len(data)]), max(K, 20), W + 2)
    pd = _profiled(eager(od, d_loss), 5)
    eng.check_device_errors()
    sp_ms, sp_n, sp_bytes = pf.get("spmm_csr_kernel", (0.0, 0, 0.0))
    spmm_gbs = sp_bytes / (sp_ms * 1e-3) / 1e9 if sp_ms > 0 else 0.0
    # what the kernel actually requests: one D-float row of X per nonzero (random columns: served by L2 / Infinity Cache / HBM
    # in whatever mix the table's size allows) + (col, val) + the output rows
    gathered = sp_n * (2.0 * nnz * (4.0 * D + 12.0) + 4.0 * (nu + ni) * D)
    gathered_gbs = gathered / (sp_ms * 1e-3) / 1e9 if sp_ms > 0 else 0.0
    gf_ = _gemm_summary(pf)
    return {
        "metric": "training interactions/sec + SpMM GB/s + GEMM TFLOP/s, FairGo_GCN(WAP) finetune emb=128 (BASELINE.json configs[3])",
        "value": round(B / (t_f + t_d), 1), "unit": "interactions/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": round((t_f + t_d) * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"FairGo_GCN finetune WAP n_layers=2, {nu} users x {ni} items, embedding_size={D}, B={B}, filters "
                               f"[128,64], discriminators [16,8,4], graph nnz={2 * nnz}; one step = one filter pass (whole-table "
                               "filter MLP forward + backward, 2 + 2 SpMM) + one discriminator pass (filtered table and its "
                               "propagations cached per pass)",
                   "filter_pass_ms": round(t_f * 1e3, 4), "dis_pass_ms": round(t_d * 1e3, 4),
                   "propagation": ("frontier-restricted (fr_spmm_csr_sel): rows of H_1 / H_2 computed = %s of %d graph rows"
                                   % (fr_rows, nu + ni)) if frontier else "whole tables",
                   "filter_pass_ms_whole_table_propagation": None if t_whole is None else round(t_whole * 1e3, 4),
                   "host_build_s": round(t_host, 1), "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
        # with the propagation restricted to the batch's frontier the whole-table filter MLP (three dense products each way) is the
        # dominant work of a filter pass: the roofline that bounds it is the fp32 MFMA peak; the SpMM figures stay next to it
        "roofline": ({"bound": "mfma", "kernel": "linear_fwd / linear_bwd_input / linear_bwd_weight (whole-table filter MLP)",
                      "achieved": gf_["gemm_tflops"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": gf_["gemm_frac_of_mfma_peak"], "traffic": None} if frontier else
                     {"bound": "hbm", "kernel": "spmm_csr_kernel", "achieved": round(spmm_gbs, 1), "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": round(spmm_gbs / HBM_PEAK_GBS, 4), "traffic": None}) | {
                     "algorithmic_bytes_per_launch": round(sp_bytes / max(pf.get("spmm_csr_kernel", (0, 1, 0))[1], 1)),
                     "spmm_ms_per_filter_pass": round(sp_ms, 3), "spmm_share_of_filter_pass": round(sp_ms / (t_f * 1e3), 3),
                     "spmm_requested_GBps": round(gathered_gbs, 1),
                     "spmm_note": "`achieved` prices the launch at SURVEY.md §8-d's algorithmic bytes (every X row read once); the "
                                  "kernel gathers one D-float row per nonzero, `spmm_requested_GBps`, which a random graph leaves "
                                  "to the caches",
                     "gemm": dict(gf_, gemm_share_of_filter_pass=round(gf_["gemm_ms_per_step"] / (t_f * 1e3), 3)),
                     "kernels_filter_pass": _kernel_table(pf), "kernels_dis_pass": _kernel_table(pd),
                     "measured": "hipExt start/stop events on every launch of an eager pass; SpMM bytes = 12 nnz + 8 n_rows D "
                                 "(SURVEY.md §8-d), FLOP = 2 M N K per dense product (fr_prof_read_work)"},
    }
