"""Round-4 hunt for the captured-step NaN of tests/test_fairgo_hip.py::test_fairgo_trainer_pretrain_then_finetune.

One process, many fits: every iteration builds the test's tiny FairGo_PMF model and runs FairGoTrainer.fit with a probe
around GraphedStep.__call__ that snapshots the engine's state before every step, looks at it after the step (one sync per
step) and, at the first non-finite tensor, says WHICH step (eager / capture+replay / replay) of WHICH graph produced it,
then re-runs that step from the snapshot (a) as another replay of the same graph and (b) eagerly, to tell a bad capture
(persistent) from a race at replay time (transient).

env: HUNT_N (iterations, 30)  HUNT_PG (1: hold a 1-rank RCCL world, as the session fixture of the test suite does)
     HUNT_BIG (1: run the full-batch data-parallel test first)  HUNT_PROBE (1: per-step probe; 0: only the epoch's NaN check)
     HUNT_POISON (1: before every capture fill freshly released memory with NaN bit patterns, see poison())
     FAIRREC_RCCL_QUIESCE_S, FAIRREC_TEST_NO_GRAPH as in the product / test.
"""
import os
import sys
import tempfile
import time
import pathlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import scipy.sparse as sp
import torch

N = int(os.environ.get("HUNT_N", "30"))
PG = os.environ.get("HUNT_PG", "1") == "1"
BIG = os.environ.get("HUNT_BIG", "0") == "1"
PROBE = os.environ.get("HUNT_PROBE", "1") == "1"
POISON = os.environ.get("HUNT_POISON", "0") == "1"


def poison(mb=512):
    """Fill `mb` MiB of device memory with 0xFF bytes (a NaN as float, -1 as int) and hand it back to the driver: what the
    next hipMalloc -- e.g. the private pool of a capture -- receives is then not zero-filled if the runtime recycles it."""
    t = torch.full((mb << 18,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()


def engine_state(eng):
    st = {}
    for k, d in eng._dense.items():
        st["dense:" + k] = d.p.data
        st["dense_m:" + k] = d.m
        st["dense_v:" + k] = d.v
    for k, t in eng._tables.items():
        st["table:" + k] = t.weight
        for a in ("m", "v", "last", "stamp"):
            x = getattr(t, a, None)
            if x is not None and x.data_ptr() != t.weight.data_ptr():
                st[f"table_{a}:" + k] = x
    if eng._counters is not None:
        st["counters"] = eng._counters
    return st


def snapshot(eng):
    return {k: v.detach().clone() for k, v in engine_state(eng).items()}


def restore(eng, snap):
    for k, v in engine_state(eng).items():
        v.copy_(snap[k])
    eng.sync_steps()


def bad_names(eng, loss=None):
    bad = [k for k, v in engine_state(eng).items() if v.is_floating_point() and not torch.isfinite(v).all()]
    if loss is not None and not torch.isfinite(loss).all():
        bad.insert(0, "LOSS")
    return bad


def install_probe(report):
    from fairrec import graph as G
    orig = G.GraphedStep.__call__

    def probed(self, inter, *args):
        eng = self.engine
        inter = inter.to(eng.device)
        kind = "eager" if (self.eager_left > 0 or getattr(self.optimizer, "clip", None)) else \
            ("capture+replay" if self.graph is None else "replay")
        self._n_calls = getattr(self, "_n_calls", 0) + 1
        if POISON and kind == "capture+replay":
            poison()
        snap = snapshot(eng)
        torch.cuda.synchronize()
        loss = orig(self, inter, *args)
        torch.cuda.synchronize()
        bad = bad_names(eng, loss)
        if bad and not report.get("first"):
            info = {"group": getattr(self.optimizer, "group", None), "call": self._n_calls, "kind": kind, "bad": bad[:6],
                    "n_bad": len(bad)}
            if self.graph is not None:
                # (a) the same graph once more from the same state
                restore(eng, snap)
                self._refresh(inter)
                self.graph.replay()
                torch.cuda.synchronize()
                info["replay_again_bad"] = len(bad_names(eng, self.loss))
                # (b) the same step eagerly from the same state
                restore(eng, snap)
                l2 = self._eager(inter, args)
                torch.cuda.synchronize()
                info["eager_twin_bad"] = len(bad_names(eng, l2))
                eng.sync_steps()
            report["first"] = info
        return loss

    G.GraphedStep.__call__ = probed


def one_fit(tmp):
    from fairrec.config import Config
    from fairrec.data.dataloader import TrainDataLoader
    from fairrec.data.dataset import InteractionDataset
    from fairrec.data.interaction import Interaction
    from fairrec.utils import get_model, get_trainer, init_seed
    init_seed(3)
    n_users, n_items, n = 40, 30, 300
    g = torch.Generator().manual_seed(2)
    inter = Interaction({"user_id": torch.randint(1, n_users, (n,), generator=g), "item_id": torch.randint(1, n_items, (n,), generator=g),
                         "rating": torch.randint(1, 6, (n,), generator=g).float()})
    users = Interaction({"user_id": torch.arange(n_users), "gender": (torch.rand(n_users, generator=g) < 0.5).float()})
    users["gender"][1:3] = torch.tensor([0.0, 1.0])
    cfg = Config(model="FairGo_PMF", dataset="synth", config_dict={
        "embedding_size": 16, "aggr_method": "WAP", "n_layers": 2, "filter_hidden_size_list": [16, 8], "dis_hidden_size_list": [8, 4],
        "train_batch_size": 100, "epochs": 2, "pretrain_epochs": 2, "train_epoch_interval": 1, "device": "cuda",
        "checkpoint_dir": str(tmp), **({"graph_train_step": False} if os.environ.get("FAIRREC_TEST_NO_GRAPH") else {})})

    class DS(InteractionDataset):
        def inter_matrix(self, form="coo", value_field=None):
            return sp.coo_matrix((self.inter_feat["rating"].numpy(), (self.inter_feat["user_id"].numpy(),
                                                                       self.inter_feat["item_id"].numpy())), shape=(n_users, n_items))

    ds = DS(cfg, inter, users, n_users, n_items)
    model = get_model("FairGo_PMF")(cfg, ds).to("cuda")
    trainer = get_trainer(None, "FairGo_PMF")(cfg, model)
    try:
        trainer.fit(TrainDataLoader(cfg, ds, shuffle=False), valid_data=None, verbose=False, saved=True)
    except ValueError as e:
        if "nan" not in str(e).lower():
            raise
        return "NAN", model
    bad = bad_names(model.hip_engine())
    return ("BADSTATE " + ",".join(bad[:4])) if bad else "ok", model


def main():
    print(f"nan_hunt: N={N} PG={PG} BIG={BIG} PROBE={PROBE} POISON={POISON} quiesce={os.environ.get('FAIRREC_RCCL_QUIESCE_S', '2.0')} "
          f"no_graph={bool(os.environ.get('FAIRREC_TEST_NO_GRAPH'))}", flush=True)
    torch.cuda.init()
    if PG:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29641")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    if BIG:
        import test_fairgo_hip as T

        class Req:
            def getfixturevalue(self, name):
                return None
        T.test_fairgo_full_batch_at_the_baseline_width(PG, Req())
        print("big test done", flush=True)
    report = {}
    if PROBE:
        install_probe(report)
    counts = {}
    t0 = time.time()
    for k in range(N):
        report.clear()
        with tempfile.TemporaryDirectory() as tmp:
            status, model = one_fit(pathlib.Path(tmp))
        counts[status.split()[0]] = counts.get(status.split()[0], 0) + 1
        if status != "ok" or report.get("first"):
            print(f"iter {k}: {status} first={report.get('first')}", flush=True)
        del model
    print(f"done in {time.time() - t0:.0f}s: {counts}", flush=True)
    # no process-group teardown: the process just ends (destroy_process_group with live graphs can block)
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main()
