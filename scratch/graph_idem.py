"""Does a captured piece of the FairGo finetune step give the SAME bits when it is replayed after the allocator's free
memory was churned (every cached free block handed out once more and filled with NaN bit patterns, 512 MiB of NaN handed
back to the driver)?  A difference = the graph reads memory it does not own: a dangling pointer to a tensor that was alive
at capture time, or an uninitialised read.  Stages narrow down WHICH part of the step does.

usage: python scratch/graph_idem.py [stage ...]      stages: filtered fwd fwd_bwd full dis_fwd dis_full pieces
"""
import os
import sys
import pathlib
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recbole-fairrec_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.join(ROOT, "scratch"))
from nan_hunt import engine_state, snapshot, restore  # noqa: E402


def churn(keep):
    """Occupy every cached free block of torch's default pool with NaN-filled tensors (kept alive in `keep`), then hand
    512 MiB of NaN to the driver."""
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    free_cached = st["reserved_bytes.all.current"] - st["allocated_bytes.all.current"]
    size = 1 << 28
    got = 0
    while size >= 512:
        while True:
            before = torch.cuda.memory_reserved()
            try:
                t = torch.empty(size, dtype=torch.uint8, device="cuda")
            except RuntimeError:
                break
            if torch.cuda.memory_reserved() > before:      # served by a NEW segment, not by a cached block: give it back
                del t
                break
            t.fill_(0xFF)
            keep.append(t)
            got += size
        size >>= 1
    t = torch.full((128 << 18,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()
    return free_cached, got


def build():
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
    tmp = tempfile.mkdtemp()
    cfg = Config(model="FairGo_PMF", dataset="synth", config_dict={
        "embedding_size": 16, "aggr_method": "WAP", "n_layers": 2, "filter_hidden_size_list": [16, 8], "dis_hidden_size_list": [8, 4],
        "train_batch_size": 100, "epochs": 2, "pretrain_epochs": 2, "train_epoch_interval": 1, "device": "cuda",
        "checkpoint_dir": tmp})

    class DS(InteractionDataset):
        def inter_matrix(self, form="coo", value_field=None):
            return sp.coo_matrix((self.inter_feat["rating"].numpy(), (self.inter_feat["user_id"].numpy(),
                                                                       self.inter_feat["item_id"].numpy())), shape=(n_users, n_items))

    ds = DS(cfg, inter, users, n_users, n_items)
    model = get_model("FairGo_PMF")(cfg, ds).to("cuda")
    trainer = get_trainer(None, "FairGo_PMF")(cfg, model)
    model.train_stage = "finetune"
    model.train()
    batches = [b.to("cuda") for b in TrainDataLoader(cfg, ds, shuffle=False)]
    return model, trainer, batches


def run_stage(name, fn, eng, batch_static, batches, reps=2):
    """fn() -> list of tensors (outputs).  Eager twice, capture, replay, churn, replay: outputs must not move."""
    from fairrec import _C
    snap = snapshot(eng)
    for _ in range(2):
        restore(eng, snap)
        out_e = [o.detach().clone() for o in fn()]
    restore(eng, snap)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        outs = fn()
    eng.sync_steps()
    restore(eng, snap)
    g.replay()
    torch.cuda.synchronize()
    o1 = [o.detach().clone() for o in outs]
    keep = []
    free_cached, got = churn(keep)
    res = []
    for r in range(reps):
        restore(eng, snap)
        g.replay()
        torch.cuda.synchronize()
        o2 = [o.detach().clone() for o in outs]
        same = [bool(torch.equal(a, b)) for a, b in zip(o1, o2)]
        fin = [bool(torch.isfinite(b).all()) if b.is_floating_point() else True for b in o2]
        res.append((same, fin))
    close = [bool(torch.allclose(a, b, rtol=1e-4, atol=1e-6, equal_nan=True)) for a, b in zip(out_e, o1)]
    del keep
    restore(eng, snap)
    ok = all(all(s) and all(f) for s, f in res)
    print(f"stage {name:12s}: {'OK ' if ok else 'BAD'} cached-free {free_cached >> 10} KiB, refilled {got >> 10} KiB; "
          f"replay1 ~ eager {close}; after churn same={[s for s, _ in res]} finite={[f for _, f in res]}", flush=True)
    return ok


def main():
    from fairrec import _C
    from fairrec.functional import Mse, RowDot, RowGather, SigmoidBce, SpMM
    stages = sys.argv[1:] or ["filtered", "fwd", "fwd_bwd", "full", "dis_fwd", "dis_full", "pieces"]
    model, trainer, batches = build()
    eng = model.hip_engine()
    eng.enable_graph_mode()
    opt_f, opt_d = trainer.optimizer_filter, trainer.optimizer_dis
    b = batches[0]
    one = _C.one(torch.device("cuda", 0))
    fparams = [p for p in model.filter_layer_dict["gender"].parameters()]
    dparams = [p for p in model.dis_layer_dict["gender"].parameters()]

    def filtered():
        with torch.no_grad():
            return [model._filtered_table(["gender"])]

    def fwd():
        with torch.no_grad():
            return [model.calculate_loss(b, ["gender"]).reshape(1)]

    def fwd_bwd():
        opt_f.zero_grad()
        loss = model.calculate_loss(b, ["gender"])
        loss.backward(one)
        return [loss.detach().reshape(1)] + [p.grad.clone() for p in fparams]

    def full():
        opt_f.zero_grad()
        loss = model.calculate_loss(b, ["gender"])
        loss.backward(one)
        opt_f.step()
        return [loss.detach().reshape(1)] + [p.data for p in fparams]

    def dis_fwd():
        with torch.no_grad():
            return [model.calculate_dis_loss(b, ["gender"]).reshape(1)]

    def dis_full():
        opt_d.zero_grad()
        loss = model.calculate_dis_loss(b, ["gender"])
        loss.backward(one)
        opt_d.step()
        return [loss.detach().reshape(1)] + [p.data for p in dparams]

    table = {"filtered": filtered, "fwd": fwd, "fwd_bwd": fwd_bwd, "full": full, "dis_fwd": dis_fwd, "dis_full": dis_full}
    bad = 0
    for s in stages:
        if s in table:
            bad += not run_stage(s, table[s], eng, None, batches)
    if "pieces" in stages:
        # the autograd Functions of the step one by one, forward + backward, on fixed inputs
        E0 = torch.randn(70, 16, device="cuda")
        user = b["user_id"]
        idx = torch.cat([user, b["item_id"] + 40])
        rating = b["rating"]
        gender = b["gender"]
        L = model._L

        def piece(make):
            def fn():
                x = E0.clone().requires_grad_(True)
                y = make(x)
                y.backward(one if y.dim() == 0 else torch.ones_like(y))
                return [y.detach().reshape(-1), x.grad]
            return fn
        pieces = {
            "p_rowgather": lambda x: RowGather.apply(x, idx, eng.err_flag).sum(),
            "p_spmm": lambda x: SpMM.apply(x, L).sum(),
            "p_mse": lambda x: Mse.apply(RowDot.apply(RowGather.apply(x, user, eng.err_flag), RowGather.apply(x, b["item_id"] + 40, eng.err_flag)), rating),
            "p_filter": lambda x: model.filter_layer_dict["gender"](x).sum(),
            "p_dis_bce": lambda x: SigmoidBce.apply(model.dis_layer_dict["gender"](RowGather.apply(x, user, eng.err_flag)), gender),
            "p_mean": lambda x: torch.stack([x, x * 2], dim=1).mean(dim=1).sum(),
        }
        for k, mk in pieces.items():
            bad += not run_stage(k, piece(mk), eng, None, batches)
    print("bad stages:", bad, flush=True)
    os._exit(0)


if __name__ == "__main__":
    main()
