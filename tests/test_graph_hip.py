"""GPU: a whole training step captured in a hipGraph (fairrec.graph.GraphedStep: device-resident step counters,
fr_table.step_dev) reproduces the reference's golden vectors exactly like the eager path -- first steps eager, one capture,
the remaining steps are replays of the same graph on new batches."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


class _DS:
    def __init__(self, n_users, n_items, gender):
        from fairrec.data.interaction import Interaction
        self._n = {"user_id": n_users, "item_id": n_items}
        self._uf = Interaction({"user_id": torch.arange(n_users), "gender": torch.from_numpy(gender)})

    def num(self, f):
        return self._n[f]

    def get_user_feature(self):
        return self._uf


@pytest.mark.parametrize("case", ["nfcf_pretrain", "nfcf_pretrain_wd", "nfcf_finetune", "nfcf_finetune_d64"])
def test_nfcf_graphed_steps_match_reference_golden(case):
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.graph import GraphedStep
    from fairrec.model.fair_recommender.nfcf import NFCF
    from fairrec.optim import FusedLazyAdam
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    lr, wd, fw, p = (float(x) for x in z["hyper"])
    assert p == 0.0            # recorded dropout masks are host inputs; the graphed path draws its own (test below)
    n_users, D = z["init.user_embedding.weight"].shape
    n_items = z["init.item_embedding.weight"].shape[0]
    cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": [int(h) for h in z["hidden"]],
                                            "dropout": p, "fair_weight": fw, "device": "cuda", "load_pretrain_path": None})
    model = NFCF(cfg, _DS(n_users, n_items, z["gender"]))
    if str(z["stage"]) == "finetune":
        model.load_pretrain_path = "reference-checkpoint"
        model.user_embedding.weight.requires_grad = False
    model.load_state_dict({k[5:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.")})
    model = model.to("cuda").train()
    eng = model.hip_engine()
    opt = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=3)
    gs = GraphedStep(eng, opt, model.calculate_loss, eager_steps=2)
    snaps = set(int(s) for s in z["snaps"])
    losses = []
    T = len(z["user_id"])
    for t in range(T):
        inter = Interaction({"user_id": torch.tensor(z["user_id"][t]), "item_id": torch.tensor(z["item_id"][t]),
                             "label": torch.tensor(z["label"][t]), "gender": torch.tensor(z["sst"][t])})
        losses.append(gs(inter).reshape(1).clone())
        if (t + 1) in snaps:
            for k, v in model.state_dict().items():
                ref = z[f"after{t + 1}." + k]
                a = v.cpu().numpy()
                assert (np.abs(a - ref) <= 1e-4 * np.abs(ref) + 2e-6).all(), (k, t + 1, np.abs(a - ref).max())
    assert gs.graph is not None and T > 3
    np.testing.assert_allclose(torch.cat(losses).cpu().numpy(), z["loss"], rtol=1e-4)
    eng.sync_steps()
    assert eng._tables["item_embedding.weight"].step == T
    assert all(d.step == T for d in eng._dense.values())
    eng.check_device_errors()


@pytest.mark.parametrize("case", ["pfcn_pmf_none", "pfcn_bmf_none", "pfcn_mlp_none", "pfcn_dmf_none"])
def test_pfcn_graphed_steps_match_reference_golden(case):
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.graph import GraphedStep
    from fairrec.optim import FusedLazyAdam
    from fairrec.utils import get_model
    from test_pfcn_hip import _DS as PDS
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    name = str(z["model"])
    lr, wd, dis_weight, p = (float(x) for x in z["hyper"])
    utab = "user_embedding" if name == "PFCN_MLP" else "user_embedding_layer"
    itab = "item_embedding" if name == "PFCN_MLP" else "item_embedding_layer"
    n_users, D = z[f"init.model.{utab}.weight"].shape
    n_items = z[f"init.model.{itab}.weight"].shape[0]
    cfg = Config(model=name, config_dict={"embedding_size": D, "sst_attr_list": [str(a) for a in z["attrs"]],
                                          "filter_mode": "none", "dis_hidden_size_list": [int(h) for h in z["dis_hidden"]],
                                          "dis_dropout": p, "dis_weight": dis_weight, "device": "cuda", "dropout": 0.0,
                                          "mlp_hidden_size_list": [8, 4], "num_layers": 2, "mlp_dropout": 0.0,
                                          "mlp_activation": "relu", "dis_activation": "leakyrelu", "activation": "leakyrelu"})
    model = get_model(name)(cfg, PDS(n_users, n_items, z))
    model.load_state_dict({k[11:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.model.")})
    model = model.to("cuda").train()
    eng = model.hip_engine()
    opt = FusedLazyAdam(eng, lr=lr, weight_decay=wd, sweep_period=2)
    gs = GraphedStep(eng, opt, lambda inter: model.calculate_loss(inter, None), eager_steps=1)
    losses = []
    for t in range(len(z["user_id"])):
        inter = Interaction({k: torch.tensor(z[k][t]) for k in ("user_id", "item_id", "neg_item_id")})
        inter["gender"] = torch.tensor(z["gender"][z["user_id"][t]])
        losses.append(gs(inter).reshape(1).clone())
    np.testing.assert_allclose(torch.cat(losses).cpu().numpy(), z["loss"], rtol=2e-4)
    sd = model.state_dict()
    for k, v in sd.items():
        ref = z["final.model." + k]
        a = v.cpu().numpy()
        assert (np.abs(a - ref) <= 1e-4 * np.abs(ref) + 2e-6).all(), (k, np.abs(a - ref).max())


def test_graphed_step_draws_fresh_dropout_masks_and_handles_odd_batches():
    """Dropout inside the captured step uses torch's graph-safe generator: two replays on the SAME batch give different
    losses; a batch of another size falls back to an eager step."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.graph import GraphedStep
    from fairrec.model.fair_recommender.nfcf import NFCF
    from fairrec.optim import FusedLazyAdam
    g = torch.Generator().manual_seed(0)
    n_users, n_items, B = 300, 200, 512
    cfg = Config(model="NFCF", config_dict={"embedding_size": 32, "mlp_hidden_size": [64, 32], "dropout": 0.5,
                                            "device": "cuda", "load_pretrain_path": None})
    model = NFCF(cfg, _DS(n_users, n_items, np.zeros(n_users, dtype=np.float32))).to("cuda").train()
    eng = model.hip_engine()
    opt = FusedLazyAdam(eng, lr=0.0, weight_decay=0.0)       # lr 0: parameters frozen, only the masks change
    gs = GraphedStep(eng, opt, model.calculate_loss, eager_steps=1)

    def batch(n):
        return Interaction({"user_id": torch.randint(1, n_users, (n,), generator=g),
                            "item_id": torch.randint(1, n_items, (n,), generator=g),
                            "label": (torch.rand(n, generator=g) < 0.5).float()})
    b0 = batch(B)
    vals = [float(gs(b0)) for _ in range(5)]
    assert gs.graph is not None
    assert len(set(round(v, 7) for v in vals[1:])) > 1 and all(np.isfinite(vals))
    assert np.isfinite(float(gs(batch(B // 2 + 3))))          # odd size: eager fallback
    eng.sync_steps()
    assert eng._tables["item_embedding.weight"].step == 6


# --- a captured piece must own everything it reads -----------------------------------------------------------------------
def _churn(keep):
    """Hand every cached free block of torch's default pool out once more, filled with 0xFF bytes (NaN as float, -1 as
    int), keep them alive, and give 128 MiB of the same back to the driver: whatever a graph reads that it does not own --
    a dangling pointer to a tensor that lived at capture time, an uninitialised buffer -- now reads NaN."""
    torch.cuda.synchronize()
    size = 1 << 26
    while size >= 512:
        while True:
            before = torch.cuda.memory_reserved()
            t = torch.empty(size, dtype=torch.uint8, device="cuda")
            if torch.cuda.memory_reserved() > before:      # a NEW segment, not a cached block: give it back, next size
                del t
                break
            keep.append(t.fill_(0xFF))
        size >>= 1
    t = torch.full((32 << 18,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()


def _replays_are_idempotent(fn, replays=3):
    """fn() -> list of output tensors.  Eager twice, capture, replay, churn, replay x3: the outputs of every replay must be
    the first replay's, bit for bit, and the first replay's the eager result."""
    for _ in range(2):
        eager = [o.detach().clone() for o in fn()]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        outs = fn()
    g.replay()
    torch.cuda.synchronize()
    first = [o.detach().clone() for o in outs]
    for a, b in zip(eager, first):
        assert torch.equal(a, b), "first replay differs from the eager result"
    keep = []
    _churn(keep)
    for r in range(replays):
        g.replay()
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(first, outs)):
            assert torch.equal(a, b), f"output {k} moved on replay {r + 2} after the allocator's free memory was churned"
    del keep


def test_captured_pieces_are_idempotent_under_allocator_churn():
    """Round 4's root cause of the FairGo 'Training loss is nan' flake, pinned: every autograd Function of the generic steps,
    forward + backward inside ONE hipGraph, replayed after the allocator's free memory was filled with NaN patterns.  Before
    the fix RowGather's backward failed here on every run: its dense gradient was cleared by hipMemsetAsync, i.e. by a memset
    NODE once captured, which this runtime executes on the graph's first launch only -- later replays summed the batch's rows
    into whatever the block's previous tenant had left."""
    from fairrec import _C
    from fairrec.functional import CsrMatrix, Mse, RowDot, RowGather, SigmoidBce, SoftmaxCe, SpMM
    from fairrec.model.layers import MLPLayers
    import scipy.sparse as sp
    g = torch.Generator().manual_seed(5)
    N, D, B = 70, 16, 96
    E0 = torch.randn(N, D, generator=g).cuda()
    idx = torch.randint(0, N, (B,), generator=g).cuda()
    idx2 = torch.randint(0, N, (B,), generator=g).cuda()
    target = torch.randint(1, 6, (B,), generator=g).float().cuda()
    label = (torch.rand(B, generator=g) < 0.5).float().cuda()
    cls = torch.randint(0, 3, (B,), generator=g).cuda()
    dense = (torch.rand(N, N, generator=g) < 0.08).float() * torch.rand(N, N, generator=g)
    L = CsrMatrix(sp.csr_matrix(dense.numpy()), "cuda")
    mlp = MLPLayers([D, 16, 8, D], activation="leakyrelu").cuda()
    mlp_bn = MLPLayers([D, 32, 1], activation="leakyrelu", bn=True).cuda().train()
    head3 = MLPLayers([D, 8, 3], activation="leakyrelu").cuda()
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    one = _C.one(torch.device("cuda", 0))

    def piece(make):
        def fn():
            x = E0.clone().requires_grad_(True)
            y = make(x)
            for p in list(mlp.parameters()) + list(mlp_bn.parameters()) + list(head3.parameters()):
                p.grad = None
            y.backward(one if y.dim() == 0 else torch.ones_like(y))
            return [y.detach().reshape(-1), x.grad]
        return fn

    pieces = {
        "RowGather": lambda x: RowGather.apply(x, idx, err).sum(),
        "SpMM": lambda x: SpMM.apply(x, L).sum(),
        "RowDot+Mse": lambda x: Mse.apply(RowDot.apply(RowGather.apply(x, idx, err), RowGather.apply(x, idx2, err)), target),
        "MLP": lambda x: mlp(x).sum(),
        "MLP+BN+SigmoidBce": lambda x: SigmoidBce.apply(mlp_bn(RowGather.apply(x, idx, err)), label),
        "MLP+SoftmaxCe": lambda x: SoftmaxCe.apply(head3(RowGather.apply(x, idx, err)), cls, err),
        "two-layer propagation": lambda x: RowGather.apply(torch.stack([SpMM.apply(x, L), SpMM.apply(SpMM.apply(x, L), L)], dim=1).mean(dim=1),
                                                           idx, err).sum(),
    }
    for name, make in pieces.items():
        try:
            _replays_are_idempotent(piece(make))
        except AssertionError as e:
            raise AssertionError(f"{name}: {e}") from None
    assert int(err.item()) == 0


def test_captured_lazy_table_step_is_idempotent_under_allocator_churn():
    """The same for the lazy-table pair of a generic step: LazyLookup (fr_table_gather_train, sort in line when capturing) ->
    loss -> backward -> FusedLazyAdam.step() (fr_table_apply_grad + dense Adam) in one graph; the state is put back before
    every replay, so every replay must produce the same table, moments and loss."""
    from fairrec import _C
    from fairrec.engine import GenericEngine
    from fairrec.functional import Mse, RowDot
    from fairrec.optim import FusedLazyAdam
    g = torch.Generator().manual_seed(6)
    NU, NI, D, B = 300, 200, 64, 256
    U = torch.nn.Parameter((torch.randn(NU, D, generator=g) * 0.1).cuda())
    I = torch.nn.Parameter((torch.randn(NI, D, generator=g) * 0.1).cuda())
    eng = GenericEngine("cuda")
    tu, ti = eng.add_table("U", U), eng.add_table("I", I)
    opt = FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-3, sweep_period=4)
    eng.enable_graph_mode()
    u = torch.randint(1, NU, (B,), generator=g).cuda()
    i = torch.randint(1, NI, (B,), generator=g).cuda()
    r = torch.randint(1, 6, (B,), generator=g).float().cuda()
    one = _C.one(torch.device("cuda", 0))
    state = lambda: [U.data, I.data, tu.m, tu.v, tu.last, ti.m, ti.v, ti.last, eng._counters]
    snap = None

    def fn():
        if snap is not None:
            for dst, src in zip(state(), snap):
                dst.copy_(src)
        opt.zero_grad()
        loss = Mse.apply(RowDot.apply(eng.lookup("U", u), eng.lookup("I", i)), r)
        loss.backward(one)
        opt.step()
        return [loss.detach().reshape(1)] + [t for t in state()[:8]]

    for _ in range(3):      # a few real steps first: rows of different staleness
        fn()
    eng.sync_steps()
    snap = [t.detach().clone() for t in state()]
    _replays_are_idempotent(fn)
    eng.sync_steps()
    eng.check_device_errors()


def test_a_flush_after_replayed_steps_is_not_skipped():
    """A replayed step runs no host code, so nothing it does may depend on host-side bookkeeping a replay cannot update.  The
    lazy tables' "some row is behind the optimizer step" flag was such bookkeeping: cleared by a flush, set again only by the
    host code of an eager or capturing step -- a second flush with only REPLAYS since the first was skipped, and a whole-table
    reader (a checkpoint per epoch, FairGo's finetune stage) got rows short of their last zero-gradient steps (found in round 5
    through the run_recbole-with-validation fixture of FairGo_PMF).  Captured run against its eager twin, a state_dict() in the
    middle and one at the end, weight decay large enough that ONE missed step is far outside the tolerance."""
    from fairrec.config import Config
    from fairrec.data.interaction import Interaction
    from fairrec.graph import GraphedStep
    from fairrec.model.fair_recommender.nfcf import NFCF
    from fairrec.optim import FusedLazyAdam
    z = np.load(os.path.join(GOLDEN, "nfcf_pretrain.npz"))
    n_users, D = z["init.user_embedding.weight"].shape
    n_items = z["init.item_embedding.weight"].shape[0]
    T = len(z["user_id"])
    assert T >= 8
    out = []
    for graphed in (True, False):
        cfg = Config(model="NFCF", config_dict={"embedding_size": D, "mlp_hidden_size": [int(h) for h in z["hidden"]],
                                                "dropout": 0.0, "fair_weight": 0.0, "device": "cuda", "load_pretrain_path": None})
        model = NFCF(cfg, _DS(n_users, n_items, z["gender"]))
        model.load_state_dict({k[5:]: torch.tensor(z[k]) for k in z.files if k.startswith("init.")})
        model = model.to("cuda").train()
        eng = model.hip_engine()
        opt = FusedLazyAdam(eng, lr=1e-3, weight_decay=1e-2, sweep_period=0)       # no sweeper: only a flush catches rows up
        gs = GraphedStep(eng, opt, model.calculate_loss, eager_steps=2 if graphed else 1 << 30)
        for t in range(T):
            inter = Interaction({"user_id": torch.tensor(z["user_id"][t]), "item_id": torch.tensor(z["item_id"][t]),
                                 "label": torch.tensor(z["label"][t]), "gender": torch.tensor(z["sst"][t])})
            gs(inter)
            if t == 3:
                model.state_dict()            # flush no. 1: after the capture (step 3) and its first replay
        assert (gs.graph is not None) == graphed
        out.append({k: v.clone() for k, v in model.state_dict().items()})     # flush no. 2: only replays since no. 1
    for k in out[0]:
        torch.testing.assert_close(out[0][k], out[1][k], rtol=2e-5, atol=1e-7, msg=k)
