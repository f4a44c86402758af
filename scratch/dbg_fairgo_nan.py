"""The FairGo trainer test's training run, REPS times in one process, with a NaN check of every parameter, buffer and table
after every optimizer step: stops at the first NaN and says where it appeared."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import scipy.sparse as sp
import torch
from fairrec.config import Config
from fairrec.data.dataloader import TrainDataLoader
from fairrec.data.dataset import InteractionDataset
from fairrec.data.interaction import Interaction
from fairrec.utils import get_model, get_trainer, init_seed
import fairrec.optim as optim_mod

REPS = int(os.environ.get("REPS", "100"))
state = {"step": 0, "bad": None, "model": None}

def check(tag):
    m = state["model"]
    if m is None or state["bad"]:
        return
    for name, t in list(m.named_parameters()) + list(m.named_buffers()):
        if t.is_floating_point() and not torch.isfinite(t).all():
            state["bad"] = (tag, state["step"], name, int((~torch.isfinite(t)).sum()), tuple(t.shape))
            return

# check after every (graphed or eager) training step, outside any capture
import fairrec.graph as graph_mod
_orig_call = graph_mod.GraphedStep.__call__
def _call(self, *a, **k):
    r = _orig_call(self, *a, **k)
    state["step"] += 1
    check("after GraphedStep (graph %s)" % ("replay" if self.graph is not None else "eager"))
    if state["bad"] is None and not torch.isfinite(r).all():
        state["bad"] = ("loss", state["step"], "loss", 1, tuple(r.shape))
    return r
graph_mod.GraphedStep.__call__ = _call

n_fail = 0
for rep in range(REPS):
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
    state.update(step=0, bad=None, model=model)
    try:
        trainer.fit(TrainDataLoader(cfg, ds, shuffle=False), valid_data=None, verbose=False, saved=True)
        err = None
    except ValueError as e:
        err = str(e)
    if err or state["bad"]:
        n_fail += 1
        print(f"rep {rep}: error={err} first non-finite: {state['bad']}", flush=True)
        if n_fail >= 3:
            break
    # some churn of the allocator between repetitions, as other tests would cause
    junk = [torch.randn(int(torch.randint(1, 200000, (1,))), device="cuda") for _ in range(8)]
    del junk
print(f"done: {n_fail} failures in {rep + 1} repetitions")
