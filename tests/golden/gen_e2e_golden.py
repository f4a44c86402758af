#!/usr/bin/env python3
"""End-to-end golden fixtures from RUNS OF THE REFERENCE'S OWN TRAINERS (SURVEY.md §8-c fixture kind 5, BASELINE.json configs[0]).

Nothing of the reference is restated here: the script drives `recbole.quick_start.run_recbole`'s own sequence
(quick_start.py:32-61: Config -> init_seed -> create_dataset -> data_preparation -> init_seed -> get_model -> get_trainer ->
trainer.fit) with the reference's classes -- `Trainer` + `FOCFDataLoader` for FOCF on recbole/dataset_example/ml-100k,
`PFCN_BiasedMFTrainer`, `FairGo_PMFTrainer` and `Trainer` (NFCF) on a small synthetic atomic dataset -- and only LISTENS:
`calculate_loss` / `calculate_dis_loss` of the model instance are wrapped so that every call leaves the batch it was given
(the columns of the Interaction), the attribute subset, and the loss it returned.  Stored per case, as data only:

  init.*            every parameter / buffer before the first step (state_dict + the dict-held filter / discriminator MLPs)
  step<k>.*         the columns of the k-th batch, in the order the trainer produced them
  kind, sst, loss   per step: which loss function ('L' calculate_loss / 'D' calculate_dis_loss), the attribute subset, the loss
  epoch_loss        what `_train_epoch` returned per epoch (train_loss_dict)
  final.*           every parameter after the last epoch
  train.* / used.*  the training split as the trainer's loader holds it, and per-user used-item sets of the sampler
  rng.np_*          numpy's global generator state when trainer.fit starts (the stream the negative sampler, FOCFDataLoader's
                    item draws and the trainers' attribute masks share); rng.torch = torch's CPU generator state

Build container only (imports /root/reference).  Usage: python tests/golden/gen_e2e_golden.py [case ...]
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
ARGV = list(sys.argv[1:])          # (_refshim.install() clears argv: the reference's Config parses it)
import _refshim  # noqa: E402

_refshim.install()
import torch  # noqa: E402
import yaml  # noqa: E402

REF = _refshim.REFERENCE_ROOT


def write_synthetic_atomic(root, name, n_users=200, n_items=120, per_user=(8, 30), seed=11):
    """A tiny dataset in the reference's atomic-file format: <name>.inter (user_id, item_id, rating, timestamp) and
    <name>.user (user_id, gender:float in {0, 1} -- the adversarial models need a float 0/1 column)."""
    rng = np.random.default_rng(seed)
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"{name}.inter"), "w") as f:
        f.write("user_id:token\titem_id:token\trating:float\ttimestamp:float\n")
        t = 1_000_000
        for u in range(1, n_users + 1):
            k = int(rng.integers(per_user[0], per_user[1]))
            # popularity-skewed items so that items have several raters
            items = np.unique((n_items * rng.random(k) ** 1.7).astype(int) + 1)
            for it in items:
                t += int(rng.integers(1, 50))
                f.write(f"{u}\t{it}\t{int(rng.integers(1, 6))}\t{t}\n")
    with open(os.path.join(d, f"{name}.user"), "w") as f:
        f.write("user_id:token\tgender:float\n")
        for u in range(1, n_users + 1):
            f.write(f"{u}\t{int(rng.integers(0, 2))}\n")
    return d


class Listener:
    """Wraps the loss functions of ONE model instance; the reference's loops run unchanged around it."""

    def __init__(self, model):
        self.steps = []
        self.depth = 0
        for kind, name in (("L", "calculate_loss"), ("D", "calculate_dis_loss")):
            fn = getattr(model, name, None)
            if fn is None:
                continue
            setattr(model, name, self._wrap(kind, fn))

    def _wrap(self, kind, fn):
        def wrapped(interaction, *args, **kw):
            self.depth += 1
            try:
                out = fn(interaction, *args, **kw)
            finally:
                self.depth -= 1
            if self.depth:          # calculate_loss of the adversarial models calls calculate_dis_loss itself: not a step
                return out
            sst = args[0] if args else kw.get("sst_list")
            cols = {k: v.detach().cpu().numpy().copy() for k, v in interaction.interaction.items()}
            loss = [float(x.item()) for x in out] if isinstance(out, tuple) else [float(out.item())]
            self.steps.append({"kind": kind, "sst": ",".join(sst) if sst else "", "cols": cols, "loss": loss})
            return out
        return wrapped


def dict_mlps(model):
    out = {}
    for attr in ("filter_layer", "filter_layer_dict", "dis_layer_dict"):
        d = getattr(model, attr, None)
        if isinstance(d, dict):
            for key, mlp in d.items():
                for k, v in mlp.state_dict().items():
                    out[f"{attr}.{key}.{k}"] = v.detach().cpu().numpy().copy()
    return out


def narrow(a):
    """ids as int32, small-integer float columns as int8 (ratings 1..5, 0/1 flags): smaller files, lossless."""
    if a.dtype == np.int64 and (a.size == 0 or (a.min() >= -2 ** 31 and a.max() < 2 ** 31)):
        return a.astype(np.int32)
    if a.dtype == np.float32 and a.size and np.all(a == np.round(a)) and np.abs(a).max() < 100:
        return a.astype(np.int8)
    return a


def run_reference(case, model_name, dataset, overrides, trainer_expected, loader_expected):
    from recbole.config import Config
    from recbole.data import create_dataset, data_preparation
    from recbole.utils import get_model, get_trainer, init_seed
    work = tempfile.mkdtemp(prefix="e2e_")
    cwd = os.getcwd()
    os.chdir(work)                       # the reference writes ./log, ./log_tensorboard, ./saved
    try:
        ypath = os.path.join(work, "override.yaml")
        with open(ypath, "w") as f:
            yaml.safe_dump(overrides, f)
        config = Config(model=model_name, dataset=dataset, config_file_list=[ypath])      # quick_start.py:32
        init_seed(config["seed"], config["reproducibility"])
        ds = create_dataset(config)
        train_data, valid_data, test_data = data_preparation(config, ds)
        init_seed(config["seed"], config["reproducibility"])
        model = get_model(config["model"])(config, train_data.dataset).to(config["device"])
        trainer = get_trainer(config["MODEL_TYPE"], config["model"])(config, model)
        assert type(trainer).__name__ == trainer_expected, type(trainer).__name__
        assert type(train_data).__name__ == loader_expected, type(train_data).__name__
        out = {}
        for k, v in model.state_dict().items():
            out["init." + k] = v.detach().cpu().numpy().copy()
        for k, v in dict_mlps(model).items():
            out["init." + k] = v
        # the training split and the sampler's used-item sets, as the loader holds them when fit starts
        feat = train_data.dataset.inter_feat
        for k in feat.interaction:
            out["train." + k] = narrow(feat[k].numpy().copy())
        uf = train_data.dataset.get_user_feature()
        for k in uf.interaction:
            out["user_feat." + k] = narrow(uf[k].numpy().copy())
        out["n_users"], out["n_items"] = np.array(train_data.dataset.user_num), np.array(train_data.dataset.item_num)
        smp = getattr(train_data, "sampler", None)
        if smp is not None and getattr(smp, "used_ids", None) is not None:
            used = smp.used_ids
            out["used.ptr"] = np.cumsum([0] + [len(s) for s in used]).astype(np.int64)
            out["used.ids"] = np.concatenate([np.sort(np.fromiter(s, dtype=np.int64, count=len(s))) for s in used]).astype(np.int32)
        st = np.random.get_state()
        out["rng.np_key"], out["rng.np_pos"] = st[1].copy(), np.array(st[2])
        out["rng.torch"] = torch.get_rng_state().numpy().copy()
        lis = Listener(model)
        epoch_losses = []
        orig_epoch = trainer._train_epoch

        def epoch(*a, **kw):
            r = orig_epoch(*a, **kw)
            epoch_losses.append([float(x) for x in r] if isinstance(r, tuple) else [float(r)])
            return r
        trainer._train_epoch = epoch
        # trainer.fit as quick_start.py:52-54 calls it; valid_data=None: no evaluation between the epochs (an evaluation draws
        # 100 negatives per positive from the same numpy stream and is pinned by its own goldens)
        trainer.fit(train_data, None, saved=True, show_progress=False, verbose=False)
        for k, v in model.state_dict().items():
            out["final." + k] = v.detach().cpu().numpy().copy()
        for k, v in dict_mlps(model).items():
            out["final." + k] = v
        out["kind"] = np.array([s["kind"] for s in lis.steps])
        out["sst"] = np.array([s["sst"] for s in lis.steps])
        width = max(len(s["loss"]) for s in lis.steps)
        out["loss"] = np.array([s["loss"] + [np.nan] * (width - len(s["loss"])) for s in lis.steps], dtype=np.float64)
        out["epoch_loss"] = np.array(json.dumps(epoch_losses))
        for k, s in enumerate(lis.steps):
            for name, col in s["cols"].items():
                out[f"step{k}.{name}"] = narrow(col)
        keep = ("embedding_size", "learning_rate", "weight_decay", "train_batch_size", "epochs", "fair_objective", "fair_weight",
                "filter_mode", "dis_weight", "dis_hidden_size_list", "dis_dropout", "activation", "train_epoch_interval",
                "sst_attr_list", "aggr_method", "vs_weights", "n_layers", "filter_hidden_size_list", "pretrain_epochs",
                "mlp_hidden_size", "dropout", "neg_sampling", "seed", "RATING_FIELD", "LABEL_FIELD", "threshold", "clip_grad_norm",
                "save_sst_embed")
        out["config"] = np.array(json.dumps({k: config[k] for k in keep if k in config.final_config_dict}, default=str))
        out["model"], out["trainer"], out["loader"] = np.array(model_name), np.array(trainer_expected), np.array(loader_expected)
        path = os.path.join(HERE, f"e2e_{case}.npz")
        np.savez_compressed(path, **out)
        n = len(lis.steps)
        print(f"{path}: {n} steps, kinds {''.join(out['kind'][:12])}..., epoch losses {epoch_losses}, "
              f"{os.path.getsize(path) / 1e6:.2f} MB, B of first steps {[len(s['cols']['user_id']) for s in lis.steps[:4]]}")
    finally:
        os.chdir(cwd)


def plain(res):
    """A metric dict (or PFCN's dict of them, one per attribute subset) as plain floats."""
    return {k: (plain(v) if isinstance(v, dict) else float(v)) for k, v in res.items()}


def run_reference_flow(case, model_name, dataset, overrides, trainer_expected, loader_expected, full_ids=False):
    """The WHOLE of quick_start.run_recbole (quick_start.py:32-61): fit WITH validation every epoch (trainer.py:332-418:
    `_valid_epoch` -> `evaluate` -> early_stopping -> `_save_checkpoint` on improvement) and the final
    `evaluate(test_data, load_best_model=True)` (trainer.py:458-515).  With `eval_args.mode: uni100` the evaluation loaders
    draw 100 negatives per positive, user by user (general_dataloader.py:141-146), from the SAME numpy stream the training
    loader's item picks / negative draws / attribute masks come from, so every evaluation moves the batches of the epochs
    after it.  Recorded on top of run_reference's keys:

      valid.* / test.*      the two evaluation splits as their loaders hold them (sorted by user)
      step_sha / step_rows  per training batch k: sha256 over the int64 user, item (and negative item) columns of the k-th training batch, its size
                            (the ids themselves for k < 2, or for every step when `full_ids`)
      eval<j>.phase         'valid<epoch>' or 'test' for the j-th evaluate() call, eval<j>.result its metric dict (json)
      eval<j>.sha / .rows   per batch k of that evaluation: sha256 over the int64 item column (positives then sampled negatives,
                            user by user) followed by the user column, and its size; eval<j>.b<k>.item_id / user_id in full for
                            k = 0 (every batch when `full_ids`)
      eval<j>.tied_users    users of that evaluation whose top-k list is decided by an exact score tie between distinct candidates
      saved_epochs          epochs at which the trainer wrote its checkpoint; best_valid_score, best_valid_result, test_result
    """
    import hashlib
    from recbole.config import Config
    from recbole.data import create_dataset, data_preparation
    from recbole.utils import get_model, get_trainer, init_seed
    work = tempfile.mkdtemp(prefix="e2e_")
    cwd = os.getcwd()
    os.chdir(work)
    try:
        ypath = os.path.join(work, "override.yaml")
        with open(ypath, "w") as f:
            yaml.safe_dump(overrides, f)
        config = Config(model=model_name, dataset=dataset, config_file_list=[ypath])
        init_seed(config["seed"], config["reproducibility"])
        ds = create_dataset(config)
        train_data, valid_data, test_data = data_preparation(config, ds)
        init_seed(config["seed"], config["reproducibility"])
        model = get_model(config["model"])(config, train_data.dataset).to(config["device"])
        trainer = get_trainer(config["MODEL_TYPE"], config["model"])(config, model)
        assert type(trainer).__name__ == trainer_expected, type(trainer).__name__
        assert type(train_data).__name__ == loader_expected, type(train_data).__name__
        assert type(valid_data).__name__ == "NegSampleEvalDataLoader", type(valid_data).__name__
        out = {}
        for k, v in model.state_dict().items():
            out["init." + k] = v.detach().cpu().numpy().copy()
        for k, v in dict_mlps(model).items():
            out["init." + k] = v
        for tag, data in (("train", train_data), ("valid", valid_data), ("test", test_data)):
            feat = data.dataset.inter_feat
            for k in feat.interaction:
                out[f"{tag}." + k] = narrow(feat[k].numpy().copy())
        uf = train_data.dataset.get_user_feature()
        for k in uf.interaction:
            out["user_feat." + k] = narrow(uf[k].numpy().copy())
        out["n_users"], out["n_items"] = np.array(train_data.dataset.user_num), np.array(train_data.dataset.item_num)
        out["eval_step_users"] = np.array([valid_data.step, test_data.step])
        st = np.random.get_state()
        out["rng.np_key"], out["rng.np_pos"] = st[1].copy(), np.array(st[2])
        out["rng.torch"] = torch.get_rng_state().numpy().copy()
        lis = Listener(model)
        epoch_losses, evals, saved_epochs = [], [], []
        orig_epoch, orig_eval, orig_save, orig_predict = trainer._train_epoch, trainer.evaluate, trainer._save_checkpoint, model.predict
        cur = {"batches": None}

        def epoch(*a, **kw):
            r = orig_epoch(*a, **kw)
            epoch_losses.append([float(x) for x in r] if isinstance(r, tuple) else [float(r)])
            return r

        def listen_eval(fn):
            def evaluate(eval_data, *a, **kw):
                cur["batches"] = []
                res = fn(eval_data, *a, **kw)
                phase = "test" if eval_data is test_data else f"valid{len(epoch_losses) - 1}"
                evals.append({"phase": phase, "result": plain(res), "batches": cur["batches"]})
                cur["batches"] = None
                return res
            return evaluate

        def predict(interaction, *a, **kw):
            out = orig_predict(interaction, *a, **kw)
            if cur["batches"] is not None:
                b = {k: v.detach().cpu().numpy().copy() for k, v in interaction.interaction.items() if k in ("user_id", "item_id")}
                # users whose top-k list is decided by an exact score tie between DISTINCT candidates (torch.topk's tie order)
                sc = out.detach().cpu().numpy().reshape(-1).astype(np.float64)
                tied = 0
                for u in np.unique(b["user_id"]):
                    m = b["user_id"] == u
                    _, first = np.unique(b["item_id"][m], return_index=True)
                    v = np.sort(sc[m][first])[::-1][:max(config["topk"]) + 1]
                    tied += int((v[1:] == v[:-1]).any())
                b["tied"] = tied
                cur["batches"].append(b)
            return out

        def save(epoch_idx, *a, **kw):
            saved_epochs.append(int(epoch_idx))
            return orig_save(epoch_idx, *a, **kw)
        trainer._train_epoch, trainer.evaluate, trainer._save_checkpoint, model.predict = epoch, listen_eval(orig_eval), save, predict
        if hasattr(trainer, "pfcn_evaluate"):       # PFCNTrainer validates through a method of its own (trainer.py:965-1030)
            trainer.pfcn_evaluate = listen_eval(trainer.pfcn_evaluate)
        best_valid_score, best_valid_result = trainer.fit(train_data, valid_data, saved=True, show_progress=False, verbose=False)
        test_result = trainer.evaluate(test_data, load_best_model=True, show_progress=False)
        for k, v in model.state_dict().items():          # = the best checkpoint's parameters (loaded for the test evaluation)
            out["final." + k] = v.detach().cpu().numpy().copy()
        out["kind"] = np.array([s["kind"] for s in lis.steps])
        out["sst"] = np.array([s["sst"] for s in lis.steps])
        width = max(len(s["loss"]) for s in lis.steps)
        out["loss"] = np.array([s["loss"] + [np.nan] * (width - len(s["loss"])) for s in lis.steps], dtype=np.float64)
        out["epoch_loss"] = np.array(json.dumps(epoch_losses))
        step_sha, step_rows = [], []
        for k, s in enumerate(lis.steps):                 # ids only (the other columns follow from the splits): a digest per
            h = hashlib.sha256()                          # training batch, the ids themselves for the first two
            for name in ("user_id", "item_id", "neg_item_id"):
                if name in s["cols"]:
                    h.update(s["cols"][name].astype(np.int64).tobytes())
                    if k < 2 or full_ids:
                        out[f"step{k}.{name}"] = narrow(s["cols"][name])
            step_sha.append(h.hexdigest())
            step_rows.append(len(s["cols"]["user_id"]))
        out["step_sha"], out["step_rows"] = np.array(step_sha), np.array(step_rows)
        n_ids = 0
        for j, ev in enumerate(evals):
            out[f"eval{j}.phase"] = np.array(ev["phase"])
            out[f"eval{j}.result"] = np.array(json.dumps(ev["result"]))
            out[f"eval{j}.n_batches"] = np.array(len(ev["batches"]))
            out[f"eval{j}.sha"] = np.array([hashlib.sha256(b["item_id"].astype(np.int64).tobytes() +
                                                           b["user_id"].astype(np.int64).tobytes()).hexdigest() for b in ev["batches"]])
            out[f"eval{j}.rows"] = np.array([len(b["item_id"]) for b in ev["batches"]])
            out[f"eval{j}.tied_users"] = np.array(sum(b["tied"] for b in ev["batches"]))
            for k, b in enumerate(ev["batches"]):
                n_ids += len(b["item_id"])
                if k == 0 or full_ids:
                    out[f"eval{j}.b{k}.item_id"] = b["item_id"].astype(np.int16 if b["item_id"].max() < 2 ** 15 else np.int32)
                    out[f"eval{j}.b{k}.user_id"] = b["user_id"].astype(np.int16 if b["user_id"].max() < 2 ** 15 else np.int32)
        out["n_evals"] = np.array(len(evals))
        out["saved_epochs"] = np.array(saved_epochs)
        out["best_valid_score"] = np.array(float(best_valid_score))
        out["best_valid_result"] = np.array(json.dumps(plain(best_valid_result)))
        out["test_result"] = np.array(json.dumps(plain(test_result)))
        keep = ("embedding_size", "learning_rate", "weight_decay", "train_batch_size", "epochs", "fair_objective", "fair_weight",
                "filter_mode", "dis_weight", "dis_hidden_size_list", "dis_dropout", "activation", "train_epoch_interval",
                "sst_attr_list", "neg_sampling", "seed", "RATING_FIELD", "LABEL_FIELD", "threshold", "clip_grad_norm",
                "save_sst_embed", "eval_step", "stopping_step", "metrics", "topk", "valid_metric", "eval_batch_size",
                "metric_decimal_place", "popularity_ratio", "eval_args", "aggr_method", "vs_weights", "n_layers",
                "filter_hidden_size_list", "pretrain_epochs", "mlp_hidden_size", "dropout", "mlp_dropout", "num_layers",
                "mlp_activation", "dis_activation")
        out["config"] = np.array(json.dumps({k: config[k] for k in keep if k in config.final_config_dict}, default=str))
        out["model"], out["trainer"], out["loader"] = np.array(model_name), np.array(trainer_expected), np.array(loader_expected)
        path = os.path.join(HERE, f"e2e_{case}.npz")
        np.savez_compressed(path, **out)
        print("  users with a tied top-k per evaluation:", [sum(b["tied"] for b in ev["batches"]) for ev in evals])
        print(f"{path}: {len(lis.steps)} steps, {len(evals)} evaluations ({n_ids} scored ids), saved at epochs {saved_epochs}, "
              f"best valid {best_valid_score}, epoch losses {epoch_losses}, {os.path.getsize(path) / 1e6:.2f} MB")
        print("  test_result", test_result)
    finally:
        os.chdir(cwd)


# the metric list, cut-off and evaluation mode of the reference's own test.yaml (test.yaml:34-43)
TEST_YAML_EVAL = {"eval_args": {"split": {"RS": [8, 1, 1]}, "group_by": "user", "order": "RO", "mode": "uni100"},
                  "metrics": ["NDCG", "Recall", "Hit", "MRR", "DifferentialFairness", "GiniIndex", "PopularityPercentage",
                              "ValueUnfairness", "AbsoluteUnfairness", "UnderUnfairness", "OverUnfairness", "NonParityUnfairness"],
                  "valid_metric": "NDCG@5", "topk": [5], "popularity_ratio": 0.1, "eval_batch_size": 8192, "eval_step": 1}


def case_flow_focf_ml100k():
    """BASELINE.json configs[0] as `run_recbole.py -m FOCF -d ml-100k` really runs: 3 epochs, validation after each, the best
    checkpoint reloaded, test evaluation -- with test.yaml's evaluation settings."""
    run_reference_flow("flow_focf_ml100k", "FOCF", "ml-100k", dict(TEST_YAML_EVAL, **{
        "data_path": os.path.join(REF, "recbole", "dataset_example") + "/", "RATING_FIELD": "rating", "LABEL_FIELD": "label",
        "threshold": {"rating": 3.0}, "load_col": {"inter": ["user_id", "item_id", "rating"], "user": ["user_id", "gender"]},
        "sst_attr_list": ["gender"], "embedding_size": 64, "fair_objective": "value", "fair_weight": 1.0, "epochs": 3,
        "seed": 2020, "use_gpu": False, "show_progress": False, "save_sst_embed": False}), "Trainer", "FOCFDataLoader")


def case_flow_pfcn_biasedmf():
    run_reference_flow("flow_pfcn_biasedmf_sm", "PFCN_BiasedMF", "synth", dict(COMMON, **dict(TEST_YAML_EVAL, **{
        "data_path": _synth_root(), "embedding_size": 16, "filter_mode": "sm", "dis_hidden_size_list": [32, 16], "dis_dropout": 0.0,
        "dis_weight": 10, "train_epoch_interval": 1, "weight_decay": 1e-4, "epochs": 3, "save_sst_embed": False})),
        "PFCN_BiasedMFTrainer", "TrainDataLoader", full_ids=True)


def case_flow_nfcf():
    run_reference_flow("flow_nfcf_pretrain", "NFCF", "synth", dict(COMMON, **dict(TEST_YAML_EVAL, **{
        "data_path": _synth_root(), "embedding_size": 16, "mlp_hidden_size": [32, 16], "dropout": 0.0, "load_pretrain_path": None,
        "weight_decay": 1e-6, "epochs": 3, "save_sst_embed": False,
        # (at the default 1e-3 a few users' top-5 lists are decided by exact ties between candidates whose ReLU output is 0 --
        # torch.topk's tie order, not the model, then fixes the metrics; at 2e-4 no list of any evaluation is)
        "learning_rate": 2e-4})), "Trainer", "TrainDataLoader", full_ids=True)


def case_flow_fairgo_pmf():
    run_reference_flow("flow_fairgo_pmf_wap", "FairGo_PMF", "synth", dict(COMMON, **dict(TEST_YAML_EVAL, **{
        "data_path": _synth_root(), "embedding_size": 16, "aggr_method": "WAP", "n_layers": 2, "dis_hidden_size_list": [16, 8, 4],
        "filter_hidden_size_list": [32, 16], "pretrain_epochs": 2, "train_epoch_interval": 1, "weight_decay": 1e-4,
        "fair_weight": 0.1, "epochs": 3, "save_sst_embed": False})), "FairGo_PMFTrainer", "TrainDataLoader", full_ids=True)


def _flow(case, model, trainer, loader="TrainDataLoader", **over):
    run_reference_flow(case, model, "synth", dict(COMMON, **dict(TEST_YAML_EVAL, **dict(
        {"data_path": _synth_root(), "embedding_size": 16, "epochs": 3, "save_sst_embed": False}, **over))), trainer, loader,
        full_ids=True)


def case_flow_pfcn_pmf_none():
    _flow("flow_pfcn_pmf_none", "PFCN_PMF", "PFCN_PMFTrainer", filter_mode="none", dis_hidden_size_list=[32, 16], dis_dropout=0.0,
          dis_weight=10, train_epoch_interval=1, weight_decay=1e-4)


def case_flow_pfcn_mlp_sm():
    _flow("flow_pfcn_mlp_sm", "PFCN_MLP", "PFCN_MLPTrainer", filter_mode="sm", dis_hidden_size_list=[32, 16], dis_dropout=0.0,
          dis_weight=10, train_epoch_interval=1, weight_decay=1e-4, mlp_hidden_size=[32, 16], dropout=0.0, mlp_dropout=0.0)


def case_flow_pfcn_dmf_sm():
    _flow("flow_pfcn_dmf_sm", "PFCN_DMF", "PFCN_DMFTrainer", filter_mode="sm", dis_hidden_size_list=[32, 16], dis_dropout=0.0,
          dis_weight=10, train_epoch_interval=2, weight_decay=1e-4, mlp_dropout=0.0, num_layers=2)


def case_flow_focf_absolute():
    _flow("flow_focf_absolute", "FOCF", "Trainer", "FOCFDataLoader", embedding_size=32, fair_objective="absolute", fair_weight=0.5,
          train_batch_size=512)


def case_flow_focf_nonparity():
    _flow("flow_focf_nonparity", "FOCF", "Trainer", "FOCFDataLoader", embedding_size=32, fair_objective="nonparity", fair_weight=0.5,
          train_batch_size=512)


def case_flow_fairgo_pmf_lba():
    _flow("flow_fairgo_pmf_lba", "FairGo_PMF", "FairGo_PMFTrainer", aggr_method="LBA", n_layers=2, dis_hidden_size_list=[16, 8, 4],
          filter_hidden_size_list=[32, 16], pretrain_epochs=2, train_epoch_interval=2, weight_decay=1e-4, fair_weight=0.1)


def case_focf_ml100k():
    """BASELINE.json configs[0]: `run_recbole.py -m FOCF -d ml-100k`, embedding_size 64, with the override yaml SURVEY.md §8-d /
    App. B-11 prescribes (the model yaml's data settings are shadowed by sample.yaml / ml-100k.yaml)."""
    run_reference("focf_ml100k", "FOCF", "ml-100k", {
        "data_path": os.path.join(REF, "recbole", "dataset_example") + "/", "RATING_FIELD": "rating", "LABEL_FIELD": "label",
        "threshold": {"rating": 3.0}, "load_col": {"inter": ["user_id", "item_id", "rating"], "user": ["user_id", "gender"]},
        "embedding_size": 64, "fair_objective": "value", "fair_weight": 1.0, "epochs": 2, "seed": 2020, "use_gpu": False,
        "show_progress": False}, "Trainer", "FOCFDataLoader")


def _synth_root():
    root = tempfile.mkdtemp(prefix="e2e_data_")
    write_synthetic_atomic(root, "synth")
    return root + "/"


COMMON = {"RATING_FIELD": "rating", "LABEL_FIELD": "label", "threshold": {"rating": 3.0}, "sst_attr_list": ["gender"],
          "load_col": {"inter": ["user_id", "item_id", "rating", "timestamp"], "user": ["user_id", "gender"]},
          "seed": 2020, "use_gpu": False, "show_progress": False, "train_batch_size": 256, "epochs": 2}


def case_pfcn_biasedmf():
    run_reference("pfcn_biasedmf_sm", "PFCN_BiasedMF", "synth", dict(COMMON, **{
        "data_path": _synth_root(), "embedding_size": 16, "filter_mode": "sm", "dis_hidden_size_list": [32, 16], "dis_dropout": 0.0,
        "dis_weight": 10, "train_epoch_interval": 1, "weight_decay": 1e-4}), "PFCN_BiasedMFTrainer", "TrainDataLoader")


def case_fairgo_pmf():
    run_reference("fairgo_pmf_wap", "FairGo_PMF", "synth", dict(COMMON, **{
        "data_path": _synth_root(), "embedding_size": 16, "aggr_method": "WAP", "n_layers": 2, "dis_hidden_size_list": [16, 8, 4],
        "filter_hidden_size_list": [32, 16], "pretrain_epochs": 2, "train_epoch_interval": 1, "weight_decay": 1e-4,
        "fair_weight": 0.1}), "FairGo_PMFTrainer", "TrainDataLoader")


def case_nfcf_pretrain():
    run_reference("nfcf_pretrain", "NFCF", "synth", dict(COMMON, **{
        "data_path": _synth_root(), "embedding_size": 16, "mlp_hidden_size": [32, 16], "dropout": 0.0, "load_pretrain_path": None,
        "weight_decay": 1e-6}), "Trainer", "TrainDataLoader")


CASES = {"focf_ml100k": case_focf_ml100k, "pfcn_biasedmf": case_pfcn_biasedmf, "fairgo_pmf": case_fairgo_pmf,
         "nfcf_pretrain": case_nfcf_pretrain, "flow_focf_ml100k": case_flow_focf_ml100k,
         "flow_pfcn_biasedmf": case_flow_pfcn_biasedmf, "flow_nfcf": case_flow_nfcf, "flow_fairgo_pmf": case_flow_fairgo_pmf,
         "flow_pfcn_pmf_none": case_flow_pfcn_pmf_none, "flow_pfcn_mlp_sm": case_flow_pfcn_mlp_sm,
         "flow_pfcn_dmf_sm": case_flow_pfcn_dmf_sm, "flow_focf_absolute": case_flow_focf_absolute,
         "flow_focf_nonparity": case_flow_focf_nonparity, "flow_fairgo_pmf_lba": case_flow_fairgo_pmf_lba}

if __name__ == "__main__":
    names = [a for a in ARGV if a in CASES] or list(CASES)
    for n in names:
        CASES[n]()
