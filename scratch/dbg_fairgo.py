import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
from fairrec.quick_start import run_recbole
graph = os.environ.get("GRAPH", "1") == "1"
if os.environ.get("NOCACHE") == "1":
    from fairrec.model.fair_recommender import fairgo_pmf
    del fairgo_pmf.FairGo_PMF.begin_dis_phase
out = run_recbole(model="FairGo_PMF", config_dict={
    "epochs": 2, "pretrain_epochs": 2, "train_epoch_interval": 1, "train_batch_size": 512, "synthetic_users": 150,
    "synthetic_items": 300, "synthetic_interactions": 4000, "device": "cuda", "checkpoint_dir": tempfile.mkdtemp(),
    "embedding_size": 16, "n_layers": 2, "dis_hidden_size_list": [16, 8, 4], "filter_hidden_size_list": [32, 16],
    "aggr_method": os.environ.get("AGGR", "LBA"), "vs_weights": [4, 1], "fair_weight": 0.1, "weight_decay": 1e-4,
    "eval_args": {"mode": "uni20"}, "topk": [5], "valid_metric": "ndcg@5",
    "metrics": ["NDCG"], "sst_attr_list": ["gender"], "eval_batch_size": 2048, "metric_decimal_place": 4,
    "neg_sampling": None, "graph_train_step": graph})
print("OK", out["test_result"])
