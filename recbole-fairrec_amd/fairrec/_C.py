"""ctypes binding of libfairrec_hip.so (the C ABI declared in include/fairrec_hip.h).

There is NO fallback: if the library is missing the import of anything that needs a kernel raises.
Build it with `python __graft_entry__.py` (or `make -C recbole-fairrec_amd/csrc`).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

# FAIRREC_HIP_LIB: another build of the SAME library (diagnostic / A-B builds made with `make VARIANT=...`); never a fallback
LIB_PATH = os.environ.get("FAIRREC_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                             "libfairrec_hip.so")

FR_SORT_MAX = 16384
DEV_ERR_INDEX_RANGE = 1
DEV_ERR_SST_GROUPS = 2
DEV_ERR_BUCKET_OVERFLOW = 4
DEV_ERR_PIPE_WAIT = 8
EUNSUPPORTED = -3      # FR_EUNSUPPORTED
WGRAD_MAX = 8          # FR_WGRAD_MAX

FOCF_OBJECTIVES = {"none": 0, "value": 1, "absolute": 2, "under": 3, "over": 4, "nonparity": 5}


class FrTable(Structure):
    _fields_ = [("p", c_void_p), ("m", c_void_p), ("v", c_void_p), ("last", c_void_p), ("stamp", c_void_p),
                ("n_rows", c_int64), ("dim", c_int32), ("step", c_int32), ("step_dev", c_void_p)]


class FrDenseDesc(Structure):   # include/fairrec_hip.h: fr_dense_desc
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", c_int64), ("step", c_int32),
                ("step_dev", c_void_p)]


class FrAdam(Structure):
    _fields_ = [("scalars", c_void_p), ("cap", c_int32), ("reserved_", c_int32), ("weight_decay", c_double),
                ("beta1", c_double), ("beta2", c_double), ("eps", c_double)]


class FrScorer(Structure):      # include/fairrec_hip.h: fr_scorer
    _fields_ = [("k0", c_int32), ("k1", c_int32), ("n1", c_int32), ("n2", c_int32), ("W1", c_void_p), ("b1", c_void_p),
                ("W2", c_void_p), ("b2", c_void_p), ("W3", c_void_p), ("b3", c_void_p), ("p", c_float), ("seed", c_uint64),
                ("off_x0", c_uint64), ("off_x1", c_uint64), ("off_h1", c_uint64), ("off_h2", c_uint64)]


class FrWgradJob(Structure):    # include/fairrec_hip.h: fr_wgrad_job
    _fields_ = [("dY", c_void_p), ("x0", c_void_p), ("k0", c_int32), ("x1", c_void_p), ("k1", c_int32), ("N", c_int32),
                ("dW", c_void_p), ("db", c_void_p), ("parts", c_void_p), ("n_parts", c_int32)]


class FrFocfBatch(Structure):
    _fields_ = [("user", c_void_p), ("item", c_void_p), ("sst", c_void_p), ("B", c_int64), ("ws", c_void_p),
                ("ws_bytes", c_size_t), ("rating", c_void_p)]


class FairrecError(RuntimeError):
    pass


# name -> (restype, argtypes); mirrors include/fairrec_hip.h one to one
_PROTOS = {
    "fr_version": (c_int, []),
    "fr_last_error": (c_char_p, []),
    "fr_sort_segments": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "fr_focf_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "fr_focf_forward": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_void_p, c_void_p, c_void_p,
                                c_void_p, c_int64, c_int32, c_float, c_int32, c_void_p, c_size_t, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "fr_focf_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int32, c_void_p, c_size_t,
                                c_void_p, c_void_p]),
    "fr_focf_prepare_many": (c_int, [POINTER(FrFocfBatch), c_int32, c_int64, c_int64, c_int32, c_void_p, c_void_p]),
    "fr_focf_clip_grad_norm": (c_int, [POINTER(FrTable), POINTER(FrTable), c_int64, c_float, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p]),
    "fr_focf_backward_adam": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_int64, c_int32,
                                      c_void_p, c_size_t, c_void_p]),
    "fr_focf_prepare_step": (c_int, [POINTER(FrFocfBatch), POINTER(c_int32), c_int32, POINTER(FrTable), POINTER(FrTable),
                                     c_int32, c_void_p, c_void_p]),
    "fr_focf_step": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_void_p, c_void_p, c_void_p, c_void_p,
                             c_int64, c_int32, c_float, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_void_p, c_int64,
                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "fr_focf_step_runs": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int64, c_int32, c_float, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_int64, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "fr_focf_step_runs_pipe": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int64, c_int32, c_float, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_int64, c_int32,
                                       c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fr_focf_step_finish": (c_int, [c_void_p, c_size_t, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                    c_void_p]),
    "fr_side_stream_handle": (c_void_p, []),
    "fr_focf_row_words": (c_size_t, [c_int64, c_int64]),
    "fr_focf_stage": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrFocfBatch), c_int32, c_int32,
                              POINTER(FrFocfBatch), c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "fr_focf_step_staged": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_void_p, c_int64, c_int32, c_float,
                                    c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_int64, c_void_p, c_void_p,
                                    c_void_p, POINTER(FrFocfBatch), c_int32, c_int32, POINTER(FrFocfBatch), c_int32, c_int32,
                                    c_void_p, c_void_p]),
    "fr_focf_step_finish_staged": (c_int, [c_void_p, c_size_t, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                           c_void_p]),
    "fr_focf_compose_epoch": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64,
                                      c_void_p, c_int64, c_void_p]),
    "fr_randperm_workspace_bytes": (c_size_t, [c_int64]),
    "fr_randperm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fr_focf_runs_many": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_void_p, c_int32, c_int32, c_float,
                                  c_int32, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                  c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fr_focf_steps_many": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_void_p, c_int32, c_int32, c_float,
                                   c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "fr_focf_predict": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_void_p, c_void_p, c_int64,
                                c_float, c_void_p, c_void_p, c_void_p]),
    "fr_table_flush": (c_int, [POINTER(FrTable), POINTER(FrAdam), c_void_p]),
    "fr_table_gather": (c_int, [POINTER(FrTable), POINTER(FrAdam), c_void_p, c_int64, c_void_p, c_void_p,
                                c_void_p]),
    "fr_table_train_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "fr_table_segments_bytes": (c_size_t, [c_int64]),
    "fr_table_gather_train": (c_int, [POINTER(FrTable), POINTER(FrAdam), c_void_p, c_int64, c_int32, c_int32, c_void_p,
                                      c_void_p, c_size_t, c_void_p, c_void_p]),
    "fr_table_gather_train_prepared": (c_int, [POINTER(FrTable), POINTER(FrAdam), c_void_p, c_int64, c_int32, c_int32,
                                               c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "fr_table_apply_grad": (c_int, [POINTER(FrTable), POINTER(FrAdam), c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                    c_int32, c_void_p, c_size_t, c_void_p]),
    "fr_shard_pack_records": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "fr_shard_unpack_records": (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "fr_shard_count_distinct": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fr_shard_post_fair": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "fr_shard_loss_finish": (c_int, [c_void_p, c_void_p, c_int32, c_int64, c_float, c_int32, c_void_p, c_void_p]),
    "fr_bucket_by_owner_sparse": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p]),
    "fr_bucket_by_owner": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int32, c_void_p, c_void_p]),
    "fr_bucket_pair_by_owner": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "fr_mt19937_seed": (c_int, [c_void_p, c_uint32, c_void_p]),
    "fr_sample_negatives_workspace_bytes": (c_size_t, [c_int64]),
    "fr_sample_negatives_calls_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "fr_sample_negatives": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int64,
                                    c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "fr_eval_hits": (c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "fr_topk_like_torch_cpu": (c_int, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p]),
    "fr_eval_topk_segments": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "fr_eval_lookup_segments": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "fr_topk_metrics_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "fr_topk_metrics": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fr_group_sums": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "fr_fair_metrics_workspace_bytes": (c_size_t, [c_int64]),
    "fr_fair_metrics_from_stats": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fr_adam_dense_multi": (c_int, [POINTER(FrDenseDesc), c_int32, POINTER(FrAdam), c_void_p]),
    "fr_sample_negatives_calls": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                          c_int64, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "fr_unbucket_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "fr_bucket_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "fr_focf_shard_score": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                    c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                    c_void_p]),
    "fr_focf_shard_fair": (c_int, [c_void_p, c_size_t, c_int64, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32,
                                   c_int32, c_float, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "fr_focf_shard_nonparity_sums": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p, c_int32,
                                             c_void_p, c_void_p]),
    "fr_focf_shard_nonparity_coef": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p, c_int64,
                                             c_float, c_void_p, c_void_p, c_void_p]),
    "fr_focf_shard_grads": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_float,
                                    c_void_p, c_int32, c_int32, c_int32, c_int64, c_int32, c_void_p, c_void_p,
                                    c_void_p]),
    "fr_table_gather_train2": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_void_p, c_void_p, c_int64,
                                       c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_size_t,
                                       c_void_p, c_void_p]),
    "fr_table_sort2": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                               c_size_t, c_void_p, c_void_p]),
    "fr_table_apply_grad2": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_int64, c_int32, c_int32,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    "fr_table_apply_grad_two": (c_int, [POINTER(FrTable), POINTER(FrTable), POINTER(FrAdam), c_int64, c_int64, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "fr_linear_fwd": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_float, c_void_p, c_void_p, c_int64,
                              c_int32, c_int32, c_void_p, c_void_p]),
    "fr_linear_bwd_input": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_float, c_int64, c_int32,
                                    c_void_p, c_int32, c_void_p, c_int32, c_void_p]),
    "fr_linear_bwd_weight_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "fr_table_join": (c_int, [c_void_p, c_void_p]),
    "fr_linear_n1_bwd": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int64, c_float, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fr_linear_bwd_input_relu": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_float, c_void_p,
                                         c_void_p]),
    "fr_linear_bwd_input_act": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p,
                                        c_void_p]),
    "fr_act_bwd": (c_int, [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "fr_act_bwd_dropped": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_void_p, c_void_p]),
    "fr_dropout_apply": (c_int, [c_void_p, c_int64, c_float, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p]),
    "fr_dropout_apply2": (c_int, [c_void_p, c_int64, c_uint64, c_void_p, c_void_p, c_int64, c_uint64, c_void_p, c_float, c_uint64,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "fr_bpr_outer2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_size_t, c_void_p]),
    "fr_rowdot_rep_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "fr_rowdot_rep_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "fr_rowdot_rep_bwd_sep": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "fr_loss_accumulate": (c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    "fr_copy_many": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "fr_linear_bwd_weight": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p,
                                     c_float, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fr_nfcf_loss_workspace_bytes": (c_size_t, [c_int64]),
    "fr_nfcf_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_size_t, c_int32, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "fr_table_lookup_pair": (c_int, [POINTER(FrTable), POINTER(FrAdam), c_void_p, c_int64, c_void_p, c_void_p, c_size_t,
                                     POINTER(FrTable), POINTER(FrAdam), c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "fr_nfcf_loss_tail": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_size_t, c_int32, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p, c_void_p]),
    "fr_scorer_supported": (c_int, [POINTER(FrScorer)]),
    "fr_scorer_blocks": (c_int64, [c_int64]),
    "fr_scorer_fwd": (c_int, [POINTER(FrScorer), c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p]),
    "fr_scorer_bwd": (c_int, [POINTER(FrScorer), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fr_parts_sum": (c_int, [c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "fr_linear_bwd_weight_multi_workspace_bytes": (c_size_t, [POINTER(FrWgradJob), c_int32, c_int64]),
    "fr_linear_bwd_weight_multi": (c_int, [POINTER(FrWgradJob), c_int32, c_int64, c_void_p, c_size_t, c_void_p]),
    "fr_nfcf_df_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "fr_nfcf_df_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int32, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "fr_nfcf_df_owner": (c_int, [c_void_p, c_size_t, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_size_t,
                                 c_int64, c_void_p, c_void_p]),
    "fr_nfcf_df_apply": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                 c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fr_frontier_mark": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "fr_frontier_expand": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "fr_frontier_count": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "fr_frontier_scatter": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "fr_bn_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "fr_linear_fwd_bnstats": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "fr_bn_fwd_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_void_p, c_float, c_uint64, c_uint64,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "fr_bn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fr_bn_fwd_drop": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_float, c_uint64, c_uint64,
                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "fr_bn_bwd": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_size_t, c_void_p]),
    "fr_bn_bwd_ex": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_size_t, c_int32, c_void_p]),
    "fr_linear_bwd_input_bnstats": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                                            c_void_p, c_size_t, c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "fr_rowdot_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "fr_rowdot_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "fr_bpr_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "fr_bpr": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fr_bpr_outer_rect_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "fr_bpr_outer_rect": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "fr_bpr_outer": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fr_softmax_ce": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                              c_void_p]),
    "fr_spmm_csr": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "fr_spmm_csr_sel": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_void_p,
                                c_void_p]),
    "fr_row_gather": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "fr_row_scatter_workspace_bytes": (c_size_t, [c_int64]),
    "fr_row_scatter_sum": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_size_t, c_void_p,
                                   c_void_p]),
    "fr_row_scatter_add": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_size_t, c_void_p,
                                   c_void_p]),
    "fr_spmm_csr_sel_act": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_void_p,
                                    c_void_p, c_int32, c_void_p, c_void_p]),
    "fr_row_scatter_add_act": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_size_t, c_void_p,
                                       c_int32, c_void_p, c_void_p]),
    "fr_mse": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "fr_prof_enable": (c_int, [c_int]),
    "fr_prof_reset": (c_int, []),
    "fr_prof_kernel_count": (c_int, []),
    "fr_prof_kernel_name": (c_char_p, [c_int]),
    "fr_prof_read": (c_int, [c_int, POINTER(c_double), POINTER(c_int64)]),
    "fr_prof_read_work": (c_int, [c_int, POINTER(c_double)]),
    "fr_adam_dense": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, POINTER(FrAdam), c_int32,
                              c_void_p]),
}

_lib = None


def exported_names():
    return sorted(_PROTOS)


def lib():
    """The loaded library; raises FairrecError (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FairrecError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the training hot path.")
        # torch first: it ships its own ROCm runtime (libamdhip64); loaded after ours, the process would hold two HIP
        # runtimes and the kernels registered with one would not see the device context of the other
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)  # AttributeError here = ABI mismatch between header and library
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().fr_last_error()
        raise FairrecError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (0 for None)."""
    return 0 if t is None else t.data_ptr()


_SIDE = {}


def side_stream(device):
    """The library's own side stream as a torch stream (None when it has none), for Tensor.record_stream: the id sort of
    fr_table_gather_train reads the id list and writes the table's workspace there, and torch's allocator recycles memory
    in the order of the streams IT knows a tensor was used on."""
    import torch
    key = (device.type, device.index)
    if key not in _SIDE:
        h = lib().fr_side_stream_handle() if device.type == "cuda" else None
        _SIDE[key] = torch.cuda.ExternalStream(h, device=device) if h else None
    return _SIDE[key]


def current_stream() -> int:
    """Raw hipStream_t of torch's current stream on the current device (the private getter triton also uses: it skips the
    Python Stream object that torch.cuda.current_stream() builds on every call, ~9 us -> <1 us)."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:      # private API moved: fall back to the public one
        return torch.cuda.current_stream().cuda_stream


_ONES = {}


def one(device):
    """A cached fp32 scalar 1 on `device`: the seed `GraphedStep` hands to `loss.backward()`.  A loss Function that finds
    exactly this tensor as its incoming gradient skips the multiplication by it (`is_one`)."""
    import torch
    device = torch.device(device)
    t = _ONES.get(device)
    if t is None:
        t = _ONES[device] = torch.ones((), dtype=torch.float32, device=device)
    return t


_ZEROS = {}


def zeros_cached(shape, device):
    """A shared all-zero fp32 tensor (the exactly-zero gradients some reference losses keep in the graph): read-only by
    contract -- nothing on this path writes into a gradient it was handed."""
    import torch
    key = (tuple(shape), torch.device(device))
    t = _ZEROS.get(key)
    if t is None:
        t = _ZEROS[key] = torch.zeros(tuple(shape), dtype=torch.float32, device=device)
    return t


def is_one(t) -> bool:
    c = _ONES.get(t.device)
    return c is not None and t.data_ptr() == c.data_ptr() and t.dim() == 0


def prof_enable(on: bool):
    check(lib().fr_prof_enable(1 if on else 0), "fr_prof_enable")


def prof_reset():
    check(lib().fr_prof_reset(), "fr_prof_reset")


def prof_read_work():
    """{kernel name: algorithmic work (FLOP / bytes, fairrec_hip.h: fr_prof_read_work)} of the kinds that account for it."""
    out = {}
    for k in range(lib().fr_prof_kernel_count()):
        w = c_double(0)
        check(lib().fr_prof_read_work(k, ctypes.byref(w)), "fr_prof_read_work")
        if w.value:
            out[lib().fr_prof_kernel_name(k).decode()] = w.value
    return out


def prof_read():
    """{kernel name: (total device ms, launches)} from the library's HIP-event profiler."""
    out = {}
    for k in range(lib().fr_prof_kernel_count()):
        ms, n = c_double(0), c_int64(0)
        check(lib().fr_prof_read(k, ctypes.byref(ms), ctypes.byref(n)), "fr_prof_read")
        if n.value:
            out[lib().fr_prof_kernel_name(k).decode()] = (ms.value, n.value)
    return out
