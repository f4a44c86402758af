"""`run_recbole`-shaped entry (reference quick_start.py:20-71) for in-memory / synthetic datasets:
Config -> init_seed -> dataset -> dataloaders -> model -> trainer.fit -> trainer.evaluate."""
from __future__ import annotations

from logging import getLogger

import torch

from . import _C
from .config import Config
from .data.dataloader import FOCFDataLoader, TrainDataLoader
from .data.dataset import InteractionDataset, synthetic_dataset
from .utils import get_model, get_trainer, init_seed


def split_dataset(dataset: InteractionDataset, ratios=(0.8, 0.1, 0.1)):
    """Random split by ratio after a torch.randperm shuffle (the RS split of the reference's eval_args)."""
    dataset.shuffle()
    n = len(dataset)
    a, b = int(n * ratios[0]), int(n * (ratios[0] + ratios[1]))
    mk = lambda sl: InteractionDataset(dataset.config, dataset.inter_feat[sl], dataset.user_feat, dataset.user_num,
                                       dataset.item_num)
    return mk(slice(0, a)), mk(slice(a, b)), mk(slice(b, n))


def worst_item_complete_batch(config, train_set) -> int:
    """Rows the largest batch FOCFDataLoader can compose from `train_set` holds: it keeps adding whole item histories until the
    batch has reached train_batch_size rows (focf_dataloader.py:37-51), so up to train_batch_size - 1 rows plus the most rated
    item's.  Every FOCF step sorts its batch's id columns in ONE workgroup (at most FR_SORT_MAX rows); the reference has no such
    bound (ML-20M's most rated item alone has ~67 k ratings).  A dataset whose worst batch would not fit is trained on
    fixed-size batches instead, loudly -- not aborted in mid-epoch."""
    items = train_set.inter_feat[config['ITEM_ID_FIELD']]
    if not len(items):
        return 0
    return int(torch.bincount(items.reshape(-1).to(torch.int64)).max().item()) + int(config['train_batch_size']) - 1


def run_recbole(model=None, dataset=None, config_file_list=None, config_dict=None, saved=True, splits=None,
                before_fit=None):
    """`dataset` is an InteractionDataset, or None for a synthetic one sized by config keys
    `synthetic_users / synthetic_items / synthetic_interactions`.  `splits` = (train, valid, test) InteractionDatasets
    takes the place of the random split (a split made elsewhere, e.g. by the reference's `data_preparation`);
    `before_fit(model, trainer)` runs right before `trainer.fit` (tests/test_e2e_hip.py loads recorded initial
    parameters and generator states there)."""
    config = Config(model=model, dataset=getattr(dataset, 'name', 'synthetic'), config_file_list=config_file_list,
                    config_dict=config_dict)
    init_seed(config['seed'], config['reproducibility'])
    logger = getLogger()
    if splits is not None:
        train_set, valid_set, test_set = splits
        for part in splits:
            part.config = config
    else:
        if dataset is None:
            dataset = synthetic_dataset(config, config['synthetic_users'] or 1000, config['synthetic_items'] or 500,
                                        config['synthetic_interactions'] or 50000, seed=config['seed'])
        train_set, valid_set, test_set = split_dataset(dataset)
    on_gpu = config['device'].type == 'cuda'
    # FOCF trains on item-complete batches (the reference registers FOCFDataLoader for it whatever the config says,
    # data/utils.py:181-186, :218-220); `item_grouped_batches: False` asks for plain fixed-size batches instead
    focf_item_batches = config['model'] == 'FOCF' and config['item_grouped_batches'] is not False
    if focf_item_batches:
        worst = worst_item_complete_batch(config, train_set)
        if worst > _C.FR_SORT_MAX:
            logger.warning(f"FOCF: an item-complete batch of this dataset can hold {worst} rows (train_batch_size - 1 + the most "
                           f"rated item's {worst - int(config['train_batch_size']) + 1}), more than the {_C.FR_SORT_MAX} a step "
                           "takes; training on fixed-size batches of train_batch_size rows instead (item_grouped_batches: False)")
            focf_item_batches = False
    if focf_item_batches:
        # the interaction columns live on the device (a host-resident dataset costs a copy per batch: 4.3 ms per step
        # instead of 0.07 at B = 8192); which interactions form a batch stays the reference's host logic
        train_data = FOCFDataLoader(config, train_set.to(config['device']) if on_gpu else train_set, shuffle=True)
    elif (config['train_neg_sample_args'] or {}).get('strategy') == 'by' and config['device'].type == 'cuda':
        # negatives are drawn on the device, bit-identical to the reference's host sampler (fairrec/sampler); the
        # interaction columns stay resident on the GPU
        from .sampler import Sampler
        sampler = Sampler(['train', 'valid', 'test'], [train_set, valid_set, test_set],
                          config['train_neg_sample_args']['distribution'], device=config['device']).set_phase('train')
        train_data = TrainDataLoader(config, train_set.to(config['device']), sampler=sampler, shuffle=True)
    else:
        train_data = TrainDataLoader(config, train_set.to(config['device']) if on_gpu else train_set, shuffle=True)
    eval_mode = (config['eval_args'] or {}).get('mode') or ''
    if (eval_mode == 'full' or eval_mode[:3] == 'uni') and config['device'].type == 'cuda':
        # ranking evaluation on the device: top-k and fairness metrics of config['metrics'] (fairrec/evaluator); `uniN`
        # draws N negatives per positive from the numpy-compatible device stream, `full` ranks the whole catalogue
        from .data.dataloader import FullSortEvalDataLoader, NegSampleEvalDataLoader
        from .sampler import Sampler
        phases = Sampler(['train', 'valid', 'test'], [train_set, valid_set, test_set], 'uniform', device=config['device'])
        loader = FullSortEvalDataLoader if eval_mode == 'full' else NegSampleEvalDataLoader
        valid_data = loader(config, valid_set, phases.set_phase('valid'))
        test_data = loader(config, test_set, phases.set_phase('test'))
    else:
        valid_data = TrainDataLoader(config, valid_set)
        test_data = TrainDataLoader(config, test_set)
    init_seed(config['seed'], config['reproducibility'])
    model_obj = get_model(config['model'])(config, train_data.dataset).to(config['device'])
    logger.info(model_obj)
    trainer = get_trainer(None, config['model'])(config, model_obj)
    if before_fit is not None:
        before_fit(model_obj, trainer)
    best_valid_score, best_valid_result = trainer.fit(train_data, valid_data, saved=saved)
    try:
        test_result = trainer.evaluate(test_data, load_best_model=saved)
    except NotImplementedError as e:       # e.g. a filtered model evaluated without a ranking evaluation loader
        logger.warning('evaluation skipped: %s', e)
        test_result = None
    return {'best_valid_score': best_valid_score, 'valid_score_bigger': config['valid_metric_bigger'],
            'best_valid_result': best_valid_result, 'test_result': test_result}
