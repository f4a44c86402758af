"""Trainer with the surface of recbole/trainer/trainer.py:62-531 (`Trainer(config, model)`, `fit`, `evaluate`,
`resume_checkpoint`, overridable `_build_optimizer / _train_epoch / _valid_epoch / _save_checkpoint`).

The step loop keeps the reference's shape (trainer.py:181-196)

    zero_grad -> calculate_loss -> backward -> step

but `optimizer` is fairrec.optim.FusedLazyAdam, so `step()` is ONE HIP launch doing the backward and the Adam
update, and the two per-step host syncs of the reference (`loss.item()`, `isnan`) are taken once per epoch
(losses are accumulated on the device; NaN still raises `ValueError('Training loss is nan')`).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from logging import getLogger
from time import time

import collections

import numpy as np
import torch

from .. import _C
from ..optim import FusedLazyAdam
from ..sampler import host_numpy_stream
from ..utils import calculate_valid_score, dict2str, early_stopping, ensure_dir, get_local_time


class AbstractTrainer:
    def __init__(self, config, model):
        self.config = config
        self.model = model

    def fit(self, train_data):
        raise NotImplementedError('Method [next] should be implemented.')

    def evaluate(self, eval_data):
        raise NotImplementedError('Method [next] should be implemented.')


class Trainer(AbstractTrainer):
    def __init__(self, config, model):
        super().__init__(config, model)
        self.logger = getLogger()
        self.learner = config['learner'] or 'adam'
        self.learning_rate = config['learning_rate']
        self.epochs = config['epochs']
        self.eval_step = min(config['eval_step'] if config['eval_step'] is not None else 1, self.epochs)
        self.stopping_step = config['stopping_step']
        self.clip_grad_norm = config['clip_grad_norm']
        self.valid_metric = (config['valid_metric'] or 'rmse').lower()
        self.valid_metric_bigger = bool(config['valid_metric_bigger'])
        self.test_batch_size = config['eval_batch_size'] or 4096
        self.device = config['device']
        self.checkpoint_dir = config['checkpoint_dir'] or 'saved'
        ensure_dir(self.checkpoint_dir)
        self.saved_model_file = os.path.join(self.checkpoint_dir, '{}-{}.pth'.format(config['model'], get_local_time()))
        self.weight_decay = config['weight_decay'] or 0.0
        self.saved_sst_embed_file = os.path.join(self.checkpoint_dir, '{}_embed.pth'.format(config['model']))

        self.start_epoch = 0
        self.cur_step = 0
        self.best_valid_score = -np.inf if self.valid_metric_bigger else np.inf
        self.best_valid_result = None
        self.train_loss_dict = dict()
        self.optimizer = self._build_optimizer()

    # --- optimizer ----------------------------------------------------------------------------------------
    def _build_optimizer(self, **kwargs):
        """reference trainer.py:114-153.  Only learner 'adam' runs on the fused HIP path (the only learner the
        reference's fair-model configs use)."""
        learner = kwargs.pop('learner', self.learner)
        learning_rate = kwargs.pop('learning_rate', self.learning_rate)
        weight_decay = kwargs.pop('weight_decay', self.weight_decay)
        engine = kwargs.pop('engine', None) or self.model.hip_engine()
        if engine is None:
            raise NotImplementedError(f'{type(self.model).__name__} exposes no HIP engine')
        if learner.lower() != 'adam':
            raise NotImplementedError(f"learner '{learner}' is not on the MI355X hot path yet (adam only)")
        return FusedLazyAdam(engine, lr=learning_rate, weight_decay=weight_decay,
                             sweep_period=self.config['lazy_adam_sweep_period'], clip_grad_norm=self.clip_grad_norm)

    # --- training -----------------------------------------------------------------------------------------
    def _graphed_step(self, key, loss_fn):
        """`graph_train_step: True`: the optimizer step (zero_grad, loss, backward, step) of a generic-engine model as
        one hipGraph, captured on its third batch and replayed afterwards (fairrec/graph.py).  One graph per
        (loss function, attribute subset, optimizer)."""
        if not self.config['graph_train_step']:
            return None
        eng = self.model.hip_engine()
        if eng is None or not hasattr(eng, 'enable_graph_mode'):
            return None                                   # FOCF: its fused engine has its own launch path
        capturable = getattr(self.model, 'step_capturable', None)
        if capturable is not None and not capturable(key[1]):
            return None                                   # e.g. FairGo's frontier-restricted filter step: data-dependent shapes
        graphs = self.__dict__.setdefault('_step_graphs', {})
        key = key + (id(self.optimizer),)
        if key not in graphs:
            from ..graph import GraphedStep
            graphs[key] = GraphedStep(eng, self.optimizer, loss_fn)
        return graphs[key]

    def _train_epoch(self, train_data, epoch_idx, loss_func=None, show_progress=False):
        self.model.train()
        loss_func = loss_func or self.model.calculate_loss
        graphed = self._graphed_step(('plain', getattr(loss_func, '__name__', str(loss_func))), loss_func)
        # a model whose optimizer.step() IS the backward pass (FOCF) needs no autograd round trip per step
        fused = bool(getattr(self.model, 'fused_backward', False)) and loss_func == self.model.calculate_loss
        total = None
        n_tuple = 0
        hint = getattr(self.model, 'hint_next_batch', None)
        if fused:
            eng = self.model.hip_engine()
            eng.defer_loss = True        # the loop below reads the loss only at the epoch end (the engine's running total)
            eng.item_runs = type(train_data).__name__ == 'FOCFDataLoader'     # item-complete batches
            eng.reset_loss_acc()
            try:
                return self._train_epoch_body(train_data, loss_func, hint, graphed, fused, total, n_tuple)
            finally:
                eng.defer_loss = False
        return self._train_epoch_body(train_data, loss_func, hint, graphed, fused, total, n_tuple)

    def _steps_per_call(self):
        """config `train_steps_per_call` (default 256; 0 = the per-batch loop even where the library could run it)."""
        v = self.config['train_steps_per_call']
        return 256 if v is None else int(v)

    def _train_epoch_runs(self, train_data):
        """The epoch of a model whose step loop the library can issue itself (`model.train_steps`, FOCF) over a loader whose
        batches are slices of the dataset (`train_data.take`): one foreign call per run of `train_steps_per_call` batches
        instead of an interpreter round trip per batch -- trainer.py:181-196 with the loop body in C."""
        per_call = self._steps_per_call()
        iter(train_data)                                       # the epoch's shuffle (general_dataloader.py:59-60)
        while True:
            got = train_data.take(per_call)
            if got is None:
                return
            run, size = got
            run = run.to(self.device)
            with torch.no_grad():
                if self.model.train_steps(run, size) is None:  # the engine's state asks for the per-batch path after all
                    sizes = size if not isinstance(size, int) else [size] * -(-len(run) // size)
                    lo = 0
                    for n in sizes:
                        self.model.calculate_loss(run[lo:lo + n])
                        self.optimizer.step()
                        lo += n

    def _train_epoch_body(self, train_data, loss_func, hint, graphed, fused, total, n_tuple):
        if (fused and graphed is None and self._steps_per_call() > 0 and getattr(train_data, 'sliceable', False)
                and getattr(self.model, 'train_steps_ready', lambda: False)()):
            self._train_epoch_runs(train_data)
            return self._epoch_loss(True, 0)
        it = iter(train_data)
        # dataloader look-ahead: a model that can use it (FOCF sorts the coming batches' ids ahead, several per launch)
        # says how many batches it wants announced
        depth = max(1, int(getattr(self.model, 'PREFETCH', 1))) if hint is not None else 1
        queue = collections.deque()

        done = []

        def fill():
            while not done and len(queue) < depth + 1:
                b = next(it, None)
                if b is None:
                    done.append(True)            # a recbole-style loader rewinds after StopIteration: never ask again
                    return
                queue.append(b.to(self.device))

        fill()
        while queue:
            interaction = queue.popleft()
            fill()
            if hint is not None:
                hint(*queue)                           # lets the model start the coming batches' index sorts early
            if graphed is not None:
                total = self._accumulate(total, graphed(interaction).view(1))
                continue
            if fused:
                with torch.no_grad():
                    loss_func(interaction)                    # the step's loss goes into the engine's running total ...
                self.optimizer.step()                         # ... on the device (FocfEngine.loss_acc), read once below
                total = True
                continue
            self.optimizer.zero_grad()
            losses = loss_func(interaction)
            if isinstance(losses, tuple):
                n_tuple = len(losses)
                loss = sum(losses)
                part = torch.stack([l.detach() for l in losses])
            else:
                loss = losses
                part = losses.detach().view(1)
            total = self._accumulate(total, part)
            loss.backward()
            self.optimizer.step()
        return self._epoch_loss(total, n_tuple)

    def _epoch_loss(self, total, n_tuple):
        if total is None:
            return 0.0
        if total is True:
            eng = self.model.hip_engine()
            eng.finish()
            total = eng.loss_acc
            n_tuple_or_one = 1
        else:
            n_tuple_or_one = max(n_tuple, 1)
        eng = self.model.hip_engine()
        err = getattr(eng, 'err_flag', None)
        word = None
        if err is not None and err.numel() == 1 and err.dtype == torch.int32 and err.device == total.device:
            # the device error word rides in the same read: the epoch's ONLY host sync
            host = torch.cat((total.view(torch.int32), err)).cpu()
            acc, word = host[:-1].view(torch.float32).tolist(), int(host[-1])
        else:
            acc = total.cpu().tolist()
        vals = acc[:n_tuple_or_one]
        self._check_nan(torch.tensor(vals), first_bad_step=int(acc[4]))
        if eng is not None:
            eng.check_device_errors() if word is None else eng.check_device_errors(word)
        return tuple(vals) if n_tuple else vals[0]

    def _accumulate(self, acc, part):
        """acc[0..n-1] += the step's n loss values on the device, plus the sticky record of the first NaN step
        (fr_loss_accumulate: one launch, what `total + part` cost before)."""
        part = part.detach().reshape(-1).to(torch.float32).contiguous()
        if acc is None:
            acc = torch.zeros(8, dtype=torch.float32, device=part.device)
        _C.check(_C.lib().fr_loss_accumulate(part.data_ptr(), part.numel(), acc.data_ptr(), _C.current_stream()),
                 "fr_loss_accumulate")
        return acc

    def _check_nan(self, loss, first_bad_step=0):
        """trainer.py:286-288 raises on the STEP whose loss is NaN, before its backward (trainer.py:192-193), at the price of
        one `.item()` host sync per step.  Here the per-step losses are summed on the device and the sum is looked at once
        per epoch: a NaN is sticky through the sum, so the same ValueError is raised for the same epoch -- but at its end,
        after the remaining steps of that epoch have been applied to parameters nobody will use (training aborts either
        way; the last checkpoint is from an earlier, finite epoch in both).  Deliberate deviation, stated in DESIGN.md §9."""
        if torch.isnan(loss).any():
            # (same message as trainer.py:286-288; the step at which the reference would have stopped rides along)
            raise ValueError('Training loss is nan' + (f' (first at step {first_bad_step} of this pass)' if first_bad_step else ''))

    def _valid_epoch(self, valid_data, show_progress=False):
        valid_result = self.evaluate(valid_data, load_best_model=False, show_progress=show_progress)
        return calculate_valid_score(valid_result, self.valid_metric), valid_result

    def _save_checkpoint(self, epoch, verbose=True, **kwargs):
        """Same keys as trainer.py:221-240; model.state_dict() flushes the lazy tables first.  The optimizer state goes out
        in torch.optim.Adam's own layout (integer parameter indices in model.named_parameters() order) whenever every
        tensor it holds is a model parameter, so that the reference's `optimizer.load_state_dict` takes the file as it is;
        resume_checkpoint reads both layouts."""
        saved_model_file = kwargs.pop('saved_model_file', self.saved_model_file)
        try:
            opt_state = self.optimizer.state_dict(param_names=[n for n, _ in self.model.named_parameters()])
        except KeyError:         # tensors outside model.parameters() (dict-held MLPs): the name-keyed layout
            opt_state = self.optimizer.state_dict()
        state = {
            'config': dict(self.config.final_config_dict) if hasattr(self.config, 'final_config_dict') else None,
            'epoch': epoch,
            'cur_step': self.cur_step,
            'best_valid_score': self.best_valid_score,
            'state_dict': self.model.state_dict(),
            'other_parameter': self.model.other_parameter(),
            'optimizer': opt_state,
        }
        torch.save(state, saved_model_file)
        if verbose:
            self.logger.info('Saving current: %s', saved_model_file)

    def resume_checkpoint(self, resume_file):
        resume_file = str(resume_file)
        self.saved_model_file = resume_file
        checkpoint = torch.load(resume_file, weights_only=False)
        self.start_epoch = checkpoint['epoch'] + 1
        self.cur_step = checkpoint['cur_step']
        self.best_valid_score = checkpoint['best_valid_score']
        self.model.load_state_dict(checkpoint['state_dict'])
        self.model.load_other_parameter(checkpoint.get('other_parameter'))
        # a checkpoint written by the reference carries torch.optim.Adam's integer parameter indices: its optimizer was
        # built over model.parameters() (trainer.py:129), so index k is the k-th named parameter
        self.optimizer.load_state_dict(checkpoint['optimizer'],
                                       param_names=[n for n, _ in self.model.named_parameters()])
        self.logger.info('Checkpoint loaded. Resume training from epoch %d', self.start_epoch)

    def _generate_train_loss_output(self, epoch_idx, s_time, e_time, losses):
        des = self.config['loss_decimal_place'] or 4
        out = 'epoch %d training [time: %.2fs, ' % (epoch_idx, e_time - s_time)
        if isinstance(losses, tuple):
            out += ', '.join(('train_loss%d: %.' + str(des) + 'f') % (i + 1, l) for i, l in enumerate(losses))
        else:
            out += ('train loss: %.' + str(des) + 'f') % losses
        return out + ']'

    def fit(self, train_data, valid_data=None, verbose=True, saved=True, show_progress=False, callback_fn=None):
        """reference trainer.py:332-418: returns (best_valid_score, best_valid_result)."""
        self._train_data_for_eval = train_data                        # eval_collector.data_collect(train_data), :341
        if saved and self.start_epoch >= self.epochs:
            self._save_checkpoint(-1, verbose=verbose)
        for epoch_idx in range(self.start_epoch, self.epochs):
            t0 = time()
            train_loss = self._train_epoch(train_data, epoch_idx, show_progress=show_progress)
            self.train_loss_dict[epoch_idx] = sum(train_loss) if isinstance(train_loss, tuple) else train_loss
            if verbose:
                self.logger.info(self._generate_train_loss_output(epoch_idx, t0, time(), train_loss))
            if self.eval_step <= 0 or not valid_data:
                if saved:
                    self._save_checkpoint(epoch_idx, verbose=verbose)
                continue
            if (epoch_idx + 1) % self.eval_step == 0:
                valid_score, valid_result = self._valid_epoch(valid_data, show_progress=show_progress)
                self.best_valid_score, self.cur_step, stop_flag, update_flag = early_stopping(
                    valid_score, self.best_valid_score, self.cur_step, max_step=self.stopping_step,
                    bigger=self.valid_metric_bigger)
                if verbose:
                    self.logger.info('epoch %d evaluating [valid_score: %f] %s', epoch_idx, valid_score,
                                     dict2str(valid_result))
                if update_flag:
                    if saved:
                        self._save_checkpoint(epoch_idx, verbose=verbose)
                    self.best_valid_result = valid_result
                if callback_fn:
                    callback_fn(epoch_idx, valid_score)
                if stop_flag:
                    break
        # store embedding and sst if the task needs an attacker after training (trainer.py:413-415)
        if self.config['save_sst_embed'] and os.path.exists(self.saved_model_file):
            self._save_sst_embed(train_data)
        return self.best_valid_score, self.best_valid_result

    def _reload_for_sst_embed(self):
        ck = torch.load(self.saved_model_file, weights_only=False)
        self.model.load_state_dict(ck['state_dict'])
        self.model.load_other_parameter(ck.get('other_parameter'))
        self.model.eval()

    def _save_sst_embed(self, data):
        """trainer.py:242-256: the saved model's user embeddings next to the users' sensitive attributes, for an attacker
        trained afterwards.  (The reference reloads the checkpoint first; the dict-held filter MLPs are not in it, are never
        put in eval mode and take one more BatchNorm batch from this pass -- SURVEY.md App. B-3 -- and so they do here.)"""
        self._reload_for_sst_embed()
        user_features = data.dataset.get_user_feature()
        torch.save(self.model.get_sst_embed(user_features[1:]), self.saved_sst_embed_file)

    # --- full-sort ranking evaluation on the device (trainer.py:420-438, :458-515) ----------------------------
    def _full_sort_scores(self, interaction, n_items, sst_list=None):
        """[users, n_items] scores of a batch of users: `model.full_sort_predict`, or -- as the reference does when a
        model has none (trainer.py:425-433) -- `predict` on every (user, item) pair, in chunks."""
        extra = () if sst_list is None else (sst_list,)
        from ..model.abstract_recommender import AbstractRecommender
        if type(self.model).full_sort_predict is not AbstractRecommender.full_sort_predict:
            try:
                return self.model.full_sort_predict(interaction, *extra).view(-1, n_items)
            except NotImplementedError:
                pass
        U = len(interaction)
        items = torch.arange(n_items, device=self.device)
        out = torch.empty((U, n_items), dtype=torch.float32, device=self.device)
        per = max(int(self.config['eval_batch_size'] or 4096) // n_items, 1)
        iid = self.config['ITEM_ID_FIELD']
        for lo in range(0, U, per):
            part = interaction[lo:lo + per].repeat_interleave(n_items)
            part.update(type(part)({iid: items.repeat(min(per, U - lo))}))
            out[lo:lo + per] = self.model.predict(part, *extra).view(-1, n_items)
        return out

    def _ranking_evaluate(self, eval_data, sst_lists=(None,)):
        """One result over everything collected: every batch scored once per entry of `sst_lists` (None = no filter
        argument; the filtered models' validation pools all attribute subsets, trainer.py:1005-1022)."""
        from ..data.dataloader import NegSampleEvalDataLoader
        from ..evaluator import Collector, Evaluator
        collector, evaluator = Collector(self.config), Evaluator(self.config)
        if getattr(self, '_train_data_for_eval', None) is not None:
            collector.data_collect(self._train_data_for_eval)         # catalogue size / training popularity (:341)
        n_items = eval_data.dataset.item_num
        if isinstance(eval_data, NegSampleEvalDataLoader):             # uniN: trainer.py:440-456
            per = int(self.config['eval_batch_size'] or 4096)
            for interaction, row_idx, positive_u, positive_i in eval_data:
                for sst_list in sst_lists:
                    extra = () if sst_list is None else (sst_list,)
                    scores = torch.cat([self.model.predict(interaction[lo:lo + per], *extra).view(-1)
                                        for lo in range(0, len(interaction), per)])
                    collector.eval_batch_collect_candidates(scores, row_idx, interaction, positive_u, positive_i, n_items)
            return OrderedDict(evaluator.evaluate(collector.get_data_struct()))
        for user_df, (hist_u, hist_i), positive_u, positive_i in eval_data:
            user_df = user_df.to(self.device)
            for sst_list in sst_lists:
                scores = self._full_sort_scores(user_df, n_items, sst_list)
                scores[:, 0] = -float('inf')                           # [PAD], trainer.py:435
                scores[hist_u, hist_i] = -float('inf')                 # items of earlier phases, :436-437
                collector.eval_batch_collect(scores, user_df, positive_u, positive_i)
        return OrderedDict(evaluator.evaluate(collector.get_data_struct()))

    def _load_for_eval(self, load_best_model, model_file):
        if load_best_model:
            checkpoint = torch.load(model_file or self.saved_model_file, weights_only=False)
            self.model.load_state_dict(checkpoint['state_dict'])
            self.model.load_other_parameter(checkpoint.get('other_parameter'))
        self.model.eval()

    @torch.no_grad()
    def evaluate(self, eval_data, load_best_model=False, model_file=None, show_progress=False):
        """A FullSortEvalDataLoader / NegSampleEvalDataLoader gets the reference's ranking evaluation (top-k and
        fairness metrics named in `config['metrics']`, fairrec/evaluator); a plain loader of (user, item, rating)
        batches the value-type RMSE / MAE of `model.predict`."""
        if not eval_data:
            return None
        from ..data.dataloader import FullSortEvalDataLoader, NegSampleEvalDataLoader
        if isinstance(eval_data, (FullSortEvalDataLoader, NegSampleEvalDataLoader)):
            self._load_for_eval(load_best_model, model_file)
            return self._ranking_evaluate(eval_data)
        if load_best_model:
            checkpoint = torch.load(model_file or self.saved_model_file, weights_only=False)
            self.model.load_state_dict(checkpoint['state_dict'])
            self.model.load_other_parameter(checkpoint.get('other_parameter'))
        self.model.eval()
        se = torch.zeros((), device=self.device)
        ae = torch.zeros((), device=self.device)
        n = 0
        rating_field = self.config['RATING_FIELD']
        max_rating = float(getattr(self.model, 'max_rating', 1.0))
        for interaction in eval_data:
            interaction = interaction.to(self.device)
            score = self.model.predict(interaction).view(-1) * max_rating
            err = score - interaction[rating_field].to(torch.float32)
            se += (err * err).sum()
            ae += err.abs().sum()
            n += err.numel()
        return OrderedDict(rmse=float((se / max(n, 1)).sqrt().item()), mae=float((ae / max(n, 1)).item()))


class PFCNTrainer(Trainer):
    """Alternating filter / discriminator schedule of recbole/trainer/trainer.py:865-930: per epoch draw a random
    non-empty subset of the sensitive attributes (`np.random.choice([0,1], sst_num)`, same RNG consumption), every
    `train_epoch_interval` epochs one pass with `optimizer_filter` on `calculate_loss`, then always one pass with
    `optimizer_dis` on `calculate_dis_loss`."""

    def __init__(self, config, model):
        self.filter_mode = config['filter_mode'].lower()
        super().__init__(config, model)
        self.train_epoch_interval = config['train_epoch_interval']
        if self.filter_mode != 'none':
            self.sst_num = len(self.config['sst_attr_list'])
            self.mask_label = {i: sst for i, sst in enumerate(self.config['sst_attr_list'])}
            # which tensors each optimizer owns is declared by the model's engine groups (trainer.py:1201-1235)
            self.optimizer_filter = self._build_optimizer(group='filter')
            self.optimizer_dis = self._build_optimizer(group='dis')

    def _build_optimizer(self, **kwargs):
        group = kwargs.pop('group', None)
        if group is None and self.filter_mode != 'none':
            return None           # the reference's default optimizer is never stepped when filters are on (PFCN, FairGo)
        from ..optim import FusedLazyAdam
        if (kwargs.get('learner', self.learner) or 'adam').lower() != 'adam':
            raise NotImplementedError('only learner adam is on the MI355X hot path')
        return FusedLazyAdam(self.model.hip_engine(), lr=kwargs.get('learning_rate', self.learning_rate),
                             weight_decay=kwargs.get('weight_decay', self.weight_decay),
                             sweep_period=self.config['lazy_adam_sweep_period'], group=group,
                             clip_grad_norm=self.clip_grad_norm)

    def _train_epoch(self, train_data, epoch_idx, loss_func=None, show_progress=False):
        dis_loss, filter_loss = 0., 0.
        if self.filter_mode != 'none':
            mask = np.zeros(self.sst_num)
            with host_numpy_stream():      # same position in numpy's stream as the reference, also with a device feed
                while mask.sum() == 0:
                    mask = np.random.choice([0, 1], self.sst_num)
            sst_list = [sst for i, sst in self.mask_label.items() if mask[i] != 0]
            if epoch_idx % self.config['train_epoch_interval'] == 0:
                self.optimizer = self.optimizer_filter
                filter_loss = self._train_epoch_with_mask(train_data, epoch_idx, self.model.calculate_loss, sst_list)
            self.optimizer = self.optimizer_dis
            if hasattr(self.model, 'begin_dis_phase'):     # what does not move during a discriminator pass is computed once
                self.model.begin_dis_phase(sst_list)
            dis_loss = self._train_epoch_with_mask(train_data, epoch_idx, self.model.calculate_dis_loss, sst_list)
            return filter_loss, dis_loss
        return self._train_epoch_with_mask(train_data, epoch_idx, self.model.calculate_loss, None)

    def _train_epoch_with_mask(self, train_data, epoch_idx, loss_func=None, sst_list=None, show_progress=False):
        self.model.train()
        total = None
        graphed = self._graphed_step(('mask', getattr(loss_func, '__name__', str(loss_func)), tuple(sst_list or ())),
                                     lambda inter: loss_func(inter, sst_list))
        for interaction in train_data:
            interaction = interaction.to(self.device)
            if graphed is not None:
                total = self._accumulate(total, graphed(interaction))
                continue
            self.optimizer.zero_grad()
            loss = loss_func(interaction, sst_list)
            total = self._accumulate(total, loss)
            loss.backward()
            self.optimizer.step()
        if total is None:
            return 0.0
        acc = total.cpu().tolist()                  # the pass's only host sync
        val = float(acc[0])
        self._check_nan(torch.tensor(val), first_bad_step=int(acc[4]))
        self.model.hip_engine().check_device_errors()
        return val

    def _save_checkpoint(self, epoch, verbose=True, **kwargs):
        """trainer.py:1133-1154: same keys, with optimizer_filter / optimizer_dis when filters are on."""
        saved_model_file = kwargs.pop('saved_model_file', self.saved_model_file)
        state = {
            'config': dict(self.config.final_config_dict), 'epoch': epoch, 'cur_step': self.cur_step,
            'best_valid_score': self.best_valid_score, 'state_dict': self.model.state_dict(),
            'other_parameter': self.model.other_parameter(),
        }
        if self.filter_mode != 'none':
            state['optimizer_filter'] = self.optimizer_filter.state_dict()
            state['optimizer_dis'] = self.optimizer_dis.state_dict()
        else:
            state['optimizer'] = self.optimizer.state_dict()
        torch.save(state, saved_model_file)

    def resume_checkpoint(self, resume_file):
        """trainer.py:1156-1186 as intended: the two optimizers of the filtered modes, the plain one for filter_mode none
        (the reference has the two branches swapped, SURVEY.md App. B-10, and raises on either)."""
        resume_file = str(resume_file)
        self.saved_model_file = resume_file
        checkpoint = torch.load(resume_file, weights_only=False)
        self.start_epoch = checkpoint['epoch'] + 1
        self.cur_step = checkpoint['cur_step']
        self.best_valid_score = checkpoint['best_valid_score']
        self.model.load_state_dict(checkpoint['state_dict'])
        self.model.load_other_parameter(checkpoint.get('other_parameter'))
        if self.filter_mode != 'none':
            self.optimizer_filter.load_state_dict(checkpoint['optimizer_filter'])
            self.optimizer_dis.load_state_dict(checkpoint['optimizer_dis'])
        else:
            self.optimizer.load_state_dict(checkpoint['optimizer'],
                                           param_names=[n for n, _ in self.model.named_parameters()])
        self.logger.info('Checkpoint loaded. Resume training from epoch %d', self.start_epoch)

    def _save_sst_embed(self, data):
        """trainer.py:1108-1131: one file per attribute subset (the filtered modes), `<model>_embed-<mode>-[a_b].pth`."""
        import itertools
        self._reload_for_sst_embed()
        user_features = data.dataset.get_user_feature()[1:]
        if self.filter_mode != 'none':
            for i in range(1, 4):
                for attr_list in (list(c) for c in itertools.combinations(self.config['sst_attr_list'], i)):
                    name = '{}_embed-{}-[{}].pth'.format(self.config['model'], self.config['filter_mode'], '_'.join(attr_list))
                    torch.save(self.model.get_sst_embed(user_features, attr_list), os.path.join(self.checkpoint_dir, name))
        else:
            name = '{}_embed-{}.pth'.format(self.config['model'], self.config['filter_mode'])
            torch.save(self.model.get_sst_embed(user_features), os.path.join(self.checkpoint_dir, name))

    def _subsets(self):
        import itertools
        attrs = self.config['sst_attr_list']
        return [list(sub) for r in range(1, len(attrs) + 1) for sub in itertools.combinations(attrs, r)]

    @torch.no_grad()
    def pfcn_evaluate(self, eval_data, load_best_model=False, model_file=None, show_progress=False):
        """trainer.py:968-1027 (validation during training): ONE result over the batches of every attribute subset."""
        from ..data.dataloader import FullSortEvalDataLoader, NegSampleEvalDataLoader
        if not eval_data:
            return None
        if not isinstance(eval_data, (FullSortEvalDataLoader, NegSampleEvalDataLoader)):
            raise NotImplementedError("evaluation of the filtered models needs a ranking evaluation loader")
        self._load_for_eval(load_best_model, model_file)
        return self._ranking_evaluate(eval_data, self._subsets() if self.filter_mode != 'none' else (None,))

    def _valid_epoch(self, valid_data, show_progress=False):
        valid_result = self.pfcn_evaluate(valid_data, load_best_model=False, show_progress=show_progress)
        return calculate_valid_score(valid_result, self.valid_metric), valid_result

    @torch.no_grad()
    def evaluate(self, eval_data, load_best_model=False, model_file=None, show_progress=False):
        """trainer.py:1047-1106: one result per non-empty subset of the sensitive attributes, keyed
        '<filter_mode>-<subset>' (filters on), or {'<filter_mode>': result}."""
        from ..data.dataloader import FullSortEvalDataLoader, NegSampleEvalDataLoader
        if eval_data and isinstance(eval_data, (FullSortEvalDataLoader, NegSampleEvalDataLoader)):
            self._load_for_eval(load_best_model, model_file)
            final = {}
            if self.filter_mode != 'none':
                for sub in self._subsets():
                    final['{}-{}'.format(self.config['filter_mode'] or self.filter_mode, sub)] = \
                        self._ranking_evaluate(eval_data, (sub,))
            else:
                final[self.config['filter_mode']] = self._ranking_evaluate(eval_data)
            return final
        raise NotImplementedError("evaluation of the filtered models needs a ranking evaluation loader "
                                  "(FullSortEvalDataLoader / NegSampleEvalDataLoader)")


class PFCN_PMFTrainer(PFCNTrainer):
    pass


class PFCN_BiasedMFTrainer(PFCNTrainer):
    pass


class PFCN_MLPTrainer(PFCNTrainer):
    pass


class PFCN_DMFTrainer(PFCNTrainer):
    pass


class FairGoTrainer(PFCNTrainer):
    """recbole/trainer/trainer.py:534-736: `pretrain_epochs` of plain MF regression with optimizer_pretrain, a pretrain
    checkpoint, then the same alternating filter / discriminator schedule as PFCN with optimizer_filter /
    optimizer_dis; `model.train_stage` is switched by the trainer."""

    def __init__(self, config, model):
        self.filter_mode = 'fairgo'            # always filtered: reuse PFCNTrainer's alternating epoch
        Trainer.__init__(self, config, model)
        self.train_epoch_interval = config['train_epoch_interval']
        self.sst_num = len(self.config['sst_attr_list'])
        self.mask_label = {i: sst for i, sst in enumerate(self.config['sst_attr_list'])}
        self.load_pretrain_weight = config['load_pretrain_weight']
        self.optimizer_filter = self._build_optimizer(group='filter')
        self.optimizer_dis = self._build_optimizer(group='dis')
        self.optimizer_pretrain = None
        self.saved_sst_embed_file = os.path.join(self.checkpoint_dir, '{}-{}_embed-[{}].pth'.format(
            config['model'], config['aggr_method'], '_'.join(config['sst_attr_list'])))     # trainer.py:557-559
        if config['pretrain_model_file_path'] is not None:
            ck = torch.load(config['pretrain_model_file_path'], weights_only=False)
            self.model.load_state_dict(ck['state_dict'])
            self.model.load_other_parameter(ck.get('other_parameter'))
            self.model.train_stage = 'finetune'
        elif self.load_pretrain_weight:
            self.model.train_stage = 'finetune'
        else:
            self.model.train_stage = 'pretrain'
            self.pretrain_epochs = config['pretrain_epochs']
            self.optimizer_pretrain = self._build_optimizer(group='pretrain')

    _valid_epoch = Trainer._valid_epoch      # the reference's FairGoTrainer evaluates like the plain Trainer (:738-772)

    def _train_epoch(self, train_data, epoch_idx, loss_func=None, show_progress=False):
        """trainer.py:687-704: the same alternating epoch as PFCN's, but the reference's FairGoTrainer returns the pair as
        (dis_loss, filter_loss) -- the order its epoch log lines and train_loss_dict consumers see."""
        filter_loss, dis_loss = PFCNTrainer._train_epoch(self, train_data, epoch_idx, loss_func, show_progress)
        return dis_loss, filter_loss

    def save_pretrained_model(self, saved_model_file):
        self._pretrain_file_written = True
        torch.save({'config': dict(self.config.final_config_dict), 'state_dict': self.model.state_dict(),
                    'optimizer': self.optimizer.state_dict(), 'other_parameter': self.model.other_parameter()},
                   saved_model_file)

    def pretrain(self, train_data, valid_data=None, verbose=True, saved=True, show_progress=False):
        """trainer.py:606-685: `pretrain_epochs` of the plain regression with optimizer_pretrain; WITH validation data every
        `eval_step`-th epoch is validated (the evaluation loader's negatives come from the numpy stream the training
        negatives come from, so this moves every later batch), early stopping runs on the trainer's usual counters, the
        pretrain checkpoint is written on improvement -- without validation data after every epoch -- and read back at the
        end (the best pretrain model, not the last one, enters the finetune stage)."""
        self.saved_pretrain_model_file = os.path.join(
            self.checkpoint_dir, '{}-{}-pretrain.pth'.format(self.config['model'], self.config['dataset']))
        self.eval_step = min(self.config['eval_step'] if self.config['eval_step'] is not None else 1, self.pretrain_epochs)
        self.optimizer = self.optimizer_pretrain
        self._pretrain_file_written = False
        self._train_data_for_eval = train_data                        # eval_collector.data_collect(train_data), :621
        for epoch_idx in range(self.start_epoch, self.pretrain_epochs):
            t0 = time()
            loss = self._train_epoch_with_mask(train_data, epoch_idx, self.model.calculate_loss, None)
            self.train_loss_dict[epoch_idx] = loss
            if verbose:
                self.logger.info(self._generate_train_loss_output(epoch_idx, t0, time(), loss))
            if self.eval_step <= 0 or not valid_data:
                if saved:
                    self.save_pretrained_model(self.saved_pretrain_model_file)
                continue
            if (epoch_idx + 1) % self.eval_step == 0:
                valid_score, valid_result = self._valid_epoch(valid_data, show_progress=show_progress)
                self.best_valid_score, self.cur_step, stop_flag, update_flag = early_stopping(
                    valid_score, self.best_valid_score, self.cur_step, max_step=self.stopping_step,
                    bigger=self.valid_metric_bigger)
                if verbose:
                    self.logger.info('epoch %d evaluating [valid_score: %f] %s', epoch_idx, valid_score, dict2str(valid_result))
                if update_flag:
                    if saved:
                        self.save_pretrained_model(self.saved_pretrain_model_file)
                    self.best_valid_result = valid_result
                if stop_flag:
                    break
        # (the reference reads the file unconditionally, :677 -- also one an EARLIER run left in checkpoint_dir when this run wrote
        # none: saved=False, no pretrain epoch, no improvement.  Only what this run wrote is read back; otherwise the parameters
        # just trained enter the finetune stage, and the log says so)
        if os.path.exists(self.saved_pretrain_model_file) and not self._pretrain_file_written:
            self.logger.info('pretrain: %s is not of this run -- not loaded', self.saved_pretrain_model_file)
        if os.path.exists(self.saved_pretrain_model_file) and self._pretrain_file_written:
            ck = torch.load(self.saved_pretrain_model_file, weights_only=False)
            self.model.load_state_dict(ck['state_dict'])
            self.model.load_other_parameter(ck.get('other_parameter'))
            if self.config['save_sst_embed']:      # trainer.py:681-682
                self._save_sst_embed(train_data, os.path.join(self.checkpoint_dir, '{}-{}-pretrain_embed[none].pth'.format(
                    self.config['model'], self.config['dataset'])))
        return self.best_valid_score, self.best_valid_result

    def reset_params(self):
        """trainer.py:560-577: the finetune stage starts from fresh counters."""
        self.epochs = self.config['epochs']
        self.eval_step = min(self.config['eval_step'] if self.config['eval_step'] is not None else 1, self.epochs)
        self.start_epoch, self.cur_step = 0, 0
        self.best_valid_score = -np.inf if self.valid_metric_bigger else np.inf
        self.best_valid_result = None
        self.train_loss_dict = dict()
        self.model.train_stage = 'finetune'

    def fit(self, train_data, valid_data=None, verbose=True, saved=True, show_progress=False, callback_fn=None):
        if self.model.train_stage == 'pretrain':
            self.pretrain(train_data, valid_data, verbose, saved, show_progress)
            self.reset_params()
        elif self.model.train_stage != 'finetune':
            raise ValueError("Please make sure that the 'train_stage' is 'pretrain' or 'finetune'!")
        return Trainer.fit(self, train_data, valid_data, verbose, saved, show_progress, callback_fn)

    @torch.no_grad()
    def evaluate(self, eval_data, load_best_model=False, model_file=None, show_progress=False):
        """trainer.py:738-772: during training the plain evaluation of the current stage; with `load_best_model` the
        pretrain checkpoint ('pretrain-<metric>') and the finetuned one ('finetune-<metric>')."""
        if not eval_data:
            return None
        if not load_best_model:
            return Trainer.evaluate(self, eval_data, False, None, show_progress)
        result = OrderedDict()
        stages = ([] if self.load_pretrain_weight else [('pretrain', self.saved_pretrain_model_file)]) + \
            [('finetune', model_file or self.saved_model_file)]
        for stage, path in stages:
            self.model.train_stage = stage
            for key, value in Trainer.evaluate(self, eval_data, True, path, show_progress).items():
                result[f'{stage}-{key}'] = value
        return result

    def _save_sst_embed(self, data, saved_sst_embed_file=None):
        """trainer.py:774-782: no checkpoint reload here; every attribute of `mask_label` at once."""
        self.model.eval()
        user_features = data.dataset.get_user_feature()[1:]
        stored = self.model.get_sst_embed(user_features, [a for a in self.mask_label.values()])
        torch.save(stored, saved_sst_embed_file or self.saved_sst_embed_file)

    def _save_checkpoint(self, epoch, verbose=True, **kwargs):
        saved_model_file = kwargs.pop('saved_model_file', self.saved_model_file)
        torch.save({'config': dict(self.config.final_config_dict), 'epoch': epoch, 'cur_step': self.cur_step,
                    'best_valid_score': self.best_valid_score, 'state_dict': self.model.state_dict(),
                    'other_parameter': self.model.other_parameter(), 'optimizer_filter': self.optimizer_filter.state_dict(),
                    'optimizer_dis': self.optimizer_dis.state_dict()}, saved_model_file)


class FairGo_PMFTrainer(FairGoTrainer):
    pass


class FairGo_GCNTrainer(FairGoTrainer):
    pass
