"""The training loop of ``morgana.experiment_builder.ExperimentBuilder`` restated around the HIP hot path.

Reference: ``train_epoch`` experiment_builder.py:431-505 (loop body :464-494), ``run_train`` :507-560, model / EMA
construction :267-281, :386-396.  Only what the hot path needs is mirrored (batch loop, Adam, LR schedules, EMA,
loss bookkeeping, checkpoints); CLI, logging, plotting and the file data loaders are out of scope.

Differences that matter on a 0.3-1 ms step: the batch loss stays on the device (the reference calls ``.item()`` and
formats the loss every batch: >= 3 host syncs per step, experiment_builder.py:480-490) and is read once per epoch;
``zero_grad`` / all-reduce / Adam work on flat buffers (morgana_amd.optim.Adam).
"""
import copy
import json
import os

import torch

from . import functional as F_hip
from . import lr_schedules
from . import ops
from . import utils
from .optim import Adam


class ExperimentBuilder(object):
    def __init__(self, model_class, model_kwargs=None, learning_rate=0.01, weight_decay=0., lr_schedule_name='constant',
                 lr_schedule_kwargs=None, ema_decay=0., device='cuda:0', start_epoch=1, end_epoch=50,
                 experiment_dir=None, model_checkpoint_interval=1, checkpoint_path=None, use_graphs=False, graph_group=10, **unused):
        self.model_class = model_class
        self.model_kwargs = model_kwargs or {}
        self.learning_rate = learning_rate
        self.weight_decay = weight_decay
        self.lr_schedule_name = lr_schedule_name
        self.lr_schedule_kwargs = lr_schedule_kwargs or {}
        self.ema_decay = ema_decay
        self.device = device
        self.start_epoch = start_epoch
        self.end_epoch = end_epoch
        self.epoch = start_epoch
        self.experiment_dir = experiment_dir
        self.model_checkpoint_interval = model_checkpoint_interval
        self.analysis_kwargs = {}
        self.use_graphs = use_graphs          # replay repeated batch shapes as HIP graphs (morgana_amd/graphs.py); off = eager launches
        # use_graphs with a RESIDENT loader (a list / tuple of device batches kept from epoch to epoch): this many consecutive batches are
        # captured into one graph that reads them in place (graphs.GraphedStepCache.step_group); 1 = a graph launch and a load per batch
        self.graph_group = int(graph_group)
        self._graph_cache = None
        self._lr_schedule = lr_schedules.init_lr_schedule(lr_schedule_name, **self.lr_schedule_kwargs)

        self.model = self.build_model(model_class, self.model_kwargs, checkpoint_path)      # :267, :386-396
        if self.ema_decay:                                                                  # :276-281
            self.ema_model = copy.deepcopy(self.model)
            self.ema = utils.ExponentialMovingAverage(self.ema_model, self.ema_decay)

    def build_model(self, model_class, model_kwargs, checkpoint_path=None):
        model = model_class(**model_kwargs)
        model = model.to(self.device)
        if checkpoint_path:
            model.load_parameters(checkpoint_path, device=self.device)
        return model

    def make_optimizer(self, **kwargs):
        # fused_loop: train_epoch below is the reference's loop body (zero_grad, forward, backward, step) and nothing else touches .grad
        kwargs.setdefault('fused_loop', True)
        return Adam(self.model.parameters(), lr=self.learning_rate, weight_decay=self.weight_decay, **kwargs)  # :516

    def _cache_for(self, optimizer):
        if self._graph_cache is None or self._graph_cache.optimizer is not optimizer:
            from . import graphs
            self._graph_cache = graphs.GraphedStepCache(self.model, optimizer)
        return self._graph_cache

    def _resident_group(self, data_loader, lr_schedule, gen_output):
        """Steps per captured graph for this epoch: ``graph_group`` when the loader is a list / tuple of batches that live on the device
        and nothing has to happen between two steps on the host (a per-batch LR schedule, EMA, per-batch analysis output), else 1."""
        if not self.use_graphs or self.graph_group <= 1 or not isinstance(data_loader, (list, tuple)) or len(data_loader) < 2:
            return 1
        if gen_output or self.ema_decay:
            return 1
        if lr_schedule is not None and self.lr_schedule_name in lr_schedules.BATCH_LR_SCHEDULES:
            return 1
        for features in data_loader:
            if not isinstance(features, dict) or not all(v.is_cuda for v in features.values() if isinstance(v, torch.Tensor)):
                return 1
        return self.graph_group

    def train_epoch(self, data_loader, optimizer, lr_schedule=None, gen_output=False, out_dir=None):
        """One pass over ``data_loader`` (an iterable of feature dicts already on the device); returns the mean loss."""
        self.model.mode = 'train'
        self.model.metrics.reset_state('train')
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)

        # the loader half of bf16 mode: the batches carry the bf16 operand table of the model's phone-level input (data.DeviceBatches
        # writes it in the pass that pads and normalises), so the step below launches no cast of the 49 MB table
        if hasattr(data_loader, 'use_bf16_tables'):
            data_loader.use_bf16_tables(self.model.bf16_table_features())

        loss = None
        n_batches = len(data_loader)
        i = -1
        import time
        t_loop = time.perf_counter()
        loss_log = None
        group = self._resident_group(data_loader, lr_schedule, gen_output)
        if group > 1:
            # resident epoch: K steps per graph launch, the batches read where they lie (no load launch); the K losses filed by one launch
            cache = self._cache_for(optimizer)
            loss_log = torch.zeros(max(n_batches, 1), dtype=torch.float32, device=self.device)
            for start in range(0, n_batches, group):
                chunk = data_loader[start:start + group]
                self.model.step = (self.epoch - 1) * n_batches + start + len(chunk)
                cache.step_group(chunk, [loss_log[start + j] for j in range(len(chunk))])
            i, ahead = n_batches - 1, None
        else:
            batches = iter(data_loader)
            ahead = next(batches, None)                 # one batch of look-ahead (the graph cache loads it beside the current step)
        while ahead is not None:
            i, features = i + 1, ahead
            ahead = next(batches, None)
            self.model.step = (self.epoch - 1) * n_batches + i + 1

            if self.use_graphs:
                # the same four calls, captured once per batch shape and replayed (graphs.GraphedStepCache)
                self._cache_for(optimizer)
                if loss_log is None:
                    loss_log = torch.zeros(max(n_batches, 1), dtype=torch.float32, device=self.device)
                slot = loss_log[i] if i < loss_log.numel() else None
                batch_loss, output_features = self._graph_cache.step(features, clone_loss=False, loss_slot=slot)
                filed = slot is not None and self._graph_cache._pending_loss is not None      # rides in the next batch's load launch
                self._graph_cache.prefetch(ahead)                                # (MORGANA_GRAPH_PREFETCH: measured, off)
            else:
                optimizer.zero_grad()                                            # :468
                batch_loss, output_features = self.model(features)               # :471
                F_hip.backward(batch_loss)                                       # :473 (loss.backward(), cached unit gradient)
                optimizer.step()                                                 # :474

            if lr_schedule is not None and self.lr_schedule_name in lr_schedules.BATCH_LR_SCHEDULES:
                lr_schedule.step()                                               # :477-478

            # :480, :487 - the batch losses are filed on the device, one slot per batch (ONE small launch per step where the running sum, the
            # metric's sum and its add and a copy of the replayed graph's loss were four); the epoch's sum and the 'loss' metric are taken
            # from the log once, behind the loop: the same sums (metrics.Mean: sum of the batch losses / their count)
            if loss_log is None:
                loss_log = torch.zeros(max(n_batches, 1), dtype=torch.float32, device=batch_loss.device)
            if i >= loss_log.numel():                                            # a loader that yields more than its len(): grow
                if self.use_graphs and self._graph_cache is not None:
                    self._graph_cache.flush()
                loss_log = torch.cat((loss_log, torch.zeros_like(loss_log)))
            if not (self.use_graphs and filed):
                loss_log[i].copy_(batch_loss.detach().reshape(()))

            if self.ema_decay:
                self.ema.update_params(self.model)                               # :483-484

            if gen_output:
                self.model.analysis_for_train_batch(features, output_features, out_dir=out_dir,
                                                    **self.analysis_kwargs)
        # how long the host took to ISSUE the epoch's steps (the device may still be working: the one sync of the epoch comes below)
        self.last_epoch_stats = {'steps': i + 1, 'host_issue_s': time.perf_counter() - t_loop}
        if self.use_graphs and self._graph_cache is not None:
            self._graph_cache.flush()                                            # the last step's loss into its slot
        if loss_log is not None:
            self.model.metrics.accumulate(self.model.mode, loss=loss_log[:i + 1])
            loss = loss_log[:i + 1].sum()
        if gen_output:
            self.model.analysis_for_train_epoch(out_dir=out_dir, **self.analysis_kwargs)
        if out_dir:
            with open(os.path.join(out_dir, 'metrics.json'), 'w') as f:          # :499-501
                json.dump(self.model.metrics.results_as_json_dict('train'), f)
        self.model.mode = ''
        ops.check_persistent_status()                                            # persistent recurrent kernels: any time-out?
        return float(loss.item()) / (i + 1)                                      # :505 (one sync per epoch)

    def _eval_model(self, model):
        """The model an evaluation pass runs: the one given, else the EMA twin when EMA is on, else the trained one (:629-632)."""
        if model is not None:
            return model
        return self.ema_model if self.ema_decay else self.model

    @torch.no_grad()
    def valid_epoch(self, data_loader, model=None, gen_output=False, out_dir=None):
        """One evaluation pass: ``model(features)`` without backward / step, the batch loss into ``metrics`` under 'valid', the
        ``analysis_for_valid_*`` hooks when output is requested, ``metrics.json``; returns the mean loss (experiment_builder.py:562-620).
        As in ``train_epoch`` the loss stays on the device and is read once per epoch (the reference reads it every batch, :596)."""
        model = self._eval_model(model)
        model.mode = 'valid'
        model.metrics.reset_state('valid')
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
        loss, n_batches, i = None, len(data_loader), -1
        for i, features in enumerate(data_loader):
            self.model.step = (self.epoch - 1) * n_batches + i + 1
            batch_loss, output_features = model(features)
            batch_loss = batch_loss.detach()
            loss = batch_loss if loss is None else loss + batch_loss
            model.metrics.accumulate(model.mode, loss=batch_loss)
            if gen_output:
                model.analysis_for_valid_batch(features, output_features, out_dir=out_dir, **self.analysis_kwargs)
        if gen_output:
            model.analysis_for_valid_epoch(out_dir=out_dir, **self.analysis_kwargs)
        if out_dir:
            with open(os.path.join(out_dir, 'metrics.json'), 'w') as f:
                json.dump(model.metrics.results_as_json_dict('valid'), f)
        model.mode = ''
        ops.check_persistent_status()
        return float(loss.item()) / (i + 1)

    @torch.no_grad()
    def test_epoch(self, data_loader, model=None, out_dir=None):
        """Generation pass: ``model.predict(features)`` and the ``analysis_for_test_*`` hooks, no loss (experiment_builder.py:639-680)."""
        model = self._eval_model(model)
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
        model.mode = 'test'
        model.metrics.reset_state('test')
        n_batches = len(data_loader)
        for i, features in enumerate(data_loader):
            self.model.step = (self.epoch - 1) * n_batches + i + 1
            output_features = model.predict(features)
            model.analysis_for_test_batch(features, output_features, out_dir=out_dir, **self.analysis_kwargs)
        model.analysis_for_test_epoch(out_dir=out_dir, **self.analysis_kwargs)
        if out_dir:
            with open(os.path.join(out_dir, 'metrics.json'), 'w') as f:
                json.dump(model.metrics.results_as_json_dict('test'), f)
        model.mode = ''
        ops.check_persistent_status()

    def run_valid(self, valid_loader, gen_output=False):
        """experiment_builder.py:622-637."""
        out_dir = os.path.join(self.experiment_dir, 'valid', 'epoch_{}'.format(self.epoch)) if self.experiment_dir else None
        return self.valid_epoch(valid_loader, gen_output=gen_output, out_dir=out_dir)

    def run_test(self, test_loader):
        """experiment_builder.py:682-693."""
        out_dir = os.path.join(self.experiment_dir, 'test', 'epoch_{}'.format(self.epoch)) if self.experiment_dir else None
        self.test_epoch(test_loader, out_dir=out_dir)

    def run_train(self, train_loader, valid_loader=None, test_loader=None):
        """Epoch loop of experiment_builder.py:507-560: train, checkpoint, optional evaluation and generation passes, epoch-level
        LR schedule ('plateau' steps on the validation loss, :549-551).  Returns the per-epoch training losses."""
        optimizer = self.make_optimizer()
        lr_schedule = self._lr_schedule(optimizer)
        history = []
        self.valid_history = []
        for self.epoch in range(self.start_epoch, self.end_epoch + 1):
            out_dir = os.path.join(self.experiment_dir, 'train', 'epoch_{}'.format(self.epoch)) \
                if self.experiment_dir else None
            history.append(self.train_epoch(train_loader, optimizer, lr_schedule, out_dir=out_dir))
            if self.experiment_dir and self.epoch % self.model_checkpoint_interval == 0:
                self.model.save_parameters(self.experiment_dir, self.epoch)      # :532-542
                if self.ema_decay:
                    self.ema_model.save_parameters(self.experiment_dir, '{}_ema'.format(self.epoch))
            if valid_loader is not None:
                valid_loss = self.run_valid(valid_loader)                        # :545-551
                self.valid_history.append(valid_loss)
                if self.lr_schedule_name == 'plateau':
                    lr_schedule.step(valid_loss)
            if test_loader is not None:
                self.run_test(test_loader)                                       # :554-556
            if self.lr_schedule_name in lr_schedules.EPOCH_LR_SCHEDULES:
                lr_schedule.step()                                               # :559-560
        return history
