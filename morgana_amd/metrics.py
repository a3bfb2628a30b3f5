"""Mirror of ``morgana.metrics`` for the train loop: ``Handler``, the ``Mean`` loss metric and the streaming metrics the shipped
acoustic model accumulates inside its ``loss`` every step (models/RNN_SPSS.py:44-48, :120-129).

Reference: morgana/metrics.py - ``Handler.accumulate`` :133-153, ``Mean`` :359-397 (sum / (count + 1e-8)), ``RMSE`` :474-499,
``MAE`` :556-576, ``F0Distortion`` / ``LF0Distortion`` :579-634, ``Distortion`` :637-669, ``MelCepDistortion`` :672-694.
Everything stays on the device: the reference pulls the frame count to the host with ``.item()`` in every accumulate call
(metrics.py:394, :610); here a call is two small launches into a (sum, count) accumulator (csrc/metrics.hip) and ``result``
reads it only when asked.  The reference's quirk that the masked count is in FRAMES while the unmasked one is in ELEMENTS
(metrics.py:388-394) is kept.
"""
import math

import torch


class Mean(object):
    """morgana/metrics.py:359-397.  Unmasked calls (the loss bookkeeping of the train loop) are a torch sum; with ``seq_len`` the
    masked sum and the frame count come from the device kernel (csrc/metrics.hip) - no ``.item()`` as at metrics.py:394."""

    def __init__(self, hidden=False):
        self.hidden = hidden
        self.reset_state()

    def reset_state(self):
        self.sum = 0.
        self.count = 0.

    def accumulate(self, tensor, seq_len=None):
        if seq_len is None:
            self.sum = self.sum + torch.sum(tensor.detach())
            self.count = self.count + tensor.numel()
            return
        from . import ops
        accum = torch.zeros(2, dtype=torch.float64, device=tensor.device)
        ops.metric_accumulate(ops.METRIC_MEAN, accum, tensor.detach().to(torch.float32), seq_len=seq_len)
        self.sum = self.sum + accum[0]
        self.count = self.count + accum[1]

    def result(self, *args):
        return self.sum / (self.count + 1e-8)

    def result_as_json(self, *args):
        return float(self.result(*args))

    def __str__(self):
        return '{:.3f}'.format(float(self.result()))


class _DeviceMetric(object):
    """(sum, count) as two doubles on the device; subclasses say which reduction (ops.METRIC_*) feeds them."""
    kind = None

    def __init__(self, hidden=False):
        self.hidden = hidden
        self.reset_state()

    def reset_state(self):
        self._accum = None

    def _add(self, kind, target, pred=None, voiced=None, seq_len=None, col0=0):
        from . import ops
        if self._accum is None:
            self._accum = torch.zeros(2, dtype=torch.float64, device=target.device)
        ops.metric_accumulate(kind, self._accum, target.detach(), None if pred is None else pred.detach(), voiced, seq_len, col0=col0)

    @property
    def sum(self):
        return 0. if self._accum is None else self._accum[0]

    @property
    def count(self):
        return 0. if self._accum is None else self._accum[1]

    def result(self, *args):
        return self.sum / (self.count + 1e-8)

    def result_as_json(self, *args):
        return float(self.result(*args))

    def __str__(self):
        return '{:.3f}'.format(float(self.result()))


class DeviceMean(_DeviceMetric):
    """metrics.Mean with seq_len support (morgana/metrics.py:383-397) - e.g. the V/UV accuracy of the shipped model."""

    def accumulate(self, tensor, seq_len=None):
        from . import ops
        self._add(ops.METRIC_MEAN, tensor, seq_len=seq_len)


class RMSE(_DeviceMetric):
    def accumulate(self, target, pred, seq_len=None):
        from . import ops
        self._add(ops.METRIC_SQDIFF, target, pred, seq_len=seq_len)

    def result(self, *args):
        return (self.sum / (self.count + 1e-8)) ** 0.5


class MAE(_DeviceMetric):
    def accumulate(self, target, pred, seq_len=None):
        from . import ops
        self._add(ops.METRIC_ABSDIFF, target, pred, seq_len=seq_len)


class F0Distortion(RMSE):
    """RMSE over frames that are voiced (and inside seq_len), morgana/metrics.py:597-609."""

    def accumulate(self, f0_target, f0_pred, is_voiced, seq_len=None):
        from . import ops
        self._add(ops.METRIC_SQDIFF_VOICED, f0_target, f0_pred, voiced=is_voiced, seq_len=seq_len)


class LF0Distortion(RMSE):
    """F0 RMSE in Hz from log-F0 inputs, morgana/metrics.py:630-634."""

    def accumulate(self, lf0_target, lf0_pred, is_voiced, seq_len=None):
        from . import ops
        self._add(ops.METRIC_SQDIFF_VOICED_EXP, lf0_target, lf0_pred, voiced=is_voiced, seq_len=seq_len)


class Distortion(_DeviceMetric):
    """Mean per-frame root of the summed squared differences, in dB (morgana/metrics.py:637-669)."""
    log_spec_dB_const = 10. / math.log(10.) * math.sqrt(2.)

    def accumulate(self, target, pred, seq_len=None):
        from . import ops
        self._add(ops.METRIC_ROOT_SQ, target, pred, seq_len=seq_len)

    def result(self, *args):
        return super(Distortion, self).result(*args) * self.log_spec_dB_const


class MelCepDistortion(RMSE):
    """RMSE ignoring c0 (morgana/metrics.py:690-694)."""

    def accumulate(self, target, pred, seq_len=None):
        from . import ops
        self._add(ops.METRIC_SQDIFF, target, pred, seq_len=seq_len, col0=1)


class Handler(object):
    """Container for running a set of metrics, morgana/metrics.py:50-186: named collections ('all', 'train', 'valid', 'test')
    of name -> metric; the constructor's metrics go to 'all', 'train' and 'valid', ``add_metrics('all', ...)`` to every
    collection, and a metric added to several collections is ONE object shared between them, as in the reference."""

    def __init__(self, **metrics):
        self.hidden = False
        self.collections = {'all': metrics, 'train': {}, 'valid': {}, 'test': {}}
        self.metrics = self.collections['all']
        self.add_metrics(('train', 'valid'), **metrics)

    def __getitem__(self, name):
        if name in self.collections:
            return self.collections[name]
        raise ValueError("No collection found by the name {}".format(name))

    def add_metrics(self, collections=('all',), **kwargs):
        if isinstance(collections, str) or not hasattr(collections, '__iter__'):
            collections = [collections]
        if 'all' in collections:
            collections = list(self.collections.keys())
        for collection_name in collections:
            self.collections[collection_name].update(kwargs)
        self.metrics.update(kwargs)

    def add_collection(self, collection, from_collections=tuple()):
        if isinstance(from_collections, str) or not hasattr(from_collections, '__iter__'):
            from_collections = [from_collections]
        self.collections[collection] = {}
        for from_collection in from_collections:
            self[collection].update(self[from_collection])

    def reset_state(self, collection, *args):
        for metric in self[collection].values():
            metric.reset_state()

    def accumulate(self, collection, **kwargs):
        for metric_name, inputs in kwargs.items():
            inputs = list(inputs) if isinstance(inputs, (tuple, list)) else [inputs]          # utils.listify
            if isinstance(inputs[-1], dict):
                inputs, kwinputs = inputs[:-1], inputs[-1]
            else:
                kwinputs = dict()
            self[collection][metric_name].accumulate(*inputs, **kwinputs)

    def result(self, collection='all', *args):
        return {name: metric.result(*args) for name, metric in self[collection].items()}

    def results_as_json_dict(self, collection='all', prefix=''):
        return {prefix + name: metric.result_as_json() for name, metric in self[collection].items() if not metric.hidden}

    def results_as_str_dict(self, collection='all', prefix=''):
        return {prefix + name: str(metric) for name, metric in self[collection].items() if not metric.hidden}

    def __str__(self):
        return ' | '.join('{} = {}'.format(name, value) for name, value in self.results_as_str_dict('all').items())
