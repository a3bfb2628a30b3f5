"""The metrics of the train loop: a registry-based ``Handler``, the ``Mean`` loss metric and the streaming metrics the shipped
acoustic model accumulates inside its ``loss`` every step (models/RNN_SPSS.py:44-48, :120-129).

Reference behaviour: morgana/metrics.py - ``Handler`` :50-186, ``Mean`` :359-397 (sum / (count + 1e-8)), ``RMSE`` :474-499,
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


def _call_spec(inputs):
    """What a caller hands ``Handler.accumulate`` for one metric -> (positional, keyword) arguments of ``metric.accumulate``: a bare
    value, or a tuple / list of positionals whose last item may be a dict of keyword arguments (``seq_len=...``)."""
    args = list(inputs) if isinstance(inputs, (tuple, list)) else [inputs]
    kwargs = args.pop() if args and isinstance(args[-1], dict) else {}
    return args, kwargs


class Handler(object):
    """The metric container of the train loop (public surface of morgana/metrics.py:50-186: ``handler[collection]``, ``add_metrics``,
    ``add_collection``, ``reset_state``, ``accumulate``, ``result``, ``results_as_json_dict``, ``results_as_str_dict``, and the live
    ``collections`` / ``metrics`` mappings).

    Every collection ('all', 'train', 'valid', 'test', or one made by ``add_collection``) is its own name -> metric mapping, handed
    out LIVE: ``handler['train'][name] = metric`` and ``handler.metrics.update(...)`` register metrics as they do in the reference,
    and the same name may hold different metric objects in different collections (``add_metrics('train', loss=a)`` then
    ``add_metrics('valid', loss=b)``).  Metrics given to the constructor join 'all', 'train' and 'valid'; ``add_metrics('all', ...)``
    joins every collection there is; whatever is added also joins 'all' (the last object added under a name is the one 'all' holds)."""

    BUILT_IN = ('all', 'train', 'valid', 'test')

    def __init__(self, **metrics):
        self.hidden = False
        self._members = {name: {} for name in self.BUILT_IN}
        self.add_metrics(('train', 'valid'), **metrics)

    # -- membership ------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _names(spec):
        return [spec] if isinstance(spec, str) or not hasattr(spec, '__iter__') else list(spec)

    def __getitem__(self, collection):
        if collection not in self._members:
            raise ValueError("No collection found by the name {}".format(collection))
        return self._members[collection]

    @property
    def metrics(self):
        return self._members['all']

    @property
    def collections(self):
        return self._members

    def add_metrics(self, collections=('all',), **metrics):
        targets = self._names(collections)
        if 'all' in targets:
            targets = list(self._members)
        for collection in dict.fromkeys(list(targets) + ['all']):
            self[collection].update(metrics)

    def add_collection(self, collection, from_collections=tuple()):
        merged = {}
        for source in self._names(from_collections):
            merged.update(self[source])
        self._members[collection] = merged

    # -- the loop's calls ------------------------------------------------------------------------------------------------------------
    def reset_state(self, collection, *args):
        for metric in self[collection].values():
            metric.reset_state()

    def accumulate(self, collection, **inputs_by_metric):
        members = self[collection]
        for name, inputs in inputs_by_metric.items():
            args, kwargs = _call_spec(inputs)
            members[name].accumulate(*args, **kwargs)          # KeyError for a name the collection does not hold, as the reference

    def _visible(self, collection):
        return [(name, metric) for name, metric in self[collection].items() if not metric.hidden]

    def result(self, collection='all', *args):
        return {name: metric.result(*args) for name, metric in self[collection].items()}

    def results_as_json_dict(self, collection='all', prefix=''):
        return {prefix + name: metric.result_as_json() for name, metric in self._visible(collection)}

    def results_as_str_dict(self, collection='all', prefix=''):
        return {prefix + name: str(metric) for name, metric in self._visible(collection)}

    def __str__(self):
        return ' | '.join('{} = {}'.format(name, text) for name, text in self.results_as_str_dict('all').items())
