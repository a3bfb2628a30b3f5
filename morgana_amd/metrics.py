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
    def __init__(self, hidden=False):
        self.hidden = hidden
        self.reset_state()

    def reset_state(self):
        self.sum = 0.
        self.count = 0.

    def accumulate(self, tensor, seq_len=None):
        if seq_len is not None:
            raise NotImplementedError('Mean.accumulate with seq_len is outside the hot path')
        self.sum = self.sum + torch.sum(tensor.detach())
        self.count += tensor.numel()

    def result(self, *args):
        return self.sum / (self.count + 1e-8)


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
    """Tracks metrics per mode ('train' / 'valid' / 'test'), morgana/metrics.py:52-186 (collection logic only)."""

    def __init__(self, **metrics):
        self._factories = {name: type(metric) for name, metric in metrics.items()}
        self.collections = {}

    def _collection(self, mode):
        if mode not in self.collections:
            self.collections[mode] = {name: cls() for name, cls in self._factories.items()}
        return self.collections[mode]

    def reset_state(self, mode):
        for metric in self._collection(mode).values():
            metric.reset_state()

    def accumulate(self, mode, **kwargs):
        for name, value in kwargs.items():
            metric = self._collection(mode)[name]
            if isinstance(value, (tuple, list)):
                metric.accumulate(*value)
            else:
                metric.accumulate(value)

    def results_as_json_dict(self, mode):
        return {name: float(metric.result()) for name, metric in self._collection(mode).items()}
