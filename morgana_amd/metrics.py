"""Minimal mirror of ``morgana.metrics`` for the train loop: ``Handler`` with the ``Mean`` loss metric.

Reference: morgana/metrics.py - ``Handler.accumulate`` :133-153, ``Mean`` :359-397 (sum / (count + 1e-8)).
The loss scalar stays on the device; ``result`` syncs only when asked (the reference formats it every batch).
"""
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
