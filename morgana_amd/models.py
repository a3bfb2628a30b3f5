"""The BASELINE model definitions, written against the plugin surface exactly as a user of the reference would.

* ``F0Model``  - README stack Linear/Sigmoid 600-512-128-32-1 (README.rst:53-97; configs C1-C3).
* ``RNNSPSS``  - Linear-512 / sigmoid / GRU-512 / Linear-256 / sigmoid / Linear-out, the layer layout of
                 models/RNN_SPSS.py:32-42 with the GRU cell of models/f0_test_model.py:32-39 (configs C4-C5).
``SequentialWithRecurrent`` returns ``(output, hiddens)`` (utils.py:418), so both unpack it.
"""
import torch.nn as nn

from . import data
from . import losses
from . import utils
from .base_models import BaseSPSS


class F0Model(BaseSPSS):
    def __init__(self, input_dim=600, hidden_dims=(512, 128, 32), output_dim=1, target_name='lf0', precision=None,
                 fused_upsample=True, fused_loss=True):
        super(F0Model, self).__init__()
        self.fused_loss = fused_loss
        dims = (input_dim,) + tuple(hidden_dims) + (output_dim,)
        mods = []
        for i in range(len(dims) - 1):
            mods.append(nn.Linear(dims[i], dims[i + 1]))
            if i < len(dims) - 2:
                mods.append(nn.Sigmoid())
        self.layers = utils.SequentialWithRecurrent(*mods, precision=precision)
        self.target_name = target_name
        self.fused_upsample = fused_upsample

    def normaliser_sources(self):
        return {
            'lab': data.MinMaxNormaliser('lab'),
            self.target_name: data.MeanVarianceNormaliser(self.target_name),
        }

    def predict(self, features):
        target = features.get('normalised_' + self.target_name)
        max_len = target.shape[1] if target is not None else None
        norm_lab_at_frame_rate = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'],
                                                               max_len=max_len, fused=self.fused_upsample)
        pred_norm, _ = self.layers(norm_lab_at_frame_rate, seq_len=features['n_frames'])
        outputs = {'pred_norm_' + self.target_name: pred_norm}
        if self.target_name in self.normalisers:
            outputs['pred_' + self.target_name] = self.normalisers[self.target_name].denormalise(pred_norm.detach())
        return outputs

    def loss(self, features, output_features):
        return losses.mse(output_features['pred_norm_' + self.target_name],
                          features['normalised_' + self.target_name], features['n_frames'])

    def forward(self, features):
        """``predict`` + ``loss`` (base_models.py:279-285) with the stack's tail and the loss fused when a target is at hand
        (``SequentialWithRecurrent.forward_mse``); identical outputs otherwise."""
        target = features.get('normalised_' + self.target_name)
        if target is None or not self.fused_loss:
            return super(F0Model, self).forward(features)
        x = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'], max_len=target.shape[1],
                                          fused=self.fused_upsample)
        loss, pred_norm = self.layers.forward_mse(x, target, seq_len=features['n_frames'])
        outputs = {'pred_norm_' + self.target_name: pred_norm}
        if self.target_name in self.normalisers:
            outputs['pred_' + self.target_name] = self.normalisers[self.target_name].denormalise(pred_norm.detach())
        return loss, outputs


class RNNSPSS(BaseSPSS):
    def __init__(self, input_dim=600, hidden_dim=512, post_dim=256, output_dim=80, target_name='mcep', precision=None,
                 fused_upsample=True):
        super(RNNSPSS, self).__init__()
        self.layers = utils.SequentialWithRecurrent(
            nn.Linear(input_dim, hidden_dim),
            nn.Sigmoid(),
            utils.RecurrentCuDNNWrapper(nn.GRU(hidden_dim, hidden_dim, batch_first=True), precision=precision),
            nn.Linear(hidden_dim, post_dim),
            nn.Sigmoid(),
            nn.Linear(post_dim, output_dim),
            precision=precision)
        self.target_name = target_name
        self.fused_upsample = fused_upsample

    def normaliser_sources(self):
        return {
            'lab': data.MinMaxNormaliser('lab'),
            self.target_name: data.MeanVarianceNormaliser(self.target_name),
        }

    def predict(self, features):
        target = features.get('normalised_' + self.target_name)
        max_len = target.shape[1] if target is not None else None
        norm_lab_at_frame_rate = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'],
                                                               max_len=max_len, fused=self.fused_upsample)
        pred_norm, _ = self.layers(norm_lab_at_frame_rate, seq_len=features['n_frames'])
        outputs = {'pred_norm_' + self.target_name: pred_norm}
        if self.target_name in self.normalisers:
            outputs['pred_' + self.target_name] = self.normalisers[self.target_name].denormalise(pred_norm.detach())
        return outputs

    def loss(self, features, output_features):
        return losses.mse(output_features['pred_norm_' + self.target_name],
                          features['normalised_' + self.target_name], features['n_frames'])
