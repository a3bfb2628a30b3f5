"""The BASELINE model definitions, written against the plugin surface exactly as a user of the reference would.

* ``F0Model``  - README stack Linear/Sigmoid 600-512-128-32-1 (README.rst:53-97; configs C1-C3).
* ``RNNSPSS``  - Linear-512 / sigmoid / GRU-512 / Linear-256 / sigmoid / Linear-out, the layer layout of
                 models/RNN_SPSS.py:32-42 with the GRU cell of models/f0_test_model.py:32-39 (configs C4-C5).
* ``GRUF0Model`` - the reference's shipped F0 model, models/f0_test_model.py:21-107: 609-dim input, Linear-256 / sigmoid /
                 GRU-64 x 3 / Linear-64 / sigmoid / Linear-3 (lf0 + deltas).
* ``LSTMAcousticModel`` - the reference's shipped acoustic model, models/RNN_SPSS.py:20-139: 609-dim input (labels +
                 counters), Linear-512 / sigmoid / 8 x LSTM-512 / Linear-256 / sigmoid / Linear-199, four output streams.
``SequentialWithRecurrent`` returns ``(output, hiddens)`` (utils.py:418), so all of them unpack it.
"""
import torch
import torch.nn as nn

from . import data
from . import losses
from . import metrics
from . import utils
from . import viz
from .base_models import BaseSPSS


def _has_delta_params(normalisers, name):
    norm = normalisers.get(name)
    return norm is not None and getattr(norm, 'delta_params_torch', None) is not None


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

    def bf16_table_features(self):
        from . import functional as F_hip
        return ('normalised_lab',) if (self.layers.precision or F_hip.get_precision()) == 'bf16' else ()

    def forward(self, features):
        """``predict`` + ``loss`` (base_models.py:279-285) with the stack's tail and the loss fused when a target is at hand
        (``SequentialWithRecurrent.forward_mse``); identical outputs otherwise."""
        target = features.get('normalised_' + self.target_name)
        if target is None or not self.fused_loss:
            return super(F0Model, self).forward(features)
        x = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'], max_len=target.shape[1],
                                          fused=self.fused_upsample, table_bf16=features.get('normalised_lab' + data.BF16_TABLE_SUFFIX))
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
        # max_len: the padded frame axis is the longest utterance (collate_fn, data.py:183-193), so the GRU wrapper need not read
        # seq_len back to crop its output (utils.py:383) - no host sync in the step, which makes it capturable as a HIP graph
        layout = utils.FrameLayout.for_batch(features, max_len) if max_len is not None else None
        pred_norm, _ = self.layers(norm_lab_at_frame_rate, seq_len=features['n_frames'], max_len=max_len, layout=layout)
        outputs = {'pred_norm_' + self.target_name: pred_norm}
        if self.target_name in self.normalisers:
            outputs['pred_' + self.target_name] = self.normalisers[self.target_name].denormalise(pred_norm.detach())
        return outputs

    def loss(self, features, output_features):
        return losses.mse(output_features['pred_norm_' + self.target_name],
                          features['normalised_' + self.target_name], features['n_frames'])


class LSTMAcousticModel(BaseSPSS):
    """models/RNN_SPSS.py:20-139 against this package: same constructor arguments, layer container (so the reference's
    state_dict keys ``layers.0.weight`` ... ``layers.{3+k}.layer.weight_ih_l0`` ... load unchanged), ``predict`` outputs and
    ``loss``, and the metrics registered at :44-48 and accumulated in ``loss`` (:120-129).  ``_prepare_output`` (:107-118:
    denormalise the delta streams, MLPG against the global delta variances, padding 100) runs on the device
    (``viz.synthesis.MLPG``, csrc/mlpg.hip) instead of through numpy on the host; it and the metrics are active whenever the
    normalisers carry delta parameters, i.e. under ``ExperimentBuilder`` as in the reference (``generate=False`` turns both off)."""

    STREAMS = ('lf0', 'vuv', 'mcep', 'bap')

    def __init__(self, input_dim=600 + 9, output_dims=None, dropout_prob=0., num_layers=8, hidden_dim=512, post_dim=256,
                 precision=None, fused_upsample=True, fused_loss=True, generate=True):
        if output_dims is None:
            output_dims = {'lf0': 1 * 3, 'vuv': 1, 'mcep': 60 * 3, 'bap': 5 * 3}
        super(LSTMAcousticModel, self).__init__()
        self.generate = generate
        self.input_dim = input_dim
        self.output_dims = output_dims
        self.dropout_prob = dropout_prob
        self.num_layers = num_layers
        self.fused_upsample = fused_upsample
        self.fused_loss = fused_loss
        self.layers = utils.SequentialWithRecurrent(
            nn.Linear(self.input_dim, hidden_dim),
            nn.Sigmoid(),
            nn.Dropout(p=self.dropout_prob),
            *[utils.RecurrentCuDNNWrapper(nn.LSTM(hidden_dim, hidden_dim, dropout=self.dropout_prob, batch_first=True),
                                          precision=precision)
              for _ in range(self.num_layers)],
            nn.Linear(hidden_dim, post_dim),
            nn.Sigmoid(),
            nn.Dropout(p=self.dropout_prob),
            nn.Linear(post_dim, sum(self.output_dims.values())),
            precision=precision)
        self.metrics.add_metrics('all',                                              # models/RNN_SPSS.py:44-48
                                 LF0_RMSE_Hz=metrics.LF0Distortion(),
                                 VUV_accuracy=metrics.Mean(),
                                 MCEP_distortion=metrics.MelCepDistortion(),
                                 BAP_distortion=metrics.Distortion())

    def normaliser_sources(self):
        return {
            'dur': data.MeanVarianceNormaliser('dur'),
            'lab': data.MinMaxNormaliser('lab'),
            'counters': data.MinMaxNormaliser('counters'),
            'lf0': data.MeanVarianceNormaliser('lf0', use_deltas=True),
            'mcep': data.MeanVarianceNormaliser('mcep', use_deltas=True),
            'bap': data.MeanVarianceNormaliser('bap', use_deltas=True),
        }

    def _run_layers(self, features):
        norm_counters = features['normalised_counters']
        norm_lab_at_frame_rate = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'],
                                                               max_len=norm_counters.shape[1], fused=self.fused_upsample)
        model_inputs = utils.concat_frame_features(norm_lab_at_frame_rate, norm_counters)
        pred_norm_deltas, _ = self.layers(model_inputs, seq_len=features['n_frames'], max_len=norm_counters.shape[1])
        return pred_norm_deltas

    def _split(self, pred_norm_deltas, pred_vuv=None):
        output_dims = [self.output_dims[n] for n in self.STREAMS]
        lf0, vuv, mcep, bap = torch.split(pred_norm_deltas, output_dims, dim=-1)
        return {
            'normalised_lf0_deltas': lf0,
            'normalised_mcep_deltas': mcep,
            'normalised_bap_deltas': bap,
            'vuv': torch.sigmoid(vuv) if pred_vuv is None else pred_vuv,
        }

    def _generating(self):
        return self.generate and all(_has_delta_params(self.normalisers, n) for n in ('lf0', 'mcep', 'bap'))

    def _prepare_output(self, name, pred_norm_deltas, seq_len=None):
        """models/RNN_SPSS.py:107-118, without leaving the device."""
        pred_deltas = self.normalisers[name].denormalise(pred_norm_deltas.detach(), deltas=True)
        return viz.synthesis.MLPG(means=pred_deltas, variances=self.normalisers[name].delta_params_torch['std_dev'] ** 2,
                                  padding_size=100, seq_len=seq_len)

    def _with_trajectories(self, outputs, n_frames):
        if self._generating():
            for name in ('lf0', 'mcep', 'bap'):                                      # :88-93
                outputs[name] = self._prepare_output(name, outputs['normalised_%s_deltas' % name], n_frames)
        return outputs

    def _accumulate_metrics(self, features, output_features):
        if not self._generating():
            return
        n_frames = features['n_frames']
        vuv = output_features['vuv'] > 0.5                                           # :121-129
        self.metrics.accumulate(
            self.mode,
            LF0_RMSE_Hz=(features['lf0'], output_features['lf0'], vuv, n_frames),
            VUV_accuracy=((features['vuv'] == vuv).type(torch.float), n_frames),
            MCEP_distortion=(features['mcep'], output_features['mcep'], n_frames),
            BAP_distortion=(features['bap'], output_features['bap'], n_frames))

    def predict(self, features):
        return self._with_trajectories(self._split(self._run_layers(features)), features['n_frames'])

    def loss(self, features, output_features):
        n_frames = features['n_frames']
        self._accumulate_metrics(features, output_features)
        loss = 0.
        loss += losses.mse(output_features['normalised_lf0_deltas'], features['normalised_lf0_deltas'], n_frames)
        loss += losses.mse(output_features['normalised_mcep_deltas'], features['normalised_mcep_deltas'], n_frames)
        loss += losses.mse(output_features['normalised_bap_deltas'], features['normalised_bap_deltas'], n_frames)
        loss += losses.bce(output_features['vuv'].type(torch.float), features['vuv'].type(torch.float), n_frames)
        return loss / 4.

    def forward(self, features):
        """``predict`` + ``loss`` (base_models.py:279-285); with ``fused_loss`` the split, the sigmoid and the four masked
        losses run as one pass over the prediction (``losses.multi_stream``), same numbers."""
        if not self.fused_loss:
            return super(LSTMAcousticModel, self).forward(features)
        pred_norm_deltas = self._run_layers(features)
        targets = [features['vuv'] if n == 'vuv' else features['normalised_%s_deltas' % n] for n in self.STREAMS]
        kinds = ['sigmoid_bce' if n == 'vuv' else 'mse' for n in self.STREAMS]
        loss, pred_vuv = losses.multi_stream(pred_norm_deltas, targets, kinds, features['n_frames'], want_prob=True)
        outputs = self._with_trajectories(self._split(pred_norm_deltas.detach(), pred_vuv), features['n_frames'])
        self._accumulate_metrics(features, outputs)
        return loss, outputs


class GRUF0Model(BaseSPSS):
    """models/f0_test_model.py:21-107 against this package: same constructor arguments, the same layer container (state_dict
    keys ``layers.0.weight``, ``layers.3.layer.weight_ih_l0`` ... load unchanged), ``predict`` / ``loss`` and the LF0 metric
    (:47-48, :101-103).  MLPG (:86-89) runs on the device (``viz.synthesis.MLPG``, csrc/mlpg.hip); it and the metric are active
    whenever the 'lf0' normaliser carries delta parameters, i.e. under ``ExperimentBuilder`` (``generate=False`` turns both off)."""

    def __init__(self, dropout_prob=0., input_dim=600 + 9, output_dim=1 * 3, precision=None, fused_upsample=True, generate=True):
        super(GRUF0Model, self).__init__()
        self.generate = generate
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.fused_upsample = fused_upsample
        self.layers = utils.SequentialWithRecurrent(
            nn.Linear(self.input_dim, 256),
            nn.Sigmoid(),
            nn.Dropout(p=dropout_prob),
            utils.RecurrentCuDNNWrapper(nn.GRU(256, 64, batch_first=True), precision=precision),
            nn.Dropout(p=dropout_prob),
            utils.RecurrentCuDNNWrapper(nn.GRU(64, 64, batch_first=True), precision=precision),
            nn.Dropout(p=dropout_prob),
            utils.RecurrentCuDNNWrapper(nn.GRU(64, 64, batch_first=True), precision=precision),
            nn.Dropout(p=dropout_prob),
            nn.Linear(64, 64),
            nn.Sigmoid(),
            nn.Dropout(p=dropout_prob),
            nn.Linear(64, self.output_dim),
            precision=precision)
        self.metrics.add_metrics('all', LF0_RMSE_Hz=metrics.LF0Distortion())         # models/f0_test_model.py:47-48

    def normaliser_sources(self):
        return {
            'dur': data.MeanVarianceNormaliser('dur'),
            'lab': data.MinMaxNormaliser('lab'),
            'counters': data.MinMaxNormaliser('counters'),
            'lf0': data.MeanVarianceNormaliser('lf0', use_deltas=True),
        }

    def predict(self, features):
        norm_counters = features['normalised_counters']
        norm_lab_at_frame_rate = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'],
                                                               max_len=norm_counters.shape[1], fused=self.fused_upsample)
        model_inputs = utils.concat_frame_features(norm_lab_at_frame_rate, norm_counters)
        n_frames = features['n_frames']
        pred_norm_lf0_deltas, _ = self.layers(model_inputs, seq_len=n_frames, max_len=norm_counters.shape[1])
        outputs = {'normalised_lf0_deltas': pred_norm_lf0_deltas}
        if self.generate and _has_delta_params(self.normalisers, 'lf0'):
            # MLPG to select the most probable trajectory given the delta and delta-delta features (:83-89)
            pred_lf0_deltas = self.normalisers['lf0'].denormalise(pred_norm_lf0_deltas.detach(), deltas=True)
            global_variance = self.normalisers['lf0'].delta_params_torch['std_dev'] ** 2
            outputs['lf0'] = viz.synthesis.MLPG(pred_lf0_deltas, global_variance, padding_size=100, seq_len=n_frames)
        return outputs

    def loss(self, features, output_features):
        seq_len = features['n_frames']
        loss = losses.mse(output_features['normalised_lf0_deltas'], features['normalised_lf0_deltas'], seq_len)
        if 'lf0' in output_features:
            self.metrics.accumulate(self.mode,                                       # :101-103
                                    LF0_RMSE_Hz=(features['lf0'], output_features['lf0'], features['vuv'], seq_len))
        return loss
