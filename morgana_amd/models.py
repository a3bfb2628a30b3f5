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
import os

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
                 fused_upsample=True, fused_loss=True, phone_rate=None):
        super(F0Model, self).__init__()
        self.fused_loss = fused_loss
        self.phone_rate = phone_rate          # order of operations of THIS model (base_models.BaseModel.phone_rate)
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
                                                               max_len=max_len, fused=self.fused_upsample, phone_rate=self.phone_rate)
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
        precision = self.layers.precision or F_hip.get_precision()
        # 'bf16': the table's bf16 copy; 'bf16x3': its [hi | lo] pair planes (data.add_bf16_table) - made once per batch by the loader
        return ('normalised_lab',) if precision == 'bf16' else ('normalised_lab:x3',) if precision == 'bf16x3' else ()

    def step_input_keys(self, features):
        # With the loader's operand table of this precision in the batch the FUSED steps read the table, never the fp32 feature - but
        # only they do: any other path of the stack (a batch too small for the fused 'bf16x3' step, a stack that is not the README's)
        # casts or splits the fp32 feature inside the step.  Decided by the very predicates the forward pass uses.
        from . import ops
        precision = self.layers.precision or utils.F_hip.get_precision()
        suffix = {'bf16': data.BF16_TABLE_SUFFIX, 'bf16x3': data.X3_TABLE_SUFFIX}.get(precision)
        lab, target = features.get('normalised_lab'), features.get('normalised_' + self.target_name)
        table = features.get('normalised_lab' + suffix) if suffix is not None else None
        if table is None or lab is None or target is None or not self.fused_loss or not self.fused_upsample:
            return None
        rows, k = lab.shape[0] * lab.shape[1] + ops.PHONE_RATE_EXTRA, lab.shape[2]
        if precision == 'bf16':
            reads_table = self.layers._fused_mse_spec(target, precision) is not None and tuple(table.shape) == (rows, ops.pad_ld(k))
        else:
            reads_table = False
            if ops.phone_rate_choice(self.phone_rate) and tuple(table.shape) == (rows, 2 * ops.pad_ld(k)):
                x = utils.upsample_to_repetitions(lab, features['dur'], max_len=target.shape[1], fused=True, table_bf16=table, phone_rate=True)
                reads_table = (isinstance(x, utils.UpsampledSequence) and
                               self.layers._fused_x3_params(x, target, features['n_frames'], precision) is not None)
        if not reads_table:
            return None
        return [k_ for k_, v in features.items() if isinstance(v, torch.Tensor) and k_ not in ('normalised_lab', 'lab')]

    def forward(self, features):
        """``predict`` + ``loss`` (base_models.py:279-285) with the stack's tail and the loss fused when a target is at hand
        (``SequentialWithRecurrent.forward_mse``); identical outputs otherwise."""
        target = features.get('normalised_' + self.target_name)
        if target is None or not self.fused_loss:
            return super(F0Model, self).forward(features)
        # the loader's operand table of the phone rows, if the batch carries the one this precision reads (bf16_table_features)
        x3 = (self.layers.precision or utils.F_hip.get_precision()) == 'bf16x3'
        table = features.get('normalised_lab' + (data.X3_TABLE_SUFFIX if x3 else data.BF16_TABLE_SUFFIX))
        x = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'], max_len=target.shape[1],
                                          fused=self.fused_upsample, table_bf16=table, phone_rate=self.phone_rate)
        loss, pred_norm = self.layers.forward_mse(x, target, seq_len=features['n_frames'])
        outputs = {'pred_norm_' + self.target_name: pred_norm}
        if self.target_name in self.normalisers:
            outputs['pred_' + self.target_name] = self.normalisers[self.target_name].denormalise(pred_norm.detach())
        return loss, outputs


class RNNSPSS(BaseSPSS):
    def __init__(self, input_dim=600, hidden_dim=512, post_dim=256, output_dim=80, target_name='mcep', precision=None,
                 fused_upsample=True, phone_rate=None):
        super(RNNSPSS, self).__init__()
        self.phone_rate = phone_rate
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
                                                               max_len=max_len, fused=self.fused_upsample, phone_rate=self.phone_rate)
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


# ---------------------------------------------------------------------------------------------------------------------------------
# The reference's two shipped models (models/RNN_SPSS.py, models/f0_test_model.py) are instances of ONE scheme: labels + counters
# -> a layer stack -> a prediction that is split into output STREAMS, each with a masked loss, optionally a delta-feature
# trajectory (MLPG) and a streaming metric.  A stream is a row of a table; the generic model below reads the table.
# ---------------------------------------------------------------------------------------------------------------------------------
class Stream(object):
    """One output stream.  ``name``: feature name ('lf0'); ``dim``: columns of the prediction; ``loss``: 'mse' against
    ``normalised_<name>_deltas`` (a delta stream: static + delta + delta-delta, denormalised and turned into a trajectory by MLPG)
    or 'sigmoid_bce' against ``<name>`` (a probability stream).  ``metric`` = (registered name, factory, kind): kind 'trajectory'
    feeds (target, trajectory, n_frames), 'voiced_trajectory' adds a voicing mask (the predicted probability stream ``voicing`` >
    0.5, or the feature of that name when the model predicts none), 'accuracy' feeds the hit rate of a probability stream."""

    def __init__(self, name, dim, loss='mse', metric=None, voicing='vuv'):
        self.name, self.dim, self.loss, self.metric, self.voicing = name, dim, loss, metric, voicing

    @property
    def is_delta(self):
        return self.loss == 'mse'

    @property
    def output_key(self):
        return 'normalised_%s_deltas' % self.name if self.is_delta else self.name


# The delta streams' MLPG launches on streams of their own (StreamModel._with_trajectories); 0 = one after the other on the current stream
TRAJECTORY_STREAMS = os.environ.get('MORGANA_TRAJECTORY_STREAMS', '1') != '0'
_traj_streams = {}


def _trajectory_streams(device, n):
    pool = _traj_streams.setdefault(device.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device))
    return pool[:n]


class StreamModel(BaseSPSS):
    """``layers`` (a ``SequentialWithRecurrent`` whose last Linear is as wide as the streams together) + a stream table ->
    ``predict`` / ``loss`` / ``forward`` as the shipped models define them: input = upsampled labels concatenated with the frame
    counters, per-stream outputs under the reference's keys, loss = mean of the streams' masked losses, trajectories and metrics on
    the device whenever the normalisers carry delta parameters (i.e. under ``ExperimentBuilder``; ``generate=False`` turns both off).
    ``fused_loss``: the split, the sigmoid and all masked losses as one pass over the prediction (``losses.multi_stream``)."""

    def __init__(self, layers, streams, fused_upsample=True, fused_loss=False, generate=True):
        super(StreamModel, self).__init__()
        self.layers = layers
        self.streams = tuple(streams)
        self.fused_upsample, self.fused_loss, self.generate = fused_upsample, fused_loss, generate
        registered = {st.metric[0]: st.metric[1]() for st in self.streams if st.metric is not None}
        if registered:
            self.metrics.add_metrics('all', **registered)

    def normaliser_sources(self):
        sources = {'dur': data.MeanVarianceNormaliser('dur'), 'lab': data.MinMaxNormaliser('lab'),
                   'counters': data.MinMaxNormaliser('counters')}
        for st in self.streams:
            if st.is_delta:
                sources[st.name] = data.MeanVarianceNormaliser(st.name, use_deltas=True)
        return sources

    # -- pieces ----------------------------------------------------------------------------------------------------------------------
    def _run_layers(self, features):
        norm_counters = features['normalised_counters']
        norm_lab_at_frame_rate = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'],
                                                               max_len=norm_counters.shape[1], fused=self.fused_upsample,
                                                               phone_rate=self.phone_rate)
        model_inputs = utils.concat_frame_features(norm_lab_at_frame_rate, norm_counters)
        prediction, _ = self.layers(model_inputs, seq_len=features['n_frames'], max_len=norm_counters.shape[1])
        return prediction

    def _split(self, prediction, probabilities=None):
        """Per-stream outputs under the reference's keys; ``probabilities``: the sigmoid of the probability streams when a fused
        loss pass has computed it already."""
        parts = torch.split(prediction, [st.dim for st in self.streams], dim=-1) if len(self.streams) > 1 else (prediction,)
        outputs = {}
        for st, part in zip(self.streams, parts):
            if st.is_delta:
                outputs[st.output_key] = part
            else:
                outputs[st.output_key] = torch.sigmoid(part) if probabilities is None else probabilities
        return outputs

    def _generating(self):
        return self.generate and all(_has_delta_params(self.normalisers, st.name) for st in self.streams if st.is_delta)

    def _trajectory(self, name, pred_norm_deltas, seq_len=None):
        """Denormalised deltas -> most probable static trajectory under the global delta variances, padding 100
        (models/RNN_SPSS.py:107-118, models/f0_test_model.py:83-89), without leaving the device."""
        normaliser = self.normalisers[name]
        pred_deltas = normaliser.denormalise(pred_norm_deltas.detach(), deltas=True)
        return viz.synthesis.MLPG(means=pred_deltas, variances=normaliser.delta_params_torch['std_dev'] ** 2, padding_size=100,
                                  seq_len=seq_len)

    _prepare_output = _trajectory      # the reference's name for it

    def _with_trajectories(self, outputs, n_frames):
        if self._generating():
            delta = [st for st in self.streams if st.is_delta]
            first = outputs[delta[0].output_key] if delta else None
            if len(delta) > 1 and TRAJECTORY_STREAMS and torch.is_tensor(first) and first.is_cuda:
                # The streams' trajectories are independent, and each MLPG launch is ONE dependent chain per (utterance, dimension) over
                # the frame axis on a few waves (lf0: 64 systems = one wave; mcep: 3,840): launched one after the other their chain
                # latencies add (the shipped acoustic model: 0.15 + 0.35 + 0.15 ms per step), side by side on streams of their own
                # they overlap.  Fork behind the current stream, join before anything reads a trajectory.
                main = torch.cuda.current_stream(first.device)
                pool = _trajectory_streams(first.device, len(delta))
                for st, side in zip(delta, pool):
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        outputs[st.name] = self._trajectory(st.name, outputs[st.output_key], n_frames)
                for st, side in zip(delta, pool):
                    main.wait_stream(side)
                    outputs[st.name].record_stream(main)
            else:
                for st in delta:
                    outputs[st.name] = self._trajectory(st.name, outputs[st.output_key], n_frames)
        return outputs

    def _accumulate_metrics(self, features, outputs):
        if not self._generating():
            return
        n_frames = features['n_frames']
        predicted = {st.name for st in self.streams if not st.is_delta}
        calls = {}
        for st in self.streams:
            if st.metric is None:
                continue
            metric_name, _, kind = st.metric
            if kind == 'accuracy':
                calls[metric_name] = ((features[st.name] == (outputs[st.name] > 0.5)).type(torch.float), n_frames)
            elif kind == 'voiced_trajectory':
                voiced = (outputs[st.voicing] > 0.5) if st.voicing in predicted else features[st.voicing]
                calls[metric_name] = (features[st.name], outputs[st.name], voiced, n_frames)
            else:
                calls[metric_name] = (features[st.name], outputs[st.name], n_frames)
        self.metrics.accumulate(self.mode, **calls)

    def _target(self, features, st):
        return features[st.output_key]

    # -- the plugin surface ----------------------------------------------------------------------------------------------------------
    def predict(self, features):
        return self._with_trajectories(self._split(self._run_layers(features)), features['n_frames'])

    def loss(self, features, output_features):
        n_frames = features['n_frames']
        self._accumulate_metrics(features, output_features)
        total = 0.
        for st in self.streams:                           # delta streams first, then the probability streams: the reference's order
            if st.is_delta:
                total = total + losses.mse(output_features[st.output_key], self._target(features, st), n_frames)
        for st in self.streams:
            if not st.is_delta:
                total = total + losses.bce(output_features[st.output_key].type(torch.float), self._target(features, st).type(torch.float),
                                           n_frames)
        return total / float(len(self.streams)) if len(self.streams) > 1 else total

    def forward(self, features):
        if not self.fused_loss:
            return super(StreamModel, self).forward(features)
        prediction = self._run_layers(features)
        targets = [self._target(features, st) for st in self.streams]
        kinds = [st.loss for st in self.streams]
        loss, probabilities = losses.multi_stream(prediction, targets, kinds, features['n_frames'], want_prob=True)
        outputs = self._with_trajectories(self._split(prediction.detach(), probabilities), features['n_frames'])
        self._accumulate_metrics(features, outputs)
        return loss, outputs


def _lstm_stack(input_dim, hidden_dim, post_dim, output_dim, num_layers, dropout_prob, precision):
    """models/RNN_SPSS.py:32-42 (the container order fixes the reference's state_dict keys)."""
    return utils.SequentialWithRecurrent(
        nn.Linear(input_dim, hidden_dim), nn.Sigmoid(), nn.Dropout(p=dropout_prob),
        *[utils.RecurrentCuDNNWrapper(nn.LSTM(hidden_dim, hidden_dim, dropout=dropout_prob, batch_first=True), precision=precision)
          for _ in range(num_layers)],
        nn.Linear(hidden_dim, post_dim), nn.Sigmoid(), nn.Dropout(p=dropout_prob),
        nn.Linear(post_dim, output_dim),
        precision=precision)


def _gru_f0_stack(input_dim, output_dim, dropout_prob, precision):
    """models/f0_test_model.py:28-45."""
    def gru(n_in):
        return utils.RecurrentCuDNNWrapper(nn.GRU(n_in, 64, batch_first=True), precision=precision)
    return utils.SequentialWithRecurrent(
        nn.Linear(input_dim, 256), nn.Sigmoid(), nn.Dropout(p=dropout_prob),
        gru(256), nn.Dropout(p=dropout_prob), gru(64), nn.Dropout(p=dropout_prob), gru(64), nn.Dropout(p=dropout_prob),
        nn.Linear(64, 64), nn.Sigmoid(), nn.Dropout(p=dropout_prob),
        nn.Linear(64, output_dim),
        precision=precision)


class LSTMAcousticModel(StreamModel):
    """The reference's shipped acoustic model (models/RNN_SPSS.py:20-139) as a stream table: lf0 / mcep / bap delta streams with
    masked MSE, a vuv probability stream with masked BCE, loss = their mean; LF0 RMSE in Hz over the frames the model calls voiced,
    V/UV accuracy, mel-cepstral and band-aperiodicity distortion (:44-48, :120-129).  Same constructor arguments and state_dict keys
    (``layers.0.weight`` ... ``layers.{3+k}.layer.weight_ih_l0`` ...)."""

    STREAMS = ('lf0', 'vuv', 'mcep', 'bap')

    def __init__(self, input_dim=600 + 9, output_dims=None, dropout_prob=0., num_layers=8, hidden_dim=512, post_dim=256,
                 precision=None, fused_upsample=True, fused_loss=True, generate=True):
        if output_dims is None:
            output_dims = {'lf0': 1 * 3, 'vuv': 1, 'mcep': 60 * 3, 'bap': 5 * 3}
        self.input_dim, self.output_dims, self.dropout_prob, self.num_layers = input_dim, output_dims, dropout_prob, num_layers
        table = {'lf0': Stream('lf0', output_dims['lf0'], 'mse', ('LF0_RMSE_Hz', metrics.LF0Distortion, 'voiced_trajectory')),
                 'vuv': Stream('vuv', output_dims['vuv'], 'sigmoid_bce', ('VUV_accuracy', metrics.Mean, 'accuracy')),
                 'mcep': Stream('mcep', output_dims['mcep'], 'mse', ('MCEP_distortion', metrics.MelCepDistortion, 'trajectory')),
                 'bap': Stream('bap', output_dims['bap'], 'mse', ('BAP_distortion', metrics.Distortion, 'trajectory'))}
        layers = _lstm_stack(input_dim, hidden_dim, post_dim, sum(output_dims.values()), num_layers, dropout_prob, precision)
        super(LSTMAcousticModel, self).__init__(layers, [table[name] for name in self.STREAMS], fused_upsample=fused_upsample,
                                                fused_loss=fused_loss, generate=generate)


class GRUF0Model(StreamModel):
    """The reference's shipped F0 model (models/f0_test_model.py:21-107) as a one-row stream table: the lf0 delta stream with masked
    MSE and the LF0 RMSE in Hz over the frames the DATA calls voiced (``features['vuv']``, :101-103).  Same constructor arguments and
    state_dict keys (``layers.0.weight``, ``layers.3.layer.weight_ih_l0`` ...)."""

    def __init__(self, dropout_prob=0., input_dim=600 + 9, output_dim=1 * 3, precision=None, fused_upsample=True, generate=True):
        self.input_dim, self.output_dim = input_dim, output_dim
        layers = _gru_f0_stack(input_dim, output_dim, dropout_prob, precision)
        streams = [Stream('lf0', output_dim, 'mse', ('LF0_RMSE_Hz', metrics.LF0Distortion, 'voiced_trajectory'))]
        super(GRUF0Model, self).__init__(layers, streams, fused_upsample=fused_upsample, fused_loss=False, generate=generate)
