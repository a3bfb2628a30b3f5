"""Host-side mirror of ``morgana.utils`` for the training hot path: same names, arguments and error behaviour, with the
per-frame compute done by the HIP kernels of libmorgana_hip.so (no torch-op or CPU fallback for the in-scope ops).

Reference: morgana/utils.py - ``sequence_mask`` :115-144, ``upsample_to_repetitions`` :175-228,
``split_to_segments`` :231-285, ``get_segment_ends`` :288-330, ``RecurrentCuDNNWrapper`` :333-393, ``SequentialWithRecurrent`` :396-418, ``ExponentialMovingAverage`` :421-456.
"""
import os

import torch
import torch.nn as nn

from . import _lib
from . import functional as F_hip
from . import ops

_LEGACY_TYPES = {
    'ByteTensor': torch.uint8, 'CharTensor': torch.int8, 'FloatTensor': torch.float32, 'DoubleTensor': torch.float64,
    'IntTensor': torch.int32, 'LongTensor': torch.int64, 'BoolTensor': torch.bool,
}


def _as_dtype(dtype):
    """Accept torch dtypes and the legacy tensor types the reference passes to ``Tensor.type`` (torch.ByteTensor)."""
    if isinstance(dtype, torch.dtype):
        return dtype
    name = getattr(dtype, '__name__', str(dtype)).split('.')[-1]
    if name in _LEGACY_TYPES:
        return _LEGACY_TYPES[name]
    raise TypeError('unsupported mask type %r' % (dtype,))


def infer_device(tensor):
    """morgana/utils.py:147-156."""
    if tensor.is_cuda:
        return torch.device('cuda:{}'.format(tensor.get_device()))
    return torch.device('cpu')


def sequence_mask(seq_len, max_len=None, dtype=torch.ByteTensor, device=None):
    """(batch_size,) lengths -> (batch_size, max_len, 1) mask.  morgana/utils.py:115-144."""
    if max_len is None:
        max_len = torch.max(seq_len).item()
    seq_len = seq_len if seq_len.dtype == torch.int64 else seq_len.long()
    return ops.sequence_mask(seq_len, int(max_len), _as_dtype(dtype))


class UpsampledSequence(object):
    """Lazy result of ``upsample_to_repetitions(..., fused=True)``: the phone-rate source plus the frame->row map.

    ``SequentialWithRecurrent`` feeds it to the first Linear through the gather-fused GEMM loader, so the
    (B, Tmax, feat) tensor never exists in HBM.  ``materialise()`` gives the ordinary dense tensor.
    """

    def __init__(self, sequence_feature, dur2d, rows, maps=None, table_bf16=None, t_cap=None, phone_rate=None):
        self.source = sequence_feature
        # the order of operations the caller chose (ops.phone_rate_choice): True = layers that commute with the repetition may run
        # on the phone rows, False = every product on the frame rows (the reference's order); the consumers read it here
        self.phone_rate = ops.phone_rate_choice(phone_rate)
        self.dur = dur2d
        # int32 (B, Tmax): b*P + phone, or -1.  None = not built yet (t_cap given): the first reader of ``rows`` launches the map
        # kernel - unless the consumer is the phone-rate loss stack, which builds the maps together with its own front
        # (functional.LinearStackMSEFn: ops.phone_front) and hands them back through ``adopt``
        self._rows = rows
        self.maps = maps                       # (seg (2, B*P) frame runs, rows with -1 -> B*P) from the same launch, or None
        self.table_bf16 = table_bf16           # the loader's bf16 copy of the source rows (data.add_bf16_table), or None
        self.t_cap = int(rows.shape[1] if rows is not None else t_cap)
        self.shape = (sequence_feature.shape[0], self.t_cap, sequence_feature.shape[2])

    @property
    def rows(self):
        if self._rows is None:
            rows, rows_mapped, seg = ops.upsample_index_maps(self.dur, self.t_cap)
            self._rows, self.maps = rows, (seg, rows_mapped.reshape(-1))
        return self._rows

    def pending(self):
        """True while no map kernel has run for this sequence."""
        return self._rows is None

    def adopt(self, rows, maps):
        self._rows, self.maps = rows, maps

    def phone_maps(self):
        """(frame runs per phone row, row map with -1 -> the first row behind the B*P phone rows) for the phone-rate paths."""
        if self.maps is None:
            n_rows = self.source.shape[0] * self.source.shape[1]
            self.maps = ops.segment_bounds(self.rows.reshape(-1), n_rows, pad_row=n_rows)
        return self.maps

    def materialise(self):
        return F_hip.UpsampleFn.apply(self.source, self.dur, self.t_cap)


def bf16_copy_of(tensor):
    """The bf16 copy a recurrence attached to its fp32 output (``RecurrentCuDNNWrapper._run_gru``), or None when there is none or the
    fp32 tensor has been written in place since (its version counter moved: the copy no longer holds its values - ADVICE round 4)."""
    entry = getattr(tensor, '_mg_bf16', None)
    if entry is None:
        return None
    copy, version = entry
    return copy if tensor._version == version else None


class PhoneTable(object):
    """Lazy frame-rate tensor whose frames repeat the rows of a TABLE: ``table`` (B*P + extra rows, F) holds one row per phone
    (the extra rows: what padding frames gather) and ``rows`` (B, T) int32 is the frame -> phone-row map of the upsample (-1 =
    padding).  Produced by ``SequentialWithRecurrent`` when Linear / Sigmoid layers directly follow an ``UpsampledSequence`` and a
    GRU wrapper comes next: those layers commute with repeating rows, so they run on the phone rows and the wrapper repeats the
    rows of ITS input projection instead (``functional.GRUFn`` with ``rows``)."""

    def __init__(self, table, rows, n_phone_rows, maps=None):
        self.table, self.rows, self.n_phone_rows, self._maps = table, rows, n_phone_rows, maps
        self.shape = (rows.shape[0], rows.shape[1], table.shape[1])
        self.ndim = 3

    def crop(self, t_out):
        return self if t_out == self.rows.shape[1] else PhoneTable(self.table, self.rows[:, :t_out].contiguous(), self.n_phone_rows)

    def maps(self):
        """(frame runs per phone row, map with -1 -> the first extra row) for the current (possibly cropped) frame axis."""
        if self._maps is None:
            self._maps = ops.segment_bounds(self.rows.reshape(-1), self.n_phone_rows, pad_row=self.n_phone_rows)
        return self._maps


PACKED_FRAMES = os.environ.get('MORGANA_PACKED_FRAMES', '1') != '0'
# The persistent GRU forward can write the bf16 copy of its output itself (mg_gru_fwd_persist_out_bf16), which saves the cast pass of the
# Linear run behind the wrapper (C4 37 us, C5 98 us).  MEASURED twice (round 4, same-box A/B): with the recurrence's computing waves
# storing their own results the 2-byte stores cost the chain more than the cast saved (C5 6.541 / 6.546 against 6.486 / 6.510 ms, C4
# 3.445 / 3.358 against 3.394 / 3.401) and it was off; since the forward kernel has dedicated storing waves (R4.9) they ride there:
# C4 3.308 / 3.301 against 3.336 / 3.334 ms, C5 6.398 / 6.423 against 6.398 / 6.419 - ON.
OUT_SHADOW = os.environ.get('MORGANA_OUT_SHADOW', '1') != '0'


# The first Linear of the 609-input models (cat(repeated phone rows, 9 frame counters), models/RNN_SPSS.py:76-81) with the labels' 600
# columns at phone rate (ops.phone_concat_layer; backward ops.segment_sum_feat).  MEASURED (round 4, same-box A/B, profiles/
# r4_phone_rate_609.txt) and OFF: the layer's output must still be written per frame in fp32 (it leaves the fused run for the recurrent
# wrapper) and its gradient read per frame twice over, so the GEMM the form removes (37 us at N = 256, M = 64,000) is replaced by
# latency-bound passes that cost as much or more - GRU-F0 3.52 against 3.44 ms, LSTM 16.15 against 16.01 ms.
CONCAT_PHONE_RATE = os.environ.get('MORGANA_CONCAT_PHONE_RATE', '0') != '0'


# The exact-fp32 modes' fused tail (functional.F0TailRowsF32Fn); 0 = the two layers and the loss as their generic launches (A/B, tests)
F0_TAIL_F32 = os.environ.get('MORGANA_F0_TAIL_F32', '1') != '0'
# Precision 'bf16x3': the README stack's phone-rate step on pair planes (functional.F0StackX3Fn); 0 = the generic row-wise path (A/B, tests)
X3_FUSED = os.environ.get('MORGANA_X3_FUSED', '1') != '0'


# Row-wise layers behind a recurrent wrapper are packed only when at least this share of the B * T rows is padding.  Measured at C5
# (64 utterances of 300-2000 frames, 41 % padding; profiles/r4_c5_packed_vs_padded.txt): the packed Linear stack + loss saves 140 us of
# GEMM / cast time and pays 257 us for the way there and back (unpack gather of the prediction 44, pack of its gradient 32, column
# sums of the padding rows' gradient 84, zero fill + scatter of the input gradient 97) - a loss of 120 us - while the recurrent
# layers' weight gradients over the valid frames only (``worthwhile``) win 109 us.  Break-even for the stack is about three
# quarters padding; the recurrent weight gradients keep the 10 % rule.
PACK_ROWS_MIN_PADDING = float(os.environ.get('MORGANA_PACK_ROWS_MIN_PADDING', '0.75'))


def set_packed_frames(enabled, rows_min_padding=None):
    """Layers behind a recurrent wrapper may work on the valid frames of a ragged batch only (``FrameLayout``); off = on all B * T
    padded rows as the reference does.  ``rows_min_padding``: share of padding rows from which the row-wise Linear / Sigmoid runs are
    packed too (default ``PACK_ROWS_MIN_PADDING``; 0.1 packs them whenever the recurrent layers are).  Results are the same either way
    (tests/test_gpu_configs.py)."""
    global PACKED_FRAMES, PACK_ROWS_MIN_PADDING
    PACKED_FRAMES = bool(enabled)
    if rows_min_padding is not None:
        PACK_ROWS_MIN_PADDING = float(rows_min_padding)


class FrameLayout(object):
    """Packed-frame maps of a ragged batch: which of the B * T rows of a zero-padded (B, T, .) tensor are real frames.

    The reference pads every batch to its longest utterance (``collate_fn``, morgana/data.py:183-193), applies nn.Linear to all
    B * T rows (morgana/utils.py:401-418) and masks the loss (losses.py:37-39); at BASELINE config C5 (300-2000 frames) 41 % of
    those rows are padding.  With a layout, ``SequentialWithRecurrent`` runs the Linear / Sigmoid layers that follow a recurrent
    wrapper on the ``total`` valid rows plus ONE representative padding row (behind the wrapper every padded frame is the same
    zero row, utils.py:383) and scatters the result back to (B, T, .) - padded predictions included, so nothing changes for the
    caller.  ``total`` comes from the host (the loader knows the lengths it collated: ``features['n_frames_total']``), so
    building the maps costs no device -> host read."""

    def __init__(self, seq_len, t, total):
        seq_len = seq_len if seq_len.dtype == torch.int64 else seq_len.long()
        self.seq_len, self.b, self.t, self.total = seq_len.contiguous(), seq_len.numel(), int(t), int(total)
        self.offsets, self.rows, self.inverse = ops.frame_layout(self.seq_len, self.t, self.total)

    @classmethod
    def for_batch(cls, features, t, seq_len_key='n_frames'):
        """Layout for ``features`` if packing is enabled and the loader recorded the host-side frame total; else None."""
        total = features.get(seq_len_key + '_total')
        if not PACKED_FRAMES or total is None or int(t) <= 0:
            return None
        return cls(features[seq_len_key], t, min(int(total), features[seq_len_key].numel() * int(t)))

    def worthwhile(self):
        """The recurrent layers' weight gradients over the valid frames only: when at least a tenth of the rows are padding."""
        return 10 * (self.total + 1) <= 9 * self.b * self.t

    def worthwhile_for_rows(self):
        """Packing a row-wise run costs a gather each way, the padding rows' column sums and a scatter: see PACK_ROWS_MIN_PADDING."""
        return self.worthwhile() and (self.total + 1) <= (1.0 - PACK_ROWS_MIN_PADDING) * self.b * self.t

    def unpack(self, packed):
        out = F_hip.UnpackRowsFn.apply(packed, self.rows, self.inverse, self.seq_len, self.b, self.t)
        return out.view(self.b, self.t, packed.shape[1])

    def frame_rows(self):
        """int32 (total,): the dense row b * t + i of every valid frame, in packed order."""
        return self.rows[:self.total]

    def state_rows(self):
        """int32 (total,): row b * (t + 1) + i of a (B, t + 1, H) state array for every valid frame (b, i) - the state the frame's
        step STARTS from (functional.state_rows restricted to the valid frames).  Made once per layout."""
        if getattr(self, '_state_rows', None) is None:
            dense = self.frame_rows()
            self._state_rows = (dense + torch.div(dense, self.t, rounding_mode='floor')).to(torch.int32).contiguous()
        return self._state_rows


class UpsampledConcat(object):
    """Lazy ``torch.cat((upsample_to_repetitions(sequence_feature, repeats), frame_feature), dim=-1)`` - the model input of
    models/RNN_SPSS.py:76-81 and models/f0_test_model.py:78-79.  ``SequentialWithRecurrent`` turns it into the first Linear's
    operand with one gather+concat pass (bf16 and padded in bf16 mode); ``materialise()`` gives the dense fp32 tensor."""

    def __init__(self, upsampled, frame_feature):
        if frame_feature.shape[:2] != tuple(upsampled.shape[:2]):
            raise RuntimeError('Sizes of tensors must match except in dimension 2. Expected %s but got %s'
                               % (tuple(upsampled.shape[:2]), tuple(frame_feature.shape[:2])))
        self.upsampled = upsampled
        self.frame_feature = frame_feature
        self.shape = tuple(upsampled.shape[:2]) + (upsampled.shape[2] + frame_feature.shape[2],)

    def operand(self, out_bf16):
        up = self.upsampled
        return ops.gather_concat(up.source.reshape(-1, up.source.shape[-1]), up.rows.reshape(-1),
                                 self.frame_feature.reshape(-1, self.frame_feature.shape[-1]), out_bf16=out_bf16)

    def materialise(self):
        return self.operand(False).view(self.shape)


def concat_frame_features(upsampled, frame_feature):
    """``torch.cat((upsampled, frame_feature), dim=-1)`` that keeps a fused upsample lazy (see ``UpsampledConcat``)."""
    if isinstance(upsampled, UpsampledSequence) and not frame_feature.requires_grad:
        return UpsampledConcat(upsampled, frame_feature)
    if isinstance(upsampled, UpsampledSequence):
        upsampled = upsampled.materialise()
    return torch.cat((upsampled, frame_feature), dim=-1)


def upsample_to_repetitions(sequence_feature, repeats, max_len=None, fused=False, table_bf16=None, phone_rate=None):
    """Copies sequence items according to a number of repetitions, as ``np.repeat`` does.  morgana/utils.py:175-228.

    sequence_feature (B, P, F) float32; repeats (B, P, 1) or (B, P) integer -> (B, max_b sum_p repeats, F).
    ``max_len`` (optional, not in the reference) supplies Tmax when the caller knows it (the padded target length),
    which removes the one device->host sync that sizing the output otherwise needs; the reference has three.
    ``phone_rate`` (not in the reference; ``fused`` only): the order of operations of the layers that consume the lazy result -
    True lets layers that commute with the repetition run once per phone row, False keeps every product on the frame rows (the
    reference's order), None = the process default (``MORGANA_PHONE_RATE``).  Same outputs either way.
    """
    if repeats.is_floating_point() or repeats.dtype == torch.bool:
        raise TypeError('upsample_to_repetitions: repeats must be an integer tensor, got %s' % repeats.dtype)
    batch_size = sequence_feature.shape[0]
    dur2d = repeats.reshape((batch_size, -1))
    dur2d = dur2d if dur2d.dtype == torch.int64 else dur2d.long()
    dur2d = dur2d.contiguous()
    if dur2d.shape[1] != sequence_feature.shape[1]:
        raise ValueError('repeats has %d items per sequence, sequence_feature has %d'
                         % (dur2d.shape[1], sequence_feature.shape[1]))
    if max_len is None:
        _, tmax = ops.upsample_lengths(dur2d)
        max_len = int(tmax.item())
    if fused and not sequence_feature.requires_grad:
        if ops.phone_rate_choice(phone_rate) and int(max_len) > 0:
            return UpsampledSequence(sequence_feature, dur2d, None, table_bf16=table_bf16, t_cap=int(max_len),
                                     phone_rate=True)                                                                # maps on first use
        _, rows = ops.upsample_index(dur2d, int(max_len))
        return UpsampledSequence(sequence_feature, dur2d, rows, table_bf16=table_bf16, phone_rate=ops.phone_rate_choice(phone_rate))
    return F_hip.UpsampleFn.apply(sequence_feature, dur2d, int(max_len))


def _segment_lens_2d(sequence_feature, segment_lens):
    if segment_lens.is_floating_point() or segment_lens.dtype == torch.bool:
        raise TypeError('segment_lens must be an integer tensor, got %s' % segment_lens.dtype)
    lens2d = segment_lens.reshape((sequence_feature.shape[0], -1))
    return (lens2d if lens2d.dtype == torch.int64 else lens2d.long()).contiguous()


def split_to_segments(sequence_feature, segment_lens):
    """Splits a sequence into zero-padded segments.  morgana/utils.py:231-285.

    sequence_feature (B, T, F) float32; segment_lens (B, S, 1) or (B, S) integer -> (B, S, max segment length, F).
    The reference builds the index array in a host loop over batch items and segments; here one kernel scans the lengths and
    writes the flat row map, a second gathers.  ``segment_lens.max().item()`` is the one sync the output shape needs (the
    reference has it too, :250)."""
    lens2d = _segment_lens_2d(sequence_feature, segment_lens)
    b, t, f = sequence_feature.shape
    max_segment_len = int(lens2d.max().item())
    split, _ = ops.segment_index(lens2d, t, max_len=max_segment_len, want_ends=False)
    out = F_hip.GatherRowsFn.apply(sequence_feature.reshape(b * t, f), split.reshape(-1))
    return out.view(b, lens2d.shape[1], max_segment_len, f)


def get_segment_ends(sequence_feature, segment_lens):
    """Feature at the last position of each segment, zeros for empty segments.  morgana/utils.py:288-330."""
    lens2d = _segment_lens_2d(sequence_feature, segment_lens)
    b, t, f = sequence_feature.shape
    _, ends = ops.segment_index(lens2d, t, want_split=False)
    out = F_hip.GatherRowsFn.apply(sequence_feature.reshape(b * t, f), ends.reshape(-1))
    return out.view(b, lens2d.shape[1], f)


class RecurrentCuDNNWrapper(nn.Module):
    """Wraps a torch recurrent layer with the sort / pack / unpack semantics of morgana/utils.py:333-393.

    ``nn.GRU`` (single layer, unidirectional, batch_first - the shape used by models/f0_test_model.py:32-39) and
    ``nn.LSTM`` (any number of layers, unidirectional, batch_first - models/RNN_SPSS.py:36-37) run on the HIP recurrences;
    the parameters stay the wrapped layer's own (``layer.weight_ih_l0`` ...), so state_dict keys match the reference.
    Other layer types (bidirectional, projections, multi-layer GRU) raise ``MorganaHipError``: there is no torch / MIOpen fallback.
    """

    def __init__(self, layer, precision=None):
        super(RecurrentCuDNNWrapper, self).__init__()
        self.layer = layer
        self.precision = precision

    def _hip_gru(self):
        layer = self.layer
        return (isinstance(layer, nn.GRU) and layer.num_layers == 1 and not layer.bidirectional and layer.batch_first
                and layer.bias and getattr(layer, 'proj_size', 0) == 0)

    def _hip_lstm(self):
        layer = self.layer
        return (isinstance(layer, nn.LSTM) and not layer.bidirectional and layer.batch_first and layer.bias and
                getattr(layer, 'proj_size', 0) == 0 and
                (layer.dropout == 0 or not layer.training or layer.num_layers == 1))      # one layer: nn.LSTM's dropout acts BETWEEN layers

    def _hip_general(self):
        """Any batch_first nn.GRU / nn.LSTM with biases and no projection - several layers, both directions: the general form of the
        reference's wrapper (morgana/utils.py:333-343 wraps any nn.RNNBase).  Runs layer by layer and direction by direction on the
        single-layer HIP recurrences (``_run_general``); the shapes the shipped models use keep their own faster paths
        (``_hip_gru`` / ``_hip_lstm``)."""
        layer = self.layer
        return (isinstance(layer, (nn.GRU, nn.LSTM)) and layer.batch_first and layer.bias and getattr(layer, 'proj_size', 0) == 0)

    def _layer_params(self, k, reverse):
        layer, sfx = self.layer, '_l%d%s' % (k, '_reverse' if reverse else '')
        return [getattr(layer, name + sfx) for name in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]

    @staticmethod
    def _reverse_rows(seq_len, b, t, device):
        """int32 (B * T,) row map that reverses every item's first ``seq_len[b]`` frames in place (-1 = zero row past the end): what
        pack_padded_sequence gives the backward direction of a bidirectional layer (morgana/utils.py:366-385).  Its own inverse."""
        steps = torch.arange(t, device=device).unsqueeze(0)
        lens = (seq_len.to(device).clamp(max=t) if seq_len is not None else torch.full((b,), t, device=device)).unsqueeze(1)
        src = lens - 1 - steps
        rows = torch.where(src >= 0, src + torch.arange(b, device=device).unsqueeze(1) * t, torch.full_like(src, -1))
        return rows.reshape(-1).to(torch.int32).contiguous()

    def _run_general(self, inputs, hidden, seq_len):
        """Multi-layer and / or bidirectional nn.GRU / nn.LSTM: layer k, direction d = one single-layer HIP recurrence (functional.GRUFn /
        LSTMFn) on the output of layer k - 1 (both directions concatenated); the backward direction runs on every item's frames
        reversed within its own length and is reversed back.  Inter-layer dropout (``layer.dropout``, training) draws its masks in the
        HIP kernel (functional.DropoutFn).  Hidden states in torch's layout: (num_layers * num_directions, B, H), index 2 k + d."""
        layer = self.layer
        is_lstm = isinstance(layer, nn.LSTM)
        precision = F_hip.recurrent_precision(self.precision or F_hip.get_precision())
        n_dir = 2 if layer.bidirectional else 1
        b, t = inputs.shape[0], inputs.shape[1]
        h0s, c0s = (hidden if is_lstm else (hidden, None)) if hidden is not None else (None, None)
        rev = self._reverse_rows(seq_len, b, t, inputs.device) if n_dir == 2 else None
        out, hns, cns = inputs.contiguous(), [], []
        for k in range(layer.num_layers):
            outs = []
            for d in range(n_dir):
                x = out
                if d == 1:
                    x = F_hip.GatherRowsFn.apply(out.reshape(b * t, -1), rev).view(b, t, -1)
                idx = k * n_dir + d
                h0 = None if h0s is None else h0s[idx:idx + 1].contiguous()
                if is_lstm:
                    c0 = None if c0s is None else c0s[idx:idx + 1].contiguous()
                    y, hn, cn = F_hip.LSTMFn.apply(precision, x.contiguous(), h0, c0, seq_len, *self._layer_params(k, d == 1))
                    cns.append(cn)
                else:
                    y, hn = F_hip.GRUFn.apply(precision, x.contiguous(), h0, seq_len, *self._layer_params(k, d == 1), None, None, None, None)
                if d == 1:
                    y = F_hip.GatherRowsFn.apply(y.reshape(b * t, -1), rev).view(b, t, -1)
                outs.append(y)
                hns.append(hn)
            out = outs[0] if n_dir == 1 else torch.cat(outs, dim=-1)
            if layer.dropout > 0 and layer.training and k + 1 < layer.num_layers:
                if layer.dropout >= 1:
                    out = out * 0.0
                else:
                    out = F_hip.DropoutFn.apply(out.contiguous(), float(layer.dropout), 7919 + k)
        hn = torch.cat(hns, 0)
        return out, ((hn, torch.cat(cns, 0)) if is_lstm else hn)

    def _lstm_params(self):
        layer = self.layer
        params = []
        for k in range(layer.num_layers):
            params += [getattr(layer, 'weight_ih_l%d' % k), getattr(layer, 'weight_hh_l%d' % k),
                       getattr(layer, 'bias_ih_l%d' % k), getattr(layer, 'bias_hh_l%d' % k)]
        return params

    def _gru_params(self):
        layer = self.layer
        return [layer.weight_ih_l0, layer.weight_hh_l0, layer.bias_ih_l0, layer.bias_hh_l0]

    def _run_lstm(self, inputs, hidden, seq_len):
        """nn.LSTM with num_layers >= 1; hidden = (h0, c0), each (num_layers, B, H), returned in the same layout
        (utils.py:374-375, 388-389).  Several layers run as one time-skewed stack (functional.LSTMStackFn)."""
        layer = self.layer
        precision = F_hip.recurrent_precision(self.precision or F_hip.get_precision())
        h0s, c0s = (None, None) if hidden is None else hidden
        if layer.num_layers == 1:
            out, hn, cn = F_hip.LSTMFn.apply(precision, inputs.contiguous(), h0s, c0s, seq_len, *self._lstm_params())
            return out, (hn, cn)
        if F_hip.lstm_stack_persistent(precision, inputs.shape[0], inputs.shape[1], layer.hidden_size, layer.num_layers):
            out, hn, cn = F_hip.LSTMStackPersistFn.apply(inputs.contiguous(), seq_len, h0s, c0s, *self._lstm_params())
            return out, (hn, cn)
        if F_hip.lstm_layerwise(precision, inputs.shape[0], inputs.shape[1], layer.hidden_size):
            # persistent recurrence: a layer is two launches, so the layers simply follow each other
            params, out, hns, cns = self._lstm_params(), inputs.contiguous(), [], []
            for k in range(layer.num_layers):
                out, hn, cn = F_hip.LSTMFn.apply(precision, out, None if h0s is None else h0s[k:k + 1],
                                                 None if c0s is None else c0s[k:k + 1], seq_len, *params[4 * k:4 * k + 4])
                hns.append(hn)
                cns.append(cn)
            return out, (torch.cat(hns, 0), torch.cat(cns, 0))
        out, hn, cn = F_hip.LSTMStackFn.apply(precision, F_hip.LSTM_STACK_LAG, inputs.contiguous(), seq_len, h0s, c0s,
                                              *self._lstm_params())
        return out, (hn, cn)

    def _run_gru(self, inputs, hidden, seq_len, layout=None):
        layer = self.layer
        precision = F_hip.recurrent_precision(self.precision or F_hip.get_precision())
        if isinstance(inputs, PhoneTable):
            seg, rows = inputs.maps()
            if layout is not None and (tuple(inputs.rows.shape) != (layout.b, layout.t) or not layout.worthwhile()):
                layout = None
            out_bf = self._out_shadow(precision, inputs.rows.shape[0], inputs.rows.shape[1], inputs.table.device)
            out, hn = F_hip.GRUFn.apply(precision, inputs.table, hidden, seq_len, layer.weight_ih_l0, layer.weight_hh_l0,
                                        layer.bias_ih_l0, layer.bias_hh_l0, rows.view(inputs.rows.shape), seg, layout, out_bf)
            if out_bf is not None:
                out._mg_bf16 = (out_bf, out._version)          # valid while nobody has written `out` in place (see bf16_copy_of)
            return out, hn
        if layout is not None and (tuple(inputs.shape[:2]) != (layout.b, layout.t) or not layout.worthwhile()):
            layout = None
        out_bf = self._out_shadow(precision, inputs.shape[0], inputs.shape[1], inputs.device)
        out, hn = F_hip.GRUFn.apply(precision, inputs, hidden, seq_len, layer.weight_ih_l0, layer.weight_hh_l0,
                                    layer.bias_ih_l0, layer.bias_hh_l0, None, None, layout, out_bf)
        if out_bf is not None:
            out._mg_bf16 = (out_bf, out._version)
        return out, hn

    def _out_shadow(self, precision, b, t, device):
        """A (B, T, H) bf16 buffer for the recurrence to fill with the bf16 copy of its output (the operand of a Linear layer that
        follows the wrapper: no cast pass over [B, T, H]) - when the caller asked for one (``want_out_shadow``: SequentialWithRecurrent
        does when a bf16 Linear run on the unpacked rows comes next; a direct call of the wrapper does not pay for a copy nobody
        reads), the persistent bf16 recurrence will run and the width needs no padding."""
        hid = self.layer.hidden_size
        if not getattr(self, 'want_out_shadow', False):
            return None
        if OUT_SHADOW and hid == ops.pad_ld(hid) and F_hip.gru_shadow_ok(precision, b, t, hid):
            return torch.empty((b, t, hid), dtype=torch.bfloat16, device=device)
        return None

    def _unsupported(self):
        layer = self.layer
        return _lib.MorganaHipError(
            'RecurrentCuDNNWrapper: %s(num_layers=%s, bidirectional=%s, batch_first=%s, bias=%s, proj_size=%s) has no HIP recurrence '
            'in libmorgana_hip.so (built: batch_first nn.GRU / nn.LSTM with biases and without projection, any number of layers, one or '
            'both directions); there is no torch / MIOpen fallback'
            % (type(layer).__name__, getattr(layer, 'num_layers', '?'), getattr(layer, 'bidirectional', '?'),
               getattr(layer, 'batch_first', '?'), getattr(layer, 'bias', '?'), getattr(layer, 'proj_size', 0)))

    def run_full_length(self, inputs, hidden=None):
        """The wrapped layer on a padded (B, T, F) batch with every item running all T steps: what calling the bare torch layer does."""
        if not (self._hip_gru() or self._hip_lstm() or self._hip_general()):
            raise self._unsupported()
        if isinstance(inputs, nn.utils.rnn.PackedSequence) or inputs.ndim != 3:
            return self.forward(inputs, hidden, None)
        run = self._run_gru if self._hip_gru() else self._run_lstm if self._hip_lstm() else self._run_general
        return run(inputs.contiguous(), hidden, None)

    def forward(self, inputs, hidden=None, seq_len=None, max_len=None, layout=None):
        """``layout`` (not in the reference): the batch's ``FrameLayout`` - a GRU layer's weight gradients then multiply the valid
        frames only (the reference gets the same from ``pack_padded_sequence``, utils.py:366-385); results unchanged.
        ``max_len`` (not in the reference): the caller's upper bound on ``max(seq_len)``, normally the padded frame axis of the
        batch.  The reference crops the output to the longest item (``pad_packed_sequence``, utils.py:383), which costs a
        device -> host read of ``seq_len`` per call; when the input's time axis already equals ``max_len`` - every batch whose longest
        utterance defines its padding, i.e. every batch ``collate_fn`` builds - the result is identical without that read, and the
        step stays capturable as a HIP graph.  ``SequentialWithRecurrent`` passes it when its own caller does."""
        if not (self._hip_gru() or self._hip_lstm() or self._hip_general()):
            raise self._unsupported()
        run = self._run_gru if self._hip_gru() else self._run_lstm if self._hip_lstm() else self._run_general
        if seq_len is None:
            if isinstance(inputs, nn.utils.rnn.PackedSequence):
                # already packed (utils.py:347-349): unpack to the padded layout the kernels work on, pack the result again
                padded, lens = nn.utils.rnn.pad_packed_sequence(inputs, batch_first=True)
                outputs, hidden = run(padded.contiguous(), hidden, lens.to(padded.device))
                return nn.utils.rnn.pack_padded_sequence(outputs, lens, batch_first=True, enforce_sorted=False), hidden
            elif inputs.ndim == 2:
                seq_dim = 1 if self.layer.batch_first else 0
                outputs, hidden = run(inputs.unsqueeze(seq_dim).contiguous(), hidden, None)
                return outputs.squeeze(seq_dim), hidden
            else:
                raise ValueError('If no seq_len is provided to RecurrentCuDNNWrapper the data must be already packed'
                                 f'or must be for one time slice only. For non-packed input got shape, {inputs.shape}')

        seq_len = seq_len if seq_len.dtype == torch.int64 else seq_len.long()
        t_in = inputs.rows.shape[1] if isinstance(inputs, PhoneTable) else inputs.shape[1]
        if max_len is not None and int(max_len) == t_in:
            t_out = t_in                                    # the longest item fills the padded axis: nothing to crop, no host read
        else:
            t_out = int(torch.max(seq_len).item())          # pad_packed_sequence crops to the longest item
        if isinstance(inputs, PhoneTable):
            if not self._hip_gru():
                raise self._unsupported()                   # (SequentialWithRecurrent hands a PhoneTable to single-layer GRU wrappers only)
            return self._run_gru(inputs.crop(t_out), hidden, seq_len.contiguous(), layout)
        if t_out != inputs.shape[1]:
            inputs = inputs[:, :t_out]
        if self._hip_gru():
            return self._run_gru(inputs.contiguous(), hidden, seq_len.contiguous(), layout)
        return run(inputs.contiguous(), hidden, seq_len.contiguous())


class SequentialWithRecurrent(nn.Sequential):
    """``nn.Sequential`` taking ``hiddens`` / ``seq_len`` for recurrent members; returns ``(output, hiddens)``.

    Reference: morgana/utils.py:396-418.  Runs of ``nn.Linear`` (+ ``nn.Sigmoid``) are executed as one fused HIP node
    (MFMA GEMM with bias+sigmoid epilogue forward; wgrad and sigmoid-grad-fused dgrad backward).  The modules stay
    ordinary ``nn.Linear`` objects, so ``state_dict`` keys (``layers.0.weight`` ...) match the reference's checkpoints.
    """

    def __init__(self, *args, precision=None):
        super(SequentialWithRecurrent, self).__init__(*args)
        self.precision = precision

    def _linear_run(self, modules, start):
        """Collect [Linear, Sigmoid?, Dropout*]+ starting at ``start``; returns (end, run) with run = [(linear, act), ...] and
        ``run.drops`` = the dropout probability behind each layer (0 = none: ``nn.Dropout`` with p == 0 or in eval mode is the
        identity - the reference's models pass dropout_prob=0., models/RNN_SPSS.py:21).  An ACTIVE dropout does not break the run
        either: the run's autograd node draws its mask in a HIP kernel behind the layer and regenerates it in its backward
        (functional.LinearStackFn, csrc/dropout.hip) - no torch kernel, no cut in the fused run.  ``run.site0`` numbers the masks."""
        run, drops, i = _Run(), [], start
        while i < len(modules) and type(modules[i]) is nn.Linear:
            act = ops.ACT_NONE
            nxt = i + 1
            if nxt < len(modules) and type(modules[nxt]) is nn.Sigmoid:
                act, nxt = ops.ACT_SIGMOID, nxt + 1
            keep = 1.0
            while nxt < len(modules) and type(modules[nxt]) is nn.Dropout:
                if modules[nxt].p != 0 and modules[nxt].training:
                    if modules[nxt].p >= 1:
                        break                                   # p == 1 zeroes everything: left to the module loop (no scale exists)
                    keep *= 1.0 - modules[nxt].p                # consecutive dropouts: one mask of the joint keep probability
                nxt += 1
            run.append((modules[i], act))
            drops.append(1.0 - keep)
            i = nxt
        run.drops, run.site0 = tuple(drops), start
        return i, run

    @staticmethod
    def _lstm_run(modules, start, hiddens, seq_len):
        """Indices of consecutive RecurrentCuDNNWrapper(single-layer nn.LSTM) modules from ``start`` (identity Dropouts between
        them skipped) that can run as one skewed stack: same hidden size, no initial hidden states, seq_len given, at most
        functional's layer limit.  Returns (index behind the run, [module indices])."""
        run, i, last_hidden = [], start, None
        while i < len(modules) and seq_len is not None and len(run) < 8:
            mod = modules[i]
            if type(mod) is nn.Dropout and (mod.p == 0 or not mod.training) and run:
                i += 1
                continue
            ok = (isinstance(mod, RecurrentCuDNNWrapper) and mod._hip_lstm() and mod.layer.num_layers == 1 and hiddens[i] is None and
                  (last_hidden is None or mod.layer.input_size == last_hidden))
            if not ok:
                break
            run.append(i)
            last_hidden = mod.layer.hidden_size
            if run and mod.layer.hidden_size != modules[run[0]].layer.hidden_size:
                run.pop()
                break
            i += 1
        end = (run[-1] + 1) if run else start
        return end, run

    @staticmethod
    def _gru_run(modules, start, hiddens, seq_len):
        """Indices of consecutive RecurrentCuDNNWrapper(single-layer nn.GRU) modules from ``start`` (identity Dropouts between them
        skipped) with one small hidden size, no initial hidden states and seq_len given - the stack of models/f0_test_model.py:31-37,
        which runs as one wavefront launch per direction.  Returns (index behind the run, [module indices])."""
        run, i, last_hidden = [], start, None
        while i < len(modules) and seq_len is not None and len(run) < 4:
            mod = modules[i]
            if type(mod) is nn.Dropout and (mod.p == 0 or not mod.training) and run:
                i += 1
                continue
            ok = (isinstance(mod, RecurrentCuDNNWrapper) and mod._hip_gru() and mod.layer.num_layers == 1 and hiddens[i] is None and
                  (last_hidden is None or (mod.layer.input_size == last_hidden and mod.layer.hidden_size == last_hidden)))
            if not ok:
                break
            run.append(i)
            last_hidden = mod.layer.hidden_size
            i += 1
        end = (run[-1] + 1) if run else start
        return end, run

    def _fused_mse_spec(self, targets, precision):
        """acts of the stack if it is [Linear, Sigmoid]* ... Linear(*,128), Sigmoid, Linear(128,32), Sigmoid, Linear(32,1) in
        bf16 mode with a 1-dimensional target (the README F0Model shape) - the case mg_f0_tail_bf16 fuses; else None."""
        modules = list(self._modules.values())
        end, run = self._linear_run(modules, 0)
        if end != len(modules) or len(run) < 3 or precision != 'bf16' or targets.shape[-1] != 1 or any(run.drops):
            return None
        dims = [lin.weight.shape for lin, _ in run]
        acts = tuple(act for _, act in run)
        ok = (tuple(dims[-1]) == (1, 32) and tuple(dims[-2]) == (32, 128) and acts[-1] == ops.ACT_NONE and
              all(a == ops.ACT_SIGMOID for a in acts[:-1]) and all(lin.bias is not None for lin, _ in run) and
              all(d[0] % 128 == 0 for d in dims[:-2]))
        return (run, acts) if ok else None

    def forward_mse(self, input, targets, seq_len=None):
        """``loss, prediction`` of ``losses.mse(self(input)[0], targets, seq_len)``.

        Same numbers as calling the container and then ``losses.mse`` (reference: README.rst:84-97), but when the stack has
        the README F0Model's tail (... -> 128 -> 32 -> 1, bf16 mode) its last two layers, the loss and their backward run
        as one kernel.  Any other stack takes the ordinary path.  ``prediction`` is detached on the fused path.
        """
        from . import losses
        precision = self.precision or F_hip.get_precision()
        fused = self._fused_mse_spec(targets, precision)
        if fused is None:
            x3 = self._fused_x3_params(input, targets, seq_len, precision)
            if x3 is not None:
                # 'bf16x3', the README stack on repeated phone rows: the whole step on [hi | lo] pair planes (functional.F0StackX3Fn)
                sl = seq_len if seq_len.dtype == torch.int64 else seq_len.long()
                return F_hip.F0StackX3Fn.apply((input, input.table_bf16), input.source.reshape(-1, input.source.shape[-1]), targets, sl, *x3)
            tail = self._phone_rate_f32_tail(input, targets, seq_len, precision)
            if tail is not None:
                # exact-fp32 modes, the README stack on repeated phone rows: the layers up to the 128-wide one on the phone rows, then
                # Sigmoid -> Linear(128, 32) -> Sigmoid -> Linear(32, 1) + the masked MSE + their backward as ONE launch (mg_f0_tail_rows_f32)
                z2, lin3, lin4 = tail
                return F_hip.F0TailRowsF32Fn.apply(z2, targets, seq_len, input, lin3.weight, lin3.bias, lin4.weight, lin4.bias)
            tail = self._frame_rate_f32_tail(input, targets, seq_len, precision)
            if tail is not None:
                z2, lin3, lin4 = tail                      # the same launch on the B * T frame rows (the reference's order of operations)
                return F_hip.F0TailRowsF32Fn.apply(z2, targets, seq_len, None, lin3.weight, lin3.bias, lin4.weight, lin4.bias)
            table = self._phone_rate_table(input, targets, seq_len, precision)
            if table is not None:
                # exact-fp32 modes, a stack of Linear / Sigmoid layers on repeated phone rows ending in ONE output column: the layers
                # run on the phone rows (as ``forward`` runs them) and the masked MSE on the per-phone predictions and target
                # statistics - the frame-rate prediction is produced for reporting only (functional.PhoneMSEFn)
                return F_hip.PhoneMSEFn.apply(table, targets, seq_len, input)
            out, _ = self.forward(input, seq_len=seq_len)
            return losses.mse(out, targets, seq_len), out
        run, acts = fused
        maps = table = phone_rate = None
        if isinstance(input, UpsampledSequence):
            phone_rate = input.phone_rate
            x2d, table = input.source.reshape(-1, input.source.shape[-1]), input.table_bf16
            if input.pending():
                rows, maps = None, input               # the stack builds the maps with its own front, or asks ``input.rows`` for them
            else:
                rows, maps = input.rows.reshape(-1), input.maps
        else:
            x2d, rows = input.reshape(-1, input.shape[-1]), None
        if seq_len is not None and seq_len.dtype != torch.int64:
            seq_len = seq_len.long()
        params = []
        for lin, _ in run:
            params += [lin.weight, lin.bias]
        return F_hip.LinearStackMSEFn.apply((acts, maps, table, phone_rate), x2d, rows, targets, seq_len, *params)

    def _fused_x3_params(self, input, targets, seq_len, precision):
        """The eight parameters of the stack when it is the README F0Model's (... -> 512 -> 128 -> 32 -> 1, sigmoids between, README.rst:65-73)
        in precision 'bf16x3' on an ``UpsampledSequence`` at phone rate with (B, T, 1) targets - what functional.F0StackX3Fn fuses; else None."""
        if precision != 'bf16x3' or not X3_FUSED or not isinstance(input, UpsampledSequence) or seq_len is None:
            return None
        found = self._readme_tail(targets, precision)
        if found is None or tuple(targets.shape[:2]) != tuple(input.shape[:2]):
            return None
        run, lin3, lin4 = found
        if len(run) != 4 or run[0][1] != ops.ACT_SIGMOID or any(lin.bias is None for lin, _ in run):
            return None
        lin1, lin2 = run[0][0], run[1][0]
        n_src = input.source.shape[0] * input.source.shape[1]
        m = input.shape[0] * input.shape[1]
        if (lin1.weight.shape[1] != input.source.shape[-1] or lin2.weight.shape[1] != lin1.weight.shape[0]
                or not ops.phone_rate_choice(input.phone_rate)
                or not ops.x3_step_ok(n_src, m, lin1.weight.shape[1], lin1.weight.shape[0], lin2.weight.shape[0])):
            return None
        params = []
        for lin, _ in run:
            params += [lin.weight, lin.bias]
        return params

    def _phone_rate_f32_tail(self, input, targets, seq_len, precision):
        """(pre-activations of the 128-wide layer on the phone rows (B * P + extra, 128), Linear(128, 32), Linear(32, 1)) when the whole
        container is Linear / Sigmoid layers ending in ``-> 128 -> Sigmoid -> 32 -> Sigmoid -> 1`` on an ``UpsampledSequence`` at
        phone rate, exact-fp32 modes, (B, T, 1) targets; else None."""
        found = self._readme_tail(targets, precision)
        if (found is None or not isinstance(input, UpsampledSequence) or seq_len is None
                or tuple(targets.shape[:2]) != tuple(input.shape[:2])):
            return None
        run, lin3, lin4 = found
        n_src = input.source.shape[0] * input.source.shape[1]
        if not ops.phone_rate_gru_ok(n_src, input.shape[0] * input.shape[1], 8, input.phone_rate):
            return None
        params = []
        for lin, _ in run[:-2]:
            params += [lin.weight, lin.bias]
        acts = tuple(act for _, act in run[:-3]) + (ops.ACT_NONE,)          # the 128-wide layer's sigmoid is taken by the tail kernel
        spec = (acts, precision, ops.PHONE_RATE_EXTRA)
        z2 = F_hip.LinearStackFn.apply(spec, input.source.reshape(-1, input.source.shape[-1]), None, *params)
        return z2, lin3, lin4

    def _readme_tail(self, targets, precision):
        """(run, Linear(128, 32), Linear(32, 1)) when the whole container is Linear / Sigmoid layers ending in ``-> 128 -> Sigmoid -> 32 ->
        Sigmoid -> 1`` without dropout, an exact-fp32 mode and (B, T, 1) targets; else None."""
        modules = list(self._modules.values())
        if (not F0_TAIL_F32 or not modules or type(modules[0]) is not nn.Linear or precision not in ('fp32', 'bf16x3')
                or targets.ndim != 3 or targets.shape[2] != 1):
            return None
        end, run = self._linear_run(modules, 0)
        if end != len(modules) or len(run) < 3 or any(run.drops):
            return None
        (lin2, act2), (lin3, act3), (lin4, act4) = run[-3], run[-2], run[-1]
        if (act2 != ops.ACT_SIGMOID or act3 != ops.ACT_SIGMOID or act4 != ops.ACT_NONE or tuple(lin3.weight.shape) != (32, 128)
                or tuple(lin4.weight.shape) != (1, 32) or lin3.bias is None or lin4.bias is None or lin2.weight.shape[0] != 128):
            return None
        return run, lin3, lin4

    def _frame_rate_f32_tail(self, input, targets, seq_len, precision):
        """As ``_phone_rate_f32_tail`` for inputs that stay at frame rate: (pre-activations of the 128-wide layer on the B * T frame
        rows, Linear(128, 32), Linear(32, 1)), or None."""
        found = self._readme_tail(targets, precision)
        if found is None or tuple(targets.shape[:2]) != tuple(input.shape[:2]):
            return None
        run, lin3, lin4 = found
        if isinstance(input, UpsampledSequence):
            n_src = input.source.shape[0] * input.source.shape[1]
            if seq_len is not None and ops.phone_rate_gru_ok(n_src, input.shape[0] * input.shape[1], 8, input.phone_rate):
                return None                                # the phone-rate form takes it
            x2d, rows = input.source.reshape(-1, input.source.shape[-1]), input.rows.reshape(-1)
        elif torch.is_tensor(input) and input.ndim == 3:
            x2d, rows = input.reshape(-1, input.shape[-1]), None
        else:
            return None
        params = []
        for lin, _ in run[:-2]:
            params += [lin.weight, lin.bias]
        acts = tuple(act for _, act in run[:-3]) + (ops.ACT_NONE,)
        spec = (acts, precision, 0, rows is not None, None)
        z2 = F_hip.LinearStackFn.apply(spec, x2d, rows, *params)
        return z2, lin3, lin4

    def _phone_rate_table(self, input, targets, seq_len, precision):
        """The (B * P + extra, 1) table of per-phone predictions when the whole container is one Linear / Sigmoid run on an
        ``UpsampledSequence`` at phone rate with a one-column output and (B, T, 1) targets; else None."""
        modules = list(self._modules.values())
        if (not isinstance(input, UpsampledSequence) or seq_len is None or not modules or type(modules[0]) is not nn.Linear
                or precision not in ('fp32', 'bf16x3') or targets.ndim != 3 or targets.shape[2] != 1
                or tuple(targets.shape[:2]) != tuple(input.shape[:2])):
            return None
        end, run = self._linear_run(modules, 0)
        n_src = input.source.shape[0] * input.source.shape[1]
        if (end != len(modules) or any(run.drops) or run[-1][0].weight.shape[0] != 1
                or not ops.phone_rate_gru_ok(n_src, input.shape[0] * input.shape[1], 8, input.phone_rate)):
            return None
        params = []
        for lin, _ in run:
            params += [lin.weight, lin.bias]
        spec = (tuple(act for _, act in run), precision, ops.PHONE_RATE_EXTRA)
        return F_hip.LinearStackFn.apply(spec, input.source.reshape(-1, input.source.shape[-1]), None, *params)

    def forward(self, input, hiddens=None, seq_len=None, max_len=None, layout=None):
        """``max_len`` (not in the reference) is handed to the recurrent wrappers: see ``RecurrentCuDNNWrapper.forward``.
        ``layout`` (not in the reference): a ``FrameLayout`` of the batch - Linear / Sigmoid runs that follow a recurrent wrapper
        then work on the valid frame rows only; the returned tensor is the same (B, T, .) either way."""
        modules = list(self._modules.values())
        if hiddens is None:
            hiddens = [None] * len(modules)
        precision = self.precision or F_hip.get_precision()
        rec_precision = F_hip.recurrent_precision(precision)      # 'bf16x3' splits the row-wise layers' operands; cells run exact fp32
        zero_padded = False          # input is (B, T, .) with exactly zero rows past seq_len (it came out of a recurrent wrapper)

        i = 0
        while i < len(modules):
            module = modules[i]
            if type(module) is nn.Linear:
                end, run = self._linear_run(modules, i)
                if (zero_padded and layout is not None and torch.is_tensor(input) and input.ndim == 3 and
                        tuple(input.shape[:2]) == (layout.b, layout.t) and layout.worthwhile_for_rows()):
                    # packed frames: the run's GEMMs take sum_b T_b + 1 rows instead of B * T; the first layer's loader (fp32) or one
                    # gather + cast pass (bf16) packs, ``layout.unpack`` restores (B, T, .) with the representative row on the padding
                    params = []
                    for lin, _ in run:
                        params += [lin.weight, lin.bias]
                    spec = (tuple(act for _, act in run), precision, 0, False, run.drop_spec())
                    packed = F_hip.LinearStackFn.apply(spec, input.reshape(-1, input.shape[-1]), layout.rows, *params)
                    input = layout.unpack(packed)
                    zero_padded = False
                    i = end
                    continue
                zero_padded = False
                nxt = modules[end] if end < len(modules) else None
                n_src = input.source.shape[0] * input.source.shape[1] if isinstance(input, UpsampledSequence) else 0
                if (isinstance(input, UpsampledSequence) and isinstance(nxt, RecurrentCuDNNWrapper) and nxt._hip_gru()
                        and seq_len is not None and not any(run.drops)         # a frame's dropout mask is its own: no phone-rate form
                        and ops.phone_rate_gru_ok(n_src, input.rows.numel(), run[-1][0].weight.shape[0], input.phone_rate)):
                    # Linear / Sigmoid commute with the row repetition of the upsample: run them on the phone rows (+ zero rows for
                    # the padding frames) and hand the GRU wrapper a table + row map; it repeats the rows of its own input projection
                    params = []
                    for lin, _ in run:
                        params += [lin.weight, lin.bias]
                    spec = (tuple(act for _, act in run), precision, ops.PHONE_RATE_EXTRA)
                    table = F_hip.LinearStackFn.apply(spec, input.source.reshape(-1, input.source.shape[-1]), None, *params)
                    input = PhoneTable(table, input.rows, n_src, maps=input.phone_maps())
                    i = end
                    continue
                if (isinstance(input, UpsampledSequence) and end == len(modules) and not any(run.drops)
                        and ops.phone_rate_gru_ok(n_src, input.rows.numel(), 8, input.phone_rate)):
                    # the stack ends in this run and sees nothing but the repeated phone rows: every layer commutes with the
                    # repetition, so the run works on the phone rows (+ zero rows for padding frames) and its OUTPUT is repeated
                    params = []
                    for lin, _ in run:
                        params += [lin.weight, lin.bias]
                    spec = (tuple(act for _, act in run), precision, ops.PHONE_RATE_EXTRA)
                    table = F_hip.LinearStackFn.apply(spec, input.source.reshape(-1, input.source.shape[-1]), None, *params)
                    seg, rows_mapped = input.phone_maps()
                    out = F_hip.RepeatTableRowsFn.apply(table, rows_mapped, seg, n_src)
                    input = out.view(input.shape[0], input.shape[1], out.shape[-1])
                    i = end
                    continue
                if isinstance(input, UpsampledSequence):
                    lead, x2d, rows = input.shape[:2], input.source.reshape(-1, input.source.shape[-1]), \
                        input.rows.reshape(-1)
                elif (isinstance(input, UpsampledConcat) and precision == 'bf16' and CONCAT_PHONE_RATE
                      and input.frame_feature.shape[-1] <= 16
                      and ops.phone_rate_gru_ok(input.upsampled.source.shape[0] * input.upsampled.source.shape[1],
                                                input.upsampled.rows.numel(), 8, input.upsampled.phone_rate)):
                    # cat(repeated phone rows, frame counters) (models/RNN_SPSS.py:76-81): W = [W_lab | W_cnt], the lab part of the
                    # first layer commutes with the repetition and runs once per phone, the counters' part per frame
                    up = input.upsampled
                    seg, rows_mapped = up.phone_maps()
                    feat = input.frame_feature.reshape(-1, input.frame_feature.shape[-1])
                    params = []
                    for lin, _ in run:
                        params += [lin.weight, lin.bias]
                    spec = (tuple(act for _, act in run), precision, ops.PHONE_RATE_EXTRA, False, run.drop_spec(), None,
                            (rows_mapped.reshape(-1), seg, feat.contiguous()))
                    out = F_hip.LinearStackFn.apply(spec, up.source.reshape(-1, up.source.shape[-1]), None, *params)
                    input = out.view(*input.shape[:2], out.shape[-1])
                    i = end
                    continue
                elif isinstance(input, UpsampledConcat):
                    lead, x2d, rows = input.shape[:2], input.operand(precision == 'bf16'), None
                else:
                    lead, x2d, rows = input.shape[:-1], input.reshape(-1, input.shape[-1]), None
                params = []
                for lin, _ in run:
                    params += [lin.weight, lin.bias]
                # rows here is the frame map of upsample_to_repetitions: runs of equal indices (a hint for the layer-1 loader)
                shadow = bf16_copy_of(input) if (torch.is_tensor(input) and precision == 'bf16') else None
                spec = (tuple(act for _, act in run), precision, 0, rows is not None, run.drop_spec(),
                        shadow.view(-1, shadow.shape[-1]) if shadow is not None else None)
                out = F_hip.LinearStackFn.apply(spec, x2d, rows, *params)
                input = out.view(*lead, out.shape[-1])
                i = end
                continue

            if isinstance(input, (UpsampledSequence, UpsampledConcat)):
                input = input.materialise()

            if isinstance(module, RecurrentCuDNNWrapper) and module._hip_gru() and torch.is_tensor(input) and input.ndim == 3:
                end, run = self._gru_run(modules, i, hiddens, seq_len)
                t_in = input.shape[1]
                if (len(run) > 1 and max_len is not None and int(max_len) == t_in and
                        F_hip.gru_stack_small(input.shape[0], t_in, modules[run[0]].layer.hidden_size, len(run))):
                    # consecutive small GRU wrappers (models/f0_test_model.py:31-37): one wavefront launch per direction
                    params = []
                    for k in run:
                        params += modules[k]._gru_params()
                    sl = seq_len if seq_len.dtype == torch.int64 else seq_len.long()
                    input, hn = F_hip.GRUStackSmallFn.apply(rec_precision, input.contiguous(), sl.contiguous(), None, *params)
                    for pos, k in enumerate(run):
                        hiddens[k] = hn[pos:pos + 1]
                    zero_padded = True
                    i = end
                    continue

            if isinstance(module, RecurrentCuDNNWrapper):
                end, run = self._lstm_run(modules, i, hiddens, seq_len)
                hid = modules[run[0]].layer.hidden_size if run else 0
                if len(run) > 1 and F_hip.lstm_stack_persistent(rec_precision, input.shape[0], input.shape[1], hid, len(run)):
                    # consecutive single-layer LSTM wrappers (models/RNN_SPSS.py:36-37): the whole stack's forward is one
                    # persistent launch (a wavefront over layers and time)
                    params = []
                    for k in run:
                        params += modules[k]._lstm_params()
                    input, hn, cn = F_hip.LSTMStackPersistFn.apply(input.contiguous(), seq_len, None, None, *params)
                    for pos, k in enumerate(run):
                        hiddens[k] = (hn[pos:pos + 1], cn[pos:pos + 1])
                    zero_padded = True
                    i = end
                    continue
                if len(run) > 1 and not F_hip.lstm_layerwise(rec_precision, input.shape[0], input.shape[1], hid):
                    # consecutive single-layer LSTM wrappers (models/RNN_SPSS.py:36-37): one time-skewed stack (per-step
                    # launches); with the persistent recurrence each wrapper is two launches and runs on its own below
                    params = []
                    for k in run:
                        params += modules[k]._lstm_params()
                    input, hn, cn = F_hip.LSTMStackFn.apply(rec_precision, F_hip.LSTM_STACK_LAG, input.contiguous(), seq_len, None,
                                                            None, *params)
                    for pos, k in enumerate(run):
                        hiddens[k] = (hn[pos:pos + 1], cn[pos:pos + 1])
                    zero_padded = True
                    i = end
                    continue

            if isinstance(module, RecurrentCuDNNWrapper):
                # the recurrence writes the bf16 copy of its output only when a bf16 Linear run on the unpacked rows reads it next
                nxt = i + 1
                while nxt < len(modules) and type(modules[nxt]) is nn.Dropout and (modules[nxt].p == 0 or not modules[nxt].training):
                    nxt += 1
                module.want_out_shadow = (precision == 'bf16' and nxt < len(modules) and type(modules[nxt]) is nn.Linear and
                                          not (layout is not None and layout.worthwhile_for_rows()))
                try:
                    input, hiddens[i] = module(input, hiddens[i], seq_len, max_len=max_len, layout=layout)
                finally:
                    module.want_out_shadow = False
                zero_padded = seq_len is not None
            elif isinstance(module, nn.RNNBase):
                # a bare recurrent layer in the container (utils.py:412-413): every item runs the full padded length
                input, hiddens[i] = RecurrentCuDNNWrapper(module, precision=rec_precision).run_full_length(input, hiddens[i])
                zero_padded = False
            elif type(module) is nn.Sigmoid:
                shape = input.shape
                input = _SigmoidFn.apply(input.reshape(-1)).view(shape)
                zero_padded = False
            elif type(module) is nn.Dropout:
                identity = module.p == 0 or not module.training
                if not identity and module.p < 1:
                    # active dropout outside a Linear run (between the recurrent wrappers of models/f0_test_model.py:31-43): the HIP
                    # mask kernel, regenerated in the backward - never torch's nn.Dropout kernel
                    input = F_hip.DropoutFn.apply(input, float(module.p), i)
                elif not identity:
                    # p == 1: everything dropped.  torch's own result (at::dropout multiplies by a zero scalar: 0 for finite values,
                    # NaN for inf / NaN, -0 for negative ones - checked against torch 2.10 in tests/test_host_logic.py; ADVICE round 4
                    # expected exact zeros, which is not what the reference's nn.Dropout returns)
                    input = input * 0.0
                zero_padded = zero_padded                             # zero rows stay zero under any mask
            else:
                input = module(input)
                zero_padded = False
            i += 1

        if isinstance(input, (UpsampledSequence, UpsampledConcat)):
            input = input.materialise()
        return input, hiddens


class _Run(list):
    """[(linear, act), ...] of ``SequentialWithRecurrent._linear_run`` plus ``drops`` (dropout probability behind each layer) and
    ``site0`` (index of the run's first module: numbers the dropout masks)."""
    drops, site0 = (), 0

    def drop_spec(self):
        return (self.drops, self.site0) if any(self.drops) else None


class _SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.sigmoid(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, grad):
        (y,) = ctx.saved_tensors
        return ops.sigmoid_grad(grad.contiguous(), y)


class ExponentialMovingAverage(object):
    """EMA helper applying updates to a separate EMA model: ``shadow = decay*shadow + (1-decay)*x``.

    Reference: morgana/utils.py:421-456.  ``shadow[name]`` aliases the EMA model's ``param.data`` as in the reference.
    """

    def __init__(self, model, decay):
        self.model = model
        self.decay = decay
        self.shadow = {}
        self._params = {}
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                self.shadow[name] = param.data
                self._params[name] = param

    def _update_param(self, name, x):
        assert name in self.shadow
        ops.ema_update(self.shadow[name], x.contiguous(), self.decay)
        # the kernel wrote the EMA model's weight through ``param.data`` (no version bump on the Parameter): say so, or a bf16 operand
        # copy of it made by an earlier forward pass of the EMA model (ops.param_shadows) would be taken for current
        ops.mark_updated(self._params[name])

    def update_params(self, other_model):
        assert other_model is not self.model
        for name, param in other_model.named_parameters():
            if name in self.shadow:
                self._update_param(name, param.data)
