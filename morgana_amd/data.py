"""The normaliser and loader side of ``morgana.data`` in front of the HIP kernels.

Reference behaviour: morgana/data.py - ``normalise_mvn`` / ``denormalise_mvn`` :533-538, ``normalise_minmax`` /
``denormalise_minmax`` :579-590, the normaliser classes :252-386, :541-616 (here: one affine-map class driven by a table of
kinds, ``FeatureNormaliser``), ``collate_fn`` :159-224, ``FilesDataset`` :60-157, ``ToDeviceWrapper`` :648-663.

As in the reference, every function works on NumPy arrays (loader side, host arithmetic: data.py:119-127) and on
torch tensors (model side, README.rst:87); device tensors go through the fused HIP elementwise kernel.
"""
import json
import os

import numpy as np
import torch

from . import ops


class _NormFn(torch.autograd.Function):
    """out = kernel(x; p0, p1).  d out / d x is the per-column scale (1/(std+eps), std, 1/scale or scale)."""

    @staticmethod
    def forward(ctx, x, p0, p1, kind):
        ctx.kind = kind
        ctx.save_for_backward(p0, p1)
        return ops.normalise(x, p0, p1, kind)

    @staticmethod
    def backward(ctx, grad):
        p0, p1 = ctx.saved_tensors
        kind = ctx.kind
        if kind == ops.NORM_MVN:
            scale = 1.0 / (p1 + 1e-8)
        elif kind == ops.DENORM_MVN:
            scale = p1
        else:
            scale = p1 - p0
            scale = torch.where(scale.abs() <= 1e-8, torch.ones_like(scale), scale)
            if kind == ops.NORM_MINMAX:
                scale = 1.0 / scale
        return grad * scale, None, None, None


def _device_norm(feature, p0, p1, kind):
    if not feature.is_cuda:
        raise RuntimeError('morgana_amd normalisers take NumPy arrays (host, loader side) or device tensors; '
                           'got a CPU torch tensor (there is no CPU fallback for the device path)')
    p0 = p0.to(device=feature.device, dtype=torch.float32).reshape(-1)
    p1 = p1.to(device=feature.device, dtype=torch.float32).reshape(-1)
    return _NormFn.apply(feature, p0, p1, kind)


# One table describes every normalisation this package knows: a normaliser is an AFFINE map per feature column,
#     normalise(x) = (x - offset) / divisor        denormalise(y) = y * multiplier + offset,
# and a kind says how (offset, divisor, multiplier) follow from its two stored parameter vectors, which JSON file holds them and
# which op code of the device kernel (mg_normalise_f32) computes the same map.  Reference arithmetic: data.py:533-538 (mvn),
# :579-590 (minmax; a zero range divides by one).
def _minmax_range(mmin, mmax):
    span = mmax - mmin
    return np.where(np.abs(span) <= 1e-8, np.ones_like(span), span)


_KINDS = {
    'mvn': {'params': ('mean', 'std_dev'), 'file': '{name}_mvn.json', 'forward': ops.NORM_MVN, 'inverse': ops.DENORM_MVN,
            'affine': lambda mean, std_dev: (mean, std_dev + 1e-8, std_dev)},
    'minmax': {'params': ('mmin', 'mmax'), 'file': '{name}_minmax.json', 'forward': ops.NORM_MINMAX, 'inverse': ops.DENORM_MINMAX,
               'affine': lambda mmin, mmax: (mmin, _minmax_range(mmin, mmax), _minmax_range(mmin, mmax))},
}


def _apply_kind(kind, feature, p0, p1, inverse):
    """The map of ``kind`` (or its inverse) over the last axis of ``feature``: NumPy on the host (the loader side, data.py:119-127),
    the HIP elementwise kernel for device tensors (the model side, README.rst:87)."""
    spec = _KINDS[kind]
    if isinstance(feature, np.ndarray):
        offset, divisor, multiplier = spec['affine'](p0, p1)
        if inverse:
            return (feature * multiplier[..., None, :]) + offset[..., None, :]
        return (feature - offset[..., None, :]) / divisor[..., None, :]
    return _device_norm(feature, p0, p1, spec['inverse' if inverse else 'forward'])


def normalise_mvn(feature, mean, std_dev):
    return _apply_kind('mvn', feature, mean, std_dev, inverse=False)


def denormalise_mvn(feature, mean, std_dev):
    return _apply_kind('mvn', feature, mean, std_dev, inverse=True)


def normalise_minmax(feature, mmin, mmax):
    return _apply_kind('minmax', feature, mmin, mmax, inverse=False)


def denormalise_minmax(feature, mmin, mmax):
    return _apply_kind('minmax', feature, mmin, mmax, inverse=True)


class _ParamGroup(object):
    """The two parameter vectors of one normaliser (its statics, or its deltas): float32 NumPy arrays for the host path and the same
    values as torch tensors on ``device`` for the device path."""

    def __init__(self, kind, values, device='cpu'):
        names = _KINDS[kind]['params']
        missing = [n for n in names if n not in values]
        if missing:
            raise KeyError('normaliser parameters %s missing (have %s)' % (missing, sorted(values)))
        self.host = {n: np.asarray(values[n], dtype=np.float32) for n in values}
        self.torch = {n: torch.tensor(v).to(device) for n, v in self.host.items()}

    @classmethod
    def from_json(cls, kind, path, device='cpu'):
        with open(path, 'r') as f:
            return cls(kind, json.load(f), device=device)


class FeatureNormaliser(object):
    """A named feature's normaliser: ``kind`` (a row of ``_KINDS``) + up to two parameter groups (the feature itself; its deltas when
    ``use_deltas``).  Public surface of the reference's normalisers (data.py:252-386): ``normalise`` / ``denormalise`` on NumPy arrays
    and tensors, ``fetch_params``, ``load_params`` from ``{name}_<kind>.json``, the ``params`` / ``params_torch`` / ``delta_params`` /
    ``delta_params_torch`` dictionaries; ``set_params`` installs values directly (synthetic runs have no JSON files)."""

    kind = None

    def __init__(self, name, use_deltas=False):
        if self.kind not in _KINDS:
            raise NotImplementedError('FeatureNormaliser is abstract: use MeanVarianceNormaliser or MinMaxNormaliser')
        self.name = name
        self.use_deltas = use_deltas
        self._groups = {}                                 # False -> the feature's parameters, True -> its deltas'

    # -- parameters ------------------------------------------------------------------------------------------------------------------
    def _view(self, deltas, side):
        group = self._groups.get(bool(deltas))
        return None if group is None else getattr(group, side)

    params = property(lambda self: self._view(False, 'host'))
    params_torch = property(lambda self: self._view(False, 'torch'))
    delta_params = property(lambda self: self._view(True, 'host'))
    delta_params_torch = property(lambda self: self._view(True, 'torch'))

    def fetch_params(self, data_type=np.ndarray, deltas=False):
        return self._view(deltas, 'torch' if data_type == torch.Tensor else 'host')

    def set_params(self, params, delta_params=None, device='cpu'):
        self._groups[False] = _ParamGroup(self.kind, params, device=device)
        if self.use_deltas and delta_params is not None:
            self._groups[True] = _ParamGroup(self.kind, delta_params, device=device)
        return self

    def load_params(self, data_dir, data_root='.', device='cpu'):
        pattern = _KINDS[self.kind]['file']
        for deltas in ((False, True) if self.use_deltas else (False,)):
            file_name = pattern.format(name=self.name + ('_deltas' if deltas else ''))
            self._groups[deltas] = _ParamGroup.from_json(self.kind, os.path.join(data_root, data_dir, file_name), device=device)

    # -- the map ---------------------------------------------------------------------------------------------------------------------
    def _map(self, feature, deltas, inverse):
        values = self.fetch_params(type(feature), deltas=deltas)
        if values is None:
            raise RuntimeError('normaliser %r has no %sparameters: call load_params or set_params first' % (
                self.name, 'delta ' if deltas else ''))
        p0, p1 = (values[n] for n in _KINDS[self.kind]['params'])
        return _apply_kind(self.kind, feature, p0, p1, inverse)

    def normalise(self, feature, deltas=False):
        return self._map(feature, deltas, inverse=False)

    def denormalise(self, feature, deltas=False):
        return self._map(feature, deltas, inverse=True)


class MeanVarianceNormaliser(FeatureNormaliser):
    """Zero mean / unit variance; ``mean`` / ``std_dev`` from ``{name}_mvn.json`` (data.py:541-564)."""
    kind = 'mvn'


class MinMaxNormaliser(FeatureNormaliser):
    """Range [0, 1]; ``mmin`` / ``mmax`` from ``{name}_minmax.json`` (data.py:593-616)."""
    kind = 'minmax'


_FeatureNormaliser = FeatureNormaliser      # the reference's name for the base class


class Normalisers(dict):
    """name -> normaliser, every member's parameters loaded from ``data_root/normalisation_dir`` on construction (data.py:225-247)."""

    def __init__(self, normaliser_sources, normalisation_dir, data_root='.', device='cpu'):
        super(Normalisers, self).__init__(normaliser_sources)
        self.normalisation_dir = os.path.join(data_root, normalisation_dir)
        self.device = device
        for normaliser in self.values():
            normaliser.load_params(self.normalisation_dir, device=device)


FRAME_COUNT_KEY = 'n_frames'


def _host_total(value):
    """Sum of a host-side length vector as a python int, or None if the lengths are already on a device (reading them back would
    cost a synchronisation)."""
    if isinstance(value, np.ndarray):
        return int(value.sum())
    if isinstance(value, torch.Tensor) and not value.is_cuda:
        return int(value.sum().item())
    return None


BF16_TABLE_SUFFIX = '__bf16_table'
X3_TABLE_SUFFIX = '__x3_table'         # the [hi | lo] pair planes of precision 'bf16x3' (ops.split_pair); asked for as 'name:x3'


def add_bf16_table(features, key='normalised_lab', extra_rows=None):
    """Loader-side half of bf16 mode: ``features[key + '__bf16_table']`` = the (B*P + extra_rows, pad_ld(F)) bf16 copy of the phone-level
    feature ``features[key]`` (B, P, F) that the first Linear's loader reads (zero padded columns, ``extra_rows`` zero rows = what
    padding frames gather in the phone-rate step).  Made ONCE when the batch is loaded - it is data preparation, like the
    float32 cast of data.py:127 - instead of one cast kernel over the 49 MB table in every training step.  The fp32 feature stays
    in the dict (the reference's key, and the operand of fp32 mode); models fall back to casting it themselves when this entry is
    missing."""
    key, _, kind = key.partition(':')
    x = features[key]
    if extra_rows is None:
        extra_rows = ops.PHONE_RATE_EXTRA
    if kind == 'x3':
        # precision 'bf16x3': the table as a [hi | lo] pair of bf16 planes (x = hi + lo to 16 significant bits) - the operand of the
        # fused step's first layer and of its weight gradient (functional.F0StackX3Fn)
        features[key + X3_TABLE_SUFFIX] = ops.split_pair(x.reshape(-1, x.shape[-1]), extra_rows=extra_rows)
    else:
        features[key + BF16_TABLE_SUFFIX] = ops.cast_pad_bf16(x.reshape(-1, x.shape[-1]), extra_rows=extra_rows)
    return features


def to_device(features, device, bf16_tables=()):
    """``ToDeviceWrapper.to_device`` over a feature dict (data.py:648-663); numpy arrays are uploaded too.

    ``bf16_tables``: names of phone-level features whose bf16 operand table the batch should carry (``add_bf16_table``; what a
    bf16-precision model's ``bf16_table_features()`` names) - the loader-side half of bf16 mode.

    One addition the reference's dict does not have: ``n_frames_total``, the python int sum of ``n_frames`` taken while the lengths
    are still on the host - it sizes the packed-frame layout of ragged batches (``utils.FrameLayout``) without a device -> host read.
    Models work without it (the build may carry a precomputed packed layout next to the reference's keys)."""
    out = {}
    for key, value in features.items():
        if isinstance(value, np.ndarray):
            value = torch.from_numpy(value)
        out[key] = value.to(device) if isinstance(value, torch.Tensor) else value
    total = _host_total(features.get(FRAME_COUNT_KEY))
    if total is not None and FRAME_COUNT_KEY + '_total' not in out:
        out[FRAME_COUNT_KEY + '_total'] = total
    for key in bf16_tables or ():
        name, _, kind = key.partition(':')
        if name in out and name + (X3_TABLE_SUFFIX if kind == 'x3' else BF16_TABLE_SUFFIX) not in out:
            add_bf16_table(out, key)
    return out


_TO_TORCH_DTYPE = {np.dtype('float16'): torch.float16, np.dtype('float32'): torch.float32,
                   np.dtype('float64'): torch.float64, np.dtype('int8'): torch.int8, np.dtype('int16'): torch.int16,
                   np.dtype('int32'): torch.int32, np.dtype('int64'): torch.int64, np.dtype('bool'): torch.bool,
                   int: torch.int64, float: torch.float32, bool: torch.bool}


def collate_fn(batch):
    """List of per-utterance feature dicts -> batched dict (reference: ``FilesDataset.collate_fn``, data.py:159-224).

    Sequence features (ndarray, ndim > 1) are zero padded to ``(B, max_len, feat_dim)``; 1-d arrays and python scalars
    become ``(B, ...)`` tensors (ints -> int64); anything else (names) stays a list.
    """
    batch_size = len(batch)
    out = {}
    for key in batch[0].keys():
        items = [item[key] for item in batch]
        first = items[0]
        if isinstance(first, np.ndarray) and first.ndim > 1:
            max_len = max(len(x) for x in items)
            dtype = _TO_TORCH_DTYPE[first.dtype]
            # padded in NumPy (a slice assignment per utterance; a torch.tensor + indexed copy per utterance was 1 ms of a 256-utterance batch)
            batched = np.zeros((batch_size, max_len) + tuple(first.shape[1:]), dtype=first.dtype)
            for i, x in enumerate(items):
                batched[i, :x.shape[0]] = x
            out[key] = torch.from_numpy(batched).to(dtype)
        elif isinstance(first, np.ndarray) and first.dtype in _TO_TORCH_DTYPE:
            out[key] = torch.tensor(np.stack(items), dtype=_TO_TORCH_DTYPE[first.dtype])
        elif not isinstance(first, np.ndarray) and type(first) in _TO_TORCH_DTYPE:
            out[key] = torch.tensor(items, dtype=_TO_TORCH_DTYPE[type(first)])
        else:
            out[key] = items
    return out


def load_utterance(features, normalisers):
    """One utterance as ``FilesDataset.__getitem__`` yields it (data.py:106-154): every feature that has a normaliser also
    gets its ``normalised_`` twin, computed on the host in NumPy and cast to float32 (data.py:119-127)."""
    out = dict(features)
    for name, normaliser in normalisers.items():
        if name in features:
            out['normalised_' + name] = normaliser.normalise(features[name]).astype(np.float32)
    return out


class _Staging(object):
    """Pinned host staging for the loader's packed features: per (device, feature) TWO buffers used in turn - the copy of the batch
    before last has certainly been issued when a buffer comes round again, and its event says when it has finished - grown
    geometrically, never returned.  (``tensor.pin_memory()`` per batch allocates and page-locks 49 MB every time: with the fresh
    ``np.concatenate`` result in front of it, 80 ms of a C2 batch's 82.)"""

    def __init__(self):
        self._slots = {}
        self._turn = {}

    def take(self, device, key, n_bytes):
        turn = self._turn.get((device, key), 0)
        self._turn[(device, key)] = 1 - turn
        slot = self._slots.get((device, key, turn))
        if slot is not None and slot[1] is not None:
            slot[1].synchronize()                         # the H2D copy that last read this buffer
        if slot is None or slot[0].numel() < n_bytes:
            size = max(int(n_bytes * 1.25), 1 << 16)
            slot = [torch.empty(size, dtype=torch.uint8).pin_memory(), None]
            self._slots[(device, key, turn)] = slot
        return slot

    def clear(self):
        self._slots.clear()
        self._turn.clear()


_STAGING = _Staging()
HOST_PACK_THREADS = max(1, min(8, (os.cpu_count() or 2) // 2))


def _pack_pinned(items, device, key):
    """``np.concatenate(items)`` of 2-D float32 arrays into a pinned staging buffer (mg_host_pack: threaded memcpy), its asynchronous
    copy to ``device`` and the row offsets (int64, ``len(items) + 1``) beside it.  Returns (packed device tensor, offsets device tensor)."""
    import ctypes
    from . import _lib
    width = items[0].shape[1]
    lens = np.array([x.shape[0] for x in items], dtype=np.int64)
    total = int(lens.sum())
    n_off = len(items) + 1
    off_bytes = (n_off * 8 + 63) // 64 * 64
    n_bytes = off_bytes + total * width * 4
    slot = _STAGING.take(str(device), key, n_bytes)
    host = slot[0]
    offsets_host = host[:n_off * 8].view(torch.int64)
    offsets_host[0] = 0
    offsets_host[1:] = torch.from_numpy(np.cumsum(lens))
    arrays = [x if (x.flags['C_CONTIGUOUS'] and x.dtype == np.float32) else np.ascontiguousarray(x, dtype=np.float32) for x in items]
    srcs = (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
    sizes = (ctypes.c_int64 * len(arrays))(*[a.nbytes for a in arrays])
    lib = _lib.load()
    _lib.check(lib.mg_host_pack(ctypes.cast(srcs, ctypes.c_void_p), ctypes.cast(sizes, ctypes.c_void_p), len(arrays),
                                ctypes.c_void_p(host.data_ptr() + off_bytes), ctypes.c_int64(host.numel() - off_bytes), HOST_PACK_THREADS),
               'mg_host_pack')
    staged = host[:n_bytes].to(device, non_blocking=True)                       # ONE copy across PCIe: offsets + rows
    slot[1] = torch.cuda.Event()
    slot[1].record(torch.cuda.current_stream(device))
    offsets = staged[:n_off * 8].view(torch.int64)
    packed = staged[off_bytes:].view(torch.float32).view(total, width)
    return packed, offsets, lens


def _small_to_device(plain, device):
    """The batch's small host tensors (durations, lengths, ...) to ``device`` as ONE asynchronous copy out of pinned staging.  A
    ``tensor.to(device)`` from pageable memory is a synchronous copy on the current stream: it returns when everything queued there
    has run - the 49 MB feature copy in front of it included - so the host could not pack the next batch beside this one's transfer."""
    device = torch.device(device)
    tensors = [(k, v.contiguous()) for k, v in plain.items() if isinstance(v, torch.Tensor)]
    out = {k: v for k, v in plain.items() if not isinstance(v, torch.Tensor)}
    if device.type != 'cuda' or not tensors:
        out.update((k, v.to(device)) for k, v in tensors)
        return out
    spans, at = [], 0
    for _, v in tensors:
        spans.append(at)
        at += (v.numel() * v.element_size() + 63) // 64 * 64
    slot = _STAGING.take(str(device), '__small__', max(at, 64))
    host = slot[0]
    import ctypes
    for (_, v), lo in zip(tensors, spans):
        n = v.numel() * v.element_size()
        if n:
            # (memmove, not tensor.copy_: above 32 K elements a torch CPU op opens an OpenMP region on every core of the host, whose
            # workers then spin - on a box with a CPU quota that throttles the whole process for the rest of the scheduler period:
            # 90 ms stalls at random points of the loop, profiles/r5_notes_loader.txt)
            ctypes.memmove(host.data_ptr() + lo, v.data_ptr(), n)
    staged = host[:max(at, 64)].to(device, non_blocking=True)
    slot[1] = torch.cuda.Event()
    slot[1].record(torch.cuda.current_stream(device))
    for (k, v), lo in zip(tensors, spans):
        n = v.numel() * v.element_size()
        out[k] = staged[lo:lo + n].view(v.dtype).view(v.shape) if n else torch.empty(v.shape, dtype=v.dtype, device=device)
    return out


def collate_to_device(batch, normalisers, device, bf16_tables=()):
    """``load_utterance`` + ``collate_fn`` + ``to_device`` for a list of RAW per-utterance feature dicts, with the float
    sequence features normalised and zero padded on the device (reference: data.py:119-127, 159-224, 648-663).

    Every float32 sequence feature travels as ONE packed host buffer (utterances back to back, pinned when possible) and one
    kernel pass (mg_pad_normalise_f32) writes the padded raw feature and - where ``normalisers`` has its name - the
    ``normalised_`` twin; the host never touches per-frame data beyond the concatenation.  Other features (integer
    durations, scalars, names) take the ordinary collate path.  Same values as the reference's host pipeline to fp32
    rounding of the normaliser arithmetic (the host version divides in float32 NumPy as well).

    ``bf16_tables``: names of normalised phone-level features (``'normalised_lab'``) whose bf16 operand table the SAME pass writes
    (``mg_pad_normalise_bf16_f32``): the batch then carries ``name + '__bf16_table'`` and a bf16-precision model's training step
    launches no cast of the phone table (reference: the float32 cast on load, data.py:127)."""
    device = torch.device(device)
    bf16_tables = tuple(bf16_tables or ())
    out, rest = {}, []
    for key in batch[0].keys():
        first = batch[0][key]
        if not (isinstance(first, np.ndarray) and first.ndim == 2 and first.dtype == np.float32):
            rest.append(key)
            continue
        items = [item[key] for item in batch]
        if device.type == 'cuda':
            # pinned staging kept from batch to batch, packed by host threads, one H2D copy for rows and offsets
            packed, offsets, lens = _pack_pinned(items, device, key)
        else:
            lens = np.array([x.shape[0] for x in items], dtype=np.int64)
            offsets = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int64))
            packed = torch.from_numpy(np.ascontiguousarray(np.concatenate(items, axis=0)))
        kind = p0 = p1 = None
        normaliser = normalisers.get(key) if normalisers is not None else None
        if isinstance(normaliser, FeatureNormaliser):
            spec, prm = _KINDS[normaliser.kind], normaliser.fetch_params(torch.Tensor)
            kind, (p0, p1) = spec['forward'], (prm[n].to(device) for n in spec['params'])
        if kind is not None and 'normalised_' + key in bf16_tables:
            raw, norm, table = ops.pad_normalise(packed, offsets, int(lens.max()), p0, p1, kind, bf16_extra_rows=ops.PHONE_RATE_EXTRA)
            out['normalised_' + key + BF16_TABLE_SUFFIX] = table
        else:
            raw, norm = ops.pad_normalise(packed, offsets, int(lens.max()), p0, p1, kind)
        out[key] = raw
        if norm is not None:
            out['normalised_' + key] = norm
    if rest:
        plain = collate_fn([{key: item[key] for key in rest} for item in batch])
        total = _host_total(plain.get(FRAME_COUNT_KEY))
        if total is not None:
            out[FRAME_COUNT_KEY + '_total'] = total          # see to_device
        for key in rest:                                  # integer sequence features with a normaliser (dur) stay on the host path
            normaliser = normalisers.get(key) if normalisers is not None else None
            if normaliser is not None and isinstance(batch[0][key], np.ndarray):
                plain['normalised_' + key] = collate_fn([{key: normaliser.normalise(item[key]).astype(np.float32)} for item in batch])[key]
        out.update(_small_to_device(plain, device))
    for key in bf16_tables:                               # features that did not take the fused pass (no normaliser, host path; pair planes)
        name, _, kind = key.partition(':')
        suffix = X3_TABLE_SUFFIX if kind == 'x3' else BF16_TABLE_SUFFIX
        if name in out and isinstance(out[name], torch.Tensor) and out[name].is_cuda and name + suffix not in out:
            add_bf16_table(out, key)
    return out


class NumpyBinarySource(object):
    """``{data_dir}/{name}/{base_name}.npy`` -> ``{name: array}``: the loader half of the un-vendored
    ``tts_data_tools.data_sources.NumpyBinarySource`` the reference's models name in ``train_data_sources``
    (models/f0_test_model.py:60-69).  ``FilesDataset`` only needs a ``use_deltas`` attribute and a call
    ``(base_name, data_dir) -> dict`` (data.py:93, :135, :142).  Arrays come back as stored: float32 ``(len, D)`` sequence
    features, integer ``(P, 1)`` durations.  ``use_deltas`` additionally loads ``{name}_deltas``."""

    def __init__(self, name, use_deltas=False, ext='npy'):
        self.name, self.use_deltas, self.ext = name, use_deltas, ext

    def file_path(self, base_name, data_dir, name=None):
        return os.path.join(data_dir, name or self.name, '{}.{}'.format(base_name, self.ext))

    def __call__(self, base_name, data_dir):
        features = {self.name: np.load(self.file_path(base_name, data_dir))}
        if self.use_deltas:
            deltas = self.name + '_deltas'
            features[deltas] = np.load(self.file_path(base_name, data_dir, deltas))
        return features


class TextSource(object):
    """``{data_dir}/{name}/{base_name}.txt`` holding one number -> ``{name: int}`` (or float): the sentence-level counts
    (``n_frames``, ``n_phones``) of the reference's data layout (README.rst:139-150)."""

    use_deltas = False

    def __init__(self, name, ext='txt'):
        self.name, self.ext = name, ext

    def __call__(self, base_name, data_dir):
        with open(os.path.join(data_dir, self.name, '{}.{}'.format(base_name, self.ext))) as f:
            text = f.read().strip()
        try:
            return {self.name: int(text)}
        except ValueError:
            return {self.name: float(text)}


class FilesDataset(object):
    """File-backed utterances in front of ``DeviceBatches``: the reference's ``FilesDataset`` (data.py:60-157) - same constructor
    arguments, id-list handling (joined to ``data_root``, not to the split directory: data.py:100) and checks (:89-94).

    ``dataset[i]`` is what the reference's ``__getitem__`` returns: the features of every data source plus, for each feature with a
    normaliser, its ``normalised_`` twin computed on the host in NumPy and cast to float32 (:119-127, :144-150).
    ``dataset.raw(i)`` is the same utterance WITHOUT the twins - what ``DeviceBatches`` takes, because ``collate_to_device`` pads
    and normalises on the device in one pass.  Speaker-dependent normalisers are out of scope."""

    def __init__(self, data_sources, data_dir, id_list, normalisers, data_root='.'):
        for name, normaliser in normalisers.items():
            if name in data_sources and normaliser.use_deltas and not data_sources[name].use_deltas:
                raise ValueError(f'To normalise deltas of {name}, set `data_source.use_deltas` to True.')
        self.data_sources = data_sources
        self.data_root = data_root
        self.data_dir = os.path.join(data_root, data_dir)
        self.id_list = os.path.join(data_root, id_list)
        with open(self.id_list, 'r') as f:
            self.file_ids = [line.strip() for line in f if line.strip()]
        self.normalisers = normalisers

    def __len__(self):
        return len(self.file_ids)

    def raw(self, index):
        base_name = self.file_ids[index]
        features = {'name': base_name}
        for data_source in self.data_sources.values():
            features.update(data_source(base_name, self.data_dir))
        return features

    def __getitem__(self, index):
        features = self.raw(index)
        for name in self.data_sources:
            normaliser = self.normalisers.get(name)
            if normaliser is None:
                continue
            features['normalised_' + name] = normaliser.normalise(features[name]).astype(np.float32)
            if normaliser.use_deltas:
                deltas = name + '_deltas'
                features['normalised_' + deltas] = normaliser.normalise(features[deltas], deltas=True).astype(np.float32)
        return features

    collate_fn = staticmethod(collate_fn)


def batch(data_generator, batch_size=32, shuffle=True, num_data_threads=0, device='cuda:0', bf16_tables=()):
    """The reference's ``data.batch`` (data.py:29-57) for a ``FilesDataset``: a loader of device-resident batches.  Files are read
    on the calling thread as each batch is formed (``num_data_threads`` is accepted for signature compatibility; worker
    subprocesses are the reference's answer to a host-bound collate, which here runs on the device)."""
    rng = np.random.RandomState(torch.initial_seed() % (2 ** 32)) if shuffle else None
    return DeviceBatches(data_generator, batch_size, data_generator.normalisers, device, shuffle=rng, bf16_tables=bf16_tables)


class DeviceBatches(object):
    """The DataLoader + ``ToDeviceWrapper`` of the reference (data.py:50-55, :648-663): an iterable of feature dicts on ``device``,
    each batch padded and normalised there by ``collate_to_device``.  ``utterances`` is a ``FilesDataset`` (read lazily, batch by
    batch, through ``raw``) or a sequence of utterances that are already in host memory.

    ``bf16_tables``: see ``collate_to_device``.  ``utterances`` is a sequence of RAW per-utterance feature dicts (what a ``_DataSource`` returns: float32 ``(len, D)``
    arrays, integer ``dur``, python ints, the name); ``normalisers`` maps feature names to normalisers (``Normalisers`` or a
    dict).  Batches are contiguous slices in the given order, or a fresh permutation per epoch from ``shuffle`` = a
    ``numpy.random.RandomState`` (the reference shuffles with torch's global generator, data.py:50); the last, smaller batch
    is kept, as ``DataLoader`` does by default.  ``ExperimentBuilder.train_epoch`` takes it like any other loader."""

    def __init__(self, utterances, batch_size, normalisers, device, shuffle=None, bf16_tables=()):
        if batch_size <= 0:
            raise ValueError('batch_size must be positive, got %r' % (batch_size,))
        self.utterances = utterances if isinstance(utterances, FilesDataset) else list(utterances)
        self.batch_size = int(batch_size)
        self.normalisers = normalisers
        self.device = torch.device(device)
        self.shuffle = shuffle
        self.bf16_tables = tuple(bf16_tables or ())

    def use_bf16_tables(self, names):
        """The loader half of bf16 mode: every batch from now on carries the bf16 operand tables of these (normalised, phone-level)
        features - ``ExperimentBuilder`` asks for what its model's ``bf16_table_features()`` names."""
        self.bf16_tables = tuple(names or ())
        return self

    def __len__(self):
        return (len(self.utterances) + self.batch_size - 1) // self.batch_size

    def resident(self):
        """One pass of this loader kept on the device: the list of its batches (operand tables included).  A corpus of this model
        family fits the 288 GB of an MI355X many times over, and a host loader cannot feed the step (a C2 batch is 49 MB of float32
        over PCIe: >= 1 ms against a 0.1 ms step) - so later epochs take the list: ``ExperimentBuilder.train_epoch(use_graphs=True)``
        replays ``graph_group`` consecutive batches per graph launch, read where they lie.  The batches keep this pass's composition
        and order (a ``shuffle`` applies to the pass that builds the list)."""
        return list(iter(self))

    def __iter__(self):
        order = np.arange(len(self.utterances))
        if self.shuffle is not None:
            order = self.shuffle.permutation(len(self.utterances))
        for start in range(0, len(order), self.batch_size):
            fetch = self.utterances.raw if isinstance(self.utterances, FilesDataset) else self.utterances.__getitem__
            batch = [fetch(int(i)) for i in order[start:start + self.batch_size]]
            yield collate_to_device(batch, self.normalisers, self.device, bf16_tables=self.bf16_tables)
