"""``morgana/viz/synthesis.py`` against this package: ``MLPG`` with the reference's signature and return conventions, solved on the
device (``csrc/mlpg.hip``) instead of on the host with ``bandmat``.

The reference calls this from ``predict`` of its shipped models (``models/f0_test_model.py:86-89``, ``models/RNN_SPSS.py:107-118``)
on every training step - a device -> host copy, B x D banded float64 solves in a Python loop, a host -> device copy - to feed the
LF0 / MCD metrics of ``loss``.  Here the delta streams never leave the device and nothing synchronises.
"""
import numpy as np
import torch

from .. import _lib, ops

DEFAULT_WINDOWS = (                                  # synthesis.py:122-127
    (0, 0, np.array([1.0])),
    (1, 1, np.array([-0.5, 0.0, 0.5])),
    (1, 1, np.array([1.0, -2.0, 1.0])),
)


def _device_of(*values):
    for v in values:                                 # synthesis.py:101-112: the first tensor's device
        if isinstance(v, torch.Tensor):
            return v.device
    return None


def MLPG(means, variances, windows=None, padding_size=0, seq_len=None):
    r"""Performs maximum-likelihood parameter generation (synthesis.py:79-178).

    means : (batch_size, seq_len, feat_dim) or (seq_len, feat_dim); variances : the same shape, or (feat_dim,) global;
    windows : list of (l, u, coefficients), default static / delta / delta-delta; padding_size : frames repeated at either end
    as burn-in; seq_len : (batch_size,) lengths (frames past them come back as zeros).

    Returns the most probable trajectory, (batch_size, seq_len, feat_dim // len(windows)) or without the batch axis for a
    single sequence: a float32 tensor on the inputs' device if any input was a tensor (as the reference), else a float64 array.
    Tensors must live on the MI355X; numpy inputs are uploaded to the current device (float32, the dtype the model emits).
    There is no host implementation: without the HIP library this raises.
    """
    device = _device_of(means, variances, seq_len)
    as_tensor = device is not None
    if device is None:
        if not torch.cuda.is_available():
            raise _lib.MorganaHipError('MLPG runs only on an MI355X device (no CPU fallback)')
        device = torch.device('cuda', torch.cuda.current_device())

    def put(x, dtype):
        if isinstance(x, torch.Tensor):
            return x.detach().to(device=device, dtype=dtype)
        return torch.as_tensor(np.ascontiguousarray(x)).to(device=device, dtype=dtype)

    means = put(means, torch.float32)
    variances = put(variances, torch.float32)
    if windows is None:
        windows = DEFAULT_WINDOWS
    using_batches = means.dim() != 2                  # :129-133
    if not using_batches:
        means = means[None]
        if variances.dim() == 2:
            variances = variances[None]               # :143-144
    if seq_len is not None:
        seq_len = put(seq_len, torch.int64).reshape(-1)
    out = ops.mlpg(means, variances, windows, padding_size=padding_size, seq_len=seq_len,
                   out_dtype=torch.float32 if as_tensor else torch.float64)
    if not using_batches:
        out = out.squeeze(0)                          # :173-174
    return out if as_tensor else out.cpu().numpy()    # :176-178
