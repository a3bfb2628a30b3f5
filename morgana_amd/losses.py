"""Mirror of ``morgana.losses`` for the hot path.  Reference: morgana/losses.py:9-51 (``sequence_loss`` / ``mse``)."""
import torch

from . import functional as F_hip


def mse(predictions, targets, seq_len=None):
    """Masked mean-squared error: per-utterance mean over valid frames, then mean over (batch, feature).

    Argument order is the reference wrapper's ``(predictions, targets, seq_len)`` (losses.py:30).  One HIP pass
    computes the loss and d loss / d predictions (the reference runs mse_loss, a host-built mask, mul, two sums, div
    and mean, then their autograd mirrors).  ``seq_len[b] == 0`` gives NaN, as in the reference.
    """
    if seq_len is not None and seq_len.dtype != torch.int64:
        seq_len = seq_len.long()
    if targets.shape[1] != predictions.shape[1]:
        raise RuntimeError('The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton dimension 1'
                           % (predictions.shape[1], targets.shape[1]))
    return F_hip.MaskedMSEFn.apply(predictions, targets, seq_len)


def bce(predictions, targets, seq_len=None):
    """Masked binary cross entropy with the same per-utterance averaging as ``mse`` (reference: losses.py:54-56; the voicing
    stream of models/RNN_SPSS.py:137).  Logs are clamped at -100, as ``F.binary_cross_entropy`` does."""
    if seq_len is not None and seq_len.dtype != torch.int64:
        seq_len = seq_len.long()
    if targets.shape[1] != predictions.shape[1]:
        raise RuntimeError('The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton dimension 1'
                           % (predictions.shape[1], targets.shape[1]))
    return F_hip.MaskedMSEFn.apply(predictions, targets, seq_len, 'bce')


def multi_stream(predictions, targets, kinds, seq_len=None, want_prob=False):
    """Mean over streams of ``mse`` / ``bce(sigmoid(.))`` on column slices of one prediction tensor - the loss of the
    reference's LSTM acoustic model (models/RNN_SPSS.py:120-139: three ``losses.mse`` + one ``losses.bce``, ``/ 4.``) in one
    pass instead of torch.split + four masked losses and their autograd mirrors.

    predictions (B, T, sum of widths); targets[k] (B, T, width_k) in column order; kinds[k] in {'mse', 'sigmoid_bce'}.
    Returns (loss, sigmoid(predictions) of the BCE stream if ``want_prob`` else None)."""
    if seq_len is not None and seq_len.dtype != torch.int64:
        seq_len = seq_len.long()
    targets = [y if y.dtype == torch.float32 else y.float() for y in targets]
    return F_hip.StreamLossFn.apply(predictions, seq_len, tuple(kinds), want_prob, *targets)
