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
