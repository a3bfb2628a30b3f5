"""ORACLE (timing + cross-check leg) - the reference's train step restated with the SAME torch-CPU ops in the same order.

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
The reference's CPU path *is* PyTorch-CPU eager fp32; its Python files cannot travel to the GPU box, so this file is
what is timed there as ``cpu_baseline`` (kind "port").  It is pinned against the imported reference by
``tests/golden/make_golden.py`` (same fixtures as ``ref_cpu.py``).

Op sequence per step (reference lines):
  optimizer.zero_grad()                      experiment_builder.py:468
  upsample_to_repetitions                    utils.py:198-226 (sum/max .item(), per-item np.repeat loop, index gather)
  Linear/Sigmoid stack or GRU wrapper        README.rst:65-73, utils.py:345-418
  losses.mse with seq_len mask               losses.py:29-44, utils.py:115-144
  loss.backward(); Adam.step()               experiment_builder.py:473-474, :516
"""
import numpy as np
import torch
import torch.nn as nn


def upsample_to_repetitions(sequence_feature, repeats):
    """utils.py:175-228, op for op."""
    batch_size, max_seq_len, feat_dim = sequence_feature.shape
    repeated_lens = torch.sum(repeats, dim=1)
    max_repeated_len = torch.max(repeated_lens).item()
    repeats = repeats.reshape((batch_size, -1))
    padder = torch.zeros((batch_size, 1, feat_dim), dtype=sequence_feature.dtype)
    with_padder = torch.cat((sequence_feature, padder), dim=1)
    batch_idxs = torch.arange(batch_size)[:, None].repeat(1, max_repeated_len)
    repeated_idxs = -1 * np.ones((batch_size, max_repeated_len), dtype=np.int64)
    seq_feats_idx = np.arange(max_seq_len)
    for b, (repeat, repeated_len) in enumerate(zip(repeats.cpu(), repeated_lens.cpu())):
        repeated_idxs[b, :repeated_len] = np.repeat(seq_feats_idx, repeat)
    repeated_idxs = torch.tensor(repeated_idxs)
    return with_padder[batch_idxs, repeated_idxs]


def sequence_mask(seq_len, max_len=None, dtype=torch.uint8):
    """utils.py:115-144."""
    if max_len is None:
        max_len = torch.max(seq_len).item()
    rng = torch.arange(max_len).type(seq_len.dtype)
    mask = rng[None, :] < seq_len[:, None]
    return mask[:, :, None].type(dtype)


def mse(predictions, targets, seq_len=None):
    """losses.py:29-51."""
    feature_loss = torch.nn.functional.mse_loss(predictions, targets, reduction='none')
    if seq_len is None:
        feature_loss = torch.sum(feature_loss, dim=1) / feature_loss.shape[1]
    else:
        mask = sequence_mask(seq_len, max_len=feature_loss.shape[1], dtype=feature_loss.dtype)
        num_valid = torch.sum(mask, dim=1)
        feature_loss = torch.sum(feature_loss * mask, dim=1) / num_valid
    return torch.mean(feature_loss)


class GRUWrapper(nn.Module):
    """RecurrentCuDNNWrapper.forward with seq_len, utils.py:366-391."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, inputs, seq_len):
        sorted_idxs = torch.argsort(seq_len, descending=True)
        packed = nn.utils.rnn.pack_padded_sequence(inputs[sorted_idxs, ...], seq_len[sorted_idxs], batch_first=True)
        packed_out, hidden = self.layer(packed)
        sorted_out, _ = nn.utils.rnn.pad_packed_sequence(packed_out, batch_first=True)
        unsort = torch.argsort(sorted_idxs)
        return sorted_out[unsort, ...], hidden[:, unsort, :]


class F0Model(nn.Module):
    """README.rst:65-97 (Linear/Sigmoid 600-512-128-32-1)."""

    def __init__(self, dims=(600, 512, 128, 32, 1)):
        super().__init__()
        mods = []
        for i in range(len(dims) - 1):
            mods.append(nn.Linear(dims[i], dims[i + 1]))
            if i < len(dims) - 2:
                mods.append(nn.Sigmoid())
        self.layers = nn.Sequential(*mods)
        self.target_key = 'normalised_lf0'

    def forward(self, features):
        x = upsample_to_repetitions(features['normalised_lab'], features['dur'])
        pred = self.layers(x)
        return mse(pred, features[self.target_key], features['n_frames']), pred


class RNNModel(nn.Module):
    """Linear-sigmoid / GRU wrapper / Linear-sigmoid / Linear (layout models/RNN_SPSS.py:32-42, GRU cell)."""

    def __init__(self, lab_dim=600, hidden=512, post=256, out_dim=80, target_key='normalised_mcep'):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.Linear(lab_dim, hidden), nn.Sigmoid(), GRUWrapper(nn.GRU(hidden, hidden, batch_first=True)),
            nn.Linear(hidden, post), nn.Sigmoid(), nn.Linear(post, out_dim)])
        self.target_key = target_key

    def forward(self, features):
        n_frames = features['n_frames']
        x = upsample_to_repetitions(features['normalised_lab'], features['dur'])
        x = self.layers[1](self.layers[0](x))
        x, _ = self.layers[2](x, n_frames)
        pred = self.layers[5](self.layers[4](self.layers[3](x)))
        return mse(pred, features[self.target_key], n_frames), pred


def load_state(model, state):
    """Copy a numpy state_dict (morgana_amd.synthetic.*_state) into the torch module, same key names."""
    own = model.state_dict()
    for key, value in state.items():
        own[key].copy_(torch.from_numpy(np.asarray(value)))
    return model


def to_torch(features):
    out = {}
    for key, value in features.items():
        out[key] = torch.from_numpy(value) if isinstance(value, np.ndarray) else value
    return out


def train_steps(model, batches, n_steps, lr=0.01, weight_decay=0.0):
    """experiment_builder.py:468-480 per step.  Returns the list of per-step losses (python floats)."""
    optimizer = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)   # experiment_builder.py:516
    losses = []
    for step in range(n_steps):
        features = batches[step % len(batches)]
        optimizer.zero_grad()
        loss, _ = model(features)
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
    return losses
