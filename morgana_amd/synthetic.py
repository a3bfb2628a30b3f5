"""Synthetic (lab, dur, n_frames, target) utterance batches and deterministic parameter initialisation.

Pure numpy: shared by the HIP path (bench / tests), the oracle timing leg and the golden-vector generator, so that
every side sees bit-identical inputs. Shapes and distributions follow SURVEY.md section 8(d):

* ``normalised_lab`` ~ U[0, 1)  (B, P, lab_dim) float32   (min-max range, reference README.rst:80)
* ``dur``            int64 (B, P, 1), every real phone >= 1 frame, trailing pad phones have dur 0
* ``n_frames``       int64 (B,)   = sum of dur
* target             ~ N(0, 1)   (B, Tmax, out_dim) float32, zero beyond n_frames (collate zero-pad, data.py:183-193)

The feature dict keys are the ones the reference's ``FilesDataset.collate_fn`` yields (data.py:131-150).
"""
import numpy as np

REFERENCE_SEED = 1234567890  # models/f0_test_model.py:141


def _durations(rng, n_frames, n_phones):
    """Split ``n_frames`` into ``n_phones`` strictly positive integer durations (sorted distinct cut points)."""
    if n_phones == 1:
        return np.array([n_frames], dtype=np.int64)
    cuts = np.sort(rng.choice(np.arange(1, n_frames), size=n_phones - 1, replace=False))
    edges = np.concatenate(([0], cuts, [n_frames]))
    return np.diff(edges).astype(np.int64)


def make_batch(batch_size, n_frames, lab_dim=600, out_dim=1, target_name='lf0', frames_per_phone=12.5,
               seed=REFERENCE_SEED, rank=0):
    """Build one feature dict of numpy arrays.

    ``n_frames`` is an int (fixed length, configs C1-C4) or a ``(lo, hi)`` tuple (C5: T_b ~ U{lo..hi}).
    """
    rng = np.random.RandomState((seed + rank) % (2 ** 32))
    if isinstance(n_frames, (tuple, list)):
        lens = rng.randint(n_frames[0], n_frames[1] + 1, size=batch_size).astype(np.int64)
    else:
        lens = np.full((batch_size,), int(n_frames), dtype=np.int64)
    n_phones = np.maximum(1, np.round(lens / frames_per_phone).astype(np.int64))
    n_phones = np.minimum(n_phones, lens)  # every phone needs at least one frame
    max_p, max_t = int(n_phones.max()), int(lens.max())

    dur = np.zeros((batch_size, max_p, 1), dtype=np.int64)
    lab = np.zeros((batch_size, max_p, lab_dim), dtype=np.float32)
    tgt = np.zeros((batch_size, max_t, out_dim), dtype=np.float32)
    for b in range(batch_size):
        p, t = int(n_phones[b]), int(lens[b])
        dur[b, :p, 0] = _durations(rng, t, p)
        lab[b, :p] = rng.random_sample((p, lab_dim)).astype(np.float32)
        tgt[b, :t] = rng.standard_normal((t, out_dim)).astype(np.float32)

    return {
        'name': ['synthetic_%d_%05d' % (rank, b) for b in range(batch_size)],
        'normalised_lab': lab,
        'dur': dur,
        'n_frames': lens,
        'n_phones': n_phones,
        'normalised_' + target_name: tgt,
    }


def shard_batch(features, rank, world_size):
    """Contiguous utterance shard ``[r*B/R, (r+1)*B/R)`` of a global batch.

    Sequence features are cropped to the shard's own maximum length, as the reference's collate would have padded a
    batch that only held these utterances (data.py:183-193).
    """
    batch_size = len(features['n_frames'])
    if batch_size % world_size != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (batch_size, world_size))
    per = batch_size // world_size
    sl = slice(rank * per, (rank + 1) * per)
    out = {}
    for key, value in features.items():
        out[key] = value[sl]
    max_t = int(out['n_frames'].max())
    max_p = int(out['n_phones'].max()) if 'n_phones' in out else None
    for key, value in out.items():
        if isinstance(value, np.ndarray) and value.ndim == 3:
            if key in ('dur', 'normalised_lab', 'lab') and max_p is not None:
                out[key] = np.ascontiguousarray(value[:, :max_p])
            elif key not in ('dur', 'normalised_lab', 'lab'):
                out[key] = np.ascontiguousarray(value[:, :max_t])
    return out


def init_linear(rng, in_dim, out_dim):
    """U(-1/sqrt(in), 1/sqrt(in)) weight (out, in) and bias (out,) - the range torch's nn.Linear default uses."""
    bound = 1.0 / np.sqrt(in_dim)
    w = rng.uniform(-bound, bound, size=(out_dim, in_dim)).astype(np.float32)
    b = rng.uniform(-bound, bound, size=(out_dim,)).astype(np.float32)
    return w, b


def init_gru(rng, in_dim, hidden):
    """U(-1/sqrt(H), 1/sqrt(H)) GRU parameters in torch's layout: gates stacked (r, z, n) along dim 0."""
    bound = 1.0 / np.sqrt(hidden)
    w_ih = rng.uniform(-bound, bound, size=(3 * hidden, in_dim)).astype(np.float32)
    w_hh = rng.uniform(-bound, bound, size=(3 * hidden, hidden)).astype(np.float32)
    b_ih = rng.uniform(-bound, bound, size=(3 * hidden,)).astype(np.float32)
    b_hh = rng.uniform(-bound, bound, size=(3 * hidden,)).astype(np.float32)
    return w_ih, w_hh, b_ih, b_hh


def f0_model_state(seed=REFERENCE_SEED, dims=(600, 512, 128, 32, 1)):
    """state_dict (numpy) of the README F0Model stack; keys are the reference's (`layers.{i}.weight`, README.rst:65-73)."""
    rng = np.random.RandomState(seed % (2 ** 32))
    state = {}
    for i in range(len(dims) - 1):
        w, b = init_linear(rng, dims[i], dims[i + 1])
        state['layers.%d.weight' % (2 * i)] = w
        state['layers.%d.bias' % (2 * i)] = b
    return state


def rnn_spss_state(seed=REFERENCE_SEED, lab_dim=600, hidden=512, post=256, out_dim=80):
    """state_dict (numpy) of Linear-sigmoid-GRU-Linear-sigmoid-Linear (layout of models/RNN_SPSS.py:32-42, one GRU)."""
    rng = np.random.RandomState(seed % (2 ** 32))
    state = {}
    state['layers.0.weight'], state['layers.0.bias'] = init_linear(rng, lab_dim, hidden)
    w_ih, w_hh, b_ih, b_hh = init_gru(rng, hidden, hidden)
    state['layers.2.layer.weight_ih_l0'] = w_ih
    state['layers.2.layer.weight_hh_l0'] = w_hh
    state['layers.2.layer.bias_ih_l0'] = b_ih
    state['layers.2.layer.bias_hh_l0'] = b_hh
    state['layers.3.weight'], state['layers.3.bias'] = init_linear(rng, hidden, post)
    state['layers.5.weight'], state['layers.5.bias'] = init_linear(rng, post, out_dim)
    return state


ACOUSTIC_STREAMS = (('lf0', 3, 'mse'), ('vuv', 1, 'sigmoid_bce'), ('mcep', 180, 'mse'), ('bap', 15, 'mse'))


def init_lstm(rng, in_dim, hidden):
    """U(-1/sqrt(H), 1/sqrt(H)) LSTM parameters in torch's layout: gates stacked (i, f, g, o) along dim 0."""
    bound = 1.0 / np.sqrt(hidden)
    w_ih = rng.uniform(-bound, bound, size=(4 * hidden, in_dim)).astype(np.float32)
    w_hh = rng.uniform(-bound, bound, size=(4 * hidden, hidden)).astype(np.float32)
    b_ih = rng.uniform(-bound, bound, size=(4 * hidden,)).astype(np.float32)
    b_hh = rng.uniform(-bound, bound, size=(4 * hidden,)).astype(np.float32)
    return w_ih, w_hh, b_ih, b_hh


def lstm_acoustic_state(seed=REFERENCE_SEED, input_dim=609, hidden=512, post=256, output_dim=199, num_layers=8):
    """state_dict (numpy) of the reference's LSTMAcousticModel (models/RNN_SPSS.py:32-42): Linear, Sigmoid, Dropout,
    ``num_layers`` single-layer LSTM wrappers, Linear, Sigmoid, Dropout, Linear - keys are the reference's."""
    rng = np.random.RandomState(seed % (2 ** 32))
    state = {}
    state['layers.0.weight'], state['layers.0.bias'] = init_linear(rng, input_dim, hidden)
    for k in range(num_layers):
        w_ih, w_hh, b_ih, b_hh = init_lstm(rng, hidden, hidden)
        prefix = 'layers.%d.layer.' % (3 + k)
        state[prefix + 'weight_ih_l0'], state[prefix + 'weight_hh_l0'] = w_ih, w_hh
        state[prefix + 'bias_ih_l0'], state[prefix + 'bias_hh_l0'] = b_ih, b_hh
    state['layers.%d.weight' % (3 + num_layers)], state['layers.%d.bias' % (3 + num_layers)] = init_linear(rng, hidden, post)
    state['layers.%d.weight' % (6 + num_layers)], state['layers.%d.bias' % (6 + num_layers)] = init_linear(rng, post, output_dim)
    return state


def make_acoustic_batch(batch_size, n_frames, lab_dim=600, counters_dim=9, streams=ACOUSTIC_STREAMS, frames_per_phone=12.5,
                        seed=REFERENCE_SEED, rank=0, with_raw=False):
    """Feature dict for the LSTM acoustic model (models/RNN_SPSS.py:60-71, 73-82): ``make_batch``'s lab / dur / n_frames plus
    frame-level ``normalised_counters`` ~ U[0,1) and one target per stream - ``normalised_<name>_deltas`` ~ N(0,1) for the
    regression streams, ``vuv`` in {0, 1}; pads beyond each utterance's length are zero."""
    feats = make_batch(batch_size, n_frames, lab_dim=lab_dim, out_dim=1, target_name='unused', frames_per_phone=frames_per_phone,
                       seed=seed, rank=rank)
    del feats['normalised_unused']
    rng = np.random.RandomState((seed + rank + 7919) % (2 ** 32))
    lens = feats['n_frames']
    max_t = int(lens.max())
    mask = (np.arange(max_t)[None, :] < lens[:, None])[..., None]
    feats['normalised_counters'] = (rng.random_sample((batch_size, max_t, counters_dim)) * mask).astype(np.float32)
    for name, width, kind in streams:
        if kind == 'mse':
            feats['normalised_%s_deltas' % name] = (rng.standard_normal((batch_size, max_t, width)) * mask).astype(np.float32)
        else:
            feats[name] = ((rng.random_sample((batch_size, max_t, width)) > 0.4) * mask).astype(np.float32)
    if with_raw:
        # the un-normalised static features the shipped models' metrics compare the MLPG trajectories with
        # (models/RNN_SPSS.py:124-129, models/f0_test_model.py:101-103); drawn from their own stream so the rest is unchanged
        raw = np.random.RandomState((seed + rank + 104729) % (2 ** 32))
        for name, width, kind in streams:
            if kind == 'mse':
                offset = 5.0 if name == 'lf0' else 0.0
                feats[name] = ((raw.standard_normal((batch_size, max_t, width // 3)) * 0.3 + offset) * mask).astype(np.float32)
        if 'vuv' not in feats:
            feats['vuv'] = ((raw.random_sample((batch_size, max_t, 1)) > 0.4) * mask).astype(np.float32)
    return feats


def acoustic_normalisers(model, device='cpu', seed=REFERENCE_SEED):
    """Install synthetic mean-variance parameters (static and delta) on the model's delta-stream normalisers, standing in for
    the ``*_mvn.json`` / ``*_deltas_mvn.json`` files ``ExperimentBuilder`` loads (morgana/data.py:362-385): with them the shipped
    models run their MLPG + metrics step as under the reference's builder."""
    rng = np.random.RandomState((seed + 15485863) % (2 ** 32))
    model.normalisers = model.normaliser_sources()
    widths = getattr(model, 'output_dims', None) or {'lf0': model.output_dim}
    for name, norm in model.normalisers.items():
        if not getattr(norm, 'use_deltas', False):
            continue
        width = widths[name]
        mean = rng.standard_normal(width).astype(np.float32) * 0.1
        mean[:width // 3] += 5.0 if name == 'lf0' else 0.0
        std = rng.uniform(0.2, 0.6, width).astype(np.float32)
        norm.set_params({'mean': mean[:width // 3], 'std_dev': std[:width // 3]}, {'mean': mean, 'std_dev': std}, device=device)
    return model.normalisers


def gru_f0_state(seed=REFERENCE_SEED, input_dim=609, d1=256, hidden=64, post=64, output_dim=3):
    """state_dict (numpy) of the shipped F0 model (models/f0_test_model.py:28-45): Linear(input, d1), Sigmoid, Dropout,
    GRU(d1, hidden), Dropout, GRU(hidden, hidden), Dropout, GRU(hidden, hidden), Dropout, Linear(hidden, post), Sigmoid,
    Dropout, Linear(post, output) - keys are the reference's (the wrappers sit at layers.3, .5, .7)."""
    rng = np.random.RandomState(seed % (2 ** 32))
    state = {}
    state['layers.0.weight'], state['layers.0.bias'] = init_linear(rng, input_dim, d1)
    for idx, in_dim in ((3, d1), (5, hidden), (7, hidden)):
        w_ih, w_hh, b_ih, b_hh = init_gru(rng, in_dim, hidden)
        prefix = 'layers.%d.layer.' % idx
        state[prefix + 'weight_ih_l0'], state[prefix + 'weight_hh_l0'] = w_ih, w_hh
        state[prefix + 'bias_ih_l0'], state[prefix + 'bias_hh_l0'] = b_ih, b_hh
    state['layers.9.weight'], state['layers.9.bias'] = init_linear(rng, hidden, post)
    state['layers.12.weight'], state['layers.12.bias'] = init_linear(rng, post, output_dim)
    return state
