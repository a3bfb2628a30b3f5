#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF (build container only).

Usage (from the repo root, in the container that has /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (ZackHodari/morgana, pure Python) is imported from /root/reference with empty in-memory stand-ins for
its un-vendored third-party imports (tts_data_tools, tensorboardX, bandmat) - SURVEY.md Appendix A.  Nothing of the
reference's source is written anywhere: the .npz files hold only inputs and the outputs the reference computed.
Model weights come from this repo's deterministic numpy generator (morgana_amd.synthetic) loaded into reference-shaped
torch modules, so fixtures stay small (curves, norms, sampled elements) and reproducible on the GPU box.

Fixtures:
  g1_upsample_index.npz   dur cases -> int64 frame->phone map with -1 pads (bit exact)
  g2_upsample_values.npz  gathered values + autograd backward (segment sum)
  g3_sequence_mask.npz    masks for uint8 / float32 / int64
  g4_masked_mse.npz       loss + grad for D in {1, 80, 187}, with / without seq_len
  g5_normalisers.npz      mvn / minmax normalise + denormalise incl. std 0 and max == min
  g6_f0_model.npz         README F0Model at C1: 20-step Adam loss curve, step-1 grad norms + samples, final checksums
  g7_gru.npz              RecurrentCuDNNWrapper(GRU): outputs, final hidden, grads; small RNN_SPSS-layout model curve
  g9_ema_lr.npz           EMA update, Noam / CyclicNoam sequences, metrics.Mean
  g10 .. g14              LSTM wrapper, BCE, the shipped LSTM acoustic model and GRU F0 model, segment ops (see the functions)
  g15_metrics.npz         streaming metrics (Mean / RMSE / MAE / Distortion / MelCepDistortion / F0 / LF0): sum, count, result
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


def import_reference():
    names = ['tts_data_tools', 'tts_data_tools.file_io', 'tts_data_tools.utils', 'tts_data_tools.data_sources',
             'tensorboardX', 'bandmat', 'bandmat.linalg']
    for name in names:
        sys.modules[name] = types.ModuleType(name)
    tdt = sys.modules['tts_data_tools']
    tdt.file_io = sys.modules['tts_data_tools.file_io']
    tdt.utils = sys.modules['tts_data_tools.utils']
    tdt.data_sources = sys.modules['tts_data_tools.data_sources']
    tdt.utils.get_file_ids = lambda *a, **k: []
    tdt.file_io.load_json = lambda path: {}
    tdt.file_io.save_json = lambda obj, path: None
    sys.modules['tensorboardX'].SummaryWriter = type('SummaryWriter', (), {'__init__': lambda self, *a, **k: None})
    sys.modules['bandmat'].linalg = sys.modules['bandmat.linalg']
    sys.path.insert(0, '/root/reference')
    import morgana  # noqa: F401
    from morgana import utils, losses, data, base_models, lr_schedules, metrics
    return utils, losses, data, base_models, lr_schedules, metrics


def main():
    import torch
    import torch.nn as nn
    from morgana_amd import synthetic

    utils, losses, data, base_models, lr_schedules, metrics = import_reference()
    torch.manual_seed(synthetic.REFERENCE_SEED)
    torch.set_num_threads(4)
    rng = np.random.RandomState(20261003)

    # ---------------------------------------------------------------- G1: index map
    g1 = {}
    cases = {
        'zeros_inside': np.array([[2, 0, 3, 1], [0, 0, 4, 0], [1, 1, 1, 1]], dtype=np.int64),
        'trailing_pads': np.array([[5, 2, 0, 0, 0], [1, 1, 1, 1, 1], [7, 0, 0, 0, 0]], dtype=np.int64),
        'single_phone': np.array([[6]], dtype=np.int64),
        'ragged_totals': rng.randint(0, 9, size=(7, 11)).astype(np.int64),
        'tmax_boundary': np.array([[3, 3, 3], [9, 0, 0], [0, 0, 9], [4, 4, 0]], dtype=np.int64),
        'all_zero_item': np.array([[0, 0, 0], [2, 1, 3]], dtype=np.int64),
        'long': rng.randint(1, 40, size=(5, 64)).astype(np.int64),
    }

    def ref_index(dur):
        # The reference builds the map internally; recover it by upsampling a (B, P, 1) tensor holding the phone index
        # (+1 so that the zero pad row decodes to -1).
        b, p = dur.shape
        src = torch.arange(1, p + 1, dtype=torch.float64)[None, :, None].repeat(b, 1, 1)
        up = utils.upsample_to_repetitions(src, torch.from_numpy(dur)[:, :, None])
        return (up[:, :, 0].numpy().astype(np.int64) - 1)

    for name, dur in cases.items():
        g1[name + '__dur'] = dur
        g1[name + '__idx'] = ref_index(dur)
    np.savez_compressed(os.path.join(HERE, 'g1_upsample_index.npz'), **g1)

    # ---------------------------------------------------------------- G2: values + backward
    g2 = {}
    x = torch.from_numpy(rng.standard_normal((3, 5, 4)).astype(np.float32)).requires_grad_(True)
    dur = torch.tensor([[2, 0, 3, 1, 1], [1, 1, 1, 1, 1], [0, 4, 0, 0, 2]], dtype=torch.int64)
    up = utils.upsample_to_repetitions(x, dur[:, :, None])
    gout = torch.from_numpy(rng.standard_normal(tuple(up.shape)).astype(np.float32))
    up.backward(gout)
    g2.update(x=x.detach().numpy(), dur=dur.numpy(), out=up.detach().numpy(), grad_out=gout.numpy(),
              grad_x=x.grad.numpy())
    up2 = utils.upsample_to_repetitions(x.detach(), dur)          # 2-D durations are accepted (utils.py:202)
    g2['out_2d_dur'] = up2.numpy()
    np.savez_compressed(os.path.join(HERE, 'g2_upsample_values.npz'), **g2)

    # ---------------------------------------------------------------- G3: sequence_mask
    g3 = {}
    seq_len = torch.tensor([6, 4, 1, 7, 0], dtype=torch.int64)
    g3['seq_len'] = seq_len.numpy()
    g3['mask_default'] = utils.sequence_mask(seq_len).numpy()
    g3['mask_float32_len9'] = utils.sequence_mask(seq_len, max_len=9, dtype=torch.float32).numpy()
    g3['mask_long_len3'] = utils.sequence_mask(seq_len, max_len=3, dtype=torch.long).numpy()
    np.savez_compressed(os.path.join(HERE, 'g3_sequence_mask.npz'), **g3)

    # ---------------------------------------------------------------- G4: masked MSE
    g4 = {}
    for dim in (1, 80, 187):
        b, t = 5, 23
        p = torch.from_numpy(rng.standard_normal((b, t, dim)).astype(np.float32)).requires_grad_(True)
        y = torch.from_numpy(rng.standard_normal((b, t, dim)).astype(np.float32))
        sl = torch.tensor([23, 1, 17, 9, 20], dtype=torch.int64)
        loss = losses.mse(p, y, sl)
        loss.backward()
        g4['d%d__pred' % dim] = p.detach().numpy()
        g4['d%d__target' % dim] = y.numpy()
        g4['d%d__seq_len' % dim] = sl.numpy()
        g4['d%d__loss' % dim] = loss.detach().numpy()
        g4['d%d__grad' % dim] = p.grad.numpy().copy()
        p.grad = None
        loss = losses.mse(p, y)
        loss.backward()
        g4['d%d__loss_nolen' % dim] = loss.detach().numpy()
        g4['d%d__grad_nolen' % dim] = p.grad.numpy().copy()
    p = torch.zeros(2, 3, 1)
    g4['zero_len_loss'] = losses.mse(p, p + 1, torch.tensor([0, 2])).numpy()       # NaN (0/0), no guard
    np.savez_compressed(os.path.join(HERE, 'g4_masked_mse.npz'), **g4)

    # ---------------------------------------------------------------- G5: normalisers
    g5 = {}
    dim = 7
    feat = rng.standard_normal((3, 6, dim)).astype(np.float32) * 3 + 1
    mean = rng.standard_normal(dim).astype(np.float32)
    std = np.abs(rng.standard_normal(dim)).astype(np.float32)
    std[2] = 0.0
    mmin = rng.standard_normal(dim).astype(np.float32)
    mmax = mmin + np.abs(rng.standard_normal(dim)).astype(np.float32)
    mmax[4] = mmin[4]
    g5.update(feat=feat, mean=mean, std=std, mmin=mmin, mmax=mmax)
    for kind, arr in (('np', lambda a: a.copy()), ('torch', lambda a: torch.from_numpy(a.copy()))):
        conv = (lambda r: r) if kind == 'np' else (lambda r: r.numpy())
        with np.errstate(all='ignore'):
            g5['mvn_norm_' + kind] = conv(data.normalise_mvn(arr(feat), arr(mean), arr(std)))
            g5['mvn_denorm_' + kind] = conv(data.denormalise_mvn(arr(feat), arr(mean), arr(std)))
            g5['minmax_norm_' + kind] = conv(data.normalise_minmax(arr(feat), arr(mmin), arr(mmax)))
            g5['minmax_denorm_' + kind] = conv(data.denormalise_minmax(arr(feat), arr(mmin), arr(mmax)))
    np.savez_compressed(os.path.join(HERE, 'g5_normalisers.npz'), **g5)

    # ---------------------------------------------------------------- G6: README F0Model at C1
    class F0Model(base_models.BaseSPSS):
        def __init__(self, dims):
            super(F0Model, self).__init__()
            mods = []
            for i in range(len(dims) - 1):
                mods.append(nn.Linear(dims[i], dims[i + 1]))
                if i < len(dims) - 2:
                    mods.append(nn.Sigmoid())
            self.layers = utils.SequentialWithRecurrent(*mods)

        def predict(self, features):
            x = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'])
            pred, _ = self.layers(x, seq_len=features['n_frames'])          # returns (out, hiddens): utils.py:418
            return {'pred_norm_lf0': pred}

        def loss(self, features, output_features):
            return losses.mse(output_features['pred_norm_lf0'], features['normalised_lf0'], features['n_frames'])

    def to_torch(features):
        return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in features.items()}

    def load(model, state):
        own = model.state_dict()
        for k, v in state.items():
            own[k].copy_(torch.from_numpy(v))

    g6 = {}
    dims = (600, 512, 128, 32, 1)
    model = F0Model(dims)
    load(model, synthetic.f0_model_state())
    batches = [to_torch(synthetic.make_batch(8, 200, frames_per_phone=12.5, seed=synthetic.REFERENCE_SEED + 100 * i))
               for i in range(4)]
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=0.0)
    curve = []
    sample_idx = {}
    for step in range(20):
        optimizer.zero_grad()
        loss, out = model(batches[step % 4])
        loss.backward()
        if step == 0:
            g6['step1_pred_sample'] = out['pred_norm_lf0'].detach().numpy()[:, ::25, 0]
            for name, prm in model.named_parameters():
                g = prm.grad.detach().numpy().ravel()
                idx = rng.choice(g.size, size=min(64, g.size), replace=False)
                sample_idx[name] = idx
                g6['step1_gradnorm__' + name] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
                g6['step1_gradidx__' + name] = idx.astype(np.int64)
                g6['step1_gradval__' + name] = g[idx]
        optimizer.step()
        curve.append(loss.item())
    g6['loss_curve'] = np.array(curve, dtype=np.float64)
    for name, prm in model.named_parameters():
        v = prm.detach().numpy().astype(np.float64)
        g6['final_sum__' + name] = v.sum()
        g6['final_abs_sum__' + name] = np.abs(v).sum()
    # Ragged batch with weight decay, 5 steps (n_frames differ -> padding rows and per-utterance normalisation matter).
    model = F0Model(dims)
    load(model, synthetic.f0_model_state())
    ragged = to_torch(synthetic.make_batch(6, (40, 120), seed=77))
    optimizer = torch.optim.Adam(model.parameters(), lr=0.005, weight_decay=1e-3)
    curve = []
    for step in range(5):
        optimizer.zero_grad()
        loss, out = model(ragged)
        loss.backward()
        optimizer.step()
        curve.append(loss.item())
    g6['ragged_loss_curve'] = np.array(curve, dtype=np.float64)
    g6['ragged_last_pred_sample'] = out['pred_norm_lf0'].detach().numpy()[:, ::7, 0]
    np.savez_compressed(os.path.join(HERE, 'g6_f0_model.npz'), **g6)

    # ---------------------------------------------------------------- G7: GRU wrapper
    g7 = {}
    for tag, (bsz, t_in, i_dim, hid, lens) in {
            'h8': (4, 9, 3, 8, [6, 4, 1, 7]),
            'h32': (5, 12, 16, 32, [12, 3, 12, 7, 1])}.items():
        gru = nn.GRU(i_dim, hid, batch_first=True)
        st = synthetic.init_gru(np.random.RandomState(5 + hid), i_dim, hid)
        with torch.no_grad():
            gru.weight_ih_l0.copy_(torch.from_numpy(st[0]))
            gru.weight_hh_l0.copy_(torch.from_numpy(st[1]))
            gru.bias_ih_l0.copy_(torch.from_numpy(st[2]))
            gru.bias_hh_l0.copy_(torch.from_numpy(st[3]))
        wrapper = utils.RecurrentCuDNNWrapper(gru)
        xin = torch.from_numpy(rng.standard_normal((bsz, t_in, i_dim)).astype(np.float32)).requires_grad_(True)
        sl = torch.tensor(lens, dtype=torch.int64)
        out, hn = wrapper(xin, None, sl)
        gout = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32))
        ghn = torch.from_numpy(rng.standard_normal(tuple(hn.shape)).astype(np.float32))
        (out * gout).sum().backward(retain_graph=True)
        g7[tag + '__x'] = xin.detach().numpy()
        g7[tag + '__seq_len'] = sl.numpy()
        g7[tag + '__out'] = out.detach().numpy()
        g7[tag + '__hn'] = hn.detach().numpy()
        g7[tag + '__grad_out'] = gout.numpy()
        g7[tag + '__grad_x'] = xin.grad.numpy().copy()
        for pname in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0'):
            g7[tag + '__grad_' + pname] = getattr(gru, pname).grad.numpy().copy()
            getattr(gru, pname).grad = None
        xin.grad = None
        # with an initial hidden state and a gradient on the final hidden
        h0 = torch.from_numpy(rng.standard_normal((1, bsz, hid)).astype(np.float32)).requires_grad_(True)
        out, hn = wrapper(xin, h0, sl)
        ((out * gout).sum() + (hn * ghn).sum()).backward()
        g7[tag + '__h0'] = h0.detach().numpy()
        g7[tag + '__grad_hn'] = ghn.numpy()
        g7[tag + '__out_h0'] = out.detach().numpy()
        g7[tag + '__hn_h0'] = hn.detach().numpy()
        g7[tag + '__grad_x_h0'] = xin.grad.numpy().copy()
        g7[tag + '__grad_h0'] = h0.grad.numpy().copy()
        g7[tag + '__grad_weight_hh_l0_h0'] = gru.weight_hh_l0.grad.numpy().copy()

    class RNNModel(base_models.BaseSPSS):
        def __init__(self, lab_dim, hidden, post, out_dim):
            super(RNNModel, self).__init__()
            self.layers = utils.SequentialWithRecurrent(
                nn.Linear(lab_dim, hidden), nn.Sigmoid(),
                utils.RecurrentCuDNNWrapper(nn.GRU(hidden, hidden, batch_first=True)),
                nn.Linear(hidden, post), nn.Sigmoid(), nn.Linear(post, out_dim))

        def predict(self, features):
            x = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'])
            pred, hiddens = self.layers(x, seq_len=features['n_frames'])
            return {'pred': pred}

        def loss(self, features, output_features):
            return losses.mse(output_features['pred'], features['normalised_mcep'], features['n_frames'])

    lab_dim, hidden, post, out_dim = 40, 32, 24, 5
    model = RNNModel(lab_dim, hidden, post, out_dim)
    state = synthetic.rnn_spss_state(seed=31, lab_dim=lab_dim, hidden=hidden, post=post, out_dim=out_dim)
    load(model, state)
    feats = to_torch(synthetic.make_batch(6, (20, 60), lab_dim=lab_dim, out_dim=out_dim, target_name='mcep',
                                          frames_per_phone=6.0, seed=99))
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01)
    curve = []
    for step in range(8):
        optimizer.zero_grad()
        loss, out = model(feats)
        loss.backward()
        if step == 0:
            g7['rnn__step1_pred'] = out['pred'].detach().numpy()
            for name, prm in model.named_parameters():
                g7['rnn__step1_grad__' + name] = prm.grad.detach().numpy().copy()
        optimizer.step()
        curve.append(loss.item())
    g7['rnn__loss_curve'] = np.array(curve, dtype=np.float64)
    g7['rnn__dims'] = np.array([lab_dim, hidden, post, out_dim], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'g7_gru.npz'), **g7)

    # ---------------------------------------------------------------- G9: EMA, LR schedules, metrics.Mean
    g9 = {}
    lin_a, lin_b = nn.Linear(4, 3), nn.Linear(4, 3)
    g9['ema_shadow0'] = np.concatenate([p.detach().numpy().ravel() for p in lin_a.parameters()])
    ema = utils.ExponentialMovingAverage(lin_a, 0.9)
    steps = []
    for step in range(3):
        with torch.no_grad():
            for p in lin_b.parameters():
                p.add_(0.1 * (step + 1))
        steps.append(np.concatenate([p.detach().numpy().ravel() for p in lin_b.parameters()]))
        ema.update_params(lin_b)
    g9['ema_params_seq'] = np.stack(steps)
    g9['ema_shadow_final'] = np.concatenate([p.detach().numpy().ravel() for p in lin_a.parameters()])
    g9['ema_decay'] = np.float64(0.9)

    def lr_sequence(cls, n, **kwargs):
        opt = torch.optim.SGD([nn.Parameter(torch.zeros(1))], lr=1.0)
        sched = cls(opt, **kwargs)
        seq = []
        for _ in range(n):
            seq.append(opt.param_groups[0]['lr'])
            opt.step()
            sched.step()
        return np.array(seq, dtype=np.float64)

    g9['noam_w4'] = lr_sequence(lr_schedules.NoamLR, 12, warmup_steps=4)
    g9['cyclic_noam_w4_c9'] = lr_sequence(lr_schedules.CyclicNoamLR, 24, warmup_steps=4, cycle_steps=9)
    g9['cyclic_noam_w4_trig'] = lr_sequence(lr_schedules.CyclicNoamLR, 30, warmup_steps=4, cycle_trigger=0.5)
    g9['constant'] = lr_sequence(lr_schedules.DummyLR, 5)
    mean = metrics.Mean()
    vals = rng.standard_normal(6).astype(np.float32)
    for v in vals:
        mean.accumulate(torch.tensor(v))
    g9['mean_inputs'] = vals
    g9['mean_result'] = np.float64(float(mean.result()))
    np.savez_compressed(os.path.join(HERE, 'g9_ema_lr.npz'), **g9)

    for name in sorted(os.listdir(HERE)):
        if name.endswith('.npz'):
            print('%-28s %8d bytes' % (name, os.path.getsize(os.path.join(HERE, name))))


if __name__ == '__main__' and len(sys.argv) == 1:
    main()
    g8_plumbing_later = True


def g8_plumbing():
    """G8: a 3-epoch run of the reference's own ExperimentBuilder (argparse defaults -> FilesDataset -> train_epoch ->
    metrics.json) on a tiny on-disk data set, with /tmp-only stand-ins for the un-vendored tts_data_tools loaders.
    Records the data set, the batch order the shuffling DataLoader produced, per-epoch losses and final checksums."""
    import json
    import tempfile
    import torch
    import torch.nn as nn
    from morgana_amd import synthetic

    utils, losses, data, base_models, lr_schedules, metrics = import_reference()
    file_io = sys.modules['tts_data_tools.file_io']
    file_io.load_json = lambda path: json.load(open(path))
    file_io.save_json = lambda obj, path: json.dump(obj, open(path, 'w'))
    sys.modules['tts_data_tools.utils'].get_file_ids = lambda *a, **k: []
    from morgana import experiment_builder

    lab_dim, dims = 40, (40, 32, 16, 8, 1)
    rng = np.random.RandomState(808)
    root = tempfile.mkdtemp(prefix='morgana_g8_')
    n_utts = 22
    names = ['utt_%03d' % i for i in range(n_utts)]
    dataset = {}
    for split in ('train',):
        for feat in ('lab', 'dur', 'lf0', 'n_frames'):
            os.makedirs(os.path.join(root, split, feat), exist_ok=True)
    for name in names:
        n_ph = int(rng.randint(3, 9))
        dur = rng.randint(1, 7, size=(n_ph, 1)).astype(np.int64)
        n_fr = int(dur.sum())
        lab = rng.uniform(-2, 3, size=(n_ph, lab_dim)).astype(np.float32)
        lf0 = (5.0 + 0.3 * rng.standard_normal((n_fr, 1))).astype(np.float32)
        dataset[name] = dict(lab=lab, dur=dur, lf0=lf0, n_frames=n_fr)
        np.save(os.path.join(root, 'train', 'lab', name + '.npy'), lab)
        np.save(os.path.join(root, 'train', 'dur', name + '.npy'), dur)
        np.save(os.path.join(root, 'train', 'lf0', name + '.npy'), lf0)
        open(os.path.join(root, 'train', 'n_frames', name + '.txt'), 'w').write(str(n_fr))
    open(os.path.join(root, 'train_file_id_list.scp'), 'w').write('\n'.join(names) + '\n')
    all_lab = np.concatenate([d['lab'] for d in dataset.values()])
    all_lf0 = np.concatenate([d['lf0'] for d in dataset.values()])
    norm = {'lab_minmax': {'mmin': all_lab.min(0).tolist(), 'mmax': all_lab.max(0).tolist()},
            'lf0_mvn': {'mean': all_lf0.mean(0).tolist(), 'std_dev': all_lf0.std(0).tolist()}}
    os.makedirs(os.path.join(root, 'processed'), exist_ok=True)
    for key, val in norm.items():
        json.dump(val, open(os.path.join(root, 'processed', key + '.json'), 'w'))

    class NpySource(object):                       # satisfies what FilesDataset calls (data.py:93,135,142)
        def __init__(self, name, use_deltas=False, as_int=False):
            self.name, self.use_deltas, self.as_int = name, use_deltas, as_int

        def __call__(self, base_name, data_dir):
            if self.as_int:
                return {self.name: int(open(os.path.join(data_dir, self.name, base_name + '.txt')).read())}
            return {self.name: np.load(os.path.join(data_dir, self.name, base_name + '.npy'))}

    batch_log = []

    class F0Model(base_models.BaseSPSS):
        def __init__(self):
            super(F0Model, self).__init__()
            mods = []
            for i in range(len(dims) - 1):
                mods.append(nn.Linear(dims[i], dims[i + 1]))
                if i < len(dims) - 2:
                    mods.append(nn.Sigmoid())
            self.layers = utils.SequentialWithRecurrent(*mods)
            own = self.state_dict()
            for k, v in synthetic.f0_model_state(seed=4242, dims=dims).items():
                own[k].copy_(torch.from_numpy(v))

        def normaliser_sources(self):
            return {'lab': data.MinMaxNormaliser('lab'), 'lf0': data.MeanVarianceNormaliser('lf0')}

        def train_data_sources(self):
            return {'n_frames': NpySource('n_frames', as_int=True), 'dur': NpySource('dur'), 'lab': NpySource('lab'),
                    'lf0': NpySource('lf0')}

        def predict(self, features):
            batch_log.append(list(features['name']))
            x = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'])
            pred, _ = self.layers(x, seq_len=features['n_frames'])
            return {'pred_norm_lf0': pred}

        def loss(self, features, output_features):
            return losses.mse(output_features['pred_norm_lf0'], features['normalised_lf0'], features['n_frames'])

    argv = ['--experiment_name', 'g8', '--data_root', root, '--experiments_base', os.path.join(root, 'experiments'),
            '--batch_size', '8', '--end_epoch', '3', '--device', 'cpu', '--no-valid',
            '--learning_rate', '0.01', '--num_data_threads', '0', '--normalisation_dir', 'processed']
    old_argv = sys.argv
    sys.argv = ['g8'] + argv
    try:
        args = experiment_builder.ExperimentBuilder.get_experiment_args()
    finally:
        sys.argv = old_argv
    torch.manual_seed(synthetic.REFERENCE_SEED)
    exp = experiment_builder.ExperimentBuilder(F0Model, **args)
    exp.run_experiment()

    out = {'names': np.array(names), 'dims': np.array(dims, dtype=np.int64)}
    for name in names:
        for feat in ('lab', 'dur', 'lf0'):
            out['data__%s__%s' % (name, feat)] = dataset[name][feat]
    for key, val in norm.items():
        for k2, v2 in val.items():
            out['norm__%s__%s' % (key, k2)] = np.array(v2, dtype=np.float32)
    n_batches = len(batch_log) // 3
    out['batch_order'] = np.array([[','.join(b) for b in batch_log[e * n_batches:(e + 1) * n_batches]] for e in range(3)])
    epoch_loss = []
    for e in (1, 2, 3):
        m = json.load(open(os.path.join(root, 'experiments', 'g8', 'train', 'epoch_%d' % e, 'metrics.json')))
        epoch_loss.append(m['loss'])
    out['epoch_metrics_loss'] = np.array(epoch_loss, dtype=np.float64)
    for k, v in exp.model.state_dict().items():
        out['final_sum__' + k] = np.float64(v.double().sum().item())
        out['final_abs_sum__' + k] = np.float64(v.double().abs().sum().item())
    ckpt = torch.load(os.path.join(root, 'experiments', 'g8', 'checkpoints', 'epoch_3.pt'))
    out['checkpoint_keys'] = np.array(sorted(ckpt.keys()))
    np.savez_compressed(os.path.join(HERE, 'g8_plumbing.npz'), **out)
    print('g8_plumbing.npz', os.path.getsize(os.path.join(HERE, 'g8_plumbing.npz')), 'bytes; epoch losses', epoch_loss)


if __name__ == '__main__' and (len(sys.argv) == 1 or sys.argv[1] == 'g8'):
    g8_plumbing()


def g10_lstm():
    """G10: RecurrentCuDNNWrapper(nn.LSTM) of the reference: 1 and 2 layers, ragged seq_len, with and without (h0, c0)."""
    import torch
    import torch.nn as nn
    from morgana_amd import synthetic
    utils, losses, data, base_models, lr_schedules, metrics = import_reference()
    rng = np.random.RandomState(1010)
    g = {}
    for tag, (layers, bsz, t_in, i_dim, hid, lens) in {'l1': (1, 4, 9, 5, 8, [6, 9, 1, 4]), 'l2': (2, 5, 11, 12, 16, [11, 3, 7, 11, 2])}.items():
        lstm = nn.LSTM(i_dim, hid, num_layers=layers, batch_first=True)
        prm_rng = np.random.RandomState(77 + hid)
        with torch.no_grad():
            for name, prm in lstm.named_parameters():
                prm.copy_(torch.from_numpy(prm_rng.uniform(-0.4, 0.4, size=tuple(prm.shape)).astype(np.float32)))
                g['%s__param__%s' % (tag, name)] = prm.detach().numpy().copy()
        wrapper = utils.RecurrentCuDNNWrapper(lstm)
        x = torch.from_numpy(rng.standard_normal((bsz, t_in, i_dim)).astype(np.float32)).requires_grad_(True)
        sl = torch.tensor(lens, dtype=torch.int64)
        out, (hn, cn) = wrapper(x, None, sl)
        gout = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32))
        (out * gout).sum().backward()
        g[tag + '__x'], g[tag + '__seq_len'] = x.detach().numpy(), sl.numpy()
        g[tag + '__out'], g[tag + '__hn'], g[tag + '__cn'] = out.detach().numpy(), hn.detach().numpy(), cn.detach().numpy()
        g[tag + '__grad_out'], g[tag + '__grad_x'] = gout.numpy(), x.grad.numpy().copy()
        for name, prm in lstm.named_parameters():
            g['%s__grad__%s' % (tag, name)] = prm.grad.numpy().copy()
            prm.grad = None
        x.grad = None
        h0 = torch.from_numpy(rng.standard_normal((layers, bsz, hid)).astype(np.float32)).requires_grad_(True)
        c0 = torch.from_numpy(rng.standard_normal((layers, bsz, hid)).astype(np.float32)).requires_grad_(True)
        ghn = torch.from_numpy(rng.standard_normal((layers, bsz, hid)).astype(np.float32))
        gcn = torch.from_numpy(rng.standard_normal((layers, bsz, hid)).astype(np.float32))
        out, (hn, cn) = wrapper(x, (h0, c0), sl)
        ((out * gout).sum() + (hn * ghn).sum() + (cn * gcn).sum()).backward()
        g[tag + '__h0'], g[tag + '__c0'], g[tag + '__grad_hn'], g[tag + '__grad_cn'] = h0.detach().numpy(), c0.detach().numpy(), ghn.numpy(), gcn.numpy()
        g[tag + '__out_s'], g[tag + '__hn_s'], g[tag + '__cn_s'] = out.detach().numpy(), hn.detach().numpy(), cn.detach().numpy()
        g[tag + '__grad_x_s'], g[tag + '__grad_h0'], g[tag + '__grad_c0'] = x.grad.numpy().copy(), h0.grad.numpy().copy(), c0.grad.numpy().copy()
        g[tag + '__grad_whh0_s'] = lstm.weight_hh_l0.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'g10_lstm.npz'), **g)
    print('g10_lstm.npz', os.path.getsize(os.path.join(HERE, 'g10_lstm.npz')), 'bytes')


if __name__ == '__main__' and (len(sys.argv) == 1 or sys.argv[1] == 'g10'):
    g10_lstm()


def g11_bce():
    """G11: losses.bce of the reference (the voicing stream, models/RNN_SPSS.py:137): loss + grad, with / without seq_len,
    including saturated probabilities 0 and 1 where torch clamps the logs at -100."""
    import torch
    utils, losses, data, base_models, lr_schedules, metrics = import_reference()
    rng = np.random.RandomState(1111)
    g = {}
    for dim in (1, 3):
        b, t = 5, 23
        pn = (1.0 / (1.0 + np.exp(-3 * rng.standard_normal((b, t, dim))))).astype(np.float32)
        pn[0, 0, 0], pn[1, 0, 0], pn[2, 3, 0], pn[3, 2, 0] = 0.0, 1.0, 1.0, 0.0      # saturated: both right and wrong
        y = torch.from_numpy((rng.rand(b, t, dim) > 0.5).astype(np.float32))
        y[0, 0, 0], y[1, 0, 0], y[2, 3, 0], y[3, 2, 0] = 0.0, 1.0, 0.0, 1.0
        p = torch.from_numpy(pn).requires_grad_(True)
        sl = torch.tensor([23, 1, 17, 9, 20], dtype=torch.int64)
        loss = losses.bce(p, y, sl)
        loss.backward()
        g['d%d__pred' % dim] = pn
        g['d%d__target' % dim] = y.numpy()
        g['d%d__seq_len' % dim] = sl.numpy()
        g['d%d__loss' % dim] = loss.detach().numpy()
        g['d%d__grad' % dim] = p.grad.numpy().copy()
        p.grad = None
        loss = losses.bce(p, y)
        loss.backward()
        g['d%d__loss_nolen' % dim] = loss.detach().numpy()
        g['d%d__grad_nolen' % dim] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'g11_bce.npz'), **g)
    print('g11_bce.npz', os.path.getsize(os.path.join(HERE, 'g11_bce.npz')), 'bytes')


if __name__ == '__main__' and (len(sys.argv) == 1 or sys.argv[1] == 'g11'):
    g11_bce()


def g12_lstm_acoustic():
    """G12: the reference's LSTM acoustic model (models/RNN_SPSS.py:20-139) at toy size, assembled from the reference's own
    building blocks (the module itself needs pyworld / tts_data_tools.wav_gen / bandmat to import): layer container and
    predict/loss written out as in that file, MLPG left out (detached post-processing).  Holds the multi-stream loss alone
    (loss + gradient on random predictions) and 6 Adam steps of the whole model."""
    import torch
    import torch.nn as nn
    from morgana_amd import synthetic
    utils, losses, data, base_models, lr_schedules, metrics = import_reference()
    g = {}
    streams = (('lf0', 3, 'mse'), ('vuv', 1, 'sigmoid_bce'), ('mcep', 6, 'mse'), ('bap', 3, 'mse'))
    widths = [w for _, w, _ in streams]
    lab_dim, counters_dim, hidden, post, num_layers = 20, 4, 16, 12, 3

    def loss_fn(features, outputs):                                             # models/RNN_SPSS.py:120-139
        n_frames = features['n_frames']
        loss = 0.
        loss += losses.mse(outputs['normalised_lf0_deltas'], features['normalised_lf0_deltas'], n_frames)
        loss += losses.mse(outputs['normalised_mcep_deltas'], features['normalised_mcep_deltas'], n_frames)
        loss += losses.mse(outputs['normalised_bap_deltas'], features['normalised_bap_deltas'], n_frames)
        loss += losses.bce(outputs['vuv'].type(torch.float), features['vuv'].type(torch.float), n_frames)
        return loss / 4.

    def split(pred):                                                            # models/RNN_SPSS.py:86-93
        lf0, vuv, mcep, bap = torch.split(pred, widths, dim=-1)
        return {'normalised_lf0_deltas': lf0, 'normalised_mcep_deltas': mcep, 'normalised_bap_deltas': bap,
                'vuv': torch.sigmoid(vuv)}

    feats_np = synthetic.make_acoustic_batch(5, (10, 30), lab_dim=lab_dim, counters_dim=counters_dim, streams=streams,
                                             frames_per_phone=5.0, seed=1212)
    feats = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in feats_np.items()}

    # the loss alone, on random predictions (large logits included so that the sigmoid saturates)
    rng = np.random.RandomState(12)
    max_t = int(feats_np['n_frames'].max())
    pred_np = (rng.standard_normal((5, max_t, sum(widths))) * 2).astype(np.float32)
    pred_np[0, 0, 3], pred_np[1, 1, 3], pred_np[2, 2, 3] = 40.0, -40.0, 110.0
    pred = torch.from_numpy(pred_np).requires_grad_(True)
    loss = loss_fn(feats, split(pred))
    loss.backward()
    g['loss__pred'] = pred_np
    g['loss__value'] = loss.detach().numpy()
    g['loss__grad'] = pred.grad.numpy().copy()
    g['loss__vuv'] = torch.sigmoid(pred.detach()[..., 3:4]).numpy()

    class Model(base_models.BaseSPSS):
        def __init__(self):
            super(Model, self).__init__()
            self.layers = utils.SequentialWithRecurrent(                        # models/RNN_SPSS.py:32-42
                nn.Linear(lab_dim + counters_dim, hidden), nn.Sigmoid(), nn.Dropout(p=0.),
                *[utils.RecurrentCuDNNWrapper(nn.LSTM(hidden, hidden, dropout=0., batch_first=True))
                  for _ in range(num_layers)],
                nn.Linear(hidden, post), nn.Sigmoid(), nn.Dropout(p=0.),
                nn.Linear(post, sum(widths)))

        def predict(self, features):                                            # models/RNN_SPSS.py:73-85
            at_frame_rate = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'])
            model_inputs = torch.cat((at_frame_rate, features['normalised_counters']), dim=-1)
            pred, _ = self.layers(model_inputs, seq_len=features['n_frames'])
            return split(pred)

        def loss(self, features, output_features):
            return loss_fn(features, output_features)

    model = Model()
    state = synthetic.lstm_acoustic_state(seed=1213, input_dim=lab_dim + counters_dim, hidden=hidden, post=post,
                                          output_dim=sum(widths), num_layers=num_layers)
    own = model.state_dict()
    assert sorted(own.keys()) == sorted(state.keys()), (sorted(own.keys()), sorted(state.keys()))
    for k, v in state.items():
        own[k].copy_(torch.from_numpy(v))
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01)
    curve = []
    for step in range(6):
        optimizer.zero_grad()
        loss, out = model(feats)
        loss.backward()
        if step == 0:
            for name in ('normalised_lf0_deltas', 'normalised_mcep_deltas', 'normalised_bap_deltas', 'vuv'):
                g['model__step1_' + name] = out[name].detach().numpy()
            for name, prm in model.named_parameters():
                g['model__step1_grad__' + name] = prm.grad.detach().numpy().copy()
        optimizer.step()
        curve.append(loss.item())
    g['model__loss_curve'] = np.array(curve, dtype=np.float64)
    g['model__dims'] = np.array([lab_dim, counters_dim, hidden, post, num_layers] + widths, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'g12_lstm_acoustic.npz'), **g)
    print('g12_lstm_acoustic.npz', os.path.getsize(os.path.join(HERE, 'g12_lstm_acoustic.npz')), 'bytes; curve', curve)


if __name__ == '__main__' and (len(sys.argv) == 1 or sys.argv[1] == 'g12'):
    g12_lstm_acoustic()


def g13_gru_f0():
    """G13: the shipped F0 model (models/f0_test_model.py:21-107) at toy size from the reference's own building blocks:
    609-style input (upsampled labels + frame-level counters), Linear / sigmoid / dropout(0) / three single-layer GRU wrappers /
    Linear / sigmoid / Linear -> lf0 deltas, losses.mse; MLPG (detached post-processing) left out.  6 Adam steps."""
    import torch
    import torch.nn as nn
    from morgana_amd import synthetic
    utils, losses, data, base_models, lr_schedules, metrics = import_reference()
    lab_dim, counters_dim, d1, hid, post, out_dim = 20, 4, 24, 16, 16, 3

    class Model(base_models.BaseSPSS):
        def __init__(self):
            super(Model, self).__init__()
            self.layers = utils.SequentialWithRecurrent(                        # models/f0_test_model.py:28-45
                nn.Linear(lab_dim + counters_dim, d1), nn.Sigmoid(), nn.Dropout(p=0.),
                utils.RecurrentCuDNNWrapper(nn.GRU(d1, hid, batch_first=True)), nn.Dropout(p=0.),
                utils.RecurrentCuDNNWrapper(nn.GRU(hid, hid, batch_first=True)), nn.Dropout(p=0.),
                utils.RecurrentCuDNNWrapper(nn.GRU(hid, hid, batch_first=True)), nn.Dropout(p=0.),
                nn.Linear(hid, post), nn.Sigmoid(), nn.Dropout(p=0.),
                nn.Linear(post, out_dim))

        def predict(self, features):                                            # models/f0_test_model.py:76-84
            at_frame_rate = utils.upsample_to_repetitions(features['normalised_lab'], features['dur'])
            model_inputs = torch.cat((at_frame_rate, features['normalised_counters']), dim=-1)
            pred, _ = self.layers(model_inputs, seq_len=features['n_frames'])
            return {'normalised_lf0_deltas': pred}

        def loss(self, features, output_features):                              # models/f0_test_model.py:99-102
            return losses.mse(output_features['normalised_lf0_deltas'], features['normalised_lf0_deltas'], features['n_frames'])

    streams = (('lf0', out_dim, 'mse'),)
    feats_np = synthetic.make_acoustic_batch(5, (10, 30), lab_dim=lab_dim, counters_dim=counters_dim, streams=streams,
                                             frames_per_phone=5.0, seed=1313)
    feats = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in feats_np.items()}
    model = Model()
    state = synthetic.gru_f0_state(seed=1314, input_dim=lab_dim + counters_dim, d1=d1, hidden=hid, post=post, output_dim=out_dim)
    own = model.state_dict()
    assert sorted(own.keys()) == sorted(state.keys()), (sorted(own.keys()), sorted(state.keys()))
    for k, v in state.items():
        own[k].copy_(torch.from_numpy(v))
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01)
    g, curve = {}, []
    for step in range(6):
        optimizer.zero_grad()
        loss, out = model(feats)
        loss.backward()
        if step == 0:
            g['step1_pred'] = out['normalised_lf0_deltas'].detach().numpy()
            for name, prm in model.named_parameters():
                g['step1_grad__' + name] = prm.grad.detach().numpy().copy()
        optimizer.step()
        curve.append(loss.item())
    g['loss_curve'] = np.array(curve, dtype=np.float64)
    g['dims'] = np.array([lab_dim, counters_dim, d1, hid, post, out_dim], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'g13_gru_f0.npz'), **g)
    print('g13_gru_f0.npz', os.path.getsize(os.path.join(HERE, 'g13_gru_f0.npz')), 'bytes; curve', curve)


if __name__ == '__main__' and (len(sys.argv) == 1 or sys.argv[1] == 'g13'):
    g13_gru_f0()


def g14_segments():
    """G14: utils.split_to_segments / utils.get_segment_ends of the reference (utils.py:231-330): values and the gradient
    w.r.t. the sequence feature, with zero-length and trailing-pad segments."""
    import torch
    utils, losses, data, base_models, lr_schedules, metrics = import_reference()
    rng = np.random.RandomState(1414)
    g = {}
    for tag, (b, t, f, lens) in {
            'small': (3, 10, 4, [[3, 0, 4, 2], [10, 0, 0, 0], [1, 1, 1, 5]]),
            'wide': (4, 37, 9, [[5, 9, 7, 1, 3, 0, 0], [12, 12, 13, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1], [0, 20, 0, 17, 0, 0, 0]])}.items():
        x = torch.from_numpy(rng.standard_normal((b, t, f)).astype(np.float32)).requires_grad_(True)
        sl = torch.tensor(lens, dtype=torch.int64).unsqueeze(-1)
        seg = utils.split_to_segments(x, sl)
        gs = torch.from_numpy(rng.standard_normal(tuple(seg.shape)).astype(np.float32))
        (seg * gs).sum().backward()
        g[tag + '__x'] = x.detach().numpy()
        g[tag + '__lens'] = sl.numpy()
        g[tag + '__split'] = seg.detach().numpy()
        g[tag + '__split_grad_out'] = gs.numpy()
        g[tag + '__split_grad_x'] = x.grad.numpy().copy()
        x.grad = None
        ends = utils.get_segment_ends(x, sl)
        ge = torch.from_numpy(rng.standard_normal(tuple(ends.shape)).astype(np.float32))
        (ends * ge).sum().backward()
        g[tag + '__ends'] = ends.detach().numpy()
        g[tag + '__ends_grad_out'] = ge.numpy()
        g[tag + '__ends_grad_x'] = x.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'g14_segments.npz'), **g)
    print('g14_segments.npz', os.path.getsize(os.path.join(HERE, 'g14_segments.npz')), 'bytes')


if __name__ == '__main__' and (len(sys.argv) == 1 or sys.argv[1] == 'g14'):
    g14_segments()


def g15_metrics():
    """G15: the streaming metrics the shipped acoustic model registers (models/RNN_SPSS.py:44-48) plus their base classes,
    morgana/metrics.py:359-695: two accumulate calls each (ragged seq_len, then none where the class allows it) -> sum, count
    and result as the reference computes them."""
    import torch
    utils, losses, data, base_models, lr_schedules, metrics = import_reference()
    rng = np.random.RandomState(1515)
    g = {}
    b, t = 5, 23
    seq = np.array([23, 7, 15, 1, 20], dtype=np.int64)
    g['seq_len'] = seq
    sl = torch.from_numpy(seq)

    def arr(d, scale=1.0, shift=0.0):
        return (rng.standard_normal((b, t, d)) * scale + shift).astype(np.float32)

    def run(name, metric, calls):
        metric.reset_state()           # as Handler.reset_state does at the start of an epoch (RMSE.__init__ does not create the sums)
        for args in calls:
            metric.accumulate(*[torch.from_numpy(a) if isinstance(a, np.ndarray) else a for a in args])
        g[name + '__sum'] = np.float64(float(metric.sum))
        g[name + '__count'] = np.float64(float(metric.count))
        g[name + '__result'] = np.float64(float(metric.result()))

    for d in (1, 5):
        x1, x2 = arr(d), arr(d)
        g['mean_d%d__x1' % d], g['mean_d%d__x2' % d] = x1, x2
        run('mean_d%d' % d, metrics.Mean(), [(x1, sl), (x2, None)])
        t1, p1, t2, p2 = arr(d), arr(d), arr(d), arr(d)
        for k, v in (('t1', t1), ('p1', p1), ('t2', t2), ('p2', p2)):
            g['pair_d%d__%s' % (d, k)] = v
        run('rmse_d%d' % d, metrics.RMSE(), [(t1, p1, sl), (t2, p2, None)])
        run('mae_d%d' % d, metrics.MAE(), [(t1, p1, sl), (t2, p2, None)])
        run('distortion_d%d' % d, metrics.Distortion(), [(t1, p1, sl), (t2, p2, None)])
    # mel-cepstral distortion ignores c0 (60-dim static mcep of the shipped model)
    t1, p1 = arr(60, 0.3), arr(60, 0.3)
    g['mcd__t1'], g['mcd__p1'] = t1, p1
    run('mcd', metrics.MelCepDistortion(), [(t1, p1, sl), (t1[:, ::-1].copy(), p1, None)])
    # LF0 distortion in Hz on voiced frames only
    lt, lp = arr(1, 0.2, 5.0), arr(1, 0.2, 5.0)
    voiced = rng.rand(b, t, 1) > 0.4
    g['lf0__t'], g['lf0__p'], g['lf0__voiced'] = lt, lp, voiced
    run('lf0', metrics.LF0Distortion(), [(lt, lp, torch.from_numpy(voiced), sl), (lp, lt, torch.from_numpy(~voiced), None)])
    run('f0', metrics.F0Distortion(), [(np.exp(lt), np.exp(lp), torch.from_numpy(voiced), sl)])
    # V/UV accuracy as the model accumulates it: Mean of a float comparison
    vuv_t, vuv_p = rng.rand(b, t, 1) > 0.5, rng.rand(b, t, 1) > 0.5
    g['vuv__t'], g['vuv__p'] = vuv_t, vuv_p
    run('vuv_acc', metrics.Mean(), [((vuv_t == vuv_p).astype(np.float32), sl)])
    np.savez_compressed(os.path.join(HERE, 'g15_metrics.npz'), **g)
    print('g15_metrics.npz', os.path.getsize(os.path.join(HERE, 'g15_metrics.npz')), 'bytes')


if __name__ == '__main__' and (len(sys.argv) == 1 or sys.argv[1] == 'g15'):
    g15_metrics()


def g16_collate():
    """G16: the batches the reference's own ``FilesDataset.__getitem__`` (normalise on load, data.py:106-154) + ``collate_fn``
    (zero-pad, data.py:159-224) yield for the G8 data set written to disk, in G8's recorded first-epoch batch order, with the
    normaliser parameters loaded through ``_FeatureNormaliser.load_params`` from the JSON files.  Stored: every tensor of every
    batch (padded raw and ``normalised_`` features, int64 durations and frame counts) and the name lists."""
    import json
    import tempfile
    utils, losses, data, base_models, lr_schedules, metrics = import_reference()
    file_io = sys.modules['tts_data_tools.file_io']
    file_io.load_json = lambda path: json.load(open(path))
    g8 = dict(np.load(os.path.join(HERE, 'g8_plumbing.npz'), allow_pickle=False))
    names = [str(n) for n in g8['names']]
    root = tempfile.mkdtemp(prefix='morgana_g16_')
    for feat in ('lab', 'dur', 'lf0', 'n_frames'):
        os.makedirs(os.path.join(root, 'train', feat), exist_ok=True)
    for name in names:
        for feat in ('lab', 'dur', 'lf0'):
            np.save(os.path.join(root, 'train', feat, name + '.npy'), g8['data__%s__%s' % (name, feat)])
        open(os.path.join(root, 'train', 'n_frames', name + '.txt'), 'w').write(str(int(g8['data__%s__dur' % name].sum())))
    open(os.path.join(root, 'train_file_id_list.scp'), 'w').write('\n'.join(names) + '\n')
    os.makedirs(os.path.join(root, 'processed'), exist_ok=True)
    for key in ('lab_minmax', 'lf0_mvn'):
        params = {k.split('__')[2]: g8[k].tolist() for k in g8 if k.startswith('norm__%s__' % key)}
        json.dump(params, open(os.path.join(root, 'processed', key + '.json'), 'w'))

    class NpySource(object):                       # satisfies what FilesDataset calls (data.py:93,135,142)
        def __init__(self, name, use_deltas=False, as_int=False):
            self.name, self.use_deltas, self.as_int = name, use_deltas, as_int

        def __call__(self, base_name, data_dir):
            if self.as_int:
                return {self.name: int(open(os.path.join(data_dir, self.name, base_name + '.txt')).read())}
            return {self.name: np.load(os.path.join(data_dir, self.name, base_name + '.npy'))}

    normalisers = data.Normalisers({'lab': data.MinMaxNormaliser('lab'), 'lf0': data.MeanVarianceNormaliser('lf0')},
                                   normalisation_dir='processed', data_root=root, device='cpu')
    sources = {'n_frames': NpySource('n_frames', as_int=True), 'dur': NpySource('dur'), 'lab': NpySource('lab'),
               'lf0': NpySource('lf0')}
    dataset = data.FilesDataset(sources, 'train', 'train_file_id_list.scp', normalisers, data_root=root)
    out = {'file_ids': np.array(dataset.file_ids)}
    order = [b.split(',') for b in g8['batch_order'][0]]
    out['n_batches'] = np.int64(len(order))
    for i, batch_names in enumerate(order):
        items = [dataset[dataset.file_ids.index(n)] for n in batch_names]
        batch = data.FilesDataset.collate_fn(items)
        for key, value in batch.items():
            if hasattr(value, 'numpy'):
                out['batch%d__%s' % (i, key)] = value.numpy()
            else:
                out['batch%d__%s' % (i, key)] = np.array(value)
    np.savez_compressed(os.path.join(HERE, 'g16_collate.npz'), **out)
    print('g16_collate.npz', os.path.getsize(os.path.join(HERE, 'g16_collate.npz')), 'bytes;', sorted(out)[:12])


if __name__ == '__main__' and (len(sys.argv) == 1 or sys.argv[1] == 'g16'):
    g16_collate()
