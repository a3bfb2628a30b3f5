"""Pins the oracle (oracle/ref_cpu.py, oracle/ref_torch.py, oracle/oracle_c.c) against the golden vectors that
tests/golden/make_golden.py produced by running the reference itself.  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from morgana_amd import synthetic
from oracle import ref_cpu, ref_torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-4   # north star: 1e-4 relative fp32


def _c_oracle():
    path = os.path.join(REPO, 'oracle', 'liboracle_c.so')
    if not os.path.exists(path):
        subprocess.check_call(['make', '-C', os.path.join(REPO, 'oracle')])
    lib = ctypes.CDLL(path)
    lib.oracle_upsample_index.restype = ctypes.c_int64
    lib.oracle_masked_mse.restype = ctypes.c_float
    return lib


def test_g1_index_map_bit_exact(golden):
    g = golden('g1_upsample_index.npz')
    lib = _c_oracle()
    names = sorted(k[:-5] for k in g if k.endswith('__dur'))
    assert len(names) >= 7
    for name in names:
        dur, want = g[name + '__dur'], g[name + '__idx']
        idx, lens = ref_cpu.upsample_index(dur)
        assert idx.dtype == np.int64 and np.array_equal(idx, want), name
        assert np.array_equal(lens, dur.sum(axis=1))
        # (B, P, 1) durations give the same map (utils.py:202)
        assert np.array_equal(ref_cpu.upsample_index(dur[:, :, None])[0], want)
        # C leg
        b, p = dur.shape
        dur_c = np.ascontiguousarray(dur)
        tmax = lib.oracle_upsample_index(dur_c.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(b), ctypes.c_int64(p),
                                         ctypes.c_int64(0), None, None)
        assert tmax == want.shape[1]
        out = np.empty((b, tmax), dtype=np.int64)
        nfr = np.empty((b,), dtype=np.int64)
        lib.oracle_upsample_index(dur_c.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(b), ctypes.c_int64(p),
                                  ctypes.c_int64(tmax), out.ctypes.data_as(ctypes.c_void_p),
                                  nfr.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(out, want), name
        assert np.array_equal(nfr, lens)


def test_g1_float_durations_raise():
    with pytest.raises(TypeError):
        ref_cpu.upsample_index(np.ones((2, 3), dtype=np.float32))


def test_g2_upsample_values_and_backward(golden):
    g = golden('g2_upsample_values.npz')
    out = ref_cpu.upsample_to_repetitions(g['x'], g['dur'][:, :, None])
    assert np.array_equal(out, g['out'])
    assert np.array_equal(ref_cpu.upsample_to_repetitions(g['x'], g['dur']), g['out_2d_dur'])
    gx = ref_cpu.upsample_backward(g['grad_out'], g['dur'], g['x'].shape[1])
    np.testing.assert_allclose(gx, g['grad_x'], rtol=1e-6, atol=1e-6)
    # torch leg
    out_t = ref_torch.upsample_to_repetitions(torch.from_numpy(g['x']), torch.from_numpy(g['dur'])[:, :, None])
    assert np.array_equal(out_t.numpy(), g['out'])


def test_g3_sequence_mask(golden):
    g = golden('g3_sequence_mask.npz')
    sl = g['seq_len']
    m = ref_cpu.sequence_mask(sl)
    assert m.dtype == np.uint8 and np.array_equal(m, g['mask_default'])
    assert np.array_equal(ref_cpu.sequence_mask(sl, 9, np.float32), g['mask_float32_len9'])
    assert np.array_equal(ref_cpu.sequence_mask(sl, 3, np.int64), g['mask_long_len3'])
    assert np.array_equal(ref_torch.sequence_mask(torch.from_numpy(sl)).numpy(), g['mask_default'])


@pytest.mark.parametrize('dim', [1, 80, 187])
def test_g4_masked_mse(golden, dim):
    g = golden('g4_masked_mse.npz')
    p, y, sl = g['d%d__pred' % dim], g['d%d__target' % dim], g['d%d__seq_len' % dim]
    np.testing.assert_allclose(ref_cpu.mse(p, y, sl), g['d%d__loss' % dim], rtol=1e-5)
    np.testing.assert_allclose(ref_cpu.mse_grad(p, y, sl), g['d%d__grad' % dim], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(ref_cpu.mse(p, y), g['d%d__loss_nolen' % dim], rtol=1e-5)
    np.testing.assert_allclose(ref_cpu.mse_grad(p, y), g['d%d__grad_nolen' % dim], rtol=1e-5, atol=1e-9)
    lib = _c_oracle()
    grad = np.empty_like(p)
    b, t, d = p.shape
    loss = lib.oracle_masked_mse(p.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p),
                                 sl.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(b), ctypes.c_int64(t),
                                 ctypes.c_int64(d), grad.ctypes.data_as(ctypes.c_void_p))
    np.testing.assert_allclose(loss, g['d%d__loss' % dim], rtol=1e-5)
    np.testing.assert_allclose(grad, g['d%d__grad' % dim], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('dim', [1, 3])
def test_g11_bce(golden, dim):
    g = golden('g11_bce.npz')
    p, y, sl = g['d%d__pred' % dim], g['d%d__target' % dim], g['d%d__seq_len' % dim]
    np.testing.assert_allclose(ref_cpu.bce(p, y, sl), g['d%d__loss' % dim], rtol=1e-5)
    np.testing.assert_allclose(ref_cpu.bce_grad(p, y, sl), g['d%d__grad' % dim], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(ref_cpu.bce(p, y), g['d%d__loss_nolen' % dim], rtol=1e-5)
    np.testing.assert_allclose(ref_cpu.bce_grad(p, y), g['d%d__grad_nolen' % dim], rtol=1e-5, atol=1e-9)


def test_g4_zero_length_is_nan(golden):
    g = golden('g4_masked_mse.npz')
    assert np.isnan(g['zero_len_loss'])
    p = np.zeros((2, 3, 1), dtype=np.float32)
    assert np.isnan(ref_cpu.mse(p, p + 1, np.array([0, 2])))


def test_g5_normalisers(golden):
    g = golden('g5_normalisers.npz')
    f = g['feat']
    with np.errstate(all='ignore'):
        for kind in ('np', 'torch'):
            np.testing.assert_allclose(ref_cpu.normalise_mvn(f, g['mean'], g['std']), g['mvn_norm_' + kind], rtol=1e-6)
            np.testing.assert_allclose(ref_cpu.denormalise_mvn(f, g['mean'], g['std']), g['mvn_denorm_' + kind],
                                       rtol=1e-6)
            np.testing.assert_allclose(ref_cpu.normalise_minmax(f, g['mmin'], g['mmax']), g['minmax_norm_' + kind],
                                       rtol=1e-6)
            np.testing.assert_allclose(ref_cpu.denormalise_minmax(f, g['mmin'], g['mmax']),
                                       g['minmax_denorm_' + kind], rtol=1e-6)
    # std == 0 column divides by the 1e-8 epsilon; max == min column uses scale 1
    assert np.all(np.abs(g['mvn_norm_np'][..., 2]) > 1e5)
    np.testing.assert_allclose(g['minmax_norm_np'][..., 4], f[..., 4] - g['mmin'][4], rtol=1e-6)


def _c1_batches():
    return [synthetic.make_batch(8, 200, seed=synthetic.REFERENCE_SEED + 100 * i) for i in range(4)]


def test_g6_f0_model_numpy_oracle(golden):
    g = golden('g6_f0_model.npz')
    state = synthetic.f0_model_state()
    batches = _c1_batches()
    loss, pred, grads = ref_cpu.f0_forward_backward(state, batches[0])
    np.testing.assert_allclose(loss, g['loss_curve'][0], rtol=RTOL)
    np.testing.assert_allclose(pred[:, ::25, 0], g['step1_pred_sample'], rtol=RTOL, atol=1e-6)
    for key in grads:
        flat = grads[key].ravel()
        np.testing.assert_allclose(np.sqrt((flat.astype(np.float64) ** 2).sum()), g['step1_gradnorm__' + key],
                                   rtol=RTOL)
        want = g['step1_gradval__' + key]
        np.testing.assert_allclose(flat[g['step1_gradidx__' + key]], want, rtol=1e-3,
                                   atol=1e-4 * np.abs(want).max())
    curve = ref_cpu.f0_train(state, batches, 20, lr=0.01)
    np.testing.assert_allclose(curve, g['loss_curve'], rtol=RTOL)
    for key in state:
        np.testing.assert_allclose(state[key].astype(np.float64).sum(), g['final_sum__' + key], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(np.abs(state[key].astype(np.float64)).sum(), g['final_abs_sum__' + key], rtol=1e-3)


def test_g6_f0_model_torch_oracle(golden):
    g = golden('g6_f0_model.npz')
    torch.set_num_threads(4)
    model = ref_torch.load_state(ref_torch.F0Model(), synthetic.f0_model_state())
    batches = [ref_torch.to_torch(b) for b in _c1_batches()]
    curve = ref_torch.train_steps(model, batches, 20, lr=0.01)
    np.testing.assert_allclose(curve, g['loss_curve'], rtol=1e-5)
    model = ref_torch.load_state(ref_torch.F0Model(), synthetic.f0_model_state())
    ragged = ref_torch.to_torch(synthetic.make_batch(6, (40, 120), seed=77))
    curve = ref_torch.train_steps(model, [ragged], 5, lr=0.005, weight_decay=1e-3)
    np.testing.assert_allclose(curve, g['ragged_loss_curve'], rtol=1e-5)


def test_g6_ragged_weight_decay_numpy_oracle(golden):
    g = golden('g6_f0_model.npz')
    state = synthetic.f0_model_state()
    ragged = synthetic.make_batch(6, (40, 120), seed=77)
    curve = ref_cpu.f0_train(state, [ragged], 5, lr=0.005, weight_decay=1e-3)
    np.testing.assert_allclose(curve, g['ragged_loss_curve'], rtol=RTOL)
    _, pred, _ = ref_cpu.f0_forward_backward(state, ragged)   # after 5 updates == prediction of step 6; only shape
    assert pred[:, ::7, 0].shape == g['ragged_last_pred_sample'].shape


@pytest.mark.parametrize('tag', ['h8', 'h32'])
def test_g7_gru_wrapper(golden, tag):
    g = golden('g7_gru.npz')
    hid = int(tag[1:])
    i_dim = g[tag + '__x'].shape[2]
    w_ih, w_hh, b_ih, b_hh = synthetic.init_gru(np.random.RandomState(5 + hid), i_dim, hid)
    x, sl = g[tag + '__x'], g[tag + '__seq_len']
    out, hn, cache = ref_cpu.gru_forward(x, sl, w_ih, w_hh, b_ih, b_hh)
    assert out.shape == g[tag + '__out'].shape            # time cropped to max(seq_len)
    np.testing.assert_allclose(out, g[tag + '__out'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(hn, g[tag + '__hn'], rtol=RTOL, atol=1e-6)
    for b, n in enumerate(sl):
        assert np.all(out[b, n:] == 0)
    gr = ref_cpu.gru_backward(g[tag + '__grad_out'], None, x, sl, w_ih, w_hh, cache)
    np.testing.assert_allclose(gr['x'], g[tag + '__grad_x'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(gr['w_ih'], g[tag + '__grad_weight_ih_l0'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(gr['w_hh'], g[tag + '__grad_weight_hh_l0'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(gr['b_ih'], g[tag + '__grad_bias_ih_l0'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(gr['b_hh'], g[tag + '__grad_bias_hh_l0'], rtol=1e-3, atol=1e-5)
    # initial hidden + gradient on the final hidden
    out, hn, cache = ref_cpu.gru_forward(x, sl, w_ih, w_hh, b_ih, b_hh, h0=g[tag + '__h0'])
    np.testing.assert_allclose(out, g[tag + '__out_h0'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(hn, g[tag + '__hn_h0'], rtol=RTOL, atol=1e-6)
    gr = ref_cpu.gru_backward(g[tag + '__grad_out'], g[tag + '__grad_hn'], x, sl, w_ih, w_hh, cache)
    np.testing.assert_allclose(gr['x'], g[tag + '__grad_x_h0'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(gr['h0'], g[tag + '__grad_h0'], rtol=1e-3, atol=1e-5)


def test_g7_rnn_model(golden):
    g = golden('g7_gru.npz')
    lab_dim, hidden, post, out_dim = [int(v) for v in g['rnn__dims']]
    state = synthetic.rnn_spss_state(seed=31, lab_dim=lab_dim, hidden=hidden, post=post, out_dim=out_dim)
    feats = synthetic.make_batch(6, (20, 60), lab_dim=lab_dim, out_dim=out_dim, target_name='mcep',
                                 frames_per_phone=6.0, seed=99)
    loss, pred, grads = ref_cpu.rnn_forward_backward(state, feats)
    np.testing.assert_allclose(loss, g['rnn__loss_curve'][0], rtol=RTOL)
    np.testing.assert_allclose(pred, g['rnn__step1_pred'], rtol=1e-3, atol=1e-5)
    for key in ref_cpu.RNN_KEYS:
        want = g['rnn__step1_grad__' + key]
        np.testing.assert_allclose(grads[key], want, rtol=1e-3, atol=1e-4 * np.abs(want).max())
    # torch leg: 8-step curve
    model = ref_torch.load_state(ref_torch.RNNModel(lab_dim, hidden, post, out_dim), state)
    curve = ref_torch.train_steps(model, [ref_torch.to_torch(feats)], 8, lr=0.01)
    np.testing.assert_allclose(curve, g['rnn__loss_curve'], rtol=1e-5)


def test_g9_ema_lr_mean(golden):
    g = golden('g9_ema_lr.npz')
    shadow = g['ema_shadow0'].copy()
    for params in g['ema_params_seq']:
        ref_cpu.ema_update(shadow, params, float(g['ema_decay']))
    np.testing.assert_allclose(shadow, g['ema_shadow_final'], rtol=1e-6)
    np.testing.assert_allclose([ref_cpu.noam_scale(s, 4) for s in range(12)], g['noam_w4'], rtol=1e-12)
    np.testing.assert_allclose([ref_cpu.cyclic_noam_scale(s, 4, cycle_steps=9) for s in range(24)],
                               g['cyclic_noam_w4_c9'], rtol=1e-12)
    np.testing.assert_allclose([ref_cpu.cyclic_noam_scale(s, 4, cycle_trigger=0.5) for s in range(30)],
                               g['cyclic_noam_w4_trig'], rtol=1e-12)
    assert np.all(g['constant'] == 1.0)
    np.testing.assert_allclose(ref_cpu.metric_mean(g['mean_inputs']), g['mean_result'], rtol=1e-5)


def _lstm_layers(g, tag, layers):
    return [[g['%s__param__%s_l%d' % (tag, name, k)] for name in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
            for k in range(layers)]


@pytest.mark.parametrize('tag,layers', [('l1', 1), ('l2', 2)])
def test_g10_lstm_wrapper(golden, tag, layers):
    """Oracle LSTM (forward + BPTT, chained for multi-layer) against the reference's RecurrentCuDNNWrapper(nn.LSTM)."""
    g = golden('g10_lstm.npz')
    params = _lstm_layers(g, tag, layers)
    x, sl = g[tag + '__x'], g[tag + '__seq_len']
    h0, c0 = g[tag + '__h0'], g[tag + '__c0']
    inp, caches, inputs, hns, cns = x, [], [], [], []
    for k in range(layers):
        inputs.append(inp)
        inp, hn, cn, cache = ref_cpu.lstm_forward(inp, sl, *params[k], h0=h0[k], c0=c0[k])
        caches.append(cache)
        hns.append(hn)
        cns.append(cn)
    np.testing.assert_allclose(inp, g[tag + '__out_s'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(np.concatenate(hns), g[tag + '__hn_s'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(np.concatenate(cns), g[tag + '__cn_s'], rtol=RTOL, atol=1e-6)
    grad = g[tag + '__grad_out']
    for k in range(layers - 1, -1, -1):
        gr = ref_cpu.lstm_backward(grad, g[tag + '__grad_hn'][k], g[tag + '__grad_cn'][k], inputs[k], params[k][0],
                                   params[k][1], caches[k])
        np.testing.assert_allclose(gr['h0'][0], g[tag + '__grad_h0'][k], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(gr['c0'][0], g[tag + '__grad_c0'][k], rtol=1e-3, atol=1e-5)
        if k == 0:
            np.testing.assert_allclose(gr['w_hh'], g[tag + '__grad_whh0_s'], rtol=1e-3, atol=1e-5)
        grad = gr['x']
    np.testing.assert_allclose(grad, g[tag + '__grad_x_s'], rtol=1e-3, atol=1e-5)
    # without initial state: outputs only
    inp = x
    for k in range(layers):
        inp, hn, cn, _ = ref_cpu.lstm_forward(inp, sl, *params[k])
    assert inp.shape == g[tag + '__out'].shape
    np.testing.assert_allclose(inp, g[tag + '__out'], rtol=RTOL, atol=1e-6)


G12_STREAMS = (('lf0', 3, 'mse'), ('vuv', 1, 'sigmoid_bce'), ('mcep', 6, 'mse'), ('bap', 3, 'mse'))


def _g12_features():
    return synthetic.make_acoustic_batch(5, (10, 30), lab_dim=20, counters_dim=4, streams=G12_STREAMS, frames_per_phone=5.0,
                                         seed=1212)


def test_g12_multi_stream_loss(golden):
    """models/RNN_SPSS.py:120-139 (3 x mse + bce(sigmoid) over 4) on random predictions with saturated logits."""
    g = golden('g12_lstm_acoustic.npz')
    feats = _g12_features()
    targets = [feats['vuv'] if kind != 'mse' else feats['normalised_%s_deltas' % name] for name, _, kind in G12_STREAMS]
    loss, grad = ref_cpu.multi_stream_loss(g['loss__pred'], targets, [k for _, _, k in G12_STREAMS], feats['n_frames'])
    np.testing.assert_allclose(loss, g['loss__value'], rtol=1e-5)
    np.testing.assert_allclose(grad, g['loss__grad'], rtol=1e-4, atol=1e-8)


def test_g12_lstm_acoustic_model(golden):
    g = golden('g12_lstm_acoustic.npz')
    lab_dim, counters_dim, hidden, post, num_layers = [int(v) for v in g['model__dims'][:5]]
    feats = _g12_features()
    state = synthetic.lstm_acoustic_state(seed=1213, input_dim=lab_dim + counters_dim, hidden=hidden, post=post,
                                          output_dim=13, num_layers=num_layers)
    loss, pred, grads = ref_cpu.lstm_acoustic_forward_backward(state, feats, G12_STREAMS, num_layers)
    np.testing.assert_allclose(loss, g['model__loss_curve'][0], rtol=1e-5)
    np.testing.assert_allclose(pred[..., 0:3], g['model__step1_normalised_lf0_deltas'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ref_cpu.sigmoid(pred[..., 3:4]), g['model__step1_vuv'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(pred[..., 4:10], g['model__step1_normalised_mcep_deltas'], rtol=1e-4, atol=1e-6)
    for key in ref_cpu.lstm_acoustic_keys(num_layers):
        ref = g['model__step1_grad__' + key]
        np.testing.assert_allclose(grads[key], ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max(), err_msg=key)
    curve = ref_cpu.lstm_acoustic_train(state, feats, G12_STREAMS, num_layers, 6, lr=0.01)
    np.testing.assert_allclose(curve, g['model__loss_curve'], rtol=1e-4)


def test_g13_gru_f0_model(golden):
    """The shipped F0 model layout (models/f0_test_model.py:28-45: three GRU wrappers, counters concat) at toy size."""
    g = golden('g13_gru_f0.npz')
    lab_dim, counters_dim, d1, hid, post, out_dim = [int(v) for v in g['dims']]
    feats = synthetic.make_acoustic_batch(5, (10, 30), lab_dim=lab_dim, counters_dim=counters_dim,
                                          streams=(('lf0', out_dim, 'mse'),), frames_per_phone=5.0, seed=1313)
    state = synthetic.gru_f0_state(seed=1314, input_dim=lab_dim + counters_dim, d1=d1, hidden=hid, post=post, output_dim=out_dim)
    loss, pred, grads = ref_cpu.gru_f0_forward_backward(state, feats)
    np.testing.assert_allclose(loss, g['loss_curve'][0], rtol=1e-5)
    np.testing.assert_allclose(pred, g['step1_pred'], rtol=1e-4, atol=1e-6)
    for key in ref_cpu.GRU_F0_KEYS:
        ref = g['step1_grad__' + key]
        np.testing.assert_allclose(grads[key], ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max(), err_msg=key)
    np.testing.assert_allclose(ref_cpu.gru_f0_train(state, feats, 6, lr=0.01), g['loss_curve'], rtol=1e-4)


@pytest.mark.parametrize('tag', ['small', 'wide'])
def test_g14_segment_ops(golden, tag):
    """split_to_segments / get_segment_ends (utils.py:231-330), incl. zero-length segments."""
    g = golden('g14_segments.npz')
    x, lens = g[tag + '__x'], g[tag + '__lens']
    assert np.array_equal(ref_cpu.split_to_segments(x, lens), g[tag + '__split'])
    assert np.array_equal(ref_cpu.get_segment_ends(x, lens), g[tag + '__ends'])


METRIC_CASES = [('mean_d1', 'mean', None), ('mean_d5', 'mean', None), ('rmse_d1', 'sqdiff', None), ('rmse_d5', 'sqdiff', None),
                ('mae_d1', 'absdiff', None), ('mae_d5', 'absdiff', None), ('distortion_d1', 'root_sq', None),
                ('distortion_d5', 'root_sq', None), ('mcd', 'sqdiff', None), ('lf0', 'sqdiff_voiced_exp', None),
                ('f0', 'sqdiff_voiced', None), ('vuv_acc', 'mean', None)]


def metric_calls(g, name):
    """The accumulate calls tests/golden/make_golden.py:g15_metrics made for `name`: [(kind kwargs)]."""
    sl = g['seq_len']
    if name.startswith('mean_d'):
        d = name[-1]
        return [dict(target=g['mean_d%s__x1' % d], seq_len=sl), dict(target=g['mean_d%s__x2' % d])]
    if name[:-3] in ('rmse', 'mae', 'distortion'):
        d = name[-1]
        return [dict(target=g['pair_d%s__t1' % d], pred=g['pair_d%s__p1' % d], seq_len=sl),
                dict(target=g['pair_d%s__t2' % d], pred=g['pair_d%s__p2' % d])]
    if name == 'mcd':
        return [dict(target=g['mcd__t1'], pred=g['mcd__p1'], seq_len=sl, col0=1),
                dict(target=g['mcd__t1'][:, ::-1].copy(), pred=g['mcd__p1'], col0=1)]
    if name == 'lf0':
        return [dict(target=g['lf0__t'], pred=g['lf0__p'], voiced=g['lf0__voiced'], seq_len=sl),
                dict(target=g['lf0__p'], pred=g['lf0__t'], voiced=~g['lf0__voiced'])]
    if name == 'f0':
        return [dict(target=np.exp(g['lf0__t']), pred=np.exp(g['lf0__p']), voiced=g['lf0__voiced'], seq_len=sl)]
    if name == 'vuv_acc':
        return [dict(target=(g['vuv__t'] == g['vuv__p']).astype(np.float32), seq_len=sl)]
    raise KeyError(name)


@pytest.mark.parametrize('name,kind,_', METRIC_CASES)
def test_g15_streaming_metrics(golden, name, kind, _):
    """oracle.ref_cpu.metric_sums / metric_result against the reference's streaming metrics (sum, count, result)."""
    g = golden('g15_metrics.npz')
    total = count = 0.0
    for call in metric_calls(g, name):
        s, c = ref_cpu.metric_sums(kind, **call)
        total, count = total + s, count + c
    np.testing.assert_allclose(count, float(g[name + '__count']), rtol=0, atol=0)
    np.testing.assert_allclose(total, float(g[name + '__sum']), rtol=2e-5)
    np.testing.assert_allclose(ref_cpu.metric_result(kind, total, count), float(g[name + '__result']), rtol=2e-5)


# ------------------------------------------------------------------------------------------------------------ MLPG
# The reference's MLPG needs `bandmat`, which this image lacks: no golden vector exists (oracle header: parity unpinned by
# reference execution).  The restatement is pinned against the definition written out densely and against cases with known
# answers.
WINDOWS_5PT = ((0, 0, (1.0,)), (2, 2, (-0.2, -0.1, 0.0, 0.1, 0.2)), (1, 1, (1.0, -2.0, 1.0)))


def mlpg_dense(mu, var, windows, pad):
    """morgana/viz/synthesis.py:39-77, :156-171 for one feature dimension with explicit N x N window matrices."""
    edge = lambda x: np.concatenate((np.repeat(x[[0]], pad, 0), x, np.repeat(x[[-1]], pad, 0)), 0)
    mu, var = edge(np.asarray(mu, np.float64)), edge(np.asarray(var, np.float64))
    n = mu.shape[0]
    prec, b = np.zeros((n, n)), np.zeros(n)
    for w, (l, u, c) in enumerate(windows):
        win = np.zeros((n, n))
        for s in range(n):
            for k in range(-l, u + 1):
                if 0 <= s + k < n:
                    win[s, s + k] = c[l + k]
        prec += win.T @ np.diag(1.0 / var[:, w]) @ win
        b += win.T @ (mu[:, w] / var[:, w])
    return np.linalg.solve(prec, b)[pad:n - pad]


@pytest.mark.parametrize('windows,pad', [(ref_cpu.MLPG_DEFAULT_WINDOWS, 0), (ref_cpu.MLPG_DEFAULT_WINDOWS, 7), (WINDOWS_5PT, 4)])
def test_mlpg_oracle_vs_dense_definition(windows, pad):
    rng = np.random.RandomState(len(windows[1][2]) + pad)
    b, t, d = 3, 26, 2
    means = rng.standard_normal((b, t, 3 * d))
    seq_len = [t, 11, 1]
    for variances in (rng.uniform(0.3, 2.0, 3 * d), rng.uniform(0.3, 2.0, (b, t, 3 * d))):
        got = ref_cpu.mlpg(means, variances, windows=windows, padding_size=pad, seq_len=seq_len)
        assert got.shape == (b, t, d)
        for i in range(b):
            n = seq_len[i]
            assert np.all(got[i, n:] == 0)
            for k in range(d):
                var = np.broadcast_to(variances[k::d], (n, 3)) if variances.ndim == 1 else variances[i, :n, k::d]
                np.testing.assert_allclose(got[i, :n, k], mlpg_dense(means[i, :n, k::d], var, windows, pad), rtol=1e-10, atol=1e-12)


def test_mlpg_oracle_known_answers():
    rng = np.random.RandomState(5)
    static = rng.standard_normal((1, 15, 1))
    # a static window alone: the trajectory is the means
    np.testing.assert_allclose(ref_cpu.mlpg(static, np.ones(1), windows=((0, 0, (1.0,)),)), static, rtol=1e-12)
    # three zero-extent windows with gains a_w: per frame x = sum_w a_w mu_w / var_w / sum_w a_w^2 / var_w
    means = rng.standard_normal((2, 15, 3))
    var, gain = np.array([1.0, 0.5, 2.0]), np.array([1.0, -2.0, 0.5])
    got = ref_cpu.mlpg(means, var, windows=tuple((0, 0, (a,)) for a in gain))
    np.testing.assert_allclose(got[..., 0], (means * gain / var).sum(-1) / (gain ** 2 / var).sum(), rtol=1e-12)
    # single sequence in, single sequence out
    assert ref_cpu.mlpg(means[0], np.array([1.0, 0.5, 0.5])).shape == (15, 1)
