"""GPU parity tests (run with ``-m gpu`` on an MI355X): every check goes through the C ABI of libmorgana_hip.so and is
compared with the oracle (oracle/ref_cpu.py) and/or the golden vectors generated from the reference.

Tolerances: integer index work is bit exact; fp32 mode is held to the north star's 1e-4 relative; bf16 throughput mode
is held to 2e-2 relative (bf16 has an 8-bit mantissa: 4e-3 per rounding), stated per test.
"""
import numpy as np
import pytest
import torch

from morgana_amd import data, losses, models, ops, optim, synthetic, utils
from morgana_amd import functional as F_hip
from oracle import ref_cpu

pytestmark = pytest.mark.gpu

RTOL = 1e-4           # fp32 parity bar (north star)
RTOL_BF16 = 2e-2      # bf16 throughput mode

DEV = 'cuda:0'


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t if dtype is None else t.to(dtype)


from parity_report import rel_err          # noqa: E402,F401  max |got - want| / max |want|, recorded per test (gpurun_out/parity_report.json)


# ------------------------------------------------------------------------------------------------------------ K1
def test_upsample_index_golden_bit_exact(golden):
    g = golden('g1_upsample_index.npz')
    for name in sorted(k[:-5] for k in g if k.endswith('__dur')):
        dur, want = g[name + '__dur'], g[name + '__idx']
        d = dev(dur)
        n_frames, tmax = ops.upsample_lengths(d)
        assert int(tmax.item()) == want.shape[1]
        assert np.array_equal(n_frames.cpu().numpy(), dur.sum(axis=1))
        idx64, rows = ops.upsample_index(d, want.shape[1], want_idx64=True)
        assert idx64.dtype == torch.int64 and np.array_equal(idx64.cpu().numpy(), want), name
        p = dur.shape[1]
        want_rows = np.where(want < 0, -1, want + np.arange(dur.shape[0])[:, None] * p)
        assert np.array_equal(rows.cpu().numpy(), want_rows)


@pytest.mark.parametrize('shape', [(256, 80, 1000), (64, 160, 2000), (3, 1, 5), (17, 300, 4097)])
def test_upsample_index_full_size_vs_oracle(shape):
    b, p, t = shape
    rng = np.random.RandomState(b + p)
    dur = np.zeros((b, p), dtype=np.int64)
    for i in range(b):
        n_ph = rng.randint(1, p + 1)
        total = rng.randint(n_ph, t + 1)
        dur[i, :n_ph] = synthetic._durations(rng, total, n_ph)
    if b > 2:
        dur[1] = 0                                     # an utterance with no frames at all
    want, lens = ref_cpu.upsample_index(dur)
    idx64, _ = ops.upsample_index(dev(dur), want.shape[1] + 3, want_idx64=True)     # t_cap beyond Tmax: -1 fill
    got = idx64.cpu().numpy()
    assert np.array_equal(got[:, :want.shape[1]], want)
    assert np.all(got[:, want.shape[1]:] == -1)


def test_upsample_values_and_backward(golden):
    g = golden('g2_upsample_values.npz')
    x = dev(g['x']).requires_grad_(True)
    out = utils.upsample_to_repetitions(x, dev(g['dur'])[:, :, None])
    assert np.array_equal(out.detach().cpu().numpy(), g['out'])           # a pure copy: bit exact
    out.backward(dev(g['grad_out']))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g['grad_x'], rtol=1e-6, atol=1e-6)
    out2 = utils.upsample_to_repetitions(x.detach(), dev(g['dur']))       # 2-D durations
    assert np.array_equal(out2.cpu().numpy(), g['out_2d_dur'])
    with pytest.raises(TypeError):
        utils.upsample_to_repetitions(x.detach(), dev(g['dur']).float())


def test_upsample_full_size_copy_and_segment_sum():
    feats = synthetic.make_batch(32, (300, 2000), lab_dim=600, seed=5)
    lab, dur = feats['normalised_lab'], feats['dur']
    want = ref_cpu.upsample_to_repetitions(lab, dur)
    x = dev(lab).requires_grad_(True)
    out = utils.upsample_to_repetitions(x, dev(dur))
    assert out.shape == want.shape
    assert np.array_equal(out.detach().cpu().numpy(), want)
    # adjoint property: <U x, g> == <x, U^T g>
    g = torch.randn(out.shape, device=out.device, generator=torch.Generator(device=out.device).manual_seed(7))
    out.backward(g)
    lhs = (out.detach().double() * g.double()).sum().item()
    rhs = (x.detach().double() * x.grad.double()).sum().item()
    scale = (out.detach().double() * g.double()).abs().sum().item()      # the two sums cancel to ~1e-5 of their terms
    assert abs(lhs - rhs) <= 1e-6 * scale
    # bf16 gather with zero padded leading dimension
    _, rows = ops.upsample_index(dev(dur[:, :, 0]), want.shape[1])
    bf = ops.gather_rows(dev(lab).view(-1, 600), rows.view(-1), out_bf16=True)
    assert bf.shape[1] == ops.pad_ld(600) and torch.all(bf[:, 600:] == 0)
    np.testing.assert_allclose(bf[:, :600].float().cpu().numpy().reshape(want.shape), want, rtol=8e-3, atol=1e-6)


@pytest.mark.parametrize('tag', ['small', 'wide'])
def test_segment_ops_golden(golden, tag):
    """split_to_segments / get_segment_ends (utils.py:231-330): values bit exact (pure copies), gradients bit exact."""
    g = golden('g14_segments.npz')
    x = dev(g[tag + '__x']).requires_grad_(True)
    lens = dev(g[tag + '__lens'])
    seg = utils.split_to_segments(x, lens)
    assert np.array_equal(seg.detach().cpu().numpy(), g[tag + '__split'])
    (seg * dev(g[tag + '__split_grad_out'])).sum().backward()
    assert np.array_equal(x.grad.cpu().numpy(), g[tag + '__split_grad_x'])
    x.grad = None
    ends = utils.get_segment_ends(x, lens)
    assert np.array_equal(ends.detach().cpu().numpy(), g[tag + '__ends'])
    (ends * dev(g[tag + '__ends_grad_out'])).sum().backward()
    assert np.array_equal(x.grad.cpu().numpy(), g[tag + '__ends_grad_x'])


def test_segment_ops_full_size_vs_oracle():
    """Phone segments of a C5-like ragged batch: frames -> (phones, frames per phone) and back (adjoint of the split)."""
    feats = synthetic.make_batch(16, (300, 2000), lab_dim=8, out_dim=80, target_name='mcep', seed=9)
    x, dur = feats['normalised_mcep'], feats['dur']
    want = ref_cpu.split_to_segments(x, dur)
    xt = dev(x).requires_grad_(True)
    seg = utils.split_to_segments(xt, dev(dur))
    assert np.array_equal(seg.detach().cpu().numpy(), want)
    assert np.array_equal(utils.get_segment_ends(xt, dev(dur)).detach().cpu().numpy(), ref_cpu.get_segment_ends(x, dur))
    seg.backward(seg.detach())                              # scatter of the split itself gives the valid frames back
    n = feats['n_frames']
    mask = (np.arange(x.shape[1])[None, :] < n[:, None])[..., None]
    assert np.array_equal(xt.grad.cpu().numpy(), x * mask)


def test_collate_to_device_matches_host_pipeline():
    """normalise-on-load + zero-padding collate + upload (data.py:119-127, 159-224, 648-663) against the device-side collate."""
    rng = np.random.RandomState(44)
    norms = {'lab': data.MinMaxNormaliser('lab').set_params({'mmin': rng.rand(20).astype(np.float32),
                                                             'mmax': (1.5 + rng.rand(20)).astype(np.float32)}, device=DEV),
             'lf0': data.MeanVarianceNormaliser('lf0').set_params({'mean': rng.randn(3).astype(np.float32),
                                                                   'std_dev': (0.5 + rng.rand(3)).astype(np.float32)}, device=DEV),
             'dur': data.MeanVarianceNormaliser('dur').set_params({'mean': np.array([7.0], np.float32),
                                                                   'std_dev': np.array([3.0], np.float32)}, device=DEV)}
    batch = []
    for i, (n_ph, n_fr) in enumerate([(5, 40), (9, 77), (1, 3), (7, 64)]):
        batch.append({'name': 'utt%d' % i, 'n_frames': n_fr, 'n_phones': n_ph,
                      'dur': rng.randint(1, 12, size=(n_ph, 1)).astype(np.int64),
                      'lab': (rng.rand(n_ph, 20) * 2).astype(np.float32),
                      'lf0': rng.randn(n_fr, 3).astype(np.float32),
                      'vuv': (rng.rand(n_fr, 1) > 0.5)})
    want = data.to_device(data.collate_fn([data.load_utterance(item, norms) for item in batch]), DEV)
    got = data.collate_to_device(batch, norms, DEV)
    assert sorted(got.keys()) == sorted(want.keys())
    for key, value in want.items():
        if isinstance(value, torch.Tensor):
            assert got[key].shape == value.shape and got[key].dtype == value.dtype, key
            if value.is_floating_point():
                np.testing.assert_allclose(got[key].cpu().numpy(), value.cpu().numpy(), rtol=1e-6, atol=1e-7, err_msg=key)
            else:
                assert torch.equal(got[key], value), key
        else:
            assert got[key] == value, key


# ------------------------------------------------------------------------------------------- mask / K4 / K5
def test_sequence_mask(golden):
    g = golden('g3_sequence_mask.npz')
    sl = dev(g['seq_len'])
    m = utils.sequence_mask(sl)
    assert m.dtype == torch.uint8 and np.array_equal(m.cpu().numpy(), g['mask_default'])
    m = utils.sequence_mask(sl, max_len=9, dtype=torch.float32)
    assert np.array_equal(m.cpu().numpy(), g['mask_float32_len9'])
    m = utils.sequence_mask(sl, max_len=3, dtype=torch.long)
    assert m.dtype == torch.int64 and np.array_equal(m.cpu().numpy(), g['mask_long_len3'])
    m = utils.sequence_mask(sl, dtype=torch.ByteTensor)                    # the reference's legacy type object
    assert np.array_equal(m.cpu().numpy(), g['mask_default'])


@pytest.mark.parametrize('dim', [1, 80, 187])
def test_masked_mse_golden(golden, dim):
    g = golden('g4_masked_mse.npz')
    p = dev(g['d%d__pred' % dim]).requires_grad_(True)
    y = dev(g['d%d__target' % dim])
    sl = dev(g['d%d__seq_len' % dim])
    loss = losses.mse(p, y, sl)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['d%d__loss' % dim], rtol=1e-5)
    np.testing.assert_allclose(p.grad.cpu().numpy(), g['d%d__grad' % dim], rtol=1e-5, atol=1e-9)
    p.grad = None
    loss = losses.mse(p, y)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['d%d__loss_nolen' % dim], rtol=1e-5)
    np.testing.assert_allclose(p.grad.cpu().numpy(), g['d%d__grad_nolen' % dim], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('dim', [1, 3])
def test_masked_bce_golden(golden, dim):
    """losses.bce (voicing stream): saturated probabilities included; tolerance 1e-5 relative fp32."""
    g = golden('g11_bce.npz')
    p = dev(g['d%d__pred' % dim]).requires_grad_(True)
    y = dev(g['d%d__target' % dim])
    sl = dev(g['d%d__seq_len' % dim])
    loss = losses.bce(p, y, sl)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['d%d__loss' % dim], rtol=1e-5)
    np.testing.assert_allclose(p.grad.cpu().numpy(), g['d%d__grad' % dim], rtol=1e-5, atol=1e-9)
    p.grad = None
    loss = losses.bce(p, y)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['d%d__loss_nolen' % dim], rtol=1e-5)
    np.testing.assert_allclose(p.grad.cpu().numpy(), g['d%d__grad_nolen' % dim], rtol=1e-5, atol=1e-9)


def test_masked_mse_zero_length_is_nan_and_full_size():
    p = torch.zeros(2, 3, 1, device=DEV)
    assert torch.isnan(losses.mse(p, p + 1, torch.tensor([0, 2], device=DEV)))
    rng = np.random.RandomState(3)
    for (b, t, d) in [(256, 1000, 1), (64, 1000, 80), (64, 1999, 187)]:
        pred = rng.standard_normal((b, t, d)).astype(np.float32)
        tgt = rng.standard_normal((b, t, d)).astype(np.float32)
        sl = rng.randint(1, t + 1, size=b).astype(np.int64)
        sl[0] = t
        pt = dev(pred).requires_grad_(True)
        loss = losses.mse(pt, dev(tgt), dev(sl))
        loss.backward()
        np.testing.assert_allclose(loss.item(), ref_cpu.mse(pred, tgt, sl), rtol=1e-5)
        np.testing.assert_allclose(pt.grad.cpu().numpy(), ref_cpu.mse_grad(pred, tgt, sl), rtol=1e-5, atol=1e-12)
        # deterministic: two runs give identical bits
        loss2 = losses.mse(pt.detach(), dev(tgt), dev(sl))
        assert loss2.item() == loss.item()


def test_normalisers(golden):
    g = golden('g5_normalisers.npz')
    f = dev(g['feat'])
    mvn = data.MeanVarianceNormaliser('lf0').set_params({'mean': g['mean'], 'std_dev': g['std']}, device=DEV)
    mm = data.MinMaxNormaliser('lab').set_params({'mmin': g['mmin'], 'mmax': g['mmax']}, device=DEV)
    np.testing.assert_allclose(mvn.normalise(f).cpu().numpy(), g['mvn_norm_torch'], rtol=1e-6)
    np.testing.assert_allclose(mvn.denormalise(f).cpu().numpy(), g['mvn_denorm_torch'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(mm.normalise(f).cpu().numpy(), g['minmax_norm_torch'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(mm.denormalise(f).cpu().numpy(), g['minmax_denorm_torch'], rtol=1e-6, atol=1e-7)
    # NumPy inputs take the host path (loader side), as in the reference
    with np.errstate(all='ignore'):
        np.testing.assert_allclose(mvn.normalise(g['feat']), g['mvn_norm_np'], rtol=1e-6)
        np.testing.assert_allclose(mm.normalise(g['feat']), g['minmax_norm_np'], rtol=1e-6)
    # round trip on a full-size lab tensor (size independent property)
    x = torch.rand(256, 80, 600, device=DEV)
    mmin = torch.rand(600) - 0.5
    mmax = mmin + torch.rand(600) + 0.1
    mm2 = data.MinMaxNormaliser('lab').set_params({'mmin': mmin.numpy(), 'mmax': mmax.numpy()}, device=DEV)
    back = mm2.denormalise(mm2.normalise(x))
    assert (back - x).abs().max().item() < 1e-5


# ------------------------------------------------------------------------------------------------------------ K2
LINEAR_SHAPES = [(1000, 600, 512), (777, 609, 256), (300, 512, 128), (513, 128, 32), (1600, 32, 1), (130, 64, 3),
                 (64, 40, 96),
                 # shapes that take the large-tile LDS-DMA kernels (M >= 2048 / 4096, N % 128 == 0), with ragged M tails
                 (4999, 600, 512), (4500, 512, 128), (4100, 128, 512), (6001, 200, 256), (5003, 600, 384),
                 (8200, 512, 1536),
                 # half-width weight-gradient tiles (k halves) without a padding tile for the bias sums (K > 608) and with all 640 columns
                 (4800, 620, 512), (4321, 640, 512), (21504, 600, 512), (21504, 512, 128)]


@pytest.mark.parametrize('shape', LINEAR_SHAPES)
@pytest.mark.parametrize('gather', [False, True])
def test_linear_kernels_fp32(shape, gather):
    m, k, n = shape
    rng = np.random.RandomState(m + k + n)
    w = rng.uniform(-0.1, 0.1, (n, k)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, (n,)).astype(np.float32)
    if gather:
        table = rng.uniform(0, 1, (max(m // 7, 2), k)).astype(np.float32)
        rows = rng.randint(-1, table.shape[0], size=m).astype(np.int32)
        a = np.where(rows[:, None] < 0, 0, table[np.maximum(rows, 0)]).astype(np.float32)
        a_dev, rows_dev = dev(table), dev(rows)
    else:
        a = rng.uniform(0, 1, (m, k)).astype(np.float32)
        a_dev, rows_dev = dev(a), None
    for act in (ops.ACT_NONE, ops.ACT_SIGMOID):
        y = ops.linear_fwd_f32(a_dev, rows_dev, m, dev(w), dev(b), act).cpu().numpy()
        want = a.astype(np.float64) @ w.T.astype(np.float64) + b
        if act:
            want = 1 / (1 + np.exp(-want))
        assert rel_err(y, want) < 1e-5, (shape, act)
    dy = rng.standard_normal((m, n)).astype(np.float32)
    dw, db = ops.linear_wgrad_f32(dev(dy), a_dev, rows_dev, n, k)
    assert rel_err(dw.cpu().numpy(), dy.T.astype(np.float64) @ a.astype(np.float64)) < 1e-5
    assert rel_err(db.cpu().numpy(), dy.astype(np.float64).sum(axis=0)) < 1e-5
    h = rng.uniform(0.05, 0.95, (m, k)).astype(np.float32)
    dx = ops.linear_dgrad_f32(dev(dy), dev(w), None).cpu().numpy()
    assert rel_err(dx, dy.astype(np.float64) @ w.astype(np.float64)) < 1e-5
    dxh = ops.linear_dgrad_f32(dev(dy), dev(w), dev(h)).cpu().numpy()
    assert rel_err(dxh, (dy.astype(np.float64) @ w.astype(np.float64)) * h * (1 - h)) < 1e-5


def _bf16_round(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize('shape', LINEAR_SHAPES)
@pytest.mark.parametrize('gather', [False, True])
def test_linear_kernels_bf16(shape, gather):
    """bf16 operands are exact inputs here (pre-rounded), so only accumulation order and the output rounding differ:
    outputs are compared at 1e-2 (bf16 output rounding 4e-3), fp32 weight gradients at 1e-4."""
    m, k, n = shape
    rng = np.random.RandomState(m + k + n + 1)
    w = _bf16_round(rng.uniform(-0.1, 0.1, (n, k)).astype(np.float32))
    b = rng.uniform(-0.1, 0.1, (n,)).astype(np.float32)
    if gather:
        table = _bf16_round(rng.uniform(0, 1, (max(m // 7, 2), k)).astype(np.float32))
        rows = rng.randint(-1, table.shape[0], size=m).astype(np.int32)
        a = np.where(rows[:, None] < 0, 0, table[np.maximum(rows, 0)]).astype(np.float32)
        a_dev, rows_dev = ops.cast_pad_bf16(dev(table)), dev(rows)
    else:
        a = _bf16_round(rng.uniform(0, 1, (m, k)).astype(np.float32))
        a_dev, rows_dev = ops.cast_pad_bf16(dev(a)), None
    w_bf = ops.cast_pad_bf16(dev(w))
    assert a_dev.shape[1] % 8 == 0 and torch.all(a_dev[:, k:] == 0)
    for act in (ops.ACT_NONE, ops.ACT_SIGMOID):
        want = a.astype(np.float64) @ w.T.astype(np.float64) + b
        if act:
            want = 1 / (1 + np.exp(-want))
        y = ops.linear_fwd_bf16(a_dev, rows_dev, m, k, w_bf, dev(b), n, act)
        assert y.dtype == torch.bfloat16 and y.shape == (m, ops.pad_ld(n))
        assert torch.all(y[:, n:] == 0)
        assert rel_err(y[:, :n].float().cpu().numpy(), want) < 1e-2, (shape, act)
        y32 = ops.linear_fwd_bf16(a_dev, rows_dev, m, k, w_bf, dev(b), n, act, out_f32=True)
        assert y32.dtype == torch.float32 and rel_err(y32[:, :n].cpu().numpy(), want) < 1e-4, (shape, act)
    dy = _bf16_round(rng.standard_normal((m, n)).astype(np.float32))
    dy_bf = ops.cast_pad_bf16(dev(dy))
    dw, db = ops.linear_wgrad_bf16(dy_bf, a_dev, rows_dev, m, n, k)
    assert rel_err(dw.cpu().numpy(), dy.T.astype(np.float64) @ a.astype(np.float64)) < 1e-4
    assert rel_err(db.cpu().numpy(), dy.astype(np.float64).sum(axis=0)) < 1e-4
    wt = ops.cast_transpose_bf16(dev(w))
    assert wt.shape == (k, ops.pad_ld(n))
    np.testing.assert_array_equal(wt[:, :n].float().cpu().numpy(), w.T)
    dx = ops.linear_dgrad_bf16(dy_bf, m, n, wt, k, None, out_f32=True)
    assert rel_err(dx[:, :k].cpu().numpy(), dy.astype(np.float64) @ w.astype(np.float64)) < 1e-4
    h = _bf16_round(rng.uniform(0.05, 0.95, (m, k)).astype(np.float32))
    dxh = ops.linear_dgrad_bf16(dy_bf, m, n, wt, k, ops.cast_pad_bf16(dev(h)))
    assert torch.all(dxh[:, k:] == 0)
    assert rel_err(dxh[:, :k].float().cpu().numpy(), (dy.astype(np.float64) @ w.astype(np.float64)) * h * (1 - h)) < 1e-2


def test_cast_params_batched():
    rng = np.random.RandomState(4)
    ws = [rng.standard_normal(sh).astype(np.float32) for sh in [(512, 600), (128, 512), (32, 128), (1, 32)]]
    plain, trans = ops.cast_params_bf16([dev(w) for w in ws], want_t=(1, 2))
    for w, wb in zip(ws, plain):
        assert wb.shape == (w.shape[0], ops.pad_ld(w.shape[1]))
        np.testing.assert_array_equal(wb[:, :w.shape[1]].float().cpu().numpy(), _bf16_round(w))
        assert torch.all(wb[:, w.shape[1]:] == 0)
    assert trans[0] is None and trans[3] is None
    for i in (1, 2):
        assert trans[i].shape == (ws[i].shape[1], ops.pad_ld(ws[i].shape[0]))
        np.testing.assert_array_equal(trans[i][:, :ws[i].shape[0]].float().cpu().numpy(), _bf16_round(ws[i]).T)


@pytest.mark.parametrize('bt', [(7, 45), (64, 1000), (3, 32)])
def test_fused_tail_kernel_vs_numpy(bt):
    """mg_f0_tail_bf16 (layers 3-4 + masked MSE, forward and backward) against a float64 restatement fed the same
    bf16-rounded H2 / W3.  Outputs are fp32 except dZ2 (bf16)."""
    b, t = bt
    m = b * t
    rng = np.random.RandomState(b * t)
    h2 = _bf16_round(rng.uniform(0.05, 0.95, (m, 128)).astype(np.float32))
    w3 = rng.uniform(-0.2, 0.2, (32, 128)).astype(np.float32)
    b3 = rng.uniform(-0.1, 0.1, 32).astype(np.float32)
    w4 = rng.uniform(-0.3, 0.3, (1, 32)).astype(np.float32)
    b4 = rng.uniform(-0.1, 0.1, 1).astype(np.float32)
    tgt = rng.standard_normal(m).astype(np.float32)
    sl = rng.randint(1, t + 1, size=b).astype(np.int64)
    sl[0] = t
    grads = torch.empty(32 * 128 + 32 + 32 + 1, device=DEV)
    pred, loss, dz2 = ops.f0_tail(ops.cast_pad_bf16(dev(h2)), dev(w3), dev(b3), dev(w4), dev(b4), dev(tgt), dev(sl), b, t,
                                  grads)
    w3r = _bf16_round(w3).astype(np.float64)
    z3 = h2.astype(np.float64) @ w3r.T + b3
    h3 = 1 / (1 + np.exp(-z3))
    p = h3 @ w4[0].astype(np.float64) + b4[0]
    mask = (np.arange(t)[None, :] < sl[:, None]).reshape(-1).astype(np.float64)
    nb = np.repeat(sl, t).astype(np.float64)
    e = p - tgt
    want_loss = (mask * e * e / nb).sum() / b
    dpred = 2 * e * mask / (nb * b)
    dz3 = dpred[:, None] * w4[0][None, :] * h3 * (1 - h3)
    want_dz2 = (dz3 @ w3r) * h2 * (1 - h2)
    np.testing.assert_allclose(pred.cpu().numpy(), p, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(loss.item(), want_loss, rtol=2e-3)
    assert rel_err(dz2[:, :128].float().cpu().numpy(), want_dz2) < 2e-2
    g = grads.cpu().numpy().astype(np.float64)
    assert rel_err(g[:4096].reshape(32, 128), dz3.T @ h2) < 1e-2
    assert rel_err(g[4096:4128], dz3.sum(axis=0)) < 1e-2
    assert rel_err(g[4128:4160], dpred @ h3) < 1e-2
    assert abs(g[4160] - dpred.sum()) < 1e-2 * np.abs(dpred).sum()


@pytest.mark.parametrize('rows_kind', ['runs', 'short_runs', 'random'])
@pytest.mark.parametrize('m,n,act', [(70000, 512, 1), (4999, 128, 0), (25600, 256, 1)])
def test_linear_fwd_run_staged_equals_frame_staged(m, n, act, rows_kind):
    """mg_linear_fwd_bf16 with MG_ACT_ROWS_RUNS (csrc/gemm_nt_runs.hip: the gathered operand staged once per distinct row, 256 x 128
    tiles, two workgroups per CU) against the same call without the hint (gemm_nt_persist, every frame row staged): the hint must
    not change a bit - same products, same order of the contraction.  'runs' = phone-like runs incl. -1 pad rows, 'short_runs' =
    runs of 1-4 frames (tiles with more than 64 runs: several passes), 'random' = every frame its own row (four passes)."""
    k = 600
    rng = np.random.RandomState(m + n)
    table = ops.cast_pad_bf16(dev(rng.uniform(0, 1, (m // 9 + 3, k)).astype(np.float32)))
    if rows_kind == 'random':
        rows = rng.randint(-1, table.shape[0], size=m)
    else:
        lens = rng.randint(1, 40 if rows_kind == 'runs' else 5, size=m)
        rows = np.repeat(rng.randint(-1, table.shape[0], size=m), lens)[:m]
    rows = dev(rows.astype(np.int32))
    (w_bf,), _ = ops.cast_params_bf16([dev(rng.uniform(-0.1, 0.1, (n, k)).astype(np.float32))])
    bias = dev(rng.uniform(-0.5, 0.5, n).astype(np.float32))
    want = ops.linear_fwd_bf16(table, rows, m, k, w_bf, bias, n, act)
    got = ops.linear_fwd_bf16(table, rows, m, k, w_bf, bias, n, act, rows_runs=True)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    # the two schedules of the run-staged kernel (MG_TUNE_FORM 0 / 16: fragments read one k-step ahead or at the head of their own step)
    from morgana_amd import _lib
    lib = _lib.load()
    other = 16 if lib.mg_set_tuning(0, 16) == 0 else None
    try:
        if other is not None:
            alt = ops.linear_fwd_bf16(table, rows, m, k, w_bf, bias, n, act, rows_runs=True)
            assert torch.equal(alt.view(torch.int16), want.view(torch.int16))
    finally:
        lib.mg_set_tuning(0, 0)


@pytest.fixture
def l2tail_variant(request):
    """MG_TUNE_AB for mg_f0_l2tail_bf16: 0 = the product kernel, 64 = the producer / consumer role split - an experiment that must
    stay correct, compiled into the lab build only (MORGANA_HIP_LIB=morgana_amd/libmorgana_hip_lab.so): the product library refuses
    the value and the case is skipped."""
    from morgana_amd import _lib
    if _lib.load().mg_set_tuning(7, request.param) != 0:
        pytest.skip('experiment kernels are in the lab build only (make -C morgana_amd/csrc lab)')
    yield request.param
    _lib.load().mg_set_tuning(7, 0)


@pytest.mark.parametrize('l2tail_variant', [0, 67, 68, 64, 69], indirect=True)  # 67 = the 16x16x32 form (f0_l2tail16_kernel), 69 = the wide form (l2tail_wide.hip)
@pytest.mark.parametrize('bt', [(7, 45), (64, 1000), (3, 32), (1, 1), (5, 333), (37, 901)])
def test_l2tail_kernel_vs_numpy_and_unfused_pair(bt, l2tail_variant):
    """mg_f0_l2tail_bf16 (the 512 -> 128 sigmoid layer inside the fused tail: README.rst:65-73 layers 2-4 + losses.py:29-51, forward
    and backward in one pass over H1) against a float64 restatement fed the same bf16 operands and the kernel's own bf16 rounding of
    H2, and against the unfused pair it replaces (mg_linear_fwd_bf16 + mg_f0_tail_bf16): those two differ only in the summation
    order inside the 512-deep dot products."""
    b, t = bt
    m = b * t
    rng = np.random.RandomState(b * t + 1)
    h1 = _bf16_round(rng.uniform(0.05, 0.95, (m, 512)).astype(np.float32))
    w2 = rng.uniform(-0.08, 0.08, (128, 512)).astype(np.float32)
    b2 = rng.uniform(-0.1, 0.1, 128).astype(np.float32)
    w3 = rng.uniform(-0.2, 0.2, (32, 128)).astype(np.float32)
    b3 = rng.uniform(-0.1, 0.1, 32).astype(np.float32)
    w4 = rng.uniform(-0.3, 0.3, (1, 32)).astype(np.float32)
    b4 = rng.uniform(-0.1, 0.1, 1).astype(np.float32)
    tgt = rng.standard_normal(m).astype(np.float32)
    sl = rng.randint(1, t + 1, size=b).astype(np.int64)
    sl[0] = t
    h1_d = ops.cast_pad_bf16(dev(h1))
    (w2_bf,), _ = ops.cast_params_bf16([dev(w2)])
    grads = torch.full((32 * 128 + 32 + 32 + 1 + 1,), float('nan'), device=DEV)
    pred, loss, dz2 = ops.f0_l2tail(h1_d, w2_bf, dev(b2), dev(w3), dev(b3), dev(w4), dev(b4), dev(tgt), dev(sl), b, t, grads)
    assert loss.data_ptr() == grads[4161:].data_ptr()                 # the loss came out of the gradients' reduce launch

    w2r, w3r = _bf16_round(w2).astype(np.float64), _bf16_round(w3).astype(np.float64)
    h2 = _bf16_round((1 / (1 + np.exp(-(h1.astype(np.float64) @ w2r.T + b2)))).astype(np.float32)).astype(np.float64)
    z3 = h2 @ w3r.T + b3
    h3 = 1 / (1 + np.exp(-z3))
    p = h3 @ w4[0].astype(np.float64) + b4[0]
    mask = (np.arange(t)[None, :] < sl[:, None]).reshape(-1).astype(np.float64)
    nb = np.repeat(sl, t).astype(np.float64)
    e = p - tgt
    want_loss = (mask * e * e / nb).sum() / b
    dpred = 2 * e * mask / (nb * b)
    dz3 = dpred[:, None] * w4[0][None, :] * h3 * (1 - h3)
    want_dz2 = (dz3 @ w3r) * h2 * (1 - h2)
    np.testing.assert_allclose(pred.cpu().numpy(), p, rtol=3e-3, atol=3e-3)
    np.testing.assert_allclose(loss.item(), want_loss, rtol=3e-3)
    assert rel_err(dz2.float().cpu().numpy(), want_dz2) < 2e-2
    g = grads.cpu().numpy().astype(np.float64)
    assert np.isfinite(g).all()
    assert rel_err(g[:4096].reshape(32, 128), dz3.T @ h2) < 1e-2
    assert rel_err(g[4096:4128], dz3.sum(axis=0)) < 1e-2
    assert rel_err(g[4128:4160], dpred @ h3) < 1e-2
    assert abs(g[4160] - dpred.sum()) < 1e-2 * np.abs(dpred).sum() + 1e-12

    # the pair it replaces
    h2_d = ops.linear_fwd_bf16(h1_d, None, m, 512, w2_bf, dev(b2), 128, ops.ACT_SIGMOID)
    grads_u = torch.empty(32 * 128 + 32 + 32 + 1 + 1, device=DEV)
    pred_u, loss_u, dz2_u = ops.f0_tail(h2_d, dev(w3), dev(b3), dev(w4), dev(b4), dev(tgt), dev(sl), b, t, grads_u)
    np.testing.assert_allclose(pred.cpu().numpy(), pred_u.cpu().numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(loss.item(), loss_u.item(), rtol=1e-3)
    assert rel_err(dz2.float().cpu().numpy(), dz2_u[:, :128].float().cpu().numpy()) < 1e-2
    assert rel_err(grads[:4161].cpu().numpy(), grads_u[:4161].cpu().numpy()) < 1e-2
    # run to run: bit identical
    grads2 = torch.empty_like(grads)
    pred2, loss2, dz2_2 = ops.f0_l2tail(h1_d, w2_bf, dev(b2), dev(w3), dev(b3), dev(w4), dev(b4), dev(tgt), dev(sl), b, t, grads2)
    assert torch.equal(pred, pred2) and torch.equal(dz2, dz2_2) and torch.equal(grads, grads2)


def test_l2tail_rows_kernel_vs_unfused_pair():
    """mg_f0_l2tail_rows_bf16 (phone-rate form: rows that stand for groups of frames, weights and mean targets from
    mg_phone_target_stats) against mg_linear_fwd_bf16 + mg_f0_tail_rows_bf16."""
    m = 21504
    rng = np.random.RandomState(11)
    h1_d = ops.cast_pad_bf16(dev(rng.uniform(0.05, 0.95, (m, 512)).astype(np.float32)))
    (w2_bf,), _ = ops.cast_params_bf16([dev(rng.uniform(-0.08, 0.08, (128, 512)).astype(np.float32))])
    b2 = dev(rng.uniform(-0.1, 0.1, 128).astype(np.float32))
    w3, b3 = dev(rng.uniform(-0.2, 0.2, (32, 128)).astype(np.float32)), dev(rng.uniform(-0.1, 0.1, 32).astype(np.float32))
    w4, b4 = dev(rng.uniform(-0.3, 0.3, (1, 32)).astype(np.float32)), dev(rng.uniform(-0.1, 0.1, 1).astype(np.float32))
    ybar = dev(rng.standard_normal(m).astype(np.float32))
    weight = dev((rng.uniform(0, 1, m) * (rng.uniform(0, 1, m) > 0.1) / m).astype(np.float32))
    grads = torch.empty(32 * 128 + 32 + 32 + 1 + 1, device=DEV)
    pred, loss, dz2 = ops.f0_l2tail_rows(h1_d, w2_bf, b2, w3, b3, w4, b4, ybar, weight, grads)
    h2_d = ops.linear_fwd_bf16(h1_d, None, m, 512, w2_bf, b2, 128, ops.ACT_SIGMOID)
    grads_u = torch.empty_like(grads)
    pred_u, loss_u, dz2_u = ops.f0_tail_rows(h2_d, w3, b3, w4, b4, ybar, weight, grads_u)
    np.testing.assert_allclose(pred.cpu().numpy(), pred_u.cpu().numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(loss.item(), loss_u.item(), rtol=1e-3)
    assert rel_err(dz2.float().cpu().numpy(), dz2_u[:, :128].float().cpu().numpy()) < 1e-2
    assert rel_err(grads[:4161].cpu().numpy(), grads_u[:4161].cpu().numpy()) < 1e-2

    # the tail's reduce folded into the launch that repeats the prediction (mg_f0_l2tail_rows_slabs_bf16 + mg_expand_column_reduce_f32)
    # against the two launches it replaces (reduce, then mg_expand_column_loss_f32): the same sums in the same order - EQUAL
    frames = 50000
    rows = dev(np.sort(rng.randint(0, m, size=frames)).astype(np.int32))
    n_table, extra = m - 1024, 1024
    partials = dev(rng.uniform(0, 1e-3, (n_table + 15) // 16 + extra // 4).astype(np.float32))    # per-block sums of the loss's constant term
    assert partials.numel() == (n_table + 15) // 16 + extra // 4
    loss_before = loss.clone()
    want_pred = ops.expand_column(pred, rows, loss_const=(partials, n_table, extra, loss))       # adds the constant to `loss` in place
    grads_f = torch.empty_like(grads)
    got_pred, got_loss, dz2_f = ops.f0_l2tail_rows_expand(h1_d, w2_bf, b2, w3, b3, w4, b4, ybar, weight, grads_f, rows,
                                                          (partials, n_table, extra))
    assert torch.equal(got_pred, want_pred) and torch.equal(dz2_f, dz2)
    assert torch.equal(grads_f[:4161], grads[:4161])
    assert got_loss.item() == loss.item() and loss.item() != loss_before.item()

    # ... and both jobs DEFERRED to the end of the backward's first launch (a step captured whole into a HIP graph:
    # mg_linear_wgrad_dgrad_expand_bf16 - rider blocks behind the weight-gradient and dgrad tiles of the 512 -> 128 layer): every
    # output of the tail and of the pair EQUAL to the separate launches; nothing of the tail's is written before that launch
    (_,), (w2_t,) = ops.cast_params_bf16([dev(rng.uniform(-0.08, 0.08, (128, 512)).astype(np.float32))], want_plain=True, want_t=(0,))
    for frames_d in (frames, 256000):
        rows_d = rows if frames_d == frames else dev(np.sort(rng.randint(0, m, size=frames_d)).astype(np.int32))
        grads_w = torch.empty_like(grads)
        want_p, want_l, dz2_w = ops.f0_l2tail_rows_expand(h1_d, w2_bf, b2, w3, b3, w4, b4, ybar, weight, grads_w, rows_d,
                                                          (partials, n_table, extra))
        want_l = want_l.clone()
        slab_w, ns_w, st_w, dx_w = ops.linear_wgrad_dgrad_bf16(dz2_w, h1_d, m, 128, 512, w2_t)
        slab_w = slab_w.view(torch.float32)[:ns_w * st_w].clone()
        grads_d = torch.full_like(grads, float('nan'))
        got_p, got_l, dz2_d, tail = ops.f0_l2tail_rows_expand(h1_d, w2_bf, b2, w3, b3, w4, b4, ybar, weight, grads_d, rows_d,
                                                              (partials, n_table, extra), defer=True)
        got_p.fill_(float('nan'))
        assert torch.isnan(grads_d).all()                                 # deferred means deferred
        slab_d, ns_d, st_d, dx_d = ops.linear_wgrad_dgrad_bf16(dz2_d, h1_d, m, 128, 512, w2_t, tail=tail)
        assert (ns_d, st_d) == (ns_w, st_w) and torch.equal(dx_d, dx_w) and torch.equal(dz2_d, dz2_w)
        assert torch.equal(slab_d.view(torch.float32)[:ns_d * st_d], slab_w)
        assert torch.equal(got_p, want_p) and torch.equal(grads_d[:4162], grads_w[:4162]) and got_l.item() == want_l.item()
        # ... or finished by a launch of its own when the backward pass has nothing for it to ride in
        grads_e = torch.full_like(grads, float('nan'))
        got_e, _, _, tail = ops.f0_l2tail_rows_expand(h1_d, w2_bf, b2, w3, b3, w4, b4, ybar, weight, grads_e, rows_d,
                                                      (partials, n_table, extra), defer=True)
        ops.finish_deferred_tail(tail)
        assert torch.equal(got_e, want_p) and torch.equal(grads_e[:4162], grads_w[:4162])
        # loss_only: the riders leave the gradients' sums to the update kernel (which takes the slabs as a source of its plan) and
        # form the chunk that holds the loss
        grads_l = torch.full_like(grads, float('nan'))
        got_lp, got_ll, dz2_l, tail = ops.f0_l2tail_rows_expand(h1_d, w2_bf, b2, w3, b3, w4, b4, ybar, weight, grads_l, rows_d,
                                                                (partials, n_table, extra), defer=True)
        ops.linear_wgrad_dgrad_bf16(dz2_l, h1_d, m, 128, 512, w2_t, tail=dict(tail, loss_only=True))
        assert torch.equal(got_lp, want_p) and got_ll.item() == want_l.item() and torch.isnan(grads_l[:4176 - 16]).all()
        sums = ops.slab_reduce(tail['ws'], tail['n_slabs'], tail['stride'], 4161, torch.empty(4161, device=DEV))
        assert torch.equal(sums, grads_w[:4161])                          # what the update kernel will form from the slabs


@pytest.mark.parametrize('m', [6144, 12288, 6000])
def test_few_tile_gemms_take_narrow_tiles_with_the_same_results(m):
    """mg_try_nt_big's tile rule (128-wide tiles when 256-wide ones would be fewer than 128: the 6,144-row GEMMs of RNN_SPSS) against
    the rule before it (MG_TUNE_AB 97: 256-wide whenever N allows) and against the 128 x 128 kernel (98): the same products in the same
    k order - EQUAL bit for bit - for a sigmoid forward, a wide forward with fp32 output, a dgrad with fp32 output and a 256-wide
    forward; and each against a float64 product of the same bf16 operands."""
    from morgana_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(m)
    for k, n, kind in ((600, 512, 'fwd_sig'), (512, 1536, 'fwd'), (1536, 512, 'dgrad'), (512, 256, 'fwd_bf')):
        a = ops.cast_pad_bf16(dev((rng.standard_normal((m, k)) * 0.5).astype(np.float32)))
        w = ops.cast_pad_bf16(dev((rng.standard_normal((n, k)) * 0.05).astype(np.float32)))
        bias = dev(rng.standard_normal(n).astype(np.float32))

        def run():
            if kind == 'dgrad':
                return ops.linear_dgrad_bf16(a, m, k, w, n, None, out_f32=True)
            act = ops.ACT_SIGMOID if kind == 'fwd_sig' else ops.ACT_NONE
            return ops.linear_fwd_bf16(a, None, m, k, w, bias, n, act, out_f32=(kind == 'fwd'))

        outs = {}
        try:
            for v in (0, 97, 98):
                lib.mg_set_tuning(7, v)
                outs[v] = run().clone()
        finally:
            lib.mg_set_tuning(7, 0)
        assert torch.equal(outs[0], outs[97]), (kind, 'rule vs wide tiles')
        if kind != 'fwd_sig':                                   # (the 128 x 128 kernel's sigmoid epilogue rounds through another path)
            assert torch.equal(outs[0][:, :n], outs[98][:, :n]), (kind, 'rule vs 128 x 128')
        ref = a[:, :k].double() @ w[:, :k].double().t()
        if kind != 'dgrad':
            ref = ref + bias.double()
        if kind == 'fwd_sig':
            ref = torch.sigmoid(ref)
        err = (outs[0][:, :n].double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < (1e-2 if outs[0].dtype == torch.bfloat16 else 1e-5), (kind, err)


@pytest.mark.parametrize('m,frames,n_slabs', [(21504, 256000, 168), (30001, 70001, 235), (21504, 1000, 3)])
def test_deferred_tail_riders_on_synthetic_slabs(m, frames, n_slabs):
    """mg_linear_wgrad_dgrad_expand_bf16 and the update kernel's mg_adam_tail against mg_expand_column_reduce_f32 on made-up inputs
    (random slabs, row map, partial sums): the repeated prediction, every element of the slab sum and the loss EQUAL bit for bit - for
    the C2 shape (riders behind the pair grid), for a row count the one-grid form does not take (the entry point then runs the
    separate launches), and for a frame count smaller than one rider's share."""
    rng = np.random.RandomState(m + frames)
    n, stride = 4162, 4164
    n_table, extra = m - 1024, 1024
    (_,), (wt,) = ops.cast_params_bf16([dev(rng.uniform(-0.1, 0.1, (128, 512)).astype(np.float32))], want_plain=True, want_t=(0,))
    h = ops.cast_pad_bf16(dev(rng.uniform(0.05, 0.95, (m, 512)).astype(np.float32)))
    dy = ops.cast_pad_bf16(dev((rng.standard_normal((m, 128)) * 0.01).astype(np.float32)))
    ws = dev((rng.standard_normal(n_slabs * stride) * 0.1).astype(np.float32))
    base = dict(pred_rows=dev(rng.standard_normal(m).astype(np.float32)), rows=dev(np.sort(rng.randint(0, m, size=frames)).astype(np.int32)),
                partials=dev(rng.uniform(0, 1e-3, (n_table + 15) // 16 + extra // 4).astype(np.float32)), n_table_rows=n_table, extra=extra,
                ws=ws.view(torch.uint8), n=n, stride=stride, n_slabs=n_slabs)

    def fresh():
        return dict(base, out=torch.full((frames,), float('nan'), device=DEV), grads_out=torch.full((n,), float('nan'), device=DEV))

    want = fresh()
    ops.finish_deferred_tail(want)
    assert not torch.isnan(want['out']).any() and not torch.isnan(want['grads_out']).any()
    got = fresh()                                                           # riders behind the pair grid: everything
    ops.linear_wgrad_dgrad_bf16(dy, h, m, 128, 512, wt, tail=got)
    assert torch.equal(got['out'], want['out']) and torch.equal(got['grads_out'], want['grads_out'])
    got = fresh()                                                           # ... the loss chunk only
    ops.linear_wgrad_dgrad_bf16(dy, h, m, 128, 512, wt, tail=dict(got, loss_only=True))
    assert torch.equal(got['out'], want['out']) and torch.equal(got['grads_out'][-2:], want['grads_out'][-2:])
    # the update launch's first blocks (mg_adam_tail): repeated prediction + the loss chunk, next to an ordinary update
    got = fresh()
    nparam = 20000
    p_, g_, m_, v_ = (torch.zeros(nparam, device=DEV) for _ in range(4))
    scal = torch.tensor([0.01, 1.0], device=DEV)
    ops.adam_step_plan(p_, g_, m_, v_, (0.9, 0.999), 1e-8, 0.0, scal, tail=got)
    assert torch.equal(got['out'], want['out']) and torch.equal(got['grads_out'][-2:], want['grads_out'][-2:])


@pytest.mark.parametrize('m,n,k,lda', [(64000, 1536, 512, 512), (40003, 2048, 500, 512), (9000, 256, 400, 512), (4100, 512, 512, 512),
                                       (70001, 768, 512, 512), (64000, 512, 600, 640), (40003, 512, 609, 640), (9000, 256, 630, 640),
                                       (70001, 256, 640, 640)])
def test_wgrad_square_tile_equals_the_wide_tile(m, n, k, lda):
    """The 256 x 256 tile of the wide weight-gradient kernel (default for a 512-wide operand when N is a multiple of 256; 256 x 320 for a
    640-wide one) against its 128 x 512 / 128 x 640 tile (MG_TUNE_AB = 91): every element is summed over the same rows in the same order
    by one wave -> EQUAL slabs, dW and db; and against a float64 product of the same bf16 operands."""
    from morgana_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(m + n)
    dy = ops.cast_pad_bf16(dev((rng.standard_normal((m, n)) * 0.05).astype(np.float32)))
    a = torch.zeros((m, lda), dtype=torch.bfloat16, device=DEV)              # a 512- / 640-wide table with k columns in use
    a[:, :k] = dev(rng.uniform(-1, 1, (m, k)).astype(np.float32)).to(torch.bfloat16)
    try:
        assert lib.mg_set_tuning(7, 91) == 0
        want_w, want_b = ops.linear_wgrad_bf16(dy, a, None, m, n, k)
        want_slab, want_ns, want_stride = ops.linear_wgrad_slabs_bf16(dy, a, None, m, n, k)
    finally:
        lib.mg_set_tuning(7, 0)
    got_w, got_b = ops.linear_wgrad_bf16(dy, a, None, m, n, k)
    got_slab, got_ns, got_stride = ops.linear_wgrad_slabs_bf16(dy, a, None, m, n, k)
    assert (got_ns, got_stride) == (want_ns, want_stride)
    gs, ws = got_slab.view(torch.float32), want_slab.view(torch.float32)
    for i in range(got_ns):
        assert torch.equal(gs[i * got_stride:i * got_stride + n * k + n], ws[i * got_stride:i * got_stride + n * k + n])
    assert torch.equal(got_w, want_w) and torch.equal(got_b, want_b)
    ref = dy.double().t() @ a.double()[:, :k]
    assert np.linalg.norm(got_w.cpu().numpy() - ref.cpu().numpy()) / np.linalg.norm(ref.cpu().numpy()) < 1e-5
    ref_b = dy.double().sum(0)
    assert (got_b.double() - ref_b).abs().max().item() <= 1e-5 * ref_b.abs().max().item() + 1e-6


@pytest.mark.parametrize('n,k,total', [(1536, 512, 9000), (384, 500, 4100), (1536, 512, 73613)])
def test_wgrad_with_both_operands_gathered(n, k, total):
    """mg_linear_wgrad_rows_bf16 (dW = sum_i dY[dy_rows[i]]^T A[rows[i]]: a recurrent layer's weight gradients over the valid frames of a
    ragged batch, picked out of padded (B, T) / (B, T + 1) arrays) against mg_linear_wgrad_bf16 on explicitly gathered copies: the same
    products in the same order -> EQUAL; and against a float64 product of the gathered operands."""
    rng = np.random.RandomState(total)
    b, t = 16, (total // 16) * 2
    lens = rng.randint(1, t + 1, size=b)
    lens[-1] = t
    while lens.sum() < total:
        lens[rng.randint(b)] = t
    dense = np.concatenate([bb * t + np.arange(lens[bb]) for bb in range(b)])[:total].astype(np.int32)
    state = (dense + dense // t).astype(np.int32)
    dy = ops.cast_pad_bf16(dev((rng.standard_normal((b * t, n)) * 0.05).astype(np.float32)))
    a = ops.cast_pad_bf16(dev(rng.uniform(-1, 1, (b * (t + 1), k)).astype(np.float32)))
    assert ops.wgrad_rows_ok(total, n, k, a.shape[1], dy.shape[1])
    dyr, ar = dev(dense), dev(state)
    got_w, got_b = ops.linear_wgrad_rows_bf16(dy, dyr, a, ar, total, n, k)
    dy_g, a_g = dy[dyr.long()].contiguous(), a[ar.long()].contiguous()
    want_w, want_b = ops.linear_wgrad_bf16(dy_g, a_g, None, total, n, k)
    assert torch.equal(got_w, want_w) and torch.equal(got_b, want_b)
    ref = dy_g.double().t() @ a_g.double()[:, :k]
    assert np.linalg.norm(got_w.cpu().numpy() - ref.cpu().numpy()) / np.linalg.norm(ref.cpu().numpy()) < 1e-5
    # rows omitted: the A operand takes the dY row indices (a layer input stored like its gate gradients)
    got_same, _ = ops.linear_wgrad_rows_bf16(dy, dyr, a, None, total, n, k)
    want_same, _ = ops.linear_wgrad_bf16(dy_g, a[dyr.long()].contiguous(), None, total, n, k)
    assert torch.equal(got_same, want_same)


@pytest.mark.parametrize('m', [21504, 21377, 8192, 30001, 4100])
def test_wgrad_dgrad_pair_equals_the_two_launches(m):
    """mg_linear_wgrad_dgrad_bf16 (one grid for a layer's weight-gradient slabs and the dgrad + sigmoid backward below it) against the
    two launches it stands for: dX EQUAL, and the slabs EQUAL to mg_linear_wgrad_slabs_bf16 cut into the same number of splits (the
    pair cuts fewer than the plan when the dgrad tiles leave less of an XCD free; a different split count is a different fp32
    summation order).  m = 21 504 is C2's phone-rate row count; 30 001 leaves too little room (the entry then runs the two launches)."""
    from morgana_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(m)
    n, k = 128, 512
    (_,), (wt,) = ops.cast_params_bf16([dev(rng.uniform(-0.1, 0.1, (n, k)).astype(np.float32))], want_plain=True, want_t=(0,))
    h = ops.cast_pad_bf16(dev(rng.uniform(0.05, 0.95, (m, k)).astype(np.float32)))
    dy = ops.cast_pad_bf16(dev((rng.standard_normal((m, n)) * 0.01).astype(np.float32)))
    slab, n_slabs, stride, dx = ops.linear_wgrad_dgrad_bf16(dy, h, m, n, k, wt)
    got = slab.view(torch.float32)[:n_slabs * stride].clone()
    want_dx = ops.linear_dgrad_bf16(dy, m, n, wt, k, h)
    assert torch.equal(dx, want_dx)
    lib.mg_set_tuning(4, n_slabs)
    try:
        slab2, n2, stride2 = ops.linear_wgrad_slabs_bf16(dy, h, None, m, n, k)
        slab2 = slab2.clone()                                  # (the workspace of the next call may be this buffer)
        dw_same, db_same = ops.linear_wgrad_bf16(dy, h, None, m, n, k)       # the same splits, reduced by the entry point itself
    finally:
        lib.mg_set_tuning(4, 0)
    assert (n2, stride2) == (n_slabs, stride)
    assert torch.equal(got, slab2.view(torch.float32)[:n2 * stride2])
    # mg_slab_reduce_f32 (what a data-parallel rank runs on the slabs before its all-reduce): dW | db in one launch, bit for bit the
    # weight-gradient entry point's own two reduces; accumulating on top of what the destination holds
    both = ops.slab_reduce(slab, n_slabs, stride, n * k + n, torch.empty(n * k + n, device=DEV))
    assert torch.equal(both[:n * k].view(n, k), dw_same) and torch.equal(both[n * k:], db_same)
    twice = ops.slab_reduce(slab, n_slabs, stride, n * k + n, both.clone(), accumulate=True)
    np.testing.assert_allclose(twice.cpu().numpy(), (both + both).cpu().numpy(), rtol=1e-5, atol=1e-5)
    # and the sum of the slabs is the weight gradient
    dw, db = ops.linear_wgrad_bf16(dy, h, None, m, n, k)
    tot = got.view(n_slabs, stride).double().sum(0)
    np.testing.assert_allclose(tot[:n * k].cpu().numpy(), dw.double().reshape(-1).cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(tot[n * k:].cpu().numpy(), db.double().cpu().numpy(), rtol=1e-4, atol=1e-6)
    lib.mg_set_tuning(7, 65)                                   # the entry's two-launch form
    try:
        slab3, n3, stride3, dx3 = ops.linear_wgrad_dgrad_bf16(dy, h, m, n, k, wt)
    finally:
        lib.mg_set_tuning(7, 0)
    assert torch.equal(dx3, want_dx)
    tot3 = slab3.view(torch.float32)[:n3 * stride3].view(n3, stride3).double().sum(0)
    np.testing.assert_allclose(tot3.cpu().numpy(), tot.cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('nbt_knob', [0, 12, 15])
def test_fused_backward_64_frame_steps_equal_32_frame_kernel(nbt_knob):
    """wgrad_fused64_kernel (csrc/bwd_fused64_bf16.hip: 64-frame steps, the run structure of a step from a ballot instead of LDS
    tables, ring allocated by groups per step, passes for steps with more runs than the ring holds) against the 32-frame-step kernel
    it replaces (MG_TUNE_STAGGER = 13), on the same frame ranges: dW, db EQUAL bit for bit where a step's runs fit the ring (the same
    MFMAs on the same operands), to fp32 rounding where steps take several passes.  Row maps: phone-like runs with pad
    frames (-1), short runs (several ring passes per step and steps that do not fit beside their predecessor: the on-demand fetch),
    no runs at all (every frame its own row: the worst case), one long run; M not a multiple of the step.
    nbt_knob: 0 = tiles fetched three steps ahead (the product form), 12 = two steps ahead with the larger ring, 15 = the woven
    single-stream experiment (lab build only)."""
    from morgana_amd import _lib
    lib = _lib.load()
    if lib.mg_set_tuning(0, nbt_knob) != 0:
        pytest.skip('experiment kernels are in the lab build only (make -C morgana_amd/csrc lab)')
    lib.mg_set_tuning(0, 0)
    rng = np.random.RandomState(5)
    k0 = 600
    for m in (40037, 8192):
        table = ops.cast_pad_bf16(dev(rng.uniform(0, 1, (m // 9, k0)).astype(np.float32)))
        (_,), (w2t,) = ops.cast_params_bf16([dev(rng.uniform(-0.1, 0.1, (128, 512)).astype(np.float32))], want_plain=True, want_t=(0,))
        h1 = ops.cast_pad_bf16(dev(rng.uniform(0.05, 0.95, (m, 512)).astype(np.float32)))
        dz2 = ops.cast_pad_bf16(dev((rng.standard_normal((m, 128)) * 0.01).astype(np.float32)))
        for kind in ('runs', 'short', 'random', 'one', 'mixed'):
            if kind == 'runs':
                rows = np.repeat(rng.randint(-1, table.shape[0], size=m), rng.randint(1, 40, size=m))[:m]
            elif kind == 'short':
                rows = np.repeat(rng.randint(-1, table.shape[0], size=m), rng.randint(1, 5, size=m))[:m]
            elif kind == 'random':
                rows = rng.randint(-1, table.shape[0], size=m)
            elif kind == 'one':
                rows = np.full(m, 7)
            else:                                          # stretches of every kind next to each other
                parts, left = [], m
                while left > 0:
                    n = min(left, int(rng.randint(50, 400)))
                    style = rng.randint(0, 4)
                    hi = (40, 5, 2, 400)[style]
                    parts.append(np.repeat(rng.randint(-1, table.shape[0], size=n), rng.randint(1, hi, size=n))[:n])
                    left -= n
                rows = np.concatenate(parts)
            rows = dev(rows.astype(np.int32))
            lib.mg_set_tuning(0, 13)
            try:
                want = ops.linear_bwd_fused_bf16(dz2, w2t, h1, table, rows, m, 512, k0)
                lib.mg_set_tuning(0, nbt_knob)
                got = ops.linear_bwd_fused_bf16(dz2, w2t, h1, table, rows, m, 512, k0)
            finally:
                lib.mg_set_tuning(0, 0)
            if kind in ('runs', 'one'):                    # every step's runs fit the ring: the same MFMAs on the same operands
                assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (m, kind)
            else:                                          # steps multiplied in passes: a 16-frame MFMA's products arrive in two
                for g, w in zip(got, want):               # instructions instead of one, i.e. one more fp32 rounding per pass
                    assert rel_err(g.cpu().numpy(), w.cpu().numpy()) < 1e-5, (m, kind)


@pytest.mark.parametrize('m', [40037, 8192, 4100])
def test_fused_backward_with_second_layer_wgrad(m):
    """mg_linear_bwd_fused2_slabs_bf16 (wgrad_fused3_kernel: the 64-frame fused backward with dW2 = dZ2^T H1 and db2 riding on the
    tiles it stages anyway - the stand-alone layer-2 weight-gradient launch and its pass over H1 go): the first layer's slabs sum to
    EXACTLY what mg_linear_bwd_fused_bf16 gives (same products in the same order; the tiles merely lie in the dual-use image), the
    second layer's to what the stand-alone kernel (mg_linear_wgrad_bf16) and a float64 restatement give (frames summed in other
    chunks: 1e-5).  Phone-like runs with pad frames, short runs (ring passes), no runs; M not a multiple of the 64-frame step (the
    zeroed tail of the dZ2 tile)."""
    rng = np.random.RandomState(m)
    k0, n1 = 600, 512
    table = ops.cast_pad_bf16(dev(rng.uniform(0, 1, (m // 9, k0)).astype(np.float32)))
    (_,), (w2t,) = ops.cast_params_bf16([dev(rng.uniform(-0.1, 0.1, (128, n1)).astype(np.float32))], want_plain=True, want_t=(0,))
    h1_np = _bf16_round(rng.uniform(0.05, 0.95, (m, n1)).astype(np.float32))
    dz2_np = _bf16_round((rng.standard_normal((m, 128)) * 0.01).astype(np.float32))
    h1, dz2 = ops.cast_pad_bf16(dev(h1_np)), ops.cast_pad_bf16(dev(dz2_np))
    want_w2 = dz2_np.astype(np.float64).T @ h1_np.astype(np.float64)
    want_b2 = dz2_np.astype(np.float64).sum(axis=0)
    alone_w2, alone_b2 = ops.linear_wgrad_bf16(dz2, h1, None, m, 128, n1)
    for kind in ('runs', 'short', 'random'):
        if kind == 'runs':
            rows = np.repeat(rng.randint(-1, table.shape[0], size=m), rng.randint(1, 40, size=m))[:m]
        elif kind == 'short':
            rows = np.repeat(rng.randint(-1, table.shape[0], size=m), rng.randint(1, 5, size=m))[:m]
        else:
            rows = rng.randint(-1, table.shape[0], size=m)
        rows = dev(rows.astype(np.int32))
        want_w1, want_b1 = ops.linear_bwd_fused_bf16(dz2, w2t, h1, table, rows, m, n1, k0)
        slab, n_slabs, (off1, st1, cnt1), (off2, st2, cnt2) = ops.linear_bwd_fused2_slabs_bf16(dz2, w2t, h1, table, rows, m, n1, k0)
        floats = slab.view(torch.float32)
        got1 = ops.slab_reduce(floats[off1:], n_slabs, st1, cnt1, torch.empty(cnt1, device=DEV))
        got2 = ops.slab_reduce(floats[off2:], n_slabs, st2, cnt2, torch.empty(cnt2, device=DEV))
        if kind == 'runs':
            assert torch.equal(got1[:n1 * k0].view(n1, k0), want_w1) and torch.equal(got1[n1 * k0:], want_b1), kind
        else:
            assert rel_err(got1[:n1 * k0].cpu().numpy(), want_w1.cpu().numpy().reshape(-1)) < 1e-5, kind
            assert rel_err(got1[n1 * k0:].cpu().numpy(), want_b1.cpu().numpy()) < 1e-5, kind
        w2, b2 = got2[:128 * n1].view(128, n1).cpu().numpy(), got2[128 * n1:].cpu().numpy()
        assert rel_err(w2, want_w2) < 1e-5 and rel_err(b2, want_b2) < 1e-5, kind
        assert rel_err(w2, alone_w2.cpu().numpy()) < 1e-5 and rel_err(b2, alone_b2.cpu().numpy()) < 1e-5, kind


@pytest.mark.parametrize('rows_kind', ['random', 'runs', 'identity'])
@pytest.mark.parametrize('m,n_hidden', [(4999, 512), (9000, 256)])
def test_fused_backward_kernel_vs_numpy(m, n_hidden, rows_kind):
    """mg_linear_bwd_fused_bf16: dW1, db1 from dZ2 without materialising dZ1 = (dZ2 W2) * H1 (1 - H1).
    rows: 'random' = every frame its own source row incl. -1 pads (the run-staged ring cannot prefetch: its slow path),
    'runs' = phone-like runs of equal rows (the case it is built for), 'identity' = no gather (the single-buffered kernel)."""
    k0 = 600
    rng = np.random.RandomState(m)
    if rows_kind == 'identity':
        table = _bf16_round(rng.uniform(0, 1, (m, k0)).astype(np.float32))
        rows, x = None, table.astype(np.float64)
    else:
        table = _bf16_round(rng.uniform(0, 1, (m // 9, k0)).astype(np.float32))
        if rows_kind == 'random':
            rows = rng.randint(-1, table.shape[0], size=m).astype(np.int32)
        else:
            lens = rng.randint(1, 40, size=m)
            rows = np.repeat(rng.randint(-1, table.shape[0], size=m), lens)[:m].astype(np.int32)
        x = np.where(rows[:, None] < 0, 0, table[np.maximum(rows, 0)]).astype(np.float64)
    w2 = _bf16_round(rng.uniform(-0.1, 0.1, (128, n_hidden)).astype(np.float32))
    h1 = _bf16_round(rng.uniform(0.05, 0.95, (m, n_hidden)).astype(np.float32))
    dz2 = _bf16_round((rng.standard_normal((m, 128)) * 0.01).astype(np.float32))
    dz1 = (dz2.astype(np.float64) @ w2.astype(np.float64)) * h1 * (1 - h1)
    dz1_bf = _bf16_round(dz1.astype(np.float32)).astype(np.float64)          # the kernel rounds dZ1 to bf16 in LDS
    want_w, want_b = dz1_bf.T @ x, dz1_bf.sum(axis=0)
    wt2 = ops.cast_transpose_bf16(dev(w2))
    dw, db = ops.linear_bwd_fused_bf16(ops.cast_pad_bf16(dev(dz2)), wt2, ops.cast_pad_bf16(dev(h1)),
                                       ops.cast_pad_bf16(dev(table)), dev(rows) if rows is not None else None, m, n_hidden, k0)
    assert rel_err(dw.cpu().numpy(), want_w) < 5e-3
    assert rel_err(db.cpu().numpy(), want_b) < 5e-3


# ---------------------------------------------------------------------------------------------- whole models
def _load_state(model, state):
    own = model.state_dict()
    for key, value in state.items():
        own[key].copy_(torch.from_numpy(value))
    return model


def _c1_batches():
    return [data.to_device(synthetic.make_batch(8, 200, seed=synthetic.REFERENCE_SEED + 100 * i), DEV)
            for i in range(4)]


def _train(model, batches, n_steps, lr, weight_decay=0.0):
    opt = optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)
    curve = []
    for step in range(n_steps):
        opt.zero_grad()
        loss, _ = model(batches[step % len(batches)])
        loss.backward()
        opt.step()
        curve.append(loss.item())
    return curve


@pytest.mark.parametrize('fused,precision', [(True, 'fp32'), (False, 'fp32'), (True, 'bf16x3'), (False, 'bf16x3')])
def test_f0_model_fp32_golden_curve_and_grads(golden, fused, precision):
    """The reference's own 20-step run (G6) at the north star's 1e-4 - in fp32 mode (exact-fp32 MFMA) and in 'bf16x3' (split-bf16
    operands: three bf16 MFMA products per fp32 product, csrc/split3.hip), the parity-grade mode at the bf16 matrix rate."""
    g = golden('g6_f0_model.npz')
    model = _load_state(models.F0Model(precision=precision, fused_upsample=fused).to(DEV), synthetic.f0_model_state())
    batches = _c1_batches()
    loss, out = model(batches[0])
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss_curve'][0], rtol=RTOL)
    np.testing.assert_allclose(out['pred_norm_lf0'].detach().cpu().numpy()[:, ::25, 0], g['step1_pred_sample'],
                               rtol=RTOL, atol=1e-6)
    for name, prm in model.named_parameters():
        flat = prm.grad.cpu().numpy().ravel()
        np.testing.assert_allclose(np.sqrt((flat.astype(np.float64) ** 2).sum()), g['step1_gradnorm__' + name],
                                   rtol=RTOL)
        # the reference's own 64 sampled gradient elements per parameter: 1e-4 of the largest of them (was
        # rtol 1e-3 + atol 1e-4 max - observed errors are a few 1e-6: gpurun_out/parity_report.json)
        want = g['step1_gradval__' + name]
        assert rel_err(flat[g['step1_gradidx__' + name]], want, 'grad samples ' + name) < RTOL, name
    model.zero_grad()
    curve = _train(model, batches, 20, lr=0.01)
    np.testing.assert_allclose(curve, g['loss_curve'], rtol=RTOL)            # 20-step Adam loss curve, 1e-4
    for name, prm in model.named_parameters():
        v = prm.detach().cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(v.sum(), g['final_sum__' + name], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(np.abs(v).sum(), g['final_abs_sum__' + name], rtol=1e-3)


def test_f0_model_fp32_ragged_weight_decay(golden):
    g = golden('g6_f0_model.npz')
    model = _load_state(models.F0Model(precision='fp32').to(DEV), synthetic.f0_model_state())
    ragged = data.to_device(synthetic.make_batch(6, (40, 120), seed=77), DEV)
    curve = _train(model, [ragged], 5, lr=0.005, weight_decay=1e-3)
    np.testing.assert_allclose(curve, g['ragged_loss_curve'], rtol=RTOL)


@pytest.mark.parametrize('fused_loss', [True, False])
def test_f0_model_bf16_tracks_golden_curve(golden, fused_loss):
    g = golden('g6_f0_model.npz')
    model = _load_state(models.F0Model(precision='bf16', fused_loss=fused_loss).to(DEV), synthetic.f0_model_state())
    curve = _train(model, _c1_batches(), 20, lr=0.01)
    np.testing.assert_allclose(curve, g['loss_curve'], rtol=RTOL_BF16)


def test_f0_model_bf16_fused_and_unfused_agree_on_ragged_batch():
    feats = data.to_device(synthetic.make_batch(12, (300, 700), seed=21), DEV)
    out = {}
    for fused in (True, False):
        model = _load_state(models.F0Model(precision='bf16', fused_loss=fused).to(DEV), synthetic.f0_model_state())
        opt = optim.Adam(model.parameters(), lr=0.01)
        opt.zero_grad()
        loss, o = model(feats)
        loss.backward()
        out[fused] = (loss.item(), o['pred_norm_lf0'].detach().cpu().numpy(),
                      opt.flat_buffers()['grad'].cpu().numpy().copy())
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-3)
    assert np.abs(out[True][1] - out[False][1]).max() < 5e-3
    assert rel_err(out[True][2], out[False][2]) < 3e-2


_C2_ORACLE = {}


def _c2_oracle():
    """(features, state, oracle loss / prediction / gradients) of BASELINE config C2, computed once per session (the oracle takes seconds)."""
    if not _C2_ORACLE:
        feats = synthetic.make_batch(256, 1000)
        state = synthetic.f0_model_state()
        _C2_ORACLE['v'] = (feats, state) + tuple(ref_cpu.f0_forward_backward(state, feats))
    return _C2_ORACLE['v']


@pytest.mark.parametrize('phone_rate', [True, False])
@pytest.mark.parametrize('precision,tol', [('fp32', RTOL), ('bf16x3', RTOL), ('bf16', RTOL_BF16)])
def test_f0_model_full_size_vs_oracle(precision, tol, phone_rate):
    """BASELINE config C2 (256 x 1000 frames) forward + backward against the numpy oracle, in BOTH orders of operations: phone rate
    (the headline's: row-wise layers once per phone row) and ``phone_rate=False`` - every product on the 256 000 frame rows, the
    reference's order, whose own kernels (gemm_nt_runs_kernel, wgrad_fused3_kernel, f0_l2tail_kernel at M = 256 000) thereby meet the
    oracle directly.  The exact modes are held to 1e-4 on loss, prediction and every gradient."""
    from parity_report import note
    feats, state, want_loss, want_pred, want_grads = _c2_oracle()
    model = _load_state(models.F0Model(precision=precision, phone_rate=phone_rate).to(DEV), state)
    loss, out = model(data.to_device(feats, DEV, bf16_tables=model.bf16_table_features()))
    loss.backward()
    note(abs(loss.item() - want_loss) / abs(want_loss), 'loss')
    np.testing.assert_allclose(loss.item(), want_loss, rtol=tol)
    pred = out['pred_norm_lf0'].detach().cpu().numpy()
    assert pred.shape == want_pred.shape
    if precision in ('fp32', 'bf16x3'):
        assert rel_err(pred, want_pred, 'prediction') < 1e-4
    else:
        # The prediction is a cancelling sum of 32 O(0.1) terms built from bf16-rounded O(1) activations, so the bf16
        # mode is held to an ABSOLUTE 5e-3 (activation scale 1), not to a relative bound on the small result.
        assert np.abs(pred - want_pred).max() < 5e-3
    for name, prm in model.named_parameters():
        assert rel_err(prm.grad.cpu().numpy(), want_grads[name], 'grad ' + name) < (RTOL if precision in ('fp32', 'bf16x3') else 5e-2), name


@pytest.mark.parametrize('phone_rate', [True, False])
def test_bf16x3_orders_of_operations_agree_and_track_fp32(phone_rate):
    """'bf16x3' at both orders of operations (F0Model(phone_rate=)) against fp32 mode on a batch the phone-rate form takes: loss and
    prediction to 1e-4, every gradient to 1e-4 of its largest element - the bar of fp32 mode, three bf16 MFMA products per product."""
    feats = data.to_device(synthetic.make_batch(32, 400, seed=12), DEV)
    got = {}
    for precision in ('fp32', 'bf16x3'):
        model = _load_state(models.F0Model(precision=precision, phone_rate=phone_rate).to(DEV), synthetic.f0_model_state())
        loss, out = model(feats)
        loss.backward()
        got[precision] = (loss.item(), out['pred_norm_lf0'].detach().cpu().numpy(),
                          {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()})
    np.testing.assert_allclose(got['bf16x3'][0], got['fp32'][0], rtol=RTOL)
    assert rel_err(got['bf16x3'][1], got['fp32'][1]) < RTOL
    for name in got['fp32'][2]:
        assert rel_err(got['bf16x3'][2][name], got['fp32'][2][name]) < RTOL, name


@pytest.mark.parametrize('rows,cols,lds_extra', [(37, 600, 0), (256, 128, 8), (5, 9, 3), (1024, 512, 0)])
def test_split3_planes(rows, cols, lds_extra):
    """mg_split3_bf16 (csrc/split3.hip): hi = bf16(x), lo = bf16(x - hi), planes [hi | hi | lo] (order 0) / [hi | lo | hi] (order 1),
    zero padding, plain and transposed - bit-exact against the same arithmetic in torch - and hi + lo carries x to 2^-16."""
    rng = np.random.RandomState(rows + cols)
    full = dev((rng.standard_normal((rows, cols + lds_extra)) * np.exp(rng.uniform(-3, 3, (rows, 1)))).astype(np.float32))
    x = full[:, :cols]                                               # a row stride larger than the width
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    assert float(((hi.float() + lo.float()) - x).abs().max() / x.abs().max()) < 2.0 ** -16
    a3, w3, t3 = ops.split3([(x, 0, False), (x, 1, False), (x, 1, True)])
    ldp = ops.pad_ld(cols)
    assert tuple(a3.shape) == (rows, 3 * ldp) and tuple(t3.shape) == (cols, 3 * ops.pad_ld(rows))
    for got, planes in ((a3, (hi, hi, lo)), (w3, (hi, lo, hi))):
        for q, want in enumerate(planes):
            assert torch.equal(got[:, q * ldp:q * ldp + cols], want), q
            assert not bool(got[:, q * ldp + cols:(q + 1) * ldp].any())
    p2 = ops.split3([(x, 2, False)])[0]
    assert tuple(p2.shape) == (2, rows, ldp)
    assert torch.equal(p2[0, :, :cols], hi) and torch.equal(p2[1, :, :cols], lo) and not bool(p2[:, :, cols:].any())
    # three row-stacked planes (the one-launch weight gradient's operands), with the column sums of the values where the plane width allows
    p4 = ops.split3([(x, 4, False)])[0]
    assert tuple(p4.shape) == (3, rows, ldp) and not bool(p4[:, :, cols:].any())
    assert torch.equal(p4[0, :, :cols], hi) and torch.equal(p4[1, :, :cols], lo) and torch.equal(p4[2, :, :cols], hi)
    if ops.split3_colsum_ok(cols):
        p3, slabs = ops.split3([(x, 3, False, 2, None, True)])[0]
        assert tuple(p3.shape) == (3, rows + 2, ldp) and not bool(p3[:, rows:].any()) and not bool(p3[:, :, cols:].any())
        assert torch.equal(p3[0, :rows, :cols], hi) and torch.equal(p3[1, :rows, :cols], hi) and torch.equal(p3[2, :rows, :cols], lo)
        want_sum = x.double().sum(0)
        got_sum = ops.slab_reduce(slabs, slabs.shape[0], slabs.shape[1], cols, torch.empty(cols, device=DEV)).double()
        assert float((got_sum - want_sum).abs().max() / x.abs().double().sum(0).max()) < 1e-6
        # the same sums from the three-plane layouts (what the backward of 'bf16x3' takes: the gradient split once, [hi | lo | hi])
        g1, slabs1 = ops.split3([(x, 1, False, 0, None, True)])[0]
        assert torch.equal(g1, w3)
        got1 = ops.slab_reduce(slabs1, slabs1.shape[0], slabs1.shape[1], cols, torch.empty(cols, device=DEV)).double()
        assert float((got1 - want_sum).abs().max() / x.abs().double().sum(0).max()) < 1e-6
    else:
        with pytest.raises(ValueError):
            ops.split3([(x, 3, False, 0, None, True)])
    # the fused sigmoid gradient: the split of x * s * (1 - s), bit for bit the separate pass followed by the split
    s = torch.sigmoid(dev(rng.standard_normal((rows, cols)).astype(np.float32)))
    y = ops.sigmoid_grad(x.contiguous(), s)
    fused, plain = ops.split3([(x, 2, False, 0, s)])[0], ops.split3([(y, 2, False)])[0]
    assert torch.equal(fused, plain)
    ldt = ops.pad_ld(rows)
    for q, want in enumerate((hi, lo, hi)):
        assert torch.equal(t3[:, q * ldt:q * ldt + rows], want.t())
        assert not bool(t3[:, q * ldt + rows:(q + 1) * ldt].any())


@pytest.mark.parametrize('tag', ['h8', 'h32'])
def test_gru_wrapper_golden(golden, tag):
    g = golden('g7_gru.npz')
    hid = int(tag[1:])
    x_np, sl_np = g[tag + '__x'], g[tag + '__seq_len']
    i_dim = x_np.shape[2]
    gru = torch.nn.GRU(i_dim, hid, batch_first=True).to(DEV)
    st = synthetic.init_gru(np.random.RandomState(5 + hid), i_dim, hid)
    with torch.no_grad():
        for prm, val in zip((gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0), st):
            prm.copy_(dev(val))
    wrapper = utils.RecurrentCuDNNWrapper(gru, precision='fp32')
    x = dev(x_np).requires_grad_(True)
    sl = dev(sl_np)
    out, hn = wrapper(x, None, sl)
    assert tuple(out.shape) == g[tag + '__out'].shape and tuple(hn.shape) == g[tag + '__hn'].shape
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[tag + '__out'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(hn.detach().cpu().numpy(), g[tag + '__hn'], rtol=RTOL, atol=1e-6)
    for b, n in enumerate(sl_np):
        assert torch.all(out[b, n:] == 0)                                 # padded steps exactly zero
    (out * dev(g[tag + '__grad_out'])).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[tag + '__grad_x'], rtol=1e-3, atol=1e-5)
    for pname in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0'):
        np.testing.assert_allclose(getattr(gru, pname).grad.cpu().numpy(), g[tag + '__grad_' + pname], rtol=1e-3,
                                   atol=1e-5)
        getattr(gru, pname).grad = None
    x.grad = None
    h0 = dev(g[tag + '__h0']).requires_grad_(True)
    out, hn = wrapper(x, h0, sl)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[tag + '__out_h0'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(hn.detach().cpu().numpy(), g[tag + '__hn_h0'], rtol=RTOL, atol=1e-6)
    ((out * dev(g[tag + '__grad_out'])).sum() + (hn * dev(g[tag + '__grad_hn'])).sum()).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[tag + '__grad_x_h0'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(h0.grad.cpu().numpy(), g[tag + '__grad_h0'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(gru.weight_hh_l0.grad.cpu().numpy(), g[tag + '__grad_weight_hh_l0_h0'], rtol=1e-3,
                               atol=1e-5)


def test_gru_wrapper_errors_and_single_step():
    gru = torch.nn.GRU(4, 8, batch_first=True).to(DEV)
    wrapper = utils.RecurrentCuDNNWrapper(gru, precision='fp32')
    with pytest.raises(ValueError):
        wrapper(torch.zeros(2, 3, 4, device=DEV))                         # 3-D input without seq_len (utils.py:361-363)
    x = torch.randn(5, 4, device=DEV)
    out, hn = wrapper(x)                                                  # one time slice (utils.py:351-358)
    ref_out, ref_hn = gru(x.unsqueeze(1))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out.squeeze(1).detach().cpu().numpy(), rtol=1e-4,
                               atol=1e-6)


@pytest.mark.parametrize('tag,layers', [('l1', 1), ('l2', 2)])
def test_lstm_wrapper_golden(golden, tag, layers):
    """RecurrentCuDNNWrapper(nn.LSTM), 1 and 2 layers, against the reference's outputs, states and gradients (G10)."""
    g = golden('g10_lstm.npz')
    x_np, sl_np = g[tag + '__x'], g[tag + '__seq_len']
    i_dim, hid = x_np.shape[2], g[tag + '__hn'].shape[2]
    lstm = torch.nn.LSTM(i_dim, hid, num_layers=layers, batch_first=True).to(DEV)
    with torch.no_grad():
        for name, prm in lstm.named_parameters():
            prm.copy_(dev(g['%s__param__%s' % (tag, name)]))
    wrapper = utils.RecurrentCuDNNWrapper(lstm, precision='fp32')
    x = dev(x_np).requires_grad_(True)
    sl = dev(sl_np)
    out, (hn, cn) = wrapper(x, None, sl)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[tag + '__out'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(hn.detach().cpu().numpy(), g[tag + '__hn'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(cn.detach().cpu().numpy(), g[tag + '__cn'], rtol=RTOL, atol=1e-6)
    for b, n in enumerate(sl_np):
        assert torch.all(out[b, n:] == 0)
    (out * dev(g[tag + '__grad_out'])).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[tag + '__grad_x'], rtol=1e-3, atol=1e-5)
    for name, prm in lstm.named_parameters():
        np.testing.assert_allclose(prm.grad.cpu().numpy(), g['%s__grad__%s' % (tag, name)], rtol=1e-3, atol=1e-5)
        prm.grad = None
    x.grad = None
    h0, c0 = dev(g[tag + '__h0']).requires_grad_(True), dev(g[tag + '__c0']).requires_grad_(True)
    out, (hn, cn) = wrapper(x, (h0, c0), sl)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[tag + '__out_s'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(hn.detach().cpu().numpy(), g[tag + '__hn_s'], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(cn.detach().cpu().numpy(), g[tag + '__cn_s'], rtol=RTOL, atol=1e-6)
    ((out * dev(g[tag + '__grad_out'])).sum() + (hn * dev(g[tag + '__grad_hn'])).sum() +
     (cn * dev(g[tag + '__grad_cn'])).sum()).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[tag + '__grad_x_s'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(h0.grad.cpu().numpy(), g[tag + '__grad_h0'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(c0.grad.cpu().numpy(), g[tag + '__grad_c0'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(lstm.weight_hh_l0.grad.cpu().numpy(), g[tag + '__grad_whh0_s'], rtol=1e-3, atol=1e-5)


def test_lstm_512_two_layers_vs_oracle():
    """The reference's shipped cell size (LSTM-512, models/RNN_SPSS.py:36-37) on the H = 512 fast path, bf16 GEMMs around."""
    rng = np.random.RandomState(12)
    b, t, hid = 6, 40, 512
    lstm = torch.nn.LSTM(hid, hid, num_layers=2, batch_first=True).to(DEV)
    params = []
    with torch.no_grad():
        for k in range(2):
            layer = []
            for name in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
                prm = getattr(lstm, '%s_l%d' % (name, k))
                val = rng.uniform(-0.05, 0.05, size=tuple(prm.shape)).astype(np.float32)
                prm.copy_(dev(val))
                layer.append(val)
            params.append(layer)
    x_np = rng.standard_normal((b, t, hid)).astype(np.float32)
    sl_np = np.array([40, 13, 27, 40, 1, 33], dtype=np.int64)
    want = x_np
    for k in range(2):
        want, hn, cn, _ = ref_cpu.lstm_forward(want, sl_np, *params[k])
    for precision, tol in (('fp32', 1e-4), ('bf16', 2e-2)):
        out, (h, c) = utils.RecurrentCuDNNWrapper(lstm, precision=precision)(dev(x_np), None, dev(sl_np))
        assert rel_err(out.detach().cpu().numpy(), want) < tol, precision
        assert rel_err(h[1].detach().cpu().numpy(), hn[0]) < tol


def test_rnn_model_golden(golden):
    g = golden('g7_gru.npz')
    lab_dim, hidden, post, out_dim = [int(v) for v in g['rnn__dims']]
    state = synthetic.rnn_spss_state(seed=31, lab_dim=lab_dim, hidden=hidden, post=post, out_dim=out_dim)
    feats = data.to_device(synthetic.make_batch(6, (20, 60), lab_dim=lab_dim, out_dim=out_dim, target_name='mcep',
                                                frames_per_phone=6.0, seed=99), DEV)
    model = _load_state(models.RNNSPSS(lab_dim, hidden, post, out_dim, precision='fp32').to(DEV), state)
    loss, out = model(feats)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['rnn__loss_curve'][0], rtol=RTOL)
    np.testing.assert_allclose(out['pred_norm_mcep'].detach().cpu().numpy(), g['rnn__step1_pred'], rtol=1e-3, atol=1e-5)
    for name, prm in model.named_parameters():
        want = g['rnn__step1_grad__' + name]
        np.testing.assert_allclose(prm.grad.cpu().numpy(), want, rtol=1e-3, atol=1e-4 * np.abs(want).max())
    model.zero_grad()
    curve = _train(model, [feats], 8, lr=0.01)
    np.testing.assert_allclose(curve, g['rnn__loss_curve'], rtol=RTOL)
    # bf16 GEMMs around the fp32 recurrence track the same curve
    model = _load_state(models.RNNSPSS(lab_dim, hidden, post, out_dim, precision='bf16').to(DEV), state)
    curve = _train(model, [feats], 8, lr=0.01)
    np.testing.assert_allclose(curve, g['rnn__loss_curve'], rtol=RTOL_BF16)


def test_rnn_model_c4_shape_vs_oracle():
    """GRU-512 model at a reduced batch of BASELINE config C4 (8 x 120 frames, ragged) against the numpy oracle."""
    feats = synthetic.make_batch(8, (60, 120), out_dim=80, target_name='mcep', seed=11)
    state = synthetic.rnn_spss_state()
    want_loss, want_pred, want_grads = ref_cpu.rnn_forward_backward(state, feats)
    model = _load_state(models.RNNSPSS(precision='fp32').to(DEV), state)
    loss, out = model(data.to_device(feats, DEV))
    loss.backward()
    np.testing.assert_allclose(loss.item(), want_loss, rtol=RTOL)
    assert rel_err(out['pred_norm_mcep'].detach().cpu().numpy(), want_pred) < 1e-4
    for name, prm in model.named_parameters():
        assert rel_err(prm.grad.cpu().numpy(), want_grads[name]) < 1e-3, name


@pytest.mark.parametrize('b,t,hid', [(5, 37, 128), (16, 50, 512), (33, 12, 256)])
def test_gru_bf16_recurrence_vs_fp32(b, t, hid):
    """mg_gru_fwd_bf16 / mg_gru_bwd_bf16 (bf16 matmul operands, fp32 cell) against the exact-fp32 recurrence on the same
    inputs: ragged lengths incl. a full and a 1-step item, partial batch tiles, an initial state, gradients on outputs and
    h_n.  Tolerance 1e-2 relative (bf16 operand rounding, 2^-9 per product, fp32 accumulation); the bf16 shadows must be
    exactly the rounded fp32 buffers."""
    rng = np.random.RandomState(hid + b)
    xproj = dev(rng.standard_normal((b, t, 3 * hid)).astype(np.float32))
    w_hh = dev((rng.uniform(-1, 1, (3 * hid, hid)) / np.sqrt(hid)).astype(np.float32))
    b_hh = dev(rng.uniform(-0.1, 0.1, 3 * hid).astype(np.float32))
    h0 = dev(rng.standard_normal((b, hid)).astype(np.float32) * 0.5)
    sl_np = rng.randint(1, t + 1, size=b).astype(np.int64)
    sl_np[0], sl_np[-1] = t, 1
    sl = dev(sl_np)
    assert ops.gru_bf16_ok(hid)
    out32, hs32, sv32 = ops.gru_fwd(xproj, w_hh, b_hh, sl, h0, b, t, hid)
    out16, hs16, sv16, hs_bf = ops.gru_fwd_bf16(xproj, w_hh, b_hh, sl, h0, b, t, hid, persistent=False)
    assert rel_err(out16.cpu().numpy(), out32.cpu().numpy()) < 1e-2
    assert rel_err(hs16.cpu().numpy(), hs32.cpu().numpy()) < 1e-2
    valid = (np.arange(t)[None, :] < sl_np[:, None])[:, :, None]          # saved gates past an item's length are never read
    assert rel_err(np.where(valid, sv16.cpu().numpy(), 0), np.where(valid, sv32.cpu().numpy(), 0)) < 1e-2
    assert torch.equal(hs_bf, hs16.to(torch.bfloat16))
    for i, n in enumerate(sl_np):
        assert torch.all(out16[i, n:] == 0)
        assert torch.equal(hs16[i, n:], hs16[i, n:n + 1].expand(t + 1 - n, hid))
    g_out = dev(rng.standard_normal((b, t, hid)).astype(np.float32))
    g_hn = dev(rng.standard_normal((b, hid)).astype(np.float32))
    # same saved tensors for both, so that only the backward recurrence differs
    dx32, dh32, d032 = ops.gru_bwd(g_out, g_hn, hs32, sv32, w_hh, sl, b, t, hid)
    dx16, dh16, d016, dh_bf = ops.gru_bwd_bf16(g_out, g_hn, hs32, sv32, w_hh, sl, b, t, hid, persistent=False)
    assert rel_err(dx16.cpu().numpy(), dx32.cpu().numpy()) < 1e-2
    assert rel_err(dh16.cpu().numpy(), dh32.cpu().numpy()) < 1e-2
    assert rel_err(d016.cpu().numpy(), d032.cpu().numpy()) < 1e-2
    assert torch.equal(dh_bf, dh16.to(torch.bfloat16))
    # deterministic
    again = ops.gru_bwd_bf16(g_out, g_hn, hs32, sv32, w_hh, sl, b, t, hid, persistent=False)
    assert torch.equal(again[0], dx16) and torch.equal(again[2], d016)


@pytest.fixture
def gru_handoff(request):
    from morgana_amd import _lib
    _lib.load().mg_set_tuning(2, request.param)       # MG_TUNE_GRU_HANDOFF: 0 = through the XCD's L2 where verified, 1 = always sc1
    yield request.param
    _lib.load().mg_set_tuning(2, 0)


@pytest.mark.parametrize('gru_handoff', [0, 1, 4], indirect=True)          # 4: the backward with 8 groups x 16-unit slots
@pytest.mark.parametrize('b,t,hid', [(64, 60, 512), (5, 37, 128), (33, 20, 256), (200, 9, 128), (130, 25, 384), (128, 12, 512), (17, 30, 384)])
def test_gru_persistent_equals_stepwise(b, t, hid, gru_handoff):
    """The one-launch recurrence (gru_persist.hip: W_hh in registers, state handed between workgroups with write-through
    stores and flags) against the launch-per-step kernels on the same inputs: same operands, same summation order and the
    same cell code (csrc/gru_cell.h, contraction pinned), so the results must be EQUAL; steps beyond a group's longest sequence
    write zero gate values.
    Repeated to catch a stale hand-off (every element of every step depends on all hand-offs before it)."""
    rng = np.random.RandomState(hid + b)
    xproj = dev(rng.standard_normal((b, t, 3 * hid)).astype(np.float32))
    w_hh = dev((rng.uniform(-1, 1, (3 * hid, hid)) / np.sqrt(hid)).astype(np.float32))
    b_hh = dev(rng.uniform(-0.1, 0.1, 3 * hid).astype(np.float32))
    h0 = dev(rng.standard_normal((b, hid)).astype(np.float32) * 0.5)
    sl_np = rng.randint(1, t + 1, size=b).astype(np.int64)
    sl_np[0] = t
    sl_np[-1] = 1
    assert ops.gru_persist_ok(b, t, hid)
    for sl in (dev(sl_np), None):
        want = ops.gru_fwd_bf16(xproj, w_hh, b_hh, sl, h0, b, t, hid, persistent=False)
        for rep in range(3):
            got = ops.gru_fwd_bf16(xproj, w_hh, b_hh, sl, h0, b, t, hid, persistent=True)
            ops.check_persistent_status()
            for name, g, w in zip(('out', 'hstate', 'saved', 'hstate_bf'), got, want):
                g, w = g.float().cpu().numpy(), w.float().cpu().numpy()
                if name == 'saved' and sl is not None:
                    live = (np.arange(t)[None, :] < sl_np[:, None])[:, :, None]
                    g, w = g * live, w * live
                np.testing.assert_array_equal(g, w, err_msg='%s rep %d' % (name, rep))
        g_out = dev(rng.standard_normal((b, t, hid)).astype(np.float32))
        g_hn = dev(rng.standard_normal((b, hid)).astype(np.float32))
        out, hstate, saved, _ = want
        for grad_hn in (g_hn, None):
            want_b = ops.gru_bwd_bf16(g_out, grad_hn, hstate, saved, w_hh, sl, b, t, hid, persistent=False)
            for rep in range(3):
                got_b = ops.gru_bwd_bf16(g_out, grad_hn, hstate, saved, w_hh, sl, b, t, hid, persistent=True)
                ops.check_persistent_status()
                for name, g, w in zip(('dxproj', 'dhproj', 'dh0', 'dhproj_bf'), got_b, want_b):
                    np.testing.assert_array_equal(g.float().cpu().numpy(), w.float().cpu().numpy(), err_msg='%s rep %d' % (name, rep))
            # the form GRUFn uses in bf16 mode: only the bf16 shadows are written
            dx_bf, dh_bf, dh0 = ops.gru_bwd_bf16(g_out, grad_hn, hstate, saved, w_hh, sl, b, t, hid, persistent=True, shadows_only=True)
            ops.check_persistent_status()
            assert torch.equal(dx_bf, want_b[0].to(torch.bfloat16)) and torch.equal(dh_bf, want_b[3]) and torch.equal(dh0, want_b[2])


@pytest.mark.parametrize('b,t,hid', [(64, 40, 512), (5, 37, 128), (33, 20, 256), (200, 9, 128), (12, 15, 384), (3, 1, 128)])
def test_lstm_persistent_vs_fp32_and_between_handoff_forms(b, t, hid):
    """The one-launch LSTM recurrence (csrc/lstm_persist.hip, bf16 matmul operands) against the exact-fp32 per-step kernels on
    the same inputs at 1e-2 relative (operand rounding 2^-9 per product, fp32 accumulation and cell), forward and backward,
    ragged lengths, initial states and gradients on the final states.  The hand-off protocol itself: the same-XCD form and
    the forced write-through form must give IDENTICAL bits, and so must repeated runs (a stale hand-off would not repeat)."""
    from morgana_amd import _lib
    rng = np.random.RandomState(hid + b)
    xproj = dev(rng.standard_normal((b, t, 4 * hid)).astype(np.float32))
    w_hh = dev((rng.uniform(-1, 1, (4 * hid, hid)) / np.sqrt(hid)).astype(np.float32))
    b_hh = dev(rng.uniform(-0.1, 0.1, 4 * hid).astype(np.float32))
    h0 = dev(rng.standard_normal((b, hid)).astype(np.float32) * 0.5)
    c0 = dev(rng.standard_normal((b, hid)).astype(np.float32) * 0.5)
    sl_np = rng.randint(1, t + 1, size=b).astype(np.int64)
    sl_np[0], sl_np[-1] = t, 1
    sl = dev(sl_np)
    live = (np.arange(t)[None, :] < sl_np[:, None])[:, :, None]
    assert ops.lstm_persist_ok(b, t, hid)
    want = ops.lstm_fwd(xproj, w_hh, b_hh, sl, h0, c0, b, t, hid)
    g_out = dev(rng.standard_normal((b, t, hid)).astype(np.float32))
    g_hn = dev(rng.standard_normal((b, hid)).astype(np.float32))
    g_cn = dev(rng.standard_normal((b, hid)).astype(np.float32))
    want_b = ops.lstm_bwd(g_out, g_hn, g_cn, want[2], want[3], w_hh, sl, b, t, hid)
    first = None
    try:
        for mode in (0, 1, 0):
            _lib.load().mg_set_tuning(2, mode)
            got = ops.lstm_fwd_bf16(xproj, w_hh, b_hh, sl, h0, c0, b, t, hid)
            # backward on the fp32 run's saved tensors, so that only the backward recurrence differs
            got_b = ops.lstm_bwd_bf16(g_out, g_hn, g_cn, want[2], want[3], w_hh, sl, b, t, hid)
            ops.check_persistent_status()
            arrays = [a.float().cpu().numpy() for a in got + got_b]
            if first is None:
                first = arrays
                for name, g, w in zip(('out', 'hstate', 'cstate', 'saved'), arrays[:4], want):
                    w = w.cpu().numpy()
                    if name == 'saved':
                        g, w = g * live, w * live
                    assert rel_err(g, w) < 1e-2, name
                assert np.array_equal(arrays[4], got[1].to(torch.bfloat16).float().cpu().numpy())       # bf16 shadow of hstate
                for i, n in enumerate(sl_np):
                    assert np.all(arrays[0][i, n:] == 0)
                for name, g, w in zip(('dgates', 'dh0', 'dc0'), arrays[5:8], want_b):
                    assert rel_err(g, w.cpu().numpy()) < 1e-2, name
                assert np.array_equal(arrays[8], got_b[0].to(torch.bfloat16).float().cpu().numpy())     # bf16 shadow of dgates
            else:
                for k, (g, w) in enumerate(zip(arrays, first)):
                    np.testing.assert_array_equal(g, w, err_msg='array %d, hand-off mode %d' % (k, mode))
    finally:
        _lib.load().mg_set_tuning(2, 0)


@pytest.mark.parametrize('b,t,i_dim,hid,n_layers', [(64, 50, 512, 512, 8), (20, 30, 24, 128, 3), (40, 25, 256, 256, 4), (9, 33, 40, 128, 2)])
def test_lstm_stack_wavefront_vs_chained_layers(b, t, i_dim, hid, n_layers):
    """functional.LSTMStackPersistFn (all layers' forward in one persistent launch, a wavefront over layer and time with the
    upper layers' input projection computed inside the step) against the same layers run one after the other: bf16 mode
    both, so equal up to the summation order of the input projection (1e-2 relative on outputs, final states and every
    gradient), and 2e-2 against the fp32 chain.  Ragged lengths with a full and a 1-step item; gradients on outputs and
    final states.  The same-XCD and the write-through hand-off forms and repeated runs must give identical bits."""
    from morgana_amd import _lib
    torch.manual_seed(hid + n_layers)
    x = torch.randn(b, t, i_dim, device=DEV, requires_grad=True)
    sl_np = np.random.RandomState(b).randint(1, t + 1, size=b).astype(np.int64)
    sl_np[0], sl_np[-1] = t, 1
    seq_len = dev(sl_np)
    params = []
    for l in range(n_layers):
        k = i_dim if l == 0 else hid
        params += [torch.randn(4 * hid, k, device=DEV) / k ** 0.5, torch.randn(4 * hid, hid, device=DEV) / hid ** 0.5,
                   torch.randn(4 * hid, device=DEV) * 0.1, torch.randn(4 * hid, device=DEV) * 0.1]
    params = [p.requires_grad_(True) for p in params]
    g_out = torch.randn(b, t, hid, device=DEV)
    g_hn, g_cn = torch.randn(n_layers, b, hid, device=DEV), torch.randn(n_layers, b, hid, device=DEV)
    assert F_hip.lstm_stack_persistent('bf16', b, t, hid, n_layers)

    # initial states for every layer in the smaller cases (a multi-layer nn.LSTM called with `hidden`), none in the first
    h0s = c0s = None
    if b != 64:
        h0s = (torch.randn(n_layers, b, hid, device=DEV) * 0.5).requires_grad_(True)
        c0s = (torch.randn(n_layers, b, hid, device=DEV) * 0.5).requires_grad_(True)

    def run(kind):
        for p in [x] + params + ([h0s, c0s] if h0s is not None else []):
            p.grad = None
        if kind == 'wavefront':
            out, hn, cn = F_hip.LSTMStackPersistFn.apply(x, seq_len, h0s, c0s, *params)
        else:
            out, hns, cns = x, [], []
            for l in range(n_layers):
                out, h, c = F_hip.LSTMFn.apply(kind, out.contiguous(), None if h0s is None else h0s[l:l + 1],
                                               None if c0s is None else c0s[l:l + 1], seq_len, *params[4 * l:4 * l + 4])
                hns.append(h)
                cns.append(c)
            hn, cn = torch.cat(hns, 0), torch.cat(cns, 0)
        ((out * g_out).sum() + (hn * g_hn).sum() + (cn * g_cn).sum()).backward()
        ops.check_persistent_status()
        extra = [h0s.grad, c0s.grad] if h0s is not None else []
        return [v.detach().cpu().numpy().copy() for v in (out, hn, cn, x.grad, *[p.grad for p in params], *extra)]

    want_bf, want_32 = run('bf16'), run('fp32')
    first = None
    try:
        for mode in (0, 1, 0):
            _lib.load().mg_set_tuning(2, mode)
            got = run('wavefront')
            if first is None:
                first = got
                for k, (g, w, w32) in enumerate(zip(got, want_bf, want_32)):
                    assert rel_err(g, w) < 1e-2, (k, 'vs bf16 chain')
                    assert rel_err(g, w32) < 2e-2, (k, 'vs fp32 chain')
                for i, n in enumerate(sl_np):
                    assert np.all(got[0][i, n:] == 0)
            else:
                for k, (g, w) in enumerate(zip(got, first)):
                    np.testing.assert_array_equal(g, w, err_msg='output %d, hand-off mode %d' % (k, mode))
        if b == 64:
            # the same under uneven load: a side stream keeps CUs and memory busy while the 512 workgroups of the wavefront start
            side = torch.cuda.Stream()
            junk = torch.randn(4096, 4096, device=DEV)
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                for _ in range(8):
                    junk @ junk
            got = run('wavefront')
            torch.cuda.synchronize()
            for k, (g, w) in enumerate(zip(got, first)):
                np.testing.assert_array_equal(g, w, err_msg='output %d under concurrent load' % k)
            # hidden units per workgroup - backward 16 (two workgroups per CU, tuning bit 0) instead of its default 32 (one per CU,
            # each hand-off tile read once), forward 32 (bit 1) instead of its default 16: the same products in the same order per
            # unit, identical bits
            for form in (2, 1, 3):
                _lib.load().mg_set_tuning(6, form)
                got = run('wavefront')
                for k, (g, w) in enumerate(zip(got, first)):
                    np.testing.assert_array_equal(g, w, err_msg='output %d, MG_TUNE_LSTM_BWD_STACK %d' % (k, form))
    finally:
        _lib.load().mg_set_tuning(2, 0)
        _lib.load().mg_set_tuning(6, 0)


@pytest.mark.parametrize('b,t,hid', [(5, 37, 64), (64, 50, 64), (1, 9, 128), (37, 21, 128)])
def test_gru_small_hidden_single_workgroup_vs_step_kernels(b, t, hid):
    """csrc/gru_small.hip (H = 64 / 128: a workgroup owns its items outright, W_hh in registers, state in LDS, one launch per
    direction, exact fp32) against the launch-per-step fp32 kernels on the same inputs: same products, summed in one chain per
    output instead of four partial chains - 1e-5 relative.  Ragged lengths, initial state, gradients on outputs and h_n."""
    from morgana_amd import _lib
    rng = np.random.RandomState(hid + b)
    xproj = dev(rng.standard_normal((b, t, 3 * hid)).astype(np.float32))
    w_hh = dev((rng.uniform(-1, 1, (3 * hid, hid)) / np.sqrt(hid)).astype(np.float32))
    b_hh = dev(rng.uniform(-0.1, 0.1, 3 * hid).astype(np.float32))
    h0 = dev(rng.standard_normal((b, hid)).astype(np.float32) * 0.5)
    sl_np = rng.randint(1, t + 1, size=b).astype(np.int64)
    sl_np[0] = t
    g_out = dev(rng.standard_normal((b, t, hid)).astype(np.float32))
    g_hn = dev(rng.standard_normal((b, hid)).astype(np.float32))
    lib = _lib.load()
    assert lib.mg_gru_small_supported(hid)
    for sl in (dev(sl_np), None):
        got = ops.gru_fwd(xproj, w_hh, b_hh, sl, h0, b, t, hid)
        got_b = ops.gru_bwd(g_out, g_hn, got[1], got[2], w_hh, sl, b, t, hid)
        lib.mg_set_tuning(3, 1)                       # the per-step kernels
        try:
            want = ops.gru_fwd(xproj, w_hh, b_hh, sl, h0, b, t, hid)
            want_b = ops.gru_bwd(g_out, g_hn, got[1], got[2], w_hh, sl, b, t, hid)
        finally:
            lib.mg_set_tuning(3, 0)
        for name, g, w in zip(('out', 'hstate', 'saved', 'dxproj', 'dhproj', 'dh0'), got + got_b, want + want_b):
            assert rel_err(g.cpu().numpy(), w.cpu().numpy()) < 1e-5, name
        if sl is not None:
            for i, n in enumerate(sl_np):
                assert torch.all(got[0][i, n:] == 0)


def test_gru_persistent_under_concurrent_load_and_edge_shapes():
    """Hand-off protocol of the persistent GRU kernels away from the quiet, lock-step case: (1) a side stream keeps the memory
    system and the CUs busy with large copies and GEMMs while the recurrence runs (uneven load: workgroups start late, polls
    and drains are delayed) - results must still EQUAL the per-step kernels; (2) T = 1, B = 1 and B = 256 (32 items per group,
    two MFMA row tiles)."""
    rng = np.random.RandomState(3)
    side = torch.cuda.Stream()
    junk_a = torch.randn(4096, 4096, device=DEV)
    junk_b = torch.empty(64 * 1024 * 1024, device=DEV)

    def case(b, t, hid, loaded):
        xproj = dev(rng.standard_normal((b, t, 3 * hid)).astype(np.float32))
        w_hh = dev((rng.uniform(-1, 1, (3 * hid, hid)) / np.sqrt(hid)).astype(np.float32))
        b_hh = dev(rng.uniform(-0.1, 0.1, 3 * hid).astype(np.float32))
        sl_np = rng.randint(1, t + 1, size=b).astype(np.int64)
        sl_np[0] = t
        sl = dev(sl_np)
        g_out = dev(rng.standard_normal((b, t, hid)).astype(np.float32))
        want = ops.gru_fwd_bf16(xproj, w_hh, b_hh, sl, None, b, t, hid, persistent=False)
        want_b = ops.gru_bwd_bf16(g_out, None, want[1], want[2], w_hh, sl, b, t, hid, persistent=False)
        torch.cuda.synchronize()
        if loaded:
            with torch.cuda.stream(side):
                for _ in range(6):
                    junk_b.copy_(junk_b.flip(0))
                    junk_a @ junk_a
        got = ops.gru_fwd_bf16(xproj, w_hh, b_hh, sl, None, b, t, hid, persistent=True)
        got_b = ops.gru_bwd_bf16(g_out, None, want[1], want[2], w_hh, sl, b, t, hid, persistent=True)
        torch.cuda.synchronize()
        ops.check_persistent_status()
        live = (np.arange(t)[None, :] < sl_np[:, None])[:, :, None]
        for name, g, w in zip(('out', 'hstate', 'saved', 'hstate_bf', 'dxproj', 'dhproj', 'dh0', 'dhproj_bf'), got + got_b, want + want_b):
            g, w = g.float().cpu().numpy(), w.float().cpu().numpy()
            if name == 'saved':
                g, w = g * live, w * live
            np.testing.assert_array_equal(g, w, err_msg='%s (B=%d T=%d H=%d loaded=%s)' % (name, b, t, hid, loaded))

    case(64, 200, 512, loaded=True)
    case(64, 200, 512, loaded=True)
    case(7, 1, 128, loaded=False)
    case(1, 33, 256, loaded=False)
    case(256, 6, 128, loaded=False)


@pytest.mark.parametrize('gru_handoff', [0, 1], indirect=True)
@pytest.mark.parametrize('b,t,hid', [(64, 40, 512), (5, 23, 256), (100, 7, 320), (1, 1, 256)])
def test_gru_persistent_fp32_equals_step_kernels(b, t, hid, gru_handoff):
    """fp32 parity mode in one launch per direction (gru_fwd_persist_f32_kernel / gru_bwd_persist_f32_kernel: fp32 hand-off tiles,
    exact-fp32 MFMA in the per-step kernels' block order, shared cell code) against the launch-per-step kernels: EQUAL bits on
    the live steps, in both hand-off forms, with ragged lengths, an initial state and a gradient on h_n."""
    rng = np.random.RandomState(hid + b)
    xproj = dev(rng.standard_normal((b, t, 3 * hid)).astype(np.float32))
    w_hh = dev((rng.uniform(-1, 1, (3 * hid, hid)) / np.sqrt(hid)).astype(np.float32))
    b_hh = dev(rng.uniform(-0.1, 0.1, 3 * hid).astype(np.float32))
    h0 = dev(rng.standard_normal((b, hid)).astype(np.float32) * 0.5)
    sl_np = rng.randint(1, t + 1, size=b).astype(np.int64)
    sl_np[0] = t
    g_out = dev(rng.standard_normal((b, t, hid)).astype(np.float32))
    g_hn = dev(rng.standard_normal((b, hid)).astype(np.float32))
    assert ops.gru_persist_f32_ok(b, t, hid)
    for sl in (dev(sl_np), None):
        live = np.ones((b, t, 1), bool) if sl is None else (np.arange(t)[None, :] < sl_np[:, None])[:, :, None]
        want = ops.gru_fwd(xproj, w_hh, b_hh, sl, h0, b, t, hid, persistent=False)
        want_b = ops.gru_bwd(g_out, g_hn, want[1], want[2], w_hh, sl, b, t, hid, persistent=False)
        for rep in range(2):
            got = ops.gru_fwd(xproj, w_hh, b_hh, sl, h0, b, t, hid, persistent=True)
            got_b = ops.gru_bwd(g_out, g_hn, want[1], want[2], w_hh, sl, b, t, hid, persistent=True)
            ops.check_persistent_status()
            for name, g, w in zip(('out', 'hstate', 'saved', 'dxproj', 'dhproj', 'dh0'), got + got_b, want + want_b):
                g, w = g.cpu().numpy(), w.cpu().numpy()
                if name == 'saved':
                    g, w = g * live, w * live
                np.testing.assert_array_equal(g, w, err_msg='%s rep %d' % (name, rep))


def test_gru_bf16_recurrence_rejects_bad_sizes():
    x = torch.zeros(2, 3, 3 * 96, device=DEV)
    assert not ops.gru_bf16_ok(96)
    with pytest.raises(ValueError, match='H % 128'):
        ops.gru_fwd_bf16(x, torch.zeros(3 * 96, 96, device=DEV), torch.zeros(3 * 96, device=DEV), None, None, 2, 3, 96)


@pytest.mark.parametrize('recurrence_bf16', [True, False])
def test_rnn_model_c4_shape_bf16_vs_oracle(recurrence_bf16):
    """GRU-512 model (BASELINE C4 widths) in bf16 precision, with the bf16-operand and with the exact-fp32 recurrence, against
    the fp32 numpy oracle at the bf16 tolerance: loss and predictions 2e-2, every gradient 5e-2 (relative L2)."""
    feats = synthetic.make_batch(8, (60, 120), out_dim=80, target_name='mcep', seed=11)
    state = synthetic.rnn_spss_state()
    want_loss, want_pred, want_grads = ref_cpu.rnn_forward_backward(state, feats)
    F_hip.set_recurrence_bf16(recurrence_bf16)
    try:
        model = _load_state(models.RNNSPSS(precision='bf16').to(DEV), state)
        loss, out = model(data.to_device(feats, DEV))
        loss.backward()
    finally:
        F_hip.set_recurrence_bf16(True)
    np.testing.assert_allclose(loss.item(), want_loss, rtol=RTOL_BF16)
    assert rel_err(out['pred_norm_mcep'].detach().cpu().numpy(), want_pred) < 2e-2
    for name, prm in model.named_parameters():
        assert rel_err(prm.grad.cpu().numpy(), want_grads[name]) < 5e-2, name


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_lstm_stack_skewed_equals_chained_layers(precision):
    """functional.LSTMStackFn (layers skewed in time, one launch per step for all of them, projections chunk by chunk) against
    the same layers run one after the other (LSTMFn): T well beyond the lag, ragged lengths, a length-0 tail chunk, gradients
    on the outputs and on the final states.  fp32 mode: the same products, summed in a different order by the stack's wide
    step kernels (whole contraction per wave instead of a 4-way split) - equal to 1e-5."""
    torch.manual_seed(5)
    b, t, i_dim, hid, n_layers = 5, 100, 24, 16, 3
    x = torch.randn(b, t, i_dim, device=DEV, requires_grad=True)
    seq_len = torch.tensor([100, 37, 64, 1, 99], device=DEV)
    params = []
    for l in range(n_layers):
        params += [torch.randn(4 * hid, i_dim if l == 0 else hid, device=DEV) * 0.3, torch.randn(4 * hid, hid, device=DEV) * 0.3,
                   torch.randn(4 * hid, device=DEV) * 0.1, torch.randn(4 * hid, device=DEV) * 0.1]
    params = [p.requires_grad_(True) for p in params]
    g_out = torch.randn(b, t, hid, device=DEV)
    g_hn, g_cn = torch.randn(n_layers, b, hid, device=DEV), torch.randn(n_layers, b, hid, device=DEV)

    def run(stacked):
        for p in [x] + params:
            p.grad = None
        if stacked:
            out, hn, cn = F_hip.LSTMStackFn.apply(precision, 32, x, seq_len, None, None, *params)
        else:
            out, hns, cns = x, [], []
            for l in range(n_layers):
                out, h, c = F_hip.LSTMFn.apply(precision, out.contiguous(), None, None, seq_len, *params[4 * l:4 * l + 4])
                hns.append(h)
                cns.append(c)
            hn, cn = torch.cat(hns, 0), torch.cat(cns, 0)
        ((out * g_out).sum() + (hn * g_hn).sum() + (cn * g_cn).sum()).backward()
        return [out.detach(), hn.detach(), cn.detach(), x.grad.clone()] + [p.grad.clone() for p in params]

    want, got = run(False), run(True)
    for w, g in zip(want, got):
        assert rel_err(g.cpu().numpy(), w.cpu().numpy()) < (1e-5 if precision == 'fp32' else RTOL_BF16)


# ------------------------------------------------------------------------ LSTM acoustic model (models/RNN_SPSS.py)
G12_STREAMS = (('lf0', 3, 'mse'), ('vuv', 1, 'sigmoid_bce'), ('mcep', 6, 'mse'), ('bap', 3, 'mse'))


def _stream_targets(feats, streams):
    return [feats[name] if kind != 'mse' else feats['normalised_%s_deltas' % name] for name, _, kind in streams]


def test_multi_stream_loss_golden(golden):
    """3 x mse + bce(sigmoid) over 4 (models/RNN_SPSS.py:120-139) in one kernel, against the reference's four losses."""
    g = golden('g12_lstm_acoustic.npz')
    feats = data.to_device(synthetic.make_acoustic_batch(5, (10, 30), lab_dim=20, counters_dim=4, streams=G12_STREAMS,
                                                         frames_per_phone=5.0, seed=1212), DEV)
    pred = dev(g['loss__pred']).requires_grad_(True)
    loss, prob = losses.multi_stream(pred, _stream_targets(feats, G12_STREAMS), [k for _, _, k in G12_STREAMS],
                                     feats['n_frames'], want_prob=True)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss__value'], rtol=1e-5)
    np.testing.assert_allclose(pred.grad.cpu().numpy(), g['loss__grad'], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(prob.cpu().numpy(), g['loss__vuv'], rtol=1e-5, atol=1e-30)


def test_multi_stream_loss_full_width_vs_oracle():
    """The shipped model's 199 columns (3 + 1 + 180 + 15), ragged lengths, one empty utterance -> NaN as in the reference."""
    feats = synthetic.make_acoustic_batch(6, (100, 333), seed=5)
    rng = np.random.RandomState(6)
    pred = (rng.standard_normal((6, int(feats['n_frames'].max()), 199)) * 1.5).astype(np.float32)
    kinds = [k for _, _, k in synthetic.ACOUSTIC_STREAMS]
    want_loss, want_grad = ref_cpu.multi_stream_loss(pred, _stream_targets(feats, synthetic.ACOUSTIC_STREAMS), kinds,
                                                     feats['n_frames'])
    d = data.to_device(feats, DEV)
    p = dev(pred).requires_grad_(True)
    loss, _ = losses.multi_stream(p, _stream_targets(d, synthetic.ACOUSTIC_STREAMS), kinds, d['n_frames'])
    loss.backward()
    np.testing.assert_allclose(loss.item(), want_loss, rtol=1e-5)
    assert rel_err(p.grad.cpu().numpy(), want_grad) < 1e-5
    seq = d['n_frames'].clone()
    seq[2] = 0
    loss, _ = losses.multi_stream(p, _stream_targets(d, synthetic.ACOUSTIC_STREAMS), kinds, seq)
    assert torch.isnan(loss)


@pytest.mark.parametrize('f,c', [(600, 9), (20, 4), (7, 1)])
def test_gather_concat_vs_numpy(f, c):
    """upsample_to_repetitions + torch.cat with frame-level counters (models/RNN_SPSS.py:76-81) as one pass: exact copies."""
    rng = np.random.RandomState(f)
    src = rng.standard_normal((50, f)).astype(np.float32)
    rows = rng.randint(-1, 50, size=777).astype(np.int32)
    extra = rng.standard_normal((777, c)).astype(np.float32)
    want = np.concatenate((np.where(rows[:, None] < 0, 0, src[np.maximum(rows, 0)]), extra), axis=1)
    got = ops.gather_concat(dev(src), dev(rows), dev(extra))
    assert np.array_equal(got.cpu().numpy(), want)
    got = ops.gather_concat(dev(src), dev(rows), dev(extra), out_bf16=True)
    assert got.shape == (777, ops.pad_ld(f + c)) and got.dtype == torch.bfloat16
    assert np.array_equal(got[:, :f + c].float().cpu().numpy(), torch.from_numpy(want).bfloat16().float().numpy())
    assert float(got[:, f + c:].float().abs().sum()) == 0.0


@pytest.mark.parametrize('fused', [True, False])
def test_lstm_acoustic_model_golden(golden, fused):
    """Linear / 3 x LSTM wrapper / Linear / Linear with counters concat and the 4-stream loss: loss, outputs, every gradient
    and 6 Adam steps against the reference (fp32 mode, north-star tolerance); bf16 GEMMs track the curve."""
    g = golden('g12_lstm_acoustic.npz')
    lab_dim, counters_dim, hidden, post, num_layers = [int(v) for v in g['model__dims'][:5]]
    dims = {'lf0': 3, 'vuv': 1, 'mcep': 6, 'bap': 3}
    state = synthetic.lstm_acoustic_state(seed=1213, input_dim=lab_dim + counters_dim, hidden=hidden, post=post,
                                          output_dim=13, num_layers=num_layers)
    feats = data.to_device(synthetic.make_acoustic_batch(5, (10, 30), lab_dim=lab_dim, counters_dim=counters_dim,
                                                         streams=G12_STREAMS, frames_per_phone=5.0, seed=1212), DEV)

    def build(precision):
        model = models.LSTMAcousticModel(lab_dim + counters_dim, dims, num_layers=num_layers, hidden_dim=hidden, post_dim=post,
                                         precision=precision, fused_upsample=fused, fused_loss=fused)
        return _load_state(model.to(DEV), state)

    model = build('fp32')
    loss, out = model(feats)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['model__loss_curve'][0], rtol=RTOL)
    for name in ('normalised_lf0_deltas', 'normalised_mcep_deltas', 'normalised_bap_deltas', 'vuv'):
        np.testing.assert_allclose(out[name].detach().cpu().numpy(), g['model__step1_' + name], rtol=1e-3, atol=1e-5)
    for name, prm in model.named_parameters():
        want = g['model__step1_grad__' + name]
        np.testing.assert_allclose(prm.grad.cpu().numpy(), want, rtol=1e-3, atol=1e-4 * np.abs(want).max(), err_msg=name)
    model.zero_grad()
    curve = _train(model, [feats], 6, lr=0.01)
    np.testing.assert_allclose(curve, g['model__loss_curve'], rtol=RTOL)
    curve = _train(build('bf16'), [feats], 6, lr=0.01)
    np.testing.assert_allclose(curve, g['model__loss_curve'], rtol=RTOL_BF16)


def test_lstm_acoustic_model_shipped_shape_vs_oracle():
    """The shipped layout (609 -> 512 -> LSTM-512 x 2 -> 256 -> 199; 2 of the 8 recurrent layers to keep the CPU oracle in
    seconds) on a ragged batch against the numpy oracle."""
    feats = synthetic.make_acoustic_batch(4, (40, 90), seed=21)
    state = synthetic.lstm_acoustic_state(num_layers=2)
    want_loss, want_pred, want_grads = ref_cpu.lstm_acoustic_forward_backward(state, feats, synthetic.ACOUSTIC_STREAMS, 2)
    model = _load_state(models.LSTMAcousticModel(num_layers=2, precision='fp32').to(DEV), state)
    loss, out = model(data.to_device(feats, DEV))
    loss.backward()
    np.testing.assert_allclose(loss.item(), want_loss, rtol=RTOL)
    assert rel_err(out['normalised_mcep_deltas'].cpu().numpy(), want_pred[..., 4:184]) < 1e-4
    for name, prm in model.named_parameters():
        assert rel_err(prm.grad.cpu().numpy(), want_grads[name]) < 1e-3, name


class _ToyGRUF0(models.StreamModel):
    """The shipped GRU F0 model's stream table (models.GRUF0Model) on the layer widths of golden G13 (the class itself fixes
    256 / 64 / 64 as the reference file does)."""

    def __init__(self, input_dim, d1, hid, post, out_dim, precision, fused):
        nn = torch.nn
        layers = utils.SequentialWithRecurrent(
            nn.Linear(input_dim, d1), nn.Sigmoid(), nn.Dropout(p=0.),
            utils.RecurrentCuDNNWrapper(nn.GRU(d1, hid, batch_first=True), precision=precision), nn.Dropout(p=0.),
            utils.RecurrentCuDNNWrapper(nn.GRU(hid, hid, batch_first=True), precision=precision), nn.Dropout(p=0.),
            utils.RecurrentCuDNNWrapper(nn.GRU(hid, hid, batch_first=True), precision=precision), nn.Dropout(p=0.),
            nn.Linear(hid, post), nn.Sigmoid(), nn.Dropout(p=0.), nn.Linear(post, out_dim), precision=precision)
        from morgana_amd import metrics
        streams = [models.Stream('lf0', out_dim, 'mse', ('LF0_RMSE_Hz', metrics.LF0Distortion, 'voiced_trajectory'))]
        models.StreamModel.__init__(self, layers, streams, fused_upsample=fused, fused_loss=False, generate=False)


@pytest.mark.parametrize('fused', [True, False])
def test_gru_f0_model_golden(golden, fused):
    """The shipped F0 model (models/f0_test_model.py): counters concat, three GRU wrappers; loss, prediction, every gradient
    and 6 Adam steps against the reference (fp32 mode)."""
    g = golden('g13_gru_f0.npz')
    lab_dim, counters_dim, d1, hid, post, out_dim = [int(v) for v in g['dims']]
    feats = data.to_device(synthetic.make_acoustic_batch(5, (10, 30), lab_dim=lab_dim, counters_dim=counters_dim,
                                                         streams=(('lf0', out_dim, 'mse'),), frames_per_phone=5.0, seed=1313), DEV)
    state = synthetic.gru_f0_state(seed=1314, input_dim=lab_dim + counters_dim, d1=d1, hidden=hid, post=post, output_dim=out_dim)
    model = _load_state(_ToyGRUF0(lab_dim + counters_dim, d1, hid, post, out_dim, 'fp32', fused).to(DEV), state)
    loss, out = model(feats)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss_curve'][0], rtol=RTOL)
    np.testing.assert_allclose(out['normalised_lf0_deltas'].detach().cpu().numpy(), g['step1_pred'], rtol=1e-3, atol=1e-5)
    for name, prm in model.named_parameters():
        want = g['step1_grad__' + name]
        np.testing.assert_allclose(prm.grad.cpu().numpy(), want, rtol=1e-3, atol=1e-4 * np.abs(want).max(), err_msg=name)
    model.zero_grad()
    np.testing.assert_allclose(_train(model, [feats], 6, lr=0.01), g['loss_curve'], rtol=RTOL)


def test_gru_f0_model_shipped_shape_vs_oracle():
    """The shipped widths (609 -> 256 -> GRU-64 x 3 -> 64 -> 3) on a ragged batch against the numpy oracle."""
    feats = synthetic.make_acoustic_batch(4, (40, 90), streams=(('lf0', 3, 'mse'),), seed=23)
    state = synthetic.gru_f0_state()
    want_loss, want_pred, want_grads = ref_cpu.gru_f0_forward_backward(state, feats)
    model = _load_state(models.GRUF0Model(precision='fp32').to(DEV), state)
    loss, out = model(data.to_device(feats, DEV))
    loss.backward()
    np.testing.assert_allclose(loss.item(), want_loss, rtol=RTOL)
    assert rel_err(out['normalised_lf0_deltas'].detach().cpu().numpy(), want_pred) < 1e-4
    for name, prm in model.named_parameters():
        assert rel_err(prm.grad.cpu().numpy(), want_grads[name]) < 1e-3, name


# ------------------------------------------------------------------------------------------- optimiser / EMA
def test_adam_and_ema_vs_oracle(golden):
    rng = np.random.RandomState(0)
    p0 = rng.standard_normal(10007).astype(np.float32)
    for wd in (0.0, 1e-2):
        p_ref = p0.copy()
        ref = ref_cpu.Adam([p_ref], lr=0.01, weight_decay=wd)
        prm = torch.nn.Parameter(dev(p0.copy()))
        opt = optim.Adam([prm], lr=0.01, weight_decay=wd)
        for step in range(5):
            gnp = rng.standard_normal(10007).astype(np.float32) * (10.0 ** rng.randint(-6, 2))
            ref.step([gnp])
            opt.zero_grad()
            prm.grad.copy_(dev(gnp))
            opt.step()
        np.testing.assert_allclose(prm.detach().cpu().numpy(), p_ref, rtol=1e-5, atol=1e-7)
    g = golden('g9_ema_lr.npz')
    shadow = dev(g['ema_shadow0'].copy())
    for params in g['ema_params_seq']:
        ops.ema_update(shadow, dev(params), float(g['ema_decay']))
    np.testing.assert_allclose(shadow.cpu().numpy(), g['ema_shadow_final'], rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------- streaming metrics
@pytest.mark.parametrize('name,cls,kind', [('mean_d5', 'DeviceMean', 'mean'), ('rmse_d5', 'RMSE', 'sqdiff'), ('mae_d1', 'MAE', 'absdiff'),
                                           ('distortion_d5', 'Distortion', 'root_sq'), ('mcd', 'MelCepDistortion', 'sqdiff'),
                                           ('lf0', 'LF0Distortion', 'sqdiff_voiced_exp'), ('f0', 'F0Distortion', 'sqdiff_voiced'),
                                           ('vuv_acc', 'DeviceMean', 'mean')])
def test_streaming_metrics_golden(golden, name, cls, kind):
    """morgana_amd.metrics device accumulators (csrc/metrics.hip) against the reference's sum / count / result for the same
    accumulate calls (tests/golden/g15_metrics.npz): count exact, sum and result to 1e-5."""
    from morgana_amd import metrics
    from test_oracle_golden import metric_calls
    g = golden('g15_metrics.npz')
    metric = getattr(metrics, cls)()
    for call in metric_calls(g, name):
        args = [dev(np.ascontiguousarray(call['target']))]
        if 'pred' in call:
            args.append(dev(np.ascontiguousarray(call['pred'])))
        if 'voiced' in call:
            args.append(dev(call['voiced']))
        metric.accumulate(*args, seq_len=dev(call['seq_len']) if 'seq_len' in call else None)
    assert float(metric.count) == float(g[name + '__count'])
    np.testing.assert_allclose(float(metric.sum), float(g[name + '__sum']), rtol=1e-5)
    np.testing.assert_allclose(float(metric.result()), float(g[name + '__result']), rtol=1e-5)


def test_streaming_metric_full_size_vs_oracle():
    """MelCepDistortion and LF0Distortion at the shipped model's batch (64 x 1000 frames, 60 mel-cepstra) against the numpy oracle."""
    from morgana_amd import metrics
    rng = np.random.RandomState(77)
    b, t = 64, 1000
    seq = rng.randint(300, t + 1, size=b).astype(np.int64)
    tm, pm = rng.standard_normal((b, t, 60)).astype(np.float32), rng.standard_normal((b, t, 60)).astype(np.float32)
    mcd = metrics.MelCepDistortion()
    mcd.accumulate(dev(tm), dev(pm), seq_len=dev(seq))
    s, c = ref_cpu.metric_sums('sqdiff', tm, pm, seq_len=seq, col0=1)
    np.testing.assert_allclose(float(mcd.result()), ref_cpu.metric_result('sqdiff', s, c), rtol=1e-5)
    lt, lp = (rng.standard_normal((b, t, 1)) * 0.2 + 5).astype(np.float32), (rng.standard_normal((b, t, 1)) * 0.2 + 5).astype(np.float32)
    voiced = rng.rand(b, t, 1) > 0.3
    lf0 = metrics.LF0Distortion()
    lf0.accumulate(dev(lt), dev(lp), dev(voiced), seq_len=dev(seq))
    s, c = ref_cpu.metric_sums('sqdiff_voiced_exp', lt, lp, voiced=voiced, seq_len=seq)
    assert float(lf0.count) == c
    np.testing.assert_allclose(float(lf0.result()), ref_cpu.metric_result('sqdiff_voiced_exp', s, c), rtol=1e-5)


# ------------------------------------------------------------------------------------------------------------ MLPG
MLPG_WINDOWS_5PT = ((0, 0, (1.0,)), (2, 2, (-0.2, -0.1, 0.0, 0.1, 0.2)), (1, 1, (1.0, -2.0, 1.0)))


@pytest.mark.parametrize('windows,pad,b,t,d', [(None, 0, 3, 26, 2), (None, 100, 5, 60, 3), (MLPG_WINDOWS_5PT, 4, 4, 33, 5),
                                               (((0, 0, (1.0,)),), 3, 2, 10, 4)])
def test_mlpg_vs_oracle(windows, pad, b, t, d):
    """mg_mlpg_f32 (csrc/mlpg.hip: band rows in one pass, banded LDL^T per system in float64) against the float64 CPU restatement
    of morgana/viz/synthesis.py:79-178: ragged lengths incl. a full and a 1-frame utterance, global and per-frame variances,
    default / 5-point / static-only windows, with and without burn-in padding.  1e-6 relative on the float32 trajectories
    (float32 rounding of the float64 solution; the summation order inside the band differs from bandmat's), 1e-11 on the
    float64 output; frames past seq_len exactly zero."""
    from morgana_amd import ops
    from morgana_amd.viz import synthesis
    n_win = 3 if windows is None else len(windows)
    rng = np.random.RandomState(b * t + pad)
    means = rng.standard_normal((b, t, n_win * d)).astype(np.float32)
    seq = rng.randint(2, t + 1, size=b).astype(np.int64)
    seq[0], seq[-1] = t, 1
    win = synthesis.DEFAULT_WINDOWS if windows is None else windows
    for variances in (rng.uniform(0.3, 2.0, n_win * d).astype(np.float32), rng.uniform(0.3, 2.0, means.shape).astype(np.float32)):
        want = ref_cpu.mlpg(means, variances, windows=windows, padding_size=pad, seq_len=seq)
        scale = np.abs(want).max()
        got = ops.mlpg(dev(means), dev(variances), win, padding_size=pad, seq_len=dev(seq))
        assert got.dtype == torch.float32 and tuple(got.shape) == (b, t, d)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=1e-6 * scale)
        got64 = ops.mlpg(dev(means), dev(variances), win, padding_size=pad, seq_len=dev(seq), out_dtype=torch.float64)
        np.testing.assert_allclose(got64.cpu().numpy(), want, rtol=0, atol=1e-11 * scale)
        for i, n in enumerate(seq):
            assert torch.all(got[i, n:] == 0)
    # no seq_len: every utterance runs to T
    want = ref_cpu.mlpg(means, variances, windows=windows, padding_size=pad)
    np.testing.assert_allclose(ops.mlpg(dev(means), dev(variances), win, padding_size=pad).cpu().numpy(), want, rtol=0,
                               atol=1e-6 * np.abs(want).max())


def test_mlpg_reference_signature():
    """morgana_amd.viz.synthesis.MLPG as the reference's call sites use it (models/f0_test_model.py:86-89: tensors, global variance,
    padding 100, seq_len tensor -> float tensor on the device; a single numpy sequence -> float64 array without the batch axis)."""
    from morgana_amd.viz import synthesis
    rng = np.random.RandomState(3)
    means = rng.standard_normal((6, 80, 3)).astype(np.float32)
    var = (rng.uniform(0.2, 1.5, 3) ** 2).astype(np.float32)
    seq = np.array([80, 41, 7, 80, 1, 63], np.int64)
    want = ref_cpu.mlpg(means, var, padding_size=100, seq_len=seq)
    got = synthesis.MLPG(dev(means), dev(var), padding_size=100, seq_len=dev(seq))
    assert isinstance(got, torch.Tensor) and got.is_cuda and got.dtype == torch.float32
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=1e-6 * np.abs(want).max())
    single = synthesis.MLPG(means[1, :41], var, padding_size=100)
    assert isinstance(single, np.ndarray) and single.dtype == np.float64 and single.shape == (41, 1)
    np.testing.assert_allclose(single, want[1, :41], rtol=0, atol=1e-11 * np.abs(want).max())
    with pytest.raises(ValueError, match='coefficients'):
        synthesis.MLPG(dev(means), dev(var), windows=[(1, 1, np.array([1.0]))] * 3)
    with pytest.raises(ValueError, match='windows'):
        synthesis.MLPG(dev(means[..., :2]), dev(var[:2]))


def test_mlpg_full_size_properties():
    """The LSTM acoustic model's streams at its batch (64 x 1000 frames, 60 mel-cepstra, padding 100; 3 840 systems of up to 1 200
    unknowns): a sample of systems against the oracle, and over ALL of them the defining property - the residual of the normal
    equations P x = b evaluated in float64 from the float64 trajectory - below 1e-9 of |b|."""
    from morgana_amd import ops
    from morgana_amd.viz import synthesis
    rng = np.random.RandomState(9)
    b, t, d = 64, 1000, 60
    means = rng.standard_normal((b, t, 3 * d)).astype(np.float32)
    var = rng.uniform(0.1, 2.0, 3 * d).astype(np.float32)
    seq = rng.randint(300, t + 1, size=b).astype(np.int64)
    seq[0] = t
    got = ops.mlpg(dev(means), dev(var), synthesis.DEFAULT_WINDOWS, padding_size=0, seq_len=dev(seq), out_dtype=torch.float64).cpu().numpy()
    padded = ops.mlpg(dev(means), dev(var), synthesis.DEFAULT_WINDOWS, padding_size=100, seq_len=dev(seq)).cpu().numpy()
    for i in (0, 17, 63):
        want = ref_cpu.mlpg(means[i:i + 1, :, [5, d + 5, 2 * d + 5]], var[[5, d + 5, 2 * d + 5]], padding_size=100, seq_len=seq[i:i + 1])
        np.testing.assert_allclose(padded[i, :, 5], want[0, :, 0], rtol=0, atol=1e-6 * np.abs(want).max())
    # residual of sum_w W_w^T diag(tau_w) (W_w x - mu_w) = 0 without padding, all systems at once
    tau = (1.0 / var).astype(np.float64).reshape(3, d)
    mu_tau = (means / var).astype(np.float64).reshape(b, t, 3, d)
    worst = 0.0
    for i in range(b):
        n = int(seq[i])
        x = got[i, :n]                                                     # (n, d)
        xp = np.concatenate((np.zeros((1, d)), x, np.zeros((1, d))), 0)
        delta = 0.5 * (xp[2:] - xp[:-2])
        ddelta = xp[2:] - 2.0 * xp[1:-1] + xp[:-2]
        res = [x * tau[0] - mu_tau[i, :n, 0], delta * tau[1] - mu_tau[i, :n, 1], ddelta * tau[2] - mu_tau[i, :n, 2]]
        rp = [np.concatenate((np.zeros((1, d)), r, np.zeros((1, d))), 0) for r in res]
        grad = res[0] + 0.5 * (rp[1][:-2] - rp[1][2:]) + (rp[2][:-2] - 2.0 * rp[2][1:-1] + rp[2][2:])
        worst = max(worst, np.abs(grad).max() / np.abs(mu_tau[i, :n]).max())
    assert worst < 1e-9, worst


def _mvn_denorm(x, params):
    return x * params['std_dev'] + params['mean']                        # morgana/data.py:536-538


@pytest.mark.parametrize('fused_loss', [True, False])
def test_lstm_acoustic_model_generation_and_metrics(fused_loss):
    """The shipped acoustic model's full step as the reference runs it under its builder (models/RNN_SPSS.py:84-129): the delta
    streams denormalised and turned into trajectories by MLPG (padding 100, global delta variances) and the four metrics
    accumulated inside loss().  Trajectories against the CPU restatement applied to the model's own delta outputs (1e-5 of the
    stream's scale: float32 denormalise + float32 trajectory), metrics against oracle.metric_sums on those trajectories (1e-4)."""
    feats_np = synthetic.make_acoustic_batch(4, (40, 90), seed=31, with_raw=True)
    model = _load_state(models.LSTMAcousticModel(num_layers=2, precision='fp32', fused_loss=fused_loss).to(DEV),
                        synthetic.lstm_acoustic_state(num_layers=2))
    synthetic.acoustic_normalisers(model, device=DEV)
    model.mode = 'train'
    model.metrics.reset_state('train')
    feats = data.to_device(feats_np, DEV)
    for _ in range(2):                                                   # two accumulate calls, as two steps of an epoch
        loss, out = model(feats)
    seq = feats_np['n_frames']
    vuv = out['vuv'].detach().cpu().numpy() > 0.5
    results = model.metrics.results_as_json_dict('train')
    kinds = {'lf0': ('LF0_RMSE_Hz', 'sqdiff_voiced_exp'), 'mcep': ('MCEP_distortion', 'sqdiff'), 'bap': ('BAP_distortion', 'root_sq')}
    for name, (metric_name, kind) in kinds.items():
        norm = model.normalisers[name]
        deltas = _mvn_denorm(out['normalised_%s_deltas' % name].detach().cpu().numpy(), norm.delta_params)
        want = ref_cpu.mlpg(deltas, norm.delta_params['std_dev'] ** 2, padding_size=100, seq_len=seq)
        got = out[name].cpu().numpy()
        assert got.shape == feats_np[name].shape
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-5 * np.abs(want).max(), err_msg=name)
        s, c = ref_cpu.metric_sums(kind, feats_np[name], got, voiced=vuv if name == 'lf0' else None, seq_len=seq,
                                   col0=1 if name == 'mcep' else 0)
        np.testing.assert_allclose(results[metric_name], ref_cpu.metric_result(kind, 2 * s, 2 * c), rtol=1e-4, err_msg=metric_name)
    s, c = ref_cpu.metric_sums('mean', (feats_np['vuv'] == vuv).astype(np.float32), seq_len=seq)
    np.testing.assert_allclose(results['VUV_accuracy'], ref_cpu.metric_result('mean', s, c), rtol=1e-6)
    # the three streams' MLPG launches run side by side on streams of their own (models.TRAJECTORY_STREAMS): the same bits as one after the other
    models.TRAJECTORY_STREAMS = False
    try:
        _, out_serial = model(feats)
    finally:
        models.TRAJECTORY_STREAMS = True
    for name in kinds:
        assert torch.equal(out_serial[name], out[name]), name
    # generation off: the training outputs only, no metric touched
    quiet = models.LSTMAcousticModel(num_layers=2, precision='fp32', generate=False).to(DEV)
    synthetic.acoustic_normalisers(quiet, device=DEV)
    assert 'lf0' not in quiet(feats)[1]


def test_gru_f0_model_generation_and_metric():
    """models/f0_test_model.py:78-105 as run under the builder: MLPG of the denormalised LF0 deltas and LF0_RMSE_Hz accumulated in
    loss() against the feature's own voicing flags; a model without normaliser parameters (plain synthetic runs) skips both."""
    feats_np = synthetic.make_acoustic_batch(6, (30, 120), streams=(('lf0', 3, 'mse'),), seed=37, with_raw=True)
    model = _load_state(models.GRUF0Model(precision='fp32').to(DEV), synthetic.gru_f0_state())
    feats = data.to_device(feats_np, DEV)
    assert set(model(feats)[1]) == {'normalised_lf0_deltas'}
    synthetic.acoustic_normalisers(model, device=DEV)
    model.mode = 'train'
    model.metrics.reset_state('train')
    loss, out = model(feats)
    loss.backward()
    norm = model.normalisers['lf0']
    deltas = _mvn_denorm(out['normalised_lf0_deltas'].detach().cpu().numpy(), norm.delta_params)
    want = ref_cpu.mlpg(deltas, norm.delta_params['std_dev'] ** 2, padding_size=100, seq_len=feats_np['n_frames'])
    np.testing.assert_allclose(out['lf0'].cpu().numpy(), want, rtol=0, atol=1e-5 * np.abs(want).max())
    assert not out['lf0'].requires_grad
    s, c = ref_cpu.metric_sums('sqdiff_voiced_exp', feats_np['lf0'], out['lf0'].cpu().numpy(), voiced=feats_np['vuv'] > 0.5,
                               seq_len=feats_np['n_frames'])
    np.testing.assert_allclose(model.metrics.results_as_json_dict('train')['LF0_RMSE_Hz'],
                               ref_cpu.metric_result('sqdiff_voiced_exp', s, c), rtol=1e-4)
    with pytest.raises(ValueError, match='No collection'):               # mode unset, as in the reference outside its builder
        model.mode = ''
        model(feats)


# ------------------------------------------------------------------------------------------ phone-rate first layer
def _ragged_rows(rng, b, p, t):
    """Frame -> table-row map as mg_upsample_index writes it: utterance i's phones repeated dur times, -1 past its end."""
    rows = np.full((b, t), -1, np.int32)
    for i in range(b):
        dur = rng.randint(0, 5, size=p)
        dur[rng.randint(p)] += 3
        idx = np.repeat(np.arange(p), dur)[:t]
        rows[i, :len(idx)] = i * p + idx
    return rows


@pytest.mark.parametrize('bf16', [True, False])
def test_phone_rate_kernels_vs_numpy(bf16):
    """mg_segment_bounds / mg_segment_sum (csrc/phone_rate.hip) against numpy on a ragged map with empty phones and padding frames:
    bounds and the mapped rows exact; segment sums of bf16 / fp32 frame rows in fp32, padding frames collected in the extra rows
    (their total checked), also when the map already carries the pad row instead of -1."""
    from morgana_amd import ops
    rng = np.random.RandomState(41 + bf16)
    b, p, t, n = 7, 9, 40, 72
    rows = _ragged_rows(rng, b, p, t)
    flat = rows.reshape(-1)
    r_tab = b * p
    seg_dev, mapped = ops.segment_bounds(dev(flat), r_tab, pad_row=r_tab)
    np.testing.assert_array_equal(mapped.cpu().numpy(), np.where(flat < 0, r_tab, flat))
    seg = ops.segment_bounds(dev(flat), r_tab).cpu().numpy()
    np.testing.assert_array_equal(seg, seg_dev.cpu().numpy())
    for r in range(r_tab):
        where = np.nonzero(flat == r)[0]
        want = (where[0], where[-1] + 1) if len(where) else (0, 0)
        assert (seg[0, r], seg[1, r]) == want
    grad = rng.standard_normal((b * t, n)).astype(np.float32)
    g_dev = dev(grad).to(torch.bfloat16) if bf16 else dev(grad)
    grad = g_dev.float().cpu().numpy()
    extra = 5
    sums = ops.segment_sum(g_dev, dev(flat), dev(seg), r_tab, n, extra=extra).float().cpu().numpy()
    assert sums.shape == (r_tab + extra, n)
    want_rows = np.stack([grad[flat == r].sum(0) for r in range(r_tab)])
    np.testing.assert_allclose(sums[:r_tab], want_rows, rtol=2 ** -7 if bf16 else 1e-5, atol=1e-5)
    np.testing.assert_allclose(sums[r_tab:].sum(0), grad[flat < 0].sum(0), rtol=0, atol=(0.15 if bf16 else 1e-4))
    again = ops.segment_sum(g_dev, mapped, dev(seg), r_tab, n, extra=extra)
    assert torch.equal(again.float().cpu(), torch.from_numpy(sums))


def test_phone_rate_first_layer_equals_frame_rate():
    """The F0Model step (bf16 mode) on a ragged batch (padding frames present) run at phone rate - every layer once per phone row, the
    prediction repeated instead of the input, the masked MSE of a phone's frames reduced to W (p - ybar)^2 + const - against the same
    step with every product at frame rate (MORGANA_PHONE_RATE=0, the reference's order of operations): the prediction is EQUAL (per
    row the same fp32 dot products, bias adds, sigmoids and bf16 roundings in the same kernels), the loss agrees to 1e-5 (the same
    squared errors, grouped by phone) and every gradient within 2e-2 relative L2 (frame gradients are summed before instead of after
    the bf16 roundings of the backward chain)."""
    from morgana_amd import ops
    feats = data.to_device(synthetic.make_batch(24, (150, 400), seed=5), DEV)

    def run(phone_rate):
        old = ops.PHONE_RATE
        ops.PHONE_RATE = phone_rate
        try:
            model = _load_state(models.F0Model(precision='bf16').to(DEV), synthetic.f0_model_state())
            loss, out = model(feats)
            loss.backward()
            return loss.item(), out['pred_norm_lf0'].detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in model.named_parameters()}
        finally:
            ops.PHONE_RATE = old

    loss_p, pred_p, grads_p = run(True)
    loss_f, pred_f, grads_f = run(False)
    np.testing.assert_allclose(loss_p, loss_f, rtol=1e-5)
    np.testing.assert_array_equal(pred_p, pred_f)
    for name in grads_f:
        assert rel_err(grads_p[name], grads_f[name]) < 2e-2, name

    # no seq_len: every frame counts, padding frames included (their prediction is the stack's value on a zero row) - the extra
    # table rows carry their loss terms and gradients
    def run_unmasked(phone_rate):
        old = ops.PHONE_RATE
        ops.PHONE_RATE = phone_rate
        try:
            model = _load_state(models.F0Model(precision='bf16').to(DEV), synthetic.f0_model_state())
            frames = utils.upsample_to_repetitions(feats['normalised_lab'], feats['dur'], max_len=feats['normalised_lf0'].shape[1],
                                                   fused=True)
            loss, pred = model.layers.forward_mse(frames, feats['normalised_lf0'], seq_len=None)
            loss.backward()
            return loss.item(), pred.detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in model.named_parameters()}
        finally:
            ops.PHONE_RATE = old

    loss_p, pred_p, grads_p = run_unmasked(True)
    loss_f, pred_f, grads_f = run_unmasked(False)
    np.testing.assert_allclose(loss_p, loss_f, rtol=1e-5)
    np.testing.assert_array_equal(pred_p, pred_f)
    for name in grads_f:
        assert rel_err(grads_p[name], grads_f[name]) < 2e-2, name


def test_graphed_train_step_equals_eager_steps():
    """graphs.GraphedTrainStep (zero_grad, forward, backward and the Adam update captured once as a HIP graph and replayed; Adam's
    step-dependent scalars read from device memory) against the same number of eager steps: same kernels in the same order on the
    same data, so parameters, both Adam moments and the loss must be EQUAL bit for bit - also across a learning-rate change."""
    from morgana_amd import graphs, optim
    feats = data.to_device(synthetic.make_batch(32, 200, seed=8), DEV)

    def fresh():
        model = _load_state(models.F0Model(precision='bf16').to(DEV), synthetic.f0_model_state())
        return model, optim.Adam(model.parameters(), lr=0.01)

    model_e, opt_e = fresh()
    losses_e = []
    for i in range(9):
        if i == 7:
            opt_e.param_groups[0]['lr'] = 0.003
        opt_e.zero_grad()
        loss, _ = model_e(feats)
        loss.backward()
        opt_e.step()
        losses_e.append(loss.item())
    model_g, opt_g = fresh()
    step = graphs.GraphedTrainStep(model_g, opt_g, feats, warmup=3)          # 3 eager steps, then the capture (which runs nothing)
    assert step.steps_done == 3
    losses_g = []
    for i in range(3, 9):                        # no host synchronisation inside the loop: the host runs ahead of the replays
        if i == 7:
            opt_g.param_groups[0]['lr'] = 0.003
        losses_g.append(step().clone())
    assert [v.item() for v in losses_g] == losses_e[3:]
    flat_e, flat_g = opt_e.flat_buffers(), opt_g.flat_buffers()
    assert flat_e['step'] == flat_g['step'] == 9
    for key in ('param', 'exp_avg', 'exp_avg_sq'):
        assert torch.equal(flat_e[key], flat_g[key]), key


@pytest.mark.parametrize('phone_rate', [True, False])
def test_graphed_step_defers_the_tail(phone_rate, monkeypatch):
    """A C2 step captured whole (graphs.GraphedTrainStep) leaves the last two small jobs of the phone-rate forward - the repeated
    prediction, the fused tail's slab sum - to rider blocks at the end of the backward's first launch
    (mg_linear_wgrad_dgrad_expand_bf16 instead of mg_expand_column_reduce_f32 + mg_linear_wgrad_dgrad_bf16: five launches per step
    instead of six); the eager loop keeps the separate launch, where forward outputs must be complete when forward returns.  Losses,
    predictions, parameters and Adam moments of the two: EQUAL bit for bit."""
    from morgana_amd import _lib, graphs, optim
    monkeypatch.setattr(ops, 'PHONE_RATE', phone_rate)       # False: the reference's order of operations - there the tail's reduce launch
    feats = data.to_device(synthetic.make_batch(256, 1000, seed=9), DEV)      # (mg_slab_reduce_f32 inside mg_f0_l2tail_bf16) is what goes
    data.add_bf16_table(feats)

    def fresh():
        model = _load_state(models.F0Model(precision='bf16').to(DEV), synthetic.f0_model_state())
        return model, optim.Adam(model.parameters(), lr=0.01, fused_loop=True)

    model_e, opt_e = fresh()
    losses_e = []
    for i in range(5):
        opt_e.zero_grad()
        loss, out_e = model_e(feats)
        F_hip.backward(loss)
        opt_e.step()
        losses_e.append(loss.item())
    pred_e = (next(iter(out_e.values())) if isinstance(out_e, dict) else out_e).clone()
    model_g, opt_g = fresh()
    _lib.CALL_LOG = []
    try:
        step = graphs.GraphedTrainStep(model_g, opt_g, feats, warmup=2)
        log = list(_lib.CALL_LOG)
    finally:
        _lib.CALL_LOG = None
    first = 'mg_phone_front_linear_fwd_bf16' if phone_rate else 'mg_upsample_index_maps'
    first = first if first in log else [c for c in log if 'upsample' in c][0]
    captured = log[len(log) - [c for c in reversed(log)].index(first) - 1:]      # the calls of the captured step
    # one rank, fused loop: the two jobs ride in the UPDATE launch's first blocks (mg_adam_tail), the tail's slabs are a source of its
    # plan; a rank whose gradients must be complete before the update takes mg_linear_wgrad_dgrad_expand_bf16 (riders behind the pair grid)
    if phone_rate:
        assert 'mg_expand_column_reduce_f32' not in captured and 'mg_linear_wgrad_dgrad_bf16' in captured, captured
        assert 'mg_expand_column_reduce_f32' in log                   # the two eager warm-up steps keep their own launch
    else:
        assert 'mg_f0_l2tail_slabs_bf16' in captured and 'mg_f0_l2tail_bf16' not in captured, captured
        assert 'mg_f0_l2tail_bf16' in log
    losses_g = [step().clone() for _ in range(3)]
    assert [v.item() for v in losses_g] == losses_e[2:]
    pred_g = next(iter(step.output.values())) if isinstance(step.output, dict) else step.output
    assert torch.equal(pred_g, pred_e)
    for key in ('param', 'exp_avg', 'exp_avg_sq'):
        assert torch.equal(opt_e.flat_buffers()[key], opt_g.flat_buffers()[key]), key


@pytest.mark.parametrize('phone_rate', [True, False])
def test_multi_step_replay_equals_single_step_replays(phone_rate, monkeypatch):
    """graphs.GraphedTrainStep(steps_per_replay=3): three training steps captured into ONE graph (each update reads its own slot of the
    scalars staged by one launch, optim.Adam.advance(3)), two batches alternating inside the replay, against single-step replays on
    the same sequence of batches: losses of every step, parameters and both Adam moments EQUAL bit for bit, at both orders of
    operations, across a learning-rate change between replays."""
    from morgana_amd import graphs, optim, ops
    monkeypatch.setattr(ops, 'PHONE_RATE', phone_rate)
    b0 = data.to_device(synthetic.make_batch(32, 200, seed=8), DEV)
    b1 = data.to_device(synthetic.make_batch(32, 200, seed=9), DEV)
    seq = [b0, b1, b0]

    def fresh():
        model = _load_state(models.F0Model(precision='bf16').to(DEV), synthetic.f0_model_state())
        return model, optim.Adam(model.parameters(), lr=0.01)

    model_1, opt_1 = fresh()
    singles = [graphs.GraphedTrainStep(model_1, opt_1, b, warmup=(2 if i == 0 else 0)) for i, b in enumerate((b0, b1))]
    losses_1 = []
    for rep in range(2):
        if rep == 1:
            opt_1.param_groups[0]['lr'] = 0.004
        for b in seq:
            losses_1.append(singles[0 if b is b0 else 1]().clone())
    model_k, opt_k = fresh()
    multi = graphs.GraphedTrainStep(model_k, opt_k, seq, warmup=2, steps_per_replay=3)      # the same two eager steps on b0 first
    losses_k = []
    for rep in range(2):
        if rep == 1:
            opt_k.param_groups[0]['lr'] = 0.004
        multi()
        losses_k += [v.clone() for v in multi.losses]
    assert multi.steps_done == 8
    assert [v.item() for v in losses_k] == [v.item() for v in losses_1]
    flat_1, flat_k = opt_1.flat_buffers(), opt_k.flat_buffers()
    assert flat_1['step'] == flat_k['step'] == 8
    for key in ('param', 'exp_avg', 'exp_avg_sq'):
        assert torch.equal(flat_1[key], flat_k[key]), key


def test_dgrad_with_table_gathered_sigmoid_outputs():
    """mg_linear_dgrad_gathered_bf16 (sigmoid outputs read from the per-phone table through a row map) against mg_linear_dgrad_bf16
    on the materialised frame-rate activation: same kernel, same arithmetic - EQUAL."""
    from morgana_amd import ops
    rng = np.random.RandomState(12)
    m, n, k, r_tab = 4096, 128, 512, 300
    dy = dev(rng.standard_normal((m, n)).astype(np.float32) * 0.1).to(torch.bfloat16)
    wt = dev(rng.standard_normal((k, n)).astype(np.float32) * 0.05).to(torch.bfloat16)
    table = torch.sigmoid(dev(rng.standard_normal((r_tab, k)).astype(np.float32))).to(torch.bfloat16)
    rows = dev(np.sort(rng.randint(0, r_tab, size=m)).astype(np.int32))
    want = ops.linear_dgrad_bf16(dy, m, n, wt, k, table[rows.long()].contiguous())
    got = ops.linear_dgrad_gathered_bf16(dy, m, n, wt, k, table, rows)
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        ops.linear_dgrad_gathered_bf16(dy[:100], 100, n, wt, k, table, rows[:100])         # below the wide-tile kernel's sizes


def test_experiment_builder_graph_replay_equals_eager_loop():
    """ExperimentBuilder(use_graphs=True): batches of a repeated shape are replayed as HIP graphs (first occurrence eager, second
    captured, later ones copied into the captured buffers), a batch of another shape in between runs eagerly - against the eager loop
    on the same batches: epoch losses and final parameters EQUAL, with a per-batch Noam learning-rate schedule."""
    from morgana_amd import experiment_builder
    batches = [data.to_device(synthetic.make_batch(16, 120, seed=40 + i), DEV) for i in range(5)]
    batches.insert(3, data.to_device(synthetic.make_batch(8, 90, seed=77), DEV))

    def train(use_graphs):
        torch.manual_seed(3)
        builder = experiment_builder.ExperimentBuilder(models.F0Model, dict(precision='bf16'), learning_rate=0.01,
                                                       lr_schedule_name='noam', lr_schedule_kwargs=dict(warmup_steps=4),
                                                       device=DEV, end_epoch=2, use_graphs=use_graphs)
        _load_state(builder.model, synthetic.f0_model_state())
        history = builder.run_train(batches)
        return history, {k: v.detach().clone() for k, v in builder.model.named_parameters()}

    hist_e, params_e = train(False)
    hist_g, params_g = train(True)
    assert hist_g == hist_e
    for name in params_e:
        assert torch.equal(params_g[name], params_e[name]), name


def test_graph_replay_survives_a_larger_shape_between_replays():
    """The per-layer slab buffers of the weight-gradient GEMMs grow with the row count.  A graph captured at the smaller shape has the
    old buffer's address baked into its weight-gradient and update nodes, so an outgrown buffer must stay alive (ops._slab_buffer
    retires it; ADVICE round 2 found it dropped: the next tensor took its memory and the replay wrote 48-256 slabs over it).  Shapes at
    slab-taking row counts (>= 4096): 16 x 300 twice (second one captured), 32 x 1000 (reallocates), then 16 x 300 replayed, with a
    burst of allocations in between that would land in freed memory - against the eager loop: losses and parameters EQUAL."""
    from morgana_amd import experiment_builder
    small = [data.to_device(synthetic.make_batch(16, 300, seed=60 + i), DEV) for i in range(4)]
    big = data.to_device(synthetic.make_batch(32, 1000, seed=70), DEV)
    batches = [small[0], small[1], big, small[2], small[3]]

    def train(use_graphs):
        torch.manual_seed(3)
        builder = experiment_builder.ExperimentBuilder(models.F0Model, dict(precision='bf16'), learning_rate=0.01, device=DEV,
                                                       end_epoch=1, use_graphs=use_graphs)
        _load_state(builder.model, synthetic.f0_model_state())
        optimizer = builder.make_optimizer()
        losses = []
        for i, feats in enumerate(batches):
            losses.append(builder.train_epoch([feats], optimizer))
            if i == 2:                                     # whatever was freed by the larger shape gets a new owner, filled with NaN
                junk = [torch.full((1 << 20,), float('nan'), device=DEV) for _ in range(64)]
                del junk
        return losses, {k: v.detach().clone() for k, v in builder.model.named_parameters()}

    old = ops.PHONE_RATE
    for phone_rate in (True, False):
        ops.PHONE_RATE = phone_rate
        try:
            loss_e, params_e = train(False)
            loss_g, params_g = train(True)
        finally:
            ops.PHONE_RATE = old
        assert loss_g == loss_e, phone_rate
        for name in params_e:
            assert torch.equal(params_g[name], params_e[name]), (phone_rate, name)


def test_graph_captured_after_a_no_grad_forward_keeps_every_operand_copy_current():
    """a model with more bf16 operand copies than the update kernel's plan holds (ADAM_MAX_SHADOWS = 8; here eleven
    Linear layers), a validation pass (no_grad forward: every copy stamped current) right before the step that is captured.  The
    copies past the plan used to be left to the next forward's version stamps - which a graph replay never moves: from the second
    replay on those layers multiplied by stale bf16 weights.  Now the update re-casts them itself (one batched launch), captured or
    not: the graph loop must train bit for bit as the eager loop, and a copy allocated AFTER the capture must not be taken for
    current after a replay."""
    from morgana_amd import experiment_builder
    same = [data.to_device(synthetic.make_batch(16, 300, seed=90 + i), DEV) for i in range(6)]

    def train(use_graphs):
        torch.manual_seed(5)
        builder = experiment_builder.ExperimentBuilder(models.F0Model, dict(precision='bf16', hidden_dims=(512,) * 8 + (128, 32)),
                                                       learning_rate=0.01, device=DEV, end_epoch=1, use_graphs=use_graphs)
        optimizer = builder.make_optimizer()
        losses = []
        for i, feats in enumerate(same):
            losses.append(builder.train_epoch([feats], optimizer))
            with torch.no_grad():                          # a validation pass between the steps (stamps every copy current)
                builder.model(feats)
        n_copies = sum(1 for p in builder.model.parameters() if getattr(p, '_mg_shadow', None) is not None)
        return losses, {k: v.detach().clone() for k, v in builder.model.named_parameters()}, n_copies, builder

    loss_e, params_e, n_e, _ = train(False)
    loss_g, params_g, n_g, builder = train(True)
    assert n_g > 8 and n_e == n_g
    assert loss_g == loss_e
    for name in params_e:
        assert torch.equal(params_g[name], params_e[name]), name
    # every copy equals a fresh cast of its weight after the last replay
    for p in builder.model.parameters():
        sh = getattr(p, '_mg_shadow', None)
        if sh is not None:
            assert torch.equal(sh['plain'][:, :p.shape[1]], p.detach().to(torch.bfloat16))
            if sh['t'] is not None:
                assert torch.equal(sh['t'][:, :p.shape[0]], p.detach().t().to(torch.bfloat16))


def test_deep_stack_with_more_slab_sources_than_the_update_plan_holds():
    """A 600-512-512-512-128-32-1 stack registers one split-M slab source per leading layer plus the tail: five, and the update
    kernel's plan holds ADAM_MAX_SLABS = 4 (optimizer.step() raised on the first step).  The surplus source is summed
    by a reduce launch of its own (optim.Adam.defer_slabs); the fused loop must train exactly as the plain loop."""
    feats = data.to_device(synthetic.make_batch(16, 400, seed=21), DEV)
    results = []
    for fused in (False, True):
        torch.manual_seed(0)
        model = models.F0Model(hidden_dims=(512, 512, 512, 128, 32), precision='bf16').to(DEV)
        opt = optim.Adam(model.parameters(), lr=0.01, fused_loop=fused)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            loss, _ = model(feats)
            F_hip.backward(loss)
            opt.step()
            losses.append(loss.item())
        results.append((losses, [p.detach().clone() for p in model.parameters()]))
    assert results[0][0] == results[1][0]
    for a, b in zip(results[0][1], results[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('precision', ['bf16', 'fp32'])
def test_phone_rate_gru_input_equals_frame_rate(precision):
    """RNN_SPSS layout (Linear-512 + Sigmoid on the upsampled labels, then GRU-512) with the Linear and the GRU's input projection run
    once per phone row (utils.PhoneTable, GRUFn with a row map; ragged batch, padding frames present) against the same step with
    every product at frame rate (MORGANA_PHONE_RATE=0): per row the same GEMM arithmetic, so loss and prediction are EQUAL; gradients
    within 2e-2 (bf16: the gate gradients are summed per phone before the bf16 GEMMs) / 1e-4 (fp32) relative L2."""
    from morgana_amd import ops
    feats = data.to_device(synthetic.make_batch(32, (150, 260), out_dim=80, target_name='mcep', seed=9, frames_per_phone=5.0), DEV)

    def run(phone_rate):
        old = ops.PHONE_RATE
        ops.PHONE_RATE = phone_rate
        try:
            model = _load_state(models.RNNSPSS(precision=precision).to(DEV), synthetic.rnn_spss_state())
            loss, out = model(feats)
            loss.backward()
            ops.check_persistent_status()
            return loss.item(), out['pred_norm_mcep'].detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in model.named_parameters()}
        finally:
            ops.PHONE_RATE = old

    n_src = feats['normalised_lab'].shape[0] * feats['normalised_lab'].shape[1]
    old_flag, ops.PHONE_RATE = ops.PHONE_RATE, True          # the batch must be one the phone-rate form takes (whatever the env says)
    try:
        assert ops.phone_rate_gru_ok(n_src, feats['normalised_mcep'].shape[0] * feats['normalised_mcep'].shape[1], 512)
    finally:
        ops.PHONE_RATE = old_flag
    loss_p, pred_p, grads_p = run(True)
    loss_f, pred_f, grads_f = run(False)
    assert loss_p == loss_f
    np.testing.assert_array_equal(pred_p, pred_f)
    for name in grads_f:
        assert rel_err(grads_p[name], grads_f[name]) < (2e-2 if precision == 'bf16' else 1e-4), name


@pytest.mark.parametrize('fused_tail', [True, False])
def test_phone_rate_fp32_stack_equals_frame_rate(fused_tail, monkeypatch):
    """fp32 parity mode of the F0Model: the stack ends in the linear run, so it runs on the phone rows - against MORGANA_PHONE_RATE=0
    on a ragged batch.  Either way - the fused exact tail (mg_f0_tail_rows_f32, the default: on phone rows with per-phone statistics,
    on frame rows with the loss's own weights) or the generic layers (utils.F0_TAIL_F32 off) - a row goes through the same kernels
    with the same arithmetic at both rates, so the prediction is EQUAL and the loss agrees to 1e-6; gradients to 1e-4 (frame
    gradients summed per phone before the GEMMs instead of inside them)."""
    from morgana_amd import ops
    monkeypatch.setattr(utils, 'F0_TAIL_F32', fused_tail)
    feats = data.to_device(synthetic.make_batch(24, (150, 400), seed=6), DEV)

    def run(phone_rate):
        old = ops.PHONE_RATE
        ops.PHONE_RATE = phone_rate
        try:
            model = _load_state(models.F0Model(precision='fp32').to(DEV), synthetic.f0_model_state())
            loss, out = model(feats)
            loss.backward()
            return loss.item(), out['pred_norm_lf0'].detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in model.named_parameters()}
        finally:
            ops.PHONE_RATE = old

    loss_p, pred_p, grads_p = run(True)
    loss_f, pred_f, grads_f = run(False)
    np.testing.assert_allclose(loss_p, loss_f, rtol=1e-6)
    np.testing.assert_array_equal(pred_p, pred_f)
    for name in grads_f:
        assert rel_err(grads_p[name], grads_f[name]) < 1e-4, name


def test_upsample_index_maps_match_separate_launches():
    """mg_upsample_index_maps (frame map, pad-mapped map and per-phone frame runs in one launch) against mg_upsample_index +
    mg_segment_bounds: EQUAL, with zero durations, a capped frame axis (phones cut or dropped by t_cap) and padding frames."""
    from morgana_amd import ops
    rng = np.random.RandomState(17)
    dur = rng.randint(0, 6, size=(9, 23)).astype(np.int64)
    dur[3] = 0
    dur[4, :5] = 40
    for t_cap in (int(dur.sum(1).max()), 60):
        _, rows = ops.upsample_index(dev(dur), t_cap)
        seg, mapped = ops.segment_bounds(rows.reshape(-1), dur.size, pad_row=dur.size)
        rows2, mapped2, seg2 = ops.upsample_index_maps(dev(dur), t_cap)
        assert torch.equal(rows2, rows) and torch.equal(mapped2.reshape(-1), mapped) and torch.equal(seg2, seg)


@pytest.mark.parametrize('case', ['ragged', 'c2', 'wide', 'many'])
def test_phone_front_equals_the_separate_launches(case):
    """mg_phone_front (frame map + per-phone loss statistics, one job per utterance) against mg_upsample_index_maps +
    mg_phone_target_stats: rows, the pad-mapped rows, the frame runs, ybar and weight EQUAL; the loss's constant term (partial sums
    grouped per utterance instead of per 16 rows) to 1e-6.  'ragged': zero durations, an empty utterance, totals short of and beyond the
    frame axis, seq_len cutting into live frames; 'c2': the headline shape; 'wide': more phones than threads per utterance; 'many':
    many short utterances (a rider block works dozens of jobs off one staged batch).
    Then mg_phone_front_linear_fwd_bf16: the same outputs plus the first layer's GEMM EQUAL to mg_linear_fwd_bf16, in one grid
    (where the GEMM leaves CUs idle: 'c2') and as its two launches (MG_TUNE_PROBE = 66)."""
    from morgana_amd import ops, _lib
    lib = _lib.load()
    rng = np.random.RandomState(31)
    if case == 'ragged':
        b, p, t, extra = 9, 23, 60, 8
        dur = rng.randint(0, 6, size=(b, p)).astype(np.int64)
        dur[3] = 0
        dur[4, :5] = 40
        seq_np = rng.randint(10, t + 1, size=b).astype(np.int64)
    elif case == 'c2':
        b, p, t, extra = 256, 80, 1000, ops.PHONE_RATE_EXTRA
        dur = rng.randint(1, 25, size=(b, p)).astype(np.int64)
        seq_np = np.minimum(dur.sum(1), t).astype(np.int64)
    elif case == 'many':                                             # 1 700 short utterances on 32 riders: 50+ jobs per block, staged
        b, p, t, extra = 1700, 16, 20, 64                            # in one batch
        dur = rng.randint(0, 3, size=(b, p)).astype(np.int64)
        seq_np = rng.randint(1, t + 1, size=b).astype(np.int64)
    else:
        b, p, t, extra = 5, 700, 2100, 16
        dur = rng.randint(0, 7, size=(b, p)).astype(np.int64)
        seq_np = None
    target = rng.standard_normal(b * t).astype(np.float32)
    seq = dev(seq_np) if seq_np is not None else None
    assert ops.phone_front_ok(b, p, t, extra)
    rows, mapped, seg = ops.upsample_index_maps(dev(dur), t)
    ybar, weight, partials = ops.phone_target_stats(dev(target), mapped.reshape(-1), seg, seq, b, t, b * p, extra)
    const = torch.zeros((), device=DEV)
    ops.phone_loss_const_add(partials, b * p, extra, const)

    def check(got):
        rows2, mapped2, seg2, ybar2, weight2, partials2 = got[:6]
        assert torch.equal(rows2, rows) and torch.equal(mapped2, mapped) and torch.equal(seg2, seg)
        assert torch.equal(ybar2, ybar) and torch.equal(weight2, weight)
        const2 = torch.zeros((), device=DEV)
        ops.phone_loss_const_add(partials2, b * p, extra, const2)
        np.testing.assert_allclose(const2.item(), const.item(), rtol=1e-6)

    check(ops.phone_front(dev(dur), dev(target), seq, t, extra))
    m, k, n = b * p + extra, 600, 512
    a = ops.cast_pad_bf16(dev(rng.uniform(0, 1, (m, k)).astype(np.float32)))
    (w_bf,), _ = ops.cast_params_bf16([dev(rng.uniform(-0.1, 0.1, (n, k)).astype(np.float32))], want_plain=True, want_t=())
    bias = dev(rng.uniform(-0.1, 0.1, n).astype(np.float32))
    want_y = ops.linear_fwd_bf16(a, None, m, k, w_bf, bias, n, ops.ACT_SIGMOID)
    for probe in (0, 66):
        lib.mg_set_tuning(7, probe)
        try:
            got = ops.phone_front(dev(dur), dev(target), seq, t, extra, linear=(a, k, w_bf, bias, n, ops.ACT_SIGMOID))
        finally:
            lib.mg_set_tuning(7, 0)
        check(got)
        assert torch.equal(got[6], want_y), probe


@pytest.mark.parametrize('masked', [True, False])
def test_phone_target_stats_reduce_the_masked_mse_exactly(masked):
    """mg_phone_target_stats / mg_phone_loss_const_add: for predictions that are constant over the frames of a table row the masked MSE
    of morgana/losses.py:29-51 equals sum_r W_r (p_r - ybar_r)^2 + const.  Checked against the frame-by-frame numpy loss on a ragged
    map with empty phones and padding frames, with seq_len (padding frames partly inside it) and without: weights exact to 1e-6,
    the reassembled loss to 1e-5."""
    from morgana_amd import ops
    rng = np.random.RandomState(23 + masked)
    b, p, t, extra = 6, 11, 48, 8
    rows = _ragged_rows(rng, b, p, t)
    flat = rows.reshape(-1)
    r_tab = b * p
    target = rng.standard_normal(b * t).astype(np.float32)
    seq_np = rng.randint(10, t + 1, size=b).astype(np.int64) if masked else None
    seg, mapped = ops.segment_bounds(dev(flat), r_tab, pad_row=r_tab)
    ybar, weight, partials = ops.phone_target_stats(dev(target), mapped, seg, dev(seq_np) if masked else None, b, t, r_tab, extra)
    loss = torch.zeros((), device=DEV)
    ops.phone_loss_const_add(partials, r_tab, extra, loss)
    # frame weights of the reference loss
    nb = seq_np if masked else np.full(b, t)
    w = ((np.arange(t)[None, :] < nb[:, None]) / (nb[:, None] * float(b))).reshape(-1)
    chunk = -(-b * t // extra)
    row_of = np.where(flat >= 0, flat, r_tab + np.arange(b * t) // chunk)          # padding frames: the extra row of their share
    want_w = np.bincount(row_of, weights=w, minlength=r_tab + extra)
    np.testing.assert_allclose(weight.cpu().numpy(), want_w, rtol=1e-5, atol=1e-9)
    pred_rows = rng.standard_normal(r_tab + extra)
    frame_loss = float((w * (pred_rows[row_of] - target) ** 2).sum())
    got = float((weight.double().cpu().numpy() * (pred_rows - ybar.double().cpu().numpy()) ** 2).sum() + loss.item())
    np.testing.assert_allclose(got, frame_loss, rtol=1e-5)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_phone_rate_loss_of_the_exact_modes_equals_the_frame_rate_loss(precision, monkeypatch):
    """The exact modes' loss on the phone rows - functional.F0TailRowsF32Fn (the 128 -> 32 -> 1 tail, the masked MSE on per-phone
    predictions and target statistics and their backward as ONE launch, mg_f0_tail_rows_f32) and functional.PhoneMSEFn (the loss
    alone, mg_phone_mse_rows_f32, behind the generic layers) - against the same model with the loss at frame rate (predict +
    losses.mse: the table's rows repeated, masked MSE over the frames, segment sums behind it): the same algebra, fp32 rounding
    apart - loss, every gradient and the reported prediction to 1e-5 (the prediction bit for bit where the layers are the same launches)."""
    from morgana_amd import _lib
    feats = data.to_device(synthetic.make_batch(32, 400, seed=21), DEV)
    got = {}
    for form in ('tail', 'rows', 'frames'):
        monkeypatch.setattr(utils, 'F0_TAIL_F32', form == 'tail')
        model = _load_state(models.F0Model(precision=precision, fused_loss=form != 'frames', phone_rate=True).to(DEV), synthetic.f0_model_state())
        calls = []
        _lib.CALL_LOG = calls
        try:
            loss, out = model(feats)
            loss.backward()
        finally:
            _lib.CALL_LOG = None
        assert (calls.count('mg_f0_tail_rows_f32') == 1) == (form == 'tail'), calls
        assert (calls.count('mg_phone_mse_rows_f32') == 1) == (form == 'rows') and (calls.count('mg_segment_sum') == 0) == (form != 'frames'), calls
        got[form] = (loss.item(), out['pred_norm_lf0'].detach().cpu().numpy(),
                     {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()})
    assert np.array_equal(got['rows'][1], got['frames'][1])
    for form in ('tail', 'rows'):
        np.testing.assert_allclose(got[form][0], got['frames'][0], rtol=1e-5)
        assert rel_err(got[form][1], got['frames'][1]) < 1e-5          # the prediction is a cancelling sum of O(1) terms: 2.5e-6 seen
        for name in got['frames'][2]:
            assert rel_err(got[form][2][name], got['frames'][2][name]) < 1e-5, (form, name)


@pytest.mark.parametrize('m', [16, 37, 4099])
def test_f0_tail_rows_f32_kernel(m):
    """mg_f0_tail_rows_f32 (csrc/tail_f32.hip) against the same tail in float64 torch autograd on random rows, weights and per-row
    statistics (some weights zero: rows without frames): prediction, loss, d loss / d Z2 and the four parameter gradients to 2e-6 of
    their largest element; two runs give the same bits."""
    rng = np.random.RandomState(m)
    z2 = dev(rng.standard_normal((m, 128)).astype(np.float32) * 2)
    w3 = dev((rng.standard_normal((32, 128)) / 8).astype(np.float32))
    b3 = dev((rng.standard_normal(32) * 0.1).astype(np.float32))
    w4 = dev((rng.standard_normal((1, 32)) / 4).astype(np.float32))
    b4 = dev(np.array([0.3], dtype=np.float32))
    ybar = dev(rng.standard_normal(m).astype(np.float32))
    wnp = rng.uniform(0, 1e-3, m).astype(np.float32)
    wnp[rng.uniform(size=m) < 0.2] = 0
    weight = dev(wnp)
    pred, dz2, flat = ops.f0_tail_rows_f32(z2, w3, b3, w4, b4, ybar, weight)
    again = ops.f0_tail_rows_f32(z2, w3, b3, w4, b4, ybar, weight)
    for a, b in zip((pred, dz2, flat[:ops.F0_TAIL_F32_GRADS + 1]), (again[0], again[1], again[2][:ops.F0_TAIL_F32_GRADS + 1])):
        assert torch.equal(a, b)
    zt, w3t, b3t, w4t, b4t = [t.double().requires_grad_(True) for t in (z2, w3, b3, w4, b4)]
    p = torch.sigmoid(torch.sigmoid(zt) @ w3t.t() + b3t) @ w4t.t() + b4t
    loss = (weight.double() * (p[:, 0] - ybar.double()) ** 2).sum()
    loss.backward()
    n_g = ops.F0_TAIL_F32_GRADS

    def close(got, want, tol=2e-6):
        return float((got.double() - want).abs().max() / want.abs().max()) < tol
    assert close(pred, p[:, 0].detach()) and close(flat[n_g], loss.detach().reshape(())) and close(dz2, zt.grad)
    assert close(flat[:4096].view(32, 128), w3t.grad) and close(flat[4096:4128], b3t.grad)
    assert close(flat[4128:4160].view(1, 32), w4t.grad) and close(flat[4160:4161], b4t.grad)




@pytest.mark.parametrize('prefetch', [False, True])
@pytest.mark.parametrize('precision,shape', [('bf16', (64, 300)), ('bf16x3', (64, 300)), ('bf16x3', (96, 500))])
def test_graph_cache_loads_the_next_batch_beside_the_running_step(precision, shape, prefetch, monkeypatch):
    """graphs.GraphedStepCache with one batch of look-ahead (ExperimentBuilder.train_epoch): a signature holds two captured steps used in
    turn, and the NEXT batch is copied into the idle one's static buffers on a side stream while the current step runs (only the tensors
    the step reads: BaseModel.step_input_keys - the operand table, not the fp32 feature it was made from).  Eight distinct batches of a
    slab-taking shape, two epochs, against the eager loop: epoch losses and final parameters EQUAL, and the replays were loaded ahead."""
    from morgana_amd import experiment_builder, graphs
    # ``prefetch`` (graphs.PREFETCH, off by default: measured equal to slower) loads ahead on the side stream; without it the batch is
    # loaded in line in front of the replay, in ONE launch that also files the previous step's loss
    monkeypatch.setattr(graphs, 'PREFETCH', prefetch)
    # (64 x 300: 2 560 table rows - the fused bf16 step, but the GENERIC 'bf16x3' path, which splits the fp32 feature inside the step: the
    # model must then name every tensor; 96 x 500: 4 864 rows - the fused 'bf16x3' step, which reads the pair table only)
    batches = [synthetic.make_batch(shape[0], shape[1], seed=300 + i) for i in range(8)]

    def train(use_graphs):
        torch.manual_seed(3)
        builder = experiment_builder.ExperimentBuilder(models.F0Model, dict(precision=precision), learning_rate=0.01, device=DEV, end_epoch=2,
                                                       use_graphs=use_graphs, graph_group=1)       # (the per-batch load path)
        _load_state(builder.model, synthetic.f0_model_state())
        dev_batches = [data.to_device(b, DEV, bf16_tables=builder.model.bf16_table_features()) for b in batches]
        history = builder.run_train(dev_batches)
        return history, {k: v.detach().clone() for k, v in builder.model.named_parameters()}, builder

    hist_e, params_e, _ = train(False)
    hist_g, params_g, builder = train(True)
    assert hist_g == hist_e
    for name in params_e:
        assert torch.equal(params_g[name], params_e[name]), name
    stats = builder._graph_cache.stats()
    assert stats['graphs'] == 2 and stats['eager'] == 1 and stats['replayed'] == 15 and (stats['prefetched'] >= 11) == prefetch, stats
    keys = builder.model.step_input_keys(data.to_device(batches[0], DEV, bf16_tables=builder.model.bf16_table_features()))
    assert (keys is None) == (precision == 'bf16x3' and shape == (64, 300))           # only the fused steps leave the fp32 feature out


@pytest.mark.parametrize('precision', ['bf16', 'bf16x3', 'fp32'])
def test_resident_epoch_groups_equal_the_eager_loop(precision):
    """ExperimentBuilder(use_graphs=True) over a RESIDENT loader (a list of device batches): ``graph_group`` consecutive batches are
    captured into one graph that reads them where they lie (graphs.GraphedStepCache.step_group) - no load launch, one graph launch per
    group, ragged shapes in one graph.  Seven batches of three shapes in groups of three, three epochs (eager, capture + replay,
    replay), against the eager loop: epoch losses and final parameters EQUAL; then one batch's CONTENT is changed in place and both
    loops run another epoch - the graph reads the batch, not a copy of it."""
    from morgana_amd import experiment_builder
    shapes = [(64, 300)] * 4 + [(96, 500)] * 2 + [(16, 120)]
    batches = [synthetic.make_batch(b, t, seed=500 + i) for i, (b, t) in enumerate(shapes)]
    other = synthetic.make_batch(64, 300, seed=599)

    def train_by_hand(use_graphs):
        torch.manual_seed(3)
        builder = experiment_builder.ExperimentBuilder(models.F0Model, dict(precision=precision), learning_rate=0.01, device=DEV,
                                                       use_graphs=use_graphs, graph_group=3)
        _load_state(builder.model, synthetic.f0_model_state())
        tables = builder.model.bf16_table_features()
        dev_batches = [data.to_device(b, DEV, bf16_tables=tables) for b in batches]
        optimizer = builder.make_optimizer()
        history = [builder.train_epoch(dev_batches, optimizer) for _ in range(3)]
        fresh = data.to_device(other, DEV, bf16_tables=tables)
        for key, value in dev_batches[1].items():          # new content at the old addresses
            if isinstance(value, torch.Tensor):
                value.copy_(fresh[key])
        history.append(builder.train_epoch(dev_batches, optimizer))
        return history, {k: v.detach().clone() for k, v in builder.model.named_parameters()}, builder

    hist_e, params_e, _ = train_by_hand(False)
    hist_g, params_g, builder = train_by_hand(True)
    assert hist_g == hist_e
    assert hist_e[3] != hist_e[2]
    for name in params_e:
        assert torch.equal(params_g[name], params_e[name]), name
    stats = builder._graph_cache.stats()
    assert stats['eager'] == 7 and stats['replayed'] == 21 and stats['group_graphs'] == 3 and stats['group_replays'] == 9, stats
    assert stats['graphs'] == 0                                # no single-step graphs, no static copies


@pytest.mark.parametrize('group', [1, 3])
def test_a_step_that_cannot_be_captured_trains_on_ordinary_launches(group):
    """ExperimentBuilder(use_graphs=True) with a model whose forward reads a device value back on every call (HIP refuses that inside a
    capture): the capture attempt fails, the cache warns once, keeps that batch shape on ordinary launches and the epoch goes on - no
    step lost, none run twice: losses and parameters EQUAL to the eager loop; per-batch form and resident groups."""
    from morgana_amd import experiment_builder

    class Syncing(models.F0Model):
        def forward(self, features):
            self.frames_seen = getattr(self, 'frames_seen', 0) + int(features['n_frames'].sum().item())      # a host read per step
            return super().forward(features)

    batches = [synthetic.make_batch(16, 120, seed=700 + i) for i in range(5)]

    def train(use_graphs):
        torch.manual_seed(3)
        builder = experiment_builder.ExperimentBuilder(Syncing, dict(precision='bf16'), learning_rate=0.01, device=DEV, use_graphs=use_graphs,
                                                       graph_group=group)
        _load_state(builder.model, synthetic.f0_model_state())
        dev_batches = [data.to_device(b, DEV, bf16_tables=builder.model.bf16_table_features()) for b in batches]
        optimizer = builder.make_optimizer()
        history = [builder.train_epoch(dev_batches, optimizer) for _ in range(3)]
        return history, {k: v.detach().clone() for k, v in builder.model.named_parameters()}, builder

    hist_e, params_e, eager = train(False)
    with pytest.warns(UserWarning, match='cannot be captured'):
        hist_g, params_g, builder = train(True)
    assert hist_g == hist_e and builder.model.frames_seen == eager.model.frames_seen
    for name in params_e:
        assert torch.equal(params_g[name], params_e[name]), name
    stats = builder._graph_cache.stats()
    assert stats['eager'] == 15 and stats['replayed'] == 0 and stats['graphs'] == 0 and stats['group_graphs'] == 0, stats
    # the thread is usable afterwards: an ordinary model still captures and replays
    plain = experiment_builder.ExperimentBuilder(models.F0Model, dict(precision='bf16'), learning_rate=0.01, device=DEV, use_graphs=True, graph_group=1)
    dev_batches = [data.to_device(b, DEV, bf16_tables=plain.model.bf16_table_features()) for b in batches]
    plain.train_epoch(dev_batches, plain.make_optimizer())
    assert plain._graph_cache.stats()['replayed'] == 4


def test_resident_groups_follow_an_epoch_level_learning_rate_schedule():
    """Resident groups (ten batches per graph) under an epoch-level schedule ('exponential', gamma 0.7) through ``run_train``: the captured
    update kernels read the step's scalars from the staging launch of every replay, so the learning rate that changed between two epochs
    is the one the replayed steps use - four epochs, losses and parameters EQUAL to the eager loop."""
    from morgana_amd import experiment_builder
    batches = [synthetic.make_batch(64, 300, seed=800 + i) for i in range(6)]

    def train(use_graphs):
        torch.manual_seed(3)
        builder = experiment_builder.ExperimentBuilder(models.F0Model, dict(precision='bf16'), learning_rate=0.02, lr_schedule_name='exponential',
                                                       lr_schedule_kwargs=dict(gamma=0.7), device=DEV, end_epoch=4, use_graphs=use_graphs,
                                                       graph_group=4)
        _load_state(builder.model, synthetic.f0_model_state())
        dev_batches = [data.to_device(b, DEV, bf16_tables=builder.model.bf16_table_features()) for b in batches]
        history = builder.run_train(dev_batches)
        return history, {k: v.detach().clone() for k, v in builder.model.named_parameters()}, builder

    hist_e, params_e, _ = train(False)
    hist_g, params_g, builder = train(True)
    assert hist_g == hist_e
    for name in params_e:
        assert torch.equal(params_g[name], params_e[name]), name
    stats = builder._graph_cache.stats()
    assert stats['group_graphs'] == 2 and stats['group_replays'] == 7 and stats['eager'] == 4, stats
