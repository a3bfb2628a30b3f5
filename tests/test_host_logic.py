"""CPU tests of the host side: synthetic generator, NumPy normaliser path, LR schedules, flat Adam bookkeeping and the
train_epoch loop semantics (experiment_builder.py:431-505), with the oracle's torch-CPU ops standing in for the device
kernels (the product itself has no CPU compute path)."""
import json
import os

import numpy as np
import pytest
import torch

from morgana_amd import data, experiment_builder, lr_schedules, metrics, optim, synthetic
from oracle import ref_cpu, ref_torch

import helpers


def test_synthetic_batches_follow_the_contract():
    f = synthetic.make_batch(256, 1000)
    assert f['normalised_lab'].shape == (256, 80, 600) and f['normalised_lab'].dtype == np.float32
    assert f['dur'].shape == (256, 80, 1) and f['dur'].dtype == np.int64 and f['dur'].min() >= 1
    assert np.all(f['dur'].sum(axis=(1, 2)) == 1000) and np.all(f['n_frames'] == 1000)
    assert f['normalised_lf0'].shape == (256, 1000, 1)
    assert 0.0 <= f['normalised_lab'].min() and f['normalised_lab'].max() < 1.0
    g = synthetic.make_batch(256, 1000)
    assert np.array_equal(f['dur'], g['dur']) and np.array_equal(f['normalised_lab'], g['normalised_lab'])
    h = synthetic.make_batch(256, 1000, rank=1)
    assert not np.array_equal(f['dur'], h['dur'])
    r = synthetic.make_batch(16, (300, 2000), out_dim=187, target_name='mcep', seed=9)
    assert r['normalised_mcep'].shape == (16, int(r['n_frames'].max()), 187)
    assert np.all(r['dur'].sum(axis=(1, 2)) == r['n_frames'])
    for b in range(16):                                         # zero padded beyond each utterance
        assert np.all(r['normalised_mcep'][b, r['n_frames'][b]:] == 0)
        assert np.all(r['dur'][b, r['n_phones'][b]:] == 0)


def test_numpy_normaliser_path(golden):
    g = golden('g5_normalisers.npz')
    mvn = data.MeanVarianceNormaliser('lf0').set_params({'mean': g['mean'], 'std_dev': g['std']})
    mm = data.MinMaxNormaliser('lab').set_params({'mmin': g['mmin'], 'mmax': g['mmax']})
    with np.errstate(all='ignore'):
        np.testing.assert_allclose(mvn.normalise(g['feat']), g['mvn_norm_np'], rtol=1e-6)
        np.testing.assert_allclose(mvn.denormalise(g['feat']), g['mvn_denorm_np'], rtol=1e-6)
        np.testing.assert_allclose(mm.normalise(g['feat']), g['minmax_norm_np'], rtol=1e-6)
        np.testing.assert_allclose(mm.denormalise(g['feat']), g['minmax_denorm_np'], rtol=1e-6)
    with pytest.raises(RuntimeError):
        mvn.normalise(torch.from_numpy(g['feat']))              # CPU torch tensors have no device path


def test_normaliser_json_loading(tmp_path):
    d = tmp_path / 'norm'
    d.mkdir()
    (d / 'lf0_mvn.json').write_text(json.dumps({'mean': [1.0, 2.0], 'std_dev': [0.5, 2.0]}))
    (d / 'lab_minmax.json').write_text(json.dumps({'mmin': [0.0, 1.0], 'mmax': [2.0, 1.0]}))
    n = data.Normalisers({'lf0': data.MeanVarianceNormaliser('lf0'), 'lab': data.MinMaxNormaliser('lab')}, 'norm',
                         data_root=str(tmp_path))
    x = np.array([[2.0, 4.0]], dtype=np.float32)
    np.testing.assert_allclose(n['lf0'].normalise(x), [[2.0, 1.0]], rtol=1e-6)
    np.testing.assert_allclose(n['lab'].normalise(x), [[1.0, 3.0]], rtol=1e-6)      # max == min -> scale 1
    assert n['lf0'].params_torch['mean'].dtype == torch.float32


def test_lr_schedules_match_reference(golden):
    g = golden('g9_ema_lr.npz')

    def sequence(cls, n, **kwargs):
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        sched = cls(opt, **kwargs)
        seq = []
        for _ in range(n):
            seq.append(opt.param_groups[0]['lr'])
            opt.step()
            sched.step()
        return seq

    np.testing.assert_allclose(sequence(lr_schedules.NoamLR, 12, warmup_steps=4), g['noam_w4'], rtol=1e-12)
    np.testing.assert_allclose(sequence(lr_schedules.CyclicNoamLR, 24, warmup_steps=4, cycle_steps=9),
                               g['cyclic_noam_w4_c9'], rtol=1e-12)
    np.testing.assert_allclose(sequence(lr_schedules.CyclicNoamLR, 30, warmup_steps=4, cycle_trigger=0.5),
                               g['cyclic_noam_w4_trig'], rtol=1e-12)
    np.testing.assert_allclose(sequence(lr_schedules.DummyLR, 5), g['constant'])
    assert lr_schedules.init_lr_schedule('noam', warmup_steps=7).keywords == {'warmup_steps': 7}
    m = metrics.Mean()
    for v in g['mean_inputs']:
        m.accumulate(torch.tensor(v))
    np.testing.assert_allclose(float(m.result()), g['mean_result'], rtol=1e-5)


def test_flat_adam_matches_torch_adam():
    torch.manual_seed(0)
    model_a = helpers.init_small(ref_torch.F0Model((12, 8, 4, 1)), seed=3)
    model_b = helpers.init_small(ref_torch.F0Model((12, 8, 4, 1)), seed=3)
    feats = ref_torch.to_torch(synthetic.make_batch(4, 30, lab_dim=12, frames_per_phone=5.0, seed=4))
    opt_a = optim.Adam(model_a.parameters(), lr=0.02, weight_decay=1e-2, kernel=helpers.cpu_adam_kernel)
    opt_b = torch.optim.Adam(model_b.parameters(), lr=0.02, weight_decay=1e-2)
    flat = opt_a.flat_buffers()
    assert flat['param'].numel() == sum(p.numel() for p in model_a.parameters())
    for p in model_a.parameters():                              # parameters and grads are views into the flat buffers
        lo, hi = flat['param'].data_ptr(), flat['param'].data_ptr() + flat['param'].numel() * 4
        assert lo <= p.data_ptr() < hi and p.grad is not None
    for _ in range(6):
        for opt, model in ((opt_a, model_a), (opt_b, model_b)):
            opt.zero_grad()
            loss, _ = model(feats)
            loss.backward()
            opt.step()
    for pa, pb in zip(model_a.parameters(), model_b.parameters()):
        np.testing.assert_allclose(pa.detach().numpy(), pb.detach().numpy(), rtol=1e-5, atol=1e-7)
    # zero_grad keeps the views (even after a caller dropped them) and really zeroes
    for p in model_a.parameters():
        p.grad = None
    opt_a.zero_grad()
    assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in model_a.parameters())
    # torch LR schedulers drive it like any torch optimiser
    sched = lr_schedules.NoamLR(opt_a, warmup_steps=4)
    sched.step()
    assert opt_a.param_groups[0]['lr'] == pytest.approx(0.02 * ref_cpu.noam_scale(1, 4))


def test_train_epoch_semantics(tmp_path):
    """Loop body order, batch-level LR schedule, loss bookkeeping and metrics.json (experiment_builder.py:464-505)."""
    batches = [ref_torch.to_torch(synthetic.make_batch(4, 40, lab_dim=24, frames_per_phone=5.0, seed=s))
               for s in (1, 2, 3)]
    eb = experiment_builder.ExperimentBuilder(helpers.CpuF0Model, model_kwargs={'dims': (24, 16, 8, 1)},
                                              learning_rate=0.05, lr_schedule_name='noam',
                                              lr_schedule_kwargs={'warmup_steps': 2}, device='cpu',
                                              experiment_dir=str(tmp_path), end_epoch=2)
    helpers.init_small(eb.model, seed=5)
    opt = eb.make_optimizer(kernel=helpers.cpu_adam_kernel)
    sched = eb._lr_schedule(opt)
    out_dir = str(tmp_path / 'train' / 'epoch_1')
    mean_loss = eb.train_epoch(batches, opt, sched, out_dir=out_dir)

    # the same three steps by hand with torch.optim.Adam + the reference-shaped Noam schedule
    ref_model = helpers.init_small(helpers.CpuF0Model(dims=(24, 16, 8, 1)), seed=5)
    ref_opt = torch.optim.Adam(ref_model.parameters(), lr=0.05)
    ref_sched = lr_schedules.NoamLR(ref_opt, warmup_steps=2)
    losses = []
    for feats in batches:
        ref_opt.zero_grad()
        loss, _ = ref_model(feats)
        loss.backward()
        ref_opt.step()
        ref_sched.step()
        losses.append(loss.item())
    assert mean_loss == pytest.approx(np.mean(losses), rel=1e-5)
    assert eb.model.step == 3 and eb.model.mode == ''
    saved = json.load(open(os.path.join(out_dir, 'metrics.json')))
    assert saved['loss'] == pytest.approx(ref_cpu.metric_mean(losses), rel=1e-5)
    for pa, pb in zip(eb.model.parameters(), ref_model.parameters()):
        np.testing.assert_allclose(pa.detach().numpy(), pb.detach().numpy(), rtol=1e-4, atol=1e-6)
    assert opt.param_groups[0]['lr'] == pytest.approx(ref_opt.param_groups[0]['lr'])


def test_valid_and_test_epochs(tmp_path):
    """Evaluation passes of experiment_builder.py:562-680: no parameter moves, mean loss and metrics.json as the reference keeps them,
    the EMA twin is the one evaluated when EMA is on, test_epoch calls predict and the analysis hooks only."""
    batches = [ref_torch.to_torch(synthetic.make_batch(4, 40, lab_dim=24, frames_per_phone=5.0, seed=s)) for s in (4, 5)]
    eb = experiment_builder.ExperimentBuilder(helpers.CpuF0Model, model_kwargs={'dims': (24, 16, 8, 1)}, device='cpu',
                                              experiment_dir=str(tmp_path), end_epoch=1, ema_decay=0.9)
    helpers.init_small(eb.model, seed=6)
    helpers.init_small(eb.ema_model, seed=7)                       # a different twin: shows which model is evaluated
    before = [p.detach().clone() for p in eb.model.parameters()]
    want = []
    with torch.no_grad():
        for feats in batches:
            want.append(eb.ema_model(feats)[0].item())
    got = eb.run_valid(batches)
    assert got == pytest.approx(np.mean(want), rel=1e-6)
    saved = json.load(open(os.path.join(str(tmp_path), 'valid', 'epoch_1', 'metrics.json')))
    assert saved['loss'] == pytest.approx(ref_cpu.metric_mean(want), rel=1e-5)
    assert eb.ema_model.mode == '' and all(torch.equal(a, b) for a, b in zip(before, eb.model.parameters()))
    assert eb.valid_epoch(batches, model=eb.model) != pytest.approx(got)          # an explicit model overrides the twin

    seen = []
    eb.ema_model.analysis_for_test_batch = lambda features, output_features, out_dir, **kw: seen.append(sorted(output_features))
    eb.run_test(batches)
    assert seen == [['pred_norm_lf0']] * 2 and os.path.exists(os.path.join(str(tmp_path), 'test', 'epoch_1', 'metrics.json'))


def test_checkpoint_round_trip(tmp_path):
    model = helpers.init_small(helpers.CpuF0Model(dims=(24, 16, 8, 1)), seed=8)
    path = model.save_parameters(str(tmp_path), 3)
    assert path.endswith(os.path.join('checkpoints', 'epoch_3.pt'))
    other = helpers.CpuF0Model(dims=(24, 16, 8, 1))
    other.load_parameters(path, device='cpu')
    for pa, pb in zip(model.parameters(), other.parameters()):
        assert torch.equal(pa, pb)


def test_early_bucket_exchange_only_when_the_stack_is_the_model():
    """The captured two-bucket exchange (graphs.GraphedTrainStep) cuts the flat gradient where the FIRING stack says its tail is
    final.  That is only true of the whole buffer when the stack's parameters are exactly the optimiser's, in order:
    a model with further trainable parameters, or a different order, keeps the single collective."""
    from morgana_amd import graphs
    torch.manual_seed(0)
    stack = [torch.nn.Parameter(torch.randn(64, 48)), torch.nn.Parameter(torch.randn(64)),
             torch.nn.Parameter(torch.randn(8, 64)), torch.nn.Parameter(torch.randn(8)),
             torch.nn.Parameter(torch.randn(1, 8)), torch.nn.Parameter(torch.randn(1))]
    opt = optim.Adam(stack, lr=0.01, kernel=helpers.cpu_adam_kernel)
    assert opt.bucket_split() == 64 * 48 + 64
    assert graphs.early_exchange_is_safe(opt, stack)
    assert not graphs.early_exchange_is_safe(opt, stack[:4])                     # the stack is not the whole model
    assert not graphs.early_exchange_is_safe(opt, stack[2:4] + stack[:2] + stack[4:])     # same parameters, another order
    extra = torch.nn.Parameter(torch.randn(5))
    opt2 = optim.Adam(stack + [extra], lr=0.01, kernel=helpers.cpu_adam_kernel)
    assert not graphs.early_exchange_is_safe(opt2, stack)                        # the model has parameters outside the stack
    small_first = [torch.nn.Parameter(torch.randn(2, 2)), torch.nn.Parameter(torch.randn(2)),
                   torch.nn.Parameter(torch.randn(64, 64)), torch.nn.Parameter(torch.randn(64))]
    opt3 = optim.Adam(small_first, lr=0.01, kernel=helpers.cpu_adam_kernel)
    assert opt3.bucket_split() == 0 and not graphs.early_exchange_is_safe(opt3, small_first)


def test_bf16_copy_of_a_recurrence_output_is_dropped_after_an_in_place_write():
    """the bf16 copy a recurrence attaches to its fp32 output (utils.bf16_copy_of) is the next Linear run's operand only
    while the fp32 tensor still holds the values it copied - an in-place write (out.mul_(), out[:, k:] = 0, a hook) moves the version
    counter and the copy is ignored (the run casts the tensor again)."""
    from morgana_amd import utils
    out = torch.rand(2, 3, 8)
    copy = out.to(torch.bfloat16)
    assert utils.bf16_copy_of(out) is None
    out._mg_bf16 = (copy, out._version)
    assert utils.bf16_copy_of(out) is copy
    out.mul_(2.0)
    assert utils.bf16_copy_of(out) is None


def test_dropout_with_p_one_equals_torch_also_for_non_finite_activations():
    """ADVICE round 4 asked for exact zeros from an active nn.Dropout(p=1) inside SequentialWithRecurrent; the reference's module is
    torch's, whose result is input * 0 (0 for finite values, NaN for inf / NaN, -0 for negative ones) with a zero gradient - that is
    the parity target, and what the container returns on any device."""
    from morgana_amd import utils
    stack = utils.SequentialWithRecurrent(torch.nn.Dropout(p=1.0))
    stack.train()
    x = torch.tensor([[[1.0, float('inf'), float('nan'), -2.0]]], requires_grad=True)
    y, _ = stack(x)
    want_in = x.detach().clone().requires_grad_(True)
    want = torch.nn.functional.dropout(want_in, p=1.0, training=True)
    assert torch.equal(torch.nan_to_num(y.detach(), nan=7.0), torch.nan_to_num(want.detach(), nan=7.0))
    assert torch.equal(torch.signbit(y.detach()), torch.signbit(want.detach()))
    y[..., 0].sum().backward()
    want[..., 0].sum().backward()
    assert torch.equal(x.grad, want_in.grad)
