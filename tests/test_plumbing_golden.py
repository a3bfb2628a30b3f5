"""G8: the reference's own ExperimentBuilder run (3 epochs, shuffled batches, on-disk data set) replayed through this
repo's loop.  CPU: host logic (normalise -> collate -> train_epoch -> metrics.json) with the oracle's torch ops as the
compute stand-in.  GPU: the same replay on the HIP path in fp32 mode, to the 1e-4 bar."""
import json
import os

import numpy as np
import pytest
import torch

from morgana_amd import data, experiment_builder, models, synthetic, utils

import helpers


def _dataset(g):
    names = [str(n) for n in g['names']]
    norm = {'lab': data.MinMaxNormaliser('lab').set_params({'mmin': g['norm__lab_minmax__mmin'],
                                                             'mmax': g['norm__lab_minmax__mmax']}),
            'lf0': data.MeanVarianceNormaliser('lf0').set_params({'mean': g['norm__lf0_mvn__mean'],
                                                                  'std_dev': g['norm__lf0_mvn__std_dev']})}
    utts = {}
    for name in names:
        raw = {'name': name, 'n_frames': int(g['data__%s__lf0' % name].shape[0]), 'dur': g['data__%s__dur' % name],
               'lab': g['data__%s__lab' % name], 'lf0': g['data__%s__lf0' % name]}
        utts[name] = data.load_utterance(raw, norm)
    return utts, norm


def _epoch_batches(g, utts, epoch, device):
    batches = []
    for joined in g['batch_order'][epoch]:
        batch = data.collate_fn([utts[n] for n in str(joined).split(',')])
        batches.append(data.to_device(batch, device))
    return batches


def test_collate_matches_reference_layout(golden):
    g = golden('g8_plumbing.npz')
    utts, _ = _dataset(g)
    first = str(g['batch_order'][0][0]).split(',')
    batch = data.collate_fn([utts[n] for n in first])
    assert batch['name'] == first
    assert batch['n_frames'].dtype == torch.int64 and tuple(batch['n_frames'].shape) == (len(first),)
    assert batch['dur'].dtype == torch.int64 and batch['dur'].dim() == 3
    t_max = int(batch['n_frames'].max())
    assert tuple(batch['normalised_lf0'].shape) == (len(first), t_max, 1) and batch['normalised_lf0'].dtype == torch.float32
    for i, n in enumerate(first):
        t = utts[n]['n_frames']
        assert torch.all(batch['normalised_lf0'][i, t:] == 0)          # zero padding (data.py:183-193)
        assert int(batch['dur'][i].sum()) == t
    assert float(batch['normalised_lab'].max()) <= 1.0 + 1e-6 and float(batch['normalised_lab'].min()) >= -1e-6


def _g16_batches(g16):
    for i in range(int(g16['n_batches'])):
        yield {k.split('__', 1)[1]: v for k, v in g16.items() if k.startswith('batch%d__' % i)}


def test_files_dataset_and_collate_equal_reference_batches(golden, tmp_path):
    """``data.FilesDataset`` (files on disk, id list, normaliser JSON) + ``collate_fn`` against G16: the batches the REFERENCE's
    FilesDataset.__getitem__ + collate_fn built from the same files (data.py:106-154, 159-224): every tensor equal (the host
    normalisation is the same float32 NumPy arithmetic), names in order, dtypes and shapes as the reference's."""
    g8, g16 = golden('g8_plumbing.npz'), golden('g16_collate.npz')
    dataset = helpers.g8_files_dataset(g8, str(tmp_path))
    assert dataset.file_ids == [str(n) for n in g16['file_ids']] and len(dataset) == len(g16['file_ids'])
    for want in _g16_batches(g16):
        names = [str(n) for n in want['name']]
        got = dataset.collate_fn([dataset[dataset.file_ids.index(n)] for n in names])
        assert got['name'] == names
        assert sorted(got.keys()) == sorted(want.keys())
        for key, value in want.items():
            if key == 'name':
                continue
            assert tuple(got[key].shape) == value.shape and got[key].numpy().dtype == value.dtype, key
            np.testing.assert_array_equal(got[key].numpy(), value, err_msg=key)
    with pytest.raises(ValueError, match='use_deltas'):
        data.FilesDataset({'lf0': data.NumpyBinarySource('lf0')}, 'train', 'train_file_id_list.scp',
                          {'lf0': data.MeanVarianceNormaliser('lf0', use_deltas=True)}, data_root=str(tmp_path))


@pytest.mark.gpu
def test_device_collate_equals_reference_batches(golden, tmp_path):
    """The device-side feed (``data.batch`` -> ``DeviceBatches`` over a ``FilesDataset``: raw utterances packed, one H2D copy per
    feature, mg_pad_normalise_f32 pads and normalises) against G16, the reference's own host-side batches of the same utterances:
    integers and raw features equal, normalised features to fp32 rounding of the normaliser arithmetic (1e-6)."""
    g8, g16 = golden('g8_plumbing.npz'), golden('g16_collate.npz')
    dataset = helpers.g8_files_dataset(g8, str(tmp_path), device='cuda:0')
    for want in _g16_batches(g16):
        names = [str(n) for n in want['name']]
        raw = [dataset.raw(dataset.file_ids.index(n)) for n in names]
        got = data.collate_to_device(raw, dataset.normalisers, 'cuda:0')
        assert got['name'] == names and got['n_frames_total'] == int(want['n_frames'].sum())
        for key, value in want.items():
            if key == 'name':
                continue
            assert got[key].is_cuda and tuple(got[key].shape) == value.shape and got[key].cpu().numpy().dtype == value.dtype, key
            if value.dtype.kind == 'f':
                np.testing.assert_allclose(got[key].cpu().numpy(), value, rtol=1e-6, atol=1e-7, err_msg=key)
            else:
                np.testing.assert_array_equal(got[key].cpu().numpy(), value, err_msg=key)
    # the loader form: same utterances batch by batch in file order, the last batch smaller
    loader = data.batch(dataset, batch_size=8, shuffle=False, device='cuda:0')
    assert len(loader) == 3
    seen = [n for feats in loader for n in feats['name']]
    assert seen == dataset.file_ids


def _run(g, device, model_class, model_kwargs, tmp_path, kernel=None):
    dims = tuple(int(d) for d in g['dims'])
    eb = experiment_builder.ExperimentBuilder(model_class, model_kwargs=model_kwargs, learning_rate=0.01, device=device,
                                              experiment_dir=str(tmp_path), end_epoch=3)
    own = eb.model.state_dict()
    for k, v in synthetic.f0_model_state(seed=4242, dims=dims).items():
        own[k].copy_(torch.from_numpy(v))
    utts, _ = _dataset(g)
    opt = eb.make_optimizer(**({'kernel': kernel} if kernel else {}))
    sched = eb._lr_schedule(opt)
    losses = []
    for epoch in range(3):
        eb.epoch = epoch + 1
        out_dir = os.path.join(str(tmp_path), 'train', 'epoch_%d' % (epoch + 1))
        losses.append(eb.train_epoch(_epoch_batches(g, utts, epoch, device), opt, sched, out_dir=out_dir))
        saved = json.load(open(os.path.join(out_dir, 'metrics.json')))
        assert saved['loss'] == pytest.approx(g['epoch_metrics_loss'][epoch], rel=2e-4)
    return eb, losses


def test_g8_replay_on_cpu_host_logic(golden, tmp_path):
    g = golden('g8_plumbing.npz')
    dims = tuple(int(d) for d in g['dims'])
    eb, losses = _run(g, 'cpu', helpers.CpuF0Model, {'dims': dims}, tmp_path, kernel=helpers.cpu_adam_kernel)
    np.testing.assert_allclose(losses, g['epoch_metrics_loss'], rtol=1e-4)
    state = eb.model.state_dict()
    for key in [str(k) for k in g['checkpoint_keys']]:
        np.testing.assert_allclose(state[key].double().sum().item(), g['final_sum__' + key], rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
def test_g8_replay_on_gpu_fp32(golden, tmp_path):
    g = golden('g8_plumbing.npz')
    dims = tuple(int(d) for d in g['dims'])
    kwargs = {'input_dim': dims[0], 'hidden_dims': dims[1:-1], 'output_dim': dims[-1], 'precision': 'fp32'}
    eb, losses = _run(g, 'cuda:0', models.F0Model, kwargs, tmp_path)
    np.testing.assert_allclose(losses, g['epoch_metrics_loss'], rtol=1e-4)
    state = eb.model.state_dict()
    assert sorted(state.keys()) == sorted(str(k) for k in g['checkpoint_keys'])     # checkpoints interchange
    for key in state:
        np.testing.assert_allclose(state[key].double().sum().item(), g['final_sum__' + key], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(state[key].double().abs().sum().item(), g['final_abs_sum__' + key], rtol=1e-3)
    path = eb.model.save_parameters(str(tmp_path), 3)
    again = models.F0Model(**kwargs).to('cuda:0')
    again.load_parameters(path)
    for a, b in zip(eb.model.parameters(), again.parameters()):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_train_epoch_with_ema_and_noam_on_gpu():
    """Loop body with a batch-level LR schedule and the EMA twin (experiment_builder.py:477-484) against the oracle."""
    from morgana_amd import lr_schedules
    from oracle import ref_cpu, ref_torch
    dims = (24, 16, 8, 1)
    feats = [synthetic.make_batch(4, (30, 60), lab_dim=24, frames_per_phone=5.0, seed=s) for s in (1, 2, 3, 4)]
    kwargs = {'input_dim': 24, 'hidden_dims': (16, 8), 'output_dim': 1, 'precision': 'fp32'}
    eb = experiment_builder.ExperimentBuilder(models.F0Model, model_kwargs=kwargs, learning_rate=0.02, device='cuda:0',
                                              lr_schedule_name='noam', lr_schedule_kwargs={'warmup_steps': 2},
                                              ema_decay=0.9)
    state = synthetic.f0_model_state(seed=9, dims=dims)
    for model in (eb.model, eb.ema_model):
        own = model.state_dict()
        for k, v in state.items():
            own[k].copy_(torch.from_numpy(v))
    opt = eb.make_optimizer()
    sched = eb._lr_schedule(opt)
    mean_loss = eb.train_epoch([data.to_device(f, 'cuda:0') for f in feats], opt, sched)

    ref = ref_torch.load_state(ref_torch.F0Model(dims), state)
    ref_opt = torch.optim.Adam(ref.parameters(), lr=0.02)
    ref_sched = lr_schedules.NoamLR(ref_opt, warmup_steps=2)
    shadow = {k: v.copy() for k, v in state.items()}
    losses = []
    for f in feats:
        ref_opt.zero_grad()
        loss, _ = ref(ref_torch.to_torch(f))
        loss.backward()
        ref_opt.step()
        ref_sched.step()
        losses.append(loss.item())
        for k, v in ref.state_dict().items():
            ref_cpu.ema_update(shadow[k], v.numpy(), 0.9)
    assert mean_loss == pytest.approx(np.mean(losses), rel=1e-4)
    for k, v in eb.model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), ref.state_dict()[k].numpy(), rtol=1e-3, atol=1e-5)
    for k, v in eb.ema_model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), shadow[k], rtol=1e-3, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('model_name', ['f0', 'rnn'])
def test_ema_twin_in_bf16_mode_sees_its_updates(model_name):
    """The EMA twin's weights are written by a HIP kernel through ``param.data`` (utils.ExponentialMovingAverage, morgana/utils.py:
    443-456), which torch's version counter does not see.  In bf16 mode the layers multiply by bf16 COPIES of the weights that live on
    the parameters (ops.param_shadows): a copy made by an earlier forward pass of the twin must not survive an EMA update.  Evaluate
    the twin, update it towards a different model, evaluate again: the second output must equal a fresh model's with the same
    weights (it did not change at all before ops.mark_updated)."""
    if model_name == 'f0':
        make = lambda: models.F0Model(precision='bf16').to('cuda:0')
        state_a, state_b = synthetic.f0_model_state(seed=3), synthetic.f0_model_state(seed=4)
        feats = data.to_device(synthetic.make_batch(16, (150, 300), seed=5), 'cuda:0')
    else:
        make = lambda: models.RNNSPSS(precision='bf16').to('cuda:0')
        state_a, state_b = synthetic.rnn_spss_state(seed=3), synthetic.rnn_spss_state(seed=4)
        feats = data.to_device(synthetic.make_batch(8, (100, 200), out_dim=80, target_name='mcep', seed=5), 'cuda:0')

    def load(model, state):
        own = model.state_dict()
        for k, v in state.items():
            own[k].copy_(torch.from_numpy(v))
        return model

    def output_of(model):
        with torch.no_grad():
            _, out = model(feats)
        out = next(iter(out.values())) if isinstance(out, dict) else out
        return out.clone()

    twin, other = load(make(), state_a), load(make(), state_b)
    ema = utils.ExponentialMovingAverage(twin, 0.25)
    before = output_of(twin)                                   # creates the twin's bf16 operand copies
    ema.update_params(other)                                   # shadow = 0.25 shadow + 0.75 other
    after = output_of(twin)
    fresh = make()
    fresh.load_state_dict(twin.state_dict())
    want = output_of(fresh)
    assert not torch.equal(after, before)
    assert torch.equal(after, want)


@pytest.mark.gpu
def test_train_epoch_from_device_batches_matches_host_collate():
    """ExperimentBuilder.train_epoch fed by data.DeviceBatches (raw utterances -> pad + normalise on the device) against the
    same epoch fed by the host pipeline (normalise-on-load, collate_fn, to_device: data.py:119-127, 159-224, 648-663): same
    batches, so the same mean loss and the same parameters afterwards (fp32 mode, 1e-5)."""
    rng = np.random.RandomState(5)
    lab_dim = 24
    norms = {'lab': data.MinMaxNormaliser('lab').set_params({'mmin': rng.rand(lab_dim).astype(np.float32) * 0.1,
                                                             'mmax': (1.0 + rng.rand(lab_dim)).astype(np.float32)}, device='cuda:0'),
             'lf0': data.MeanVarianceNormaliser('lf0').set_params({'mean': np.array([5.0], np.float32),
                                                                   'std_dev': np.array([0.3], np.float32)}, device='cuda:0')}
    utterances = []
    for i in range(10):
        n_ph = int(rng.randint(3, 9))
        dur = rng.randint(1, 8, size=(n_ph, 1)).astype(np.int64)
        n_fr = int(dur.sum())
        utterances.append({'name': 'utt%02d' % i, 'n_frames': n_fr, 'n_phones': n_ph, 'dur': dur,
                           'lab': rng.rand(n_ph, lab_dim).astype(np.float32),
                           'lf0': (5.0 + 0.3 * rng.randn(n_fr, 1)).astype(np.float32)})
    kwargs = {'input_dim': lab_dim, 'hidden_dims': (16, 8), 'output_dim': 1, 'precision': 'fp32'}
    state = synthetic.f0_model_state(seed=9, dims=(lab_dim, 16, 8, 1))

    def run(loader):
        eb = experiment_builder.ExperimentBuilder(models.F0Model, model_kwargs=kwargs, learning_rate=0.02, device='cuda:0')
        own = eb.model.state_dict()
        for k, v in state.items():
            own[k].copy_(torch.from_numpy(v))
        mean_loss = eb.train_epoch(loader, eb.make_optimizer())
        return mean_loss, {k: v.cpu().numpy().copy() for k, v in eb.model.state_dict().items()}

    device_loader = data.DeviceBatches(utterances, 4, norms, 'cuda:0')
    assert len(device_loader) == 3
    host_loader = [data.to_device(data.collate_fn([data.load_utterance(u, norms) for u in utterances[i:i + 4]]), 'cuda:0')
                   for i in range(0, 10, 4)]
    got_loss, got = run(device_loader)
    want_loss, want = run(host_loader)
    assert got_loss == pytest.approx(want_loss, rel=1e-5)
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=1e-4, atol=1e-6, err_msg=k)
    shuffled = data.DeviceBatches(utterances, 4, norms, 'cuda:0', shuffle=np.random.RandomState(0))
    names = [n for batch in shuffled for n in batch['name']]
    assert sorted(names) == sorted(u['name'] for u in utterances) and names != [u['name'] for u in utterances]


@pytest.mark.gpu
def test_bf16_loader_table_and_launches_of_the_product_step():
    """The loader half of bf16 mode is in the product: ``data.DeviceBatches`` - asked by
    ``ExperimentBuilder.train_epoch`` for what the model's ``bf16_table_features()`` names - writes the bf16 operand table of the phone
    input in the pass that pads and normalises (mg_pad_normalise_bf16_f32), bit for bit what casting the normalised feature gives
    (zero padding columns, zero extra rows); the training step then launches NO cast of the phone table, and at C2-like shapes it is
    the six-entry phone-rate step that bench.py times (entry-point calls counted through ``_lib.CALL_LOG``)."""
    from morgana_amd import _lib, ops
    rng = np.random.RandomState(11)
    lab_dim, n_utt, n_ph, per_batch_utts = 600, 192, 40, 96          # 96 x 40 + 1,024 = 4,864 table rows: the slab-taking phone-rate step
    norms = {'lab': data.MinMaxNormaliser('lab').set_params({'mmin': (rng.rand(lab_dim) * 0.1).astype(np.float32),
                                                             'mmax': (1.0 + rng.rand(lab_dim)).astype(np.float32)}, device='cuda:0'),
             'lf0': data.MeanVarianceNormaliser('lf0').set_params({'mean': np.array([5.0], np.float32),
                                                                   'std_dev': np.array([0.3], np.float32)}, device='cuda:0')}
    utterances = []
    for i in range(n_utt):
        dur = rng.randint(5, 21, size=(n_ph, 1)).astype(np.int64)
        n_fr = int(dur.sum())
        utterances.append({'name': 'utt%03d' % i, 'n_frames': n_fr, 'n_phones': n_ph, 'dur': dur,
                           'lab': rng.rand(n_ph, lab_dim).astype(np.float32),
                           'lf0': (5.0 + 0.3 * rng.randn(n_fr, 1)).astype(np.float32)})
    loader = data.DeviceBatches(utterances, per_batch_utts, norms, 'cuda:0')
    # the table, against the stand-alone cast of the normalised feature
    batch = next(iter(loader.use_bf16_tables(('normalised_lab',))))
    table = batch['normalised_lab' + data.BF16_TABLE_SUFFIX]
    lab = batch['normalised_lab']
    want = ops.cast_pad_bf16(lab.reshape(-1, lab_dim), extra_rows=ops.PHONE_RATE_EXTRA)
    assert table.shape == want.shape and torch.equal(table.view(torch.int16), want.view(torch.int16))
    plain = next(iter(data.DeviceBatches(utterances, per_batch_utts, norms, 'cuda:0')))
    assert 'normalised_lab' + data.BF16_TABLE_SUFFIX not in plain and torch.equal(plain['normalised_lab'], lab)

    eb = experiment_builder.ExperimentBuilder(models.F0Model, model_kwargs={'precision': 'bf16'}, learning_rate=0.01, device='cuda:0')
    own = eb.model.state_dict()
    for k, v in synthetic.f0_model_state().items():
        own[k].copy_(torch.from_numpy(v))
    loader = data.DeviceBatches(utterances, per_batch_utts, norms, 'cuda:0')          # no tables requested: train_epoch asks for them
    optimizer = eb.make_optimizer()
    eb.train_epoch(loader, optimizer)                                     # first epoch: operand shadows, buffers
    assert loader.bf16_tables == ('normalised_lab',)
    _lib.CALL_LOG = []
    try:
        eb.train_epoch(loader, optimizer)
        log = list(_lib.CALL_LOG)
    finally:
        _lib.CALL_LOG = None
    loader_calls = ('mg_pad_normalise_f32', 'mg_pad_normalise_bf16_f32', 'mg_host_pack')      # (host pack: one per float feature and batch)
    step_calls = [c for c in log if c not in loader_calls]
    assert not any('cast' in c for c in step_calls), step_calls
    assert log.count('mg_pad_normalise_bf16_f32') == len(loader) and log.count('mg_host_pack') == 2 * len(loader), log
    # per batch: front + layer-1 GEMM | layers 2-4 + loss + their backward | prediction expansion + tail reduce | layer-2 wgrad + dgrad |
    # layer-1 wgrad | Adam's scalars | update - bench.py's graph replay stages the scalars once per ten steps, hence its "six"
    # (eager launches, as here; a step captured into a HIP graph has one entry point less: the third rides at the end of the fourth,
    # mg_linear_wgrad_dgrad_expand_bf16 - tests/test_gpu_parity.py::test_graphed_step_defers_the_tail)
    want = ['mg_phone_front_linear_fwd_bf16', 'mg_f0_l2tail_rows_slabs_bf16', 'mg_expand_column_reduce_f32', 'mg_linear_wgrad_dgrad_bf16',
            'mg_linear_wgrad_slabs_bf16', 'mg_store_pair_f32', 'mg_adam_step_plan_f32']
    n = len(loader)
    per_epoch = [c for c in step_calls if c not in want and not c.endswith('_status')]     # the epoch's time-out check reads status words
    assert sorted(c for c in step_calls if c in want) == sorted(want * n), step_calls
    assert len(per_epoch) <= 3, per_epoch                    # the epoch's reads of the loss metric / the persistent-kernel status


@pytest.mark.gpu
def test_streamed_batches_do_not_share_staging():
    """data.collate_to_device packs every batch into pinned staging that is REUSED (two buffers per feature, in turn, an event each) and
    copies it asynchronously: eight batches are built back to back with no synchronisation in between - the host runs far ahead of the
    copies - and only then compared with the host pipeline (collate_fn on normalised utterances): every batch holds its own rows.
    Ragged lengths, batches of different sizes (the staging grows), and ``DeviceBatches.resident`` = the same list."""
    rng = np.random.RandomState(21)
    lab_dim = 600
    norms = {'lab': data.MinMaxNormaliser('lab').set_params({'mmin': (rng.rand(lab_dim) * 0.1).astype(np.float32),
                                                             'mmax': (1.0 + rng.rand(lab_dim)).astype(np.float32)}, device='cuda:0'),
             'lf0': data.MeanVarianceNormaliser('lf0').set_params({'mean': np.array([5.0], np.float32),
                                                                   'std_dev': np.array([0.3], np.float32)}, device='cuda:0')}
    utterances = []
    for i in range(8 * 48 + 17):
        n_ph = int(rng.randint(20, 90))
        dur = rng.randint(5, 21, size=(n_ph, 1)).astype(np.int64)
        n_fr = int(dur.sum())
        utterances.append({'name': 'utt%04d' % i, 'n_frames': n_fr, 'n_phones': n_ph, 'dur': dur,
                           'lab': rng.rand(n_ph, lab_dim).astype(np.float32),
                           'lf0': (5.0 + 0.3 * rng.randn(n_fr, 1)).astype(np.float32)})
    loader = data.DeviceBatches(utterances, 48, norms, 'cuda:0', bf16_tables=('normalised_lab',))
    batches = loader.resident()                                  # nine batches (the last of 17 utterances), no sync in between
    assert len(batches) == len(loader) == 9
    torch.cuda.synchronize()
    for b, got in enumerate(batches):
        items = utterances[48 * b:48 * (b + 1)]
        want = data.collate_fn([data.load_utterance(u, norms) for u in items])
        assert got['name'] == [u['name'] for u in items]
        for key in ('lab', 'lf0', 'dur', 'n_frames', 'n_phones'):
            assert torch.equal(got[key].cpu(), want[key]), (b, key)
        for key in ('normalised_lab', 'normalised_lf0'):
            np.testing.assert_allclose(got[key].cpu().numpy(), want[key].numpy(), rtol=0, atol=2e-6, err_msg='%d %s' % (b, key))
        table = got['normalised_lab' + data.BF16_TABLE_SUFFIX]
        rows = got['normalised_lab'].shape[0] * got['normalised_lab'].shape[1]
        assert torch.equal(table[:rows, :lab_dim], got['normalised_lab'].reshape(rows, lab_dim).to(torch.bfloat16))
