"""world_size-2 gloo tests on CPU of the data-parallel path: contiguous utterance shards + ONE flat all-reduce per step
must reproduce the single-process run on the global batch."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from morgana_amd import distributed, optim, synthetic
from oracle import ref_torch

import helpers

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _single_process(n_steps, ragged):
    torch.set_num_threads(1)
    model = helpers.init_small(helpers.CpuF0Model(dims=(24, 16, 8, 1)), seed=1)
    frames = (30, 90) if ragged else 50
    batch = ref_torch.to_torch(synthetic.make_batch(8, frames, lab_dim=24, frames_per_phone=5.0, seed=17))
    opt = optim.Adam(model.parameters(), lr=0.01, weight_decay=1e-3, kernel=helpers.cpu_adam_kernel)
    losses = []
    for _ in range(n_steps):
        opt.zero_grad()
        loss, _ = model(batch)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return opt.flat_buffers()['param'].numpy().copy(), losses


@pytest.mark.parametrize('world', [2, 4])
@pytest.mark.parametrize('mode', ['fixed', 'ragged'])
def test_data_parallel_equals_global_batch(tmp_path, world, mode):
    out = str(tmp_path / 'dp.npz')
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, 'tests', '_dist_worker.py'), out, '4', mode],
                                      env=env, cwd=REPO))
    for p in procs:
        assert p.wait(timeout=240) == 0
    got = np.load(out)
    want_flat, want_losses = _single_process(4, mode == 'ragged')
    for replica in got['replicas']:                         # replicas stay bit-identical to each other
        assert np.array_equal(replica, got['replicas'][0])
    np.testing.assert_allclose(got['flat'], want_flat, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got['losses'], want_losses, rtol=1e-5)


def test_shard_slices_and_cropping():
    batch = synthetic.make_batch(8, (30, 90), lab_dim=6, seed=2)
    seen = []
    for rank in range(4):
        shard = synthetic.shard_batch(batch, rank, 4)
        sl = distributed.shard_slice(8, rank, 4)
        assert np.array_equal(shard['n_frames'], batch['n_frames'][sl])
        t = int(shard['n_frames'].max())
        assert shard['normalised_lf0'].shape[1] == t
        assert np.array_equal(shard['normalised_lf0'], batch['normalised_lf0'][sl, :t])
        assert np.array_equal(shard['dur'].sum(axis=(1, 2)), shard['n_frames'])
        seen += shard['name']
    assert seen == batch['name']
    with pytest.raises(ValueError):
        distributed.shard_slice(10, 0, 4)


def _bucket_worker(rank, world, port, out_dir):
    """Two-bucket gradient exchange (optim.Adam.exchange_gradients('early' / 'late'), graphs.GraphedTrainStep's multi-rank path) against
    the single whole-buffer all-reduce: every element reduced exactly once, bit-identical sums."""
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    from morgana_amd import optim
    import helpers
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        results = {}
        for mode in ('whole', 'buckets'):
            model = helpers.init_small(helpers.CpuF0Model(dims=(24, 16, 8, 1)), seed=3)
            opt = optim.Adam(model.parameters(), lr=0.01, kernel=helpers.cpu_adam_kernel)
            rng = np.random.RandomState(100 + rank)
            flat = opt.flat_buffers()
            flat['grad'].copy_(torch.from_numpy(rng.standard_normal(flat['grad'].numel()).astype(np.float32)))
            split = opt.bucket_split()
            assert split == 24 * 16 + 16                      # the first Linear's weight + bias: produced last by the backward pass
            if mode == 'whole':
                opt.exchange_gradients()
            else:
                opt.exchange_gradients('early')
                opt.exchange_gradients('late')
            results[mode] = flat['grad'].clone()
        assert torch.equal(results['whole'], results['buckets'])
        np.save(os.path.join(out_dir, 'bucket_rank%d.npy' % rank), results['buckets'].numpy())
    finally:
        dist.destroy_process_group()


def test_two_bucket_exchange_equals_single_all_reduce(tmp_path):
    import numpy as np
    import torch.multiprocessing as mp
    port = 29641
    mp.spawn(_bucket_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / 'bucket_rank0.npy'), np.load(tmp_path / 'bucket_rank1.npy')
    assert np.array_equal(a, b)                           # replicas hold identical reduced gradients
    want = np.random.RandomState(100).standard_normal(a.size).astype(np.float32) + np.random.RandomState(101).standard_normal(a.size).astype(np.float32)
    np.testing.assert_array_equal(a, want)


def _exchange_record_worker(rank, world, port, out_dir):
    import json
    import os
    import sys
    import torch
    import torch.distributed as dist
    from morgana_amd import optim
    import helpers
    sys.path.insert(0, REPO)
    import bench
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        model = helpers.init_small(helpers.CpuF0Model(dims=(24, 16, 8, 1)), seed=3)
        opt = optim.Adam(model.parameters(), lr=0.01, kernel=helpers.cpu_adam_kernel)
        rec = bench.exchange_record(opt, None, world, False, torch.device('cpu'), calls=3)
        alone = optim.Adam(helpers.init_small(helpers.CpuF0Model(dims=(24, 16, 8, 1)), seed=3).parameters(), lr=0.01,
                           kernel=helpers.cpu_adam_kernel, exchange_never=True)
        rec['alone_exchanging'] = alone.exchanging()
        rec['alone_world'] = alone._world()
        with open(os.path.join(out_dir, 'exchange_rank%d.json' % rank), 'w') as f:
            json.dump(rec, f)
    finally:
        dist.destroy_process_group()


def test_exchange_record_fields_over_gloo(tmp_path):
    """bench.py's `exchange` record on a live two-rank gloo group, no GPU: the ranks are COUNTED by a real
    all-reduce, the exchange of the flat gradient is timed on its own, the capture probe reports why nothing was captured, and the
    measurement switch `exchange_never` gives an optimiser that does not exchange (the `exposed_us` leg's step)."""
    import json
    import torch.multiprocessing as mp
    mp.spawn(_exchange_record_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    recs = [json.load(open(tmp_path / ('exchange_rank%d.json' % r))) for r in range(2)]
    for rec in recs:
        assert rec['world_size'] == 2 and rec['ranks_counted'] == 2 and rec['backend'] == 'gloo'
        assert rec['mode'] == 'eager all-reduce'
        assert rec['us'] is not None and rec['us'] > 0 and rec['bytes'] == 4 * (24 * 16 + 16 + 16 * 8 + 8 + 8 + 1)
        assert 'exposed_us' in rec and rec['capture_probe']['verdict'] is False and 'error' in rec['capture_probe']
        assert rec['alone_exchanging'] is False and rec['alone_world'] == 1
    assert recs[0]['us'] == recs[1]['us']                             # the MAX over ranks: every rank holds the same number
    # one rank, no process group: the record still has every field
    sys.path.insert(0, REPO)
    import bench
    solo = bench.exchange_record(None, None, 1, False, torch.device('cpu'))
    assert solo['ranks_counted'] == 1 and solo['mode'] == 'none (one rank)' and solo['us'] is None and solo['capture_probe'] is None


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher must start N ranks itself (torch.distributed.run as a child process, before
    any GPU call) and report n_gpus = N; a WORLD_SIZE that disagrees with --gpus is an error, never a silent single-rank run
    (VERDICT round 2, item 3).  --dry-run-ranks stops behind the rendezvous, so this runs on the CPU box over gloo."""
    import json
    env = dict(os.environ, MG_DIST_BACKEND='gloo')
    for key in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(key, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--dry-run-ranks'], env=env, timeout=300,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['ranks_counted'] == 2 and line['backend'] == 'gloo'
    bad = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--dry-run-ranks'], timeout=120,
                         env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True)
    assert bad.returncode != 0 and 'does not match WORLD_SIZE' in bad.stderr
