"""Where a streamed C2 batch's host time goes (data.collate_to_device): per-call wall times of the pieces, no synchronisation in between."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from morgana_amd import data, ops
dev = torch.device('cuda:0')
rng = np.random.RandomState(1)
lab_dim, n_ph = 600, 80
norms = {'lab': data.MinMaxNormaliser('lab').set_params({'mmin': (rng.rand(lab_dim) * 0.1).astype(np.float32), 'mmax': (1.0 + rng.rand(lab_dim)).astype(np.float32)}, device=dev),
         'lf0': data.MeanVarianceNormaliser('lf0').set_params({'mean': np.array([5.0], np.float32), 'std_dev': np.array([0.3], np.float32)}, device=dev)}
utts = []
for i in range(256 * 4):
    dur = np.full((n_ph, 1), 12, np.int64); dur[::2] += 1
    utts.append({'name': 'u%d' % i, 'n_frames': int(dur.sum()), 'n_phones': n_ph, 'dur': dur, 'lab': rng.rand(n_ph, lab_dim).astype(np.float32),
                 'lf0': rng.randn(int(dur.sum()), 1).astype(np.float32)})
import morgana_amd.data as D
orig_pack, orig_small, orig_pad = D._pack_pinned, D._small_to_device, ops.pad_normalise
T = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3); return r
    return w
D._pack_pinned = timed('pack+h2d issue', orig_pack)
D._small_to_device = timed('small', orig_small)
ops.pad_normalise = timed('pad_normalise issue', orig_pad)
for rep in range(3):
    T.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(4):
        t1 = time.perf_counter()
        out = data.collate_to_device(utts[256 * b:256 * (b + 1)], norms, dev, bf16_tables=('normalised_lab',))
        T.setdefault('collate total', []).append((time.perf_counter() - t1) * 1e3)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    print('rep', rep, 'host %.2f ms, +sync %.2f ms' % ((t2 - t0) * 1e3, (time.perf_counter() - t2) * 1e3))
    for k, v in T.items():
        print('   %-22s %s' % (k, ' '.join('%.2f' % x for x in v)))
