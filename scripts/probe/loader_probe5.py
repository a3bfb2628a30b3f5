"""Is the ~90 ms stall of the all-asynchronous loader a Python garbage collection?  gc callbacks time every collection."""
import sys, time, gc
import numpy as np, torch
sys.path.insert(0, '.')
from morgana_amd import data
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
GC = []
state = {}
def cb(phase, info):
    if phase == 'start':
        state['t'] = time.perf_counter()
    else:
        GC.append((info['generation'], round((time.perf_counter() - state['t']) * 1e3, 2)))
gc.callbacks.append(cb)
for mode in ('gc on', 'gc off', 'gc on'):
    if mode == 'gc off':
        gc.disable()
    else:
        gc.enable()
    for rep in range(3):
        GC.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        per = []
        for b in range(4):
            t1 = time.perf_counter()
            out = data.collate_to_device(utts[256 * b:256 * (b + 1)], norms, dev, bf16_tables=('normalised_lab',))
            per.append(round((time.perf_counter() - t1) * 1e3, 2))
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        print(mode, 'rep', rep, 'host %.2f ms +sync %.2f' % ((t2 - t0) * 1e3, (time.perf_counter() - t2) * 1e3), per, 'collections', GC)
