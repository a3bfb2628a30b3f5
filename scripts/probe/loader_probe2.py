"""Which call of the all-asynchronous loader stalls: C-level call times through cProfile over four streamed C2 batches."""
import sys, time, cProfile, pstats
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
def epoch():
    for b in range(4):
        out = data.collate_to_device(utts[256 * b:256 * (b + 1)], norms, dev, bf16_tables=('normalised_lab',))
    return out
epoch(); torch.cuda.synchronize()
epoch(); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
epoch()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
