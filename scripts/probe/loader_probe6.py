"""The all-asynchronous loader's ~90 ms stalls: process CPU time across them and the cgroup's throttling counters."""
import sys, time, os, glob
import numpy as np, torch
sys.path.insert(0, '.')
from morgana_amd import data
def cpu_stat():
    for path in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat'):
        if os.path.exists(path):
            return ' '.join(l.strip() for l in open(path) if 'thrott' in l)
    return 'no cpu.stat'
def cpu_max():
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        if os.path.exists(path):
            return open(path).read().strip()
    return '?'
print('cpu.max', cpu_max(), 'cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'threads', data.HOST_PACK_THREADS)
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
for threads in (data.HOST_PACK_THREADS, 1):
    data.HOST_PACK_THREADS = threads
    print('pack threads', threads, cpu_stat())
    for rep in range(6):
        per = []
        for b in range(4):
            t1, c1 = time.perf_counter(), time.process_time()
            out = data.collate_to_device(utts[256 * b:256 * (b + 1)], norms, dev, bf16_tables=('normalised_lab',))
            per.append('%.1f/%.1f' % ((time.perf_counter() - t1) * 1e3, (time.process_time() - c1) * 1e3))
        torch.cuda.synchronize()
        print('  rep', rep, 'wall/cpu ms', per, cpu_stat())
