"""Statement-level times inside data._pack_pinned over streamed C2 batches (a copy of the function with timers)."""
import sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, '.')
from morgana_amd import data, _lib
import morgana_amd.data as D
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
LOG = []
def pack(items, device, key):
    t = [time.perf_counter()]
    width = items[0].shape[1]
    lens = np.array([x.shape[0] for x in items], dtype=np.int64)
    total = int(lens.sum())
    n_off = len(items) + 1
    off_bytes = (n_off * 8 + 63) // 64 * 64
    n_bytes = off_bytes + total * width * 4
    slot = D._STAGING.take(str(device), key, n_bytes)
    t.append(time.perf_counter())
    host = slot[0]
    offsets_host = host[:n_off * 8].view(torch.int64)
    offsets_host[0] = 0
    offsets_host[1:] = torch.from_numpy(np.cumsum(lens))
    t.append(time.perf_counter())
    arrays = items
    srcs = (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
    sizes = (ctypes.c_int64 * len(arrays))(*[a.nbytes for a in arrays])
    lib = _lib.load()
    t.append(time.perf_counter())
    rc = lib.mg_host_pack(ctypes.cast(srcs, ctypes.c_void_p), ctypes.cast(sizes, ctypes.c_void_p), len(arrays),
                          ctypes.c_void_p(host.data_ptr() + off_bytes), ctypes.c_int64(host.numel() - off_bytes), D.HOST_PACK_THREADS)
    t.append(time.perf_counter())
    staged = host[:n_bytes].to(device, non_blocking=True)
    t.append(time.perf_counter())
    slot[1] = torch.cuda.Event()
    slot[1].record(torch.cuda.current_stream(device))
    t.append(time.perf_counter())
    offsets = staged[:n_off * 8].view(torch.int64)
    packed = staged[off_bytes:].view(torch.float32).view(total, width)
    LOG.append((key, [round((b - a) * 1e3, 2) for a, b in zip(t, t[1:])]))
    return packed, offsets, lens
D._pack_pinned = pack
print('[take, offsets, ctypes arrays, host_pack, to(), event]')
for rep in range(12):
    LOG.clear()
    per = []
    for b in range(4):
        t1 = time.perf_counter()
        out = data.collate_to_device(utts[256 * b:256 * (b + 1)], norms, dev, bf16_tables=('normalised_lab',))
        per.append(round((time.perf_counter() - t1) * 1e3, 1))
    torch.cuda.synchronize()
    print('rep', rep, per, [(k, v) for k, v in LOG if max(v) > 10])
