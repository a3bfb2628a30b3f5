"""mg_host_pack into pinned memory: alone, and while an H2D copy of another pinned buffer is in flight (1 and 8 threads)."""
import sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, '.')
from morgana_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
items = [np.random.rand(80, 600).astype(np.float32) for _ in range(256)]
srcs = (ctypes.c_void_p * len(items))(*[a.ctypes.data for a in items])
sizes = (ctypes.c_int64 * len(items))(*[a.nbytes for a in items])
n = sum(a.nbytes for a in items)
pin = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
page = torch.empty(n, dtype=torch.uint8)
dst = torch.empty(n, dtype=torch.uint8, device=dev)
def pack(buf, threads):
    t0 = time.perf_counter()
    rc = lib.mg_host_pack(ctypes.cast(srcs, ctypes.c_void_p), ctypes.cast(sizes, ctypes.c_void_p), len(items), ctypes.c_void_p(buf.data_ptr()), ctypes.c_int64(n), threads)
    assert rc == 0
    return (time.perf_counter() - t0) * 1e3
for threads in (1, 8):
    torch.cuda.synchronize()
    print('threads', threads)
    print('  pinned, gpu idle      ', ' '.join('%.2f' % pack(pin[0], threads) for _ in range(6)))
    print('  pageable, gpu idle    ', ' '.join('%.2f' % pack(page, threads) for _ in range(6)))
    ts = []
    for i in range(6):
        dst.copy_(pin[1], non_blocking=True)
        ts.append(pack(pin[0], threads))
    torch.cuda.synchronize()
    print('  pinned, beside an H2D ', ' '.join('%.2f' % t for t in ts))
    ts = []
    for i in range(6):
        dst.copy_(pin[1], non_blocking=True)
        ts.append(pack(page, threads))
    torch.cuda.synchronize()
    print('  pageable, beside H2D  ', ' '.join('%.2f' % t for t in ts))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); dst.copy_(pin[1], non_blocking=True); e1.record(); e1.synchronize()
print('H2D of %.1f MB: %.2f ms' % (n / 1e6, e0.elapsed_time(e1)))
