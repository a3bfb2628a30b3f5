"""Layer-1 forward GEMM of the phone-rate step (21 504 x 600 -> 512) alone, with its timing probes (lab build: MG_TUNE_FORM 32 + mask,
1 = no A pieces, 2 = no B pieces, 4 = no fragment reads, 8 = no MFMAs, 16 = no epilogue), each call timed with HIP events inside one
graph; `cold` puts a 512 MB fill between two calls (the rest of a training step moves 300 MB: the table is not in L2 when the step
comes back to it).  Run on the GPU box: python scripts/kbench_fwd_phone.py"""
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _lab  # noqa: E402,F401

import numpy as np
import torch
from morgana_amd import ops, _lib

lib = _lib.load()
dev = torch.device('cuda:0')
rng = np.random.RandomState(0)
m, k, n = 21504, 600, 512
a = ops.cast_pad_bf16(torch.from_numpy(rng.uniform(0, 1, (m, k)).astype(np.float32)).to(dev))
(w_bf,), _ = ops.cast_params_bf16([torch.from_numpy(rng.uniform(-0.1, 0.1, (n, k)).astype(np.float32)).to(dev)], want_plain=True, want_t=())
bias = torch.zeros(n, device=dev)
flush = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)


def timed(fn, cold, iters=20, reps=10):
    """us per call inside one HIP graph of `iters` calls; cold: a 512 MB fill in front of every call, its own graph subtracted"""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()

    def graph_of(body):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                body()
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            e.synchronize()
            ts.append(s.elapsed_time(e) / iters * 1e3)
        ts.sort()
        return ts[len(ts) // 2], ts[0]

    if not cold:
        return graph_of(fn)

    def both():
        flush.fill_(1.0)
        fn()
    a_med, a_min = graph_of(both)
    b_med, b_min = graph_of(lambda: flush.fill_(1.0))
    return a_med - b_med, a_min - b_min


forms = [int(v) for v in _os.environ.get('MG_FORMS', '0,33,34,35,36,40,48,47,39').split(',')]
for cold in (False, True):
    for form in forms:
        lib.mg_set_tuning(0, form)
        med, mn = timed(lambda: ops.linear_fwd_bf16(a, None, m, k, w_bf, bias, n, ops.ACT_SIGMOID), cold)
        lib.mg_set_tuning(0, 0)
        print('%-5s form %3d   median %7.1f us   min %7.1f us' % ('cold' if cold else 'warm', form, med, mn), flush=True)
