"""Timing of the phone-rate front beside the first layer's GEMM (C2 shapes): one grid, two launches, and the grid with idle rider blocks
(MG_TUNE_AB = 67, results garbage).  Run on the GPU box: python scripts/kbench_front.py"""
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (the lab library, before morgana_amd loads one)

import numpy as np
import torch
from morgana_amd import ops, _lib

lib = _lib.load()
dev = torch.device('cuda:0')
rng = np.random.RandomState(0)
b, p, t, extra = 256, 80, 1000, ops.PHONE_RATE_EXTRA
dur = torch.from_numpy(rng.randint(1, 25, size=(b, p)).astype(np.int64)).to(dev)
target = torch.from_numpy(rng.standard_normal(b * t).astype(np.float32)).to(dev)
seq = torch.full((b,), t, dtype=torch.int64, device=dev)
m, k, n = b * p + extra, 600, 512
a = ops.cast_pad_bf16(torch.from_numpy(rng.uniform(0, 1, (m, k)).astype(np.float32)).to(dev))
(w_bf,), _ = ops.cast_params_bf16([torch.from_numpy(rng.uniform(-0.1, 0.1, (n, k)).astype(np.float32)).to(dev)], want_plain=True, want_t=())
bias = torch.zeros(n, device=dev)


def timed(fn, iters=20, reps=10):
    """us per call, 20 calls captured into one HIP graph (the host's launch path costs more than these kernels take)."""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / (iters * reps) * 1e3


print('gemm alone            %7.1f us' % timed(lambda: ops.linear_fwd_bf16(a, None, m, k, w_bf, bias, n, ops.ACT_SIGMOID)))
print('front alone           %7.1f us' % timed(lambda: ops.phone_front(dur, target, seq, t, extra)))
for probe, name in ((0, 'one grid'), (66, 'two launches'), (67, 'one grid, idle rider'), (71, 'no utterance compute'), (72, 'no extra compute'),
                    (74, 'no extra jobs'), (73, 'staging only'), (78, 'rider alone'), (79, 'rider alone, extras only'),
                    (80, 'rider alone, no extra compute'), (82, 'rider alone, utterances only'), (81, 'rider alone, staging only')):
    lib.mg_set_tuning(7, probe)
    us = timed(lambda: ops.phone_front(dur, target, seq, t, extra, linear=(a, k, w_bf, bias, n, ops.ACT_SIGMOID)))
    lib.mg_set_tuning(7, 0)
    print('%-21s %7.1f us' % (name, us))
