#!/usr/bin/env python
"""
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (the lab library, before morgana_amd loads one)
Micro-benchmark of mg_f0_l2tail_bf16 (layer 2 + tail in one pass over H1) against the pair it replaces, at the C2 frame-rate and
phone-rate shapes, with the kernel's timing probes (MG_TUNE_AB).  Usage: python scripts/kbench_l2tail.py [iters]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from morgana_amd import _lib, ops  # noqa: E402

PROBES = {0: 'product', 68: 'rolling prefetch', 67: '16x16x32 form (f0_l2tail16_kernel)', 64: 'role split (experiment)', 1: 'no H1 loads in the loop', 2: 'no tail', 3: 'no loads, no tail', 4: 'no layer-2 MFMAs', 6: 'loads only',
          7: 'nothing but the loop', 8: 'no sigmoid on H2', 16: 'no steps 8-9', 32: 'no step 9 (dW3)', 17: 'no loads, no steps 8-9'}


FAST = os.environ.get('MG_KB_FAST') == '1'      # only the two product forms


def timed(fn, iters):
    for _ in range(2):
        fn()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / iters * 1e3


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = 'cuda:0'
    lib = _lib.load()
    for b, t in ((256, 1000), (1, 21504)):
        m = b * t
        h1 = torch.rand(m, 512, device=dev).to(torch.bfloat16)
        w2b = (torch.randn(128, 512, device=dev) * 0.05).to(torch.bfloat16)
        b2 = torch.zeros(128, device=dev)
        w3, b3 = torch.randn(32, 128, device=dev) * 0.1, torch.zeros(32, device=dev)
        w4, b4 = torch.randn(1, 32, device=dev) * 0.1, torch.zeros(1, device=dev)
        tgt = torch.randn(m, device=dev)
        sl = torch.full((b,), t, dtype=torch.int64, device=dev)
        grads = torch.empty(4162, device=dev)

        def pair():
            h2 = ops.linear_fwd_bf16(h1, None, m, 512, w2b, b2, 128, ops.ACT_SIGMOID)
            ops.f0_tail(h2, w3, b3, w4, b4, tgt, sl, b, t, grads)

        print('M = %d rows' % m)
        print('  unfused pair (gemm_nt_persist<128> + f0_tail)  %8.1f us' % timed(pair, iters))
        for probe, what in [(k, v) for k, v in PROBES.items() if not FAST or k in (0, 68, 67, 64)]:
            lib.mg_set_tuning(7, probe)
            us = timed(lambda: ops.f0_l2tail(h1, w2b, b2, w3, b3, w4, b4, tgt, sl, b, t, grads), iters)
            print('  l2tail probe %d (%-26s)  %8.1f us   %6.2f TB/s of H1 + dZ2' % (probe, what, us, (m * 1280) / us / 1e6))
        lib.mg_set_tuning(7, 0)


if __name__ == '__main__':
    main()
