#!/usr/bin/env python
"""
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (the lab library, before morgana_amd loads one)
Micro-benchmark of the phone-rate step's GEMM kernels (M = B*P + 1024 = 21 504 table rows at C2) under tuning variants:
interleaved rounds in one process, HIP-event timing on the launch stream.  Usage: python scripts/kbench_phone.py [iters]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from morgana_amd import _lib, ops  # noqa: E402


def timed(fn, iters):
    fn()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / iters * 1e3


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = 'cuda:0'
    lib = _lib.load()
    r_tab, k, n1, n2 = 21504, 600, 512, 128
    tab = (torch.rand(r_tab, 640, device=dev)).to(torch.bfloat16)
    tab[:, 600:] = 0
    dz1 = (torch.randn(r_tab, n1, device=dev) * 0.01).to(torch.bfloat16)
    dz2 = (torch.randn(r_tab, n2, device=dev) * 0.01).to(torch.bfloat16)
    h1 = torch.rand(r_tab, n1, device=dev).to(torch.bfloat16)
    w1b = (torch.randn(n1, 640, device=dev) * 0.05).to(torch.bfloat16)
    b1 = torch.zeros(n1, device=dev)
    cases = [('wgrad1 (dZ1^T table, 512x600)', 2.0 * r_tab * k * n1, lambda: ops.linear_wgrad_bf16(dz1, tab, None, r_tab, n1, k)),
             ('wgrad2 (dZ2^T H1, 128x512)', 2.0 * r_tab * n1 * n2, lambda: ops.linear_wgrad_bf16(dz2, h1, None, r_tab, n2, n1))]
    variants = [(0, 0), (8, 3), (12, 3), (16, 3), (24, 3), (32, 3), (42, 3)]
    rounds = 3
    for skip in (1, 0):
        lib.mg_set_tuning(1, skip)
        print('--- %s' % ('GEMM kernel alone' if skip else 'with the slab reduce'))
        for name, flops, fn in cases:
            times = {v: [] for v in variants}
            for _ in range(rounds):
                for v in variants:
                    lib.mg_set_tuning(4, v[0])
                    lib.mg_set_tuning(5, v[1])
                    times[v].append(timed(fn, iters))
            for v in variants:
                ts = sorted(times[v])
                print('%-32s splits %3d order %d  median %7.1f us (min %7.1f)  %7.1f TFLOP/s' % (name, v[0], v[1], ts[1], ts[0], flops / ts[1] / 1e6))
    lib.mg_set_tuning(1, 0)
    lib.mg_set_tuning(4, 0)
    lib.mg_set_tuning(5, 0)
    l1 = lambda: ops.linear_fwd_bf16(tab, None, r_tab, k, w1b, b1, n1, ops.ACT_SIGMOID)
    print('fwd1 phone rate: %.1f us' % timed(l1, iters))
    w2t = (torch.randn(n1, n2, device=dev) * 0.05).to(torch.bfloat16)
    ident = torch.arange(r_tab, device=dev, dtype=torch.int32)
    dg = lambda: ops.linear_dgrad_bf16(dz2, r_tab, n2, w2t, n1, h1)
    wg = lambda: ops.linear_wgrad_bf16(dz1, tab, None, r_tab, n1, k)
    fz = lambda: ops.linear_bwd_fused_bf16(dz2, w2t, h1, tab, ident, r_tab, n1, k)
    fz0 = lambda: ops.linear_bwd_fused_bf16(dz2, w2t, h1, tab, None, r_tab, n1, k)
    for name, fn in (('dgrad2', dg), ('wgrad1 + reduce', wg), ('fused bwd (identity rows, pipelined) + reduce', fz), ('fused bwd (no rows, single buffered) + reduce', fz0)):
        ts = sorted(timed(fn, iters) for _ in range(3))
        print('%-50s median %7.1f us' % (name, ts[1]))
    lib.mg_set_tuning(1, 1)
    for name, fn in (('wgrad1 alone', wg), ('fused bwd (identity rows) alone', fz), ('fused bwd (no rows) alone', fz0)):
        ts = sorted(timed(fn, iters) for _ in range(3))
        print('%-50s median %7.1f us' % (name, ts[1]))
    lib.mg_set_tuning(1, 0)


if __name__ == '__main__':
    main()
