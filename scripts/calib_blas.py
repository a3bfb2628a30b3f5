#!/usr/bin/env python
"""Calibration only (never on the product path): what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on this box for the three
large products of the frame-rate step, on random bf16 operands.  Gives the kernels in csrc/ a same-box, same-data yardstick."""
import sys
import torch


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dev = 'cuda:0'
    m = 256000
    x = torch.randn(m, 640, device=dev).to(torch.bfloat16)
    w1 = torch.randn(512, 640, device=dev).to(torch.bfloat16)
    h1 = torch.randn(m, 512, device=dev).to(torch.bfloat16)
    dz1 = torch.randn(m, 512, device=dev).to(torch.bfloat16)
    dz2 = torch.randn(m, 128, device=dev).to(torch.bfloat16)
    w2 = torch.randn(128, 512, device=dev).to(torch.bfloat16)
    out1 = torch.empty(m, 512, device=dev, dtype=torch.bfloat16)
    cases = [
        ('fwd1  [M,640]x[640,512]', 2.0 * m * 640 * 512, lambda: torch.matmul(x, w1.t(), out=out1)),
        ('wgrad1 [512,M]x[M,640]', 2.0 * m * 640 * 512, lambda: torch.matmul(dz1.t(), x)),
        ('wgrad2 [128,M]x[M,512]', 2.0 * m * 128 * 512, lambda: torch.matmul(dz2.t(), h1)),
        ('dgrad2 [M,128]x[128,512]', 2.0 * m * 128 * 512, lambda: torch.matmul(dz2, w2, out=out1)),
        ('square 8192^3', 2.0 * 8192 ** 3, None),
    ]
    a8 = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    b8 = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    cases[-1] = (cases[-1][0], cases[-1][1], lambda: torch.matmul(a8, b8.t()))
    for name, flops, fn in cases:
        ms = timeit(fn)
        print('%-28s %8.1f us  %7.1f TFLOP/s' % (name, ms * 1e3, flops / ms / 1e9), flush=True)


if __name__ == '__main__':
    main()
