#!/usr/bin/env python
"""Time the MLPG launches of the shipped acoustic model's three delta streams (64 x 1000 frames): usage kbench_mlpg.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morgana_amd import ops  # noqa: E402
from morgana_amd.viz import synthesis  # noqa: E402

dev = 'cuda:0'
b, t = 64, 1000
g = torch.Generator(device=dev).manual_seed(0)
seq = torch.full((b,), t, dtype=torch.int64, device=dev)
for name, d in (('lf0', 1), ('bap', 5), ('mcep', 60)):
    means = torch.randn(b, t, 3 * d, device=dev, generator=g)
    var = torch.rand(3 * d, device=dev, generator=g) + 0.5
    for _ in range(3):
        out = ops.mlpg(means, var, synthesis.DEFAULT_WINDOWS, padding_size=100, seq_len=seq)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = ops.mlpg(means, var, synthesis.DEFAULT_WINDOWS, padding_size=100, seq_len=seq)
    e1.record()
    torch.cuda.synchronize()
    print('%-5s D=%2d  %.1f us per call' % (name, d, e0.elapsed_time(e1) * 1000 / 20))
