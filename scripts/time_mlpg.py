"""Time ops.mlpg at the shipped models' stream shapes (64 x 1000 frames, padding 100): lf0 (1 dim), bap (5), mcep (60)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morgana_amd import ops
from morgana_amd.viz import synthesis

rng = np.random.RandomState(0)
b, t = 64, 1000
seq = torch.full((b,), t, dtype=torch.int64, device='cuda')
for name, d in (('lf0', 1), ('bap', 5), ('mcep', 60)):
    means = torch.from_numpy(rng.standard_normal((b, t, 3 * d)).astype(np.float32)).cuda()
    var = torch.from_numpy(rng.uniform(0.1, 1.0, 3 * d).astype(np.float32)).cuda()
    for _ in range(3):
        ops.mlpg(means, var, synthesis.DEFAULT_WINDOWS, padding_size=100, seq_len=seq)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.mlpg(means, var, synthesis.DEFAULT_WINDOWS, padding_size=100, seq_len=seq)
    e1.record()
    torch.cuda.synchronize()
    print('%-5s D=%2d  %d systems x %d unknowns: %.3f ms per call' % (name, d, b * d, t + 200, e0.elapsed_time(e1) / 20))
