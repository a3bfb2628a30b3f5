"""Kernel timeline of ExperimentBuilder.train_epoch(use_graphs=True) over distinct C2 batches (run under rocprofv3 --kernel-trace):
prints nothing itself; scripts/train_epoch_gaps.py reads the trace.  usage: python scripts/train_epoch_trace.py [precision] [n_batches] [steps per graph]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morgana_amd import data, experiment_builder, models, synthetic   # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
group = int(sys.argv[3]) if len(sys.argv) > 3 else 10          # steps per graph over the resident list (1 = the per-batch load path)
dev = torch.device('cuda:0')
host = [synthetic.make_batch(256, 1000, seed=synthetic.REFERENCE_SEED + 1000 + i) for i in range(2)]
eb = experiment_builder.ExperimentBuilder(models.F0Model, model_kwargs={'precision': precision}, learning_rate=0.01, device=dev, use_graphs=True,
                                          graph_group=group)
batches = []
for i in range(n):
    b = data.to_device(host[i % 2], dev)
    b['normalised_lab'] = torch.rand(b['normalised_lab'].shape, device=dev)
    for name in eb.model.bf16_table_features():
        data.add_bf16_table(b, name)
    batches.append(b)
opt = eb.make_optimizer()
eb.train_epoch(batches, opt)
eb.train_epoch(batches, opt)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
eb.train_epoch(batches, opt)
print('ms per step', (time.perf_counter() - t0) / n * 1e3, eb.last_epoch_stats, eb._graph_cache.stats())
