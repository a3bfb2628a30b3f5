#!/usr/bin/env python
"""A few eager C2 steps in precision 'bf16x3' at phone rate (for rocprofv3 --pmc passes: the bench's clock-ramp warm-up is too long under counters)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morgana_amd import data, models, optim, synthetic  # noqa: E402
from morgana_amd import functional as F_hip  # noqa: E402

dev = 'cuda:0'
feats = data.to_device(synthetic.make_batch(256, 1000, seed=1), dev)
model = models.F0Model(precision=os.environ.get('PRECISION', 'bf16x3'), phone_rate=os.environ.get('PHONE_RATE', '1') == '1').to(dev)
opt = optim.Adam(model.parameters(), lr=1e-3)
for i in range(int(os.environ.get('STEPS', '6'))):
    opt.zero_grad()
    loss, _ = model(feats)
    F_hip.backward(loss)
    opt.step()
torch.cuda.synchronize()
print(float(loss))
