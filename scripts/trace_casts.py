#!/usr/bin/env python
"""Which Python call sites launch cast_pad_bf16 / cast_transpose_bf16 / torch element-wise ops in one training step of a recurrent
config (C4 by default): shapes and the two innermost morgana_amd frames per call.  Usage: python scripts/trace_casts.py [c4|lstm]"""
import os
import sys
import traceback
from collections import Counter

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from morgana_amd import data, models, ops, optim, synthetic  # noqa: E402
from morgana_amd import functional as F_hip  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'c4'
    dev = 'cuda:0'
    if which == 'c4':
        feats_np = synthetic.make_batch(64, 1000, out_dim=80, target_name='mcep')
        model = models.RNNSPSS(precision='bf16')
        acoustic = False
    else:
        feats_np = synthetic.make_acoustic_batch(64, 1000, with_raw=True)
        model = models.LSTMAcousticModel(precision='bf16', generate=True)
        acoustic = True
    model = model.to(dev)
    if acoustic:
        synthetic.acoustic_normalisers(model, device=dev)
        model.mode = 'train'
        model.metrics.reset_state('train')
    feats = data.to_device(feats_np, dev, bf16_tables=model.bf16_table_features())
    opt = optim.Adam(model.parameters(), lr=0.01, fused_loop=True)

    def step():
        opt.zero_grad()
        loss, _ = model(feats)
        F_hip.backward(loss)
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    seen = Counter()

    def wrap(name):
        orig = getattr(ops, name)

        def traced(x, *a, **k):
            frames = [f for f in traceback.extract_stack()[:-1] if 'morgana_amd' in f.filename][-3:]
            seen[(name, tuple(x.shape), ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(frames)))] += 1
            return orig(x, *a, **k)
        setattr(ops, name, traced)
    for n in ('cast_pad_bf16', 'cast_transpose_bf16', 'cast_params_bf16'):
        wrap(n)
    step()
    torch.cuda.synchronize()
    for (name, shape, where), c in sorted(seen.items(), key=lambda kv: -kv[1]):
        print('%-20s %-18s x%d  %s' % (name, shape, c, where))
    # torch ops of one step
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
        step()
    torch.cuda.synchronize()
    for evt in sorted(prof.key_averages(), key=lambda e: -e.count)[:25]:
        if evt.key.startswith('aten::'):
            print('%-40s x%d' % (evt.key, evt.count))


if __name__ == '__main__':
    main()
