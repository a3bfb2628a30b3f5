#!/usr/bin/env python
"""C4 / C5 step time with eager launches and as a HIP-graph replay (graphs.GraphedTrainStep).  Usage: python scripts/graph_vs_eager.py"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from morgana_amd import data, graphs, models, ops, optim, synthetic  # noqa: E402
from morgana_amd import functional as F_hip  # noqa: E402


def run(name, feats_np, model_kwargs, state):
    dev = 'cuda:0'
    for mode in ('eager', 'graph'):
        model = models.RNNSPSS(precision='bf16', **model_kwargs).to(dev)
        own = model.state_dict()
        for k, v in state.items():
            own[k].copy_(torch.from_numpy(v))
        feats = data.to_device(feats_np, dev)
        opt = optim.Adam(model.parameters(), lr=0.01, fused_loop=True)

        def step():
            opt.zero_grad()
            loss, _ = model(feats)
            F_hip.backward(loss)
            opt.step()
        if mode == 'graph':
            step = graphs.GraphedTrainStep(model, opt, feats, warmup=2)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        print('%s %s: %.3f ms/step' % (name, mode, (time.perf_counter() - t0) / 10 * 1e3))
        ops.check_persistent_status()
        import collections
        for key, ws in ops._PERSIST_WORKSPACES.items():          # where the last persistent launch's workgroups sat (XCC id per group)
            words = ws.view(torch.int32)[256:512].cpu().reshape(8, 32)
            print('   XCC ids per group: ' + '  '.join('g%d:%s' % (g, ','.join('%dx%d' % (k, n) for k, n in sorted(
                collections.Counter(int(v) - 1 for v in words[g] if int(v) > 0).items()))) for g in range(8)))


run('C4', synthetic.make_batch(64, 1000, out_dim=80, target_name='mcep'), {}, synthetic.rnn_spss_state())
run('C5', synthetic.make_batch(64, (300, 2000), out_dim=187, target_name='mcep'), {'output_dim': 187}, synthetic.rnn_spss_state(out_dim=187))
