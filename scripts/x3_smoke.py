"""Smoke of the fused 'bf16x3' phone-rate step (functional.F0StackX3Fn): against fp32 mode and the generic 'bf16x3' path on one batch,
then timed through graphs.GraphedTrainStep.  usage: python scripts/x3_smoke.py [B] [T]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morgana_amd import _lib, data, graphs, models, optim, synthetic, utils   # noqa: E402
from morgana_amd import functional as F_hip                                   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device('cuda:0')


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


feats_np = synthetic.make_batch(B, T)
state = synthetic.f0_model_state()
got = {}
for name, precision, fused in (('fp32', 'fp32', True), ('x3_generic', 'bf16x3', False), ('x3_fused', 'bf16x3', True)):
    utils.X3_FUSED = fused
    model = models.F0Model(precision=precision).to(dev)
    own = model.state_dict()
    for k, v in state.items():
        own[k].copy_(torch.from_numpy(v))
    feats = data.to_device(feats_np, dev, bf16_tables=model.bf16_table_features())
    _lib.CALL_LOG = []
    loss, out = model(feats)
    loss.backward()
    torch.cuda.synchronize()
    print(name, 'calls:', len(_lib.CALL_LOG), _lib.CALL_LOG if name == 'x3_fused' else '')
    _lib.CALL_LOG = None
    got[name] = (loss.item(), out['pred_norm_lf0'].detach().cpu().numpy(), {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()})
for name in ('x3_generic', 'x3_fused'):
    print(name, 'loss rel', abs(got[name][0] - got['fp32'][0]) / abs(got['fp32'][0]), 'pred rel', rel(got[name][1], got['fp32'][1]))
    for n in got['fp32'][2]:
        print('   grad', n, rel(got[name][2][n], got['fp32'][2][n]))

utils.X3_FUSED = True
for precision in ('bf16', 'bf16x3'):
    model = models.F0Model(precision=precision).to(dev)
    feats = data.to_device(feats_np, dev, bf16_tables=model.bf16_table_features())
    opt = optim.Adam(model.parameters(), lr=0.01, fused_loop=True)
    n = 10
    step = graphs.GraphedTrainStep(model, opt, feats, steps_per_replay=n)
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    print(precision, 'graph step ms', (time.perf_counter() - t0) / (20 * n) * 1e3, 'loss', float(step.loss) if hasattr(step, 'loss') else '')
