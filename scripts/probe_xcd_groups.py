#!/usr/bin/env python
"""Which XCD does every workgroup of the persistent GRU launches land on?  The kernels publish their XCC id per (group, slot) at start
(persist_common.h: gp_group_on_one_xcd) and take the fast L2 hand-off only when a group's slots all agree.  This runs C4 steps one
by one, and after every step that took longer than `--slow` ms prints the table the LAST launch left: per group the XCC ids seen.
Usage: python scripts/probe_xcd_groups.py [--steps 400] [--slow 4.5]"""
import argparse
import collections
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from morgana_amd import data, models, ops, optim, synthetic  # noqa: E402
from morgana_amd import functional as F_hip  # noqa: E402


def table(ws):
    words = ws.view(torch.int32)[256:512].cpu().reshape(8, 32)          # XCC id + 1 per (group, slot); 0 = never written
    out = []
    for g in range(8):
        cnt = collections.Counter(int(v) - 1 for v in words[g] if int(v) > 0)
        out.append('g%d:%s' % (g, ','.join('%dx%d' % (k, n) for k, n in sorted(cnt.items()))))
    return '  '.join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--slow', type=float, default=4.5)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    feats = data.to_device(synthetic.make_batch(64, 1000, out_dim=80, target_name='mcep'), dev)
    model = models.RNNSPSS(precision='bf16').to(dev)
    own = model.state_dict()
    for k, v in synthetic.rnn_spss_state().items():
        own[k].copy_(torch.from_numpy(v))
    opt = optim.Adam(model.parameters(), lr=0.01, fused_loop=True)
    times, shown = [], 0
    for i in range(args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _ = model(feats)
        F_hip.backward(loss)
        opt.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        times.append(ms)
        if i == 2 or (ms > args.slow and i > 2 and shown < 12):
            shown += 1
            for key, ws in ops._PERSIST_WORKSPACES.items():
                print('step %d %.2f ms  workspace %s: %s' % (i, ms, key[2:] if len(key) > 2 else key, table(ws)), flush=True)
    ts = sorted(times[3:])
    print('steps %d: median %.2f ms, max %.2f, over %.1f ms: %d' % (len(ts), ts[len(ts) // 2], ts[-1], args.slow, sum(1 for t in ts if t > args.slow)))
    ops.check_persistent_status()


if __name__ == '__main__':
    main()
