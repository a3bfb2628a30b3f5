#!/usr/bin/env python
"""GEMMs at the phone-rate row counts of the recurrent models (RNN_SPSS at C4 / C5: 6,144 table rows), tile choice A/B:
MG_TUNE_AB 0 = the launcher's rule, 97 = 256-wide tiles whenever N allows (the rule before), 99 = 128-wide, 98 = the 128 x 128 kernel."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from morgana_amd import _lib, ops  # noqa: E402


def main():
    dev = 'cuda:0'
    lib = _lib.load()
    rows = [int(v) for v in os.environ.get('MG_ROWS', '6144,12288,24576').split(',')]
    for m in rows:
        shapes = [('fwd 600->512 sigmoid', 600, 512, 'fwd_sig'), ('fwd 512->1536', 512, 1536, 'fwd'), ('dgrad 1536->512', 1536, 512, 'dgrad'),
                  ('fwd 512->256', 512, 256, 'fwd')]
        for name, k, n, kind in shapes:
            a = (torch.randn(m, ops.pad_ld(k), device=dev) * 0.5).to(torch.bfloat16)
            w = (torch.randn(n, ops.pad_ld(k), device=dev) * 0.05).to(torch.bfloat16)
            bias = torch.randn(n, device=dev)
            if kind == 'dgrad':
                fn = lambda: ops.linear_dgrad_bf16(a, m, k, w, n, None, out_f32=True)      # dY [m, k] . WT [n, k]^T -> dX [m, n]
            else:
                act = ops.ACT_SIGMOID if kind == 'fwd_sig' else ops.ACT_NONE
                fn = lambda: ops.linear_fwd_bf16(a, None, m, k, w, bias, n, act, out_f32=(kind == 'fwd'))
            res = []
            for v in (0, 97, 99, 98):
                lib.mg_set_tuning(7, v)
                for _ in range(3):
                    fn()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        fn()
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
                res.append('%d: %6.1f' % (v, sorted(ts)[2]))
            lib.mg_set_tuning(7, 0)
            print('M %6d  %-22s us per launch  %s' % (m, name, '   '.join(res)), flush=True)


if __name__ == '__main__':
    main()
