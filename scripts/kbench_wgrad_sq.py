#!/usr/bin/env python
"""The wide weight-gradient kernel at frame-rate row counts of a 512-wide operand: the square 256 x 256 tile (default where N % 256 == 0)
against the 128 x 512 tile (MG_TUNE_AB = 91) - results against a product of the same bf16 operands, kernel time alone (slabs left for
the update).   Usage: python scripts/kbench_wgrad_sq.py [iters]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from morgana_amd import ops, _lib  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = 'cuda:0'
    lib = _lib.load()
    torch.manual_seed(0)
    shapes = [('LSTM dW (64000 x 2048 x 512)', 64000, 2048, 512, 512),
              ('GRU dW (64000 x 1536 x 512)', 64000, 1536, 512, 512),
              ('ragged (74003 x 1536 x 500)', 74003, 1536, 500, 512),
              ('out layer (64000 x 256 x 512)', 64000, 256, 512, 512),
              ('N 512 (64000 x 512 x 512)', 64000, 512, 512, 512),
              ('K 400 (40000 x 1024 x 400)', 40000, 1024, 400, 512)]
    for name, m, n, k, lda in shapes:
        dz = (torch.randn(m, n, device=dev) * 0.05).to(torch.bfloat16)
        a = torch.zeros(m, lda, device=dev, dtype=torch.bfloat16)
        a[:, :k] = torch.randn(m, k, device=dev).to(torch.bfloat16)
        ref_w = dz.float().t() @ a[:, :k].float()
        ref_b = dz.float().sum(0)
        line = '%-32s' % name
        for ab in (0, 91):
            lib.mg_set_tuning(7, ab)
            dw, db = ops.linear_wgrad_bf16(dz, a, None, m, n, k)
            ew = ((dw - ref_w).abs().max() / ref_w.abs().max()).item()
            eb = ((db - ref_b).abs().max() / ref_b.abs().max()).item()
            slab = None
            for _ in range(2):
                slab, ns, stride = ops.linear_wgrad_slabs_bf16(dz, a, None, m, n, k, slab=slab)
            best = []
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    slab, ns, stride = ops.linear_wgrad_slabs_bf16(dz, a, None, m, n, k, slab=slab)
                e.record()
                e.synchronize()
                best.append(s.elapsed_time(e) / iters * 1e3)
            line += '   AB %2d: %7.1f us (S %3d, err %.1e / %.1e)' % (ab, min(best), ns, ew, eb)
        lib.mg_set_tuning(7, 0)
        print(line, flush=True)


if __name__ == '__main__':
    main()
