#!/usr/bin/env python
"""The persistent bf16 GRU backward at the C4 shape (64 x 1000 x 512): wide slots (16 groups x 32 units; default) against narrow
(8 groups x 16 units; MG_TUNE_GRU_HANDOFF bit 2), kernel time of the form training uses (bf16 shadows only) and with fp32 results.
Usage: python scripts/kbench_gru_bwd.py [iters]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from morgana_amd import ops, _lib  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = 'cuda:0'
    lib = _lib.load()
    b, t, h = 64, 1000, 512
    g = torch.Generator(device=dev).manual_seed(0)
    xproj = torch.randn(b, t, 3 * h, device=dev, generator=g)
    w_hh = torch.randn(3 * h, h, device=dev, generator=g) / h ** 0.5
    b_hh = torch.zeros(3 * h, device=dev)
    out, hstate, saved, _ = ops.gru_fwd_bf16(xproj, w_hh, b_hh, None, None, b, t, h, persistent=True)
    g_out = torch.randn(b, t, h, device=dev, generator=g)
    for shadows in (True, False):
        for rnd in range(2):
            for tune in (0, 4):
                lib.mg_set_tuning(2, tune)
                for _ in range(2):
                    ops.gru_bwd_bf16(g_out, None, hstate, saved, w_hh, None, b, t, h, persistent=True, shadows_only=shadows)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    ops.gru_bwd_bf16(g_out, None, hstate, saved, w_hh, None, b, t, h, persistent=True, shadows_only=shadows)
                e.record()
                e.synchronize()
                ops.check_persistent_status()
                print('shadows_only %-5s  %s  %8.1f us per call' % (shadows, 'wide  ' if tune == 0 else 'narrow', s.elapsed_time(e) / iters * 1e3), flush=True)
    lib.mg_set_tuning(2, 0)


if __name__ == '__main__':
    main()
