#!/usr/bin/env python
"""
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (the lab library, before morgana_amd loads one)
The NT GEMM's schedule variants (MG_TUNE_FORM) must compute the same bits: layer-1 forward at the C2 frame-rate and phone-rate
shapes, ragged M, with and without the gather.  Usage: python scripts/check_nt_variants.py 0 2 4"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from morgana_amd import _lib, ops  # noqa: E402


def main():
    variants = [int(v) for v in sys.argv[1:]] or [0, 2]
    lib = _lib.load()
    dev = 'cuda:0'
    torch.manual_seed(0)
    bad = 0
    for m, r_tab, gather, k, n in ((256000, 20480, True, 600, 512), (21504, 21504, False, 600, 512), (4099, 700, True, 600, 512),
                                   (70000, 70000, False, 600, 512), (256000, 256000, False, 512, 128), (21504, 21504, False, 512, 128),
                                   (5000, 5000, False, 512, 128)):
        kp = (k + 63) // 64 * 64
        tab = torch.rand(r_tab, kp, device=dev).to(torch.bfloat16)
        tab[:, k:] = 0
        w = (torch.randn(n, kp, device=dev) * 0.05).to(torch.bfloat16)
        w[:, k:] = 0
        bias = torch.randn(n, device=dev) * 0.1
        rows = None
        if gather:
            rows = torch.sort(torch.randint(0, r_tab, (m,), device=dev, dtype=torch.int32)).values
            rows[::97] = -1
        outs = []
        for v in variants:
            lib.mg_set_tuning(0, v)
            for act in (ops.ACT_SIGMOID, ops.ACT_NONE):
                outs.append((v, act, ops.linear_fwd_bf16(tab, rows, m, k, w, bias, n, act).clone()))
        lib.mg_set_tuning(0, 0)
        torch.cuda.synchronize()
        ref = {act: o for v, act, o in outs if v == variants[0]}
        for v, act, o in outs:
            same = torch.equal(o, ref[act])
            bad += not same
            print('M=%6d N=%d gather=%d variant %d act %d: %s' % (m, n, gather, v, act, 'equal' if same else 'DIFFERENT (max %g)' % (o.float() - ref[act].float()).abs().max().item()))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
