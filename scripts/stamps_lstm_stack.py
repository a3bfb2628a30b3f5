#!/usr/bin/env python
"""Per-phase cycles per step of the LSTM stack wavefront kernels (diagnostic library, in-kernel stamps), by layer: the forward, or with
the argument `bwd` the backward (its `loads` = until the first hand-off tile has landed, `mfma+sum` = all four tile products).
Usage: MORGANA_HIP_LIB=morgana_amd/libmorgana_hip_diag.so python scripts/stamps_lstm_stack.py [bwd]"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault('MORGANA_HIP_LIB', os.path.join(REPO, 'morgana_amd', 'libmorgana_hip_diag.so'))
from morgana_amd import _lib, ops  # noqa: E402

SLOTS, BLOCKS = 16, 4096


def main():
    dev = 'cuda:0'
    lib = _lib.load()
    lib.mg_set_tuning(6, int(os.environ.get('MG_WIDTH', '0')))
    b, t, h, n_layers = 64, 1000, 512, 8
    g = torch.Generator(device=dev).manual_seed(0)
    xproj = torch.randn(b, t, 4 * h, device=dev, generator=g)
    w_ih = [torch.randn(4 * h, h, device=dev, generator=g) / h ** 0.5 for _ in range(n_layers)]
    w_hh = [torch.randn(4 * h, h, device=dev, generator=g) / h ** 0.5 for _ in range(n_layers)]
    bias = [torch.zeros(4 * h, device=dev) for _ in range(n_layers)]
    n_wg = 512
    for _ in range(3):
        _, _, cstate, saved, _ = ops.lstm_pstack_fwd(xproj, w_ih, w_hh, bias, bias, None, None, None, b, t, h)
    if len(sys.argv) > 1 and sys.argv[1] == 'bwd':
        g_out = torch.randn(b, t, h, device=dev, generator=g)
        for _ in range(3):
            ops.lstm_pstack_bwd(g_out, None, None, cstate, saved, w_ih, w_hh, None, b, t, h)
        n_wg = 256 if os.environ.get('MG_WIDTH', '0') == '0' else 512
    torch.cuda.synchronize()
    ops.check_persistent_status()
    buf = np.zeros(BLOCKS * 2 * SLOTS, dtype=np.uint64)
    fn = lib.mg_diag_read_stamps_lps
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert fn(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
    st = buf.reshape(BLOCKS, 2, SLOTS)[:n_wg, 0].astype(np.int64)
    life = st[:, 1] - st[:, 0]
    real_ns = (st[:, 3] - st[:, 2]) * 10.0
    steps = st[:, 9]
    layer = st[:, 10] // 1000
    print('wave 0 of %d workgroups, %d steps: loop %.0f cycles/step = %.2f us/step; clock %.2f GHz; same-XCD groups: %d of %d workgroups' % (
        len(st), int(np.median(steps)), np.median(life / steps), np.median(real_ns / steps) / 1e3, np.median(life / real_ns),
        int((st[:, 10] % 1000).sum()), len(st)))
    for l in range(n_layers):
        sel = layer == l
        parts = ['%s %5.0f' % (name, np.median(st[sel, col] / steps[sel])) for name, col in
                 (('poll', 4), ('loads', 5), ('mfma+sum', 6), ('cell', 7), ('publish', 8))]
        print('  layer %d: %s   (cycles per step, median over %d workgroups; total %.0f)' % (
            l, '  '.join(parts), int(sel.sum()), np.median(life[sel] / steps[sel])))


if __name__ == '__main__':
    main()
