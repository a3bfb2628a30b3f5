#!/usr/bin/env python
"""Per-phase cycle shares of the persistent GRU forward kernel (both directions), from the DIAGNOSTIC library's in-kernel stamps.
Usage: MORGANA_HIP_LIB=morgana_amd/libmorgana_hip_diag.so python scripts/stamps_gru.py   (MG_TUNE=2:1: the write-through hand-off)"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault('MORGANA_HIP_LIB', os.path.join(REPO, 'morgana_amd', 'libmorgana_hip_diag.so'))
os.environ.setdefault('MG_TUNE', '2:0')
from morgana_amd import _lib, ops  # noqa: E402

SLOTS, BLOCKS = 16, 4096


def main():
    dev = 'cuda:0'
    lib = _lib.load()
    b, t, h = 64, 1000, 512
    g = torch.Generator(device=dev).manual_seed(0)
    xproj = torch.randn(b, t, 3 * h, device=dev, generator=g)
    w_hh = torch.randn(3 * h, h, device=dev, generator=g) / h ** 0.5
    b_hh = torch.zeros(3 * h, device=dev)
    for _ in range(3):
        out, hstate, saved, _ = ops.gru_fwd_bf16(xproj, w_hh, b_hh, None, None, b, t, h, persistent=True)
    g_out = torch.randn(b, t, h, device=dev, generator=g)
    for _ in range(3):
        ops.gru_bwd_bf16(g_out, None, hstate, saved, w_hh, None, b, t, h, persistent=True)
    torch.cuda.synchronize()
    ops.check_persistent_status()
    for title, name in (('forward', 'mg_diag_read_stamps_gp'), ('backward', 'mg_diag_read_stamps_gpb')):
        buf = np.zeros(BLOCKS * 2 * SLOTS, dtype=np.uint64)
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        assert fn(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
        st = buf.reshape(BLOCKS, 2, SLOTS)[:256, 0].astype(np.int64)
        life = st[:, 1] - st[:, 0]
        real_ns = (st[:, 3] - st[:, 2]) * 10.0
        steps = st[:, 9]
        print('%s: wave 0 of %d workgroups, %d steps: loop %.0f cycles/step = %.2f us/step; clock %.2f GHz' % (
            title, len(st), int(np.median(steps)), np.median(life / steps), np.median(real_ns / steps) / 1e3, np.median(life / real_ns)))
        for label, col in (('poll + barrier', 4), ('hand-off loads landed', 5), ('MFMA + LDS partials + barrier', 6),
                           ('cell + LDS tile + barrier', 7), ('publish: stores, drain, flag', 8)):
            v = st[:, col] / steps
            print('  %-36s %7.0f cycles/step  %5.1f %%   (min %.0f max %.0f over workgroups)' % (
                label, np.median(v), 100 * np.median(st[:, col] / life), v.min(), v.max()))


if __name__ == '__main__':
    main()
