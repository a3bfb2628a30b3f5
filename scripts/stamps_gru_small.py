#!/usr/bin/env python
"""Per-phase cycle shares of the small GRU stack's forward wavefront (gru_stack_fwd_small64_kernel), from the DIAGNOSTIC library's stamps.
Usage: MORGANA_HIP_LIB=morgana_amd/libmorgana_hip_diag.so python scripts/stamps_gru_small.py"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault('MORGANA_HIP_LIB', os.path.join(REPO, 'morgana_amd', 'libmorgana_hip_diag.so'))
from morgana_amd import _lib, data, models, synthetic  # noqa: E402
from morgana_amd import functional as F_hip  # noqa: E402

SLOTS, BLOCKS = 16, 4096


def main():
    dev = 'cuda:0'
    lib = _lib.load()
    feats = data.to_device(synthetic.make_acoustic_batch(64, 1000, streams=(('lf0', 3, 'mse'),), seed=5), dev)
    model = models.GRUF0Model(precision='bf16', generate=False).to(dev)
    for _ in range(3):
        loss, _ = model(feats)
        F_hip.backward(loss)
    torch.cuda.synchronize()
    buf = np.zeros(BLOCKS * 2 * SLOTS, dtype=np.uint64)
    fn = lib.mg_diag_read_stamps_gss
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert fn(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
    st = buf.reshape(BLOCKS, 2, SLOTS)[:48, 0].astype(np.int64)
    steps = 1000
    for layer in range(3):
        rows = st[st[:, 9] == layer]
        life = rows[:, 1] - rows[:, 0]
        real_ns = (rows[:, 3] - rows[:, 2]) * 10.0
        print('layer %d: wave 0 of %d workgroups: loop %.0f cycles/step = %.2f us/step; clock %.2f GHz' % (
            layer, len(rows), np.median(life) / steps, np.median(real_ns) / steps / 1e3, np.median(life / real_ns)))
        for label, col in (('MFMA + LDS writes', 4), ('barrier 1', 5), ('cell + stores issued', 6), ('take x of the layer below', 7), ('barrier 2', 8)):
            print('  %-28s %7.0f cycles/step' % (label, np.median(rows[:, col]) / steps))


if __name__ == '__main__':
    main()
