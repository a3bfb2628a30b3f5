#!/usr/bin/env python
"""In-kernel cycles and clock of the layer-1 forward (gemm_nt_persist<256>) under the timing probes of MG_TUNE_STAGGER, from the
DIAGNOSTIC library, after a second of back-to-back launches per variant (the chip lowers its clock under MFMA load: compare
cycles AND microseconds).  Usage: MG_VARIANTS=0,48,55,60 python scripts/stamps_nt_probe.py [launches]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault('MORGANA_HIP_LIB', os.path.join(REPO, 'morgana_amd', 'libmorgana_hip_diag.so'))
from morgana_amd import _lib, ops, synthetic, data  # noqa: E402
from stamps import read  # noqa: E402


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    dev = 'cuda:0'
    lib = _lib.load()
    feats = data.to_device(synthetic.make_batch(256, 1000), dev)
    lab = feats['normalised_lab']
    b, p, k = lab.shape
    m = b * 1000
    _, rows = ops.upsample_index(feats['dur'].reshape(b, -1).contiguous(), 1000)
    rows = rows.view(-1)
    st = synthetic.f0_model_state()
    w1b = ops.cast_pad_bf16(torch.from_numpy(st['layers.0.weight']).to(dev))
    b1 = torch.from_numpy(st['layers.0.bias']).to(dev)
    tab = ops.cast_pad_bf16(lab.view(b * p, k))
    for v in [int(x) for x in os.environ.get('MG_VARIANTS', '0').split(',')]:
        lib.mg_set_tuning(0, v)
        for _ in range(launches):
            ops.linear_fwd_bf16(tab, rows, m, k, w1b, b1, 512, ops.ACT_SIGMOID)
        torch.cuda.synchronize()
        s = read(lib, 'mg_diag_read_stamps_ntp', 256)
        life = s[..., 3] - s[..., 0]
        real = (s[..., 5] - s[..., 4]) * 10.0
        loop = s[..., 2] - s[..., 1]
        print('variant %3d: wave 0 lifetime %7.0f cycles = %6.1f us, clock %.2f GHz; main loop %7.0f cycles (%.0f per k-step of 152); '
              'vmcnt wait + barrier %6.0f (wave 0) %6.0f (wave 4); epilogues %6.0f' % (
                  v, np.median(life[:, 0]), np.median(real[:, 0]) / 1e3, np.median(life / np.maximum(real, 1)),
                  np.median(loop[:, 0]), np.median(loop[:, 0]) / 152.0, np.median(s[:, 0, 6]), np.median(s[:, 1, 6]),
                  np.median(s[:, 0, 7])))
    lib.mg_set_tuning(0, 0)


if __name__ == '__main__':
    main()
