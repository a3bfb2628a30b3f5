#!/usr/bin/env python
"""Micro-benchmark of the individual GEMM kernels at the C2 shapes (HIP-event timing on the launch stream).
Usage: python scripts/kbench.py [iters] [which,comma,separated]   (which: fwd1,fwd1r,fwd2,dgrad2,wgrad1,wgrad2,fused; fwd1r = fwd1 with the run-staged loader)"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# experiments and timing probes live in the lab build (make -C morgana_amd/csrc lab); the product library refuses their knobs
if os.path.exists(os.path.join(REPO, 'morgana_amd', 'libmorgana_hip_lab.so')):
    os.environ.setdefault('MORGANA_HIP_LIB', os.path.join(REPO, 'morgana_amd', 'libmorgana_hip_lab.so'))
sys.path.insert(0, REPO)
from morgana_amd import ops, synthetic, data  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    which = sys.argv[2].split(',') if len(sys.argv) > 2 else ['fwd1', 'fwd2', 'dgrad2', 'wgrad1', 'wgrad2']
    dev = 'cuda:0'
    feats = data.to_device(synthetic.make_batch(256, 1000), dev)
    lab = feats['normalised_lab']
    b, p, k = lab.shape
    t = 1000
    m = b * t
    _, rows = ops.upsample_index(feats['dur'].reshape(b, -1).contiguous(), t)
    rows = rows.view(-1)
    st = synthetic.f0_model_state()
    w1 = torch.from_numpy(st['layers.0.weight']).to(dev)
    b1 = torch.from_numpy(st['layers.0.bias']).to(dev)
    w2 = torch.from_numpy(st['layers.2.weight']).to(dev)
    b2 = torch.from_numpy(st['layers.2.bias']).to(dev)
    tab = ops.cast_pad_bf16(lab.view(b * p, k))
    w1b, w2b = ops.cast_pad_bf16(w1), ops.cast_pad_bf16(w2)
    w2t = ops.cast_transpose_bf16(w2)
    h1 = ops.linear_fwd_bf16(tab, rows, m, k, w1b, b1, 512, ops.ACT_SIGMOID)
    dz1 = (torch.randn(m, 512, device=dev) * 0.01).to(torch.bfloat16)
    dz2 = (torch.randn(m, 128, device=dev) * 0.01).to(torch.bfloat16)
    if os.environ.get('MG_ZERO') == '1':                  # power probe: the same launches on all-zero operands (no bit toggles in the
        for tns in (tab, w1b, w2b, w2t, h1, dz1, dz2):    # matrix pipe or on the memory buses; MI355X_MICROARCH.md, DVFS give-back item 1)
            tns.zero_()
        b1, b2 = torch.zeros_like(b1), torch.zeros_like(b2)
        print('all operands zero (power probe)')
    bufs = {}

    def keep(key, result):
        bufs[key] = result[0]

    cases = {
        'fwd1': (2.0 * m * 600 * 512, lambda: ops.linear_fwd_bf16(tab, rows, m, k, w1b, b1, 512, ops.ACT_SIGMOID)),
        'fwd1r': (2.0 * m * 600 * 512, lambda: ops.linear_fwd_bf16(tab, rows, m, k, w1b, b1, 512, ops.ACT_SIGMOID, rows_runs=True)),
        'fwd2': (2.0 * m * 512 * 128, lambda: ops.linear_fwd_bf16(h1, None, m, 512, w2b, b2, 128, ops.ACT_SIGMOID)),
        'dgrad2': (2.0 * m * 512 * 128, lambda: ops.linear_dgrad_bf16(dz2, m, 128, w2t, 512, h1)),
        'wgrad1': (2.0 * m * 600 * 512, lambda: ops.linear_wgrad_bf16(dz1, tab, rows, m, 512, 600)),
        'wgrad2': (2.0 * m * 512 * 128, lambda: ops.linear_wgrad_bf16(dz2, h1, None, m, 128, 512)),
        'fused': (2.0 * m * 600 * 512 + 2.0 * m * 512 * 128, lambda: ops.linear_bwd_fused_bf16(dz2, w2t, h1, tab, rows, m, 512, 600)),
        # the kernels alone (their slabs left for the update kernel, as the training step does): fused backward, layer-2 weight gradient,
        # and the fused backward that carries the layer-2 weight gradient (one launch for both)
        'fuseds': (2.0 * m * 600 * 512 + 2.0 * m * 512 * 128, lambda: keep('f', ops.linear_bwd_fused_slabs_bf16(dz2, w2t, h1, tab, rows, m, 512, 600, slab=bufs.get('f')))),
        'wgrad2s': (2.0 * m * 512 * 128, lambda: keep('w', ops.linear_wgrad_slabs_bf16(dz2, h1, None, m, 128, 512, slab=bufs.get('w')))),
        'fused2': (2.0 * m * 600 * 512 + 4.0 * m * 512 * 128, lambda: keep('f2', ops.linear_bwd_fused2_slabs_bf16(dz2, w2t, h1, tab, rows, m, 512, 600, slab=bufs.get('f2')))),
    }
    from morgana_amd import _lib
    lib = _lib.load()
    variants = [int(v) for v in os.environ.get('MG_VARIANTS', '0').split(',')]
    rounds = int(os.environ.get('MG_ROUNDS', '3'))
    for name in which:
        flops, fn = cases[name]
        for _ in range(2):
            fn()
        # interleaved rounds in one process: variants x rounds, median and min per variant
        times = {v: [] for v in variants}
        for _ in range(rounds):
            for v in variants:
                lib.mg_set_tuning(0, v)
                fn()
                start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record()
                for _ in range(iters):
                    fn()
                end.record()
                end.synchronize()
                times[v].append(start.elapsed_time(end) / iters)
        lib.mg_set_tuning(0, 0)
        for v in variants:
            ts = sorted(times[v])
            med = ts[len(ts) // 2]
            print('%-8s variant %d  median %8.1f us (min %8.1f)  %8.1f TFLOP/s' % (name, v, med * 1e3, ts[0] * 1e3, flops / med / 1e9))


if __name__ == '__main__':
    main()
