#!/usr/bin/env python
"""Where the large bf16 GEMM kernels spend their cycles: runs them from the DIAGNOSTIC library (make -C morgana_amd/csrc
diag; in-kernel s_memtime stamps, cdna_hip_programming.md section 7) at the C2 shapes and prints the per-phase shares.
Read shares, not run times: the stamps' fences cost about 10 % and forbid overlaps the product build has.
Usage: MORGANA_HIP_LIB=morgana_amd/libmorgana_hip_diag.so python scripts/stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault('MORGANA_HIP_LIB', os.path.join(REPO, 'morgana_amd', 'libmorgana_hip_diag.so'))
from morgana_amd import _lib, ops, synthetic, data  # noqa: E402

SLOTS, BLOCKS = 16, 4096


def read(lib, name, n_blocks):
    buf = np.zeros(BLOCKS * 2 * SLOTS, dtype=np.uint64)
    fn = getattr(lib, name)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    rc = fn(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    assert rc == 0, rc
    return buf.reshape(BLOCKS, 2, SLOTS)[:n_blocks].astype(np.int64)


def report(title, st, extra):
    life = st[..., 3] - st[..., 0]
    real = (st[..., 5] - st[..., 4]) * 10.0           # ns (100 MHz)
    clk = life / np.maximum(real, 1) * 1e0             # cycles per ns = GHz
    print('%s: %d blocks; wave lifetime median %.0f cycles = %.2f us; in-kernel clock %.2f GHz' % (
        title, st.shape[0], np.median(life), np.median(real) / 1e3, np.median(clk)))
    parts = [('entry -> first tile landed', st[..., 1] - st[..., 0]), ('main loop', st[..., 2] - st[..., 1]),
             ('epilogue + store drain', st[..., 3] - st[..., 2])] + extra
    for half, tag in ((0, 'wave 0'), (1, 'wave 4')):
        print('  %s' % tag)
        for name, v in parts:
            print('    %-34s %9.0f cycles  %5.1f %%' % (name, np.median(v[:, half]), 100.0 * np.median(v[:, half] / life[:, half])))


def main():
    only_fused = len(sys.argv) > 1 and sys.argv[1] == 'fused'
    dev = 'cuda:0'
    lib = _lib.load()
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
    tab = ops.cast_pad_bf16(lab.view(b * p, k))
    w1b = ops.cast_pad_bf16(w1)
    w2t = ops.cast_transpose_bf16(w2)
    dz1 = (torch.randn(m, 512, device=dev) * 0.01).to(torch.bfloat16)
    dz2 = (torch.randn(m, 128, device=dev) * 0.01).to(torch.bfloat16)
    h1 = ops.linear_fwd_bf16(tab, rows, m, k, w1b, b1, 512, ops.ACT_SIGMOID)
    if not only_fused:
        for _ in range(3):
            h1 = ops.linear_fwd_bf16(tab, rows, m, k, w1b, b1, 512, ops.ACT_SIGMOID)
        torch.cuda.synchronize()
        s = read(lib, 'mg_diag_read_stamps_ntp', 256)
        report('gemm_nt_persist<256, sigmoid> (layer-1 forward; "main loop" spans all tiles of the workgroup)', s,
               [('  vmcnt wait + barrier', s[..., 6]), ('  epilogues', s[..., 7])])
        w2b = ops.cast_pad_bf16(w2)
        b2 = torch.from_numpy(st['layers.2.bias']).to(dev)
        for _ in range(3):
            ops.linear_fwd_bf16(h1, None, m, 512, w2b, b2, 128, ops.ACT_SIGMOID)
        torch.cuda.synchronize()
        s = read(lib, 'mg_diag_read_stamps_ntp', 256)
        report('gemm_nt_persist<128, sigmoid> (layer-2 forward)', s,
               [('  vmcnt wait + barrier', s[..., 6]), ('  epilogues', s[..., 7])])
        for _ in range(3):
            ops.linear_wgrad_bf16(dz1, tab, rows, m, 512, 600)
        torch.cuda.synchronize()
        s = read(lib, 'mg_diag_read_stamps_wg', 256)
        report('wgrad_big<10> (layer-1 weight gradient)', s,
               [('  loop: vmcnt wait + barrier', s[..., 6]), ('  loop: LDS-DMA issue', s[..., 7])])
    for knob, title in ((0, 'wgrad_fused64<3>'), (12, 'wgrad_fused64<2>')):
        lib.mg_set_tuning(0, knob)
        for _ in range(3):
            ops.linear_bwd_fused_bf16(dz2, w2t, h1, tab, rows, m, 512, 600)
        torch.cuda.synchronize()
        lib.mg_set_tuning(0, 0)
        s = read(lib, 'mg_diag_read_stamps_f64', 256)
        report(title + ' (dgrad2 + wgrad1, gathered input, 64-frame steps)', s,
               [('  loop: wait + barrier', s[..., 6]), ('  loop: run scan of the next step', s[..., 10]), ('  loop: fetch (DMA issue)', s[..., 7]),
                ('  loop: P1', s[..., 8]), ('  loop: P2', s[..., 9])])
    for _ in range(3):
        ops.linear_bwd_fused2_slabs_bf16(dz2, w2t, h1, tab, rows, m, 512, 600)
    torch.cuda.synchronize()
    s = read(lib, 'mg_diag_read_stamps_f3', 256)
    report('wgrad_fused3 (dgrad2 + wgrad1 + wgrad2, 64-frame steps, W2^T in LDS, two tile buffers)', s,
           [('  loop: wait + barrier', s[..., 6]), ('  loop: run scan of the next step', s[..., 10]), ('  loop: fetch (DMA issue)', s[..., 7]),
            ('  loop: P1 + P3', s[..., 8]), ('  loop: P2', s[..., 9])])
    lib.mg_set_tuning(0, 15)
    for _ in range(3):
        ops.linear_bwd_fused_bf16(dz2, w2t, h1, tab, rows, m, 512, 600)
    torch.cuda.synchronize()
    lib.mg_set_tuning(0, 0)
    s = read(lib, 'mg_diag_read_stamps_f64w', 256)
    report('wgrad_fused64w (woven stream)', s, [('  loop: wait + barrier', s[..., 6]), ('  loop: the stream', s[..., 7])])
    lib.mg_set_tuning(0, 13)
    for _ in range(3):
        ops.linear_bwd_fused_bf16(dz2, w2t, h1, tab, rows, m, 512, 600)
    torch.cuda.synchronize()
    lib.mg_set_tuning(0, 0)
    s = read(lib, 'mg_diag_read_stamps_fp', 256)
    report('wgrad_fused_pipe (dgrad2 + wgrad1, gathered input, 32-frame steps: the kernel it replaced)', s,
           [('  loop: barrier', s[..., 6]), ('  loop: DMA issue + look-ahead reads', s[..., 7]), ('  loop: P1 and P2', s[..., 8]),
            ('    of it: P2', s[..., 9]), ('    of it: what precedes P2 (waves 0-3: P1)', s[..., 10])])


if __name__ == '__main__':
    main()
