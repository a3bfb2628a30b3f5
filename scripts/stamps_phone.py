#!/usr/bin/env python
"""In-kernel stamps (diagnostic library, `make -C morgana_amd/csrc diag`) of the phone-rate step's GEMM kernels: M = 21 504 table rows,
one tile (or one split) per workgroup.  Usage: MORGANA_HIP_LIB=morgana_amd/libmorgana_hip_diag.so python scripts/stamps_phone.py"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'scripts'))
sys.path.insert(0, REPO)
os.environ.setdefault('MORGANA_HIP_LIB', os.path.join(REPO, 'morgana_amd', 'libmorgana_hip_diag.so'))
from stamps import read, report  # noqa: E402
from morgana_amd import _lib, ops  # noqa: E402


def main():
    dev = 'cuda:0'
    lib = _lib.load()
    r_tab, k, n1, n2 = 21504, 600, 512, 128
    tab = torch.rand(r_tab, 640, device=dev).to(torch.bfloat16)
    tab[:, 600:] = 0
    w1b = (torch.randn(n1, 640, device=dev) * 0.05).to(torch.bfloat16)
    w2b = (torch.randn(n2, n1, device=dev) * 0.05).to(torch.bfloat16)
    b1, b2 = torch.zeros(n1, device=dev), torch.zeros(n2, device=dev)
    dz1 = (torch.randn(r_tab, n1, device=dev) * 0.01).to(torch.bfloat16)
    dz2 = (torch.randn(r_tab, n2, device=dev) * 0.01).to(torch.bfloat16)
    h1 = None
    for _ in range(3):
        h1 = ops.linear_fwd_bf16(tab, None, r_tab, k, w1b, b1, n1, ops.ACT_SIGMOID)
    torch.cuda.synchronize()
    s = read(lib, 'mg_diag_read_stamps_ntp', 168)
    report('gemm_nt_persist<256> layer-1 forward at phone rate (168 workgroups, one 256 x 256 x 640 tile each)', s,
           [('  vmcnt wait + barrier', s[..., 6]), ('  epilogues', s[..., 7])])
    for _ in range(3):
        ops.linear_fwd_bf16(h1, None, r_tab, n1, w2b, b2, n2, ops.ACT_SIGMOID)
    torch.cuda.synchronize()
    s = read(lib, 'mg_diag_read_stamps_ntp', 84)
    report('gemm_nt_persist<128> layer-2 forward at phone rate (84 workgroups)', s,
           [('  vmcnt wait + barrier', s[..., 6]), ('  epilogues', s[..., 7])])
    for _ in range(3):
        ops.linear_wgrad_bf16(dz1, tab, None, r_tab, n1, k)
    torch.cuda.synchronize()
    s = read(lib, 'mg_diag_read_stamps_wg', 256)
    report('wgrad_big<5> layer-1 weight gradient at phone rate (32 splits x 8 tiles of 128 x 320, 21 steps each)', s,
           [('  loop: vmcnt wait + barrier', s[..., 6]), ('  loop: LDS-DMA issue', s[..., 7])])
    for _ in range(3):
        ops.linear_wgrad_bf16(dz2, h1, None, r_tab, n2, n1)
    torch.cuda.synchronize()
    s = read(lib, 'mg_diag_read_stamps_wg', 96)
    report('wgrad_big<4> layer-2 weight gradient at phone rate (48 splits x 2 tiles of 128 x 256, 14 steps each)', s,
           [('  loop: vmcnt wait + barrier', s[..., 6]), ('  loop: LDS-DMA issue', s[..., 7])])

    # layers 2-4 + loss + backward in one pass (one 32-row tile per wave at this row count)
    w3, b3 = torch.randn(32, 128, device=dev) * 0.1, torch.zeros(32, device=dev)
    w4, b4 = torch.randn(1, 32, device=dev) * 0.1, torch.zeros(1, device=dev)
    ybar, weight = torch.randn(r_tab, device=dev), torch.rand(r_tab, device=dev) / r_tab
    grads = torch.empty(4162, device=dev)
    for _ in range(3):
        ops.f0_l2tail_rows(h1, w2b, b2, w3, b3, w4, b4, ybar, weight, grads)
    torch.cuda.synchronize()
    s = read(lib, 'mg_diag_read_stamps_lt', 168)
    t_start, t_end = s[:, 0, 4], s[:, 0, 5]                      # 100 MHz real-time clock at wave start / end, wave 0 of each workgroup
    print('f0_l2tail_kernel: workgroup starts spread over %.2f us, ends over %.2f us; first start -> last end %.2f us' % (
        (t_start.max() - t_start.min()) / 100.0, (t_end.max() - t_end.min()) / 100.0, (t_end.max() - t_start.min()) / 100.0))
    report('f0_l2tail_kernel at phone rate (168 workgroups, one tile per wave; "main loop" = the tile, "entry" = the prologue)', s,
           [('  tile: loop top -> layer-2 MFMAs issued', s[..., 6]), ('  tile: the tail (sigmoid .. dW3)', s[..., 7]),
            ('  prologue: loads + W2 DMA issued', s[..., 8]), ('  prologue: W3 landed, cast, staged', s[..., 9]),
            ('  prologue: barrier', s[..., 10]), ('  prologue: fragment tables', s[..., 11]),
            ('  prologue: rest of DMA / first tile landed', s[..., 12]), ('  prologue: barrier + bias', s[..., 13])])


if __name__ == '__main__':
    main()
