#!/usr/bin/env python
"""The LSTM stack's backward alone at the acoustic model's shape (64 x 1000 frames, 8 x LSTM-512): the wavefront launch
(mg_lstm_pstack_bwd_bf16) with 32 and with 16 hidden units per slot, and the layer-by-layer form it replaces (8 persistent launches
+ 7 input-gradient GEMMs).  HIP-event time per backward.  Usage: python scripts/kbench_lstm_bwd.py [iters]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from morgana_amd import _lib, ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = 'cuda:0'
    b, t, hid, n_layers = 64, 1000, 512, 8
    torch.manual_seed(0)
    lib = _lib.load()
    seq_len = torch.randint(300, t + 1, (b,), device=dev, dtype=torch.int64)
    seq_len[0] = t
    w_ih = [torch.randn(4 * hid, hid, device=dev) / hid ** 0.5 for _ in range(n_layers)]
    w_hh = [torch.randn(4 * hid, hid, device=dev) / hid ** 0.5 for _ in range(n_layers)]
    b_ih = [torch.randn(4 * hid, device=dev) * 0.1 for _ in range(n_layers)]
    b_hh = [torch.randn(4 * hid, device=dev) * 0.1 for _ in range(n_layers)]
    xproj0 = torch.randn(b, t, 4 * hid, device=dev)
    _, _, cstate, saved, _ = ops.lstm_pstack_fwd(xproj0, w_ih, w_hh, b_ih, b_hh, seq_len, None, None, b, t, hid)
    g_out = torch.randn(b, t, hid, device=dev)
    m = b * t

    def layerwise():
        g = g_out
        for l in range(n_layers - 1, -1, -1):
            _, _, _, dg_bf = ops.lstm_bwd_bf16(g, None, None, cstate[l], saved[l], w_hh[l], seq_len, b, t, hid, want_f32=False)
            if l > 0:
                g = ops.linear_dgrad_bf16(dg_bf.view(m, 4 * hid), m, 4 * hid, ops.cast_transpose_bf16(w_ih[l]), hid, None,
                                          out_f32=True).view(b, t, hid)

    def wavefront():
        ops.lstm_pstack_bwd(g_out, None, None, cstate, saved, w_ih, w_hh, seq_len, b, t, hid)

    for name, fn, width in (('wavefront, 32 units per slot', wavefront, 0), ('wavefront, 16 units per slot', wavefront, 1),
                            ('layer by layer', layerwise, 0)):
        lib.mg_set_tuning(6, width)
        fn()
        torch.cuda.synchronize()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            fn()
        end.record()
        end.synchronize()
        ops.check_persistent_status()
        print('%-32s %8.3f ms per backward' % (name, start.elapsed_time(end) / iters))
    lib.mg_set_tuning(6, 0)


if __name__ == '__main__':
    main()
