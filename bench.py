#!/usr/bin/env python
"""bench.py - acoustic frames/sec of one training step (zero_grad + forward + backward + [all-reduce] + Adam) of the
README F0Model 600->512->128->32->1 on synthetic 256 x 1000-frame utterance batches per GPU (BASELINE config C2; C3 at N>1).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Inputs are resident in HBM before the timed region.  `roofline` is measured live with
HIP events on the kernel's own stream for the dominant kernel; `cpu_baseline` times the oracle's torch-CPU restatement
of the reference step on a bounded sample on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from morgana_amd import _lib, data, distributed, models, ops, optim, synthetic  # noqa: E402
from morgana_amd import functional as F_hip  # noqa: E402

WARM_STEPS_GRAPH = 400        # graph-replayed C2 steps of warm-up in front of the clock (~40 ms of device work)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
# Algorithmic FLOPs per frame of the F0Model step: fwd 753,728 + wgrad 753,728 + dgrad(L2-4) 139,328
F0_FLOPS_PER_FRAME = 1646784.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None,
                    help='timed steps (default 200 for c2 - a 0.13 ms step: the barrier + synchronize around the timed region and the gaps '
                         'between graph launches weigh 4 %% on 30 steps - and 30 for the recurrent configs)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed steps in front (default 20 for c2, 5 otherwise)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32', 'bf16x3'])
    ap.add_argument('--config', default='c2', choices=['c2', 'c4', 'c5', 'lstm', 'f0gru'],
                    help='c2: F0Model 256x1000 (headline); c4: GRU-512 600->80, 64x1000; c5: GRU-512 600->187, 64 ragged 300-2000; '
                         'lstm: the shipped LSTMAcousticModel 609->512->8xLSTM-512->256->199, 64x1000')
    ap.add_argument('--batch', type=int, default=None, help='utterances per GPU (default 256 for c2, 64 for c4)')
    ap.add_argument('--frames', type=int, default=1000)
    ap.add_argument('--no-generate', action='store_true',
                    help='lstm / f0gru: leave out the MLPG + metrics part of the step (the reference runs it inside predict / loss)')
    ap.add_argument('--steps-per-replay', type=int, default=0,
                    help='C2: training steps captured into one HIP graph (0 = the largest divisor of --steps up to 25)')
    ap.add_argument('--no-graph', action='store_true',
                    help='c2: launch every kernel of the step from Python instead of replaying the captured HIP graph')
    ap.add_argument('--no-compare', action='store_true',
                    help='c2: skip the extra leg that times the frame-rate order of operations (frame_rate_order in the JSON line)')
    ap.add_argument('--rehearse-exchange', action='store_true',
                    help='one rank only: run the MULTI-rank code path (gradient buckets, all-reduce through a world-size-1 RCCL group, '
                         'no deferred slabs) - what a rank of an N-GPU job executes per step, without the peers')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dry-run-ranks', action='store_true',
                    help='rendezvous only: every rank joins the process group, one all-reduce counts them, rank 0 prints '
                         '{"n_gpus": N, "dry_run": true} - checks the launcher path without touching a GPU')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.config == 'c2' else 30
    if args.warmup is None:
        args.warmup = 20 if args.config == 'c2' else 5
    return args


def time_kernel(fn, iters=10, warm=2, graph=False):
    """Average duration (ms) of `fn` (which launches on torch's current stream), measured with HIP events on that stream around
    replays of a HIP graph holding `iters` calls: the kernels of the phone-rate step take 20-30 us, less than the host needs to
    issue one of them through Python, so timed launch by launch the result would be the host's rate (graph=True: the F0 step's kernels;
    falls back to timing the launches themselves where a call cannot be captured)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    entry = torch.cuda.current_stream()
    try:
        if not graph:
            raise RuntimeError('launch by launch')
        captured = torch.cuda.CUDAGraph()
        with torch.cuda.graph(captured):
            for _ in range(iters):
                fn()
        captured.replay()
        torch.cuda.synchronize()
        start.record()
        for _ in range(3):
            captured.replay()
        end.record()
        end.synchronize()
        return start.elapsed_time(end) / (3 * iters)
    except Exception:
        torch.cuda.set_stream(entry)
        torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / iters


def l2tail_flops(rows):
    """Products of mg_f0_l2tail_bf16 per row: layer 2 forward (512 x 128), layer 3 forward, dH2 and dW3 (3 x 128 x 32), layer 4 forward
    and dW4 (2 x 32)."""
    return 2.0 * rows * (512 * 128 + 3 * 128 * 32 + 2 * 32)


def roofline_f0(features, model, precision):
    """Time the GEMM kernels of the step in isolation (HIP events on the launch stream) and report the dominant one
    against the MFMA roof.  FLOPs are the algorithmic 2*M*K*N of the products the launch replaces."""
    lab = features['normalised_lab']
    b, p, k = lab.shape
    t = features['normalised_lf0'].shape[1]
    m = b * t
    dur2d = features['dur'].reshape(b, -1).contiguous()
    _, rows = ops.upsample_index(dur2d, t)
    rows = rows.view(-1)
    lins = [mod for mod in model.layers if isinstance(mod, torch.nn.Linear)]
    w1, b1, w2, b2 = lins[0].weight.detach(), lins[0].bias.detach(), lins[1].weight.detach(), lins[1].bias.detach()
    n1, n2 = w1.shape[0], w2.shape[0]
    kernels = []
    slab_bufs = {}

    def wgrad_alone(key, dy, a, r, m_, n_, k_):
        """The weight-gradient GEMM kernel without its slab-reduce launch (the entry point that leaves the slabs to the caller),
        into one persistent slab buffer per kernel; shapes without slabs run the complete entry point."""
        if ops.wgrad_wide_ok(m_, n_, k_, a.shape[1], dy.shape[1]):
            slab_bufs[key] = ops.linear_wgrad_slabs_bf16(dy, a, r, m_, n_, k_, slab=slab_bufs.get(key))[0]
        else:
            ops.linear_wgrad_bf16(dy, a, r, m_, n_, k_)

    def fused_alone(*args):
        slab_bufs['fused'] = ops.linear_bwd_fused_slabs_bf16(*args, slab=slab_bufs.get('fused'))[0]

    bound = {}
    algo_bytes = {}      # least bytes a launch of the kernel must move: operands once + result once (bf16 operands, fp32 gradients)
    if precision == 'bf16' and ops.phone_rate_table_ok(b * p, m, n1, n2, ops.ACT_SIGMOID):
        # the step as LinearStackMSEFn runs it (whole stack at phone rate): every layer on the B*P phone rows + the extra zero rows,
        # the masked MSE reduced per phone, the backward chain on the same rows
        extra = ops.PHONE_RATE_EXTRA
        r_tab = b * p + extra
        seg, rows_p = ops.segment_bounds(rows, b * p, pad_row=b * p)
        tab = ops.cast_pad_bf16(lab.view(b * p, k), extra_rows=extra)
        (w1b, w2b), (_, w2t) = ops.cast_params_bf16([w1, w2], want_t=(1,))
        h_tab = ops.linear_fwd_bf16(tab, None, r_tab, k, w1b, b1, n1, ops.ACT_SIGMOID)
        dz2 = (torch.randn(r_tab, ops.pad_ld(n2), device=lab.device) * 0.01).to(torch.bfloat16)
        dz1 = ops.linear_dgrad_bf16(dz2, r_tab, n2, w2t, n1, h_tab)
        target = features['normalised_lf0'].reshape(-1)
        seq_len = features['n_frames']
        if ops.phone_front_ok(b, p, t, extra):
            kernels.append(('phone_front_gemm_kernel<1>: layer-1 forward at phone rate (%d rows, 600->512 + bias + sigmoid) with the frame map and '
                            'the per-phone loss statistics riding on the CUs the GEMM leaves idle' % r_tab, 2.0 * r_tab * k * n1,
                            lambda: ops.phone_front(dur2d, target, seq_len, t, extra, linear=(tab, k, w1b, b1, n1, ops.ACT_SIGMOID))))
        else:
            kernels.append(('gemm_nt_persist_kernel<256>: layer-1 forward at phone rate (%d rows, 600->512 + bias + sigmoid)' % r_tab,
                            2.0 * r_tab * k * n1, lambda: ops.linear_fwd_bf16(tab, None, r_tab, k, w1b, b1, n1, ops.ACT_SIGMOID)))
            kernels.append(('phone_target_stats_kernel: per-phone weight / mean target / constant of the masked MSE (reads the M targets)', 0.0,
                            lambda: ops.phone_target_stats(target, rows_p, seg, seq_len, b, t, b * p, extra)))
        w3, b3, w4, b4 = lins[2].weight.detach(), lins[2].bias.detach(), lins[3].weight.detach(), lins[3].bias.detach()
        if ops.l2tail_ok(w2, w3, w4, ops.ACT_SIGMOID):
            ybar, weight, _ = ops.phone_target_stats(target, rows_p, seg, seq_len, b, t, b * p, extra)
            tail_grads = torch.empty(w3.numel() + b3.numel() + w4.numel() + b4.numel() + 1, device=lab.device)
            kernels.append(('f0_l2tail_kernel<0>: layer 2 + layers 3-4 + masked MSE + their backward at phone rate, one pass over H1', l2tail_flops(r_tab),
                            lambda: ops.f0_l2tail_rows(h_tab, w2b, b2, w3, b3, w4, b4, ybar, weight, tail_grads)))
            algo_bytes['f0_l2tail_kernel<0>'] = 2.0 * (r_tab * n1 + n2 * n1 + r_tab * n2) + 12.0 * r_tab
        else:
            kernels.append(('gemm_nt_persist_kernel<128>: layer-2 forward at phone rate (512->128 + bias + sigmoid)',
                            2.0 * r_tab * n1 * n2, lambda: ops.linear_fwd_bf16(h_tab, None, r_tab, n1, w2b, b2, n2, ops.ACT_SIGMOID)))
        bound['phone_target_stats_kernel'] = ('hbm', m * 4.0 * 2 + r_tab * 8.0)
        ldk = tab.shape[1]
        algo_bytes.update({'phone_front_gemm_kernel<1>': 2.0 * (r_tab * ldk + n1 * ldk + r_tab * n1) + m * 12.0 + b * p * 8.0 + r_tab * 16.0,
                           'wgrad_dgrad_pair_kernel<4>': 2.0 * (r_tab * n2 + n1 * n2 + 2 * r_tab * n1) + 4.0 * n2 * n1,
                           'gemm_nt_persist_kernel<256>': 2.0 * (r_tab * ldk + n1 * ldk + r_tab * n1),
                           'gemm_nt_persist_kernel<128>': 2.0 * (r_tab * n1 + n2 * n1 + r_tab * n2),
                           'wgrad_big_kernel<8>': 2.0 * (r_tab * n2 + r_tab * n1) + 4.0 * n2 * n1,
                           'gemm_nt_big_kernel<256>': 2.0 * (r_tab * n2 + n1 * n2 + 2 * r_tab * n1),
                           'wgrad_big_kernel<5>': 2.0 * (r_tab * n1 + r_tab * ldk) + 4.0 * n1 * k})
        if ops.wgrad_slabs_ok(r_tab, n2, n1, h_tab.shape[1], dz2.shape[1]):
            kernels.append(('wgrad_dgrad_pair_kernel<4>: layer-2 wgrad slabs (dZ2^T table) and layer-2 dgrad + sigmoid-grad at phone rate in '
                            'one grid', 4.0 * r_tab * n1 * n2, lambda: ops.linear_wgrad_dgrad_bf16(dz2, h_tab, r_tab, n2, n1, w2t)))
        else:
            kernels.append(('wgrad_big_kernel<8>: layer-2 wgrad at phone rate (dZ2^T table)', 2.0 * r_tab * n1 * n2,
                            lambda: wgrad_alone('w2p', dz2, h_tab, None, r_tab, n2, n1)))
            kernels.append(('gemm_nt_big_kernel<256>: layer-2 dgrad + sigmoid-grad at phone rate', 2.0 * r_tab * n1 * n2,
                            lambda: ops.linear_dgrad_bf16(dz2, r_tab, n2, w2t, n1, h_tab)))
        kernels.append(('wgrad_big_kernel<5>: layer-1 wgrad at phone rate (dZ1^T lab; 128 x 320 tiles, 32 split-M slabs)', 2.0 * r_tab * k * n1,
                        lambda: wgrad_alone('w1p', dz1, tab, None, r_tab, n1, k)))
        peak = MFMA_BF16_PEAK_TFLOPS
    elif precision == 'bf16':
        tab = ops.cast_pad_bf16(lab.view(b * p, k))
        (w1b, w2b), (_, w2t) = ops.cast_params_bf16([w1, w2], want_t=(1,))
        h1 = ops.linear_fwd_bf16(tab, rows, m, k, w1b, b1, n1, ops.ACT_SIGMOID)
        dz2 = (torch.randn(m, ops.pad_ld(n2), device=lab.device) * 0.01).to(torch.bfloat16)
        kernels.append(('gemm_nt_runs_kernel<1>: layer-1 forward (gathered operand staged by runs, 600->512 + bias + sigmoid)',
                        2.0 * m * k * n1, lambda: ops.linear_fwd_bf16(tab, rows, m, k, w1b, b1, n1, ops.ACT_SIGMOID, rows_runs=True)))
        w3, b3, w4, b4 = lins[2].weight.detach(), lins[2].bias.detach(), lins[3].weight.detach(), lins[3].bias.detach()
        if ops.l2tail_ok(w2, w3, w4, ops.ACT_SIGMOID):
            tail_grads = torch.empty(w3.numel() + b3.numel() + w4.numel() + b4.numel() + 1, device=lab.device)
            tgt, n_fr = features['normalised_lf0'].reshape(-1), features['n_frames']
            kernels.append(('f0_l2tail_kernel<0>: layer 2 + layers 3-4 + masked MSE + their backward, one pass over H1', l2tail_flops(m),
                            lambda: ops.f0_l2tail(h1, w2b, b2, w3, b3, w4, b4, tgt, n_fr, b, t, tail_grads)))
        else:
            kernels.append(('gemm_nt_persist_kernel<128>: layer-2 forward (512->128 + bias + sigmoid)', 2.0 * m * n1 * n2,
                            lambda: ops.linear_fwd_bf16(h1, None, m, n1, w2b, b2, n2, ops.ACT_SIGMOID)))
        if ops.can_fuse_bwd(m, n2, n1, k, tab.shape[1]):
            kernels.append(('wgrad_fused64_kernel<3>: layer-2 dgrad + sigmoid-grad + layer-1 wgrad (gather-fused), dZ1 on chip',
                            2.0 * m * n1 * n2 + 2.0 * m * k * n1,
                            lambda: fused_alone(dz2, w2t, h1, tab, rows, m, n1, k)))
        else:
            dz1 = (torch.randn(m, n1, device=lab.device) * 0.01).to(torch.bfloat16)
            kernels.append(('wgrad_big_kernel<10>: layer-1 wgrad (gather-fused)', 2.0 * m * k * n1,
                            lambda: wgrad_alone('w1f', dz1, tab, rows, m, n1, k)))
        kernels.append(('wgrad_big_kernel<8>: layer-2 wgrad (dZ2^T H1)', 2.0 * m * n1 * n2,
                        lambda: wgrad_alone('w2f', dz2, h1, None, m, n2, n1)))
        n_tab, ldk = tab.shape
        algo_bytes.update({'gemm_nt_runs_kernel<1>': 2.0 * (n_tab * ldk + n1 * ldk + m * n1) + 4.0 * m,
                           'f0_l2tail_kernel<0>': 2.0 * (m * n1 + n2 * n1 + m * n2) + 12.0 * m,
                           'gemm_nt_persist_kernel<128>': 2.0 * (m * n1 + n2 * n1 + m * n2),
                           'wgrad_fused64_kernel<3>': 2.0 * (m * n2 + m * n1 + n_tab * ldk) + 4.0 * m + 4.0 * n1 * k,
                           'wgrad_big_kernel<10>': 2.0 * (m * n1 + n_tab * ldk) + 4.0 * m + 4.0 * n1 * k,
                           'wgrad_big_kernel<8>': 2.0 * (m * n2 + m * n1) + 4.0 * n2 * n1})
        peak = MFMA_BF16_PEAK_TFLOPS
    else:
        tab = lab.view(b * p, k)
        h1 = ops.linear_fwd_f32(tab, rows, m, w1, b1, ops.ACT_SIGMOID)
        dz1 = torch.randn(m, n1, device=lab.device)
        dz2 = torch.randn(m, n2, device=lab.device)
        kernels.append(('gemm_f32_kernel: layer-1 forward (gather-fused 600->512 + bias + sigmoid)', 2.0 * m * k * n1,
                        lambda: ops.linear_fwd_f32(tab, rows, m, w1, b1, ops.ACT_SIGMOID)))
        kernels.append(('wgrad_f32_kernel: layer-1 wgrad (gather-fused)', 2.0 * m * k * n1,
                        lambda: ops.linear_wgrad_f32(dz1, tab, rows, n1, k)))
        kernels.append(('gemm_f32_kernel: layer-2 dgrad (dZ2 W2 * H1(1-H1))', 2.0 * m * n1 * n2,
                        lambda: ops.linear_dgrad_f32(dz2, w2, h1)))
        peak = MFMA_F32_PEAK_TFLOPS
    measured = []
    for name, flops, fn in kernels:
        ms = time_kernel(fn, graph=True)
        measured.append({'kernel': name, 'ms': round(ms, 4), 'gflop': round(flops / 1e9, 1),
                         'tflops': round(flops / (ms * 1e-3) / 1e12, 2)})
    dom = max(measured, key=lambda r: r['ms'])
    short = dom['kernel'].split(':')[0]
    # HBM bytes per launch of the dominant kernel from the committed PMC passes of this round (scripts/gpu_profile.sh:
    # 2 x FETCH_SIZE + WRITE_SIZE on gfx950, MI355X_MICROARCH.md), and the bytes the launch has to move at the very least
    traffic = traffic_source = None
    for table_name in (('r5_hbm_traffic.json', 'r4_hbm_traffic.json', 'r3_hbm_traffic.json') if ops.PHONE_RATE else
                       ('r5fr_hbm_traffic.json', 'r4fr_hbm_traffic.json', 'r3fr_hbm_traffic.json')):
        try:
            table = json.load(open(os.path.join(REPO, 'profiles', table_name)))
            if table.get(short) is not None:
                traffic = table.get(short)
                traffic_source = ('profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on the committed kernels '
                                  '(2 x FETCH_SIZE + WRITE_SIZE per launch), not re-measured in this run' % table_name)
                break
        except (OSError, ValueError):
            pass
    algorithmic_bytes = algo_bytes.get(short)
    out = {'kernel': dom['kernel'], 'ms_per_launch': dom['ms'], 'traffic': traffic, 'traffic_source': traffic_source,
           'algorithmic_bytes': algorithmic_bytes}
    # which roof bounds the launch is read off the counters: a kernel that moves more than 0.6 of the 8 TB/s spec is memory bound
    # whatever its FLOPs are (the first round labelled such a kernel "mfma")
    moved = traffic if traffic is not None else algorithmic_bytes
    hbm_gbs = moved / (dom['ms'] * 1e-3) / 1e9 if moved else 0.0
    if short in bound or hbm_gbs > 0.6 * HBM_PEAK_GBS:
        nbytes = bound[short][1] if short in bound else algorithmic_bytes
        gbs = nbytes / (dom['ms'] * 1e-3) / 1e9
        out.update({'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
                    'measured_traffic_gbs': round(hbm_gbs, 1), 'mfma_tflops': dom['tflops']})
    else:
        out.update({'bound': 'mfma', 'achieved': dom['tflops'], 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(dom['tflops'] / peak, 4),
                    'measured_traffic_gbs': round(hbm_gbs, 1)})
    # Both fractions, whichever roof `bound` names, and what the kernel's own measurements say limits it: at
    # phone-rate row counts the wide tile programs wait for LDS-DMA line requests - a CU takes in about one 128-byte line per 10 cycles
    # from beyond its XCD's L2 (4 from it) - not for the matrix pipe and not for HBM bandwidth (profiles/r4_notes_falsified_kernel_ideas.txt;
    # round 5: the pair-plane weight gradient went from 63 to 45 us when its line requests per row went from 21 to 14, same FLOPs, same bytes)
    out['frac_mfma'] = round(dom['tflops'] / peak, 4)
    if traffic is not None:
        out['frac_hbm_measured'] = round(traffic / (dom['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if algorithmic_bytes:
        out['frac_hbm_algorithmic'] = round(algorithmic_bytes / (dom['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if ops.PHONE_RATE and precision == 'bf16':
        out['bound_detail'] = ('neither roof: LDS-DMA line-request intake (about one 128-byte line per 10 cycles and CU from beyond the XCD\'s L2, '
                               '~7 TB/s over the chip) - `bound` names the roof the contraction would meet first, `frac` the fraction of it')
    out['kernels'] = measured
    return out


def roofline_gru(features, model, precision, target):
    """Time the two recurrence launches of the GRU layer in isolation.  T dependent steps of a [B,H] x [H,3H] product: the
    roof that the arithmetic sees is MFMA, but the launch is bound by the per-step hand-off latency between workgroups
    (DESIGN.md section 5); the fraction of the MFMA roof is reported as it is."""
    b, t = features[target].shape[:2]
    gru = [mod for mod in model.modules() if isinstance(mod, torch.nn.GRU)][0]
    hid = gru.hidden_size
    dev = features[target].device
    seq_len = features['n_frames'].to(dev).view(-1)
    steps = int(seq_len.max().item())
    w_hh, b_hh = gru.weight_hh_l0.detach(), gru.bias_hh_l0.detach()
    xproj = torch.randn(b, t, 3 * hid, device=dev)
    g_out = torch.randn(b, t, hid, device=dev) * 0.01
    flops = 2.0 * float(seq_len.sum().item()) * 3 * hid * hid
    if precision == 'bf16' and ops.gru_bf16_ok(hid):
        tag = 'persist' if ops.gru_persist_ok(b, t, hid) else 'step_bf16'
        out, hstate, saved, _ = ops.gru_fwd_bf16(xproj, w_hh, b_hh, seq_len, None, b, t, hid)
        runs = [('gru_fwd_%s_kernel: GRU forward recurrence, %d dependent steps' % (tag, steps),
                 lambda: ops.gru_fwd_bf16(xproj, w_hh, b_hh, seq_len, None, b, t, hid)),
                ('gru_bwd_%s_kernel: GRU backward recurrence, %d dependent steps' % (tag, steps),
                 lambda: ops.gru_bwd_bf16(g_out, None, hstate, saved, w_hh, seq_len, b, t, hid))]
        peak = MFMA_BF16_PEAK_TFLOPS
    else:
        out, hstate, saved = ops.gru_fwd(xproj, w_hh, b_hh, seq_len, None, b, t, hid)
        runs = [('gru_fwd_step_kernel x %d launches' % steps, lambda: ops.gru_fwd(xproj, w_hh, b_hh, seq_len, None, b, t, hid)),
                ('gru_bwd_step_kernel x %d launches' % steps,
                 lambda: ops.gru_bwd(g_out, None, hstate, saved, w_hh, seq_len, b, t, hid))]
        peak = MFMA_F32_PEAK_TFLOPS
    measured = []
    for name, fn in runs:
        ms = time_kernel(fn, iters=5, warm=2)
        measured.append({'kernel': name, 'ms': round(ms, 4), 'gflop': round(flops / 1e9, 1),
                         'tflops': round(flops / (ms * 1e-3) / 1e12, 2), 'us_per_step': round(ms * 1e3 / steps, 3)})
    ops.check_persistent_status()
    dom = max(measured, key=lambda r: r['ms'])
    return {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': dom['tflops'], 'peak': peak, 'unit': 'TFLOP/s',
            'frac': round(dom['tflops'] / peak, 4), 'traffic': None, 'ms_per_launch': dom['ms'],
            'note': 'latency-bound chain: %.2f us per dependent step (hand-off between workgroups), not an MFMA-rate limit'
                    % dom['us_per_step'], 'kernels': measured}


def roofline_lstm(features, model, precision):
    """Time the recurrence launches of the LSTM stack in isolation: the whole-stack forward wavefront (one launch) and one
    layer's backward recurrence.  Same remark as roofline_gru: dependent chains, the MFMA fraction is reported as it is."""
    wrappers = [mod for mod in model.modules() if isinstance(mod, torch.nn.LSTM)]
    n_layers, hid = len(wrappers), wrappers[0].hidden_size
    b = features['n_frames'].shape[0]
    dev = features['n_frames'].device
    seq_len = features['n_frames'].to(dev).view(-1)
    t = int(seq_len.max().item())
    w_ih = [m.weight_ih_l0.detach() for m in wrappers]
    w_hh = [m.weight_hh_l0.detach() for m in wrappers]
    b_ih = [m.bias_ih_l0.detach() for m in wrappers]
    b_hh = [m.bias_hh_l0.detach() for m in wrappers]
    xproj = torch.randn(b, t, 4 * hid, device=dev)
    g_out = torch.randn(b, t, hid, device=dev) * 0.01
    frames = float(seq_len.sum().item())
    measured = []
    if precision == 'bf16' and ops.lstm_pstack_ok(b, t, hid, n_layers):
        out, hstate, cstate, saved, hstate_bf = ops.lstm_pstack_fwd(xproj, w_ih, w_hh, b_ih, b_hh, seq_len, None, None, b, t, hid)
        flops_f = 2.0 * frames * 4 * hid * hid * (2 * n_layers - 1)       # W_hh of every layer + W_ih of the layers above the first
        ms = time_kernel(lambda: ops.lstm_pstack_fwd(xproj, w_ih, w_hh, b_ih, b_hh, seq_len, None, None, b, t, hid), iters=3, warm=1)
        measured.append({'kernel': 'lstm_stack_fwd_persist_kernel: %d layers, %d wavefront steps' % (n_layers, t + n_layers - 1),
                         'ms': round(ms, 4), 'gflop': round(flops_f / 1e9, 1), 'tflops': round(flops_f / (ms * 1e-3) / 1e12, 2),
                         'us_per_step': round(ms * 1e3 / (t + n_layers - 1), 3)})
        flops_b = 2.0 * frames * 4 * hid * hid
        ms = time_kernel(lambda: ops.lstm_bwd_bf16(g_out, None, None, cstate[0], saved[0], w_hh[0], seq_len, b, t, hid, want_f32=False),
                         iters=3, warm=1)
        measured.append({'kernel': 'lstm_bwd_persist_kernel: one layer, %d dependent steps (x %d layers per training step)' % (t, n_layers),
                         'ms': round(ms, 4), 'gflop': round(flops_b / 1e9, 1), 'tflops': round(flops_b / (ms * 1e-3) / 1e12, 2),
                         'us_per_step': round(ms * 1e3 / t, 3)})
        ops.check_persistent_status()
    if not measured:
        return None
    dom = max(measured, key=lambda r: r['ms'])
    peak = MFMA_BF16_PEAK_TFLOPS
    return {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': dom['tflops'], 'peak': peak, 'unit': 'TFLOP/s',
            'frac': round(dom['tflops'] / peak, 4), 'traffic': None, 'ms_per_launch': dom['ms'],
            'note': 'latency / L2-traffic bound chain: %.2f us per dependent step, not an MFMA-rate limit' % dom['us_per_step'],
            'kernels': measured}


def host_cores():
    """CPU threads this process may really use: affinity, capped by the cgroup CPU quota (the GPU box gives a 1-GPU job
    a share of the host, not all of its cores) and by 32 (the 64 x 1000-frame sample does not scale further)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            text = open(path).read().split()
            if path.endswith('cpu.max'):
                if text[0] != 'max':
                    n = min(n, max(1, int(int(text[0]) / int(text[1]))))
            else:
                quota = int(text[0])
                period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 32))


def cpu_baseline_f0(per_gpu, frames):
    """Oracle torch-CPU restatement of the reference step (oracle/ref_torch.py: the same torch ops in the reference's order, proven
    equal to the imported reference by the golden vectors) on the FULL C2 batch, on the host cores of this box: all threads the
    process may use, then one thread.  Bounded: the N-thread leg times 5 steps after 2 warm-ups (about 1 s
    per step), the 1-thread leg as many steps as fit in 30 s after one warm-up (at least one)."""
    from oracle import ref_torch
    feats = ref_torch.to_torch(synthetic.make_batch(per_gpu, frames))
    model = ref_torch.load_state(ref_torch.F0Model(), synthetic.f0_model_state())
    opt = torch.optim.Adam(model.parameters(), lr=0.01)

    def step():
        opt.zero_grad()
        loss, _ = model(feats)
        loss.backward()
        opt.step()

    def timed(n_threads, warm, max_steps, budget_s):
        torch.set_num_threads(n_threads)
        for _ in range(warm):
            step()
        n_timed, t0 = 0, time.perf_counter()
        while n_timed < max_steps and (n_timed == 0 or (time.perf_counter() - t0) < budget_s):
            step()
            n_timed += 1
        return n_timed, (time.perf_counter() - t0) / n_timed

    n_threads = host_cores()
    n_multi, dt_multi = timed(n_threads, 2, 5, 40.0)
    n_single, dt_single = timed(1, 1, 5, 30.0)
    torch.set_num_threads(n_threads)
    total = per_gpu * frames
    return {'value': round(total / dt_multi, 1), 'unit': 'frames/s', 'cores': n_threads, 'kind': 'port',
            'single_thread': {'value': round(total / dt_single, 1), 'unit': 'frames/s', 'cores': 1, 'steps': n_single,
                              's_per_step': round(dt_single, 3)},
            's_per_step': round(dt_multi, 3),
            'sample': '%d timed steps (2 warm-up) on %d threads and %d timed step(s) (1 warm-up) on 1 thread of the full %d x %d-frame C2 '
                      'batch, torch-CPU fp32 restatement of the reference step (oracle/ref_torch.py)'
                      % (n_multi, n_threads, n_single, per_gpu, frames)}


def loss_curve_deviation(dev):
    """SURVEY.md section 8d "Reporting": maximum relative deviation of this build's loss curve from the REFERENCE's own curve (golden G6:
    20 Adam steps of the README F0Model at config C1, produced by the imported reference, tests/golden/make_golden.py), in the
    benchmarked bf16 mode and in fp32 parity mode.  The north star's 1e-4 is a claim about fp32 mode; bf16 is stated next to it."""
    path = os.path.join(REPO, 'tests', 'golden', 'g6_f0_model.npz')
    if not os.path.exists(path):
        return None
    g = dict(np.load(path, allow_pickle=False))
    want = np.asarray(g['loss_curve'], dtype=np.float64)
    # the batches G6 was generated on: four C1 batches (8 x 200 frames) cycled, lr 0.01 (tests/golden/make_golden.py, tests/test_gpu_parity.py)
    batches = [data.to_device(synthetic.make_batch(8, 200, seed=synthetic.REFERENCE_SEED + 100 * i), dev) for i in range(4)]
    out = {'reference': 'tests/golden/g6_f0_model.npz: 20 Adam steps (lr 0.01) of the README F0Model at config C1, computed by the imported '
                        'reference on CPU in fp32', 'what': 'max over the 20 steps of |loss - reference loss| / reference loss'}
    for precision in ('bf16', 'bf16x3', 'fp32'):
        model = models.F0Model(precision=precision).to(dev)
        own = model.state_dict()
        for key, value in synthetic.f0_model_state().items():
            own[key].copy_(torch.from_numpy(value))
        opt = optim.Adam(model.parameters(), lr=0.01)
        curve = []
        for i in range(len(want)):
            opt.zero_grad()
            loss, _ = model(batches[i % len(batches)])
            F_hip.backward(loss)
            opt.step()
            curve.append(loss.item())
        out[precision] = float('%.3g' % float(np.max(np.abs(np.asarray(curve) - want) / np.abs(want))))
    return out


def timed_leg(step, steps, warmup):
    """ms per step of an eager loop.  Python's cyclic collector is run first and kept off while the clock runs: the legs before this
    one leave captured HIP graphs and their private memory pools in reference cycles, and a collection that happens to fall into the
    timed steps destroys them there - device frees that stall the host for tens of milliseconds with every kernel at its usual
    duration (seen as 17 -> 25-43 ms on the LSTM leg, at random; profiles/r3_notes_power.txt, last section)."""
    import gc
    gc.collect()                                   # in front of the warm-up: no idle device between the warm-up steps and the clock
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        # two blocks of `steps`; the MEAN is reported, both blocks are printed, and blocks that differ by more than 20 % are flagged
        # (`anomaly`): a slow block is an event to explain, not to hide behind a minimum
        blocks = [_timed_steps(step, steps) for _ in range(2)]
        LAST_BLOCKS[:] = [round(b, 4) for b in blocks]
        return sum(blocks) / len(blocks)
    finally:
        if gc_was_on:
            gc.enable()


LAST_BLOCKS = []       # ms per step of the two blocks the last timed_leg ran


def blocks_report(steps, warmup):
    """The fields every eager leg carries about its two timed blocks."""
    out = {'blocks_ms_per_step': list(LAST_BLOCKS),
           'timing': 'mean of two blocks of %d steps after %d warm-up steps' % (steps, warmup)}
    if LAST_BLOCKS and max(LAST_BLOCKS) > 1.2 * min(LAST_BLOCKS):
        out['anomaly'] = 'the two timed blocks differ by %.0f %% (%s ms per step)' % (
            100.0 * (max(LAST_BLOCKS) / min(LAST_BLOCKS) - 1.0), ' / '.join('%.4f' % b for b in LAST_BLOCKS))
    return out


def _timed_steps(step, steps):
    debug = os.environ.get('MG_BENCH_DEBUG') == '1'
    if debug:
        st0 = torch.cuda.memory_stats()
        each = []
        for _ in range(steps):                       # (debug only: every step on its own, synchronised)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            step()
            torch.cuda.synchronize()
            each.append('%.2f' % ((time.perf_counter() - ta) * 1e3))
        sys.stderr.write('timed_leg: single steps (ms) %s\n' % ' '.join(each))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    if debug:
        st1 = torch.cuda.memory_stats()
        sys.stderr.write('timed_leg: host issue %.2f ms/step, total %.2f ms/step, device mallocs %d frees %d retries %d reserved %.1f GB\n' % (
            t_host / steps * 1e3, (time.perf_counter() - t0) / steps * 1e3, st1['num_device_alloc'] - st0['num_device_alloc'],
            st1['num_device_free'] - st0['num_device_free'], st1['num_alloc_retries'] - st0['num_alloc_retries'],
            st1['reserved_bytes.all.current'] / 1e9))
    return (time.perf_counter() - t0) / steps * 1e3


def sustained_mfma_tflops(dev, launches=4, trips=20000):
    """The bf16 MFMA rate this chip sustains on RANDOM operands when nothing but its matrix pipe works (mg_calib_mfma_bf16: register
    operands, two waves per SIMD on every CU, no LDS or memory traffic in the loop) - about 20 ms of launches timed with HIP events
    on the launch stream, right after the frame-rate leg so that the chip is in the state the step was timed in.  The guide's
    2.5 PFLOP/s dense peak assumes 2.4 GHz; under its power limit the chip holds less on real data (MI355X_MICROARCH.md, DVFS
    give-back), and the step's fraction of THAT rate is what separates "the kernels leave matrix cycles unused" from "the chip gives
    fewer cycles" (VERDICT round 3, item 1c)."""
    import ctypes
    lib = _lib.load()
    n_wg = torch.cuda.get_device_properties(dev).multi_processor_count
    operands = (torch.rand(4096 * 64, device=dev) * 2.0 - 1.0).to(torch.bfloat16)
    sink = torch.empty(n_wg * 512, dtype=torch.float32, device=dev)
    flop = ctypes.c_double(0.0)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch():
        _lib.check(lib.mg_calib_mfma_bf16(ctypes.c_void_p(operands.data_ptr()), ctypes.c_void_p(sink.data_ptr()), n_wg, trips,
                                          ctypes.byref(flop), stream), 'mg_calib_mfma_bf16')
    launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        launch()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1)
    return launches * flop.value / (ms * 1e-3) / 1e12, ms, n_wg


def roofline_f0_x3(features, model):
    """The five launches of the fused 'bf16x3' phone-rate step (functional.F0StackX3Fn) timed in isolation with HIP events through a
    graph, the dominant one against the bf16 MFMA roof.  FLOPs = the bf16 products the matrix cores execute: three per product of the
    step (the exact-fp32 tail's products counted once); algorithmic bytes = pair-plane operands once + results once."""
    lab = features['normalised_lab']
    b, p, k = lab.shape
    target = features['normalised_lf0'].reshape(-1)
    t = features['normalised_lf0'].shape[1]
    seq_len = features['n_frames']
    extra = ops.PHONE_RATE_EXTRA
    r_tab = b * p + extra
    lins = [mod for mod in model.layers if isinstance(mod, torch.nn.Linear)]
    w1, b1, w2, b2, w3, b3, w4, b4 = [t_.detach() for lin in lins for t_ in (lin.weight, lin.bias)]
    n1, n2 = w1.shape[0], w2.shape[0]
    tab = features.get('normalised_lab' + data.X3_TABLE_SUFFIX)
    if tab is None:
        tab = ops.split_pair(lab.view(b * p, k), extra_rows=extra)
    w1p, w2p, w2tp = ops.split_pair(w1), ops.split_pair(w2), ops.split_pair(w2, transpose=True)
    dur2d = features['dur'].reshape(b, -1).contiguous()
    front = (dur2d, target, seq_len, t, extra)
    res = ops.phone_front_x3(front, (tab, k, w1p, b1, n1, ops.ACT_SIGMOID))
    ybar, weight, h1 = res[3], res[4], res[6]
    _, dz2, _, _, _ = ops.f0_l2tail_x3(h1, w2p, b2, w3, b3, w4, b4, ybar, weight)
    dz2 = dz2.clone()
    bufs = {}

    def pair():
        bufs['s2'], _, _, dz1, bufs['cs'], _ = ops.linear_wgrad_dgrad_x3(dz2, h1, r_tab, n2, n1, w2tp, slab=bufs.get('s2'), colsum=bufs.get('cs'))
        return dz1
    dz1 = pair().clone()

    def wgrad1():
        bufs['s1'], _, _ = ops.linear_wgrad_slabs_x3(dz1, tab, r_tab, n1, k, slab=bufs.get('s1'))
    ldk = tab.shape[1] // 2
    kernels = [
        ('phone_front_gemm_kernel<1,192,2>: layer-1 forward on pair planes (%d rows, 600->512 + bias + sigmoid, output split) with the frame map and '
         'the per-phone loss statistics riding' % r_tab, 3 * 2.0 * r_tab * k * n1,
         2.0 * (2 * r_tab * ldk + 2 * n1 * ldk + 2 * r_tab * n1) + b * t * 12.0,
         lambda: ops.phone_front_x3(front, (tab, k, w1p, b1, n1, ops.ACT_SIGMOID))),
        ('f0_l2tail_x3_kernel: layer 2 on pair planes (Z2 on chip) + the exact-fp32 tail + masked MSE + their backward', 3 * 2.0 * r_tab * n1 * n2 + 2.0 * r_tab * (3 * 128 * 32 + 64),
         2.0 * (2 * r_tab * n1 + 2 * n2 * n1 + 2 * r_tab * n2) + 12.0 * r_tab,
         lambda: ops.f0_l2tail_x3(h1, w2p, b2, w3, b3, w4, b4, ybar, weight)),
        ('wgrad_dgrad_pair_kernel<4,256,1>: layer-2 wgrad slabs and layer-2 dgrad + sigmoid-grad (output split, bias column sums) in one grid',
         3 * 4.0 * r_tab * n1 * n2, 2.0 * (2 * r_tab * n2 + 2 * n1 * n2 + 4 * r_tab * n1) + 4.0 * n2 * n1, pair),
        ('wgrad_big_kernel<5,1,2>: layer-1 wgrad on pair planes (all four planes per stage, 32 split-M slabs)', 3 * 2.0 * r_tab * k * n1,
         2.0 * (2 * r_tab * n1 + 2 * r_tab * ldk) + 4.0 * n1 * k, wgrad1),
    ]
    measured = []
    for name, flops, nbytes, fn in kernels:
        ms = time_kernel(fn, graph=True)
        measured.append({'kernel': name, 'ms': round(ms, 4), 'gflop': round(flops / 1e9, 1), 'tflops': round(flops / (ms * 1e-3) / 1e12, 2),
                         'algorithmic_bytes': round(nbytes), 'algorithmic_gbs': round(nbytes / (ms * 1e-3) / 1e9, 1)})
    dom = max(measured, key=lambda r: r['ms'])
    traffic = traffic_source = None
    try:
        table = json.load(open(os.path.join(REPO, 'profiles', 'r5x3_hbm_traffic.json')))
        import re
        head = dom['kernel'].split(':')[0]
        m_ = re.match(r'([^<]+)<([^,>]+)', head)
        traffic = table.get('%s<%s>' % (m_.group(1), m_.group(2)) if m_ else head)          # summarise_profile.short's key
        traffic_source = 'profiles/r5x3_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (2 x FETCH_SIZE + WRITE_SIZE per launch), not re-measured in this run'
    except (OSError, ValueError):
        pass
    return {'kernel': dom['kernel'], 'ms_per_launch': dom['ms'], 'bound': 'mfma', 'achieved': dom['tflops'], 'peak': MFMA_BF16_PEAK_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(dom['tflops'] / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_source': traffic_source,
            'algorithmic_bytes': dom['algorithmic_bytes'],
            'frac_hbm_algorithmic': round(dom['algorithmic_gbs'] / HBM_PEAK_GBS, 4),
            'bound_detail': 'LDS-DMA line-request intake, as the bf16 step\'s kernels (roofline.bound_detail); tflops counts three bf16 products per product',
            'kernels': measured}


def bf16x3_legs(dev, features, state_dict, steps, frames_per_step):
    """Precision mode 'bf16x3' (split-bf16 operands: three bf16 MFMA products per fp32 product, fp32 activations - the mode that
    meets the reference's loss curve to 1e-4, `loss_curve_deviation.bf16x3`) timed on the C2 batch in the same run as the bf16
    headline, in BOTH orders of operations, graph replayed like the headline.  tflops counts the bf16
    products the matrix cores execute: three per product of the order's step."""
    from morgana_amd import graphs
    import gc
    out = {}
    for key, phone_rate in (('phone_rate', True), ('frame_rate_order', False)):
        try:
            model = models.F0Model(precision='bf16x3', phone_rate=phone_rate).to(dev)
            model.load_state_dict(state_dict)
            # the loader's half of the mode, as ExperimentBuilder.train_epoch asks for it (DeviceBatches.use_bf16_tables): the phone
            # table's [hi | lo] pair planes, made once per batch
            for name in model.bf16_table_features():
                data.add_bf16_table(features, name)
            opt = optim.Adam(model.parameters(), lr=0.01, fused_loop=True)
            n = max(k for k in range(1, min(10, max(steps // 2, 1)) + 1) if steps % k == 0)
            step = graphs.GraphedTrainStep(model, opt, features, steps_per_replay=n)
            gc.collect()
            gc.disable()
            for _ in range(max(400 // n, 2)):           # the headline's warm-up: ~400 steps in front of the clock (see main)
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps // n):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            gc.enable()
            out[key] = {'ms_per_step': round(ms, 4), 'value': round(frames_per_step / (ms * 1e-3), 1), 'unit': 'frames/s',
                        'steps': steps, 'launch': 'hip graph replay, %d steps per graph' % n}
            if not phone_rate:
                tf = 3.0 * F0_FLOPS_PER_FRAME * frames_per_step / (ms * 1e-3) / 1e12
                out[key].update({'tflops_bf16_products': round(tf, 1), 'frac_of_mfma_peak': round(tf / MFMA_BF16_PEAK_TFLOPS, 4)})
            else:
                import morgana_amd._lib as _l
                _l.CALL_LOG = []
                try:
                    probe = graphs.GraphedTrainStep(model, opt, features, warmup=0)       # a capture runs nothing: counts the step's entry points
                    out[key]['launches_per_step'] = len(_l.CALL_LOG)
                    out[key]['entry_points'] = list(_l.CALL_LOG)
                    del probe
                finally:
                    _l.CALL_LOG = None
                try:
                    out[key]['roofline'] = roofline_f0_x3(features, model)
                except Exception as exc:                  # noqa: BLE001
                    out[key]['roofline'] = {'error': str(exc).splitlines()[0][:200]}
            del step, model, opt
        except Exception as exc:                          # noqa: BLE001 - a leg that fails is reported, the headline stands
            out[key] = {'error': str(exc).splitlines()[0][:200]}
        finally:
            gc.enable()
    out['what'] = ("F0Model(precision='bf16x3'): x = hi + lo in bf16, x w ~= hi hi + hi lo + lo hi, fp32 accumulators.  phone_rate = the headline's "
                   "order of operations as the FUSED step on [hi | lo] pair planes (functional.F0StackX3Fn: five launches - layer 1 with the front "
                   "riding, layers 2-4 + loss + their backward with Z2 on chip and the 128 -> 32 -> 1 tail exact fp32, the pair grid, the "
                   "layer-1 weight gradient, the update, which keeps the weights' pairs current); frame_rate_order = every product on the B*T "
                   "frame rows through the generic row-wise path (three-plane operands split in passes of their own, csrc/split3.hip)")
    return out


def train_epoch_block(dev, state_dict, headline_ms, n_batches=200):
    """The loop the north star names - ``ExperimentBuilder.train_epoch`` (/root/reference/morgana/experiment_builder.py:464-494) - timed over
    DISTINCT batches: ``n_batches`` C2-shaped batches resident on the device, each with its own operand table
    (the loader's half of the precision mode, made per batch by data.add_bf16_table and timed beside the loop), through the product's
    ``train_epoch`` as eager launches and with ``use_graphs=True`` (graphs.GraphedStepCache: a list of resident batches is captured ten
    consecutive batches per graph, read where they lie - ``*_graphs``; ``*_graphs_load_per_batch`` = the streaming form: a batch signature
    is captured once and every later batch is copied into the graph's buffers, only what the step reads), for the bf16 headline and for 'bf16x3';
    and a ragged variant (600-1000 frames per utterance, padded lengths bucketed to four shapes) with the cache's hits counted.  Inputs
    are in HBM when the clock starts; one synchronisation per epoch, as the product's loop has (the reference reads the loss every batch)."""
    import gc
    from morgana_amd import experiment_builder
    out = {'what': 'ExperimentBuilder.train_epoch over %d distinct device-resident batches (lab drawn on the device per batch; durations and '
                   'targets from four host-made batches rolled along the batch axis): ms per step = epoch wall time / steps incl. the '
                   'epoch\'s one synchronisation; host_us_per_step = time the host took to issue the epoch' % n_batches}

    def resident(host_batches):
        batches = []
        for i in range(n_batches):
            h = host_batches[i % len(host_batches)]
            roll = (i // len(host_batches)) * 7
            feats = {k: (np.roll(v, roll, axis=0) if isinstance(v, np.ndarray) else v[roll % len(v):] + v[:roll % len(v)]) for k, v in h.items()}
            b = data.to_device(feats, dev)
            keep = (b['dur'] > 0).to(torch.float32)                                    # phones past an utterance's end stay zero rows
            b['normalised_lab'] = torch.rand(b['normalised_lab'].shape, device=dev) * keep
            batches.append(b)
        return batches

    def run(batches, precision, use_graphs, frames, graph_group=10):
        eb = experiment_builder.ExperimentBuilder(models.F0Model, model_kwargs={'precision': precision}, learning_rate=0.01, device=dev,
                                                  use_graphs=use_graphs, graph_group=graph_group)
        eb.model.load_state_dict(state_dict)
        tables = eb.model.bf16_table_features()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for b in batches:
            for stale in [k for k in b if k.endswith(data.BF16_TABLE_SUFFIX) or k.endswith(data.X3_TABLE_SUFFIX)]:
                del b[stale]
            for name in tables:
                data.add_bf16_table(b, name)
        e1.record()
        e1.synchronize()
        table_us = e0.elapsed_time(e1) * 1e3 / len(batches)
        optimizer = eb.make_optimizer()
        eb.train_epoch(batches, optimizer)                       # first epoch: buffers, operand shadows, the graphs of the shapes
        if use_graphs and graph_group > 1:
            eb.train_epoch(batches, optimizer)                   # resident groups: first epoch eager, second captures, third on replay
        gc.collect()
        gc.disable()
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eb.train_epoch(batches, optimizer)
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        ms = dt / len(batches) * 1e3
        rec = {'ms_per_step': round(ms, 4), 'value': round(frames / (ms * 1e-3), 1), 'unit': 'frames/s', 'steps': len(batches),
               'host_us_per_step': round(eb.last_epoch_stats['host_issue_s'] / eb.last_epoch_stats['steps'] * 1e6, 1),
               'loader_table_us_per_batch': round(table_us, 1),
               'vs_headline': round(ms / headline_ms, 3) if precision == 'bf16' else None}
        if use_graphs and eb._graph_cache is not None:
            rec['graph_cache'] = eb._graph_cache.stats()          # over all epochs run: eager = first batch of a signature / first epoch of a group
            rec['steps_per_graph'] = graph_group
        del eb, optimizer
        return rec

    try:
        fixed_host = [synthetic.make_batch(256, 1000, seed=synthetic.REFERENCE_SEED + 1000 + i) for i in range(4)]
        fixed = resident(fixed_host)
        frames = int(fixed_host[0]['n_frames'].sum())
        out['fixed_shape'] = {'batches': '%d x (256 utterances x 1000 frames)' % n_batches}
        for precision in ('bf16', 'bf16x3'):
            for use_graphs in (True, False):
                out['fixed_shape']['%s_%s' % (precision, 'graphs' if use_graphs else 'eager')] = run(fixed, precision, use_graphs, frames)
        # the streaming form (a loader that hands over a fresh batch every step): one graph per shape, every batch loaded into its buffers
        out['fixed_shape']['bf16_graphs_load_per_batch'] = run(fixed, 'bf16', True, frames, graph_group=1)
        del fixed
        ragged_host = []
        for i, t_pad in enumerate((700, 800, 900, 1000)):
            h = synthetic.make_batch(256, (600, t_pad), seed=synthetic.REFERENCE_SEED + 2000 + i)
            p_pad = int(round(t_pad / 12.5))
            for key in list(h):
                v = h[key]
                if isinstance(v, np.ndarray) and v.ndim == 3:           # pad to the bucket's shape (collate pads to the batch's longest item)
                    want = p_pad if key in ('dur', 'normalised_lab') else t_pad
                    if v.shape[1] < want:
                        h[key] = np.concatenate([v, np.zeros((v.shape[0], want - v.shape[1], v.shape[2]), v.dtype)], axis=1)
            ragged_host.append(h)
        ragged = resident(ragged_host)
        mean_frames = int(np.mean([int(h['n_frames'].sum()) for h in ragged_host]))
        out['ragged'] = {'batches': '%d x 256 utterances of 600-1000 frames, padded lengths 700 / 800 / 900 / 1000 in turn' % n_batches,
                         'frames_per_step_mean': mean_frames}
        for use_graphs in (True, False):
            out['ragged']['bf16_%s' % ('graphs' if use_graphs else 'eager')] = run(ragged, 'bf16', use_graphs, mean_frames)
        out['ragged']['bf16_graphs_load_per_batch'] = run(ragged, 'bf16', True, mean_frames, graph_group=1)
        del ragged
        out['streaming'] = streaming_epoch(dev, state_dict)
    except Exception as exc:                          # noqa: BLE001 - a block that fails is reported, the headline stands
        out['error'] = str(exc).splitlines()[0][:300] if str(exc) else type(exc).__name__
    gc.collect()
    torch.cuda.empty_cache()
    return out


def streaming_epoch(dev, state_dict, n_batches=6):
    """The PCIe-inclusive rate (never the headline's ``value``): ``data.DeviceBatches`` over raw host utterances - what the reference's
    DataLoader + ToDeviceWrapper hand over (/root/reference/morgana/data.py:50-55, :648-663) - feeding ``train_epoch(use_graphs=True)``:
    per batch the host packs 256 utterances (49 MB of float32 phone features), one pinned copy crosses PCIe and one kernel pads,
    normalises and writes the bf16 operand table; then the same batches kept on the device (``DeviceBatches.resident``)."""
    from morgana_amd import experiment_builder
    rng = np.random.RandomState(synthetic.REFERENCE_SEED + 77)
    lab_dim, n_ph, per_batch = 600, 80, 256
    norms = {'lab': data.MinMaxNormaliser('lab').set_params({'mmin': (rng.rand(lab_dim) * 0.1).astype(np.float32),
                                                             'mmax': (1.0 + rng.rand(lab_dim)).astype(np.float32)}, device=dev),
             'lf0': data.MeanVarianceNormaliser('lf0').set_params({'mean': np.array([5.0], np.float32),
                                                                   'std_dev': np.array([0.3], np.float32)}, device=dev)}
    utterances = []
    for i in range(per_batch * n_batches):
        dur = np.full((n_ph, 1), 12, np.int64)
        dur[::2] += 1                                                           # 12.5 frames per phone, 1000 frames per utterance
        n_fr = int(dur.sum())
        utterances.append({'name': 'utt%05d' % i, 'n_frames': n_fr, 'n_phones': n_ph, 'dur': dur,
                           'lab': rng.rand(n_ph, lab_dim).astype(np.float32),
                           'lf0': (5.0 + 0.3 * rng.randn(n_fr, 1)).astype(np.float32)})
    frames = per_batch * 1000
    eb = experiment_builder.ExperimentBuilder(models.F0Model, model_kwargs={'precision': 'bf16'}, learning_rate=0.01, device=dev, use_graphs=True)
    eb.model.load_state_dict(state_dict)
    optimizer = eb.make_optimizer()
    loader = data.DeviceBatches(utterances, per_batch, norms, dev)
    eb.train_epoch(loader, optimizer)                                           # graphs, buffers, pinned-memory pools
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eb.train_epoch(loader, optimizer)
    ms_stream = (time.perf_counter() - t0) / n_batches * 1e3
    t0 = time.perf_counter()
    for _ in loader:
        pass
    torch.cuda.synchronize()
    ms_loader = (time.perf_counter() - t0) / n_batches * 1e3
    resident = loader.resident()
    for _ in range(2):
        eb.train_epoch(resident, optimizer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eb.train_epoch(resident, optimizer)
    ms_res = (time.perf_counter() - t0) / n_batches * 1e3
    mb = per_batch * (n_ph * lab_dim + 1000) * 4 / 1e6
    del resident, loader
    return {'what': 'data.DeviceBatches over %d raw host utterances, %d per batch (C2 shape): host packing + pinned H2D copy of %.1f MB per batch + '
                    'mg_pad_normalise_bf16_f32 + the graphed step; then the same batches resident' % (len(utterances), per_batch, mb),
            'pcie_inclusive': {'ms_per_step': round(ms_stream, 3), 'value': round(frames / (ms_stream * 1e-3), 1), 'unit': 'frames/s',
                               'loader_alone_ms_per_batch': round(ms_loader, 3), 'host_mb_per_batch': round(mb, 1)},
            'resident_after_first_epoch': {'ms_per_step': round(ms_res, 4), 'value': round(frames / (ms_res * 1e-3), 1), 'unit': 'frames/s',
                                           'steps': n_batches, 'note': 'six-step epochs: the epoch\'s fixed cost (one sync, first launch) is in'}}


def c4_leg(dev, precision):
    """BASELINE config C4 (RNN_SPSS GRU-512, 600 -> 80, 64 x 1000 frames) timed in the same run as the headline, so that the recurrent
    model's number is a driver-timed one as well: a few eager steps (4 ms each; the recurrences are two persistent launches)."""
    feats_np = synthetic.make_batch(64, 1000, out_dim=80, target_name='mcep')
    model = models.RNNSPSS(precision=precision).to(dev)
    own = model.state_dict()
    for key, value in synthetic.rnn_spss_state().items():
        own[key].copy_(torch.from_numpy(value))
    feats = data.to_device(feats_np, dev)
    opt = optim.Adam(model.parameters(), lr=0.01, fused_loop=True)

    def step():
        opt.zero_grad()
        loss, _ = model(feats)
        F_hip.backward(loss)
        opt.step()

    ms = timed_leg(step, 10, 3)
    blocks = blocks_report(10, 3)
    ops.check_persistent_status()
    if os.environ.get('MG_BENCH_DEBUG') == '1':          # where the last persistent launch's workgroups sat: XCC ids per group
        import collections
        for key, ws in ops._PERSIST_WORKSPACES.items():
            words = ws.view(torch.int32)[256:512].cpu().reshape(8, 32)
            sys.stderr.write('c4 leg blocks %s; XCC ids per group of the last launch: %s\n' % (LAST_BLOCKS, '  '.join(
                'g%d:%s' % (g, ','.join('%dx%d' % (k, n) for k, n in sorted(collections.Counter(int(v) - 1 for v in words[g] if int(v) > 0).items())))
                for g in range(8))))
    frames = int(feats_np['n_frames'].sum())
    # fwd MACs per frame 307,200 + 786,432 + 786,432 + 131,072 + 20,480; backward = wgrad of everything + dgrad of all
    # but the first layer
    macs = 307200 + 786432 + 786432 + 131072 + 20480
    flops = 2.0 * frames * (3 * macs - 307200)
    return {'workload': 'C4: RNN_SPSS Linear-512 / GRU-512 / Linear-256 / 80, 64 x 1000 frames, eager launches, %s' % precision,
            'ms_per_step': round(ms, 4), 'value': round(frames / (ms * 1e-3), 1), 'unit': 'frames/s',
            'tflops': round(flops / (ms * 1e-3) / 1e12, 2), 'frac_of_mfma_peak': round(flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            'note': 'T = 1000 dependent steps per direction bound the step (latency chain), not the MFMA rate', **blocks}


def other_workloads(dev, precision):
    """The other workloads of `--config` (C5, the shipped LSTM acoustic model, the shipped GRU F0 model) timed in the same run as the
    headline - 5 eager steps after 2 warm-up steps each - so that their numbers are driver-timed as well (C4 has its own leg)."""
    out = {}
    specs = [
        ('c5', 'C5: RNN_SPSS Linear-512 / GRU-512 / Linear-256 / 187, 64 utterances of 300-2000 frames',
         lambda: (synthetic.make_batch(64, (300, 2000), out_dim=187, target_name='mcep'), models.RNNSPSS(output_dim=187, precision=precision),
                  synthetic.rnn_spss_state(out_dim=187), False)),
        ('lstm', 'shipped LSTM acoustic model 609-512-8xLSTM512-256-199, 64 x 1000 frames, with its per-step MLPG + 4 streaming metrics',
         lambda: (synthetic.make_acoustic_batch(64, 1000, with_raw=True), models.LSTMAcousticModel(precision=precision, generate=True),
                  synthetic.lstm_acoustic_state(), True)),
        ('f0gru', 'shipped GRU F0 model 609-256-3xGRU64-64-3, 64 x 1000 frames, with its per-step MLPG + LF0 metric',
         lambda: (synthetic.make_acoustic_batch(64, 1000, streams=(('lf0', 3, 'mse'),), with_raw=True),
                  models.GRUF0Model(precision=precision, generate=True), synthetic.gru_f0_state(), True)),
    ]
    for key, what, make in specs:
        try:
            feats_np, model, state, acoustic = make()
            model = model.to(dev)
            own = model.state_dict()
            for k, v in state.items():
                own[k].copy_(torch.from_numpy(v))
            if acoustic:
                synthetic.acoustic_normalisers(model, device=dev)
                model.mode = 'train'
                model.metrics.reset_state('train')
            feats = data.to_device(feats_np, dev, bf16_tables=model.bf16_table_features())
            opt = optim.Adam(model.parameters(), lr=0.01, fused_loop=True)

            def step():
                opt.zero_grad()
                loss, _ = model(feats)
                F_hip.backward(loss)
                opt.step()

            ms = timed_leg(step, 5, 2)
            ops.check_persistent_status()
            frames = int(feats_np['n_frames'].sum())
            out[key] = {'workload': what + ', eager launches, %s' % precision, 'ms_per_step': round(ms, 4),
                        'value': round(frames / (ms * 1e-3), 1), 'unit': 'frames/s', **blocks_report(5, 2)}
            if key == 'c5':
                # the same batch (a) with every row-wise product AND the recurrent weight gradients on all B * T padded rows, (b) with
                # the row-wise runs packed as well (the default packs the recurrent weight gradients only: utils.PACK_ROWS_MIN_PADDING)
                from morgana_amd import utils as mg_utils
                default_min = mg_utils.PACK_ROWS_MIN_PADDING
                try:
                    mg_utils.set_packed_frames(False)
                    out[key]['padded_rows_ms_per_step'] = round(timed_leg(step, 5, 2), 4)
                    mg_utils.set_packed_frames(True, rows_min_padding=0.1)
                    out[key]['all_runs_packed_ms_per_step'] = round(timed_leg(step, 5, 2), 4)
                finally:
                    mg_utils.set_packed_frames(True, rows_min_padding=default_min)
                out[key]['packing'] = ('default: recurrent weight gradients over the valid frames, row-wise runs on the padded rows '
                                       '(utils.PACK_ROWS_MIN_PADDING = %.2f)' % default_min)
            del model, opt, feats
        except Exception as exc:                          # noqa: BLE001 - a leg that fails is reported, the headline stands
            out[key] = {'error': str(exc).splitlines()[0][:200]}
    return out


def exchange_record(optimizer, step, world, rehearse, device, calls=20):
    """First-contact diagnostics of the gradient exchange, carried by EVERY bench line:
    how many ranks a real all-reduce counted, how the exchange ran (mode), what one exchange of the model's flat gradient costs on
    its own (`us`: device events around `calls` back-to-back all-reduces of a buffer of that size, per-call mean, MAX over ranks;
    host clock when the group's tensors live on the CPU), and what the capture probe found (`capture_probe`: its verdict, whether the
    captured graph held a collective at all, whether the replayed sums equalled the world size).  `exposed_us` - the step with the
    exchange minus the step without it - is filled in by the caller (it needs a second timed loop).  Works on any process group:
    the CPU / gloo test drives it without a GPU."""
    dist = torch.distributed
    live = dist.is_available() and dist.is_initialized()
    rec = {'world_size': dist.get_world_size() if live else 1, 'backend': dist.get_backend() if live else None,
           'mode': (getattr(step, 'exchange_mode', None) or ('none (one rank)' if world == 1 and not rehearse else 'eager all-reduce')),
           'ranks_counted': 1, 'us': None, 'bytes': None, 'exposed_us': None, 'capture_probe': None}
    if not live:
        return rec
    on_gpu = dist.get_backend() == 'nccl'
    where = device if on_gpu else torch.device('cpu')
    count = torch.ones(1, dtype=torch.float32, device=where)
    dist.all_reduce(count)                                             # a real collective: every rank that is really there adds 1
    rec['ranks_counted'] = int(round(float(count.item())))
    flat = optimizer.flat_buffers() if optimizer is not None else None
    if flat is not None and optimizer.exchanging():
        scratch = torch.zeros(flat['grad'].numel(), dtype=torch.float32, device=flat['grad'].device)
        rec['bytes'] = int(scratch.numel() * 4)
        dist.all_reduce(scratch)                                       # communicator / buffers warm
        if scratch.is_cuda:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(calls):
                dist.all_reduce(scratch)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) / calls * 1e3
        else:
            t0 = time.perf_counter()
            for _ in range(calls):
                dist.all_reduce(scratch)
            us = (time.perf_counter() - t0) / calls * 1e6
        worst = torch.tensor([us], dtype=torch.float64, device=where)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        rec['us'] = round(float(worst.item()), 2)
    if on_gpu:
        from morgana_amd import graphs
        probe = dict(graphs.rccl_capture_report(device))
        if probe['world'] > 1 and probe['verdict']:
            # more than one rank and a green verdict: the replayed graph DID sum over the ranks
            assert probe['graph_held_collective'] and probe['replayed_sum_equals_world'], probe
        rec['capture_probe'] = probe
    else:
        rec['capture_probe'] = {'verdict': False, 'world': rec['world_size'], 'backend': rec['backend'],
                                'error': 'not an RCCL group: nothing to capture'}
    return rec


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver's command line would
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>), as a CHILD
    process - this process has made no GPU call yet and makes none - and exit with its code.  Rank 0 of the children prints the
    JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sock:                    # a free port on the loopback interface
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be at least 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    rank, local_rank, world = distributed.init()
    if world != args.gpus:                           # never a silent N = 1: the line's n_gpus is the number of ranks that ran
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world))
    n_gpus = world
    if args.dry_run_ranks:
        count = torch.ones(1)
        if world > 1:
            if torch.distributed.get_backend() == 'nccl':
                count = count.cuda()
            torch.distributed.all_reduce(count)
        if rank == 0:
            print(json.dumps({'n_gpus': n_gpus, 'ranks_counted': int(count.item()), 'dry_run': True,
                              'backend': torch.distributed.get_backend() if world > 1 else None}))
        return
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)          # several ranks may share one GPU in the gloo rehearsal on a 1-GPU box
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)

    torch.manual_seed(synthetic.REFERENCE_SEED)
    if args.config == 'c2':
        per_gpu = args.batch or 256
        feats_np = synthetic.make_batch(per_gpu, args.frames, rank=rank)
        model = models.F0Model(precision=args.precision).to(dev)
        state = synthetic.f0_model_state()
        name, target = 'F0Model 600-512-128-32-1 (README.rst:65-73)', 'normalised_lf0'
    elif args.config == 'c4':
        per_gpu = args.batch or 64
        feats_np = synthetic.make_batch(per_gpu, args.frames, out_dim=80, target_name='mcep', rank=rank)
        model = models.RNNSPSS(precision=args.precision).to(dev)
        state = synthetic.rnn_spss_state()
        name, target = 'RNN_SPSS Linear-512/GRU-512/Linear-256/80 (models/RNN_SPSS.py:32-42 layout)', 'normalised_mcep'
    elif args.config == 'f0gru':
        per_gpu = args.batch or 64
        feats_np = synthetic.make_acoustic_batch(per_gpu, args.frames, streams=(('lf0', 3, 'mse'),), rank=rank, with_raw=True)
        model = models.GRUF0Model(precision=args.precision, generate=not args.no_generate).to(dev)
        state = synthetic.gru_f0_state()
        name, target = 'shipped F0 model 609-256-3xGRU64-64-3 (models/f0_test_model.py:28-45)', 'lf0 deltas'
    elif args.config == 'lstm':
        per_gpu = args.batch or 64
        feats_np = synthetic.make_acoustic_batch(per_gpu, args.frames, rank=rank, with_raw=True)
        model = models.LSTMAcousticModel(precision=args.precision, generate=not args.no_generate).to(dev)
        state = synthetic.lstm_acoustic_state()
        name, target = 'LSTMAcousticModel 609-512-8xLSTM512-256-199 (models/RNN_SPSS.py:32-42)', 'lf0/vuv/mcep/bap streams'
    else:
        per_gpu = args.batch or 64
        feats_np = synthetic.make_batch(per_gpu, (300, 2000), out_dim=187, target_name='mcep', rank=rank)
        model = models.RNNSPSS(output_dim=187, precision=args.precision).to(dev)
        state = synthetic.rnn_spss_state(out_dim=187)
        name, target = 'RNN_SPSS Linear-512/GRU-512/Linear-256/187 WORLD params, ragged 300-2000 frames', 'normalised_mcep'
    own = model.state_dict()
    for key, value in state.items():
        own[key].copy_(torch.from_numpy(value))
    if args.config in ('lstm', 'f0gru'):
        # delta-stream normaliser parameters as ExperimentBuilder would load them: the step then includes the shipped models'
        # MLPG + streaming metrics (models/RNN_SPSS.py:84-129, models/f0_test_model.py:78-105), all on the device
        synthetic.acoustic_normalisers(model, device=dev)
        model.mode = 'train'
        model.metrics.reset_state('train')
    # the product's loader call (data.to_device = ToDeviceWrapper, data.py:648-663): in bf16 precision the batch carries the bf16
    # operand table of the model's phone-level input, exactly as ExperimentBuilder.train_epoch's batches do
    features = data.to_device(feats_np, dev, bf16_tables=model.bf16_table_features())
    frames_per_step = int(feats_np['n_frames'].sum())
    rehearse = bool(args.rehearse_exchange) and world == 1
    if rehearse:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(dev)
        torch.distributed.init_process_group(backend='nccl', rank=0, world_size=1)
    # the loop below is the reference's loop body
    optimizer = optim.Adam(model.parameters(), lr=0.01, fused_loop=True, exchange_always=rehearse)

    def step():
        optimizer.zero_grad()
        loss, _ = model(features)
        F_hip.backward(loss)             # what ExperimentBuilder.train_epoch calls: loss.backward() with a cached unit gradient
        optimizer.step()
        return loss

    # The F0Model step's kernels sum to ~0.3 ms, less than the Python / autograd / launch path around them: replay the step as a
    # HIP graph (same kernels, same buffers; morgana_amd/graphs.py).  The longer recurrent steps gain nothing from it.
    graph_note = None
    per_call = 1                                      # training steps one call of step() performs
    if args.config == 'c2' and not args.no_graph:
        # K steps per graph: the idle time between two graph launches (8-9 us) and the launch that stages Adam's step-dependent
        # scalars (4.7 us) are then paid once per K steps; every step still does all of its work (morgana_amd/graphs.py)
        # ... and at least two graph launches in the timed region where --steps allows (the driver's --steps 20: 2 x 10): a graph's
        # kernels start once its launch has been enqueued, which the second launch does while the first one runs
        want_k = args.steps_per_replay or max(k for k in range(1, min(25, max(args.steps // 2, 1)) + 1) if args.steps % k == 0)
        if args.steps % want_k:
            raise SystemExit('--steps-per-replay must divide --steps')
        from morgana_amd import graphs
        try:
            # with an eager gradient exchange (more than one rank and no RCCL capture) the object falls back to one step per replay
            step = graphs.GraphedTrainStep(model, optimizer, features, steps_per_replay=want_k)
            per_call = step.steps_per_replay
            how = 'forward+backward graph, eager all-reduce, update kernel' if step.exchange_mode == 'eager' else \
                ('%d steps per graph' % per_call if per_call > 1 else 'one graph per step')
            graph_note = 'hip graph replay (%s)' % how
        except Exception as exc:                      # capture refused: time the eager loop and say so
            graph_note = 'eager launches (graph capture failed: %s)' % str(exc).splitlines()[0][:200]
            torch.cuda.synchronize()
    import gc
    # no cyclic collection (and none of the device frees it can trigger) inside a timed region - and the collection itself (tens of
    # milliseconds of host time) goes IN FRONT of the warm-up steps: the device must not sit idle between them and the clock, or the
    # --steps 20 of the driver's run (one graph launch, 2 ms) is timed on a chip that has dropped its clocks
    gc.collect()
    gc.disable()
    # Warm-up: --warmup steps, and for the graph replay at least WARM_MS of device work - a replayed F0Model step is 0.1 ms, and a chip
    # that has idled through model construction and capture needs tens of milliseconds of load before it holds the clock of a training
    # run (20 timed steps behind 10 warm-up steps read 0.115 ms, behind 400: 0.106 - the number a run of 200 steps gives either way).
    # The steps actually run are reported (config.warmup_steps_run); every timed step does all of its work.
    warm_calls = -(-args.warmup // per_call)
    if graph_note is not None and not graph_note.startswith('eager'):
        warm_calls = max(warm_calls, -(-WARM_STEPS_GRAPH // per_call))
    for _ in range(warm_calls):
        loss = step()
    torch.cuda.synchronize()
    distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps // per_call):          # per_call divides --steps (it is want_k or 1): exactly --steps steps are timed
        loss = step()
    torch.cuda.synchronize()
    distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ops.check_persistent_status()        # raises if a persistent recurrent kernel timed out (results would be invalid)
    final_loss = float(distributed.mean_scalar(loss.detach()).item())

    # the exchange's own record (every rank takes part in its collectives), and - where the step has an exchange - the same step
    # timed WITHOUT it right after: exposed_us = what the data-parallel form costs a step beyond the one-rank step
    exchange = exchange_record(optimizer, step, world, rehearse, dev)
    if optimizer.exchanging() and args.config == 'c2' and not args.no_compare:
        try:
            from morgana_amd import graphs
            alone_model = models.F0Model(precision=args.precision).to(dev)
            alone_model.load_state_dict(model.state_dict())
            alone_opt = optim.Adam(alone_model.parameters(), lr=0.01, fused_loop=True, exchange_never=True)
            if per_call > 1 or (graph_note is not None and not graph_note.startswith('eager')):
                alone = graphs.GraphedTrainStep(alone_model, alone_opt, features, steps_per_replay=per_call)
            else:
                def alone():
                    alone_opt.zero_grad()
                    alone_loss, _ = alone_model(features)
                    F_hip.backward(alone_loss)
                    alone_opt.step()
            gc.collect()
            gc.disable()
            for _ in range(max(warm_calls // 4, 2)):
                alone()
            torch.cuda.synchronize()
            distributed.barrier()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps // per_call):
                alone()
            torch.cuda.synchronize()
            distributed.barrier()
            torch.cuda.synchronize()
            alone_s = time.perf_counter() - t2
            gc.enable()
            if world > 1:
                tmax2 = torch.tensor([alone_s], dtype=torch.float64, device=dev)
                torch.distributed.all_reduce(tmax2, op=torch.distributed.ReduceOp.MAX)
                alone_s = float(tmax2.item())
            exchange['step_without_exchange_ms'] = round(alone_s / args.steps * 1e3, 4)
            exchange['exposed_us'] = round((elapsed - alone_s) / args.steps * 1e6, 2)
        except Exception as exc:                                            # noqa: BLE001 - a diagnostic leg: reported, never fatal
            exchange['exposed_us_error'] = str(exc).splitlines()[0][:200]
        finally:
            gc.enable()

    # C2, one rank: the same model and batch with every product at frame rate (the reference's order of operations,
    # MORGANA_PHONE_RATE=0), timed the same way right after - reported next to the headline value, never as it
    frame_rate = None
    if (args.config == 'c2' and world == 1 and args.precision == 'bf16' and ops.PHONE_RATE and not args.no_graph
            and not args.no_compare):
        try:
            from morgana_amd import graphs
            # the order of operations is the MODEL's choice (base_models.BaseModel.phone_rate): no process-wide switch is written
            fr_model = models.F0Model(precision=args.precision, phone_rate=False).to(dev)
            fr_model.load_state_dict(model.state_dict())
            fr_step = graphs.GraphedTrainStep(fr_model, optim.Adam(fr_model.parameters(), lr=0.01, fused_loop=True), features,
                                              steps_per_replay=per_call)
            gc.collect()
            gc.disable()
            for _ in range(max(-(-args.warmup // per_call), -(-WARM_STEPS_GRAPH // (4 * per_call)))):      # 0.5 ms steps: a quarter as many
                fr_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps // per_call):
                fr_step()
            torch.cuda.synchronize()
            fr_ms = (time.perf_counter() - t1) / args.steps * 1e3
            gc.enable()
            fr_tflops = F0_FLOPS_PER_FRAME * frames_per_step / (fr_ms * 1e-3) / 1e12
            try:
                sustained, calib_ms, calib_wgs = sustained_mfma_tflops(dev)
                calibration = {'sustained_bf16_mfma_tflops': round(sustained, 1), 'frac_of_mfma_peak': round(sustained / MFMA_BF16_PEAK_TFLOPS, 4),
                               'what': 'mg_calib_mfma_bf16 right after the frame-rate leg: register-operand v_mfma_f32_16x16x32_bf16 loop on '
                                       'random operands, %d workgroups x 8 waves, %.1f ms of launches under HIP events - the matrix rate THIS '
                                       'chip holds under its power limit (the 2.5 PFLOP/s peak assumes 2.4 GHz)' % (calib_wgs, calib_ms)}
            except Exception as exc:                                        # noqa: BLE001 - calibration only
                sustained, calibration = None, {'error': str(exc).splitlines()[0][:200]}
            frame_rate = {'ms_per_step': round(fr_ms, 4), 'value': round(frames_per_step / (fr_ms * 1e-3), 1), 'unit': 'frames/s',
                          'tflops': round(fr_tflops, 1), 'frac_of_mfma_peak': round(fr_tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                          'frac_of_sustained': (round(fr_tflops / sustained, 4) if sustained else None), 'calibration': calibration,
                          'what': 'F0Model(phone_rate=False): the reference\'s order of operations - every product on the B*T frame rows '
                                  '(gather-fused layer-1 GEMM, fused dgrad+wgrad), same graph replay; tflops = the algorithmic '
                                  '%.1f GFLOP per step (SURVEY.md section 8d) over this time: an ACHIEVED rate, the matrix cores '
                                  'execute every one of those products' % (F0_FLOPS_PER_FRAME * frames_per_step / 1e9)}
        except Exception as exc:
            frame_rate = {'error': str(exc).splitlines()[0][:200]}
        finally:
            gc.enable()

    result = None
    form_note = ''
    phone_rate_step = False
    if args.config == 'c2':
        lab_shape = feats_np['normalised_lab'].shape
        phone_rate_step = (args.precision == 'bf16' and ops.phone_rate_table_ok(lab_shape[0] * lab_shape[1], frames_per_step, 512, 128,
                                                                                ops.ACT_SIGMOID))
        if phone_rate_step:
            form_note = ('; STEP AT PHONE RATE: the model has no frame-level input, so Linear / Sigmoid commute with '
                         'upsample_to_repetitions and every layer runs on the %d phone rows (+ %d zero rows) instead of the %d frame '
                         'rows - %.1fx fewer products than the reference\'s order of operations, same outputs (frame_rate_order '
                         'times that order); bf16 phone table prepared by the loader'
                         % (lab_shape[0] * lab_shape[1], ops.PHONE_RATE_EXTRA, frames_per_step,
                            frames_per_step / float(lab_shape[0] * lab_shape[1] + ops.PHONE_RATE_EXTRA)))
        elif args.precision in ('fp32', 'bf16x3') and ops.phone_rate_gru_ok(lab_shape[0] * lab_shape[1], frames_per_step, 8):
            # fp32 parity mode takes the generic phone-rate form (utils.SequentialWithRecurrent: the row-wise stack on the phone rows,
            # its output repeated, exact-fp32 MFMA products): not the reference's order of operations either
            form_note = ('; %s parity mode AT PHONE RATE as well (generic form: the Linear / Sigmoid stack on the %d phone rows + %d zero '
                         'rows, its output repeated to the %d frames; %s)'
                         % (args.precision, lab_shape[0] * lab_shape[1], ops.PHONE_RATE_EXTRA, frames_per_step,
                            'exact-fp32 MFMA products, the loss curve matches the reference to 1e-7' if args.precision == 'fp32' else
                            'split-bf16 operands: three bf16 MFMA products per fp32 product'))
        else:
            form_note = '; every product at frame rate (the reference\'s order of operations)'
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_gpus * frames_per_step * args.steps / elapsed
        result = {
            'metric': {'c2': 'acoustic frames/sec (fwd+bwd+step), F0Model 600->1, batch 256x1000',
                       'c4': 'acoustic frames/sec (fwd+bwd+step), RNN_SPSS GRU-512 600->80, batch 64x1000',
                       'c5': 'acoustic frames/sec (fwd+bwd+step), RNN_SPSS GRU-512 600->187, batch 64 x 300-2000 frames',
                       'lstm': 'acoustic frames/sec (fwd+bwd+step), LSTMAcousticModel 8xLSTM-512 609->199, batch 64x1000',
                       'f0gru': 'acoustic frames/sec (fwd+bwd+step), shipped GRU F0 model 3xGRU-64 609->3, batch 64x1000'}[args.config],
            'value': round(value, 1), 'unit': 'frames/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': '%s, %d utterances x %d frames per GPU, P=%d phones, synthetic lab/dur/%s, '
                                   'Adam lr 0.01, fp32 master weights%s' % (name, per_gpu, args.frames,
                                                                            feats_np['dur'].shape[1], target, form_note),
                       'global_batch': per_gpu * n_gpus, 'frames_per_utterance': args.frames,
                       'parallelism': 'dp%d' % n_gpus},
            'final_loss': round(final_loss, 6),
        }
        result['exchange'] = exchange          # how the gradients crossed the ranks (exchange_record), all ranks took part
        result['ranks_counted'] = exchange['ranks_counted']
        # top-level `warmup` = the flag (what the driver cross-checks against its command line: ADVICE round 4); the untimed steps that
        # actually ran in front of the clock (>= the flag: see the warm-up comment above) are reported beside it
        result['config']['warmup_steps_run'] = warm_calls * per_call
        if graph_note is not None:
            result['config']['launch'] = graph_note
            result['config']['steps_per_graph_launch'] = per_call      # every step does all of its work; the launch gap is shared
        if rehearse:
            result['config']['rehearsal'] = 'multi-rank code path on one rank (world-size-1 RCCL group): not the headline form'
        if frame_rate is not None:
            result['frame_rate_order'] = frame_rate
        if args.config == 'c2':
            peak = MFMA_BF16_PEAK_TFLOPS if args.precision == 'bf16' else MFMA_F32_PEAK_TFLOPS
            algorithmic = F0_FLOPS_PER_FRAME * frames_per_step
            if phone_rate_step:
                r_tab = lab_shape[0] * lab_shape[1] + ops.PHONE_RATE_EXTRA
                fwd = lab_shape[2] * 512 + 512 * 128 + 128 * 32 + 32
                executed = 2.0 * r_tab * (2 * fwd + (512 * 128 + 128 * 32 + 32))      # forward + wgrad of 4 layers, dgrad of layers 2-4
            else:
                executed = algorithmic
            ex_tflops = executed / (ms_per_step * 1e-3) / 1e12
            result['step_executed'] = {'gflop_per_step': round(executed / 1e9, 1), 'tflops': round(ex_tflops, 2),
                                       'frac_of_mfma_peak': round(ex_tflops / peak, 4),
                                       'what': 'what the matrix cores multiply in one step of the headline form'}
            if phone_rate_step and args.precision == 'bf16':
                # The whole step against its floors.  Algorithmic bytes = every operand once per kernel that
                # needs it + every result once, in the precision the step stores them (no split-M slabs: those are the implementation's):
                #   layer-1 forward   table bf16 + W1 + H1 write | frame map + targets + statistics
                #   layers 2-4 + loss H1 read + dZ2 write (bf16) + the per-phone prediction
                #   pair              dZ2 + H1 + W2^T read, dZ1 write, dW2
                #   layer-1 wgrad     dZ1 + table read, dW1
                #   update            p, m, v read + written, gradient read, bf16 operand copies written; prediction repeated to frames
                ldk = ops.pad_ld(lab_shape[2])
                n_par = sum(p.numel() for p in model.parameters())
                table_b, h1_b, dz2_b, dz1_b = 2.0 * r_tab * ldk, 2.0 * r_tab * 512, 2.0 * r_tab * 128, 2.0 * r_tab * 512
                w_b = 2.0 * (512 * ldk + 2 * 128 * 512)
                step_bytes = ((table_b + w_b + h1_b + 12.0 * frames_per_step + 16.0 * r_tab) + (h1_b + dz2_b + 8.0 * r_tab) +
                              (dz2_b + h1_b + dz1_b + 4.0 * 128 * 512) + (dz1_b + table_b + 4.0 * 512 * lab_shape[2]) +
                              (28.0 * n_par + w_b + 8.0 * frames_per_step))
                floor_hbm_ms = step_bytes / (HBM_PEAK_GBS * 1e9) * 1e3
                floor_mfma_ms = executed / (peak * 1e12) * 1e3
                floor_ms = max(floor_hbm_ms, floor_mfma_ms)
                result['step_floor'] = {'algorithmic_bytes': round(step_bytes), 'gflop': round(executed / 1e9, 1),
                                        'hbm_floor_ms': round(floor_hbm_ms, 4), 'mfma_floor_ms': round(floor_mfma_ms, 4),
                                        'floor_ms': round(floor_ms, 4), 'ms_per_step_over_floor': round(ms_per_step / floor_ms, 2),
                                        'what': 'the step\'s algorithmic HBM bytes (operands and results once per kernel, no split-M slabs) at 8 TB/s and '
                                                'its executed FLOPs at the 2.5 PFLOP/s bf16 MFMA peak; the larger is the floor.  The step is five '
                                                'launches of one-tile latency chains: DESIGN.md says where the factor goes'}
            if phone_rate_step:
                # NOT a fraction of peak: the reference algorithm's FLOPs over a step that performs 12x fewer of them
                result['reference_equivalent'] = {'gflop_per_step': round(algorithmic / 1e9, 1),
                                                  'tflops_equivalent': round(algorithmic / (ms_per_step * 1e-3) / 1e12, 1),
                                                  'what': 'FLOPs of the reference\'s order of operations (SURVEY.md section 8d) divided by the '
                                                          'phone-rate step time: an equivalence figure, not an achieved rate and not a '
                                                          'fraction of any roof - the achieved frame-rate figure is frame_rate_order'}
    if rank == 0 and args.config == 'c2' and not args.no_roofline:
        result['roofline'] = roofline_f0(features, model, args.precision)
    if rank == 0 and args.config in ('c4', 'c5') and not args.no_roofline:
        result['roofline'] = roofline_gru(features, model, args.precision, target)
    if rank == 0 and args.config == 'lstm' and not args.no_roofline:
        roof = roofline_lstm(features, model, args.precision)
        if roof is not None:
            result['roofline'] = roof
    if rank == 0 and n_gpus == 1 and args.config == 'c2' and args.precision == 'bf16' and not args.no_compare:
        try:
            result['c4'] = c4_leg(dev, args.precision)
        except Exception as exc:
            result['c4'] = {'error': str(exc).splitlines()[0][:200]}
        result['other_workloads'] = other_workloads(dev, args.precision)
        result['bf16x3'] = bf16x3_legs(dev, features, model.state_dict(), args.steps, frames_per_step)
        result['train_epoch'] = train_epoch_block(dev, model.state_dict(), ms_per_step)
        try:
            result['loss_curve_deviation'] = loss_curve_deviation(dev)
        except Exception as exc:
            result['loss_curve_deviation'] = {'error': str(exc).splitlines()[0][:200]}
    if rank == 0 and n_gpus == 1 and args.config == 'c2' and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline_f0(per_gpu, args.frames)
    distributed.barrier()
    if rank == 0:
        print(json.dumps(result))
    if torch.distributed.is_initialized():           # the ranks' group, or the one-rank rehearsal's
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
