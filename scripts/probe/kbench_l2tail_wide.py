"""mg_f0_l2tail_bf16 at the C2 frame-rate shape: f0_l2tail_kernel (the product kernel) and - lab library only - the wide form
(MG_TUNE_AB 69, l2tail_wide.hip) with its timing probes.  usage: python scripts/probe/kbench_l2tail_wide.py [iters]   (MORGANA_HIP_LIB=morgana_amd/libmorgana_hip_lab.so for the probes)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morgana_amd import _lib, ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = 'cuda:0'
lib = _lib.load()
b, t = 256, 1000
m = b * t
h1 = torch.rand(m, 512, device=dev).to(torch.bfloat16)
w2b = (torch.randn(128, 512, device=dev) * 0.05).to(torch.bfloat16)
b2 = torch.zeros(128, device=dev)
w3, b3 = torch.randn(32, 128, device=dev) * 0.1, torch.zeros(32, device=dev)
w4, b4 = torch.randn(1, 32, device=dev) * 0.1, torch.zeros(1, device=dev)
tgt = torch.randn(m, device=dev)
sl = torch.full((b,), t, dtype=torch.int64, device=dev)
grads = torch.empty(4162, device=dev)
def timed(fn):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters * 1e3
names = {69: 'wide form', 0: 'f0_l2tail_kernel', 101: 'wide: no phase 2', 102: 'wide: no fragment reads / MFMAs', 103: 'wide: stream only',
         104: 'wide: no stream', 105: 'wide: phase 1 compute only', 106: 'wide: phase 2 only', 108: 'wide: no dZ2 / pred stores'}
for probe, what in names.items():
    if lib.mg_set_tuning(7, probe) != 0:
        continue
    print('%-40s %8.1f us' % (what, timed(lambda: ops.f0_l2tail(h1, w2b, b2, w3, b3, w4, b4, tgt, sl, b, t, grads))))
lib.mg_set_tuning(7, 0)
