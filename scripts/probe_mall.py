"""Does a buffer written by one launch come back faster when the next launch reads its LAST-written part first?

MI355X has 8 x 4 MB of L2 and a 256 MB memory-side cache; H1 of the frame-rate step is 262 MB, written ascending by the layer-1
forward and read ascending by the next launch - the worst case for an LRU-like cache of about the buffer's size.  This probe writes a
buffer of `--mb` megabytes in 16 chunks (ascending) and reads it back chunk by chunk ascending or descending, same launch count.
Calibration only (torch elementwise kernels); nothing of the product is involved.
"""
import argparse

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mb', type=int, nargs='+', default=[64, 128, 192, 262, 400])
    ap.add_argument('--chunks', type=int, default=16)
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    for mb in args.mb:
        n = mb * 1024 * 1024 // 2
        buf = torch.empty(n, dtype=torch.bfloat16, device=dev)
        src = torch.randn(n // args.chunks, dtype=torch.float32, device=dev).to(torch.bfloat16)
        sink = [None] * args.chunks
        chunks = list(buf.chunk(args.chunks))
        res = {}
        for order in ('ascending', 'descending', 'ascending', 'descending'):
            idx = list(range(len(chunks)))
            if order == 'descending':
                idx.reverse()
            times = []
            for _ in range(args.reps):
                for c in chunks:                              # the producer: ascending
                    c.copy_(src[:c.numel()])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in idx:
                    sink[i] = chunks[i].view(torch.int16).sum(dtype=torch.int32)
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) * 1e3)
            times.sort()
            res.setdefault(order, []).append(times[len(times) // 2])
        print('%4d MB written ascending, read back in %d launches: ascending %s us, descending %s us  (%.2f / %.2f TB/s)' % (
            mb, args.chunks, ' '.join('%.1f' % t for t in res['ascending']), ' '.join('%.1f' % t for t in res['descending']),
            mb * 1.048576e6 / min(res['ascending']) / 1e6, mb * 1.048576e6 / min(res['descending']) / 1e6), flush=True)


if __name__ == '__main__':
    main()
