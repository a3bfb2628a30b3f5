"""Worker for tests/test_distributed_cpu.py: one rank of a gloo data-parallel run on CPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

from morgana_amd import distributed, optim, synthetic  # noqa: E402
from oracle import ref_torch  # noqa: E402
import helpers  # noqa: E402


def main():
    out_path, n_steps, ragged = sys.argv[1], int(sys.argv[2]), sys.argv[3] == 'ragged'
    rank, _, world = distributed.init(backend='gloo')
    torch.set_num_threads(1)
    model = helpers.init_small(helpers.CpuF0Model(dims=(24, 16, 8, 1)), seed=1)
    distributed.broadcast_parameters(model)
    frames = (30, 90) if ragged else 50
    global_batch = synthetic.make_batch(8, frames, lab_dim=24, frames_per_phone=5.0, seed=17)
    shard = ref_torch.to_torch(synthetic.shard_batch(global_batch, rank, world))
    opt = optim.Adam(model.parameters(), lr=0.01, weight_decay=1e-3, kernel=helpers.cpu_adam_kernel)
    losses = []
    for _ in range(n_steps):
        opt.zero_grad()
        loss, _ = model(shard)
        loss.backward()
        opt.step()
        losses.append(float(distributed.mean_scalar(loss.detach())))
    flat = opt.flat_buffers()['param'].numpy().copy()
    gathered = [torch.zeros_like(opt.flat_buffers()['param']) for _ in range(world)]
    dist.all_gather(gathered, opt.flat_buffers()['param'])
    if rank == 0:
        np.savez(out_path, flat=flat, losses=np.array(losses), replicas=np.stack([g.numpy() for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
