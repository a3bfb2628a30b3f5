"""Worker for tests/test_gpu_configs.py::test_two_ranks_on_one_gpu_equal_the_global_batch: one rank of a data-parallel run of the
PRODUCT step (HIP kernels through the C ABI, device tensors) whose ranks SHARE the one GPU of the box; the process group is gloo
(RCCL refuses two ranks on one device), so the gradient exchange is the eager all-reduce of `morgana_amd.optim.Adam.step` /
`graphs.GraphedTrainStep` on device buffers.  argv: out.npz n_steps precision form(graph|eager) ragged(0|1) [model: f0 | rnn187 | lstm]."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from morgana_amd import data, distributed, graphs, models, optim, synthetic  # noqa: E402
from morgana_amd import functional as F_hip  # noqa: E402


def global_batch(which, ragged):
    """The global batch of 16 utterances a model of the test is trained on (sharded 2 x 8 across the ranks)."""
    if which == 'rnn187':           # BASELINE config C5's model: RNN_SPSS GRU-512 with 187 WORLD outputs, variable lengths
        return synthetic.make_batch(16, (120, 400) if ragged else 250, out_dim=187, target_name='mcep', seed=23)
    if which == 'lstm':             # the reference's shipped acoustic model (models/RNN_SPSS.py), four output streams
        return synthetic.make_acoustic_batch(16, (60, 160) if ragged else 120, seed=23, with_raw=True)
    return synthetic.make_batch(16, (120, 400) if ragged else 250, seed=23)


def make_model(which, precision, dev):
    if which == 'rnn187':
        model, state = models.RNNSPSS(precision=precision, output_dim=187).to(dev), synthetic.rnn_spss_state(out_dim=187)
    elif which == 'lstm':
        model, state = models.LSTMAcousticModel(precision=precision, generate=False).to(dev), synthetic.lstm_acoustic_state()
    else:
        model, state = models.F0Model(precision=precision).to(dev), synthetic.f0_model_state()
    own = model.state_dict()
    for key, value in state.items():
        own[key].copy_(torch.from_numpy(value))
    return model


def run_steps(feats_np, n_steps, precision, form, dev, which='f0'):
    model = make_model(which, precision, dev)
    feats = data.to_device(feats_np, dev, bf16_tables=model.bf16_table_features())
    opt = optim.Adam(model.parameters(), lr=0.01, fused_loop=True)

    def eager_step():
        opt.zero_grad()
        loss, _ = model(feats)
        F_hip.backward(loss)
        opt.step()
        return loss

    step = graphs.GraphedTrainStep(model, opt, feats, warmup=1) if form == 'graph' else eager_step
    losses = [float(distributed.mean_scalar(step().detach().clone()).item()) for _ in range(n_steps)]
    torch.cuda.synchronize()
    return opt.flat_buffers()['param'].clone(), losses, getattr(step, 'exchange_mode', 'eager loop')


def main():
    out_path, n_steps, precision, form, ragged = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5] == '1'
    which = sys.argv[6] if len(sys.argv) > 6 else 'f0'
    rank, _, world = distributed.init(backend='gloo')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    shard = synthetic.shard_batch(global_batch(which, ragged), rank, world)
    flat, losses, mode = run_steps(shard, n_steps, precision, form, dev, which)
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        np.savez(out_path, losses=np.array(losses), replicas=np.stack([g.cpu().numpy() for g in gathered]), mode=np.array(mode))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
