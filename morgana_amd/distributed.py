"""Data-parallel plumbing: one process per GPU, utterances sharded contiguously by rank, ONE all-reduce per step.

The reference has no distributed code at all (single device string, experiment_builder.py:262-263).  Utterances are
independent in ``predict`` and the loss is a mean over utterances of per-utterance means (losses.py:37-42), so with
equal per-rank batch sizes  L_global = mean_r L_r  and  grad L_global = (1/R) sum_r grad L_r  exactly.
The exchange itself lives in ``morgana_amd.optim.Adam.step`` (flat fp32 bucket, RCCL over xGMI on the GPU box,
gloo in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init(backend=None):
    """Initialise the default process group from torchrun's environment; returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = os.environ.get('MG_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_slice(batch_size, rank, world):
    """Rows [r*B/R, (r+1)*B/R) of the global batch."""
    if batch_size % world != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (batch_size, world))
    per = batch_size // world
    return slice(rank * per, (rank + 1) * per)


def broadcast_parameters(model, src=0):
    """Identical initial replicas (ranks normally already agree through the shared seed)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src=src)
            p._mg_updates = getattr(p, '_mg_updates', 0) + 1      # written through .data: bf16 operand copies of it are stale (ops.mark_updated)


def mean_scalar(value):
    """Average a 0-d device tensor over ranks (reporting only; not on the per-step path)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        value = value.clone()
        dist.all_reduce(value, op=dist.ReduceOp.SUM)
        value /= dist.get_world_size()
    return value


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
