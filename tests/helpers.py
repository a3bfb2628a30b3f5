"""Shared test helpers (CPU side).  The CPU Adam kernel below is the oracle's update rule applied to torch CPU tensors:
it is injected through ``morgana_amd.optim.Adam(kernel=...)`` so that host logic (flat buckets, sharding, the single
all-reduce, LR schedules, the epoch loop) can be tested without a GPU.  The product never uses it."""
import numpy as np
import torch

from morgana_amd import base_models
from oracle import ref_torch


def cpu_adam_kernel(param, grad, exp_avg, exp_avg_sq, lr, betas, eps, weight_decay, step, grad_scale=1.0):
    b1, b2 = betas
    g = grad * grad_scale
    if weight_decay != 0:
        g = g + weight_decay * param
    exp_avg.lerp_(g, 1 - b1)
    exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (exp_avg_sq.sqrt() / (bc2 ** 0.5)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-(lr / bc1))


class CpuF0Model(base_models.BaseSPSS):
    """BaseSPSS subclass whose predict/loss run the oracle's torch-CPU ops: a stand-in for host-logic tests only."""

    def __init__(self, dims=(24, 16, 8, 1)):
        super(CpuF0Model, self).__init__()
        self.inner = ref_torch.F0Model(dims)
        self.layers = self.inner.layers

    def predict(self, features):
        x = ref_torch.upsample_to_repetitions(features['normalised_lab'], features['dur'])
        return {'pred_norm_lf0': self.layers(x)}

    def loss(self, features, output_features):
        return ref_torch.mse(output_features['pred_norm_lf0'], features['normalised_lf0'], features['n_frames'])


def init_small(model, seed=0):
    rng = np.random.RandomState(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.from_numpy(rng.uniform(-0.3, 0.3, size=tuple(p.shape)).astype(np.float32)))
    return model


def write_g8_dataset(g8, root):
    """Lay the G8 data set out on disk the way the reference run that produced it did (tests/golden/make_golden.py: one .npy per
    utterance and feature under {root}/train/{feature}/, n_frames as .txt, the id list and the normaliser JSON files)."""
    import json
    import os
    names = [str(n) for n in g8['names']]
    for feat in ('lab', 'dur', 'lf0', 'n_frames'):
        os.makedirs(os.path.join(root, 'train', feat), exist_ok=True)
    for name in names:
        for feat in ('lab', 'dur', 'lf0'):
            np.save(os.path.join(root, 'train', feat, name + '.npy'), g8['data__%s__%s' % (name, feat)])
        with open(os.path.join(root, 'train', 'n_frames', name + '.txt'), 'w') as f:
            f.write(str(int(g8['data__%s__dur' % name].sum())))
    with open(os.path.join(root, 'train_file_id_list.scp'), 'w') as f:
        f.write('\n'.join(names) + '\n\n')
    os.makedirs(os.path.join(root, 'processed'), exist_ok=True)
    for key in ('lab_minmax', 'lf0_mvn'):
        params = {k.split('__')[2]: g8[k].tolist() for k in g8 if k.startswith('norm__%s__' % key)}
        with open(os.path.join(root, 'processed', key + '.json'), 'w') as f:
            json.dump(params, f)
    return names


def g8_files_dataset(g8, root, device='cpu'):
    """``data.FilesDataset`` over the G8 data set on disk, normaliser parameters loaded from their JSON files."""
    from morgana_amd import data
    write_g8_dataset(g8, root)
    normalisers = data.Normalisers({'lab': data.MinMaxNormaliser('lab'), 'lf0': data.MeanVarianceNormaliser('lf0')},
                                   'processed', data_root=root, device=device)
    sources = {'n_frames': data.TextSource('n_frames'), 'dur': data.NumpyBinarySource('dur'), 'lab': data.NumpyBinarySource('lab'),
               'lf0': data.NumpyBinarySource('lf0')}
    return data.FilesDataset(sources, 'train', 'train_file_id_list.scp', normalisers, data_root=root)
