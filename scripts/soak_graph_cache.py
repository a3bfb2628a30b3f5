#!/usr/bin/env python
"""Soak of ExperimentBuilder(use_graphs=True) against the eager loop: C2-sized batches of a few shapes in a shuffled order, two
epochs, with the EMA twin - parameters of model and twin, Adam moments and epoch losses must be EQUAL bit for bit (the graph replays
defer the end of the forward to the update launch, the eager steps do not: tests/test_gpu_parity.py::test_graphed_step_defers_the_tail
holds one shape; this holds the cache across shapes).  Usage: python scripts/soak_graph_cache.py [batches per epoch]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from morgana_amd import data, experiment_builder, models, synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    dev = 'cuda:0'
    shapes = [(256, 1000), (128, 800), (256, 600), (64, 1000)]
    rng = np.random.RandomState(3)
    order = [shapes[i] for i in rng.randint(0, len(shapes), size=n)]
    batches = [synthetic.make_batch(b, t, seed=100 + i) for i, (b, t) in enumerate(order)]
    results = {}
    for use_graphs in (False, True):
        eb = experiment_builder.ExperimentBuilder(models.F0Model, model_kwargs={'precision': 'bf16'}, learning_rate=0.01, device=dev,
                                                  ema_decay=0.99, use_graphs=use_graphs)
        state = synthetic.f0_model_state()
        for model in (eb.model, eb.ema_model):
            own = model.state_dict()
            for k, v in state.items():
                own[k].copy_(torch.from_numpy(v))
        opt = eb.make_optimizer()
        losses = []
        for epoch in (1, 2):
            eb.epoch = epoch
            loader = [data.to_device(f, dev, bf16_tables=eb.model.bf16_table_features()) for f in batches]
            losses.append(eb.train_epoch(loader, opt))
        torch.cuda.synchronize()
        flat = opt.flat_buffers()
        results[use_graphs] = (losses, flat['param'].clone(), flat['exp_avg'].clone(), flat['exp_avg_sq'].clone(),
                               torch.cat([p.detach().reshape(-1) for p in eb.ema_model.parameters()]).clone())
        if use_graphs and eb._graph_cache is not None:
            print('graph cache:', eb._graph_cache.stats())
    (le, *te), (lg, *tg) = results[False], results[True]
    print('epoch losses eager', le, 'graphs', lg)
    ok = le == lg and all(torch.equal(a, b) for a, b in zip(te, tg))
    print('EQUAL' if ok else 'DIFFERENT')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
