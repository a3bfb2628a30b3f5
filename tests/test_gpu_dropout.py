"""Active dropout on the HIP path (``-m gpu``): ``nn.Dropout(p)`` of the shipped models (/root/reference/models/RNN_SPSS.py:19,34,40,
/root/reference/models/f0_test_model.py:22,31-43) in training mode runs as mg_dropout (csrc/dropout.hip) - a counter-based Philox
mask behind the Linear + Sigmoid layers of a fused run and between the recurrent wrappers, regenerated in the backward - never as
torch's nn.Dropout kernel.  Bit parity with torch's mask stream is not attainable (its Philox offsets follow its launch geometry), so
the tests are properties: the generator against its published known-answer vectors and a host restatement, keep fraction and scale,
forward / backward mask identity, the run node against the same computation in torch with the SAME masks, p = 0 and eval mode equal
to today's bits, a new mask per step also under graph replay.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn as nn

from morgana_amd import _lib, data, models, ops, optim, synthetic, utils
from morgana_amd import functional as F_hip

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _philox(counter, key):
    c, k, out = (ctypes.c_uint32 * 4)(*counter), (ctypes.c_uint32 * 2)(*key), (ctypes.c_uint32 * 4)()
    _lib.load().mg_philox4x32_10(c, k, out)
    return [int(v) for v in out]


def _set_counter(value):
    ops.dropout_draw(torch.device(DEV))                                # makes sure the device's counter exists
    ops._dropout_state[torch.device(DEV).index or 0].fill_(int(value))


def test_mask_equals_the_host_restatement_and_has_the_right_statistics():
    n, p, seed, site = 4099, 0.3, 0x1234567890ABCDEF, 7
    used = torch.tensor([5], dtype=torch.int64, device=DEV)
    x = torch.arange(1, n + 1, dtype=torch.float32, device=DEV)
    y = ops.dropout(x, p, seed, site, used)
    thr = int(p * 4294967296.0)
    keep = np.zeros(n, dtype=bool)
    for q in range((n + 3) // 4):
        words = _philox([q, 0, 5, site ^ 0], [seed & 0xFFFFFFFF, seed >> 32])
        for e in range(4):
            if 4 * q + e < n:
                keep[4 * q + e] = words[e] >= thr
    want = np.where(keep, np.arange(1, n + 1, dtype=np.float32) * np.float32(1.0 / (1.0 - p)), 0).astype(np.float32)
    assert np.array_equal(y.cpu().numpy(), want)                         # bit-exact: same generator, same scale product
    # statistics on a larger draw: keep fraction 1 - p, survivors scaled by 1 / (1 - p)
    big = torch.ones(1 << 22, device=DEV)
    yb = ops.dropout(big, p, seed, site, used)
    frac = float((yb != 0).float().mean())
    assert abs(frac - (1 - p)) < 2e-3
    assert torch.all((yb == 0) | (yb == torch.tensor(1.0 / (1.0 - p), device=DEV)))
    # another site, another counter value, another seed: other masks; the same numbers: the same mask; bf16 draws the same mask
    assert not torch.equal(yb, ops.dropout(big, p, seed, site + 1, used))
    assert not torch.equal(yb, ops.dropout(big, p, seed, site, used + 1))
    assert not torch.equal(yb, ops.dropout(big, p, seed + 1, site, used))
    assert torch.equal(yb, ops.dropout(big, p, seed, site, used))
    assert torch.equal(ops.dropout(big.to(torch.bfloat16), p, seed, site, used) != 0, yb != 0)
    with pytest.raises(ValueError):
        ops.dropout(big, 1.0, seed, site, used)


def test_dropout_fn_uses_one_mask_forward_and_backward_and_a_new_one_per_call():
    torch.manual_seed(11)
    x = (torch.rand(64, 333, device=DEV) + 0.5).requires_grad_(True)
    y = F_hip.DropoutFn.apply(x, 0.4, 3)
    g = torch.rand_like(y) + 0.5
    y.backward(g)
    mask = y != 0
    assert torch.equal(x.grad != 0, mask)                               # the backward's mask is the forward's
    assert torch.allclose(y[mask], x.detach()[mask] / 0.6) and torch.allclose(x.grad[mask], g[mask] / 0.6)
    y2 = F_hip.DropoutFn.apply(x, 0.4, 3)
    assert not torch.equal(y2 != 0, mask)                               # the step counter moved on


def _mlp(dims, p, precision):
    mods = []
    for i in range(len(dims) - 1):
        mods.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            mods += [nn.Sigmoid(), nn.Dropout(p=p)]
    return utils.SequentialWithRecurrent(*mods, precision=precision).to(DEV)


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-5), ('bf16x3', 1e-4), ('bf16', 3e-2)])
def test_run_with_dropout_equals_the_same_masks_in_torch(precision, tol):
    """Linear-Sigmoid-Dropout-Linear-Sigmoid-Dropout-Linear as ONE autograd node with HIP masks, against torch fp32 arithmetic with
    the very same masks (drawn through ops.dropout with the node's seed / sites / counter value): outputs and every gradient."""
    torch.manual_seed(3)
    p = 0.25
    net = _mlp((48, 160, 96, 24), p, precision)
    net.train()
    x = torch.randn(6, 50, 48, device=DEV)
    _set_counter(1000)
    calls = []
    _lib.CALL_LOG = calls
    try:
        out, _ = net(x)
    finally:
        _lib.CALL_LOG = None
    assert calls.count('mg_dropout') == 2 and calls.count('mg_dropout_advance') == 1          # one node, one draw, two masks
    w = torch.randn_like(out)
    (out * w).sum().backward()
    got = [out.detach()] + [prm.grad.detach().clone() for prm in net.parameters()]

    used = torch.tensor([1000], dtype=torch.int64, device=DEV)
    seed = ops.dropout_seed()
    lins = [m for m in net if isinstance(m, nn.Linear)]
    m = 6 * 50
    masks = []
    for i, lin in enumerate(lins[:-1]):
        shape = (m, ops.pad_ld(lin.out_features)) if precision == 'bf16' else (m, lin.out_features)
        ones = torch.ones(shape, dtype=torch.bfloat16 if precision == 'bf16' else torch.float32, device=DEV)
        masks.append(ops.dropout(ones, p, seed, 0 + i, used).float()[:, :lin.out_features])     # site0 = index of the run's first module
    ref = [nn.Linear(l.in_features, l.out_features).to(DEV) for l in lins]
    for r, l in zip(ref, lins):
        r.load_state_dict(l.state_dict())
    h = x.reshape(m, 48)
    for i, r in enumerate(ref):
        h = r(h)
        if i < len(ref) - 1:
            h = torch.sigmoid(h) * masks[i]
    want_out = h.view(6, 50, 24)
    (want_out * w).sum().backward()
    want = [want_out.detach()] + [prm.grad for r in ref for prm in r.parameters()]
    for a, b in zip(got, want):
        assert float((a - b).abs().max() / b.abs().max()) < tol


def test_eval_mode_and_p_zero_are_todays_bits_and_training_calls_no_torch_dropout(monkeypatch):
    feats = data.to_device(synthetic.make_acoustic_batch(4, 60, streams=(('lf0', 3, 'mse'),), seed=9, with_raw=True), DEV)

    def build(p):
        torch.manual_seed(1)
        model = models.GRUF0Model(dropout_prob=p, precision='bf16', generate=False).to(DEV)
        own = model.state_dict()
        for k, v in synthetic.gru_f0_state().items():
            own[k].copy_(torch.from_numpy(v))
        return model

    base = build(0.0)
    base.train()
    loss0, out0 = base(feats)
    dropped = build(0.3)
    dropped.eval()
    loss_eval, out_eval = dropped(feats)
    assert torch.equal(loss_eval, loss0)                                # eval mode: the identity, bit for bit
    for k in out0:
        if torch.is_tensor(out0[k]):
            assert torch.equal(out_eval[k], out0[k]), k

    def no_torch_dropout(*a, **kw):
        raise AssertionError('torch dropout on the HIP path')
    monkeypatch.setattr(torch.nn.functional, 'dropout', no_torch_dropout)
    monkeypatch.setattr(torch, 'dropout', no_torch_dropout)
    dropped.train()
    calls = []
    _lib.CALL_LOG = calls
    try:
        loss_a, _ = dropped(feats)
        F_hip.backward(loss_a)
        loss_b, _ = dropped(feats)
    finally:
        _lib.CALL_LOG = None
    assert calls.count('mg_dropout') >= 5                               # five Dropout modules in the shipped F0 model
    assert torch.isfinite(loss_a) and loss_a.item() != loss0.item() and loss_b.item() != loss_a.item()   # a new mask every step
    for prm in dropped.parameters():
        assert prm.grad is not None and torch.isfinite(prm.grad).all()
    ops.check_persistent_status()


def test_graph_replay_draws_a_new_mask_every_replay():
    torch.manual_seed(5)
    x = torch.rand(128, 256, device=DEV) + 0.5
    static_y = torch.empty_like(x)
    F_hip.DropoutFn.apply(x, 0.5, 1)                                    # warm-up outside the capture (allocations, the counter)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_y.copy_(F_hip.DropoutFn.apply(x, 0.5, 1))
    masks = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        masks.append((static_y != 0).clone())
    assert not torch.equal(masks[0], masks[1]) and not torch.equal(masks[1], masks[2])
    for mk in masks:
        assert abs(float(mk.float().mean()) - 0.5) < 0.02


def test_graphed_step_with_warmup_zero_still_draws_new_masks(monkeypatch):
    """the dropout step counter must never be created inside a stream capture (every replay would reset it and repeat its
    masks).  ops.dropout_state refuses to; graphs.GraphedTrainStep creates it before its capture opens, so warmup=0 - the form
    GraphedStepCache uses - replays with a new mask per step."""
    from morgana_amd import graphs, optim
    monkeypatch.setattr(ops, '_dropout_state', {})                    # a process that has never drawn a mask
    graph = torch.cuda.CUDAGraph()
    x = torch.rand(64, 64, device=DEV)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='inside a stream capture'):
        with torch.cuda.graph(graph):
            F_hip.DropoutFn.apply(x, 0.5, 1)
    torch.cuda.synchronize()
    assert ops._dropout_state == {}

    torch.manual_seed(3)
    stack = utils.SequentialWithRecurrent(torch.nn.Linear(32, 64), torch.nn.Sigmoid(), torch.nn.Dropout(0.5), torch.nn.Linear(64, 8)).to(DEV)

    class _Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = stack
            self.seen = None

        def forward(self, feats):
            h, _ = self.layers(feats['x'])
            self.seen = h
            return (h * h).mean(), {'h': h}

    model = _Model().to(DEV).train()
    feats = {'x': torch.rand(4, 16, 32, device=DEV)}
    opt = optim.Adam(model.parameters(), lr=0.0)                       # lr 0: the weights stay, only the masks change the output
    step = graphs.GraphedTrainStep(model, opt, feats, warmup=0)
    outs = []
    for _ in range(3):
        step()
        torch.cuda.synchronize()
        outs.append(step.output['h'].detach().clone())
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
